"""The C++ host mirror of the reference's interface (include/render_engine_hip.hpp): compiles with g++ against the C ABI,
its host-side pieces agree with the oracle (CPU), and the sample scene driven through it is bit-exact on the GPU."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BIN = os.path.join(HERE, "cpp", "_build", "sample_scene_test")


def build_binary():
    import oracle
    from render_engine_amd import build as libbuild
    oracle.build()
    lib_dir = os.path.dirname(libbuild.build_library())
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    oracle_dir = os.path.join(ROOT, "oracle")
    cmd = ["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", oracle_dir,
           os.path.join(HERE, "cpp", "sample_scene_test.cpp"), "-o", BIN,
           "-L", lib_dir, "-lrender_engine_hip", "-L", oracle_dir, "-l:libre_oracle.so", "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + oracle_dir]
    subprocess.check_call(cmd)
    return BIN


def write_scene(path):
    from test_sample_scene import scene
    import render_engine_amd as R
    ents, _, _ = scene()
    with open(path, "w") as f:
        for e in ents:
            kind = 0 if int(e["flags"]) & R.F_USER else (1 if int(e["flags"]) & R.F_HAS_ROTVEL else 2)
            vals = [int(e["model_index"]), int(e["sortable"]), kind, *e["pos"], e["scale"][0], *e["rot_axis"], e["rot_angle"], *e["rotvel_axis"], e["rotvel"], *e["original"]]
            f.write(" ".join(repr(float(v)) if i >= 3 else str(v) for i, v in enumerate(vals)) + "\n")


def test_cpp_mirror_builds_and_host_pieces_match_the_oracle(tmp_path):
    exe = build_binary()
    scene_file = tmp_path / "scene.txt"
    write_scene(scene_file)
    out = subprocess.run([exe, str(scene_file), "--host-only"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_mirror_sample_scene_gpu(tmp_path):
    exe = build_binary()
    scene_file = tmp_path / "scene.txt"
    write_scene(scene_file)
    out = subprocess.run([exe, str(scene_file)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bit-exact" in out.stdout
