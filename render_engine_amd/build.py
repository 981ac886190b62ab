"""Builds librender_engine_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

hipcc cross-compiles without a GPU; the .so travels with the source tree to the GPU box.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "librender_engine_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
SOURCES = ["re_kernels.hip", "re_rebucket.hip", "re_api.hip", "re_lighting.hip", "re_collide.hip", "re_sort.hip", "re_history.cpp"]
HEADERS = ["re_kernels.h", "re_math.h", "re_guard.h", os.path.join(INCLUDE, "re_hip.h")]
# -ffp-contract=off: the visible set must be bit-exact with the reference's Rust arithmetic (no FMA contraction)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-fgpu-rdc", "-Wall", "-Wno-unused-result",
         # gfx950 hands the first kernel-argument dwords to each wave in SGPRs at launch: k_scan_cull's key pointer and count
         # arrive without a scalar-load round trip in front of the key loads
         "-mllvm", "-amdgpu-kernarg-preload-count=14"]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # (RE_BUILD_DEFINES="-DRE_EXP_STAGES": development builds for tools/stage_stop.py, tools/timeline.py)
    cmd = [hipcc] + FLAGS + os.environ.get("RE_BUILD_DEFINES", "").split() + ["-I", INCLUDE, "-I", CSRC] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
