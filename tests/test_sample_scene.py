"""configs[0]: the reference's own sample scene (45 entities of `space_logic`) through the CPU oracle (plumbing,
runs everywhere) and, on the GPU box, through the HIP path against the oracle.

The model AABBs come from tests/golden/sample_scene_models.json (tools/gen_sample_scene_models.py read the
reference's OBJ assets once; the fixture is data only)."""
import json
import os

import numpy as np
import pytest

import oracle as ro
from helpers import to_oracle, oracle_camera, assert_render_equal, expand_vis

HERE = os.path.dirname(os.path.abspath(__file__))


def load_models():
    m = json.load(open(os.path.join(HERE, "golden", "sample_scene_models.json")))["models"]
    m = {k: np.array(v, np.float32) for k, v in m.items()}
    m["_user"] = np.array([-5, 5, -5, 5, -5, 5], np.float32)                  # main.rs:35-41
    return m


def scene():
    from render_engine_amd import synthetic
    return synthetic.sample_scene(load_models()), synthetic.SAMPLE_SCENE_WORLD, synthetic.SAMPLE_SCENE_CAMERA


def test_sample_scene_through_the_oracle():
    ents, world, camd = scene()
    assert len(ents) == 45
    w = ro.World(world["outline_length"], world["atomic_length"])
    assert w.register(to_oracle(ents)) == 0
    # the user entity: identity matrix, +-5 box at the camera (flows/pipeline.rs:125-144)
    u = w.entity(0)
    np.testing.assert_array_equal(u["mat"], np.eye(4, dtype=np.float32).ravel())
    np.testing.assert_array_equal(u["aabb"], np.array([995, 1005, 995, 1005, 1145, 1155], np.float32))
    # a star: scale 10 about (950,1000,965) -> AABB of the scaled model box
    s = w.entity(1)
    m = load_models()["yellowStar"]
    np.testing.assert_allclose(s["aabb"], np.array([950 + 10 * m[0], 950 + 10 * m[1], 990, 1010, 955, 975], np.float32), rtol=0, atol=1e-3)
    cam = ro.make_camera(camd["position"], camd["direction"], camd["far"])
    vis = w.cull(cam)
    r = w.render(cam)
    # everything sits within 200 units of the camera inside the frustum looking down -z: every entity is drawn exactly once
    assert r["total"] == 45 and len(vis) > 0
    assert sorted(int(i) for i in r["ids"]) == list(range(45))
    per_model, per_sortable = {}, {}
    for g in r["groups"]:
        mi = int(g["model_index"]) & 0x1FFFFFF                               # LOD lives in bits 25.. (model_definitions.rs:31-59)
        per_model[mi] = per_model.get(mi, 0) + int(g["count"]); per_sortable[int(g["sortable"])] = per_sortable.get(int(g["sortable"]), 0) + int(g["count"])
    assert per_model == {6: 1, 0: 1, 1: 1, 2: 40, 3: 1, 4: 1}
    assert per_sortable == {0: 43, 1: 2}                                       # the two stars sit in sortable bucket 1
    assert len(w.shared_sections()) > 0                                        # asteroids straddle section borders
    # 10 frames of ticking: all 43 rotating bodies change every frame (their sections are visible), nothing leaves the world
    for _ in range(10):
        n, oob = w.tick(cam, 1.0 / 60.0)
        assert n == 43 and len(oob) == 0
        w.cull(cam); r2 = w.render(cam)
        assert r2["total"] == r["total"]
    w.close()


@pytest.mark.gpu
def test_sample_scene_gpu_parity():
    import render_engine_amd as R
    ents, world, camd = scene()
    p = R.Pipeline(world["outline_length"], world["atomic_length"])
    assert p.register_model_instances(ents) == 0
    w = ro.World(world["outline_length"], world["atomic_length"])
    assert w.register(to_oracle(ents)) == 0
    cam = R.Camera(camd["position"], camd["direction"], camd["far"])
    oc = oracle_camera(cam)
    C = R._capi
    for frame in range(8):
        vis_o = w.cull(oc)
        g = p.cull_and_pack(cam, emit_duplicates=(frame % 2 == 1))
        keys, mult = p.visible_sections()
        np.testing.assert_array_equal(expand_vis(keys, mult), vis_o)
        assert_render_equal(g, w.render(oc, emit_duplicates=(frame % 2 == 1)))
        n_o, oob_o = w.tick(oc, 1.0 / 60.0)
        t = p.tick(1.0 / 60.0)
        assert t["n_changed"] == n_o == 43 and t["n_out_of_bounds"] == len(oob_o) == 0
    for e in ents:
        o = w.entity(int(e["id"]))
        np.testing.assert_array_equal(p.read_component(int(e["id"]), C.C_TRANSFORMATION), o["mat"])
        np.testing.assert_array_equal(p.read_component(int(e["id"]), C.C_STATIC_AABB), o["aabb"])
    p.close(); w.close()
