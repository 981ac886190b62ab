"""Light sets of the world sections and the per-frame light query (world/bounding_box_tree_v2.rs:157-228; flows/shadow_flow.rs:455-513,
called from RenderFlow::render, flows/render_flow.rs:249-254, and upload_*_lights, render_system/render_system.rs:676-800).
CPU: the oracle against a brute-force restatement from the definitions (numpy, no shared code); GPU: re_visible_lights against the oracle."""
import numpy as np
import pytest

import oracle as ro
from helpers import to_oracle, oracle_camera

LIGHTS = (ro.F_LIGHT_DIRECTIONAL, ro.F_LIGHT_POINT, ro.F_LIGHT_SPOT)


def lit_world(R, n, seed, spread, atomic=64, frac=0.08):
    ents = R.synthetic.mixed_world(n, seed=seed, spread=spread, atomic=atomic)
    rng = np.random.default_rng(seed)
    pick = rng.random(n) < frac
    ents["flags"][pick] |= rng.choice(np.array(LIGHTS, np.uint32), size=int(pick.sum()))
    return ents


def brute_force(w, ents, campos, far, type_flag, atomic, outline=16384):
    """definition: a light is found when a unique world section holding it -- or, for an entity spanning several sections (a shared section), one of
    the sections it spans -- has a grid box that meets the cube [camera +- far] (closed intervals) and is a candidate of the whole-world query, i.e.
    lies inside the cube clamped to the positive octant, enumerated per level from floor(min / side) for ceil(extent / side) sections"""
    out = []
    campos = np.asarray(campos, np.float32); far = np.float32(far)
    lo, hi = campos - far, campos + far
    clo = np.maximum(lo, np.float32(0))
    for e in ents:
        if not (int(e["flags"]) & type_flag):
            continue
        kind, keys = w.lookup(int(e["id"]))              # entities_index_lookup: the unique section, or the sections a shared section links
        if kind == 0:
            continue
        hit = False
        for key in keys:
            lv, x, z, y = (key >> 48) & 0xFFFF, (key >> 32) & 0xFFFF, (key >> 16) & 0xFFFF, key & 0xFFFF
            side = np.float32(atomic * (1 << lv))
            idx = np.array([x, y, z], np.float32)
            base = np.floor(clo / side); cnt = np.ceil((hi - clo) / side)
            if np.any(idx < base) or np.any(idx >= base + cnt):
                continue
            gmin = idx * side; gmax = gmin + side
            if np.all(lo <= gmax) and np.all(hi >= gmin):
                hit = True
        if hit:
            out.append(int(e["id"]))
    return np.array(sorted(out), np.uint32)


CAMS = [((8192.0, 8192.0, 8500.0), (0.0, 0.0, -1.0), 300.0), ((8000.0, 8300.0, 8100.0), (0.3, 0.1, -1.0), 700.0),
        ((8800.0, 8000.0, 8200.0), (-1.0, 0.0, 0.2), 150.0), ((150.0, 120.0, 90.0), (1.0, 0.2, 0.3), 400.0)]


def test_oracle_light_query_against_brute_force():
    import render_engine_amd as R
    ents = lit_world(R, 1500, 17, 700.0)
    extra = lit_world(R, 200, 18, 150.0); extra["id"] += 100000
    for k in range(3):
        extra["pos"][:, k] = extra["pos"][:, k] - np.float32(8192.0) + np.float32(200.0)     # a cluster near the world's corner: the clamped candidate box matters
    ents = np.concatenate([ents, extra])
    w = ro.World(16384, 64)
    w.register(to_oracle(ents))
    total = 0
    for pos, d, far in CAMS:
        cam = R.Camera(pos, d, far); oc = oracle_camera(cam)
        for t in LIGHTS:
            got = w.visible_lights(oc, t)
            want = brute_force(w, ents, pos, far, t, 64)
            np.testing.assert_array_equal(got, want)
            total += len(got)
    assert total > 20
    w.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,atomic", [(5, 64), (6, 32)])
def test_visible_lights_parity(seed, atomic):
    """re_visible_lights against the oracle: unique and shared sections, several cameras, after ticks with movers (lights that change section),
    after deletes and make-static / wake-up requests"""
    import render_engine_amd as R
    from test_gpu_parity import random_changes
    ents = lit_world(R, 2500, seed, 600.0, atomic=atomic, frac=0.12)
    p = R.Pipeline(16384, atomic); w = ro.World(16384, atomic)
    assert p.register_model_instances(ents) == w.register(to_oracle(ents))
    rng = np.random.default_rng(seed)
    seen = 0
    for f, (pos, d, far) in enumerate(CAMS + CAMS[:2]):
        cam = R.Camera(pos, d, far); oc = oracle_camera(cam)
        for t in LIGHTS:
            got = p.visible_lights(cam, t); want = w.visible_lights(oc, t)
            np.testing.assert_array_equal(got, want, err_msg=f"frame {f} type {t:#x}")
            seen += len(got)
        w.cull(oc); p.cull_and_pack(cam)
        w.tick(oc, 0.05); p.tick(0.05)
        if f % 2 == 1:
            ch = random_changes(R, ents, rng, 80, set())
            p.apply_changes(ch); w.apply_changes(ch.view(ro.CHANGE_DT))
    assert seen > 30
    # capacity smaller than the result: n reports the full count
    import ctypes as C
    cam = R.Camera(*CAMS[1][:2], CAMS[1][2]); camc = cam.to_c(); n = C.c_uint32(); few = np.zeros(2, np.uint32)
    assert p._L.re_visible_lights(p._h, C.byref(camc), ro.F_LIGHT_POINT, few.ctypes.data, 2, C.byref(n)) == 0
    full = w.visible_lights(oracle_camera(cam), ro.F_LIGHT_POINT)
    assert n.value == len(full) and list(few[:min(2, len(full))]) == list(full[:2])
    assert p._L.re_visible_lights(p._h, C.byref(camc), 0x6000, few.ctypes.data, 2, C.byref(n)) != 0     # two type bits: refused
    st = p.stats(); assert st["n_seal_waits"] == 0 and st["n_sync_fallbacks"] == 0
    p.close(); w.close()
