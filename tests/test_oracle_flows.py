"""CPU-only checks of the oracle's frame flows and of its arithmetic against an independent
numpy-float32 mirror (the nalgebra-backed formulas have no reference tests: 'parity unpinned')."""
import numpy as np
import pytest

import oracle as ro
from render_engine_amd import synthetic, Camera
from helpers import to_oracle, oracle_camera

f32 = np.float32


def np_norm3(x, y, z):
    return np.sqrt((x * x + y * y) + z * z, dtype=f32)


def test_sincos_close_to_libm_and_symmetric():
    xs = np.concatenate([np.linspace(-50, 50, 20001), np.array([0.0, 1e-8, 1e5, -1e5, 3e6, 1e9])]).astype(f32)
    for x in xs[::7]:
        s, c = ro.sincos(x)
        rs, rc = np.sin(np.float64(x)), np.cos(np.float64(x))
        assert abs(float(s) - rs) <= 1.2e-7 * max(1.0, abs(rs)) + 1e-9, x
        assert abs(float(c) - rc) <= 1.2e-7 * max(1.0, abs(rc)) + 1e-9, x
    assert ro.sincos(f32(0.0)) == (f32(0.0), f32(1.0))
    s, c = ro.sincos(f32(np.inf))
    assert np.isnan(s) and np.isnan(c)


def test_sincos_is_correctly_rounded_almost_always():
    rng = np.random.default_rng(5)
    xs = (rng.uniform(-20, 20, 4000)).astype(f32)
    bad = 0
    for x in xs:
        s, c = ro.sincos(x)
        bad += (s != f32(np.sin(np.float64(x)))) + (c != f32(np.cos(np.float64(x))))
    assert bad <= 4          # f64 evaluation rounded once: differs from correct rounding only at near-ties


def test_planes_match_numpy_mirror():
    cam = Camera((8192, 8192, 8192), (0, 0, -1), 1000.0)
    got = ro.make_planes(cam.projection_view)
    pv = cam.projection_view.reshape(4, 4)           # [col][row]
    rows = pv.T                                       # rows of P*V
    exp = np.zeros((6, 4), f32)
    raw = [rows[3] + rows[0], rows[3] - rows[0], rows[3] + rows[1], rows[3] - rows[1], rows[3] - f32(0), rows[3] - rows[2]]
    for k, p in enumerate(raw):
        exp[k] = p / np_norm3(p[0], p[1], p[2])
    np.testing.assert_array_equal(got, exp)


def test_trs_and_aabb_match_numpy_mirror():
    rng = np.random.default_rng(1)
    for _ in range(200):
        pos = rng.uniform(0, 16000, 3).astype(f32); axis = rng.uniform(-1, 1, 3).astype(f32); ang = f32(rng.uniform(-7, 7))
        scl = rng.uniform(0.2, 3, 3).astype(f32)
        m = ro.trs_matrix(pos, axis, ang, scl).reshape(4, 4)            # [col][row]
        n = np_norm3(axis[0], axis[1], axis[2]); ux, uy, uz = axis / n
        s, c = ro.sincos(ang); omc = f32(1) - c
        R = np.array([[ux * ux + (f32(1) - ux * ux) * c, ux * uy * omc - uz * s, ux * uz * omc + uy * s],
                      [ux * uy * omc + uz * s, uy * uy + (f32(1) - uy * uy) * c, uy * uz * omc - ux * s],
                      [ux * uz * omc - uy * s, uy * uz * omc + ux * s, uz * uz + (f32(1) - uz * uz) * c]], f32)   # [row][col]
        exp = np.zeros((4, 4), f32)
        for col in range(3):
            for row in range(3):
                exp[col, row] = R[row, col] * scl[col]
        exp[3, :3] = pos; exp[3, 3] = 1
        np.testing.assert_allclose(m, exp, rtol=0, atol=0)
        box = ro.aabb(np.array([-1, 2, -3, 1, -2, 5], f32))
        out = ro.lib().ro_apply_transformation(box, ro._fp(np.ascontiguousarray(m.reshape(16))))
        c0 = np.array([-1, -3, -2, 1], f32); c1 = np.array([2, 1, 5, 1], f32)

        def mv(v):
            r = np.zeros(4, f32)
            for i in range(4):
                y = m[0, i] * v[0]; y = m[1, i] * v[1] + y; y = m[2, i] * v[2] + y; y = m[3, i] * v[3] + y
                r[i] = y
            return r
        a, b = mv(c0), mv(c1)
        assert out.tup() == (min(a[0], b[0]), max(a[0], b[0]), min(a[1], b[1]), max(a[1], b[1]), min(a[2], b[2]), max(a[2], b[2]))


def test_frustum_logic_distance_against_mirror():
    rng = np.random.default_rng(2)
    cam = Camera((1000, 1000, 1150), (0, 0, -1), 1000.0)
    planes = ro.make_planes(cam.projection_view)
    L = ro.lib()
    for _ in range(500):
        lo = rng.uniform(0, 2200, 3).astype(f32); sz = rng.choice([64, 128, 256, 512]).astype(f32)
        box = np.array([lo[0], lo[0] + sz, lo[1], lo[1] + sz, lo[2], lo[2] + sz], f32)
        vis = True
        for k in range(6):
            any_in = False
            for x in box[0:2]:
                for y in box[2:4]:
                    for z in box[4:6]:
                        d = planes[k, 0] * x + planes[k, 1] * y + planes[k, 2] * z + planes[k, 3]
                        any_in |= not (d < 0)
            vis &= any_in
        assert bool(L.ro_frustum_aabb_visible(ro._fp(np.ascontiguousarray(planes.reshape(24))), ro.aabb(box))) == vis
        best = min(np_norm3(x - cam.position[0], y - cam.position[1], z - cam.position[2]) for x in box[0:2] for y in box[2:4] for z in box[4:6])
        assert bool(L.ro_logic_aabb_in_view(f32(64), ro._fp(cam.position), ro.aabb(box))) == bool(best <= f32(64))
        h = max(box[1] - box[0], box[3] - box[2], box[5] - box[4]) / f32(2)
        rad = np.sqrt((h * h) * f32(3), dtype=f32)
        ctr = [(box[0] + box[1]) / f32(2), (box[2] + box[3]) / f32(2), (box[4] + box[5]) / f32(2)]
        exp = max(np_norm3(cam.position[0] - ctr[0], cam.position[1] - ctr[1], cam.position[2] - ctr[2]) - rad, f32(0))
        assert L.ro_distance_to_aabb(ro.aabb(box), ro._fp(cam.position)) == exp


def test_lod_bands():
    lo, hi = ro.default_lod(1000.0)
    L = ro.lib()
    assert L.ro_lod_adjusted_model_index(5, f32(0.0), 5, ro._fp(lo), ro._fp(hi)) == 5
    assert L.ro_lod_adjusted_model_index(5, f32(150.0), 5, ro._fp(lo), ro._fp(hi)) == 5 | (1 << 25)
    assert L.ro_lod_adjusted_model_index(5, f32(100.0), 5, ro._fp(lo), ro._fp(hi)) == 5           # first band wins on the shared edge
    assert L.ro_lod_adjusted_model_index(5, f32(5000.0), 5, ro._fp(lo), ro._fp(hi)) == 5 | (7 << 25)
    assert L.ro_lod_adjusted_model_index(5, f32(np.nan), 5, ro._fp(lo), ro._fp(hi)) == 5 | (7 << 25)


def run_frames(w, cams, dt=0.016):
    out = []
    for cam in cams:
        oc = oracle_camera(cam)
        vis = w.cull(oc)
        r = w.render(oc)
        rd = w.render(oc, emit_duplicates=True)
        n, oob = w.tick(oc, dt)
        out.append((vis, r, rd, n, oob))
    return out


def test_mixed_world_frames_invariants():
    ents = synthetic.mixed_world(3000)
    w = ro.World(16384, 64)
    assert w.register(to_oracle(ents)) == 0
    cams = [Camera((8192 + 40 * i, 8192, 8192 + 300 - 60 * i), (0, 0, -1), 1000.0) for i in range(6)]
    res = run_frames(w, cams)
    ids = set(ents["id"].tolist())
    assert res[0][1]["total"] > 100
    for vis, r, rd, n, oob in res:
        assert set(r["ids"].tolist()) <= ids
        assert len(np.unique(r["ids"])) == len(r["ids"])               # set semantics: each instance once
        assert rd["total"] >= r["total"] and set(rd["ids"].tolist()) == set(r["ids"].tolist())
        assert len(np.unique(vis)) <= len(vis)
    assert sum(x[3] for x in res) > 0                                  # something moved
    w.close()


def test_static_cache_first_sight_quirk():
    """A static section out of draw distance when first cached stays invisible (render_flow.rs:549-594,749-754)."""
    ents = synthetic.lattice_world(cells_per_axis=12, first_cell=120)
    w = ro.World(16384, 64)
    w.register(to_oracle(ents))
    far_cam = oracle_camera(Camera((100, 100, 100), (0, 0, -1), 300.0))
    near_cam = oracle_camera(Camera((8000, 8000, 8300), (0, 0, -1), 1000.0))
    w.cull(far_cam); assert w.render(far_cam)["total"] == 0
    w.tick(far_cam, 0.016)
    w.cull(near_cam)
    assert len(w.cull(near_cam)) > 0 and w.render(near_cam)["total"] == 0      # cached empty at first sight
    w2 = ro.World(16384, 64); w2.register(to_oracle(ents))
    w2.cull(near_cam); assert w2.render(near_cam)["total"] > 0
    w.close(); w2.close()


def _ent(i, pos, half, flags, vel=(0, 0, 0)):
    e = np.zeros(1, ro.ENTITY_DT)[0]
    e["id"] = i; e["flags"] = flags
    e["original"] = (-half, half, -half, half, -half, half)
    e["pos"] = pos; e["scale"] = (1, 1, 1); e["rot_axis"] = (1, 0, 0); e["vel"] = vel
    return e


def test_collision_pass_hand_case():
    """handle_collisions (flows/logic_flow.rs:452-651) on a case worked out by hand.  One level-0 world section next to the camera
    (listed twice in visible_sections_vec: logic box and frustum, so its moved entities are pushed twice, :214-223, 443-446):
      1 mover, large, CanCauseCollisions         -- the moved entity
      2 at rest (no Velocity), touches 1         -- (1,2) and (2,1): both collision functions run (:640-647)
      3 mover with CanCauseCollisions, touches 1 -- (1,3) from 1's pass and (3,1) from 3's pass, each only-to-self (:625-637)
      4 static, touches 1                        -- static_entities are not searched (find_related_entities returns local_entities)
      5 at rest, apart                           -- no overlap
      6 mover WITHOUT CanCauseCollisions, touches 1 -- not a moved entity: treated like 2
      7 at rest, touches 1 only along a face (closed intervals, range.rs:71)
    """
    F = ro
    w = ro.World(16384, 64)
    mv = F.F_HAS_VEL | F.F_CAN_COLLIDE
    ents = np.array([
        _ent(1, (8210, 8210, 8210), 5.0, mv, (1, 0, 0)),
        _ent(2, (8216, 8210, 8210), 2.0, 0),
        _ent(3, (8204, 8210, 8210), 2.0, mv, (0, 1, 0)),
        _ent(4, (8210, 8216, 8210), 2.0, F.F_STATIC),
        _ent(5, (8240, 8240, 8240), 2.0, 0),
        _ent(6, (8210, 8204, 8210), 2.0, F.F_HAS_VEL, (0, 0, 1)),
        _ent(7, (8210, 8210, 8217), 2.0, 0),
    ], ro.ENTITY_DT)
    assert w.register(ents) == 0
    cam = oracle_camera(Camera((8210, 8210, 8290), (0, 0, -1), 1000.0))
    vis = w.cull(cam)
    key = ro.pack_key(0, 8210 // 64, 8210 // 64, 8210 // 64)
    assert (vis == key).sum() == 2
    pairs = sorted(map(tuple, w.collide(cam).tolist()))
    once = [(1, 2), (2, 1), (1, 3), (3, 1), (1, 6), (6, 1), (1, 7), (7, 1)]
    assert pairs == sorted(once * 2)
    # farther than 200 units from the section: nothing is tested (:553-558), although the section is still visible
    cam_far = oracle_camera(Camera((8210, 8210, 8210 + 64 + 260), (0, 0, -1), 1000.0))
    assert key in set(w.cull(cam_far).tolist())
    assert len(w.collide(cam_far)) == 0
    w.close()


def test_collision_shared_first_touch_and_related_sections():
    """a moved entity stored under a Shared lookup that is the first to touch a world section creates the section's entry WITHOUT
    being pushed into it (logic_flow.rs:488-498), so alone it collides with nothing; a second moved entity in the same sections
    then does.  A large entity one level up is found through related_world_sections (the parent section)."""
    F = ro
    mv = F.F_HAS_VEL | F.F_CAN_COLLIDE
    base = [
        _ent(10, (8256, 8210, 8210), 4.0, mv, (1, 0, 0)),          # straddles x = 8256: shared section of two level-0 sections
        _ent(11, (8250, 8210, 8210), 3.0, 0),                        # at rest in the left section, touches 10
        _ent(12, (8256, 8256, 8256), 50.0, 0),                       # level-1 section (parent of both), touches 10
    ]
    w = ro.World(16384, 64)
    assert w.register(np.array(base, ro.ENTITY_DT)) == 0
    assert w.lookup(10)[0] == 2 and w.lookup(12)[0] == 1 and ro.unpack_key(w.lookup(12)[1][0])[0] == 1
    cam = oracle_camera(Camera((8240, 8210, 8290), (0, 0, -1), 1000.0))
    w.cull(cam)
    assert len(w.collide(cam)) == 0                                  # 10 created both entries and is in neither
    w.close()
    w = ro.World(16384, 64)
    assert w.register(np.array(base + [_ent(20, (8256, 8212, 8212), 3.0, mv, (0, 1, 0))], ro.ENTITY_DT)) == 0   # same shared section, larger id
    w.cull(cam)
    pairs = sorted(map(tuple, w.collide(cam).tolist()))
    # 20 is pushed into both sections' entries (10 created them): per section, 20 against 10 (moved: only-to-self), 11 and 12 (at rest: both ways)
    per_section = [(20, 10), (20, 11), (11, 20), (20, 12), (12, 20)]
    assert pairs == sorted(per_section * 2)
    w.close()


@pytest.mark.parametrize("threads", [1, 4])
def test_optimised_cpu_row_matches_the_oracle(threads):
    """the "optimised CPU" baseline of bench.py (oracle/re_cpu_soa.c: sorted keys, SoA, OpenMP) draws exactly what the port of the
    reference draws, frame after frame (incl. the frozen static cache: sections beyond the draw distance at the first frame stay empty)"""
    ents = synthetic.lattice_world(cells_per_axis=36, first_cell=110)
    w = ro.World(16384, 64); assert w.register(to_oracle(ents)) == 0
    s = ro.SoaWorld(w, to_oracle(ents), threads=threads)
    from helpers import assert_render_equal
    for pos, d, far in [((8192, 8192, 8300), (0, 0, -1), 700.0), ((7800.5, 8100.25, 9000), (0.6, 0.0, -0.8), 1500.0), ((8192, 8192, 8192), (0.3, 0.2, -1), 3000.0)]:
        cam = oracle_camera(Camera(pos, d, far))
        vis = w.cull(cam)
        o = w.render(cam)
        g = s.frame(cam, cap=o["total"] + 8)
        assert g["n_visible_vec"] == len(vis)
        assert_render_equal(g, o)
        order_o = {int(i): m for i, m in zip(o["ids"], o["mats"])}
        for i, m in zip(g["ids"], g["mats"]):
            assert (order_o[int(i)] == m).all()
        w.tick(cam, 0.016)                                     # the logic phase of the frame clears changed_static_unique (pipeline.rs:271)
    s.close(); w.close()
