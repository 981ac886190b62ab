"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bar: visible sections, visible entity-ID sets, group tables and 4x4 matrices BIT-EXACT
(north_star asks 1e-5 abs on matrices; the shared deterministic sin/cos makes them exact)."""
import numpy as np
import pytest

import oracle as ro
from helpers import to_oracle, oracle_camera, assert_render_equal, expand_vis

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    import render_engine_amd as R
    return R


def build_pair(R, ents, outline=16384, atomic=64, flags=0, max_instances=0):
    p = R.Pipeline(outline, atomic, flags=flags, max_instances=max_instances)
    plain_close = p.close

    def close_checked():                      # every test that closes its pipeline also asserts the publication counters
        if getattr(p, "_h", None):
            assert_clean_publication(p)
        plain_close()
    p.close = close_checked
    rej = p.register_model_instances(ents)
    w = ro.World(outline, atomic)
    rej_o = w.register(to_oracle(ents))
    assert rej == rej_o
    return p, w


def check_sections(p, w):
    s, c = p.sections(), w.cells()
    np.testing.assert_array_equal(s["keys"], c["keys"])
    np.testing.assert_array_equal(s["n_local"], c["n_local"])
    np.testing.assert_array_equal(s["n_static"], c["n_static"])
    np.testing.assert_array_equal(s["is_static_section"] != 0, c["is_static_section"] != 0)
    tight = np.stack([c["tight"][k] for k in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")], axis=1)
    np.testing.assert_array_equal(s["tight"], tight)
    gs, cs = p.shared_sections(), w.shared_sections()                  # shared world sections: ids, AABB of the last member, members (re_debug_get_shared_sections also
    assert [g["keys"] for g in gs] == [o["keys"] for o in cs]          # checks the library's host mirrors against its device table)
    for g, o in zip(gs, cs):
        np.testing.assert_array_equal(g["active"], np.sort(o["active"]), err_msg=f"active members of shared section {o['keys']}")
        np.testing.assert_array_equal(g["static"], np.sort(o["static"]), err_msg=f"static members of shared section {o['keys']}")
        assert g["aabb"] == tuple(np.float32(v) for v in o["aabb"]), (o["keys"], g["aabb"], o["aabb"])


def check_frame(R, p, w, cam, dups, force_large_pack=False):
    oc = oracle_camera(cam)
    vis_o = w.cull(oc)
    g = p.cull_and_pack(cam, emit_duplicates=dups, force_large_pack=force_large_pack)
    keys, mult = p.visible_sections()
    np.testing.assert_array_equal(expand_vis(keys, mult), vis_o)
    assert g["n_visible_vec"] == len(vis_o) and g["n_visible_sections"] == len(np.unique(vis_o))
    o = w.render(oc, emit_duplicates=dups)
    assert_render_equal(g, o)
    assert_clean_publication(p)
    return g, o


def assert_clean_publication(p):
    """every result block the host polled in mapped memory (frame result + InstanceRange table, tick counters, collision header) agreed
    with its seal at first sight: the slow paths behind the publication protocol (re_kernels.h: publish_to_host) were never taken"""
    st = p.stats()
    assert st["n_seal_waits"] == 0 and st["n_sync_fallbacks"] == 0, st


def check_entities(R, p, w, ents):
    C = R._capi
    for e in ents:
        eid = int(e["id"])
        o = w.entity(eid)
        fl = int(p.read_component(eid, C.C_FLAGS)[0])
        if o is None:
            assert fl & 0x80000000
            continue
        np.testing.assert_array_equal(p.read_component(eid, C.C_TRANSFORMATION), o["mat"], err_msg=f"mat of {eid}")
        np.testing.assert_array_equal(p.read_component(eid, C.C_STATIC_AABB), o["aabb"])
        np.testing.assert_array_equal(p.read_component(eid, C.C_POSITION), o["pos"])
        assert (fl & (C.F_HAS_MOVED | C.F_HAS_ROTATED)) == (o["flags"] & (C.F_HAS_MOVED | C.F_HAS_ROTATED)), eid
        # components an entity does not carry read as None (ECS::get_copy): the presence bits agree with the oracle's, the values where present
        present = C.F_HAS_ROT | C.F_HAS_SCALE | C.F_HAS_VEL | C.F_HAS_ACC | C.F_HAS_ROTVEL | C.F_HAS_ROTACC
        assert (fl & present) == (o["flags"] & present), (eid, hex(fl), hex(o["flags"]))
        for bit, comp, key in ((C.F_HAS_ROT, C.C_ROTATION, "rot"), (C.F_HAS_ROTVEL, C.C_ROTATION_VEL, "rotvel"), (C.F_HAS_VEL, C.C_VELOCITY, "vel")):
            if fl & bit:
                np.testing.assert_array_equal(p.read_component(eid, comp), o[key])
            else:
                with pytest.raises(R.RenderEngineError):
                    p.read_component(eid, comp)


def test_library_loaded_and_fails_loudly(R):
    assert R._capi.load().re_abi_version() == 3
    with pytest.raises(R.RenderEngineError):
        R.Pipeline(16384, 64, device=99)


def test_empty_world(R):
    p = R.Pipeline()
    assert p.register_model_instances(np.zeros(0, R.ENTITY_DT)) == 0
    g = p.cull_and_pack(R.Camera((100, 100, 100), (0, 0, -1), 1000.0))
    assert g["total"] == 0 and len(g["groups"]) == 0 and g["n_visible_sections"] == 0
    p.close()


@pytest.mark.parametrize("far", [1000.0, 3000.0])
def test_static_lattice(R, far):
    """config 2 at reduced size: all static, one entity per level-0 section"""
    ents = R.synthetic.lattice_world(cells_per_axis=40, first_cell=108)
    p, w = build_pair(R, ents)
    check_sections(p, w)
    for pos, d in [((8192, 8192, 8192), (0, 0, -1)), ((7500.5, 8100.25, 9000), (0.6, 0.0, -0.8)), ((8192, 8192, 8192), (0, 1, 0))]:
        for dups in (False, True):
            g, o = check_frame(R, p, w, R.Camera(pos, d, far), dups)
    assert g["total"] > 0
    p.close(); w.close()


def test_static_cache_first_sight_quirk(R):
    ents = R.synthetic.lattice_world(cells_per_axis=12, first_cell=120)
    p, w = build_pair(R, ents)
    far_cam, near_cam = R.Camera((100, 100, 100), (0, 0, -1), 300.0), R.Camera((8000, 8000, 8300), (0, 0, -1), 1000.0)
    g, o = check_frame(R, p, w, far_cam, False)
    assert g["total"] == 0
    w.tick(oracle_camera(far_cam), 0.016); p.tick(0.016)
    g, o = check_frame(R, p, w, near_cam, False)
    assert g["total"] == 0 and g["n_visible_sections"] > 0
    p.close(); w.close()


def test_mixed_world_multi_frame(R):
    """unique + shared sections, static + active, spinners, movers that stay in their section or not"""
    ents = R.synthetic.mixed_world(4000)
    p, w = build_pair(R, ents)
    check_sections(p, w)
    s = p.stats()
    assert s["n_shared_sections"] == w.L.ro_num_shared(w.h) and s["n_shared_sections"] > 10
    cam = R.Camera((8192, 8192, 8500), (0, 0, -1), 1000.0)
    for dups in (False, True):
        check_frame(R, p, w, cam, dups)
    check_entities(R, p, w, ents[:400])


def test_spinners_tick_parity(R):
    """config 3 at reduced size: rotating bodies stay inside their sections; several frames"""
    ents = R.synthetic.lattice_world(cells_per_axis=32, first_cell=112, spinner_every=7)
    p, w = build_pair(R, ents)
    cams = [R.Camera((8192 + 30 * i, 8192, 8300 - 25 * i), (0, 0, -1), 1000.0) for i in range(5)]
    total_changed = 0
    for cam in cams:
        g, o = check_frame(R, p, w, cam, False)
        n_o, oob_o = w.tick(oracle_camera(cam), 0.016)
        t = p.tick(0.016)
        assert t["n_changed"] == n_o and t["n_rebucket"] == 0 and t["n_out_of_bounds"] == len(oob_o) == 0
        total_changed += n_o
    assert total_changed > 0
    spin = ents[(ents["flags"] & R.F_HAS_ROTVEL) != 0]
    check_entities(R, p, w, spin[:300])
    check_sections(p, w)            # a body that turns inside its section still changes the section's tight AABB (add_entity + end_of_changes)
    # a tick with dt == 0 mirrors the reference's assert
    with pytest.raises(R.RenderEngineError):
        p.tick(0.0)
    p.close(); w.close()


def test_kinematics_one_tick_all_branches(R):
    """every apply_kinematics branch (acc/vel/rotacc/rotvel, zero velocity, always-execute) for one tick"""
    ents = R.synthetic.mixed_world(2500, seed=99, spread=500.0)
    p, w = build_pair(R, ents)
    cam = R.Camera((8192, 8192, 8400), (0, 0, -1), 1000.0)
    check_frame(R, p, w, cam, False)
    n_o, oob_o = w.tick(oracle_camera(cam), 0.016)
    t = p.tick(0.016)
    assert t["n_changed"] == n_o and n_o > 50
    assert t["n_out_of_bounds"] == len(oob_o)
    check_entities(R, p, w, ents)
    p.close(); w.close()


def test_pack_paths_agree(R):
    """small pack (k_pack_small), multi-kernel pack, and the overflow hand-over between them"""
    ents = R.synthetic.mixed_world(6000, seed=7, spread=700.0)
    p, w = build_pair(R, ents)
    cam_small = R.Camera((8192, 8192, 9300), (0, 0, -1), 700.0)
    cam_big = R.Camera((8192, 8192, 9300), (0, 0, -1), 4000.0)
    g1, _ = check_frame(R, p, w, cam_small, False)                       # small path
    g2, _ = check_frame(R, p, w, cam_small, False, force_large_pack=True)
    assert g1["total"] == g2["total"] > 0
    w.tick(oracle_camera(cam_small), 0.016); p.tick(0.016)
    big = R.synthetic.lattice_world(cells_per_axis=48, first_cell=104)   # > 16384 visible instances: the fused pack must hand over
    p2, w2 = build_pair(R, big)
    g3, _ = check_frame(R, p2, w2, R.Camera((8192, 8192, 11000), (0, 0, -1), 6000.0), False)
    assert g3["total"] > 16384
    g4, _ = check_frame(R, p2, w2, R.Camera((8192, 8192, 11000), (0, 0, -1), 6000.0), True)   # predicted large now
    assert g4["total"] >= g3["total"]
    g5, _ = check_frame(R, p2, w2, R.Camera((8192, 8192, 8192), (0, 0, -1), 500.0), False)    # and back to a small set
    for q in (p, p2):
        q.close()
    w.close(); w2.close()


def test_large_pack_paths_by_group_table_size(R):
    """the pack of a large visible set: group tables of <= 512 slots take the one-launch pack (k_pack_large; the scan counts the instances per
    group while it expands them, or k_emit_count_sharded does when the scan did not expect a large set), larger tables the count / scan /
    scatter kernels.  Movers and spinners in between: counted frames, cancelled / replayed frames and uncounted frames alternate."""
    for n_models, expect_slots in ((8, 64), (100, 800)):
        ents = R.synthetic.lattice_world(cells_per_axis=30, first_cell=113, n_models=n_models, spinner_every=7, mover_every=11)
        p, w = build_pair(R, ents)
        assert len(np.unique(ents["model_index"])) * 8 == expect_slots
        cam = R.Camera((8192, 8192, 9300), (0, 0, -1), 4000.0)
        for f in range(6):
            forced = f % 3 != 2
            g, o = check_frame(R, p, w, cam, bool(f & 1), force_large_pack=forced)
            n_o, oob_o = w.tick(oracle_camera(cam), 0.05); t = p.tick(0.05)
            assert t["n_changed"] == n_o
        assert g["total"] > 8192                                   # (2 x total > 16383: the unforced frames take the predicted-large path as well)
        # asynchronous frames through the large pack, speculation across ticks included
        for f in range(4):
            p.cull_and_pack(cam, asynchronous=True, copy=False, force_large_pack=True); p.tick(0.05, asynchronous=True)
            oc = oracle_camera(cam); w.cull(oc); w.render(oc); w.tick(oc, 0.05)
        check_frame(R, p, w, cam, False, force_large_pack=True)
        p.close(); w.close()


def test_per_model_level_of_view_bands(R):
    """level_views.custom (render_flow.rs:495-499, 889-893): models registered with custom bands take their LOD from those, the others from the camera's
    default bands -- active entities, the static cache, shared sections, both pack paths; bands that leave gaps fall back to LOD 7"""
    ents = R.synthetic.mixed_world(5000, seed=5, spread=600.0)
    p, w = build_pair(R, ents)
    bands = {(1, 0): ([0, 150, 300], [150, 300, 450]),                     # gaps beyond 450 -> LOD 7
             (3, 1): ([0, 40, 90, 200, 420, 650, 800, 900], [40, 90, 200, 420, 650, 800, 900, 5000]),
             (4, 0): ([100], [250])}                                        # a single band that most distances miss
    for (m, rs), (lo, hi) in bands.items():
        p.set_model_lod(m, rs, lo, hi); w.set_model_lod(m, rs, lo, hi)
    cams = [R.Camera((8192, 8192, 9000), (0, 0, -1), 1500.0), R.Camera((8000, 8300, 8700), (0.3, -0.1, -1), 900.0)]
    for f, cam in enumerate(cams * 2):
        g, o = check_frame(R, p, w, cam, bool(f & 1), force_large_pack=f >= 2)
        n_o, oob = w.tick(oracle_camera(cam), 0.03); t = p.tick(0.03)
        assert t["n_changed"] == n_o
    lods = {(int(m) & 0x1FFFFFF, int(rs)): set() for m, rs in zip(g["groups"]["model_index"], g["groups"]["render_system"])}
    for m, rs in zip(g["groups"]["model_index"], g["groups"]["render_system"]):
        lods[(int(m) & 0x1FFFFFF, int(rs))].add(int(m) >> 25)
    assert 7 in lods[(1, 0)] and max(lods[(3, 1)]) >= 5                      # the custom tables were in effect
    p.set_model_lod(1, 0, [], []); w.set_model_lod(1, 0, [], [])             # removing a model's bands returns it to the default ones
    check_frame(R, p, w, cams[0], False)
    p.close(); w.close()


def test_async_frames_match_sync(R):
    ents = R.synthetic.lattice_world(cells_per_axis=32, first_cell=112, spinner_every=5)
    p, w = build_pair(R, ents)
    cams = [R.Camera((8192 + 11 * i, 8192, 8350 - 9 * i), (0, 0, -1), 1000.0) for i in range(6)]
    for cam in cams:                                                     # async: enqueue everything, wait once
        p.cull_and_pack(cam, asynchronous=True)
        p.tick(0.016, asynchronous=True)
        oc = oracle_camera(cam); w.cull(oc); o = w.render(oc); n_o, _ = w.tick(oc, 0.016)
    vis, tick = p.wait(copy=True)
    assert_render_equal(vis, o)
    assert tick["n_changed"] == n_o
    spin = ents[(ents["flags"] & R.F_HAS_ROTVEL) != 0]
    check_entities(R, p, w, spin[:200])
    p.close(); w.close()


@pytest.mark.parametrize("full_rebuild", [False, True])
def test_movers_rebucket_parity(R, full_rebuild):
    """entities that leave their world section (unique <-> unique, unique <-> shared, new and emptied sections): the section table,
    tight AABBs (stale ones included), static flags, visible set and matrices must follow the reference's apply_change semantics"""
    ents = R.synthetic.mixed_world(3000, seed=21, spread=600.0)
    ents["vel"] *= 12.0                                              # fast movers: many section changes per tick
    p, w = build_pair(R, ents, flags=R._capi.CFG_FULL_REBUILD if full_rebuild else 0)   # in-place patches, and the from-scratch fallback
    check_sections(p, w)
    cams = [R.Camera((8192 + 25 * i, 8192 - 10 * i, 8600 - 30 * i), (0, 0, -1), 1500.0) for i in range(6)]
    moved = 0
    for cam in cams:
        check_frame(R, p, w, cam, False)
        n_o, oob_o = w.tick(oracle_camera(cam), 0.05)
        t = p.tick(0.05)
        assert t["n_changed"] == n_o and t["n_out_of_bounds"] == len(oob_o)
        moved += t["n_rebucket"]
        check_sections(p, w)
        s = p.stats()
        assert s["n_shared_sections"] == w.L.ro_num_shared(w.h)
    assert moved > 50
    check_entities(R, p, w, ents[:500])
    p.close(); w.close()


def test_truncation_reports(R):
    ents = R.synthetic.lattice_world(cells_per_axis=24, first_cell=116)
    p = R.Pipeline(16384, 64, max_instances=100)
    p.register_model_instances(ents)
    g = p.cull_and_pack(R.Camera((8192, 8192, 8192), (0, 0, -1), 2000.0))
    assert g["total"] > 100 and g["n_written"] == 100 and len(g["ids"]) == 100
    assert int(g["groups"]["count"].sum()) == g["total"]
    p.close()


def random_changes(R, ents, rng, n, frozen, centre=(8192.0, 8192.0, 8192.0)):
    """a batch of user-logic change requests touching every kind and component.  `frozen`: ids of entities that were static when
    the static render cache froze (first render) that the batch must leave alone apart from MakeObjectStatic (empty: none -- waking,
    moving or deleting them leaves ghost instances in the cache, see re_apply_changes)."""
    C = R._capi
    ch = np.zeros(n, R.CHANGE_DT)
    ids = ents["id"]; fl = ents["flags"]
    free = ids[~np.isin(ids, list(frozen))]
    dyn = ids[(fl & (R.F_HAS_VEL | R.F_HAS_ROTVEL)) != 0]
    for i in range(n):
        k = rng.random()
        if k < 0.45:
            ch[i] = (C.CHANGE_MODIFY, rng.choice(free), C.C_POSITION, 0, tuple(np.float32(centre) + rng.uniform(-700, 700, 3).astype(np.float32)) + (0,))
        elif k < 0.6:
            ch[i] = (C.CHANGE_MODIFY, rng.choice(free), C.C_ROTATION, 0, tuple(rng.uniform(-1, 1, 3).astype(np.float32) + np.float32([0, 1.5, 0])) + (np.float32(rng.uniform(-3, 3)),))
        elif k < 0.7:
            ch[i] = (C.CHANGE_MODIFY, rng.choice(free), C.C_SCALE, 0, tuple(rng.uniform(0.5, 3, 3).astype(np.float32)) + (0,))
        elif k < 0.78:
            comp = rng.choice([C.C_VELOCITY, C.C_ACCELERATION, C.C_ROTATION_VEL, C.C_ROTATION_ACC])
            v = rng.uniform(-20, 20, 4).astype(np.float32)
            if comp in (C.C_ROTATION_VEL, C.C_ROTATION_ACC):
                v[:3] = rng.uniform(-1, 1, 3) + np.array([1.5, 0, 0]); v[3] = rng.uniform(-1, 1)
            ch[i] = (C.CHANGE_MODIFY, rng.choice(dyn), comp, 0, tuple(v))
        elif k < 0.81:       # RemoveComponent: presence bit off, nothing recomputed (Rotation is left alone: the reference unwraps it for entities with VelocityRotation)
            ch[i] = (C.CHANGE_REMOVE_COMPONENT, rng.choice(free), rng.choice([C.C_SCALE, C.C_ACCELERATION, C.C_ROTATION_ACC, C.C_VELOCITY]), 0, (0, 0, 0, 0))
        elif k < 0.86:
            ch[i] = (C.CHANGE_MAKE_STATIC, rng.choice(ids), 0, 0, (0, 0, 0, 0))
        elif k < 0.94:
            ch[i] = (C.CHANGE_WAKE_UP, rng.choice(free), 0, 0, (0, 0, 0, 0))
        else:
            ch[i] = (C.CHANGE_DELETE, rng.choice(free), 0, 0, (0, 0, 0, 0))
    return ch


@pytest.mark.parametrize("batch", [150, 700])
def test_apply_changes_parity(R, batch):
    """apply_change for user change requests: Modify (all kinematic components), Delete, MakeObjectStatic, WakeUpRequest;
    several batches interleaved with frames and ticks; sections, entities and visible sets stay bit-exact.
    batch 150: one launch per batch (k_apply_small: at most 256 component writes and 256 moved entities); 700: the general path"""
    ents = R.synthetic.mixed_world(3000, seed=21, spread=600.0)
    p, w = build_pair(R, ents)
    rng = np.random.default_rng(5)
    cams = [R.Camera((8192 + 40 * i, 8192, 8500 - 30 * i), (0.05 * i, 0, -1), 1200.0) for i in range(5)]
    alive = ents
    frozen = set()                                           # static entities of the frozen cache are fair game: ghost instances
    for f, cam in enumerate(cams):
        check_frame(R, p, w, cam, f % 2 == 1)
        n_o, oob_o = w.tick(oracle_camera(cam), 0.016)
        t = p.tick(0.016)
        assert t["n_changed"] == n_o and t["n_out_of_bounds"] == len(oob_o)
        ch = random_changes(R, alive, rng, batch, frozen)
        if f == 2:                                           # push a few entities out of the world: with and without OutOfBoundsLogic
            far = np.zeros(6, R.CHANGE_DT)
            for i in range(6):
                far[i] = (R._capi.CHANGE_MODIFY, [e for e in alive["id"] if int(e) not in frozen][10 + i], R._capi.C_POSITION, 0, (-500.0, 8192.0, 20000.0, 0))
            ch = np.concatenate([ch, far])
        n_a, oob_a = w.apply_changes(ch.view(ro.CHANGE_DT))
        g = p.apply_changes(ch)
        assert g["n_changed"] == n_a and g["n_out_of_bounds"] == len(oob_a), (g, n_a, len(oob_a))
        assert sorted(p.out_of_bounds()) == sorted(int(i) for i in oob_a)
        check_sections(p, w)
        check_entities(R, p, w, ents)
    check_frame(R, p, w, cams[0], True)
    check_frame(R, p, w, cams[1], False)
    p.close(); w.close()


def test_add_entities_between_frames(R):
    """Pipeline::register_model_instances after the first frame (flows/pipeline.rs:186-208): new rows, slots of the dynamic table, new (ModelId, sortable)
    groups, shared sections created by straddlers, an instance out of bounds -- in a world with unique and shared sections, spinners and movers.  The added
    entities here are non-static (a static one re-caches its section: the lattice test below)."""
    C = R._capi
    ents = R.synthetic.mixed_world(2500, seed=31, spread=500.0)
    p, w = build_pair(R, ents)
    cams = [R.Camera((8192 + 30 * i, 8192, 8450 - 20 * i), (0.04 * i, 0, -1), 1100.0) for i in range(6)]
    all_ents = ents
    next_id = int(ents["id"].max()) + 1
    for f, cam in enumerate(cams):
        check_frame(R, p, w, cam, f % 2 == 1)
        n_o, oob_o = w.tick(oracle_camera(cam), 0.016); t = p.tick(0.016)
        assert t["n_changed"] == n_o and t["n_out_of_bounds"] == len(oob_o)
        if f in (0, 2, 3):
            new = R.synthetic.mixed_world(180 + 40 * f, seed=100 + f, spread=450.0)
            new["id"] = np.arange(next_id, next_id + len(new), dtype=np.uint32); next_id += len(new)
            new["flags"] &= ~np.uint32(C.F_STATIC)
            new["model_index"] += np.uint32(3 * f)                       # model ids the upload did not know: new group classes
            if f == 3:
                new["model_index"] = 100 + np.arange(len(new), dtype=np.uint32) % 170      # 170 more models: the group table outgrows the result block's InstanceRange capacity (regrow_groups) and the one-launch packs' 512 slots
            if f == 2:
                new["pos"][5] = (-900.0, 100.0, 100.0)                   # out of bounds: created, not inserted
            host_before = p.stats()["n_host_rebuckets"]
            assert p.register_model_instances(new) == w.register(to_oracle(new)) == (1 if f == 2 else 0)
            if f == 0: assert p.stats()["n_host_rebuckets"] == host_before, "the section inserts of non-static added entities left the device path"      # (a later batch may find the spare slots of a level used up: host path, full rebuild)
            all_ents = np.concatenate([all_ents, new])
            check_sections(p, w)
            assert p.stats()["n_entities"] == len(all_ents)
    check_entities(R, p, w, all_ents)
    assert len(p.get_indexes_for_components([C.C_VELOCITY])) == len([e for e in all_ents if w.entity(int(e["id"])) is not None and w.entity(int(e["id"]))["flags"] & C.F_HAS_VEL])
    p.close(); w.close()


def test_add_entities_static_recache_and_in_frame(R):
    """A world without shared sections (one entity per level-0 section).  Between frames a static instance re-caches its section at the next render
    (the changed-static set survives until then): ghosts parked there go, rows hidden there show again.  Inside a frame (AddEntity of apply_change,
    helper_things/entity_change_helpers.rs:48-107) the list is processed in order: the added entity is modified, deleted and its id added again in the
    same batch; a static one stays undrawn until a later re-cache; sortable components come and go; Velocity written to an entity registered without it."""
    C = R._capi
    ents = R.synthetic.lattice_world(cells_per_axis=16, first_cell=120, spinner_every=5)
    p, w = build_pair(R, ents)
    cam = R.Camera((8192, 8192, 8500), (0, 0, -1), 900.0); oc = oracle_camera(cam)

    def frame(dups=False):
        check_frame(R, p, w, cam, dups)
        n_o, _ = w.tick(oc, 0.016); t = p.tick(0.016)
        assert t["n_changed"] == n_o

    def small(ids, cells, static, model=2, vel=None):
        e = np.zeros(len(ids), R.ENTITY_DT)
        e["id"] = ids; e["model_index"] = model
        e["flags"] = C.F_STATIC if static else 0
        for k in range(3):
            e["original"][:, 2 * k] = -1.0; e["original"][:, 2 * k + 1] = 1.0
        e["scale"] = 1.0; e["rot_axis"][:, 0] = 1; e["rotvel_axis"][:, 0] = 1; e["rotacc_axis"][:, 0] = 1
        for i, (cx, cy, cz) in enumerate(cells):
            e["pos"][i] = (cx * 64.0 + 20.0 + 3 * i, cy * 64.0 + 30.0, cz * 64.0 + 25.0)
        if vel is not None:
            e["flags"] |= C.F_HAS_VEL; e["vel"][:] = vel
        return e

    frame(); frame(True)
    static_ids = [int(e["id"]) for e in ents if e["flags"] & C.F_STATIC]
    # a change batch that leaves ghosts and hidden rows behind: wake a cached static entity, move another, make an active one static
    ch = np.zeros(4, R.CHANGE_DT)
    ch[0] = (C.CHANGE_WAKE_UP, static_ids[700], 0, 0, (0, 0, 0, 0))
    ch[1] = (C.CHANGE_MODIFY, static_ids[701], C.C_POSITION, 0, tuple(ents[ents["id"] == static_ids[701]]["pos"][0] + np.float32(3.0)) + (0,))
    ch[2] = (C.CHANGE_MAKE_STATIC, int(ents["id"][0]), 0, 0, (0, 0, 0, 0))
    ch[3] = (C.CHANGE_WAKE_UP, static_ids[702], 0, 0, (0, 0, 0, 0))                    # a ghost nobody re-caches: it stays through the growth of the row columns below
    w.apply_changes(ch.view(ro.CHANGE_DT)); p.apply_changes(ch)
    frame(); check_sections(p, w)
    # between frames: static and active instances; two of the static ones land in the sections that hold the ghost / the hidden row
    def cell_of(eid):
        q = ents[ents["id"] == eid]["pos"][0]
        return (int(q[0] // 64), int(q[1] // 64), int(q[2] // 64))
    cells = [cell_of(static_ids[700]), cell_of(static_ids[701]), cell_of(int(ents["id"][0])), (125, 126, 127), (126, 126, 131)]
    nid = 900000
    new = np.concatenate([small(np.arange(nid, nid + 5, dtype=np.uint32), cells, True, model=11),
                          small(np.arange(nid + 5, nid + 9, dtype=np.uint32), [(124, 125, 130), (125, 125, 130), (127, 128, 131), (60, 300, 60)], False, model=12, vel=(15.0, 0.0, -4.0))])
    assert p.register_model_instances(new) == w.register(to_oracle(new)) == 1         # (60, 300, 60): y = 19230 lies outside the world
    check_sections(p, w)
    frame(True); frame()
    # inside a frame: AddEntity, interleaved with changes of the added entities
    add = np.concatenate([small(np.array([nid + 20, nid + 21, nid + 22], np.uint32), [(126, 127, 130), (127, 127, 130), (128, 127, 130)], False, model=13),
                          small(np.array([nid + 23], np.uint32), [(126, 128, 130)], True, model=13),
                          small(np.array([nid + 21], np.uint32), [(129, 127, 130)], False, model=14)])            # the id of a deleted entity, added again
    A = lambda k, i: (C.CHANGE_ADD_ENTITY, int(add["id"][i]), 0, i, (0, 0, 0, 0))
    ch = np.zeros(12, R.CHANGE_DT)
    ch[0] = A(0, 0)
    ch[1] = (C.CHANGE_MODIFY, nid + 20, C.C_POSITION, 0, (126 * 64.0 + 40.0, 127 * 64.0 + 31.0, 130 * 64.0 + 25.0, 0))      # stays in its section: add_entity returns early
    ch[2] = A(0, 1)
    ch[3] = (C.CHANGE_MODIFY, nid + 21, C.C_VELOCITY, 0, (5.0, 0.0, 0.0, 0))
    ch[4] = (C.CHANGE_DELETE, nid + 21, 0, 0, (0, 0, 0, 0))
    ch[5] = A(0, 4)                                                                                                    # id nid + 21 again (ECS::create_entity reuses freed ids)
    ch[6] = A(0, 2)
    ch[7] = (C.CHANGE_MODIFY, nid + 22, C.C_POSITION, 0, (140 * 64.0 + 10.0, 127 * 64.0 + 31.0, 130 * 64.0 + 25.0, 0))      # leaves for another section
    ch[8] = A(0, 3)                                                                                                    # static, inside a frame: not drawn until re-cached
    ch[9] = (C.CHANGE_ADD_SORTABLE, int(ents["id"][5]), 2, 0, (0, 0, 0, 0))                                            # an active spinner
    ch[10] = (C.CHANGE_ADD_SORTABLE, static_ids[300], 3, 0, (0, 0, 0, 0))                                              # a cached static entity: the snapshot keeps its old bucket
    ch[11] = (C.CHANGE_MODIFY, static_ids[301], C.C_VELOCITY, 0, (0.0, 9.0, 0.0, 0))                                   # Velocity on an entity registered without one
    n_a, oob_a = w.apply_changes(ch.view(ro.CHANGE_DT), added=to_oracle(add)); g = p.apply_changes(ch, added=add)
    assert g["n_changed"] == n_a and g["n_out_of_bounds"] == len(oob_a)
    check_sections(p, w)
    frame(); frame(True)
    ch = np.zeros(3, R.CHANGE_DT)
    ch[0] = (C.CHANGE_REMOVE_SORTABLE, int(ents["id"][5]), 0, 0, (0, 0, 0, 0))
    ch[1] = (C.CHANGE_WAKE_UP, static_ids[301], 0, 0, (0, 0, 0, 0))                                                    # now it moves with its new velocity
    ch[2] = (C.CHANGE_MODIFY, nid + 21, C.C_ROTATION_VEL, 0, (0.0, 1.0, 0.0, 0.3))
    w.apply_changes(ch.view(ro.CHANGE_DT)); p.apply_changes(ch)
    for _ in range(3):
        frame()
    # a static instance between frames into the section of the in-frame static one: the re-cache shows both
    more = small(np.array([nid + 40], np.uint32), [(126, 128, 130)], True, model=13)
    assert p.register_model_instances(more) == w.register(to_oracle(more)) == 0
    frame(True); frame()
    everything = np.concatenate([ents, new, add[[0, 2, 3, 4]], more])
    check_entities(R, p, w, everything)
    check_sections(p, w)
    st = p.stats(); assert st["n_entities"] == len(ents) + len(new) + len(add) + len(more)
    p.close(); w.close()


def test_frozen_static_cache_ghosts(R):
    """the static render cache is a snapshot taken at the first render (render_flow.rs:549-594; pipeline.rs:271 clears the
    changed set): a cached static entity that is deleted, woken or moved stays in the picture with its old matrix; the section that
    cached it keeps drawing it after being emptied and re-created; an entity made static later is not drawn at all"""
    C = R._capi
    ents = R.synthetic.lattice_world(cells_per_axis=10, first_cell=124, straddler_fraction=0.05)
    p, w = build_pair(R, ents)
    cam = R.Camera((8192 + 120, 8192 + 100, 8192 + 900), (0, 0, -1), 2500.0)
    g0, _ = check_frame(R, p, w, cam, False)                   # the freeze
    assert g0["total"] > 500
    def frame(dups=False):
        n_o, oob_o = w.tick(oracle_camera(cam), 0.016); t = p.tick(0.016)
        assert t["n_changed"] == n_o
        return check_frame(R, p, w, cam, dups)[0]
    def apply(rows):
        ch = np.zeros(len(rows), R.CHANGE_DT)
        for i, r in enumerate(rows): ch[i] = r
        n_a, oob_a = w.apply_changes(ch.view(ro.CHANGE_DT)); g = p.apply_changes(ch)
        assert g["n_changed"] == n_a and g["n_out_of_bounds"] == len(oob_a)
        check_sections(p, w)
    drawn = sorted(int(i) for i in g0["ids"][:g0["total"]])
    dele, woke, moved, mover = drawn[0:20], drawn[20:40], drawn[40:60], drawn[60:70]
    # 1. delete / wake up / move away (the sections of `moved` become empty and disappear)
    apply([(C.CHANGE_DELETE, i, 0, 0, (0, 0, 0, 0)) for i in dele] + [(C.CHANGE_WAKE_UP, i, 0, 0, (0, 0, 0, 0)) for i in woke]
          + [(C.CHANGE_MODIFY, i, C.C_POSITION, 0, (8192.0 + 3 * k, 8192.0, 9500.0, 0)) for k, i in enumerate(moved)])
    g1 = frame()
    ids1 = g1["ids"][:g1["total"]].tolist()
    assert all(ids1.count(i) <= 1 for i in dele)                # deleted: at most the ghost (none where the delete emptied the section: no section, no cache lookup)
    assert all(ids1.count(i) == 2 for i in woke)                # woken: the ghost and the live active entity
    # 2. other entities move into the emptied sections: those sections exist again and their cache entries (the ghosts) show again
    home = {int(e["id"]): e["pos"] for e in ents[np.isin(ents["id"], moved)]}
    apply([(C.CHANGE_WAKE_UP, i, 0, 0, (0, 0, 0, 0)) for i in mover]
          + [(C.CHANGE_MODIFY, i, C.C_POSITION, 0, tuple(home[moved[k]]) + (0,)) for k, i in enumerate(mover)])
    g2 = frame(True)
    ids2 = g2["ids"][:g2["total"]].tolist()
    assert sum(ids2.count(i) >= 1 for i in moved[:len(mover)]) >= 7     # (a straddler among them was cached by another section)
    # 3. back to static: never drawn live again (hidden rows), the ghosts stay
    apply([(C.CHANGE_MAKE_STATIC, i, 0, 0, (0, 0, 0, 0)) for i in woke + mover])
    g3 = frame()
    ids3 = g3["ids"][:g3["total"]].tolist()
    assert all(ids3.count(i) == 1 for i in woke)
    # 4. and awake once more, rotated and scaled
    apply([(C.CHANGE_WAKE_UP, i, 0, 0, (0, 0, 0, 0)) for i in woke[:10]] + [(C.CHANGE_MODIFY, i, C.C_SCALE, 0, (2.0, 2.0, 2.0, 0)) for i in woke[:10]]
          + [(C.CHANGE_MODIFY, i, C.C_ROTATION, 0, (0.0, 1.0, 0.0, 0.7)) for i in dele[:5] + woke[5:15]])
    for dups in (False, True): frame(dups)
    check_frame(R, p, w, cam, True, force_large_pack=True)      # ghosts and hidden rows through the multi-kernel pack as well
    check_entities(R, p, w, ents[::7])
    p.close(); w.close()


@pytest.mark.parametrize("seed,atomic", [(3, 64), (57, 16), (101, 64)])
def test_probe_path_soak(R, seed, atomic):
    """RE_CFG_PROBE: the visibility query through hash probes of the candidate cells (k_probe_cull) instead of the key stream --
    same randomized soak as above (movers, change batches, emptied and re-created sections keep the key -> slot table in step),
    every frame compared with the oracle, and the probe path must really have served the narrow frames"""
    rng = np.random.default_rng(seed)
    ents = R.synthetic.mixed_world(2500 + 500 * (seed % 4), seed=seed, spread=350.0 + 60.0 * (seed % 5), atomic=atomic)
    ents["vel"] *= 8.0
    p, w = build_pair(R, ents, atomic=atomic, flags=R._capi.CFG_PROBE | R._capi.CFG_PROBE_ALWAYS)
    for f in range(40):
        pos = (8192 + rng.uniform(-300, 300), 8192 + rng.uniform(-200, 200), 8192 + rng.uniform(-100, 500))
        d = rng.uniform(-1, 1, 3); d[2] -= 1.5
        cam = R.Camera(pos, tuple(d / np.linalg.norm(d)), float(rng.choice([150.0, 300.0, 2500.0])))   # narrow frusta probe, the wide one streams
        oc = oracle_camera(cam)
        if f % 3:
            p.cull_and_pack(cam, asynchronous=True, copy=False); p.tick(0.04, asynchronous=True)
            w.cull(oc); w.render(oc); w.tick(oc, 0.04)
        else:
            check_frame(R, p, w, cam, bool(f % 2))
            n_o, oob_o = w.tick(oc, 0.04); t = p.tick(0.04)
            assert t["n_changed"] == n_o and t["n_out_of_bounds"] == len(oob_o)
        if f % 4 == 1:
            ch = random_changes(R, ents, rng, 60, set())
            w.apply_changes(ch.view(ro.CHANGE_DT)); p.apply_changes(ch)
    p.wait()
    check_sections(p, w)
    st = p.stats()
    assert st["n_probe_frames"] >= 10, st
    check_frame(R, p, w, R.Camera((8192, 8192, 8400), (0, 0, -1), 200.0), True)
    p.close(); w.close()


def test_probe_path_static_lattice(R):
    """configs[1] at reduced size through the probe path, and the same frames through the stream (RE_CULL_FORCE_STREAM): identical"""
    ents = R.synthetic.lattice_world(cells_per_axis=40, first_cell=108)
    p, w = build_pair(R, ents, flags=R._capi.CFG_PROBE | R._capi.CFG_PROBE_ALWAYS)
    for pos, d, far in [((8192, 8192, 8192), (0, 0, -1), 500.0), ((7500.5, 8100.25, 9000), (0.6, 0.0, -0.8), 700.0), ((8192, 8192, 8300), (0.2, 0.3, -1), 400.0)]:
        for dups in (False, True):
            g, o = check_frame(R, p, w, R.Camera(pos, d, far), dups)
            w.tick(oracle_camera(R.Camera(pos, d, far)), 0.016); p.tick(0.016)
    assert p.stats()["n_probe_frames"] == 6 and g["total"] > 0
    p.close(); w.close()


@pytest.mark.parametrize("atomic,two_lanes", [(64, False), (16, False), (64, True), (16, True)])
def test_deferred_pack_static_fly_through(R, atomic, two_lanes):
    """RE_CULL_DEFER_PACK: asynchronous frames of a static world leave their pack to the next frame's launch (k_scan_cull_fused, one
    launch per frame).  A camera flying through the lattice: whatever frame is waited for, copied or followed by a synchronous frame
    is bit-exact with the oracle, and the fused launches did happen"""
    # atomic 16: more than 512 sections per axis, so the fused launch runs over the full 64-bit stream keys
    first = (16384 // atomic - 40) // 2
    ents = R.synthetic.lattice_world(cells_per_axis=40, first_cell=first, atomic=atomic, straddler_fraction=0.03)
    p, w = build_pair(R, ents, atomic=atomic)
    s_ = atomic / 64.0
    cams = [R.Camera((8192 + 9.5 * i * s_, 8192 - 4.25 * i * s_, 8192 + (308 - 11.0 * i) * s_), (0.02 * i, 0.01 * i, -1), 900.0 * s_) for i in range(24)]
    def oracle_frame(cam, dups=False):
        oc = oracle_camera(cam); w.cull(oc); o = w.render(oc, emit_duplicates=dups); w.tick(oc, 0.016); return o
    check_frame(R, p, w, cams[0], False); w.tick(oracle_camera(cams[0]), 0.016); p.tick(0.016)          # the static cache freezes here
    for i, cam in enumerate(cams[1:], 1):
        if i % 6 == 0:                                            # a synchronous frame in between picks up the deferred pack of its predecessor
            check_frame(R, p, w, cam, bool(i % 4 == 0)); w.tick(oracle_camera(cam), 0.016); p.tick(0.016)
            continue
        p.cull_and_pack(cam, asynchronous=True, copy=False, defer_pack=True, two_lanes=two_lanes); p.tick(0.016, asynchronous=True)
        o = oracle_frame(cam)
        if i % 5 == 0:                                            # wait for this very frame: its pack is sent off on its own
            vis, _ = p.wait(copy=True)
            assert_render_equal(vis, o)
    vis, _ = p.wait(copy=True)
    assert_render_equal(vis, o)
    st = p.stats()
    assert st["n_fused_frames"] >= (6 if two_lanes else 10), st
    assert (st["reserved"] >= 12) == two_lanes, st               # lane switches (RE_CULL_TWO_LANES: frames alternate between two streams)
    p.close(); w.close()


@pytest.mark.parametrize("two_lanes", [False, True])
def test_deferred_pack_with_change_batches(R, two_lanes):
    """deferred packs and user change batches: a batch between two asynchronous frames must find the pending pack done first (it reads
    the matrices the batch rewrites), and the frames after it see the changed world -- ghosts of the frozen static cache included"""
    C = R._capi
    ents = R.synthetic.lattice_world(cells_per_axis=24, first_cell=116)
    p, w = build_pair(R, ents, flags=C.CFG_PROBE | (C.CFG_FULL_REBUILD if two_lanes else 0))   # two lanes: every batch rebuilds the table, the parked lane's stamps must follow; (the key -> slot table just rides along: narrow frames take the probe kernel and send a pending pack off on its own)
    rng = np.random.default_rng(12)
    cam0 = R.Camera((8192, 8192, 8192 + 700), (0, 0, -1), 1500.0)
    check_frame(R, p, w, cam0, False); w.tick(oracle_camera(cam0), 0.016); p.tick(0.016)
    ids = ents["id"]
    for i in range(1, 16):
        cam = R.Camera((8192 + 6.0 * i, 8192 - 3.0 * i, 8192 + 700 - 9.0 * i), (0.01 * i, 0, -1), 1500.0 if i % 4 else 250.0)
        oc = oracle_camera(cam)
        p.cull_and_pack(cam, asynchronous=True, copy=False, defer_pack=True, two_lanes=two_lanes); p.tick(0.016, asynchronous=True)
        w.cull(oc); o = w.render(oc); w.tick(oc, 0.016)
        if i % 3 == 0:
            ch = np.zeros(12, R.CHANGE_DT)
            for k in range(12):
                e = int(rng.choice(ids)); kind = rng.integers(0, 4)
                ch[k] = ((C.CHANGE_MODIFY, e, C.C_POSITION, 0, (8192 + rng.uniform(-300, 300), 8192 + rng.uniform(-300, 300), 8192 + rng.uniform(-300, 300), 0)) if kind == 0 else
                         (C.CHANGE_WAKE_UP, e, 0, 0, (0, 0, 0, 0)) if kind == 1 else (C.CHANGE_MAKE_STATIC, e, 0, 0, (0, 0, 0, 0)) if kind == 2 else (C.CHANGE_DELETE, e, 0, 0, (0, 0, 0, 0)))
            n_a, _ = w.apply_changes(ch.view(ro.CHANGE_DT)); g = p.apply_changes(ch)
            assert g["n_changed"] == n_a
    vis, _ = p.wait(copy=True)
    assert_render_equal(vis, o)
    check_sections(p, w)
    st = p.stats()
    assert st["n_fused_frames"] >= (3 if two_lanes else 5) and st["n_probe_frames"] >= 2, st
    check_frame(R, p, w, cam0, True)
    p.close(); w.close()


def test_context_reuse_and_call_order_errors(R):
    """one context, several worlds and every optional path (probe table, collision scratch, deferred packs): a second upload starts
    from scratch; calls out of order fail with RE_E_STATE instead of computing on stale state"""
    C = R._capi
    p = R.Pipeline(16384, 64, flags=C.CFG_PROBE)
    with pytest.raises(R.RenderEngineError):
        p.cull_and_pack(R.Camera((8192, 8192, 8192), (0, 0, -1), 500.0))          # no world yet
    a = collision_world(R, 1500, 4, 200.0)
    p.register_model_instances(a)
    with pytest.raises(R.RenderEngineError):
        p.collide()                                                                # no visibility query yet
    with pytest.raises(R.RenderEngineError):
        p.tick(0.016)
    w = ro.World(16384, 64); w.register(to_oracle(a))
    cam = R.Camera((8192, 8192, 8400), (0, 0, -1), 600.0); oc = oracle_camera(cam)
    check_frame(R, p, w, cam, False)
    got, n = p.collide(); assert n == len(w.collide(oc))
    w.tick(oc, 0.02); p.tick(0.02)
    w.close()
    # a different world into the same context: static lattice, deferred packs, an empty view, then a populated one
    b = R.synthetic.lattice_world(cells_per_axis=20, first_cell=118)
    p.replace_world(b)
    w = ro.World(16384, 64); w.register(to_oracle(b))
    empty = R.Camera((200, 200, 200), (0, 0, -1), 100.0)
    check_frame(R, p, w, empty, False); w.tick(oracle_camera(empty), 0.016); p.tick(0.016)      # the cache freezes with nothing in range
    for i in range(5):
        p.cull_and_pack(empty, asynchronous=True, copy=False, defer_pack=True); p.tick(0.016, asynchronous=True)
        w.cull(oracle_camera(empty)); w.render(oracle_camera(empty)); w.tick(oracle_camera(empty), 0.016)
    vis, _ = p.wait(copy=True); assert vis["total"] == 0
    near = R.Camera((8192, 8192, 8500), (0, 0, -1), 900.0)
    g, o = check_frame(R, p, w, near, True)
    assert g["total"] == 0 and g["n_visible_sections"] > 0                       # first-sight quirk: nothing was cached when the cache froze
    got, n = p.collide(); assert n == len(w.collide(oracle_camera(near))) == 0
    p.close(); w.close()


def test_rejected_calls_leave_the_world_untouched(R):
    """argument errors are reported and change nothing: unknown entity, component that cannot be modified, an AddEntity without the entity,
    delta_time 0 with rotating entities (the reference asserts, movement_components.rs:287), a batch
    with one bad request in the middle -- afterwards the world still equals the oracle's, which saw none of it"""
    C = R._capi
    ents = R.synthetic.mixed_world(1200, seed=8, spread=300.0)
    p, w = build_pair(R, ents)
    cam = R.Camera((8192, 8192, 8450), (0, 0, -1), 800.0)
    check_frame(R, p, w, cam, False)
    some = int(ents["id"][5])
    def one(kind, eid, comp=0, v=(1, 2, 3, 0)):
        ch = np.zeros(1, R.CHANGE_DT); ch[0] = (kind, eid, comp, 0, v); return ch
    for bad in (one(C.CHANGE_MODIFY, 0xFFFFFF0), one(C.CHANGE_MODIFY, some, 9), one(77, some), one(C.CHANGE_ADD_ENTITY, some)):      # (Velocity on an entity registered without one is served since round 3: it gets a slot of the dynamic table)
        with pytest.raises(R.RenderEngineError):
            p.apply_changes(bad)
    batch = np.concatenate([one(C.CHANGE_MODIFY, some, C.C_POSITION, (8000, 8000, 8000, 0)), one(C.CHANGE_DELETE, 0xFFFFFF0), one(C.CHANGE_WAKE_UP, some)])
    with pytest.raises(R.RenderEngineError):
        p.apply_changes(batch)                                    # refused whole: the first request must not have been applied
    with pytest.raises(R.RenderEngineError):
        p.tick(0.0)
    check_sections(p, w)
    check_entities(R, p, w, ents[::3])
    check_frame(R, p, w, cam, True)
    n_o, oob_o = w.tick(oracle_camera(cam), 0.02); t = p.tick(0.02)
    assert t["n_changed"] == n_o
    check_frame(R, p, w, cam, False)
    p.close(); w.close()


def test_tick_with_many_movers_ticks_every_entity(R):
    """a tick whose first workgroups find entities that change world section raises the "tree is stale" word for the frames behind it;
    the workgroups of the same tick that start later must still process their entities (regression: they read the word and returned,
    so a box-dependent handful of entities was never ticked).  Enough dynamic entities for ~100 workgroups, most of them movers."""
    ents = R.synthetic.mixed_world(60000, seed=77, spread=1500.0)
    ents["vel"] *= 10.0
    p, w = build_pair(R, ents)
    cam = R.Camera((8192, 8192, 9800), (0, 0, -1), 4000.0)
    for f in range(4):
        check_frame(R, p, w, cam, False)
        n_o, oob_o = w.tick(oracle_camera(cam), 0.05)
        t = p.tick(0.05)
        assert t["n_changed"] == n_o and t["n_out_of_bounds"] == len(oob_o), (f, t, n_o)
        assert t["n_rebucket"] > 100
    check_sections(p, w)
    check_entities(R, p, w, ents[::50])
    # the same through asynchronous frames (speculation: the frames behind a stale tick cancel themselves and are replayed)
    for f in range(4):
        p.cull_and_pack(cam, asynchronous=True, copy=False); p.tick(0.05, asynchronous=True)
        oc = oracle_camera(cam); w.cull(oc); w.render(oc); w.tick(oc, 0.05)
    p.wait()
    check_sections(p, w)
    check_entities(R, p, w, ents[::50])
    p.close(); w.close()


def sorted_pairs(a):
    a = np.asarray(a, np.uint32).reshape(-1, 2)
    return a[np.lexsort((a[:, 1], a[:, 0]))]


def collision_world(R, n, seed, spread, atomic=64):
    ents = R.synthetic.mixed_world(n, seed=seed, spread=spread, atomic=atomic)
    u = R.synthetic.uniform(seed, np.arange(n, dtype=np.uint64), 77)
    ents["flags"][u < 0.7] |= R.F_CAN_COLLIDE
    return ents


@pytest.mark.parametrize("seed,n,spread,atomic", [(5, 2500, 160.0, 64), (9, 4000, 260.0, 64), (13, 1500, 150.0, 16)])
def test_collision_broad_phase_parity(R, seed, n, spread, atomic):
    """re_collide == LogicFlow::handle_collisions (flows/logic_flow.rs:452-651): the (this, other) pairs of every collision-logic
    invocation as a multiset, frame after frame with ticks (movers change sections), user change batches and a roaming camera;
    dense clusters so that unique sections of several levels, shared sections and the related-section closure all take part"""
    ents = collision_world(R, n, seed, spread, atomic)
    ents["vel"] *= 3.0
    p, w = build_pair(R, ents, atomic=atomic)
    rng = np.random.default_rng(seed)
    total = 0
    for f in range(10):
        pos = (8192 + rng.uniform(-spread, spread) * 0.6, 8192 + rng.uniform(-spread, spread) * 0.6, 8192 + rng.uniform(-0.3, 1.2) * spread)
        d = rng.uniform(-1, 1, 3); d[2] -= 1.2
        cam = R.Camera(pos, tuple(d / np.linalg.norm(d)), float(rng.choice([400.0, 1500.0])))
        oc = oracle_camera(cam)
        check_frame(R, p, w, cam, bool(f % 2))
        want = sorted_pairs(w.collide(oc))
        got, n_total = p.collide()
        assert n_total == len(want), (f, n_total, len(want))
        np.testing.assert_array_equal(sorted_pairs(got), want, err_msg=f"frame {f}")
        total += n_total
        if f == 3:                                                   # truncation reports the total and fills what fits
            part, n2 = p.collide(capacity=min(7, n_total))
            assert n2 == n_total and len(part) == min(7, n_total)
        n_o, oob_o = w.tick(oc, 0.05); t = p.tick(0.05)
        assert t["n_changed"] == n_o and t["n_out_of_bounds"] == len(oob_o)
        if f % 3 == 2:
            ch = random_changes(R, ents, rng, 40, set())
            w.apply_changes(ch.view(ro.CHANGE_DT)); p.apply_changes(ch)
    assert total > 50
    p.close(); w.close()


def test_collision_user_entity_and_lattice(R):
    """the user entity always causes collisions (UserAlwaysCausesCollisions, pipeline.rs:136, logic_flow.rs:236-240); config-3 style
    world (one entity per section, spinners) around it"""
    ents = R.synthetic.lattice_world(cells_per_axis=12, first_cell=122, spinner_every=3, straddler_fraction=0.1)
    ents["flags"][(ents["flags"] & R.F_HAS_ROTVEL) != 0] |= R.F_CAN_COLLIDE
    user = ents[:1].copy()
    user["id"] = 9_000_000; user["flags"] = R.F_USER; user["pos"] = (8192 + 32, 8192 + 32, 8192 + 32); user["original"] = (-30, 30, -30, 30, -30, 30)   # inside section (128,128,128), whose entity (id 942) is a spinner
    ents = np.concatenate([ents, user])
    p, w = build_pair(R, ents)
    for i in range(3):
        cam = R.Camera((8192 + 40 + 20 * i, 8192 + 40, 8192 + 40), (0, 0, -1), 1000.0)
        oc = oracle_camera(cam)
        check_frame(R, p, w, cam, False)
        want = sorted_pairs(w.collide(oc))
        got, n_total = p.collide()
        assert n_total == len(want) and n_total > 0
        np.testing.assert_array_equal(sorted_pairs(got), want)
        assert [9_000_000, 942] in want.tolist() and [942, 9_000_000] in want.tolist()
        w.tick(oc, 0.016); p.tick(0.016)
    p.close(); w.close()


def test_async_frames_with_movers_are_replayed(R):
    """frames of a world with movers enqueued without waiting (speculation): a tick that finds section changes cancels the frames
    behind it and the library replays them on the patched tree -- the end state equals the frame-by-frame reference"""
    ents = R.synthetic.mixed_world(3000, seed=21, spread=600.0)
    ents["vel"] *= 12.0
    p, w = build_pair(R, ents)
    cams = [R.Camera((8192 + 25 * i, 8192 - 10 * i, 8600 - 30 * i), (0, 0, -1), 1500.0) for i in range(7)]
    for cam in cams[:-1]:                                            # reference: one frame after the other
        oc = oracle_camera(cam); w.cull(oc); w.render(oc); w.tick(oc, 0.05)
    for cam in cams[:-1]:                                            # GPU path: everything enqueued, nothing awaited
        p.cull_and_pack(cam, asynchronous=True, copy=False)
        p.tick(0.05, asynchronous=True)
    vis, tick = p.wait()
    assert p.stats()["n_rebuilds"] >= 3 if "n_rebuilds" in p.stats() else True
    check_sections(p, w)
    check_entities(R, p, w, ents[:600])
    check_frame(R, p, w, cams[-1], True)
    # the same again with a frame loop that mixes styles
    for i, cam in enumerate(cams):
        oc = oracle_camera(cam); w.cull(oc); w.render(oc); w.tick(oc, 0.03)
        p.cull_and_pack(cam, asynchronous=(i % 3 != 0), copy=False)
        p.tick(0.03, asynchronous=(i % 2 == 0))
    p.wait()
    check_sections(p, w)
    check_entities(R, p, w, ents[:600])
    p.close(); w.close()


def test_world_with_more_than_512_sections_per_axis(R):
    """outline / atomic = 1024: section indices need more than 9 bits, so the stream runs over the full 64-bit keys with the two
    packed 16-bit box tests instead of the compact 32-bit keys; several frames with ticks and movers"""
    ents = R.synthetic.mixed_world(3000, seed=33, spread=500.0, atomic=16)
    ents["vel"] *= 4.0
    p, w = build_pair(R, ents, atomic=16)
    check_sections(p, w)
    for i in range(4):
        cam = R.Camera((8192 + 15 * i, 8192, 8400 - 20 * i), (0.1 * i, 0, -1), 700.0)
        check_frame(R, p, w, cam, i % 2 == 1)
        n_o, oob_o = w.tick(oracle_camera(cam), 0.05)
        t = p.tick(0.05)
        assert t["n_changed"] == n_o and t["n_out_of_bounds"] == len(oob_o)
    check_sections(p, w)
    check_entities(R, p, w, ents[:300])
    p.close(); w.close()


@pytest.mark.parametrize("seed,atomic,tight", [(3, 64, False), (11, 64, True), (29, 64, False), (57, 16, False), (101, 64, True), (202, 16, True), (303, 64, False), (404, 16, False), (505, 64, True)])
def test_soak_random_frames(R, seed, atomic, tight):
    """a longer randomized run: movers, spinners, user change batches (every kind), cameras that jump around, synchronous and
    asynchronous frames mixed -- section table, entities and the rendered set are compared with the oracle every few frames.
    Exercises the in-place table patches (slot reuse, relocated row segments, emptied and re-created sections, shared-section churn)."""
    rng = np.random.default_rng(seed)
    ents = R.synthetic.mixed_world(2500 + 500 * (seed % 4), seed=seed, spread=350.0 + 60.0 * (seed % 5), atomic=atomic)
    ents["vel"] *= 8.0
    # atomic 16: more than 512 sections per axis -> full 64-bit stream keys; tight: hardly any slack -> patches and full rebuilds alternate
    p, w = build_pair(R, ents, atomic=atomic, flags=R._capi.CFG_TIGHT_SLACK if tight else 0)
    frozen = set()
    cam = R.Camera((8192, 8192, 8500), (0, 0, -1), 1000.0)
    for f in range(48):
        pos = (8192 + rng.uniform(-300, 300), 8192 + rng.uniform(-200, 200), 8192 + rng.uniform(-100, 500))
        d = rng.uniform(-1, 1, 3); d[2] -= 1.5
        cam = R.Camera(pos, tuple(d / np.linalg.norm(d)), float(rng.choice([600.0, 1000.0, 2500.0])))
        oc = oracle_camera(cam)
        asynchronous = bool(f % 3)
        if asynchronous:
            p.cull_and_pack(cam, asynchronous=True, copy=False); p.tick(0.04, asynchronous=True)
            w.cull(oc); w.render(oc); w.tick(oc, 0.04)
        else:
            check_frame(R, p, w, cam, bool(f % 2))
            n_o, oob_o = w.tick(oc, 0.04)
            t = p.tick(0.04)
            assert t["n_changed"] == n_o and t["n_out_of_bounds"] == len(oob_o)
        if f % 4 == 1:
            ch = random_changes(R, ents, rng, 60, frozen)
            n_a, oob_a = w.apply_changes(ch.view(ro.CHANGE_DT))
            g = p.apply_changes(ch)
            assert g["n_changed"] == n_a and g["n_out_of_bounds"] == len(oob_a)
        if f % 6 == 5:
            p.wait()
            check_sections(p, w)
            assert p.stats()["n_shared_sections"] == w.L.ro_num_shared(w.h)
    p.wait()
    check_sections(p, w)
    check_entities(R, p, w, ents[::5])
    check_frame(R, p, w, cam, True)
    st = p.stats()
    # the device path stays in use after the first change batch has left ghost instances in the frozen cache (round 3: a batch falls back only when it touches a section the ghost books know)
    assert tight or st["n_device_rebuckets"] >= 8, st
    p.close(); w.close()


def hopping_world(R, dims=(14, 14, 14), first=121, atomic=64, every=2):
    return R.synthetic.hopping_lattice(dims, first, atomic, every)


@pytest.mark.parametrize("straddlers,general", [(False, False), (True, False), (True, True)])
def test_rebucket_on_the_device(R, straddlers, general, monkeypatch):
    """batches of movers between world sections: the bookkeeping runs on the device (re_rebucket.hip: k_rb2_*), the host only notes which sections
    changed.  Sections are emptied (-> padding slots), created (free slots of the level run), outgrow their segment (relocated); the host mirrors are
    fetched on demand (a change-request batch, the debug getters) and the device path resumes afterwards.
    straddlers: some movers are wider than a world section (shared sections, higher levels): shared sections are created, emptied and re-created,
    the unique sections they link gain and lose links (and with the last one their existence), in the order of the reference's single pass -- on the
    device too: no part of a tick's batch is left to the host.
    general: the shortcuts of small batches are off (k_rb2_static_small, phase 3 chained behind phase 2 without a read-back): the kernels of large batches see these"""
    if general: monkeypatch.setenv("RE_EXP_RB2_GENERAL", "1")
    ents = hopping_world(R)
    if straddlers:
        mv = np.nonzero((ents["flags"] & R.F_HAS_VEL) != 0)[0][::23]
        for k, i in enumerate(mv):
            h = np.float32(20.0 + 7.0 * (k % 4))
            ents["original"][i] = (-h, h, -h, h, -h, h)
    p, w = build_pair(R, ents)
    assert (p.stats()["n_shared_sections"] > 0) == straddlers
    cams = [R.Camera((8192 + 10 * i, 8192 - 8 * i, 8192 + 2600), (0.02 * i, 0, -1), 6000.0) for i in range(7)]
    rng = np.random.default_rng(2)
    moved = 0
    for f, cam in enumerate(cams):
        check_frame(R, p, w, cam, f % 2 == 0)
        n_o, oob_o = w.tick(oracle_camera(cam), 1.0)
        t = p.tick(1.0)
        assert t["n_changed"] == n_o and t["n_out_of_bounds"] == len(oob_o) == 0, (f, t, n_o)
        moved += t["n_rebucket"]
        if f < 3: assert p.stats()["n_host_rebuckets"] == 0 and p.stats()["n_device_rebuckets"] == f + 1, p.stats()      # shared sections included
        if f % 2 == 1:                                           # looks at the table: host mirrors fetched from the device
            check_sections(p, w)
        if f == 3:                                               # a host-path batch in between (overlay and capacities change on the host)
            ch = np.zeros(40, R.CHANGE_DT)
            ids = rng.choice(ents["id"][(ents["flags"] & R.F_HAS_VEL) != 0], 40, replace=False)          # movers (no ghosts of the static cache), dropped at
            for k in range(40):                                                                          # section centres (no shared sections)
                ch[k] = (R._capi.CHANGE_MODIFY, ids[k], R._capi.C_POSITION, 0, (64.0 * (100 + k) + 32.0, 64.0 * 110 + 32.0, 64.0 * (140 - k) + 32.0, 0))
            g = p.apply_changes(ch); n_a, oob_a = w.apply_changes(ch.view(ro.CHANGE_DT))
            assert g["n_changed"] == n_a
            check_sections(p, w)
    st = p.stats()
    assert moved > 2000 and st["n_device_rebuckets"] >= 5 and st["n_table_rebuilds"] <= 1, (moved, st)      # (the small lattice has few spare slots per level: one batch may find no room and rebuild)
    check_sections(p, w)
    check_entities(R, p, w, ents[::7])
    check_frame(R, p, w, cams[0], False)
    p.close(); w.close()


def test_delete_and_move_batches_on_the_device(R):
    """change batches that move and DELETE non-static entities (DeleteRequest -> remove_entity inline, before the kinematic re-adds): the tree bookkeeping of the whole
    batch runs on the device -- the deleted rows ride at the end of the mover list with a remove op only, ordered in front of every mover.  Entities in unique and in
    shared sections, sections emptied by a deletion and re-created by a mover of the same batch; compared with the oracle after every batch."""
    ents = hopping_world(R, dims=(10, 10, 10), first=123)
    mv = np.nonzero((ents["flags"] & R.F_HAS_VEL) != 0)[0]
    for k, i in enumerate(mv[::7]):                                             # some wide movers: members of shared sections
        h = np.float32(22.0 + 6.0 * (k % 3)); ents["original"][i] = (-h, h, -h, h, -h, h)
    p, w = build_pair(R, ents)
    rng = np.random.default_rng(9)
    C = R._capi
    alive = list(ents["id"][mv])
    cam = R.Camera((8192, 8192, 8192 + 2200), (0, 0, -1), 5000.0)
    for f in range(6):
        check_frame(R, p, w, cam, f % 2 == 0)
        n_o, oob_o = w.tick(oracle_camera(cam), 1.0); t = p.tick(1.0)
        assert t["n_changed"] == n_o and t["n_out_of_bounds"] == len(oob_o) == 0
        host_before = p.stats()["n_host_rebuckets"]
        rng.shuffle(alive)
        dele, move = alive[:12], alive[12:40]; alive = alive[12:]
        ch = np.zeros(len(dele) + len(move), R.CHANGE_DT); o = 0
        for k in range(max(len(dele), len(move))):                               # deletions and moves interleaved in the list
            if k < len(move): ch[o] = (C.CHANGE_MODIFY, move[k], C.C_POSITION, 0, (64.0 * (123 + int(rng.integers(0, 10))) + 32.0, 64.0 * (123 + int(rng.integers(0, 10))) + 32.0, 64.0 * (123 + int(rng.integers(0, 10))) + 32.0, 0)); o += 1
            if k < len(dele): ch[o] = (C.CHANGE_DELETE, dele[k], 0, 0, (0, 0, 0, 0)); o += 1
        n_a, oob_a = w.apply_changes(ch.view(ro.CHANGE_DT)); g = p.apply_changes(ch)
        assert g["n_changed"] == n_a and g["n_out_of_bounds"] == len(oob_a)
        assert p.stats()["n_host_rebuckets"] == host_before, "a batch of moves and deletions of non-static entities left the device path"
        check_sections(p, w)
    check_entities(R, p, w, ents[::3])
    check_frame(R, p, w, cam, True)
    p.close(); w.close()


@pytest.mark.parametrize("k,wide", [(1, False), (113, False), (114, False), (255, False), (256, False), (257, False), (1023, False), (1024, False), (1025, False), (150, True), (600, True)])
def test_change_batch_size_boundaries(R, k, wide):
    """k Position changes that each carry a dynamic entity into another section: around the limits of the one-launch path of small batches (256 writes / moved
    entities: k_apply_small vs the general path) and of the one-workgroup sort of the device re-bucket (2,048 ops = 1,024 movers: k_rb2_sort_small vs the radix sorts; up to 1,024 movers phases 1-3 are one launch, k_rb2_plan_small, IF the link ops of the batch fit the sort as well).
    wide: the entities are wider than a world section -- every move empties one shared section and creates another, each with link ops for the sections it links: at k = 600 far
    more than the 848 the one-workgroup sort has room for next to the 1,200 member ops, so the plan of k_rb2_plan_small is void and the batch is planned again by the kernels of large batches"""
    ents = hopping_world(R, dims=(16, 16, 16), first=120, every=2)
    if wide:
        for j, i in enumerate(np.nonzero((ents["flags"] & R.F_HAS_VEL) != 0)[0]):
            h = np.float32(20.0 + 6.0 * (j % 3)); ents["original"][i] = (-h, h, -h, h, -h, h)
    p, w = build_pair(R, ents)
    if wide: assert p.stats()["n_shared_sections"] > 600
    C = R._capi
    dyn = ents["id"][(ents["flags"] & R.F_HAS_VEL) != 0]
    assert len(dyn) >= 1025
    cam = R.Camera((8192, 8192, 8192 + 2200), (0, 0, -1), 5000.0)
    rng = np.random.default_rng(k)
    for rep in range(2):
        check_frame(R, p, w, cam, rep == 1)
        ids = rng.choice(dyn, k, replace=False)
        ch = np.zeros(k, R.CHANGE_DT)
        for i in range(k):
            ch[i] = (C.CHANGE_MODIFY, ids[i], C.C_POSITION, 0, (64.0 * (120 + int(rng.integers(0, 16))) + 32.0, 64.0 * (120 + int(rng.integers(0, 16))) + 32.0, 64.0 * (120 + int(rng.integers(0, 16))) + 32.0, 0))
        host_before = p.stats()["n_host_rebuckets"]
        n_a, oob_a = w.apply_changes(ch.view(ro.CHANGE_DT)); g = p.apply_changes(ch)
        assert g["n_changed"] == n_a == k and g["n_out_of_bounds"] == len(oob_a) == 0
        if not wide or rep == 0: assert p.stats()["n_host_rebuckets"] == host_before      # (wide: the second batch may find the spare slots of a level used up, a legitimate fallback)
        check_sections(p, w)
        if wide: assert p.stats()["n_shared_sections"] == w.L.ro_num_shared(w.h)
    check_entities(R, p, w, ents[::11])
    check_frame(R, p, w, cam, False)
    p.close(); w.close()


def test_one_launch_synchronous_frames(R):
    """RE_CULL_ONE_LAUNCH: the scan's last workgroup publishes the frame's result itself (k_scan_cull_sync: ticket over the workgroups, list entries written
    through to memory, one wave builds the InstanceRange table), k_pack_small only moves the instances.  Frames with movers, shared sections, duplicates
    mode, an empty view, and a visible set too large for the small pack (the tail declines, the frame is redone through the large path)."""
    ents = R.synthetic.mixed_world(3000, seed=33, spread=500.0)
    ents["vel"] *= 6.0
    p, w = build_pair(R, ents)
    cams = [R.Camera((8192 + 40 * i, 8192 - 25 * i, 8700 - 20 * i), (0.02 * i, 0, -1), 900.0 + 300.0 * (i % 3)) for i in range(8)]
    cams.append(R.Camera((200.0, 200.0, 200.0), (1, 0, 0), 50.0))                       # nothing in view
    for f, cam in enumerate(cams):
        oc = oracle_camera(cam)
        vis_o = w.cull(oc)
        g = p.cull_and_pack(cam, emit_duplicates=bool(f % 2), one_launch=True)
        keys, mult = p.visible_sections()
        np.testing.assert_array_equal(expand_vis(keys, mult), vis_o)
        assert_render_equal(g, w.render(oc, emit_duplicates=bool(f % 2)))
        assert_clean_publication(p)
        n_o, oob_o = w.tick(oc, 0.05); t = p.tick(0.05)
        assert t["n_changed"] == n_o and t["n_out_of_bounds"] == len(oob_o)
    check_sections(p, w)
    p.close(); w.close()
    many = R.synthetic.mixed_world(1500, seed=34, spread=400.0)                    # more group slots than the tail takes (40 models x 8 bands > 256): the flag is ignored, two launches
    many["model_index"] = np.arange(len(many), dtype=np.uint32) % 40
    p, w = build_pair(R, many)
    for f in range(2):
        cam = R.Camera((8192 + 20 * f, 8192, 8600), (0, 0, -1), 1200.0); oc = oracle_camera(cam)
        w.cull(oc); g = p.cull_and_pack(cam, one_launch=True)
        assert_render_equal(g, w.render(oc)); assert_clean_publication(p)
    p.close(); w.close()
    big = R.synthetic.lattice_world(cells_per_axis=40, first_cell=108)               # > 16 K visible instances: the small path declines
    p, w = build_pair(R, big)
    cam = R.Camera((8192, 8192, 10500), (0, 0, -1), 5000.0); oc = oracle_camera(cam)
    for f in range(3):
        w.cull(oc); g = p.cull_and_pack(cam, one_launch=True)
        o = w.render(oc)
        assert o["total"] > 16384
        assert_render_equal(g, o)
    p.close(); w.close()


@pytest.mark.parametrize("seed,tight", [(5, False), (6, True), (7, False)])
def test_device_rebucket_long_soak(R, seed, tight):
    """40 ticks of a world whose movers drift across section borders (about half of the placement changes involve a shared section): batch after batch
    on the device without the host looking at the table in between (holes of the shared table reused, retired ids looked up again, sections
    that exist through links only), interleaved with asynchronous frames (cancelled and replayed) and -- tight: hardly any slack -- with batches that
    find no room and fall back to the host path, which rebuilds the shared table compactly.  Compared with the oracle every 8 ticks and at the end."""
    rng = np.random.default_rng(seed)
    ents = R.synthetic.mixed_world(3000, seed=seed, spread=500.0)
    mv = (ents["flags"] & R.F_HAS_VEL) != 0
    ents["flags"][mv] &= ~np.uint32(R.F_STATIC)
    ents["vel"] *= 6.0
    p, w = build_pair(R, ents, flags=R._capi.CFG_TIGHT_SLACK if tight else 0)
    for f in range(40):
        cam = R.Camera((8192 + rng.uniform(-200, 200), 8192 + rng.uniform(-200, 200), 8800), (0, 0, -1), 2500.0)
        oc = oracle_camera(cam)
        if f % 4 == 3:
            check_frame(R, p, w, cam, False)
            n_o, oob_o = w.tick(oc, 0.05); t = p.tick(0.05)
            assert t["n_changed"] == n_o and t["n_out_of_bounds"] == len(oob_o)
        else:
            p.cull_and_pack(cam, asynchronous=True, copy=False); p.tick(0.05, asynchronous=True)
            w.cull(oc); w.render(oc); w.tick(oc, 0.05)
        if f % 8 == 7:
            p.wait(); check_sections(p, w)
    p.wait()
    st = p.stats()
    assert st["n_device_rebuckets"] >= (0 if tight else 30) and (st["n_host_rebuckets"] > 0 or not tight), st      # (tight: a table without slack -- every batch may end on the host path)
    check_sections(p, w)
    check_entities(R, p, w, ents[::9])
    check_frame(R, p, w, cam, True)
    p.close(); w.close()


@pytest.mark.parametrize("straddlers", [False, True])
def test_async_frames_with_the_device_rebucket(R, straddlers):
    """frames enqueued without waiting while every tick moves entities between world sections: each such tick cancels the frames behind it, the
    library patches the tree (on the device: the movers between unique sections) and replays them -- the end state equals the frame-by-frame reference"""
    ents = hopping_world(R, dims=(12, 12, 12), first=122)
    if straddlers:
        mv = np.nonzero((ents["flags"] & R.F_HAS_VEL) != 0)[0][::19]
        for k, i in enumerate(mv):
            h = np.float32(22.0 + 5.0 * (k % 3))
            ents["original"][i] = (-h, h, -h, h, -h, h)
    p, w = build_pair(R, ents)
    cams = [R.Camera((8192 + 12 * i, 8192 - 6 * i, 8192 + 2400), (0.01 * i, 0, -1), 6000.0) for i in range(6)]
    for cam in cams[:-1]:
        oc = oracle_camera(cam); w.cull(oc); w.render(oc); w.tick(oc, 1.0)
    for cam in cams[:-1]:
        p.cull_and_pack(cam, asynchronous=True, copy=False)
        p.tick(1.0, asynchronous=True)
    p.wait()
    st = p.stats()
    assert st["n_device_rebuckets"] >= 3, st
    check_sections(p, w)
    check_entities(R, p, w, ents[::5])
    check_frame(R, p, w, cams[-1], True)
    for i, cam in enumerate(cams):                                   # mixed styles
        oc = oracle_camera(cam); w.cull(oc); w.render(oc); w.tick(oc, 1.0)
        p.cull_and_pack(cam, asynchronous=(i % 3 != 0), copy=False)
        p.tick(1.0, asynchronous=(i % 2 == 0))
    p.wait()
    check_sections(p, w)
    check_entities(R, p, w, ents[::5])
    check_frame(R, p, w, cams[0], False)
    p.close(); w.close()


@pytest.mark.parametrize("dims,first,atomic,every", [((9, 17, 6), 200, 32, 3), ((20, 5, 11), 118, 64, 2), ((8, 8, 8), 60, 128, 2), ((6, 6, 6), 249, 64, 2)])
def test_device_rebucket_soak(R, dims, first, atomic, every):
    """more shapes for the device-side re-bucket: other section lengths, flat and slab-shaped worlds; the table is compared with the oracle after every tick"""
    ents = R.synthetic.hopping_lattice(dims, first, atomic, every)         # (the last shape sits at the edge of the world: movers hop out of bounds)
    ents["flags"][::5] |= np.uint32(R.F_LIGHT_POINT)
    p, w = build_pair(R, ents, atomic=atomic)
    c = (first + max(dims) / 2.0) * atomic
    for f in range(6):
        cam = R.Camera((c + 5.0 * f, c, c + 30.0 * atomic), (0.0, 0.0, -1.0), 90.0 * atomic)
        check_frame(R, p, w, cam, f % 2 == 0)
        n_o, oob_o = w.tick(oracle_camera(cam), 1.0)
        t = p.tick(1.0)
        assert t["n_changed"] == n_o and t["n_out_of_bounds"] == len(oob_o), (f, t, n_o, len(oob_o))
        assert sorted(p.out_of_bounds()) == sorted(int(i) for i in oob_o)
        check_sections(p, w)
        np.testing.assert_array_equal(p.visible_lights(cam, R.F_LIGHT_POINT), w.visible_lights(oracle_camera(cam), ro.F_LIGHT_POINT))     # lights follow the sections the device moved them to
    assert p.stats()["n_device_rebuckets"] >= 1, p.stats()
    check_entities(R, p, w, ents[::3])
    p.close(); w.close()


def crowded_world(R, n_crowd=150000, n_lattice=50000):
    """n_lattice static entities one per level-0 section + n_crowd entities (a third of them active) whose 300-unit boxes all fall into ONE level-3 world section"""
    C = R._capi
    lat = R.synthetic.lattice_world(cells_per_axis=37, first_cell=100)[:n_lattice]
    e = np.zeros(n_crowd, R.ENTITY_DT)
    idx = np.arange(n_crowd, dtype=np.uint64)
    e["id"] = (idx + np.uint64(1_000_000)).astype(np.uint32)
    e["model_index"] = (idx % np.uint64(5)).astype(np.uint32)
    e["flags"] = np.where(idx % np.uint64(3) == 0, 0, C.F_STATIC).astype(np.uint32)
    for k in range(3):
        e["pos"][:, k] = np.float32(8448.0) + np.float32(120.0) * (R.synthetic.uniform(77, idx, k) - np.float32(0.5))
        e["original"][:, 2 * k] = -150.0; e["original"][:, 2 * k + 1] = 150.0
    e["scale"] = 1.0; e["rot_axis"][:, 0] = 1; e["rotvel_axis"][:, 0] = 1; e["rotacc_axis"][:, 0] = 1
    return np.concatenate([lat, e])


def test_crowded_section_overflows_a_cursor_segment(R):
    """200,000 entities, 150,000 of them in one level-3 world section that both visibility queries return (duplicates mode: 300,000 instances from
    one reservation of one wave).  The instance list is eight cursor segments and a wave reserves in segment (wave index mod 8), so this frame cannot
    fit its segment: the pack kernels report it, the library redoes the frame with the list as one segment (re_stats.n_segment_redos) and keeps that
    layout -- the reference never fails a frame for where its entities sit (round 2 failed it with RE_E_CAPACITY)."""
    ents = crowded_world(R)
    p, w = build_pair(R, ents, max_instances=400000)
    s = p.sections()
    big = s["keys"][(s["keys"] >> np.uint64(48)) == 3]
    assert len(big) == 1 and int(s["n_local"][s["keys"] == big[0]][0]) == 50000 and int(s["n_static"][s["keys"] == big[0]][0]) == 100000
    cam = R.Camera((8680, 8680, 8690), (-0.57, -0.57, -0.57), 2000.0)          # within the logic culler's reach of the section's corner AND in front of the frustum
    for f, dups in enumerate((True, False, True)):
        g, o = check_frame(R, p, w, cam, dups)
        assert g["total"] >= (300000 if dups else 150000)
        n_o, _ = w.tick(oracle_camera(cam), 0.016); t = p.tick(0.016)
        assert t["n_changed"] == n_o
    st = p.stats()
    assert st["n_segment_redos"] == 1                                            # the first frame found out; the later ones start with one segment
    # the forced multi-kernel pack (count / scan / scatter) on the same world
    g, o = check_frame(R, p, w, cam, True, force_large_pack=True)
    p.close(); w.close()
    # the same world seen for the first time through the large pack path (the prediction of a fresh context is "small")
    p, w = build_pair(R, ents, max_instances=400000)
    g, o = check_frame(R, p, w, cam, True, force_large_pack=True)
    assert p.stats()["n_segment_redos"] == 1 and g["total"] >= 300000
    p.close(); w.close()


@pytest.mark.parametrize("n,atomic", [(1, 64), (2, 64), (3, 32), (7, 128), (63, 64), (64, 64), (65, 32), (100, 256), (512, 64), (513, 128), (1000, 16)])
def test_tiny_and_odd_worlds(R, n, atomic):
    """worlds of a handful of entities, section lengths from 16 to 256: every frame path (small pack, large pack forced, duplicates), ticks, the table"""
    ents = R.synthetic.mixed_world(n, seed=100 + n, spread=min(600.0, 40.0 + 6.0 * n), atomic=atomic)
    p, w = build_pair(R, ents, atomic=atomic)
    check_sections(p, w)
    cams = [R.Camera((8192 + 30 * i, 8192 - 20 * i, 8192 + 500 - 60 * i), (0.1 * i - 0.1, 0.05 * i, -1), 700.0 + 200.0 * i) for i in range(4)]
    for f, cam in enumerate(cams):
        check_frame(R, p, w, cam, f % 2 == 1, force_large_pack=(f == 2))
        n_o, oob_o = w.tick(oracle_camera(cam), 0.05)
        t = p.tick(0.05)
        assert t["n_changed"] == n_o and t["n_out_of_bounds"] == len(oob_o), (f, t, n_o)
        check_sections(p, w)
    check_entities(R, p, w, ents)
    p.close(); w.close()
