"""Spreading a world over several GPUs by contiguous ranges of the first-section key: every unique world section's entities, and every
shared section together with the unique section that caches its static entities, live on one shard -- also when the sections a shared section links
fall into different shards' key ranges.  The union of the shards' visible instances then equals the single-pipeline (and the oracle's) frame."""
import numpy as np
import pytest

import oracle as ro
from helpers import to_oracle, oracle_camera, assert_render_equal


def world(R, n=3000, seed=21):
    ents = R.synthetic.mixed_world(n, seed=seed, spread=600.0)
    ents["flags"] &= ~np.uint32(R.F_HAS_VEL | R.F_HAS_ACC | R.F_HAS_ROTVEL | R.F_HAS_ROTACC)     # nobody changes section: movers that cross shards are a separate matter
    return ents


def test_first_section_keys_and_shards_against_the_oracle_tree():
    import render_engine_amd as R
    ents = world(R)
    keys = R.first_section_keys(ents)
    w = ro.World(16384, 64); w.register(to_oracle(ents))
    n_shared = 0
    for e, k in zip(ents, keys):
        kind, ks = w.lookup(int(e["id"]))
        n_shared += kind == 2
        assert int(k) == (min(ks) if kind else 0), (int(e["id"]), kind, ks, int(k))
    assert n_shared > 500
    shards = R.shard_by_first_section(ents, 3)
    assert sorted(np.concatenate(shards).tolist()) == list(range(len(ents))) and min(len(s) for s in shards) > 800
    owner = {}
    for r, idx in enumerate(shards):                             # a first section never appears on two shards
        for k in keys[idx]:
            assert owner.setdefault(int(k), r) == r
    # the shard boundaries do cut through shared sections: some shared section links sections whose keys lie in different shards' ranges
    lo = [int(keys[idx].min()) for idx in shards]
    def range_of(key):
        return max(r for r in range(3) if lo[r] <= key)
    straddling = 0
    for sh in w.shared_sections():
        if len({range_of(k) for k in sh["keys"]}) > 1:
            straddling += 1
    assert straddling > 0
    w.close()


@pytest.mark.gpu
def test_union_of_shards_equals_the_single_pipeline_frame():
    import render_engine_amd as R
    ents = world(R)
    shards = R.shard_by_first_section(ents, 3, halo=True)
    full = R.Pipeline(16384, 64); full.register_model_instances(ents)
    w = ro.World(16384, 64); w.register(to_oracle(ents))
    parts = []; n_halo = 0
    for own, halo in shards:
        rep = ents[halo].copy(); rep["flags"] |= R.F_PHANTOM               # replicas of entities other shards own: in the tree, never drawn
        n_halo += len(halo)
        p = R.Pipeline(16384, 64); assert p.register_model_instances(np.concatenate([ents[own], rep])) == 0; parts.append(p)
    assert 0 < n_halo < len(ents)
    cams = [R.Camera((8192 + 40 * i, 8192 - 25 * i, 8700 - 60 * i), (0.05 * i - 0.1, 0.02 * i, -1), 900.0 + 150.0 * i) for i in range(5)]
    for f, cam in enumerate(cams):
        dups = f % 2 == 1
        oc = oracle_camera(cam); w.cull(oc); o = w.render(oc, emit_duplicates=dups)
        g = full.cull_and_pack(cam, emit_duplicates=dups)
        assert_render_equal(g, o)
        res = [p.cull_and_pack(cam, emit_duplicates=dups) for p in parts]
        assert sum(r["total"] for r in res) == g["total"], f
        # the union, group by group: (model + LOD, render system, sortable) -> ids, and every matrix bit for bit
        def keyed(r):
            out = []
            for grp in r["groups"]:
                b, c = int(grp["begin"]), int(grp["count"])
                for k in range(b, b + c):
                    out.append((int(grp["model_index"]), int(grp["render_system"]), int(grp["sortable"]), int(r["ids"][k]), r["mats"][k].view(np.uint32).tobytes()))
            return out
        want = sorted(keyed(g)); got = sorted(x for r in res for x in keyed(r))
        assert got == want, f
        for p in parts + [full]:
            p.tick(0.016)
        w.tick(oc, 0.016)
    for p in parts + [full]:
        st = p.stats(); assert st["n_seal_waits"] == 0 and st["n_sync_fallbacks"] == 0
        p.close()
    w.close()


def _keyed(r):
    out = []
    for grp in r["groups"]:
        b, c = int(grp["begin"]), int(grp["count"])
        for k in range(b, b + c):
            out.append((int(grp["model_index"]), int(grp["render_system"]), int(grp["sortable"]), int(r["ids"][k]), r["mats"][k].view(np.uint32).tobytes()))
    return out


@pytest.mark.gpu
def test_movers_that_cross_the_shard_boundary_migrate():
    """SURVEY 8e's second, sparse exchange: two pipelines own the two halves of the key space of a lattice in which every second entity hops one or two
    world sections per tick.  After every tick each pipeline hands over the entities whose section left its key range (re_list_migrants /
    re_export_entities / RE_CHANGE_DELETE) and registers the ones that arrived (re_add_entities).  Frame by frame the union of the two pipelines'
    visible instances -- and of their section tables -- equals the CPU oracle's single world."""
    import render_engine_amd as R
    from render_engine_amd import parallel
    ents = R.synthetic.hopping_lattice(dims=(12, 12, 12), first_cell=122, every=2)
    keys = R.first_section_keys(ents)
    cut = np.sort(keys)[len(keys) // 2]
    ranges = parallel.key_ranges_from_cuts([cut])
    w = ro.World(16384, 64); w.register(to_oracle(ents))
    parts = []
    for lo, hi in ranges:
        p = R.Pipeline(16384, 64)
        assert p.register_model_instances(ents[(keys >= np.uint64(lo)) & (keys < np.uint64(hi))]) == 0
        p.set_shard_range(lo, hi); parts.append(p)
    c = (122 + 6) * 64.0
    cams = [R.Camera((c + 20 * i, c, c + 500 - 30 * i), (0, 0, -1), 1500.0) for i in range(6)]
    moved_total = 0
    for f, cam in enumerate(cams):
        dups = f % 2 == 1
        oc = oracle_camera(cam); w.cull(oc); o = w.render(oc, emit_duplicates=dups)
        res = [p.cull_and_pack(cam, emit_duplicates=dups) for p in parts]
        assert sum(r["total"] for r in res) == o["total"], f
        assert sorted(x for r in res for x in _keyed(r)) == sorted(_keyed(o)), f
        n_o, _ = w.tick(oc, 1.0)
        assert sum(p.tick(1.0)["n_changed"] for p in parts) == n_o
        # the sparse exchange, between frames
        states = [p.take_migrants() for p in parts]
        for r, p in enumerate(parts):
            inc = [states[q][parallel.route_migrants(states[q], ranges)[r]] for q in range(len(parts)) if q != r]
            inc = np.concatenate(inc) if inc else np.zeros(0, R.ENTITY_DT)
            if len(inc):
                assert p.register_model_instances(inc) == 0
        moved_total += sum(len(s) for s in states)
        # the two section tables together are the oracle's tree: keys, member counts, tight AABBs
        secs = [p.sections() for p in parts]
        cells = w.cells()
        k_all = np.concatenate([s["keys"] for s in secs]); order = np.argsort(k_all, kind="stable")
        np.testing.assert_array_equal(k_all[order], cells["keys"])
        np.testing.assert_array_equal(np.concatenate([s["n_local"] for s in secs])[order], cells["n_local"])
        np.testing.assert_array_equal(np.concatenate([s["n_static"] for s in secs])[order], cells["n_static"])
        tight = np.stack([cells["tight"][k] for k in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")], axis=1)
        np.testing.assert_array_equal(np.concatenate([s["tight"] for s in secs])[order], tight)
        for (lo, hi), s in zip(ranges, secs):                      # and every section sits on the pipeline that owns its key
            assert np.all((s["keys"] >= np.uint64(lo)) & (s["keys"] < np.uint64(hi)))
    assert moved_total > 50
    for p in parts:
        st = p.stats(); assert st["n_seal_waits"] == 0 and st["n_sync_fallbacks"] == 0
        p.close()
    w.close()
