"""N>1 path on CPU: world_size-2 gloo.  Each rank owns a contiguous section-key range of the world,
computes its visible set (here with the CPU oracle as the per-shard stand-in -- the product's
exchange code is device-agnostic), and the variable-length all-gather must reproduce the
single-process result in rank order."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS, FIRST = (24, 12, 12), 118


def _camera():
    from render_engine_amd import Camera
    from helpers import oracle_camera
    return oracle_camera(Camera(((FIRST + 12) * 64.0, (FIRST + 6) * 64.0, (FIRST + 6) * 64.0 + 200), (0, 0, -1), 1000.0))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as ro
    from render_engine_amd import synthetic, parallel
    from helpers import to_oracle
    n_total = DIMS[0] * DIMS[1] * DIMS[2]
    lo, hi = parallel.shard_bounds(n_total, world)[rank]
    ents = synthetic.box_world(DIMS, first_cell=FIRST, index_range=(lo, hi), spinner_every=9)
    w = ro.World(16384, 64); w.register(to_oracle(ents))
    cam = _camera()
    w.cull(cam); r = w.render(cam)
    cap = 4096
    ids = torch.zeros(cap, dtype=torch.int32); mats = torch.zeros(cap, 16)
    ids[:r["total"]] = torch.from_numpy(r["ids"].astype(np.int32)); mats[:r["total"]] = torch.from_numpy(r["mats"])
    ids_all, mats_all, counts = parallel.allgather_packed(ids, mats, r["total"], dist)
    # ---- the fixed-slab exchange of the frame loop (SlabAllGather / re_allgather_visible): header {written, total, frame, 0} + ids + matrices
    slab_res = {}
    for name, scap in (("fits", 1024), ("overflow", 100)):
        slab = parallel.fill_slab(torch.zeros(parallel.slab_words(scap), dtype=torch.int32), ids, mats, r["total"], frame=7)
        recv = torch.zeros(parallel.slab_words(scap) * world, dtype=torch.int32)
        dist.all_gather_into_tensor(recv, slab)
        try:
            si, sm, sc = parallel.parse_slabs(recv, world, scap)
            slab_res[name] = ("ok", sc, si.numpy().copy(), sm.numpy().copy())
        except parallel.SlabOverflow as e:                # every rank reads the same headers: all of them fall back to the variable-length gather
            fi, fm, fc = parallel.allgather_packed(ids, mats, r["total"], dist)
            slab_res[name] = ("overflow", e.totals, fi.numpy().copy(), fm.numpy().copy())
    # a frame that cross-frame speculation cancelled on rank 1: the marker in its header makes every rank gather again after the replay
    slab = parallel.fill_slab(torch.zeros(parallel.slab_words(1024), dtype=torch.int32), ids, mats, r["total"])
    if rank == 1:
        slab[0] = parallel.SLAB_CANCELLED
    recv = torch.zeros(parallel.slab_words(1024) * world, dtype=torch.int32)
    dist.all_gather_into_tensor(recv, slab)
    try:
        parallel.parse_slabs(recv, world, 1024); cancelled = None
    except parallel.SlabCancelled as e:
        cancelled = e.args[0]
        slab = parallel.fill_slab(slab, ids, mats, r["total"])            # (the replay packs into the same slab)
        dist.all_gather_into_tensor(recv, slab)
        slab_res["regather"] = ("ok",) + tuple(x if isinstance(x, list) else x.numpy().copy() for x in [parallel.parse_slabs(recv, world, 1024)[k] for k in (2, 0, 1)])
    q.put((rank, r["total"], counts, ids_all.numpy().copy(), mats_all.numpy().copy(), slab_res, cancelled))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds():
    from render_engine_amd import parallel
    assert parallel.shard_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert parallel.shard_bounds(8, 8)[-1] == (7, 8)
    b = parallel.shard_bounds(10077696 * 8, 8)
    assert all(hi - lo == 10077696 for lo, hi in b)


@pytest.mark.timeout(180)
def test_two_rank_allgather_matches_single_process():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as ro
    from render_engine_amd import synthetic
    from helpers import to_oracle
    world = 2
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(30); assert p.exitcode == 0
    ents = synthetic.box_world(DIMS, first_cell=FIRST, spinner_every=9)          # single-process result over the whole world
    w = ro.World(16384, 64); w.register(to_oracle(ents))
    cam = _camera()
    w.cull(cam); full = w.render(cam)
    counts = res[0][2]
    assert counts == [res[0][1], res[1][1]] and sum(counts) == full["total"] and min(counts) > 0
    for rank, n, cts, ids_all, mats_all, slab_res, cancelled in res:
        assert cts == counts
        np.testing.assert_array_equal(ids_all, res[0][3])            # every rank ends with the same buffer
        np.testing.assert_array_equal(mats_all, res[0][4])
        # the slab exchange: a slab that fits reproduces the variable-length result; one that does not makes every rank see the overflow in the
        # headers (true totals) and fall back; a cancelled frame is detected on every rank and gathered again
        kind, sc, si, sm = slab_res["fits"]
        assert kind == "ok" and sc == counts
        np.testing.assert_array_equal(si.astype(np.uint32), res[0][3].astype(np.uint32)); np.testing.assert_array_equal(sm, res[0][4])
        kind, totals, fi, fm = slab_res["overflow"]
        assert kind == "overflow" and totals == counts and max(counts) > 100
        np.testing.assert_array_equal(fi, res[0][3]); np.testing.assert_array_equal(fm, res[0][4])
        assert cancelled == [1]
        kind, rc, ri, rm = slab_res["regather"]
        assert rc == counts
        np.testing.assert_array_equal(ri.astype(np.uint32), res[0][3].astype(np.uint32)); np.testing.assert_array_equal(rm, res[0][4])
    ids_all, mats_all = res[0][3].astype(np.uint32), res[0][4]
    assert set(ids_all.tolist()) == set(full["ids"].tolist()) and len(ids_all) == full["total"]
    o1, o2 = np.argsort(ids_all), np.argsort(full["ids"])
    np.testing.assert_array_equal(mats_all[o1], full["mats"][o2])
    half = DIMS[0] * DIMS[1] * DIMS[2] // 2
    assert np.all(ids_all[:counts[0]] < half) and np.all(ids_all[counts[0]:] >= half)      # rank order


# ---- the frame's second, sparse exchange: movers that cross the shard boundary -----------------------------------------------------
class OracleShard:
    """one rank's share of the world with the interface parallel.exchange_migrants drives (take_migrants / register_model_instances): the CPU oracle as
    the per-shard stand-in, as everywhere in this file -- the exchange code is device-agnostic"""

    def __init__(self, ents, key_range):
        import oracle as ro
        from helpers import to_oracle
        self.ro, self.to_oracle = ro, to_oracle
        self.w = ro.World(16384, 64); self.w.register(to_oracle(ents))
        self.rec = {int(e["id"]): e.copy() for e in ents}
        self.lo, self.hi = key_range

    def take_migrants(self):
        from render_engine_amd import ENTITY_DT
        import render_engine_amd as R
        out = []
        for eid in sorted(self.rec):
            kind, ks = self.w.lookup(eid)
            if not kind or self.lo <= min(ks) < self.hi:
                continue
            st = self.w.entity(eid); e = self.rec.pop(eid)
            e["pos"] = st["pos"]; e["vel"] = st["vel"]; e["flags"] = (st["flags"] & ~np.uint32(R.F_STATIC))      # the components as they are now
            out.append(e)
        if out:
            ch = np.zeros(len(out), self.ro.CHANGE_DT); ch["kind"] = 1; ch["entity_id"] = [int(e["id"]) for e in out]      # DeleteRequest
            self.w.apply_changes(ch, end_of_frame=False)
        return np.array(out, ENTITY_DT) if out else np.zeros(0, ENTITY_DT)

    def register_model_instances(self, ents):
        for e in ents:
            self.rec[int(e["id"])] = e.copy()
        return self.w.register(self.to_oracle(ents))


def _migration_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import render_engine_amd as R
    from render_engine_amd import synthetic, parallel
    from helpers import oracle_camera
    ents = synthetic.hopping_lattice(dims=(10, 10, 10), first_cell=123, every=2)
    keys = R.first_section_keys(ents)
    cut = np.sort(keys)[len(keys) // 2]
    ranges = parallel.key_ranges_from_cuts([cut])
    lo, hi = ranges[rank]
    sh = OracleShard(ents[(keys >= np.uint64(lo)) & (keys < np.uint64(hi))], (lo, hi))
    c = (123 + 5) * 64.0
    cam = oracle_camera(R.Camera((c, c, c + 450), (0, 0, -1), 1400.0))
    frames = []
    for f in range(5):
        sh.w.cull(cam); r = sh.w.render(cam)
        n, _ = sh.w.tick(cam, 1.0)
        sent, got = parallel.exchange_migrants(sh, dist, ranges)
        cells = sh.w.cells()
        frames.append((np.sort(r["ids"]).copy(), n, sent, got, cells["keys"].copy(), cells["n_local"].copy()))
    q.put((rank, frames))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_migration_of_shard_crossing_movers():
    """two gloo ranks, each with half of the key space of a lattice whose movers hop one or two world sections per tick: after every tick
    parallel.exchange_migrants hands the entities whose section left a rank's range to the rank that owns it.  Frame by frame the two ranks' visible
    ids, ticked counts and section tables add up to the single-process oracle world."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as ro
    import render_engine_amd as R
    from render_engine_amd import synthetic
    from helpers import to_oracle, oracle_camera
    world = 2
    port = 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_migration_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(30); assert p.exitcode == 0
    ents = synthetic.hopping_lattice(dims=(10, 10, 10), first_cell=123, every=2)
    w = ro.World(16384, 64); w.register(to_oracle(ents))
    c = (123 + 5) * 64.0
    cam = oracle_camera(R.Camera((c, c, c + 450), (0, 0, -1), 1400.0))
    moved = 0
    for f in range(5):
        w.cull(cam); r = w.render(cam); n, _ = w.tick(cam, 1.0)
        a, b = res[0][f], res[1][f]
        np.testing.assert_array_equal(np.sort(np.concatenate([a[0], b[0]])), np.sort(r["ids"]))
        assert a[1] + b[1] == n
        assert a[2] == b[3] and b[2] == a[3]                       # what one rank sent, the other received
        moved += a[2] + b[2]
        cells = w.cells()
        k = np.concatenate([a[4], b[4]]); o = np.argsort(k)
        np.testing.assert_array_equal(k[o], cells["keys"])
        np.testing.assert_array_equal(np.concatenate([a[5], b[5]])[o], cells["n_local"])
    assert moved > 20
    w.close()
