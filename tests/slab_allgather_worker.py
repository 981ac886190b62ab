"""Worker of test_parallel_gpu.py: 2 ranks (gloo, sharing GPU 0) run the sharded frame loop with SlabAllGather and compare the
gathered visible set with a single-pipeline run over the whole world."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import render_engine_amd as R
from render_engine_amd import parallel, synthetic


def main():
    dist.init_process_group(os.environ.get("RE_TEST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dims, first, atomic = (24, 24, 24), 116, 64
    total = dims[0] * dims[1] * dims[2]
    lo, hi = parallel.shard_bounds(total, world)[rank]
    mine = synthetic.box_world(dims, first_cell=first, atomic=atomic, index_range=(lo, hi), spinner_every=9)
    p = R.Pipeline(16384, atomic, max_instances=1 << 14)
    p.register_model_instances(mine)
    full = R.Pipeline(16384, atomic, max_instances=1 << 14)
    full.register_model_instances(synthetic.box_world(dims, first_cell=first, atomic=atomic, spinner_every=9))
    g = parallel.SlabAllGather(p, 2048, dist)
    cams = [R.Camera((8192 + 20 * i, 8192, 8500 - 25 * i), (0, 0, -1), 900.0) for i in range(6)]
    for cam in cams:                                  # everything enqueued, nothing awaited
        g.begin_frame()
        p.cull_and_pack(cam, asynchronous=True, copy=False)
        b = g.exchange()
        p.tick(0.016, asynchronous=True)
    p.wait(); g.finish()
    ids, mats, counts = g.gathered(b)
    import oracle as ro
    from helpers import to_oracle, oracle_camera, assert_render_equal
    w = ro.World(16384, atomic); w.register(to_oracle(synthetic.box_world(dims, first_cell=first, atomic=atomic, spinner_every=9)))
    for cam in cams:
        ref = full.cull_and_pack(cam)
        oc = oracle_camera(cam); w.cull(oc); o = w.render(oc)
        assert_render_equal(ref, o)                     # the single pipeline the gathered set is compared with equals the CPU oracle, frame by frame
        full.tick(0.016); w.tick(oc, 0.016)
    w.close()
    assert sum(counts) == ref["total"], (counts, ref["total"])
    got = np.sort(ids.cpu().numpy().astype(np.uint32)); want = np.sort(ref["ids"])
    np.testing.assert_array_equal(got, want)
    o1 = np.argsort(ids.cpu().numpy().astype(np.uint32), kind="stable"); o2 = np.argsort(ref["ids"], kind="stable")
    np.testing.assert_array_equal(mats.cpu().numpy()[o1], ref["mats"][o2])
    p.close(); full.close()
    # the same with the one-launch-per-frame loop of a static world: packs deferred into the next frame's launch, exchange one frame late
    mine = synthetic.box_world(dims, first_cell=first, atomic=atomic, index_range=(lo, hi))
    p = R.Pipeline(16384, atomic, max_instances=1 << 14); p.register_model_instances(mine)
    full = R.Pipeline(16384, atomic, max_instances=1 << 14); full.register_model_instances(synthetic.box_world(dims, first_cell=first, atomic=atomic))
    g = parallel.SlabAllGather(p, 2048, dist)
    for cam in cams:
        g.begin_frame()
        p.cull_and_pack(cam, asynchronous=True, copy=False, defer_pack=True)
        g.exchange_lagged()
        p.tick(0.016, asynchronous=True)
    p.wait(); b = g.finish_lagged()
    ids, mats, counts2 = g.gathered(b)
    for cam in cams:
        ref = full.cull_and_pack(cam); full.tick(0.016)
    assert sum(counts2) == ref["total"], (counts2, ref["total"])
    np.testing.assert_array_equal(np.sort(ids.cpu().numpy().astype(np.uint32)), np.sort(ref["ids"]))
    o1 = np.argsort(ids.cpu().numpy().astype(np.uint32), kind="stable"); o2 = np.argsort(ref["ids"], kind="stable")
    np.testing.assert_array_equal(mats.cpu().numpy()[o1], ref["mats"][o2])
    st = p.stats()
    assert st["n_fused_frames"] >= 1, ("fused frames", st["n_fused_frames"], st["n_table_rebuilds"], st["n_seal_waits"], st["n_sync_fallbacks"])   # (how many launches carried a pack depends on how early the size hint of an asynchronous frame lands)
    # and with two frame lanes: frames alternate between two streams, the slab of frame g - 2 goes out behind launch g
    p.close()
    p = R.Pipeline(16384, atomic, max_instances=1 << 14); p.register_model_instances(mine)
    g2 = parallel.SlabAllGatherLanes(p, 2048, dist)
    cams2 = cams + [R.Camera((8192 + 11 * i, 8192 - 7 * i, 8450 - 13 * i), (0.01 * i, 0, -1), 900.0) for i in range(9)]
    for cam in cams2:
        g2.begin_frame()
        p.cull_and_pack(cam, asynchronous=True, copy=False, defer_pack=True, two_lanes=True)
        g2.after_cull()
        p.tick(0.016, asynchronous=True)
    p.wait(); b = g2.finish()
    ids, mats, counts3 = g2.gathered(b)
    for cam in cams2[len(cams):]:
        ref = full.cull_and_pack(cam); full.tick(0.016)
    assert sum(counts3) == ref["total"], (counts3, ref["total"])
    np.testing.assert_array_equal(np.sort(ids.cpu().numpy().astype(np.uint32)), np.sort(ref["ids"]))
    o1 = np.argsort(ids.cpu().numpy().astype(np.uint32), kind="stable"); o2 = np.argsort(ref["ids"], kind="stable")
    np.testing.assert_array_equal(mats.cpu().numpy()[o1], ref["mats"][o2])
    ids_prev, _, counts_prev = g2.gathered((b - 1) % g2.DEPTH)             # the frame before the last one went through the other lane
    st = p.stats()
    assert sum(counts_prev) > 0 and st["reserved"] >= 2, ("previous frame / lane switches", counts_prev, st["reserved"], st["n_fused_frames"])
    dist.barrier()
    if rank == 0:
        print("OK slab all-gather", counts, counts2, counts3)
    dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except AssertionError as e:                      # one short line the parent test can find in front of the launcher's own traceback
        import traceback
        tb = traceback.extract_tb(e.__traceback__)[-1]
        print("WORKER-FAIL rank %s line %d: %s" % (os.environ.get("RANK"), tb.lineno, str(e)[:600]), flush=True)
        raise
