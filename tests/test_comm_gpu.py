"""The multi-GPU exchange behind the C ABI (re_comm_*, re_allgather_visible) on the one GPU of the test box: a communicator of ONE rank runs the
whole protocol through RCCL -- the stream-ordered slab all-gather, the second, variable-length round when the visible set outgrows the slab, and the
re-gather of a frame that cross-frame speculation had cancelled; every gathered buffer is compared with the CPU ORACLE's frame (ids and matrices, bit for bit).  (Two ranks cannot share one GPU under RCCL; the N > 1 logic of the harness in
render_engine_amd/parallel.py is covered by the world_size-2 gloo tests on CPU, the sharded frame loop by tests/test_parallel_gpu.py.)"""
import numpy as np
import pytest

import oracle as ro
from helpers import to_oracle, oracle_camera

pytestmark = pytest.mark.gpu


def oracle_pair(ents):
    w = ro.World(16384, 64); w.register(to_oracle(ents))
    return w


def oracle_frame(w, cam, dt):
    """one frame of the CPU oracle (the checker): the visible instances the gathered buffer must hold, then the tick"""
    oc = oracle_camera(cam)
    w.cull(oc); o = w.render(oc)
    w.tick(oc, dt)
    return o


@pytest.fixture(scope="module")
def R():
    import render_engine_amd as R
    return R


def by_id(ids, mats):
    o = np.argsort(ids, kind="stable")
    return ids[o], mats[o]


@pytest.mark.parametrize("slab,expect_overflow", [(4096, False), (64, True)])
def test_single_rank_exchange_matches_the_local_frame(R, slab, expect_overflow):
    ents = R.synthetic.lattice_world(cells_per_axis=24, first_cell=116, spinner_every=9)
    p = R.Pipeline(16384, 64)
    p.register_model_instances(ents)
    ref = oracle_pair(ents)
    p.comm_init(R.Pipeline.comm_unique_id(), 0, 1, slab)
    for f in range(4):
        cam = R.Camera((8192 + 15 * f, 8192, 8500 - 20 * f), (0, 0, -1), 900.0)
        g = p.cull_and_pack(cam, copy=False)
        got = p.allgather_visible()
        p.tick(0.016)
        want = oracle_frame(ref, cam, 0.016)
        assert got["counts"] == [want["total"]] and got["overflowed"] == expect_overflow and want["total"] > 64
        a, b = by_id(got["ids"], got["mats"]), by_id(want["ids"], want["mats"])
        np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1])
        assert g["total"] == want["total"]
    p.close(); ref.close()


def test_large_pack_through_the_second_round(R):
    """a visible set beyond the small pack's limit AND beyond the slab: the second round packs again through k_pack_large"""
    ents = R.synthetic.lattice_world(cells_per_axis=40, first_cell=108)
    p = R.Pipeline(16384, 64); p.register_model_instances(ents)
    ref = oracle_pair(ents)
    p.comm_init(R.Pipeline.comm_unique_id(), 0, 1, 1000)
    cam = R.Camera((8192, 8192, 10500), (0, 0, -1), 5000.0)
    for f in range(3):
        p.cull_and_pack(cam, copy=False); got = p.allgather_visible(); p.tick(0.016)
        want = oracle_frame(ref, cam, 0.016)
        assert want["total"] > 16384 and got["overflowed"] and got["counts"] == [want["total"]]
        a, b = by_id(got["ids"], got["mats"]), by_id(want["ids"], want["mats"])
        np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1])
    p.close(); ref.close()


def test_cancelled_frames_are_gathered_again(R):
    """asynchronous frames of a world with movers: a tick that finds section changes cancels the frames enqueued behind it; their slabs carry the
    cancel marker, and re_gather_wait replays and gathers again -- the result equals the CPU oracle's frame"""
    ents = R.synthetic.mixed_world(3000, seed=21, spread=400.0)
    ents["vel"] *= 10.0
    p = R.Pipeline(16384, 64); p.register_model_instances(ents)
    ref = oracle_pair(ents)
    p.comm_init(R.Pipeline.comm_unique_id(), 0, 1, 8192)
    cams = [R.Camera((8192 + 30 * f, 8192, 8600 - 10 * f), (0, 0, -1), 1000.0) for f in range(6)]
    replays = 0
    for cam in cams:
        p.cull_and_pack(cam, asynchronous=True, copy=False)
        p.allgather_visible(asynchronous=True)
        p.tick(0.05, asynchronous=True)
        got = p.gather_wait()
        want = oracle_frame(ref, cam, 0.05)
        assert got["counts"] == [want["total"]]
        a, b = by_id(got["ids"], got["mats"]), by_id(want["ids"], want["mats"])
        np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1])
    p.wait()
    p.close(); ref.close()
