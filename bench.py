#!/usr/bin/env python3
"""bench.py -- entities/sec through cull + transform on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over the resident world: re_cull_pack (visibility query over the
spatial hash + frustum/logic cull + instance pack) followed by re_tick (ECS kinematic / TRS->mat4
tick) -- Pipeline::execute of the reference (flows/pipeline.rs:212-276).  Inputs are resident in HBM
before the timed region starts.  N=1: BASELINE.json configs[1] (10,077,696 static entities, one
per level-0 world section, uniform spatial-hash fill, one camera frustum).  N>1 (weak scaling):
N x 10,077,696 entities sharded by contiguous section-key range, one process per GPU, with an RCCL
all-gather of every GPU's packed visible-instance buffer each step (configs[3] at N=8).
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TIMING_EVERY = 8
SLAB_INSTANCES = 4096           # all-gather slab per rank and frame (instances); ~2.8x the visible set of a rank at far=1000
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
PER_GPU_AXIS = 216             # 216^3 = 10,077,696 sections/entities per GPU


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--far", type=float, default=1000.0, help="camera far draw distance (reference default 1000, main.rs:23)")
    ap.add_argument("--axis", type=int, default=PER_GPU_AXIS, help="sections per axis per GPU (216 -> 10,077,696 entities)")
    ap.add_argument("--spinner-every", type=int, default=0, help="0 = configs[1] (all static); 100 = configs[2] (100k rotating bodies)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-large-pack", action="store_true", help="always use the multi-kernel pack")
    ap.add_argument("--cpu-sample-axis", type=int, default=100)
    ap.add_argument("--no-defer-pack", action="store_true", help="launch every frame's pack on its own instead of letting the next frame's launch carry it (RE_CULL_DEFER_PACK)")
    ap.add_argument("--one-lane", action="store_true", help="keep all frames on one HIP stream (no RE_CULL_TWO_LANES)")
    ap.add_argument("--probe", action="store_true", help="opt-in variant: visibility query by hash probes of the candidate cells (RE_CFG_PROBE) instead of the key stream")
    return ap.parse_args()


def world_dims(n_gpus, axis):
    """N x axis^3 sections as a box whose x extent grows first (key order is x-major)."""
    f = [1, 1, 1]
    n, i = n_gpus, 0
    while n > 1:
        if n % 2:
            raise SystemExit("--gpus must be a power of two")
        f[i % 3] *= 2; n //= 2; i += 1
    return axis * f[0], axis * f[1], axis * f[2]     # nx, nz, ny


def make_shard(rank, n_gpus, axis, atomic, spinner_every):
    """Entities of this rank: a contiguous range of the x-major section index space."""
    from render_engine_amd import synthetic
    nx, nz, ny = world_dims(n_gpus, axis)
    total = nx * nz * ny
    per = total // n_gpus
    lo, hi = rank * per, (rank + 1) * per if rank < n_gpus - 1 else total
    first = (16384 // atomic - max(nx, nz, ny)) // 2
    return synthetic.box_world((nx, nz, ny), first_cell=first, atomic=atomic, index_range=(lo, hi), spinner_every=spinner_every), (nx, nz, ny), first


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # RE_BENCH_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks (all ranks share device 0)
    backend = os.environ.get("RE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    import render_engine_amd as R
    from render_engine_amd import parallel

    atomic = 64 if world == 1 else 32            # 432 sections per axis need atomic 32
    t0 = time.time()
    ents, dims, first = make_shard(rank, world, a.axis, atomic, a.spinner_every)
    n_local = len(ents)
    cap = 1 << 16 if a.far <= 2000 else max(1 << 16, n_local // 2)
    p = R.Pipeline(16384, atomic, device=local, max_instances=cap, flags=R._capi.CFG_PROBE if a.probe else 0)
    p.register_model_instances(ents)
    stats = p.stats()
    del ents
    t_setup = time.time() - t0
    centre = [(first + d / 2.0) * atomic for d in (dims[0], dims[2], dims[1])]      # x, y, z
    cam = R.Camera(centre, (0.0, 0.0, -1.0), a.far)
    camc = cam.to_c()
    # N > 1: every rank packs into a fixed slab and the slabs are all-gathered in stream order, double-buffered -- no host round trip
    lanes = not (a.one_lane or a.no_defer_pack)
    gather = None
    if world > 1:
        gather = parallel.SlabAllGatherLanes(p, SLAB_INSTANCES, dist) if lanes else parallel.SlabAllGather(p, SLAB_INSTANCES, dist)

    def step(sync_each):
        if gather is not None:
            gather.begin_frame()
            p.cull_and_pack(camc, asynchronous=True, copy=False, force_large_pack=a.force_large_pack, defer_pack=not a.no_defer_pack, two_lanes=lanes)
            if lanes:
                gather.after_cull()            # frames alternate between two streams; the slab of frame g - 2 goes out behind launch g
            elif a.no_defer_pack:
                gather.exchange()
            else:
                gather.exchange_lagged()       # this launch carried the previous frame's pack: that frame's slab goes out now
            p.tick(0.016, asynchronous=True)
            if sync_each:
                p.wait()
                if lanes or a.no_defer_pack: gather.finish()
                else: gather.finish_lagged()
        else:
            # asynchronous frames of a static world leave their pack to the next frame's launch (one launch per frame); the last one is
            # sent off by the fence.  Worlds with dynamic entities pack every frame before its tick (the library ignores the flag there).
            p.cull_and_pack(camc, asynchronous=not sync_each, copy=False, force_large_pack=a.force_large_pack, defer_pack=not a.no_defer_pack, two_lanes=lanes)
            p.tick(0.016, asynchronous=not sync_each)

    def fence():
        p.wait()
        if gather is not None:
            if lanes or a.no_defer_pack: gather.finish()
            else: gather.finish_lagged()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(a.warmup):
        step(False)
    fence()
    fused_before = p.stats()["n_fused_frames"]
    p.timing_begin(a.steps, every=TIMING_EVERY)     # HIP events on every 8th launch of the dominant kernel: timed dispatches cost queue time
    t_start = time.perf_counter()
    for _ in range(a.steps):
        step(False)
    fence()
    elapsed = time.perf_counter() - t_start
    k1_us = p.timing_collect(a.steps)
    vis, _ = p.wait()
    n_cand = p.last_candidates()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); elapsed = float(t.item())
        tot = torch.tensor([n_local], dtype=torch.int64, device="cuda"); dist.all_reduce(tot); n_total = int(tot.item())
    else:
        n_total = n_local

    # second timing leg, outside the timed region: the scan kernel on its own (every frame's pack launched separately), so that the
    # roofline of the scan can be read next to that of the fused launch the timed region runs
    scan_only_us = None
    fused = not a.no_defer_pack and (p.stats()["n_fused_frames"] - fused_before) * 2 >= a.steps    # the launches of the timed region carried the packs
    if fused and rank == 0 and gather is None:
        nleg = 96
        p.timing_begin(nleg, every=TIMING_EVERY)
        for _ in range(nleg):
            p.cull_and_pack(camc, asynchronous=True, copy=False, force_large_pack=a.force_large_pack, defer_pack=False); p.tick(0.016, asynchronous=True)
        p.wait()
        scan_only_us = p.timing_collect(nleg)

    # per-frame latency with a host sync after every call (what a frame loop that draws each frame sees)
    lat = []
    for _ in range(min(50, a.steps)):
        t1 = time.perf_counter(); step(True); lat.append(time.perf_counter() - t1)     # both calls return with their results complete
    p.timings_us(); step(True)                       # kernel event timing is off until asked for: switch it on, time one more synchronous frame
    tm = p.timings_us()

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        # roofline of the dominant kernel (k_scan_cull): algorithmic bytes per launch, see DESIGN.md "Roofline accounting":
        # the stream key of every section slot (4 B compact when the world has <= 512 sections per axis, else 8 B) + one level word per
        # 512-key chunk + 41 B per visible section (flags, tight AABB, counts, begin read; stamp written) + 16 B per visible instance
        # (row index and group class read, instance-list entry written).  Candidate sections that turn out invisible cost no bytes:
        # their test runs on the key alone.
        C_sections = stats["n_section_slots"] if "n_section_slots" in stats else stats["n_sections"]
        n_entries = vis["n_visible_sections"]
        key_bytes = 4 if (16384 + atomic - 1) // atomic <= 512 and not os.environ.get("RE_EXP_KEY64") else 8
        alg_bytes = key_bytes * C_sections + 4 * ((C_sections + 511) // 512) + 41 * n_entries + 16 * vis["total"]
        probed = a.probe and p.stats()["n_probe_frames"] > 0
        if probed:      # the probe kernel reads one 16-byte table entry per candidate cell instead of the key stream
            alg_bytes = 16 * int(p.last_candidates()) + 41 * n_entries + 16 * vis["total"]
        k1_mean = float(np.mean(k1_us)) if len(k1_us) else float("nan")
        achieved = alg_bytes / (k1_mean * 1e-6) / 1e9 if k1_mean > 0 else None
        traffic = None
        prof = os.path.join(ROOT, "profiles", "r01_pmc_k_scan_cull.json")
        # the committed PMC measurement is of the default workload on one GPU; other workloads report null
        default_workload = world == 1 and a.axis == 216 and a.far == 1000.0 and not a.spinner_every and not a.force_large_pack and not a.probe
        if default_workload and os.path.exists(prof):
            try:
                traffic = json.load(open(prof)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "entities/sec through cull+transform; visible-set ms/frame @ 10M entities",
            "value": n_total / (elapsed / a.steps), "unit": "entities/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("configs[1]: %d static entities, one per level-0 world section (uniform spatial-hash fill), 1 camera frustum far=%g"
                                    % (n_total, a.far)) if not a.spinner_every else
                                   ("configs[2]: %d entities incl. every %dth rotating (ECS tick + cull), far=%g" % (n_total, a.spinner_every, a.far)),
                       "entities": n_total, "sections": C_sections * world, "dynamic_entities": stats["n_dynamic"] * world,
                       "visible_sections": vis["n_visible_sections"], "visible_instances": vis["total"], "far": a.far,
                       "sharding": "none" if world == 1 else "contiguous section-key ranges; per frame one RCCL all_gather_into_tensor of fixed %d-instance slabs (count header written by the pack kernel), slabs in rotation, stream-ordered behind the launch that carries the pack" % SLAB_INSTANCES},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel": "k_probe_cull" if probed else (("k_scan_cull_fused (scan of frame f + pack of frame f-2 in one launch; frames alternate between two streams)" if lanes else "k_scan_cull_fused (scan of frame f+1 + pack of frame f in one launch)") if fused else "k_scan_cull"), "algorithmic_bytes_per_launch": alg_bytes, "stream_key_bytes": key_bytes,
                         "mean_launch_us": k1_mean, "launches_timed": int(len(k1_us)), "timed_every": TIMING_EVERY},
            "frame_latency_ms_sync": float(np.median(lat) * 1e3),
            "kernel_us_last_frame": tm, "setup_s": t_setup,
        }
        if fused:
            # the fused launch also carries the previous frame's pack: 8 B per instance-list entry read, id + matrix read and written
            out["roofline"]["algorithmic_bytes_per_launch"] = alg_bytes + (8 + 68 + 68) * vis["total"]
            out["roofline"]["achieved"] = out["roofline"]["algorithmic_bytes_per_launch"] / (k1_mean * 1e-6) / 1e9
            out["roofline"]["frac"] = out["roofline"]["achieved"] / HBM_PEAK_GBS
            if scan_only_us is not None and len(scan_only_us):
                so = float(np.mean(scan_only_us))
                out["roofline_scan_only"] = {"bound": "hbm", "kernel": "k_scan_cull (second leg after the timed region: every pack launched on its own)", "mean_launch_us": so,
                                             "algorithmic_bytes_per_launch": alg_bytes, "achieved": alg_bytes / (so * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": alg_bytes / (so * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "launches_timed": int(len(scan_only_us))}
            out["roofline"]["traffic"] = None            # the committed PMC figure `traffic` above is of the scan kernel alone
            if default_workload and os.path.exists(prof):
                try:
                    out["roofline"]["traffic"] = json.load(open(prof))["fused_launch"]["hbm_bytes_per_launch"]
                except Exception:
                    pass
        # with two frame lanes the launches of consecutive frames overlap: next to the per-launch figure, the algorithmic bytes of the timed
        # region over its wall time (what the device as a whole sustained)
        rf = out["roofline"]
        rf["timed_region_achieved"] = rf["algorithmic_bytes_per_launch"] * a.steps / elapsed / 1e9 if rf.get("algorithmic_bytes_per_launch") else None
        rf["timed_region_frac"] = rf["timed_region_achieved"] / HBM_PEAK_GBS if rf["timed_region_achieved"] else None
        if fused and lanes and world == 1 and rf["timed_region_achieved"]:
            # Two launches are in flight at any time (frame lanes): the HIP events of one launch then span a period in which the device moves the
            # bytes of about two.  achieved / frac are therefore the algorithmic bytes of the timed region over its wall time -- the bandwidth
            # the device sustains while this kernel runs; the event figures of the single launch stay next to them.
            rf["per_launch_event"] = {"achieved": rf["achieved"], "frac": rf["frac"], "mean_launch_us": rf["mean_launch_us"],
                                      "note": "one launch timed alone with HIP events while a second one shares the device"}
            rf["achieved"], rf["frac"], rf["concurrent_launches"] = rf["timed_region_achieved"], rf["timed_region_frac"], 2
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"], out["cpu_optimised"] = cpu_baseline(a, atomic, n_total)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(a, atomic, n_total):
    """The CPU oracle (oracle/, a port of the reference algorithm: hash-map spatial index, candidate-box
    enumeration + probe in chunks of 25, per-entity 64-byte append, sequential apply_change), timed on
    the host cores of this box over a bounded sample."""
    import oracle as ro
    from render_engine_amd import synthetic, Camera
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import to_oracle, oracle_camera
    ax = min(a.cpu_sample_axis, a.axis)
    first = (16384 // atomic - ax) // 2
    ents = synthetic.lattice_world(cells_per_axis=ax, first_cell=first, atomic=atomic, spinner_every=a.spinner_every)
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    par = max(1, min(avail, 16))                   # a 1-GPU box gives this job a 16-core share; rayon would size its pool to the cores it may use
    w = ro.World(16384, atomic, threads=1)
    w.register(to_oracle(ents))
    c = (first + ax / 2.0) * atomic
    cam = oracle_camera(Camera((c, c, c), (0.0, 0.0, -1.0), a.far))
    L, h = w.L, w.h
    def frame():
        L.ro_frame_cull(h, C.byref(cam), 0, None)
        L.ro_frame_render(h, C.byref(cam), 0, 0, None, None, 0, None, None)
        L.ro_frame_tick(h, C.byref(cam), np.float32(0.016), 0, None, None)
    best = None
    for threads in sorted({1, par}):               # the reference's par_chunks(25) sites with 1 thread and with the pool; the faster one is reported
        L.ro_set_threads(h, threads)
        for _ in range(3):
            frame()
        n, t0 = 0, time.perf_counter()
        while True:
            frame(); n += 1
            el = time.perf_counter() - t0
            if el > 6.0 or n >= 20000:
                break
        if best is None or el / n < best[0]:
            best = (el / n, threads, n)
    el, threads, n = best[0] * best[2], best[1], best[2]
    per_frame = el / n
    opt = None
    if not a.spinner_every:
        # second, clearly labelled row: what a CPU gets with sorted keys + SoA + OpenMP instead of the reference's hash maps
        # (oracle/re_cpu_soa.c, checked against the port in tests/test_oracle_flows.py); static worlds only, so no tick
        s = ro.SoaWorld(w, to_oracle(ents), threads=1)
        cap = 1 << 16; ids = np.zeros(cap, np.uint32); mats = np.zeros((cap, 16), np.float32); groups = np.zeros(4096, ro.GROUP_DT); ng, nv = C.c_uint32(), C.c_uint32()
        def sframe():
            return L.soa_frame(s.h, C.byref(cam), cap, ids.ctypes.data, mats.ctypes.data, 4096, groups.ctypes.data, C.byref(ng), C.byref(nv), s.mark.ctypes.data)
        sbest = None
        for threads_o in sorted({1, par}):
            s.close(); s = ro.SoaWorld(w, to_oracle(ents), threads=threads_o)
            for _ in range(3):
                sframe()
            k, t0 = 0, time.perf_counter()
            while True:
                sframe(); k += 1
                el2 = time.perf_counter() - t0
                if el2 > 3.0 or k >= 50000:
                    break
            if sbest is None or el2 / k < sbest[0]:
                sbest = (el2 / k, threads_o, k)
        s.close()
        opt = {"value": n_total / sbest[0], "unit": "entities/s", "cores": sbest[1], "kind": "optimised port (sorted keys + SoA + OpenMP, no hash maps)",
               "ms_per_frame": sbest[0] * 1e3, "sample": "%d frames, same sub-lattice and camera as cpu_baseline; cull + render gather incl. the 64-byte appends" % sbest[2]}
    w.close()
    return {"value": n_total / per_frame, "unit": "entities/s", "cores": threads, "kind": "port",
            "ms_per_frame": per_frame * 1e3,
            "sample": ("%d frames over the %d^3-section sub-lattice (%d entities) centred on the camera: it contains every world section the "
                       "reference's candidate-box enumeration touches at far=%g, and the hash-based CPU path does no work for sections outside "
                       "the box, so its frame time equals that of the full %d-entity world; value = %d / that frame time"
                       % (n, ax, len(ents), a.far, n_total, n_total))}, opt


if __name__ == "__main__":
    main()
