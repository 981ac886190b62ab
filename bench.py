#!/usr/bin/env python3
"""bench.py -- entities/sec through cull + transform on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W

A step is ONE FRAME of the hot path over the resident world, as a frame loop that draws every frame sees it:
re_cull_pack (visibility query over the spatial hash + frustum / logic cull + instance pack, with the
InstanceRange table and the counts available to the host when the call returns) followed by re_tick (ECS
kinematic / TRS->mat4 tick) -- Pipeline::execute of the reference (flows/pipeline.rs:212-276).  Both calls are
synchronous.  Inputs are resident in HBM before the timed region starts.

N=1: BASELINE.json configs[1] -- 10,077,696 static entities, one per level-0 world section (uniform spatial-hash
fill), one camera frustum, far = 1000.  `value` = entities / median frame time (median of >= 100 frames);
`ms_per_step` = wall time of exactly K timed frames / K.  The same run also reports, as labelled extras: the
pipelined throughput of asynchronous frames, far = 8192 (wide frustum, ~500 K instances: the large pack path),
configs[2] (100,777 rotating bodies: tick + cull) incl. a tick of every dynamic entity, configs[4] (deferred
lighting), the CPU rows, and a full-size check of the visible set against the CPU oracle.
N>1 (weak scaling, configs[3] at N=8): N x 10,077,696 entities sharded by contiguous section-key range, one process
per GPU, an RCCL all-gather of every GPU's packed visible-instance slab each frame; the frame ends when the
gathered buffer is complete.  Rank 0 prints ONE JSON line.
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

TIMING_EVERY = 4               # HIP events on every 4th launch of the scan kernel inside the timed region (timed dispatches cost queue time)
SLAB_INSTANCES = 4096          # all-gather slab per rank and frame (instances); ~2.8x the visible set of a rank at far=1000
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
VALU_PEAK_TFLOPS = 157.3       # fp32 vector peak (MI355X_MICROARCH.md), the bound of the lighting kernel
PER_GPU_AXIS = 216             # 216^3 = 10,077,696 sections/entities per GPU
MEDIAN_FRAMES = 128            # median of >= 100 frames
K1_MIN_LAUNCHES = 96           # launches of the dominant kernel timed with HIP events, whatever --steps is
CLOCK_WARMUP_FRAMES = 8000     # untimed asynchronous frames (~7 us each) in front of the warm-up steps: the run starts on an idle-clock GPU


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=["default", "visible", "lighting"], default="default",
                    help="default: the visible-set headline + every extra leg; visible: headline only; lighting: configs[4] only")
    ap.add_argument("--far", type=float, default=1000.0, help="camera far draw distance (reference default 1000, main.rs:23)")
    ap.add_argument("--axis", type=int, default=PER_GPU_AXIS, help="sections per axis per GPU (216 -> 10,077,696 entities)")
    ap.add_argument("--spinner-every", type=int, default=0, help="0 = configs[1] (all static); 100 = configs[2] (100k rotating bodies)")
    ap.add_argument("--tick-all", action="store_true", help="tick every dynamic entity (RE_TICK_ALL_DYNAMIC) instead of the visible ones")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the far=8192 / configs[2] / lighting legs")
    ap.add_argument("--force-large-pack", action="store_true", help="always use the multi-kernel pack")
    ap.add_argument("--cpu-sample-axis", type=int, default=100)
    ap.add_argument("--probe", action="store_true", help="opt-in variant: visibility query by hash probes of the candidate cells (RE_CFG_PROBE) instead of the key stream")
    ap.add_argument("--pipelined-only", action="store_true", help="profiling aid: only the asynchronous two-lane loop (round 1's headline)")
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


def kernel_times(p, camc, frames, tick_all=False, R=None):
    """device time of the frame's kernels (HIP events on the pipeline's stream, re_get_timings) over a few synchronous frames: medians, us"""
    p.timings_us()                                     # switches the event recording on
    cull, pack, tick = [], [], []
    for _ in range(frames):
        p.cull_and_pack(camc, copy=False); p.tick(0.016, all_dynamic=tick_all)
        t = p.timings_us(); cull.append(t["cull"]); pack.append(t["pack"]); tick.append(t["tick"])
    p.timings_off()                                    # the event records cost ~12 us per synchronous frame
    return {"cull": float(np.median(cull)), "pack": float(np.median(pack)), "tick": float(np.median(tick))}


def sync_frames(p, camc, n, tick_all=False):
    import render_engine_amd as R
    us, vis, tr = p.run_frames(camc, n, 0.016, 0, R._capi.TICK_ALL_DYNAMIC if tick_all else 0)
    return us, vis, tr


def pipelined_frames(p, camc, n, tick_all=False):
    """asynchronous frames: static worlds defer every pack to the next launch and alternate between two frame lanes"""
    import render_engine_amd as R
    F = R._capi
    p.wait()
    t0 = time.perf_counter()
    p.run_frames(camc, n, 0.016, F.CULL_ASYNC | F.CULL_DEFER_PACK | F.CULL_TWO_LANES, F.TICK_ASYNC | (F.TICK_ALL_DYNAMIC if tick_all else 0))
    p.wait()
    return (time.perf_counter() - t0) / n


def launch_us(p, camc, kernel, n, pipelined, tick_all=False):
    """mean duration of one kernel's own launches (HIP events tied to the dispatch, re_timing_begin) over n synchronous or asynchronous frames"""
    p.wait()
    p.timing_begin(n, 1, kernel=kernel)
    if pipelined:
        pipelined_frames(p, camc, n, tick_all)
    else:
        sync_frames(p, camc, n, tick_all)
    us = p.timing_collect()
    return float(np.mean(us)) if len(us) else None


def pin_near_gpu(index):
    """Runs this process on the CPUs of the GPU's NUMA node (sysfs local_cpulist), before anything is allocated: the result block the kernels publish into is
    pinned host memory of the calling thread's node and the thread polls it, so a thread (or a block) on the other socket adds a socket hop to every frame
    (tools/affinity_probe.py: 24.7 us per synchronous frame on a local CPU, 25.2 us on a remote one; runs whose block landed on the remote node were up to 3 us slower).
    Returns a description for the bench line; does nothing when sysfs does not say or the CPU set is restricted away from that node."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        cpus = set()
        for part in open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read().strip().split(","):
            if part:
                lo, _, hi = part.partition("-"); cpus.update(range(int(lo), int(hi or lo) + 1))
        local = os.sched_getaffinity(0) & cpus
        if not local:
            return "unchanged (no allowed CPU on the GPU's NUMA node)"
        os.sched_setaffinity(0, local)
        return "%d CPUs of the GPU's NUMA node (%s)" % (len(local), bdf)
    except Exception as e:      # noqa: BLE001
        return "unchanged (%s)" % str(e)[:80]


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
    affinity = pin_near_gpu(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if a.config == "lighting":
        if rank == 0:
            out = lighting_leg(headline=True, steps=a.steps, warmup=a.warmup)
            print(json.dumps(out))
        if dist is not None:
            dist.destroy_process_group()
        return

    import render_engine_amd as R
    from render_engine_amd import parallel
    F = R._capi

    atomic = 64 if world == 1 else 32            # 432 sections per axis need atomic 32
    extras = world == 1 and a.config == "default" and not a.no_extras and a.far == 1000.0 and not a.spinner_every and a.axis == PER_GPU_AXIS and not a.probe
    t0 = time.time()
    ents, dims, first = make_shard(rank, world, a.axis, atomic, a.spinner_every)
    n_local = len(ents)
    cap = max(1 << 16, n_local // 2) if (a.far > 2000 or extras) else 1 << 16      # the far=8192 leg packs ~500 K instances
    p = R.Pipeline(16384, atomic, device=local, max_instances=cap, flags=F.CFG_PROBE if a.probe else 0)
    p.register_model_instances(ents)
    stats = p.stats()
    if not extras:
        del ents
    t_setup = time.time() - t0
    centre = [(first + d / 2.0) * atomic for d in (dims[0], dims[2], dims[1])]      # x, y, z
    cam = R.Camera(centre, (0.0, 0.0, -1.0), a.far)
    camc = cam.to_c()
    tick_flags = F.TICK_ALL_DYNAMIC if a.tick_all else 0
    cull_flags = F.CULL_FORCE_LARGE_PACK if a.force_large_pack else 0
    # N > 1: the frame's one exchange step runs behind the C ABI (re_comm_init + re_allgather_visible: RCCL called by the library on the pipeline's
    # stream).  The unique id travels over torch.distributed, which this script also uses for the barrier and the max over ranks.  Where the
    # library cannot set the communicator up (e.g. the rehearsal with several ranks on one GPU) the torch.distributed harness takes over.
    gather, exchange = None, "none"
    if world > 1:
        try:
            box = [R.Pipeline.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            p.comm_init(box[0], rank, world, SLAB_INSTANCES)
            exchange = "re_allgather_visible (RCCL behind the C ABI)"
        except Exception as e:          # noqa: BLE001
            ok = torch.tensor([0], dtype=torch.int32, device="cuda")
            exchange = "torch.distributed all_gather_into_tensor (harness; re_comm_init failed: %s)" % str(e)[:120]
        else:
            ok = torch.tensor([1], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:         # every rank takes the same path
            try:
                p._L.re_comm_destroy(p._h)
            except Exception:           # noqa: BLE001
                pass
            gather = parallel.SlabAllGather(p, SLAB_INSTANCES, dist)
            if not exchange.startswith("torch"):
                exchange = "torch.distributed all_gather_into_tensor (harness; another rank could not set up the communicator)"

    def frames(n, record=None):
        """n synchronous frames; N > 1: the frame ends when the all-gathered visible-instance buffer is complete on this rank"""
        if gather is None:
            us, vis, tr = p.run_frames(camc, n, 0.016, cull_flags, tick_flags)        # the two C ABI calls in a native loop
            if record is not None:
                record.extend(us.tolist())
            return vis
        cv, ct = F.Visible(), F.TickResult()            # the C ABI straight through ctypes: no per-frame conversion of the results
        for _ in range(n):
            t1 = time.perf_counter()
            gather.begin_frame()
            p._check(p._L.re_cull_pack(p._h, C.byref(camc), cull_flags, C.byref(cv)), "re_cull_pack")
            gather.exchange()                           # enqueued on the pipeline's stream: ordered behind the pack
            p._check(p._L.re_tick(p._h, np.float32(0.016), tick_flags, C.byref(ct)), "re_tick")
            gather.finish()                             # the gathered buffer is complete on this rank
            if record is not None:
                record.append((time.perf_counter() - t1) * 1e6)
        return p._visible_to_py(cv, copy=False) if n else None

    def fence():
        p.wait()
        if gather is not None:
            gather.finish()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    if a.pipelined_only:
        pipelined_frames(p, camc, a.warmup, a.tick_all); per = pipelined_frames(p, camc, a.steps, a.tick_all)
        if rank == 0:
            print(json.dumps({"pipelined_ms_per_frame": per * 1e3, "entities_per_s": n_local / per}))
        return

    # Clock warm-up, untimed and in front of the W warm-up steps: the run starts behind seconds of host-side setup with the GPU in its idle power
    # state, and W = 5 synchronous frames (~0.1 ms) do not bring the clocks up -- round 2's driver run timed its five launches at 17.4 us where a
    # warm GPU needs 11.  ~50 ms of back-to-back frames, the same treatment the far = 8192 and lighting legs already had.
    if world == 1:
        pipelined_frames(p, camc, CLOCK_WARMUP_FRAMES, a.tick_all)
    else:
        frames(CLOCK_WARMUP_FRAMES // 8)
    frames(a.warmup)
    fence()
    p.timing_begin(a.steps, every=TIMING_EVERY)
    wall = []
    t_start = time.perf_counter()
    vis = frames(a.steps, wall)
    fence()
    elapsed = time.perf_counter() - t_start
    k1_region = p.timing_collect(a.steps)
    n_cand = p.last_candidates()
    med_frames = list(wall)
    if len(med_frames) < MEDIAN_FRAMES:                # median of >= 100 frames
        frames(MEDIAN_FRAMES - len(med_frames), med_frames)
        fence()
    # the dominant kernel's own launches: at least K1_MIN_LAUNCHES of them whatever --steps was (the timed region alone holds steps / TIMING_EVERY),
    # every launch of further synchronous frames of the same loop, directly behind the timed region
    n_more = max(K1_MIN_LAUNCHES - len(k1_region), 0)
    k1_more = np.zeros(0, np.float32)
    if n_more:
        p.timing_begin(n_more, every=1)
        frames(n_more)
        fence()
        k1_more = p.timing_collect(n_more)
    k1_us = np.concatenate([np.asarray(k1_region, np.float32), np.asarray(k1_more, np.float32)])
    if dist is not None:
        t = torch.tensor([elapsed, float(np.median(med_frames))], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, med_us = float(t[0].item()), float(t[1].item())
        tot = torch.tensor([n_local], dtype=torch.int64, device="cuda"); dist.all_reduce(tot); n_total = int(tot.item())
    else:
        med_us = float(np.median(med_frames)); n_total = n_local

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        slots = stats["n_section_slots"]
        n_entries, V = vis["n_visible_sections"], vis["total"]
        key_bytes = 4 if (16384 + atomic - 1) // atomic <= 512 and not os.environ.get("RE_EXP_KEY64") else 8
        # Two byte bases for the dominant kernel (the section-key scan + cull, k_scan_cull), see DESIGN.md "Roofline accounting":
        #  survey_8d   SURVEY 8(d): 40 B per section (key, tight AABB, begin/count) + 8 B per entity (id, model word) + 132 B per visible instance.
        #              It assumes every section's AABB / ranges and every entity's id word are streamed; this layout never touches them for
        #              sections outside the frustum, so bytes_survey_8d / time exceeds the HBM peak: the kernel does not move those bytes.
        #  compulsory  what this layout must read: the stream key of every section slot + one level word per 512 slots + 41 B per visible
        #              section (flags, tight AABB, counts, begin; stamp written) + 16 B per visible instance (row index, group class; list entry).
        bytes_8d = 40 * stats["n_sections"] + 8 * stats["n_entities"] + 132 * V
        bytes_comp = key_bytes * slots + 4 * ((slots + 511) // 512) + 41 * n_entries + 16 * V
        probed = a.probe and p.stats()["n_probe_frames"] > 0
        if probed:
            bytes_comp = 16 * int(n_cand) + 41 * n_entries + 16 * V
        k1_mean = float(np.mean(k1_us)) if len(k1_us) else float("nan")
        gbs = lambda b, us: b / (us * 1e-6) / 1e9 if us and us > 0 else None
        traffic, tr_src = committed_traffic("scan") if (world == 1 and a.axis == 216 and a.far == 1000.0 and not a.spinner_every and not probed) else (None, None)
        wl = ("configs[1]: %d static entities, one per level-0 world section (uniform spatial-hash fill), 1 camera frustum far=%g" % (n_total, a.far)) if not a.spinner_every else \
             ("configs[2]: %d entities incl. every %dth rotating (ECS tick + cull), far=%g" % (n_total, a.spinner_every, a.far))
        out = {
            "metric": "entities/sec through cull+transform; visible-set ms/frame @ 10M entities",
            "value": n_total / (med_us * 1e-6), "unit": "entities/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "value_basis": "entities / median wall time of one synchronous frame (re_cull_pack + re_tick, results available to the host%s) over %d frames; ms_per_step = wall of the %d timed frames / %d"
                           % ("; N>1: incl. the all-gather of the packed visible-instance slabs" if world > 1 else "", len(med_frames), a.steps, a.steps),
            "frame_ms_median": med_us * 1e-3,
            "config": {"workload": wl, "entities": n_total, "sections": stats["n_sections"] * world, "dynamic_entities": stats["n_dynamic"] * world,
                       "visible_sections": n_entries, "visible_instances": V, "far": a.far, "frame": "synchronous re_cull_pack + re_tick%s" % (" (RE_TICK_ALL_DYNAMIC)" if a.tick_all else ""),
                       "sharding": "none" if world == 1 else "contiguous section-key ranges; per frame one all-gather of fixed %d-instance slabs (header written by the pack kernel), stream-ordered behind the pack" % SLAB_INSTANCES,
                       "exchange": exchange},
            "roofline": {"bound": "hbm", "kernel": "k_probe_cull" if probed else "k_scan_cull", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "mean_launch_us": k1_mean, "median_launch_us": float(np.median(k1_us)) if len(k1_us) else None, "min_launch_us": float(np.min(k1_us)) if len(k1_us) else None,
                         "launches_timed": int(len(k1_us)), "launches_in_timed_region": int(len(k1_region)), "timed_every_in_region": TIMING_EVERY,
                         "mean_launch_us_timed_region": float(np.mean(k1_region)) if len(k1_region) else None,
                         "bytes_compulsory": bytes_comp, "bytes_survey_8d": bytes_8d, "stream_key_bytes": key_bytes,
                         "achieved": gbs(bytes_comp, k1_mean), "frac": (gbs(bytes_comp, k1_mean) or 0) / HBM_PEAK_GBS if k1_mean == k1_mean else None,
                         "achieved_survey_8d": gbs(bytes_8d, k1_mean), "frac_survey_8d": (gbs(bytes_8d, k1_mean) or 0) / HBM_PEAK_GBS if k1_mean == k1_mean else None,
                         "basis": ("achieved / frac use bytes_compulsory (what this layout must move) over the mean of launches_timed launches (HIP events on the dispatches: those of the timed region "
                                   "+ further synchronous frames of the same loop); frac_survey_8d > 1 means the kernel does not move SURVEY 8d's bytes: sections outside the frustum cost only their "
                                   "%d-byte stream key.  The %d MB key array is re-read every frame and stays resident in the 256 MiB Infinity Cache between frames, so this stream is served by the "
                                   "Infinity Cache, not by HBM: the fraction of the 8 TB/s HBM peak is a label for the byte rate, not a bound the kernel runs against (gfx950's FETCH_SIZE counts "
                                   "Infinity-Cache hits too)" % (key_bytes, key_bytes * slots // 1000000)),
                         "clock_warmup": "%d untimed asynchronous frames in front of the %d warm-up steps" % (CLOCK_WARMUP_FRAMES, a.warmup),
                         "traffic": traffic, "traffic_source": tr_src},
            "setup_s": t_setup, "cpu_affinity": affinity,
        }
        if world == 1:
            kt = kernel_times(p, camc, 16, a.tick_all)
            out["kernel_us"] = kt
            per = pipelined_frames(p, camc, 64, a.tick_all); per = pipelined_frames(p, camc, max(a.steps, 200), a.tick_all)
            out["pipelined"] = {"ms_per_frame": per * 1e3, "entities_per_s": n_total / per,
                                "note": "asynchronous frames, no result read by the host per frame; static worlds defer each pack to the next launch and alternate two frame lanes (round 1's headline figure)"}
            # opt-in variant (RE_CULL_ONE_LAUNCH): the scan's last workgroup publishes the frame itself -- one launch between the call and its answer.  Reported beside the default;
            # its scan launch contains the publication chain, so its duration is not comparable with roofline.achieved above
            F = R._capi
            p.run_frames(camc, 64, 0.016, F.CULL_ONE_LAUNCH, F.TICK_ALL_DYNAMIC if a.tick_all else 0)
            us1, _, _ = p.run_frames(camc, max(a.steps, 200), 0.016, F.CULL_ONE_LAUNCH, F.TICK_ALL_DYNAMIC if a.tick_all else 0)
            p.wait(); p.timing_begin(64, 1, kernel="scan"); p.run_frames(camc, 64, 0.016, F.CULL_ONE_LAUNCH, F.TICK_ALL_DYNAMIC if a.tick_all else 0); k1 = p.timing_collect()
            out["one_launch_sync"] = {"frame_ms_median": float(np.median(us1)) * 1e-3, "entities_per_s": n_total / (float(np.median(us1)) * 1e-6),
                                      "scan_launch_us_incl_publication": float(np.mean(k1)) if len(k1) else None,
                                      "note": "RE_CULL_ONE_LAUNCH: k_scan_cull_sync + k_pack_small(move only); opt-in, not the default (DESIGN.md section 4)"}
        if not a.no_cpu_baseline and world == 1:
            gpu_frame = p.cull_and_pack(camc, copy=True) if not a.spinner_every else None
            out["cpu_baseline"], out["cpu_optimised"], out["full_size_check"] = cpu_baseline(a, atomic, n_total, dims, first, gpu_frame)
        p.close()
        if extras:
            # a fresh upload of the same entities: the reference's static render cache freezes with the camera of the first frame after a
            # registration (render_flow.rs:549-594), so a world first seen at far=1000 would draw only those sections at far=8192
            out["far_8192"] = far_leg(R, ents, atomic, centre, n_total, key_bytes)
            del ents
            out["configs_2"] = spinner_leg(R, a, atomic)
            out["lighting"] = lighting_leg(headline=False, steps=50, warmup=150)      # (warm-up ~40 ms: the leg starts on an idle GPU)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def committed_traffic(which):
    """HBM traffic per launch from the committed PMC profile (separate --pmc passes, gfx950 FETCH_SIZE correction applied); not measured in this run"""
    for name in ("r03_pmc.json", "r02_pmc.json", "r01_pmc_k_scan_cull.json"):
        f = os.path.join(ROOT, "profiles", name)
        if os.path.exists(f):
            try:
                d = json.load(open(f))
                v = d.get("hbm_bytes_per_launch") if which == "scan" else d.get(which, {}).get("hbm_bytes_per_launch")
                if v:
                    return v, "profiles/%s (committed PMC profile of this workload, not measured in this run)" % name
            except Exception:
                pass
    return None, None


def far_leg(R, ents, atomic, centre, n_total, key_bytes):
    """far = 8192 on the configs[1] world: ~500 K visible instances, the large pack path (SURVEY 8d asks for both far values)"""
    p = R.Pipeline(16384, atomic, max_instances=max(1 << 16, len(ents) // 2))
    p.register_model_instances(ents)
    stats = p.stats()
    cam = R.Camera(centre, (0.0, 0.0, -1.0), 8192.0).to_c()
    sync_frames(p, cam, 8)
    pipelined_frames(p, cam, 1500)               # untimed warm-up (~75 ms of back-to-back launches: this leg starts after the CPU baseline has left the GPU idle for tens of seconds)
    us, vis, _ = sync_frames(p, cam, 48)
    kt = kernel_times(p, cam, 12)
    per = pipelined_frames(p, cam, 16); per = pipelined_frames(p, cam, 64)
    own = {"k_scan_cull": {"sync_frames": launch_us(p, cam, "scan", 48, False), "async_frames": launch_us(p, cam, "scan", 96, True)},
           "k_pack_large": {"sync_frames": launch_us(p, cam, "pack_large", 48, False), "async_frames": launch_us(p, cam, "pack_large", 96, True)}}
    V, S, slots = vis["total"], vis["n_visible_sections"], stats["n_section_slots"]
    b_scan = key_bytes * slots + 4 * ((slots + 511) // 512) + 41 * S + 16 * V
    b_pack = (8 + 4 + 64 + 68) * V               # list entry + id + matrix read, id + matrix written
    dev = kt["cull"] + kt["pack"]
    p.close()
    return {"workload": "configs[1] world (fresh upload), far=8192", "visible_sections": S, "visible_instances": V,
            "frame_ms_median_sync": float(np.median(us)) * 1e-3, "entities_per_s_sync": n_total / (float(np.median(us)) * 1e-6),
            "pipelined_ms_per_frame": per * 1e3, "entities_per_s_pipelined": n_total / per,
            "kernel_us": kt,
            "launch_us": dict(own, note="each kernel's own launches (dispatch-bound HIP events); kernel_us = event pairs around the calls of synchronous frames"),
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                         "scan": {"kernel": "k_scan_cull_wide (the scan of frames with a large visible set: k_scan_cull with a wave's key loads requested together)", "us": kt["cull"], "bytes_compulsory": b_scan, "frac": b_scan / (kt["cull"] * 1e-6) / 1e9 / HBM_PEAK_GBS if kt["cull"] > 0 else None},
                         "pack": {"kernel": "instance pack (large path)", "us": kt["pack"], "bytes_compulsory": b_pack, "frac": b_pack / (kt["pack"] * 1e-6) / 1e9 / HBM_PEAK_GBS if kt["pack"] > 0 else None},
                         "frame": {"us": dev, "bytes_compulsory": b_scan + b_pack, "frac": (b_scan + b_pack) / (dev * 1e-6) / 1e9 / HBM_PEAK_GBS if dev > 0 else None,
                                   "bytes_survey_8d": 40 * stats["n_sections"] + 8 * stats["n_entities"] + 132 * V,
                                   "frac_survey_8d": (40 * stats["n_sections"] + 8 * stats["n_entities"] + 132 * V) / (dev * 1e-6) / 1e9 / HBM_PEAK_GBS if dev > 0 else None}}}


def spinner_leg(R, a, atomic):
    """configs[2]: the 10,077,696-entity world with every 100th entity a rotating body (100,777 dynamic entities)"""
    from render_engine_amd import synthetic
    first = (16384 // atomic - a.axis) // 2
    ents = synthetic.lattice_world(cells_per_axis=a.axis, first_cell=first, atomic=atomic, spinner_every=100)
    n = len(ents)
    p = R.Pipeline(16384, atomic, max_instances=max(1 << 16, n // 2))
    p.register_model_instances(ents)
    st = p.stats()
    del ents
    c = (first + a.axis / 2.0) * atomic
    out = {"workload": "configs[2]: %d entities incl. %d rotating bodies (space_logic asteroid spin), ECS tick + cull" % (n, st["n_dynamic"])}
    B_TICK = 184                                     # 80 B read + 104 B written per ticking entity
    # (the static render cache of this upload freezes with the first camera, far=1000: the wide camera afterwards draws the rotating bodies it
    # sees plus the static entities cached then -- it is here for the tick: >= 10 K visible rotating bodies)
    legs = [("far_1000", R.Camera((c, c, c), (0, 0, -1), 1000.0), False),
            ("far_1000_tick_all", R.Camera((c, c, c), (0, 0, -1), 1000.0), True),
            ("wide_camera", R.Camera((c, c, (first + a.axis) * atomic - 100.0), (0, 0, -1), 16384.0), False)]
    for name, cam, tick_all in legs:
        camc = cam.to_c()
        sync_frames(p, camc, 6, tick_all)
        us, vis, tr = sync_frames(p, camc, 40, tick_all)
        kt = kernel_times(p, camc, 10, tick_all)
        per = pipelined_frames(p, camc, 8, tick_all); per = pipelined_frames(p, camc, 200, tick_all)
        t_sync, t_async = launch_us(p, camc, "tick", 40, False, tick_all), launch_us(p, camc, "tick", 200, True, tick_all)
        med = float(np.median(us))
        out[name] = {"camera": "far=%g%s" % (cam.far_draw_distance, ", every dynamic entity ticks (RE_TICK_ALL_DYNAMIC)" if tick_all else ", entities of visible active sections tick (reference semantics)"),
                     "visible_instances": vis["total"], "entities_ticked": tr["n_changed"],
                     "frame_ms_median_sync": med * 1e-3, "entities_per_s_sync": n / (med * 1e-6),
                     "pipelined_ms_per_frame": per * 1e3, "entities_per_s_pipelined": n / per, "kernel_us": kt,
                     "tick_roofline": {"bound": "hbm", "kernel": "k_tick", "bytes_per_entity_survey_8d": B_TICK, "bytes": B_TICK * tr["n_changed"],
                                       "us_sync_frames": t_sync, "us_async_frames": t_async,
                                       "frac_sync_frames": B_TICK * tr["n_changed"] / (t_sync * 1e-6) / 1e9 / HBM_PEAK_GBS if t_sync else None,
                                       "frac_async_frames": B_TICK * tr["n_changed"] / (t_async * 1e-6) / 1e9 / HBM_PEAK_GBS if t_async else None,
                                       "note": "k_tick's own launches (dispatch-bound HIP events) in synchronous and in asynchronous frame loops; at ~100 K ticking entities the launch holds "
                                               "1.5 waves per SIMD and is bound by the length of one wave's instruction stream, not by HBM (dense_tick_all fills the machine)"}}
    # user change requests (apply_change, helper_things/entity_change_helpers.rs:32-189): the reference's frame loop re-inserts the user entity at the camera position every frame
    # (logic_flow.rs:246-251) -- one Position change per frame; wall time of the call from Python (the ctypes call itself is ~8 us of it)
    try:
        cam0 = R.Camera((c, c, c), (0, 0, -1), 1000.0).to_c()
        ids_dyn = p.get_indexes_for_components([R._capi.C_ROTATION_VEL])
        eid = int(ids_dyn[len(ids_dyn) // 2])
        base = np.array(p.read_component(eid, R._capi.C_POSITION)[:3], np.float32)
        chg = {}
        for label, step in (("inside_its_section", 0.01), ("across_a_section_border", float(atomic))):
            ts = []
            for f in range(40):
                p.cull_and_pack(cam0, copy=False)
                pos = base + np.float32(step * (f % 2 if step > 1 else f))
                ch = np.zeros(1, R.CHANGE_DT); ch[0] = (R._capi.CHANGE_MODIFY, eid, R._capi.C_POSITION, 0, (pos[0], pos[1], pos[2], 0))
                t0 = time.perf_counter(); p.apply_changes(ch); ts.append(time.perf_counter() - t0)
                p.tick(1e-4)
            chg[label + "_us"] = float(np.median(ts[8:])) * 1e6
        chg["note"] = "one Position change of a dynamic entity per frame (re_apply_changes), median wall time of the call; k_apply_small + the device-side re-bucket when the section changes"
        out["change_request"] = chg
    except Exception as e:      # noqa: BLE001
        out["change_request"] = {"error": str(e)[:200]}
    p.close()
    # the same kernel with the machine full: every 10th entity a rotating body (1,007,770 ticking entities, ~15 waves per SIMD instead of 1.5)
    ents = synthetic.lattice_world(cells_per_axis=a.axis, first_cell=first, atomic=atomic, spinner_every=10)
    p = R.Pipeline(16384, atomic, max_instances=1 << 16)
    p.register_model_instances(ents)
    nd = p.stats()["n_dynamic"]
    del ents
    camc = R.Camera((c, c, c), (0, 0, -1), 1000.0).to_c()
    sync_frames(p, camc, 6, True)
    us, vis, tr = sync_frames(p, camc, 40, True)
    per = pipelined_frames(p, camc, 8, True); per = pipelined_frames(p, camc, 100, True)
    t_sync, t_async = launch_us(p, camc, "tick", 40, False, True), launch_us(p, camc, "tick", 100, True, True)
    out["dense_tick_all"] = {"workload": "every 10th entity a rotating body: %d dynamic entities, all ticking (RE_TICK_ALL_DYNAMIC), far=1000" % nd, "entities_ticked": tr["n_changed"],
                             "frame_ms_median_sync": float(np.median(us)) * 1e-3, "pipelined_ms_per_frame": per * 1e3,
                             "tick_roofline": {"bound": "hbm", "kernel": "k_tick", "bytes_per_entity_survey_8d": B_TICK, "bytes": B_TICK * tr["n_changed"],
                                               "us_sync_frames": t_sync, "us_async_frames": t_async,
                                               "frac_sync_frames": B_TICK * tr["n_changed"] / (t_sync * 1e-6) / 1e9 / HBM_PEAK_GBS if t_sync else None,
                                               "frac_async_frames": B_TICK * tr["n_changed"] / (t_async * 1e-6) / 1e9 / HBM_PEAK_GBS if t_async else None}}
    p.close()
    return out


def lighting_leg(headline, steps=20, warmup=3):
    """configs[4]: deferred lighting of a 4096 x 4096 synthetic G-buffer with 4096 radius ('spot') lights as a HIP compute kernel"""
    import oracle as ro
    from render_engine_amd import lighting
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    W = H = 4096; NL = 4096
    t0 = time.time()
    pos, nrm, alb = lighting.synthetic_gbuffer(W, H)
    L = lighting.synthetic_lights(n_spot=NL, n_point=0)
    dl = lighting.DeferredLighting(W, H, max_spot_lights=NL, max_point_lights=64)
    dl.upload_gbuffer(pos, nrm, alb); dl.set_lights(L)
    t_setup = time.time() - t0
    for _ in range(warmup):
        dl.run()
    us = [dl.run() for _ in range(steps)]
    t = float(np.median(us))
    keep = []
    S = lighting.fill_lights_struct(ro.LightsC(), L, keep)
    pairs = ro.lighting_spot_pairs(pos, S)             # exact count of (pixel, light within its radius) pairs: the oracle is the counter, not the thing measured
    # sampled check of the full-size image against the CPU evaluation of the GLSL (tolerance 1e-4, BASELINE.json)
    idx = (np.arange(2048, dtype=np.uint64) * np.uint64(8191 * 4099 + 7) % np.uint64(W * H)).astype(np.uint32)
    err = float(np.abs(dl.read_pixels(idx) - ro.deferred_lighting(pos, nrm, alb, S, idx=idx)).max())
    if err > 1e-4:
        raise SystemExit("lighting: GPU image differs from the CPU evaluation of the GLSL by %g" % err)
    dl.close()
    npix = W * H
    flops = pairs * 150 + 20 * npix                  # ~75 flop per evaluated (pixel, light) term, the spot term is evaluated twice, + 20 per pixel epilogue
    bytes_ = npix * (16 + 16 + 4 + 16)               # gPosition + gNormal RGBA32F, gAlbedoSpec RGBA8 read; FragColor RGBA32F written (gLightPosition is not needed)
    tf, gb = flops / (t * 1e-6) / 1e12, bytes_ / (t * 1e-6) / 1e9
    # What binds the kernel is decided by the SQ counters, not by which of the two model ratios below is larger: SQ_INSTS_VALU = 92.8 M wave-instructions per
    # launch fill 75-85 % of the VALU issue slots of the launch (tools/pmc_sq.sh); valu_frac prices only the 150-flop model of the in-radius
    # (pixel, light) pairs, about a quarter of what a wave executes (it evaluates a listed light for all 64 lanes when any lane is inside the radius).
    binds = "fp32 VALU issue (SQ_INSTS_VALU: 75-85 % of the issue slots; hbm_frac and valu_frac are model ratios, not utilisations)"
    res = {"workload": "configs[4]: deferred lighting, %dx%d synthetic G-buffer, %d radius-40 'spot' lights (second_pass_frag.glsl:20-139)" % (W, H, NL),
           "kernel": "k_deferred_lighting", "kernel_us": t, "pixels_per_s": npix / (t * 1e-6),
           "pixel_light_pairs_in_radius": pairs, "pairs_per_pixel": pairs / npix, "flop_model": "150 flop per pair (75 per evaluated term, spot term twice) + 20 per pixel",
           "gflops": flops / (t * 1e-6) / 1e9, "valu_frac": tf / VALU_PEAK_TFLOPS, "valu_peak_tflops": VALU_PEAK_TFLOPS,
           "gbs": gb, "hbm_frac": gb / HBM_PEAK_GBS, "hbm_bytes": bytes_, "binding_bound": binds,
           "max_abs_err_vs_cpu_glsl_2048_pixels": err, "tolerance": 1e-4, "setup_s": t_setup}
    if not headline:
        return res
    return {"metric": "deferred-lighting pixels/sec (configs[4])", "value": res["pixels_per_s"], "unit": "pixels/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
            "ms_per_step": t * 1e-3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": res["workload"]},
            "roofline": {"bound": "hbm", "note": "the kernel is fp32-VALU-issue bound (no MFMA: per-light branchy shading), so this HBM fraction is low by construction; the VALU figures are in `lighting`",
                         "achieved": gb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / HBM_PEAK_GBS,
                         "traffic": committed_traffic("lighting")[0], "traffic_source": committed_traffic("lighting")[1]},
            "lighting": res}


def cpu_baseline(a, atomic, n_total, dims, first, gpu_frame):
    """The CPU oracle (oracle/, a port of the reference algorithm: hash-map spatial index, candidate-box
    enumeration + probe in chunks of 25, per-entity 64-byte append, sequential apply_change), timed on
    the host cores of this box over a bounded sample -- and used as the checker of the full-size GPU frame."""
    import oracle as ro
    from render_engine_amd import synthetic, Camera
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import to_oracle, oracle_camera
    ax = min(a.cpu_sample_axis, a.axis)
    off = (a.axis - ax) // 2
    # the sub-lattice around the camera, with the SAME entities as the full world (ids, sizes, positions): section (i, j, k) of the sample is
    # section (off + i, off + j, off + k) of the world
    full_ids = synthetic.sub_box_indices(dims, (off, off, off), (ax, ax, ax))
    ents = synthetic.box_world(dims, first_cell=first, atomic=atomic, spinner_every=a.spinner_every, indices=full_ids)
    ents["id"] = np.arange(len(ents), dtype=np.uint32)            # dense ids for the oracle's tables; full_ids maps back
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    par = max(1, min(avail, 16))                   # a 1-GPU box gives this job a 16-core share; rayon would size its pool to the cores it may use
    w = ro.World(16384, atomic, threads=1)
    w.register(to_oracle(ents))
    c = (first + a.axis / 2.0) * atomic
    pcam = Camera((c, c, c), (0.0, 0.0, -1.0), a.far)
    cam = oracle_camera(pcam)
    L, h = w.L, w.h
    # ---- full-size self-check: the 10,077,696-entity GPU frame against the oracle on the sub-lattice that contains every candidate section
    check = None
    if gpu_frame is not None:
        vis_o = w.cull(cam); o = w.render(cam)
        ids_o = np.sort(full_ids[o["ids"].astype(np.int64)].astype(np.uint32))
        ids_g = np.sort(gpu_frame["ids"])
        ok = (gpu_frame["total"] == o["total"] and gpu_frame["n_visible_vec"] == len(vis_o) and gpu_frame["n_visible_sections"] == len(np.unique(vis_o))
              and len(ids_g) == len(ids_o) and bool(np.all(ids_g == ids_o)))
        if ok:                                       # matrices, bit for bit, per entity
            og, oo = np.argsort(gpu_frame["ids"], kind="stable"), np.argsort(full_ids[o["ids"].astype(np.int64)], kind="stable")
            ok = bool(np.all(gpu_frame["mats"][og].view(np.uint32) == o["mats"][oo].view(np.uint32)))
        check = {"ok": ok, "visible_sections": int(gpu_frame["n_visible_sections"]), "visible_instances": int(gpu_frame["total"]),
                 "oracle_visible_sections": int(len(np.unique(vis_o))), "oracle_visible_instances": int(o["total"]),
                 "what": "visible_sections, visible_sections_vec, visible_instances, the sorted entity-id set and every 4x4 matrix (bit-exact) of the %d-entity GPU frame against the CPU oracle on the %d^3 sub-lattice around the camera (same entities, ids mapped back)" % (n_total, ax)}
        if not ok:
            raise SystemExit("bench.py: full-size check FAILED: " + json.dumps(check))
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
                       % (n, ax, len(ents), a.far, n_total, n_total))}, opt, check


if __name__ == "__main__":
    main()
