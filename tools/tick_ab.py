#!/usr/bin/env python3
"""k_tick at configs[2] (100,777 rotating bodies) and with 1,007,770: the kernel's own launches and the synchronous frame, every dynamic entity ticking
and visibility-gated.  tools/tick_ab.py [label]"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import render_engine_amd as R
from render_engine_amd import synthetic
F = R._capi
axis, atomic = 216, 64
first = (16384 // atomic - axis) // 2
c = (first + axis / 2.0) * atomic
out = {"label": sys.argv[1] if len(sys.argv) > 1 else ""}
for every in (100, 10):
    ents = synthetic.lattice_world(cells_per_axis=axis, first_cell=first, atomic=atomic, spinner_every=every)
    p = R.Pipeline(16384, atomic, max_instances=1 << 16)
    p.register_model_instances(ents); nd = p.stats()["n_dynamic"]; del ents
    cam = R.Camera((c, c, c), (0, 0, -1), 1000.0).to_c()
    res = {"n_dynamic": nd}
    for tick_all in (True, False):
        tf = F.TICK_ALL_DYNAMIC if tick_all else 0
        p.run_frames(cam, 8, 0.016, 0, tf)
        p.run_frames(cam, 1500, 0.016, F.CULL_ASYNC, F.TICK_ASYNC | tf); p.wait()
        us, vis, tr = p.run_frames(cam, 64, 0.016, 0, tf)
        p.timing_begin(64, 1, kernel="tick"); p.run_frames(cam, 64, 0.016, 0, tf); t = p.timing_collect()
        p.timing_begin(200, 1, kernel="tick"); p.run_frames(cam, 200, 0.016, F.CULL_ASYNC, F.TICK_ASYNC | tf); p.wait(); ta = p.timing_collect()
        res["tick_all" if tick_all else "gated"] = {"ticked": tr["n_changed"], "frame_median_us": float(np.median(us)), "tick_us_sync": float(np.mean(t)), "tick_us_async": float(np.mean(ta)),
                                                     "frac_async": 184.0 * tr["n_changed"] / (float(np.mean(ta)) * 1e-6) / 8e12}
    out["every_%d" % every] = res
    p.close()
print(json.dumps(out))
