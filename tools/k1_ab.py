#!/usr/bin/env python3
"""K1 at configs[1] (far = 1000) and far = 8192: mean / median / min of the kernel's own launches in synchronous frames.  One process per library variant
(environment switches are read when the library loads): tools/k1_ab.py [label]"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import render_engine_amd as R
from render_engine_amd import synthetic
F = R._capi
axis, atomic = 216, 64
first = (16384 // atomic - axis) // 2
ents = synthetic.lattice_world(cells_per_axis=axis, first_cell=first, atomic=atomic)
c = (first + axis / 2.0) * atomic
out = {"label": sys.argv[1] if len(sys.argv) > 1 else "", "env": {k: v for k, v in os.environ.items() if k.startswith("RE_EXP")}}
for far in (1000.0, 8192.0):
    p = R.Pipeline(16384, atomic, max_instances=max(1 << 16, len(ents) // 2))
    p.register_model_instances(ents)
    cam = R.Camera((c, c, c), (0, 0, -1), far).to_c()
    p.run_frames(cam, 8, 0.016, 0, 0)
    p.run_frames(cam, 3000 if far < 2000 else 800, 0.016, F.CULL_ASYNC, F.TICK_ASYNC); p.wait()
    us, vis, _ = p.run_frames(cam, 64, 0.016, 0, 0)
    res = {"frame_median_us": float(np.median(us)), "visible": vis["total"], "sections": vis["n_visible_sections"]}
    for kern in (("scan",) if far < 2000 else ("scan", "pack_large")):
        p.timing_begin(64, 1, kernel=kern); p.run_frames(cam, 64, 0.016, 0, 0); t = p.timing_collect()
        res[kern] = {"mean": float(np.mean(t)), "median": float(np.median(t)), "min": float(np.min(t))}
    out["far_%d" % far] = res
    p.close()
print(json.dumps(out))
