#!/usr/bin/env python3
"""k_scan_cull at far = 8192 with the kernel cut short (a development build: RE_BUILD_DEFINES=-DRE_EXP_STAGES python render_engine_amd/build.py; then RE_EXP_STAGE_STOP=2: no stage B,
4: stage B without the expansion, 8: no candidate work at all -- the key stream and its tests alone): where the launch's time goes.
Only the launch is timed; the frames' results are not looked at."""
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
for far in (1000.0, 8192.0):
    p = R.Pipeline(16384, atomic, max_instances=max(1 << 16, len(ents) // 2))
    p.register_model_instances(ents)
    cam = R.Camera((c, c, c), (0, 0, -1), far).to_c()
    fl = (0 if os.environ.get("SS_SYNC") else F.CULL_ASYNC) | (F.CULL_FORCE_LARGE_PACK if far > 2000 else 0)      # SS_SYNC=1: synchronous frames (the device idles between the launches)
    p.run_frames(cam, 600, 0.016, fl, F.TICK_ASYNC); p.wait()
    p.timing_begin(64, 1, kernel="scan"); p.run_frames(cam, 64, 0.016, fl, F.TICK_ASYNC); p.wait(); t = p.timing_collect()
    print("RE_EXP_STAGE_STOP=%s far %g: scan mean %.2f median %.2f min %.2f us" % (os.environ.get("RE_EXP_STAGE_STOP", "0"), far, np.mean(t), np.median(t), np.min(t)), flush=True)
    p.close()
