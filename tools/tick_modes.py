"""k_tick duration by calling mode (run under rocprofv3 --kernel-trace; tools/README.md): 40 frames each of
A: cull sync, tick sync   B: cull sync, tick async + wait   C: cull async, tick async, wait per frame   D: cull async, tick async, no wait (pipelined)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import render_engine_amd as R
from render_engine_amd import synthetic
axis = int(sys.argv[1]) if len(sys.argv) > 1 else 216
first = (256 - axis) // 2
ents = synthetic.lattice_world(cells_per_axis=axis, first_cell=first, spinner_every=100)
p = R.Pipeline(16384, 64, max_instances=1 << 16)
p.register_model_instances(ents)
c = (first + axis / 2.0) * 64
cam = R.Camera((c, c, c), (0, 0, -1), 1000.0).to_c()
N = 40
for mode in "ABCD":
    for f in range(N):
        if mode == "A":
            p.cull_and_pack(cam, copy=False); p.tick(0.016, all_dynamic=True)
        elif mode == "B":
            p.cull_and_pack(cam, copy=False); p.tick(0.016, all_dynamic=True, asynchronous=True); p.wait()
        elif mode == "C":
            p.cull_and_pack(cam, copy=False, asynchronous=True); p.tick(0.016, all_dynamic=True, asynchronous=True); p.wait()
        else:
            p.cull_and_pack(cam, copy=False, asynchronous=True); p.tick(0.016, all_dynamic=True, asynchronous=True)
    p.wait()
    time.sleep(0.01)
print("done")
