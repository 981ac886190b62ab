"""wall time of re_collide on the 10M-entity world with 100k spinners (configs[2] plus CanCauseCollisions on the spinners)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import render_engine_amd as R
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 216
ents = R.synthetic.lattice_world(cells_per_axis=cells, first_cell=20, spinner_every=100)
ents["flags"][(ents["flags"] & R.F_HAS_ROTVEL) != 0] |= R.F_CAN_COLLIDE
p = R.Pipeline(16384, 64); p.register_model_instances(ents)
cam = R.Camera((8192, 8192, 8192), (0, 0, -1), 1000.0)
g = p.cull_and_pack(cam)
pairs, n = p.collide()
ts = []
for i in range(30):
    t = time.perf_counter(); p.collide(capacity=max(n, 1)); ts.append(time.perf_counter() - t)
print("entities", len(ents), "visible instances", g["total"], "collision invocations", n, "re_collide wall us: min %.1f median %.1f" % (min(ts) * 1e6, np.median(ts) * 1e6))
