import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import render_engine_amd as R, oracle as ro
from helpers import to_oracle, oracle_camera
ents = R.synthetic.mixed_world(3000, seed=21, spread=600.0)
p = R.Pipeline(16384, 64); p.register_model_instances(ents)
w = ro.World(16384, 64); w.register(to_oracle(ents))
cam = R.Camera((8192, 8192, 8500), (0, 0, -1), 1200.0)
for force in (False, True):
    oc = oracle_camera(cam); w.cull(oc); o = w.render(oc)
    g = p.cull_and_pack(cam, force_large_pack=force)
    gi, oi = np.sort(g["ids"][:g["total"]]), np.sort(o["ids"])
    missing = np.setdiff1d(oi, gi); extra = np.setdiff1d(gi, oi)
    print("force_large", force, "gpu", g["total"], "cpu", o["total"], "missing", missing, "extra", extra)
    for m in missing:
        e = w.entity(int(m)); print(m, "lookup", e.get("lookup"), "static", e["flags"] & 1, e["aabb"], w.lookup(int(m)) if hasattr(w, "lookup") else "")
    p.tick(0.016); w.tick(oc, 0.016)
