import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import render_engine_amd as R, oracle as ro
from helpers import to_oracle, oracle_camera
from test_gpu_parity import random_changes
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ents = R.synthetic.mixed_world(2500 + 500 * (seed % 4), seed=seed, spread=350.0 + 60.0 * (seed % 5))
ents["vel"] *= 8.0
p = R.Pipeline(16384, 64, flags=flags); p.register_model_instances(ents)
w = ro.World(16384, 64); w.register(to_oracle(ents))
frozen = set(int(i) for i in ents["id"][(ents["flags"] & R.F_STATIC) != 0])
def cmp(tag):
    s, c = p.sections(), w.cells()
    if not np.array_equal(s["keys"], c["keys"]): print(tag, "KEYS differ"); return False
    tight = np.stack([c["tight"][k] for k in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")], axis=1)
    bad = np.where((s["tight"] != tight).any(axis=1))[0]
    for i in bad[:6]:
        print(tag, "section", ro.unpack_key(s["keys"][i]), "nl/ns gpu", s["n_local"][i], s["n_static"][i], "cpu", c["n_local"][i], c["n_static"][i], "nshared", c["n_shared"][i], "\n   gpu", s["tight"][i], "\n   cpu", tight[i])
    return len(bad) == 0
for f in range(48):
    pos = (8192 + rng.uniform(-300, 300), 8192 + rng.uniform(-200, 200), 8192 + rng.uniform(-100, 500))
    d = rng.uniform(-1, 1, 3); d[2] -= 1.5
    cam = R.Camera(pos, tuple(d / np.linalg.norm(d)), float(rng.choice([600.0, 1000.0, 2500.0])))
    oc = oracle_camera(cam)
    p.cull_and_pack(cam, copy=False); w.cull(oc); w.render(oc)
    t = p.tick(0.04); w.tick(oc, 0.04)
    if not cmp(f"frame {f} after tick (rebucket {t['n_rebucket']})"): break
    if f % 4 == 1:
        ch = random_changes(R, ents, rng, 60, frozen)
        w.apply_changes(ch.view(ro.CHANGE_DT)); g = p.apply_changes(ch)
        if not cmp(f"frame {f} after changes {g}"): break
print("done")
