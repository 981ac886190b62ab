import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from collections import Counter
import render_engine_amd as R, oracle as ro
from helpers import to_oracle, oracle_camera
from test_gpu_parity import collision_world
seed, n, spread, atomic = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
ents = collision_world(R, n, seed, spread, atomic); ents["vel"] *= 3.0
p = R.Pipeline(16384, atomic); p.register_model_instances(ents)
w = ro.World(16384, atomic); w.register(to_oracle(ents))
rng = np.random.default_rng(seed)
pos = (8192 + rng.uniform(-spread, spread) * 0.6, 8192 + rng.uniform(-spread, spread) * 0.6, 8192 + rng.uniform(-0.3, 1.2) * spread)
d = rng.uniform(-1, 1, 3); d[2] -= 1.2
cam = R.Camera(pos, tuple(d / np.linalg.norm(d)), float(rng.choice([400.0, 1500.0]))); oc = oracle_camera(cam)
p.cull_and_pack(cam); vk = w.cull(oc); w.render(oc)
want = Counter(map(tuple, w.collide(oc).tolist())); got_a, nt = p.collide(); got = Counter(map(tuple, got_a.tolist()))
print("cpu", sum(want.values()), "gpu", nt, "distinct cpu", len(want), "gpu", len(got))
ratio = Counter()
for k in set(want) | set(got): ratio[(want[k], got[k])] += 1
print("(cpu count, gpu count) histogram:", sorted(ratio.items()))
fl = {int(e["id"]): int(e["flags"]) for e in ents}
shown = 0
for k in set(want) | set(got):
    if want[k] != got[k] and shown < 8:
        a, b = k
        print(k, "cpu", want[k], "gpu", got[k], "flags", hex(fl[a]), hex(fl[b]), "lookup", w.lookup(a)[0], w.lookup(b)[0],
              "vis mult a", [int((vk == kk).sum()) for kk in w.lookup(a)[1]], "b", [int((vk == kk).sum()) for kk in w.lookup(b)[1]])
        shown += 1
