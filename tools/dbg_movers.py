import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import render_engine_amd as R, oracle as ro
from helpers import to_oracle, oracle_camera
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ents = R.synthetic.mixed_world(n, seed=77, spread=1500.0); ents["vel"] *= 10.0
p = R.Pipeline(16384, 64, flags=flags); p.register_model_instances(ents)
w = ro.World(16384, 64); w.register(to_oracle(ents))
cam = R.Camera((8192, 8192, 9800), (0, 0, -1), 4000.0); oc = oracle_camera(cam)
for f in range(4):
    p.cull_and_pack(cam); w.cull(oc); w.render(oc)
    n_o, oob = w.tick(oc, 0.05); t = p.tick(0.05)
    if f < 3 and os.environ.get('EVERY') != '1': print('frame', f, 'tick', t, 'cpu', n_o, flush=True); continue
    s, c = p.sections(), w.cells()
    gk, ck = set(s["keys"].tolist()), set(c["keys"].tolist())
    print("frame", f, "tick", t, "cpu n", n_o, "sections gpu", len(gk), "cpu", len(ck), "extra on gpu", len(gk - ck), "missing on gpu", len(ck - gk), "stats", p.stats()["n_table_rebuilds"], flush=True)
    if gk != ck:
        ex = sorted(gk - ck)[:6]
        for k in ex:
            i = int(np.searchsorted(s["keys"], k))
            print("  extra", ro.unpack_key(k), "gpu nl/ns", s["n_local"][i], s["n_static"][i], "tight", s["tight"][i])
        sh = w.shared_sections(cap=64)
        linked = set(k for x in sh for k in x["keys"])
        print("  extra keys linked by a CPU shared section:", sum(1 for k in gk - ck if k in linked), "of", len(gk - ck))
        break
