import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import render_engine_amd as R, oracle as ro
from helpers import to_oracle, oracle_camera
from test_gpu_parity import random_changes
ents = R.synthetic.mixed_world(3000, seed=21, spread=600.0)
p = R.Pipeline(16384, 64); p.register_model_instances(ents)
w = ro.World(16384, 64); w.register(to_oracle(ents))
rng = np.random.default_rng(5)
cams = [R.Camera((8192 + 40 * i, 8192, 8500 - 30 * i), (0.05 * i, 0, -1), 1200.0) for i in range(5)]
batches = []
def diff(cam, dups, tag):
    oc = oracle_camera(cam); w.cull(oc); o = w.render(oc, emit_duplicates=dups)
    g = p.cull_and_pack(cam, emit_duplicates=dups)
    gi, oi = np.sort(g["ids"][:g["total"]]), np.sort(o["ids"])
    if len(gi) != len(oi) or (gi != oi).any():
        from collections import Counter
        cg, co = Counter(gi.tolist()), Counter(oi.tolist())
        for k in set(cg) | set(co):
            if cg[k] != co[k]:
                e = w.entity(int(k)); row = ents[ents["id"] == k][0]
                hist = [(bi, int(c["kind"]), int(c["component"])) for bi, b in enumerate(batches) for c in b if int(c["entity_id"]) == k]
                print(tag, "id", k, "gpu", cg[k], "cpu", co[k], "upload flags", hex(int(row["flags"])), "alive", e is not None, "history (batch, kind, comp)", hist, "lookup", w.lookup(int(k))[0])
        return False
    return True
for f, cam in enumerate(cams):
    if not diff(cam, f % 2 == 1, f"frame {f}"): break
    w.tick(oracle_camera(cam), 0.016); p.tick(0.016)
    ch = random_changes(R, ents, rng, 150, set())
    if f == 2:
        far = np.zeros(6, R.CHANGE_DT)
        for i in range(6): far[i] = (R._capi.CHANGE_MODIFY, ents["id"][10 + i], R._capi.C_POSITION, 0, (-500.0, 8192.0, 20000.0, 0))
        ch = np.concatenate([ch, far])
    batches.append(ch)
    w.apply_changes(ch.view(ro.CHANGE_DT)); p.apply_changes(ch)
diff(cams[0], True, "final")
print("done")
