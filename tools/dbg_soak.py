import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import render_engine_amd as R, oracle as ro
from helpers import to_oracle, oracle_camera
from test_gpu_parity import random_changes
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
atomic = int(sys.argv[3]) if len(sys.argv) > 3 else 64
TRACE_ID = int(os.environ.get("TRACE_ID", "0")); TRACE_KEY = int(os.environ.get("TRACE_KEY", "0"), 16)
rng = np.random.default_rng(seed)
ents = R.synthetic.mixed_world(2500 + 500 * (seed % 4), seed=seed, spread=350.0 + 60.0 * (seed % 5), atomic=atomic)
ents["vel"] *= 8.0
p = R.Pipeline(16384, atomic, flags=flags); p.register_model_instances(ents)
w = ro.World(16384, atomic); w.register(to_oracle(ents))
frozen = set()
batches = []
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
    g = p.cull_and_pack(cam, emit_duplicates=bool(f % 2)); vk = w.cull(oc); o = w.render(oc, emit_duplicates=bool(f % 2))
    gi, oi = np.sort(g["ids"][:g["total"]]), np.sort(o["ids"])
    if TRACE_ID:
        s_, c_ = p.sections(), w.cells()
        info = ""
        if TRACE_KEY:
            i_ = np.searchsorted(s_["keys"], TRACE_KEY)
            if i_ < len(s_["keys"]) and s_["keys"][i_] == TRACE_KEY: info = f"owner nl/ns gpu {s_['n_local'][i_]} {s_['n_static'][i_]} cpu {c_['n_local'][i_]} {c_['n_static'][i_]} vis {TRACE_KEY in set(vk.tolist())} gpuvis {TRACE_KEY in set(p.visible_sections()[0].tolist())}"
            else: info = "owner section absent"
        print("trace frame", f, "gpu", int((gi == TRACE_ID).sum()), "cpu", int((oi == TRACE_ID).sum()), info)
    if len(gi) != len(oi) or (gi != oi).any():
        from collections import Counter
        cg, co = Counter(gi.tolist()), Counter(oi.tolist())
        for k in set(cg) | set(co):
            if cg[k] != co[k]:
                row = ents[ents["id"] == k][0]
                hist = [(bi, int(c["kind"]), int(c["component"])) for bi, b in enumerate(batches) for c in b if int(c["entity_id"]) == k]
                print("cpu mats t:", [tuple(np.round(m[12:15], 1)) for m in o["mats"][o["ids"] == k]], "gpu mats t:", [tuple(np.round(m[12:15], 1)) for m in g["mats"][:g["total"]][g["ids"][:g["total"]] == k]], "live pos", np.round(w.entity(int(k))["pos"], 1), "flags", hex(int(w.entity(int(k))["flags"])))
                print("frame", f, "id", k, "gpu", cg[k], "cpu", co[k], "upload flags", hex(int(row["flags"])), "alive", w.entity(int(k)) is not None, "history", hist, "lookup", w.lookup(int(k))[0], "stats", p.stats())
        break
    t = p.tick(0.04); w.tick(oc, 0.04)
    if not cmp(f"frame {f} after tick (rebucket {t['n_rebucket']})"): break
    if f % 4 == 1:
        ch = random_changes(R, ents, rng, 60, frozen)
        batches.append(ch)
        w.apply_changes(ch.view(ro.CHANGE_DT)); g = p.apply_changes(ch)
        if not cmp(f"frame {f} after changes {g}"): break
print("done")
