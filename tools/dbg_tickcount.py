import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import render_engine_amd as R, oracle as ro
from helpers import to_oracle, oracle_camera
from test_gpu_parity import collision_world
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    seed, n, spread, atomic = 9, 4000, 260.0, 64
    ents = collision_world(R, n, seed, spread, atomic); ents["vel"] *= 3.0
    p = R.Pipeline(16384, atomic); p.register_model_instances(ents)
    w = ro.World(16384, atomic); w.register(to_oracle(ents))
    rng = np.random.default_rng(seed)
    for f in range(10):
        pos = (8192 + rng.uniform(-spread, spread) * 0.6, 8192 + rng.uniform(-spread, spread) * 0.6, 8192 + rng.uniform(-0.3, 1.2) * spread)
        d = rng.uniform(-1, 1, 3); d[2] -= 1.2
        cam = R.Camera(pos, tuple(d / np.linalg.norm(d)), float(rng.choice([400.0, 1500.0]))); oc = oracle_camera(cam)
        g = p.cull_and_pack(cam); w.cull(oc); w.render(oc)
        if os.environ.get("WITH_COLLIDE", "1") == "1":
            w.collide(oc); p.collide()
        n_o, _ = w.tick(oc, 0.05); t = p.tick(0.05)
        if t["n_changed"] != n_o:
            bad += 1
            # which entities differ?
            nd = 0
            for e in ents[::7]:
                eid = int(e["id"]); o = w.entity(eid)
                if o is None: continue
                gp = p.read_component(eid, R._capi.C_POSITION)
                if not np.array_equal(np.asarray(gp, np.float32)[:3], o["pos"]): nd += 1
            print("rep", rep, "frame", f, "gpu n_changed", t["n_changed"], "cpu", n_o, "sampled entities with wrong position:", nd, "of", len(ents[::7]))
            break
    p.close(); w.close()
print("mismatching runs:", bad)
