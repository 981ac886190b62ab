import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import render_engine_amd as R, oracle as ro
from helpers import to_oracle, oracle_camera
from test_gpu_parity import random_changes, check_frame, build_pair
bad = 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for rep in range(reps):
    seed, atomic = [3, 29, 303, 11][rep % 4], 64
    rng = np.random.default_rng(seed)
    ents = R.synthetic.mixed_world(2500 + 500 * (seed % 4), seed=seed, spread=350.0 + 60.0 * (seed % 5), atomic=atomic)
    ents["vel"] *= 8.0
    p, w = build_pair(R, ents, atomic=atomic)
    for f in range(48):
        pos = (8192 + rng.uniform(-300, 300), 8192 + rng.uniform(-200, 200), 8192 + rng.uniform(-100, 500))
        d = rng.uniform(-1, 1, 3); d[2] -= 1.5
        cam = R.Camera(pos, tuple(d / np.linalg.norm(d)), float(rng.choice([600.0, 1000.0, 2500.0]))); oc = oracle_camera(cam)
        if f % 3:
            p.cull_and_pack(cam, asynchronous=True, copy=False); p.tick(0.04, asynchronous=True)
            w.cull(oc); w.render(oc); w.tick(oc, 0.04)
        else:
            check_frame(R, p, w, cam, bool(f % 2))
            n_o, oob_o = w.tick(oc, 0.04); t = p.tick(0.04)
            if t["n_changed"] != n_o:
                bad += 1
                nd = 0; nf = 0
                for e in ents:
                    eid = int(e["id"]); o = w.entity(eid)
                    if o is None: continue
                    gp = np.asarray(p.read_component(eid, R._capi.C_POSITION), np.float32)[:3]
                    gf = int(np.asarray(p.read_component(eid, R._capi.C_FLAGS)).view(np.uint32)[0]) if hasattr(np.asarray(p.read_component(eid, R._capi.C_FLAGS)), "view") else 0
                    if not np.array_equal(gp, o["pos"]): nd += 1
                print("rep", rep, "seed", seed, "frame", f, "gpu n_changed", t["n_changed"], "cpu", n_o, "rebucket", t["n_rebucket"], "| entities with wrong position:", nd, "of", len(ents), flush=True)
                break
        if f % 4 == 1:
            ch = random_changes(R, ents, rng, 60, set())
            w.apply_changes(ch.view(ro.CHANGE_DT)); p.apply_changes(ch)
    p.close(); w.close()
print("mismatching runs:", bad, "of", reps)
