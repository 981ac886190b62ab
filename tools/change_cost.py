"""Cost of change-request batches (re_apply_changes) in the 10M-entity world (development aid): the per-frame re-insertion of the user entity at the camera
position (logic_flow.rs:246-251 -> one Position change), the same crossing a section border, and a batch of 64 Position changes of dynamic entities."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import render_engine_amd as R
from render_engine_amd import synthetic
axis = int(sys.argv[1]) if len(sys.argv) > 1 else 216
atomic = 64
first = (16384 // atomic - axis) // 2
ents = synthetic.box_world((axis,) * 3, first_cell=first, atomic=atomic, mover_every=1000)
p = R.Pipeline(16384, atomic, device=0, max_instances=1 << 16)
p.register_model_instances(ents)
dyn = ents["id"][(ents["flags"] & R.F_HAS_VEL) != 0]
print(f"world {len(ents)} entities, {len(dyn)} dynamic", flush=True)
c0 = (first + axis / 2.0) * atomic
cam = R.Camera((c0, c0, c0), (0.0, 0.0, -1.0), 1000.0)
F = R._capi
def one(pos, eid):
    ch = np.zeros(1, R.CHANGE_DT); ch[0] = (F.CHANGE_MODIFY, eid, F.C_POSITION, 0, (pos[0], pos[1], pos[2], 0))
    return ch
user = int(dyn[len(dyn) // 2])
base = np.array([c0 + 5.0, c0 + 5.0, c0 + 5.0], np.float32)
for label, step in (("inside its section", 0.01), ("crossing a section border every frame", 64.0)):
    ts = []
    for f in range(40):
        p.cull_and_pack(cam, copy=False)
        pos = base + np.float32(step * (f % 2 if step > 1 else f))
        t0 = time.perf_counter(); p.apply_changes(one(pos, user)); ts.append(time.perf_counter() - t0)
        p.tick(0.0001)
    ts = np.array(ts[8:]) * 1e6
    print(f"one Position change, {label}: median {np.median(ts):9.1f} us  min {ts.min():9.1f}  max {ts.max():9.1f}   host re-buckets so far {p.stats()['n_host_rebuckets']}", flush=True)
rng = np.random.default_rng(1)
ts = []
for f in range(24):
    p.cull_and_pack(cam, copy=False)
    ids = rng.choice(dyn, 64, replace=False)
    ch = np.zeros(64, R.CHANGE_DT)
    for k in range(64):
        ch[k] = (F.CHANGE_MODIFY, ids[k], F.C_POSITION, 0, (c0 + rng.uniform(-3000, 3000), c0 + rng.uniform(-3000, 3000), c0 + rng.uniform(-3000, 3000), 0))
    t0 = time.perf_counter(); p.apply_changes(ch); ts.append(time.perf_counter() - t0)
    p.tick(0.0001)
ts = np.array(ts[4:]) * 1e6
print(f"64 Position changes to random places: median {np.median(ts):9.1f} us  min {ts.min():9.1f}  max {ts.max():9.1f}", flush=True)
