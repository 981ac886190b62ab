"""Cost of re-bucketing movers in the 10M-entity world (development aid): sync frames, movers every k-th entity."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import render_engine_amd as R
from render_engine_amd import synthetic
axis = int(sys.argv[1]) if len(sys.argv) > 1 else 216
every = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
hop = len(sys.argv) > 3 and sys.argv[3] == "hop"          # movers that hop whole sections (no shared sections: the device-side re-bucket takes the batch)
atomic = 64
first = (16384 // atomic - axis) // 2
t0 = time.time()
ents = synthetic.hopping_lattice((axis,) * 3, first, atomic, every) if hop else synthetic.box_world((axis,) * 3, first_cell=first, atomic=atomic, mover_every=every)
p = R.Pipeline(16384, atomic, device=0, max_instances=1 << 16)
p.register_model_instances(ents)
print(f"world {len(ents)} entities, {int(((ents['flags'] & R.F_HAS_VEL) != 0).sum())} movers, setup {time.time() - t0:.1f} s", flush=True)
centre = [(first + axis / 2.0) * atomic] * 3
cam = R.Camera(centre, (0.0, 0.0, -1.0), 1000.0)
for f in range(8):
    t1 = time.perf_counter(); p.cull_and_pack(cam, copy=False); t2 = time.perf_counter()
    t = p.tick(1.0 if hop else 0.5, all_dynamic=True); t3 = time.perf_counter()
    print(f"frame {f}: cull {1e3 * (t2 - t1):8.2f} ms  tick+rebucket {1e3 * (t3 - t2):9.2f} ms  changed {t['n_changed']} rebucket {t['n_rebucket']} oob {t['n_out_of_bounds']} on-device batches {p.stats()['n_device_rebuckets']}", flush=True)
