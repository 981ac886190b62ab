"""Host issue rate vs GPU completion rate of the pipelined frame loop (development aid)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import render_engine_amd as R
from render_engine_amd import synthetic
axis = int(sys.argv[1]) if len(sys.argv) > 1 else 216
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
atomic = 64
first = (16384 // atomic - axis) // 2
ents = synthetic.box_world((axis,) * 3, first_cell=first, atomic=atomic)
p = R.Pipeline(16384, atomic, device=0, max_instances=1 << 16)
p.register_model_instances(ents)
centre = [(first + axis / 2.0) * atomic] * 3
camc = R.Camera(centre, (0.0, 0.0, -1.0), 1000.0).to_c()
for _ in range(20):
    p.cull_and_pack(camc, asynchronous=True, copy=False); p.tick(0.016, asynchronous=True)
p.wait()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(K):
        p.cull_and_pack(camc, asynchronous=True, copy=False); p.tick(0.016, asynchronous=True)
    t1 = time.perf_counter()
    p.wait()
    t2 = time.perf_counter()
    print(f"axis {axis}: issue {1e6*(t1-t0)/K:.1f} us/frame, total {1e6*(t2-t0)/K:.1f} us/frame", flush=True)
