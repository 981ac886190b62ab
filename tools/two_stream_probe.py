"""feasibility probe: do the one-launch frames of two independent contexts (two HIP streams) overlap on the GPU?
two pipelines over the same 10M-entity world, frames issued alternately from one thread"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import render_engine_amd as R
ents = R.synthetic.lattice_world(cells_per_axis=216, first_cell=20)
ps = []
for k in range(2):
    p = R.Pipeline(16384, 64, max_instances=1 << 16); p.register_model_instances(ents); ps.append(p)
cam = R.Camera((8192, 8192, 8192), (0, 0, -1), 1000.0).to_c()
def run(pipes, n):
    for p in pipes: p.cull_and_pack(cam, asynchronous=True, copy=False, defer_pack=True); p.tick(0.016, asynchronous=True)
    for p in pipes: p.wait()
    t = time.perf_counter()
    for i in range(n):
        for p in pipes:
            p.cull_and_pack(cam, asynchronous=True, copy=False, defer_pack=True); p.tick(0.016, asynchronous=True)
    for p in pipes: p.wait()
    return (time.perf_counter() - t) / (n * len(pipes)) * 1e6
for rep in range(2):
    print("one context : %.2f us per frame" % run(ps[:1], 400))
    print("two contexts: %.2f us per frame (aggregate)" % run(ps, 400))
