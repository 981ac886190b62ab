import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import render_engine_amd as R
from render_engine_amd import synthetic
ents = synthetic.lattice_world(216, 20)
p = R.Pipeline(16384, 64, max_instances=1 << 16)
p.register_model_instances(ents)
cam = R.Camera((8192, 8192, 8192), (0, 0, -1), 1000.0)
p.cull_and_pack(cam); p.tick(0.016)
L = p._L
L.re_debug_bench_cull.restype = C.c_int; L.re_debug_bench_cull.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
us = np.zeros(60, np.float32)
assert L.re_debug_bench_cull(p._h, 60, us.ctypes.data) == 0
print("k_cull_sections alone: median %.2f us  min %.2f us" % (np.median(us[10:]), us[10:].min()))
g = p.cull_and_pack(cam); print("still consistent:", g["total"], g["n_visible_sections"])
