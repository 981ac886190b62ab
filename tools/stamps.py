import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import render_engine_amd as R
from render_engine_amd import synthetic
ents = synthetic.lattice_world(216, 20)
p = R.Pipeline(16384, 64, max_instances=1 << 16)
p.register_model_instances(ents)
cam = R.Camera((8192, 8192, 8192), (0, 0, -1), 1000.0)
L = p._L
L.re_debug_get_stamps.restype = C.c_int; L.re_debug_get_stamps.argtypes = [C.c_void_p, C.c_void_p]
for i in range(5):
    p.cull_and_pack(cam); p.tick(0.016)
    st = np.zeros(8, np.uint64); L.re_debug_get_stamps(p._h, st.ctypes.data)
    s = st.astype(np.int64)
    names = ["K1b first block start -> last-block ticket", "ticket -> pack start(after fence)", "shared sections", "ranks", "scan", "scatter", "final"]
    t = [s[7] - s[6], s[0] - s[7], s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4]]
    print(" | ".join("%s %.2fus" % (n, v / 100.0) for n, v in zip(names, t)))
