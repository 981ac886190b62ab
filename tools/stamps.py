"""Phase stamps of k_cull_sections (needs a -DRE_EXP_STAMPS build: RE_HIP_LIBRARY=render_engine_amd/lib/exp_stamps.so)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import render_engine_amd as R
from render_engine_amd import synthetic, _capi
axis = int(sys.argv[1]) if len(sys.argv) > 1 else 216
atomic = 64
first = (16384 // atomic - axis) // 2
ents = synthetic.box_world((axis,) * 3, first_cell=first, atomic=atomic)
p = R.Pipeline(16384, atomic, device=0, max_instances=1 << 16)
p.register_model_instances(ents)
centre = [(first + axis / 2.0) * atomic] * 3
camc = R.Camera(centre, (0.0, 0.0, -1.0), 1000.0).to_c()
lib = _capi.load()
lib.re_debug_get_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
acc = []
for i in range(30):
    p.cull_and_pack(camc, copy=False); p.tick(0.016)
    out = (ctypes.c_ulonglong * 8)()
    lib.re_debug_get_stamps(p._h, out)
    s = np.array(list(out), dtype=np.int64)
    if i >= 10:
        acc.append(s)
a = np.array(acc)
names = {6: "block0 start", 7: "last block after ticket", 0: "pack start", 1: "hist zeroed", 2: "ranks done", 3: "scan done", 5: "results written"}
base = a[:, 6]
for k in (6, 7, 0, 1, 2, 3, 5):
    print(f"{names[k]:28s} +{np.median(a[:, k] - base) / 100.0:7.2f} us")
print(p.timings_us())
