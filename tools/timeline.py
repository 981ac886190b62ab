"""Per-wave timeline of k_scan_cull (needs a -DRE_EXP_STAMPS build: RE_HIP_LIBRARY=render_engine_amd/lib/exp_stamps.so)."""
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
nw = (axis ** 3 + 511) // 512
buf = np.zeros((nw, 8), dtype=np.uint64)
for i in range(20):
    p.cull_and_pack(camc, copy=False); p.tick(0.016)
lib.re_debug_get_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
assert lib.re_debug_get_timeline(p._h, buf.ctypes.data, nw) == 0
t = buf.astype(np.int64)
t0 = t[:, 0].min()
start, keys, end, cand = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0, t[:, 3] != 0
print(f"waves {nw}, candidate waves {cand.sum()}, kernel span {end.max():.2f} us")
print(f"non-candidate waves: start->keys median {np.median((keys - start)[~cand]):.2f} us, keys->end median {np.median((end - keys)[~cand]):.2f} us")
print(f"candidate waves:     start->keys median {np.median((keys - start)[cand]):.2f} us, keys->end median {np.median((end - keys)[cand]):.2f} us  max {(end - keys)[cand].max():.2f}")
pred, emit = (t[:, 4] - t0) / 100.0, (t[:, 5] - t0) / 100.0
print(f"candidate waves: keys->predicates done median {np.median((pred - keys)[cand]):.2f} us, predicates->emitted median {np.median((emit - pred)[cand]):.2f} us, emitted->end median {np.median((end - emit)[cand]):.2f} us")
print(f"candidate waves start between {start[cand].min():.2f} and {start[cand].max():.2f} us, end by {end[cand].max():.2f} us")
print(f"last non-candidate wave ends {end[~cand].max():.2f} us; waves starting after 10 us: {(start > 10).sum()}")
hist, edges = np.histogram(start, bins=np.arange(0, end.max() + 1, 1.0))
print("wave starts per us:", hist.tolist())
hist, edges = np.histogram(end, bins=np.arange(0, end.max() + 1, 1.0))
print("wave ends per us:  ", hist.tolist())

