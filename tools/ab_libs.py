"""A/B of builds of the library inside ONE process (development aid): python tools/ab_libs.py <lib.so> <lib.so> ...
Between processes the synchronous frame differs by several per cent (where the result block and the thread landed); inside one process the median of 400 frames
repeats to +-0.1 us, so every library gets its own pipeline over the same world and the measurements alternate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import render_engine_amd as R
from render_engine_amd import synthetic, _capi

libs = sys.argv[1:] or [_capi.library_path()]
far = float(os.environ.get("AB_FAR", "1000"))
print("affinity:", bench.pin_near_gpu(0))
axis, atomic = 216, 64
first = (16384 // atomic - axis) // 2
ents = synthetic.lattice_world(cells_per_axis=axis, first_cell=first, atomic=atomic)
c = (first + axis / 2.0) * atomic
cam = R.Camera((c, c, c), (0.0, 0.0, -1.0), far).to_c()
pipes = []
for lib in libs:
    _capi._lib = None; os.environ["RE_HIP_LIBRARY"] = os.path.abspath(lib)
    p = R.Pipeline(16384, atomic, max_instances=(len(ents) // 2 if far > 2000 else 1 << 16)); p.register_model_instances(ents)
    bench.sync_frames(p, cam, 60)
    pipes.append(p)
rows = {lib: {"frame": [], "scan": [], "pack": []} for lib in libs}
for rep in range(6):
    for lib, p in zip(libs, pipes):
        us, _, _ = bench.sync_frames(p, cam, 400)
        rows[lib]["frame"].append(float(np.median(us)))
        rows[lib]["scan"].append(bench.launch_us(p, cam, "scan", 200, False))
        if far > 2000: rows[lib]["pack"].append(bench.launch_us(p, cam, "pack_large", 100, False))
for lib in libs:
    r = rows[lib]
    print("%-28s frame us %s | scan us %s%s" % (os.path.basename(lib), " ".join("%.2f" % v for v in r["frame"]), " ".join("%.2f" % v for v in r["scan"]),
                                               (" | pack us " + " ".join("%.2f" % v for v in r["pack"])) if r["pack"] else ""), flush=True)
for p in pipes: p.close()
