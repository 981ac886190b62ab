"""far = 8192 on the configs[1] world: own launch time of k_scan_cull and k_pack_large for the library RE_HIP_LIBRARY points at
(A/B of builds with other -DRE_PACK_THREADS / -DRE_PACK_CHUNK; see tools/README.md)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import render_engine_amd as R
from render_engine_amd import synthetic

axis, atomic = 216, 64
first = (16384 // atomic - axis) // 2
ents = synthetic.lattice_world(cells_per_axis=axis, first_cell=first, atomic=atomic)
c = (first + axis / 2.0) * atomic
p = R.Pipeline(16384, atomic, max_instances=len(ents) // 2)
p.register_model_instances(ents)
cam = R.Camera((c, c, c), (0.0, 0.0, -1.0), 8192.0).to_c()
bench.sync_frames(p, cam, 8)
out = {"lib": os.path.basename(os.environ.get("RE_HIP_LIBRARY", "default"))}
for rep in range(3):
    out[f"pack_sync_{rep}"] = round(bench.launch_us(p, cam, "pack_large", 48, False), 2)
    out[f"pack_async_{rep}"] = round(bench.launch_us(p, cam, "pack_large", 96, True), 2)
out["scan_sync"] = round(bench.launch_us(p, cam, "scan", 48, False), 2)
kt = bench.kernel_times(p, cam, 12); out["kernel_us"] = {k: round(v, 2) for k, v in kt.items()}
print(json.dumps(out))
p.close()
