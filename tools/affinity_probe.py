"""Does the CPU the calling thread runs on matter for the synchronous frame?  (development aid)
Prints the process's CPU set, the GPU's NUMA node / local CPUs from sysfs, and the median synchronous frame for a few affinity choices."""
import os, sys, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import render_engine_amd as R
from render_engine_amd import synthetic

def cpulist(s):
    out = set()
    for part in s.strip().split(","):
        if not part: continue
        a, _, b = part.partition("-"); out.update(range(int(a), int(b or a) + 1))
    return out

allowed = os.sched_getaffinity(0)
print("allowed cpus:", sorted(allowed))
p0 = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (p0.pci_domain_id, p0.pci_bus_id, p0.pci_device_id)
local = set(); node = None
try:
    node = open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip()
    local = cpulist(open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read())
except OSError as e:
    print("sysfs:", e)
print("gpu", bdf, "numa node", node, "local cpus", len(local), "allowed & local:", sorted(allowed & local)[:8], "...")
axis, atomic = 216, 64
first = (16384 // atomic - axis) // 2
ents = synthetic.lattice_world(cells_per_axis=axis, first_cell=first, atomic=atomic)
c = (first + axis / 2.0) * atomic
p = R.Pipeline(16384, atomic, max_instances=1 << 16); p.register_model_instances(ents)
cam = R.Camera((c, c, c), (0.0, 0.0, -1.0), 1000.0).to_c()
bench.sync_frames(p, cam, 50)
def run(label, cpus):
    if cpus:
        try: os.sched_setaffinity(0, cpus)
        except OSError as e: print(label, "setaffinity failed:", e); return
    meds = []
    for rep in range(5):
        us, _, _ = bench.sync_frames(p, cam, 400); meds.append(float(np.median(us)))
    print("%-34s median sync frame us: %s" % (label, " ".join("%.2f" % m for m in meds)), flush=True)
run("default", None)
a = sorted(allowed)
run("one cpu (first allowed)", {a[0]})
run("one cpu (last allowed)", {a[-1]})
if allowed & local: run("local cpus", allowed & local); run("one local cpu", {sorted(allowed & local)[0]})
if allowed - local: run("one non-local cpu", {sorted(allowed - local)[0]})
os.sched_setaffinity(0, allowed); run("default again", None)
p.close()
