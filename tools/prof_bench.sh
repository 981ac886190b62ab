#!/bin/bash
# usage (on the GPU box): tools/prof_bench.sh <outdir-under-gpurun_out> [bench args...]
# kernel trace + stats of bench.py; prints the per-kernel averages
out=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$out; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$out -- python $root/bench.py "$@" > $root/gpurun_out/$out.log 2>&1
cd $root
python - <<PY
import csv, glob
import shutil
f = glob.glob("/tmp/prof_$out/*/*kernel_stats.csv")[0]
shutil.copy(f, "gpurun_out/${out}_kernel_stats.csv")      # (the raw trace stays on the box: gpurun_out is copied back only up to 64 MiB)
for r in csv.DictReader(open(f)):
    print("%-28s calls %5s avg %9.2f us  min %8.2f  max %8.2f" % (r["Name"].split("(")[0][:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
tail -c 300 gpurun_out/$out.log | head -c 300
