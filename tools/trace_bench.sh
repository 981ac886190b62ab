#!/bin/bash
# kernel timeline of the last frames of bench.py (durations and gaps)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_bench
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/trace_bench.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/trace_bench/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-260:-240]
prev = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-36s dur %7.2f us  gap %7.2f us" % (r["Kernel_Name"][:36], (e - s) / 1000.0, (s - prev) / 1000.0 if prev else 0.0))
    prev = e
PY
