#!/bin/bash
# usage: tools/trace_lib.sh <lib.so> [axis]  -- per-kernel average durations of the pipelined frame loop with an alternative build
cd /tmp && export TMPDIR=/tmp
L=$1; AX=${2:-216}; T=$(basename $L .so)
export RE_HIP_LIBRARY=$GRAFT_REPO_ROOT/$L
rm -rf $GRAFT_REPO_ROOT/gpurun_out/tl_$T
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_$T -- python3 $GRAFT_REPO_ROOT/tools/issue_rate.py $AX 200 > $GRAFT_REPO_ROOT/gpurun_out/tl_$T.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/tl_$T/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) > 100: print("$T %-24s calls %5s avg %8.2f us min %8.2f" % (r["Name"].split("(")[0][:24], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
grep "total" $GRAFT_REPO_ROOT/gpurun_out/tl_$T.log | tail -1
