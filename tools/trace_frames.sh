#!/bin/bash
# kernel-trace the pipelined frame loop and print per-kernel durations and gaps for the last frames
cd /tmp && export TMPDIR=/tmp
AX=${1:-216}
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_$AX
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$AX -- python3 $GRAFT_REPO_ROOT/tools/issue_rate.py $AX 200 > $GRAFT_REPO_ROOT/gpurun_out/trace_$AX.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/trace_$AX/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-24:]
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{r["Kernel_Name"][:40]:40s} grid {r.get("Grid_Size_X","?"):>8s} dur {(e-s)/1000:7.2f} us  gap {((s-prev)/1000 if prev else 0):7.2f} us')
    prev = e
PY
