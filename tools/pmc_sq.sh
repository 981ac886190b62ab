#!/bin/bash
# usage (on the GPU box): tools/pmc_sq.sh <tag> [bench args...] ; wave / instruction counters of the frame kernels (per-dispatch means), one --pmc pass per group
tag=$1; shift
args=${@:---spinner-every 100 --tick-all --no-extras}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" "SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $root/gpurun_out/sq_${tag}_$i -- python $root/bench.py --steps 40 --warmup 5 --no-cpu-baseline $args > $root/gpurun_out/sq_${tag}_$i.log 2>&1
done
cd $root
python - <<PY
import csv, glob, json, collections
out = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/sq_${tag}_*/*/*counter_collection.csv"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, ctr), v in acc.items():
        if any(n in k for n in ("k_scan_cull", "k_pack_small", "k_pack_large", "k_tick")) and "publish" not in k:
            v = v[len(v) // 3:]
            out[k][ctr] = sum(v) / len(v)
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/sq_${tag}.json", "w"), indent=1)
PY
