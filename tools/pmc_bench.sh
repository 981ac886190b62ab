#!/bin/bash
# usage (on the GPU box): tools/pmc_bench.sh <tag> [bench args...] ; collects FETCH_SIZE and WRITE_SIZE in two separate passes
# (default bench args: the headline leg alone, so that a kernel name means one workload)
tag=$1; shift
args=${@:---config visible --no-extras}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${tag}_$ctr; rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_${tag}_$ctr -- python $root/bench.py --steps 60 --warmup 10 --no-cpu-baseline $args > $root/gpurun_out/pmc_${tag}_$ctr.log 2>&1
done
cd $root
python - <<PY
import csv, glob, json, collections
out = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pmc_${tag}_%s/*/*counter_collection.csv" % ctr)
    if not f:
        print("no counter file for", ctr); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") == ctr:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if any(n in k for n in ("k_scan_cull", "k_pack_small", "k_pack_large", "k_tick", "k_probe", "k_emit", "k_group_scan", "k_deferred_lighting")):
            v = v[len(v) // 3:]                      # steady state
            out.setdefault(k, {})[ctr] = sum(v) / len(v)
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc_${tag}.json", "w"), indent=1)
PY
