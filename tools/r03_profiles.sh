#!/bin/bash
# round 3: the rocprofv3 summaries behind DESIGN.md section 7 / profiles/r03_* (run on the GPU box: gpurun -- tools/r03_profiles.sh)
# kernel-trace + stats per leg (each leg alone, so that a kernel name means one workload), then FETCH_SIZE / WRITE_SIZE in separate --pmc passes
set -x
tools/prof_bench.sh r3_prof_head --config visible --no-extras --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/r3_prof_head.txt 2>&1 || exit 1
tools/prof_bench.sh r3_prof_far --config visible --far 8192 --no-extras --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r3_prof_far.txt 2>&1 || exit 1
tools/prof_bench.sh r3_prof_c2 --config visible --spinner-every 100 --tick-all --no-extras --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/r3_prof_c2.txt 2>&1 || exit 1
tools/prof_bench.sh r3_prof_dense --config visible --spinner-every 10 --tick-all --no-extras --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r3_prof_dense.txt 2>&1 || exit 1
tools/prof_bench.sh r3_prof_light --config lighting --steps 50 --warmup 150 > gpurun_out/r3_prof_light.txt 2>&1 || exit 1
tools/pmc_bench.sh r3 --config visible --no-extras > gpurun_out/r3_pmc.txt 2>&1 || exit 1
tools/pmc_bench.sh r3_far --config visible --far 8192 --no-extras > gpurun_out/r3_pmc_far.txt 2>&1 || exit 1
tools/pmc_bench.sh r3_c2 --config visible --spinner-every 100 --tick-all --no-extras > gpurun_out/r3_pmc_c2.txt 2>&1 || exit 1
tools/pmc_bench.sh r3_dense --config visible --spinner-every 10 --tick-all --no-extras > gpurun_out/r3_pmc_dense.txt 2>&1 || exit 1
tools/pmc_bench.sh r3_light --config lighting > gpurun_out/r3_pmc_light.txt 2>&1 || exit 1
echo done
