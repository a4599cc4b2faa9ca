#!/bin/bash
# re-profile the headline kernel for the present sources: PMC traffic (two passes) + kernel-trace stats of the default bench
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/gpu_pmc.sh 2>&1 | tail -6
rm -rf gpurun_out/stats_r
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_r -o stats -- python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/bench_r_prof.json 2> gpurun_out/bench_r_prof.err
f=$(find gpurun_out/stats_r -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r02_r_kernel_stats_anymdp_2a.csv
head -5 gpurun_out/r02_r_kernel_stats_anymdp_2a.csv
cat gpurun_out/bench_r_prof.json | cut -c1-400
