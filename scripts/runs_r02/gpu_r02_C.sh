#!/bin/bash
# after the bucket search: default bench, driver-flag bench, PMC traffic + kernel stats, envs sweep
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_C.json 2> gpurun_out/bench_C.err; cut -c1-1500 gpurun_out/bench_C.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_C20.json 2> gpurun_out/bench_C20.err; cut -c1-400 gpurun_out/bench_C20.json
bash scripts/gpu_pmc.sh 2>&1 | tail -3
rm -rf gpurun_out/stats_C
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_C -o stats -- python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/bench_C_prof.json 2> gpurun_out/bench_C_prof.err
f=$(find gpurun_out/stats_C -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02_C_kernel_stats_anymdp_2a.csv; head -4 gpurun_out/r02_C_kernel_stats_anymdp_2a.csv
timeout 1500 python bench.py --sweep-envs 16384,32768,65536,131072,262144 --sweep-out gpurun_out/r02_C_anymdp_envs_sweep.json --steps 1000 --warmup 100 --no-cpu-baseline 2>&1 | tail -3
