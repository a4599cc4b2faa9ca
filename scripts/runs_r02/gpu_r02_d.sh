#!/bin/bash
# round 2 main visit: full GPU suite, bench lines, envs sweep, PMC traffic of the AnyMDP step, kernel traces
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
echo "== bench 2a"; timeout 900 python bench.py --fused > gpurun_out/r02_bench_2a.json 2> gpurun_out/bench_2a.err; echo "rc=$?"; cut -c1-400 gpurun_out/r02_bench_2a.json
echo "== bench 2a driver flags"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_2a_s20.json 2>/dev/null; cut -c1-330 gpurun_out/r02_bench_2a_s20.json
echo "== bench 2b"; timeout 600 python bench.py --tasks 1024 --fused --no-cpu-baseline > gpurun_out/r02_bench_2b.json 2>/dev/null; cut -c1-330 gpurun_out/r02_bench_2b.json
echo "== sweep"; timeout 1500 python bench.py --sweep-envs 16384,32768,65536,131072,262144 --steps 500 --warmup 50 --sweep-out gpurun_out/r02_anymdp_envs_sweep.json > /dev/null 2> gpurun_out/sweep.err; echo "rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r02_anymdp_envs_sweep.json'))
for r in d['rows']: print(r)"
echo "== pmc traffic 2a"; bash scripts/gpu_pmc.sh "--steps 200 --warmup 20 --no-cpu-baseline" 2a 2>&1 | tail -4
echo "== kernel trace bench"
rm -rf gpurun_out/prof_r02
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02 -o r02 -- python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof.err; echo "rc=$?"
for f in $(find gpurun_out/prof_r02 -name "*kernel_stats.csv"); do head -6 $f | cut -c1-200; done
echo "== families"; timeout 900 python scripts/bench_families.py --steps 400 --warmup 40 --families linds,cartpole,acrobot,maze64,maze256,mixed,anymdp_tok > gpurun_out/r02_bench_families.jsonl 2> gpurun_out/fam.err; cut -c1-420 gpurun_out/r02_bench_families.jsonl
