#!/bin/bash
# round 2, first GPU visit: parity (incl. the new full-size tests), bench lines (default + the driver's flags),
# LinDS counters
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
echo "== bench default"; timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "rc=$?"; cut -c1-1500 gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
echo "== bench driver flags"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_s20.json 2> gpurun_out/bench_s20.err; echo "rc=$?"; cut -c1-700 gpurun_out/bench_s20.json
rocprofv3 -L > gpurun_out/rocprof_counters.txt 2>&1
echo "== linds counters"
bash scripts/pmc_kernel.sh linds linds_step scripts/bench_families.py --families linds --steps 60 --warmup 10 2>&1 | tail -70
