#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench (2a + 2b), rocprofv3 kernel trace.  Logs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" ; timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
echo "== bench 2a"; timeout 600 python bench.py --steps 2000 --warmup 200 --fused > gpurun_out/bench_2a.json 2> gpurun_out/bench_2a.err; echo "rc=$?"; cat gpurun_out/bench_2a.json; tail -3 gpurun_out/bench_2a.err
echo "== bench 2b"; timeout 600 python bench.py --steps 2000 --warmup 200 --tasks 1024 --fused --no-cpu-baseline > gpurun_out/bench_2b.json 2> gpurun_out/bench_2b.err; echo "rc=$?"; cat gpurun_out/bench_2b.json; tail -3 gpurun_out/bench_2b.err
echo "== rocprofv3 kernel trace"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -o r01 -- python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof.err; echo "rc=$?"; cat gpurun_out/prof_bench.json
find gpurun_out/prof_r01 -name "*stats*" | head; for f in $(find gpurun_out/prof_r01 -name "*kernel_stats.csv"); do head -12 $f; done
