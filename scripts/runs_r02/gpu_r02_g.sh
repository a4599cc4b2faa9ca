#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_maze.py tests/test_gpu_cartpole.py tests/test_gpu_fullsize.py tests/test_gpu_mixed.py tests/test_gpu_anymdp.py -m gpu -x -q > gpurun_out/pytest_g.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/pytest_g.log
timeout 600 python scripts/bench_families.py --steps 400 --warmup 40 --families maze64_m1,maze64_m3,maze64_m9,cartpole 2>/dev/null | cut -c1-360
