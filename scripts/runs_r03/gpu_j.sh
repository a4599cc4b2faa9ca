#!/bin/bash
# round 3, visit J: fused mixed-batch step; the typing fixture on the device
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest mixed + maze"; timeout 1500 python -m pytest tests/test_gpu_mixed.py tests/test_gpu_maze.py tests/test_gpu_fullsize.py tests/test_gpu_cartpole.py -x -q > gpurun_out/j_pytest.log 2>&1; echo "rc=$?"; grep -n "passed\|failed\|Error\|assert" gpurun_out/j_pytest.log | head
echo "== mixed bench"; timeout 300 python scripts/bench_families.py --families mixed 2>/dev/null | cut -c1-700
