#!/bin/bash
# round 3, visit Q: tiled packed textures (ray caster) — parity, then both filters at 64x64 and 256x256
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_maze.py tests/test_gpu_maze_agent.py tests/test_gpu_fullsize.py -q -x > gpurun_out/q_pytest.log 2>&1; echo "rc=$?"; grep -n "passed\|failed" gpurun_out/q_pytest.log
timeout 900 python scripts/bench_families.py --families maze64,maze64_f32,maze256,maze256_f32 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['workload'][-16:], d['filter'], d['us_per_step'])"
