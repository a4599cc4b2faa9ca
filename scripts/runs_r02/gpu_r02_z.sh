#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_maze.py tests/test_gpu_fullsize.py -q -m gpu -x -k "maze" > gpurun_out/pytest_z.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_z.log
for f in maze64 maze64_m1 maze64_m3; do timeout 600 python scripts/bench_families.py --families $f --steps 400 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['move_kernel'], d['us_per_step'])"; done
