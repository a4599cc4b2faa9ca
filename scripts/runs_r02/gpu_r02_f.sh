#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_maze.py -m gpu -x -q -k "nine_lane or golden_trajectory or batch_vs_oracle" > gpurun_out/pytest_maze2.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/pytest_maze2.log
timeout 600 python scripts/bench_families.py --steps 400 --warmup 40 --families maze64,maze64_f32 2>/dev/null | cut -c1-330
bash scripts/pmc_kernel.sh maze_exact maze_raycast scripts/bench_families.py --families maze64 --steps 120 --warmup 10 2>&1 | grep -A45 "^void maze_raycast" | grep "SQ_\|VGPR\|hbm\|LDS\|Scratch\|^void" | head -40
bash scripts/pmc_kernel.sh maze_f32 maze_raycast scripts/bench_families.py --families maze64_f32 --steps 120 --warmup 10 2>&1 | grep -A45 "^void maze_raycast" | grep "SQ_\|VGPR\|hbm\|LDS\|Scratch\|^void" | head -40
