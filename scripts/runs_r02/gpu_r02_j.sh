#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_maze.py -m gpu -x -q > gpurun_out/pytest_j.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/pytest_j.log
timeout 600 python scripts/bench_families.py --steps 400 --warmup 40 --families maze64_m3,maze64_m9 2> gpurun_out/fam_j.err | cut -c1-330
