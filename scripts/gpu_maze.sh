#!/bin/bash
# MazeWorld: GPU parity tests + per-kernel timing on one box.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_maze.py -m gpu -x -q > gpurun_out/pytest_maze.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_maze.log
timeout 600 python scripts/bench_families.py --families maze64,maze256 2>&1 | tail -12
