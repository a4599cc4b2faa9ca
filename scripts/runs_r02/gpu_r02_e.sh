#!/bin/bash
# maze: nine-lane move kernel + fp32 filter: parity and timing
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_maze.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/pytest_maze.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/pytest_maze.log
timeout 900 python scripts/bench_families.py --steps 400 --warmup 40 --families maze64,maze64_f32,maze256,maze256_f32 > gpurun_out/r02_bench_maze.jsonl 2> gpurun_out/maze.err; cut -c1-330 gpurun_out/r02_bench_maze.jsonl; tail -3 gpurun_out/maze.err
