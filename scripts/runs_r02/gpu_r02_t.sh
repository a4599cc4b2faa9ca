#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_maze.py tests/test_gpu_maze_agent.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -8
timeout 600 python scripts/bench_families.py --families maze64 --steps 400 2>/dev/null | cut -c1-400
