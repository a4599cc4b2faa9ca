#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_maze.py tests/test_gpu_maze_agent.py tests/test_gpu_fullsize.py -q -x 2>&1 | grep -E "passed|failed|Error" | head -5
timeout 600 python scripts/devtools/probe_maze_move.py 2>&1 | grep -E "envs|Error"
timeout 600 python scripts/bench_families.py --families maze64 2>/dev/null | cut -c1-420
