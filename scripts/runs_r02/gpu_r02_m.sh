#!/bin/bash
# round 2, call m: maze teachers on the device
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_maze_agent.py -x -q -m gpu > gpurun_out/pytest_m.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_m.log
tail -30 gpurun_out/pytest_m.log
