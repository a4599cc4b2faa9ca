#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_anymdp.py -q -m gpu > gpurun_out/pytest_D.log 2>&1; grep -E "passed|failed|Error" gpurun_out/pytest_D.log | tail -5
timeout 900 python bench.py --sweep-envs 65536 --sweep-out gpurun_out/sweep_D.json --steps 1000 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-700
