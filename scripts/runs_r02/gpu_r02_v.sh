#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_linds.py tests/test_gpu_mixed.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python scripts/bench_families.py --families linds --steps 400 --warmup 50 2>/dev/null | cut -c1-330
