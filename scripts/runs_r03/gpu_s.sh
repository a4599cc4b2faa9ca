#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_linds.py tests/test_gpu_mixed.py -q -x 2>&1 | grep -E "passed|failed|Error" | head -3
timeout 300 python scripts/bench_families.py --families linds_mfma,linds_mfma --steps 400 --warmup 40 2>/dev/null | cut -c1-330
