#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== base"
timeout 300 python scripts/bench_families.py --families linds_mfma --steps 400 --warmup 40 2>/dev/null | cut -c1-330
echo "== accv"
export XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_accv.so
timeout 900 python -m pytest tests/test_gpu_linds.py tests/test_gpu_mixed.py tests/test_gpu_fullsize.py -q -x 2>&1 | grep -E "passed|failed|Error" | head -3
timeout 300 python scripts/bench_families.py --families linds_mfma --steps 400 --warmup 40 2>/dev/null | cut -c1-330
unset XV_LIB_PATH
echo "== base again"
timeout 300 python scripts/bench_families.py --families linds_mfma --steps 400 --warmup 40 2>/dev/null | cut -c1-330
