#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_acrobot.py tests/test_gpu_mixed.py -q -x 2>&1 | grep -E "passed|failed|Error" | head -5
timeout 600 python scripts/bench_families.py --families acrobot 2>/dev/null | cut -c1-420
