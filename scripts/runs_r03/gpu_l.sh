#!/bin/bash
# round 3, visit L: maze move kernel without the per-sub-step square root; whole GPU suite
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/l_pytest_gpu.log 2>&1; echo "rc=$?"; grep -n "passed\|failed\|Error" gpurun_out/l_pytest_gpu.log | head -5
echo "== maze"; timeout 600 python scripts/bench_families.py --families maze64,maze64_m1,maze64_m3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['move_kernel'], d['us_per_step'])"
