#!/bin/bash
# round 6, visit p: the CPython trampoline in the eager step paths — parity tests, Python loops with and without it
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_anymdp.py tests/test_gpu_linds.py tests/test_gpu_capture.py tests/test_gpu_mixed.py -x -q --timeout 600 > $O/p_pytest.log 2>&1; echo "rc=$?"; grep -n "passed\|failed" $O/p_pytest.log | tail -1
for v in fast ctypes fast ctypes; do
  if [ $v = ctypes ]; then export XV_NO_FAST=1; else unset XV_NO_FAST; fi
  timeout 600 python scripts/bench_families.py --families python_loop --steps 3000 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', {k: round(x, 2) for k, x in d['us_per_vector_step'].items()}, 'linds', {k: round(x, 2) for k, x in d['other_families']['linds']['us_per_vector_step'].items()})
"
done | tee $O/p_python_loop_ab.txt
