#!/bin/bash
# round 3, visit H: cooperative multi-token POMDP step
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest tok"; timeout 900 python -m pytest tests/test_gpu_anymdp_tok.py tests/test_gpu_anymdp.py -x -q > gpurun_out/h_pytest.log 2>&1; echo "rc=$?"; grep -n "passed\|failed\|Error\|assert" gpurun_out/h_pytest.log | head
echo "== tok bench"; timeout 300 python scripts/bench_families.py --families anymdp_tok 2>/dev/null | cut -c1-500
