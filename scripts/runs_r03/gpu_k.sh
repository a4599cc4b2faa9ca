#!/bin/bash
# round 3, visit K: device observation-model sampler
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest sampler"; timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_anymdp_tok.py -x -q > gpurun_out/k_pytest.log 2>&1; echo "rc=$?"; grep -n "passed\|failed\|Error\|assert" gpurun_out/k_pytest.log | head
