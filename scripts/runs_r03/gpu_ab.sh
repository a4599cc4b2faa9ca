#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_anymdp_tok.py tests/test_gpu_fullsize.py tests/test_gpu_sampler.py -q -x 2>&1 | grep -E "passed|failed|Error|error" | head -5
timeout 600 python scripts/devtools/probe_real_tasks.py 2>&1 | tail -4
