#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_anymdp.py tests/test_gpu_anymdp_tok.py tests/test_gpu_mixed.py tests/test_gpu_fullsize.py tests/test_gpu_sampler.py tests/test_gpu_teacher.py -q -x 2>&1 | grep -E "passed|failed|Error" | head -5
timeout 600 python scripts/devtools/probe_real_tasks.py 2>&1 | tail -4
timeout 600 python bench.py --no-cpu-baseline --no-families 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('bench 2a', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
timeout 600 python bench.py --no-cpu-baseline --no-families --tasks 1024 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('bench 2b', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
timeout 300 python scripts/bench_families.py --families anymdp_tok,mixed 2>/dev/null | cut -c1-330
