#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_anymdp.py tests/test_gpu_mixed.py tests/test_gpu_fullsize.py -q -x 2>&1 | grep -E "passed|failed|Error" | head -5
timeout 600 python scripts/devtools/probe_wall.py 2>&1 | grep -E "graph" | cut -c1-300
for i in 1 2; do
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-families 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('steps20', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['timing'])"
done
timeout 600 python bench.py --gpus 1 --no-cpu-baseline --no-families 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('default', d['steps'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
