#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_anymdp.py -q -m gpu -k "graph_replay or s64_wave" 2>&1 | grep -E "passed|failed"
for st in 20 2000; do
timeout 900 python bench.py --gpus 1 --steps $st --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('steps $st', 'value %.4g' % d['value'], 'us/step %.3f' % (d['ms_per_step']*1e3), 'kernel %.3f' % d['roofline']['avg_launch_us'], d['config']['launch'])"
done
