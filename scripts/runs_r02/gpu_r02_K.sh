#!/bin/bash
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_anymdp.py -q -m gpu -k "s64_wave or golden_64x8 or bucket or fused_rollout or adversarial" 2>&1 | grep -E "passed|failed"
for i in 1 2; do
timeout 900 python bench.py --steps 2000 --warmup 100 --repeats 9 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('value %.4g' % d['value'], 'us/step %.3f' % (d['ms_per_step']*1e3), 'kernel %.3f' % d['roofline']['avg_launch_us'])"
done
