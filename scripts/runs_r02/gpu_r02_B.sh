#!/bin/bash
for b in 16 32; do
timeout 900 python bench.py --search bucket --buckets $b --steps 1000 --warmup 100 --repeats 9 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('buckets', $b, d['config'].get('bucket_lines_gib_per_gpu'), 'kernel us', d['roofline']['avg_launch_us'], 'value', d['value'])"
done
timeout 1800 python -m pytest tests/test_gpu_anymdp.py -q -m gpu -x -k "s64_wave or golden_64x8 or bucket" 2>&1 | grep -E "passed|failed"
