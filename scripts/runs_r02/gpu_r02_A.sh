#!/bin/bash
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_anymdp.py -q -m gpu -x -k "s64_wave or golden_64x8 or bucket" > gpurun_out/pytest_A.log 2>&1; grep -E "passed|failed|Error" gpurun_out/pytest_A.log | tail -5
for s in fence bucket; do
timeout 900 python bench.py --search $s --steps 1000 --warmup 100 --repeats 9 --no-cpu-baseline 2> gpurun_out/bench_A_$s.err | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['search'], d['config'].get('bucket_lines_gib_per_gpu'), 'us/step', d['ms_per_step']*1e3, 'kernel', d['roofline']['avg_launch_us'], 'value', d['value'], 'frac', d['roofline']['frac'], 'errs', d['config']['device_error_flags'])"
tail -2 gpurun_out/bench_A_$s.err
done
