#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_anymdp.py tests/test_gpu_anymdp_tok.py tests/test_gpu_mixed.py tests/test_gpu_sampler.py -x -q 2>&1 | tail -1
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/p_bench.json; python -c "
import json
d=json.load(open('gpurun_out/p_bench.json')); print('value %.4e us/step %.3f frac %.3f traffic %s' % (d['value'], d['ms_per_step']*1e3, d['roofline']['frac'], d['roofline']['traffic']))
for k,v in d['families'].items(): print(k, v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'))"
timeout 300 python scripts/bench_families.py --families anymdp_tok 2>/dev/null | cut -c1-400
