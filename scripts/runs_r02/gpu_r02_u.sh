#!/bin/bash
mkdir -p gpurun_out
for st in 0 1 2 3 4 6 8 12; do
  echo "stagger=$st"
  XV_ANYMDP_STAGGER=$st timeout 600 python bench.py --steps 1000 --warmup 100 --repeats 7 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step']*1e3, d['roofline']['avg_launch_us'], d['value'])"
done
