#!/bin/bash
# AnyMDP headline: search modes x task sharing on one box (same device, back to back).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for mode in fence binary; do
  for tasks in 0 1024; do
    echo "== search=$mode tasks=$tasks"
    timeout 600 python bench.py --steps 2000 --warmup 200 --search $mode --tasks $tasks --fused --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.3e  us/step %.2f  frac %.3f  fused %.3e' % (d['value'], r['avg_launch_us'], r['frac'], d.get('fused_rollout_env_steps_per_s_rank0',0)))"
  done
done
