#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/fam_s.jsonl
for hc in 64 32 16; do
  for f in maze64 maze64_f32 maze256; do
    echo "HC=$hc $f" >> gpurun_out/fam_s.jsonl
    XV_MAZE_HC=$hc timeout 600 python scripts/bench_families.py --families $f --steps 400 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(json.dumps({'us': d['us_per_step'], 'filter': d['filter']}))" >> gpurun_out/fam_s.jsonl
  done
done
cat gpurun_out/fam_s.jsonl
