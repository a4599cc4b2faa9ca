#!/bin/bash
# round 6, visit o: rows per LDS chunk of the 64 x 64 ray caster again, now that re-run pixels are spread over the wave
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
for hc in 64 48 32 16; do
  XV_MAZE_HC=$hc timeout 300 python scripts/bench_families.py --families maze64,maze128 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('HC $hc', d['workload'][-16:], {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
done
done | tee gpurun_out/o_maze_hc_ab.txt
