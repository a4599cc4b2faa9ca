#!/bin/bash
# round 6, visit zz25: the final kernels at 64 x 64 by window fetch (XV_MAZE_FILT): pair copy + prefetch (default) / row-major copy +
# prefetch / rows mapping
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
run() {  # tag families
  timeout 600 python scripts/bench_families.py --families $2 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['workload'][-16:], d['filter'], {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
}
for rep in 1 2; do
  unset XV_MAZE_FILT
  run pairs_prefetch maze64
  XV_MAZE_FILT=3 run rowmajor_prefetch maze64
  XV_MAZE_FILT=5 run rows maze64
done | tee $O/zz25_maze64_fetch_final.txt
