#!/bin/bash
# round 6, visit zz26: rows mapping at 64 x 64 with the frame chunk for half the columns (12.8 instead of 19 KB per one-wave
# workgroup: three waves per SIMD instead of two)
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
  XV_MAZE_FILT=5 run rows maze64
  XV_MAZE_FILT=5 XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzhalfall.so run rows_half_chunk maze64
done | tee $O/zz26_maze64_rows_half.txt
