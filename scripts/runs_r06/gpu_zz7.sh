#!/bin/bash
# round 6, visit zz7: what the 256 x 256 ray cast (rows mapping) would run at with three waves per SIMD instead of the two its 80 KB
# of LDS per workgroup leave (timing-only build, XV_MAZE_TIMING_ALIAS as in visit t: wrong frames, same instructions)
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
  unset XV_LIB_PATH
  run base maze256
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzalias.so run alias maze256
done | tee $O/zz7_maze256_alias.txt
