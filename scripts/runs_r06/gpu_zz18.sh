#!/bin/bash
# round 6, visit zz18: columns per sub-pass of the rows mapping's frame chunk: 112 / 96 / 64 (LDS per workgroup 51.1 / 48.0 / 41.7 KB)
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
for v in mzn112 mzn96 mzn64 mzwhole; do
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_$v.so run $v maze256
done | tee $O/zz18_maze256_nsub.txt
