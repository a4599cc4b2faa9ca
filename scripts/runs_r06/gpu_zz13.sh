#!/bin/bash
# round 6, visit zz13 (run twice: six waves with a smaller redo list; then TWELVE waves, one workgroup per CU):
# 79,872 B that fit a CU twice
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
  run six_waves maze256,maze64
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzfour.so run four_waves maze256
done | tee $O/zz13_maze256_six_waves_ab.txt
