#!/bin/bash
# round 6, visit zz9: (first run) the direct rows stores without the nontemporal hint; (second run, this text) each wave stages 16 column segments and writes them out itself
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
  run staged_per_wave maze256
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mznodirect.so run lds_chunk maze256
done | tee $O/zz9_maze256_direct_ab.txt
