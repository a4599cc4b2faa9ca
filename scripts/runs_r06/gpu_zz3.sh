#!/bin/bash
# round 6, visit zz3: prefetch distance 2 (three windows in flight, 241 registers) against distance 1 (206) and the tree
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzpfd2.so timeout 900 python -m pytest tests/test_gpu_maze.py -x -q --timeout 600 > $O/zz3_pytest.log 2>&1; echo "rc=$?"; tail -2 $O/zz3_pytest.log
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
  run base maze64
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzpf2.so run prefetch_1_ahead maze64
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzpfd2.so run prefetch_2_ahead maze64
done | tee $O/zz3_maze_prefetch_ab.txt
