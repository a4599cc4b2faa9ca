#!/bin/bash
# round 6, visit zz2: ray caster (columns mapping), the next pixel's window requested before the current pixel is filtered
# (-DXV_MAZE_PREFETCH=1) at 2 and 3 waves per SIMD, against the tree: parity of the variant, A/B
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzpf2.so timeout 900 python -m pytest tests/test_gpu_maze.py -x -q --timeout 600 > $O/zz2_pytest.log 2>&1; echo "rc=$?"; tail -2 $O/zz2_pytest.log
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
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzpf2.so run prefetch_2_waves maze64
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzpf3.so run prefetch_3_waves maze64
done | tee $O/zz2_maze_prefetch_ab.txt
