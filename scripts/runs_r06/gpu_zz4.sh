#!/bin/bash
# round 6, visit zz4: the prefetching pixel loops (columns and rows mappings, two waves per SIMD) as the tree's default against
# -DXV_MAZE_PREFETCH=0: parity of the tree, soak, A/B at 64 x 64 and 256 x 256
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_maze.py tests/test_gpu_maze_agent.py tests/test_gpu_fullsize.py -x -q --timeout 600 > $O/zz4_pytest.log 2>&1; echo "rc=$?"; tail -3 $O/zz4_pytest.log
PYTHONPATH=.:tests timeout 400 python tests/soak_maze.py 300 > $O/zz4_soak_maze.txt 2>&1; echo "soak rc=$?"; tail -2 $O/zz4_soak_maze.txt
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
  run prefetch maze64,maze256
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mznopf.so run no_prefetch maze64,maze256
done | tee $O/zz4_maze_prefetch_ab.txt
