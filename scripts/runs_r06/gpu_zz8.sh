#!/bin/bash
# round 6, visit zz8: rows mapping with the bytes stored straight to the frame (no frame chunk in LDS: 30 instead of 80 KB per
# workgroup at 256 x 256, three waves per SIMD instead of two) against -DXV_MAZE_ROWS_DIRECT=0: parity, soak, A/B
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_maze.py tests/test_gpu_maze_agent.py tests/test_gpu_fullsize.py -x -q --timeout 600 > $O/zz8_pytest.log 2>&1; echo "rc=$?"; tail -3 $O/zz8_pytest.log
PYTHONPATH=.:tests timeout 400 python tests/soak_maze.py 300 > $O/zz8_soak_maze.txt 2>&1; echo "soak rc=$?"; tail -2 $O/zz8_soak_maze.txt
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
  run direct maze256
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mznodirect.so run lds_chunk maze256
done | tee $O/zz8_maze256_direct_ab.txt
