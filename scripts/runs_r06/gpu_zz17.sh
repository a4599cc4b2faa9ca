#!/bin/bash
# round 6, visit zz17: rows mapping at 256 columns with the frame chunk holding HALF the columns at a time (54,272 B of LDS per
# workgroup: three per CU) against the whole pass (-DXV_MAZE_ROWS_HALF=0: 79,360 B, two per CU): parity, A/B, soak
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_maze.py tests/test_gpu_maze_agent.py tests/test_gpu_fullsize.py -x -q --timeout 600 > $O/zz17_pytest.log 2>&1; echo "rc=$?"; tail -3 $O/zz17_pytest.log
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
  run half_columns maze256,maze64
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzwhole.so run whole_pass maze256
done | tee $O/zz17_maze256_half_ab.txt
PYTHONPATH=.:tests timeout 400 python tests/soak_maze.py 300 > $O/zz17_soak_maze.txt 2>&1; echo "soak rc=$?"; tail -2 $O/zz17_soak_maze.txt
