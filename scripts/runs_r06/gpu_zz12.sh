#!/bin/bash
# round 6, visit zz12: rows mapping at 256 columns and more with SIX waves per workgroup on the same 80 KB of LDS (two workgroups
# per CU either way: three waves per SIMD instead of two) against four (-DXV_MAZE_ROWS_SIX_WAVES=0): parity, A/B, soak
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_maze.py tests/test_gpu_maze_agent.py tests/test_gpu_fullsize.py -x -q --timeout 600 > $O/zz12_pytest.log 2>&1; echo "rc=$?"; tail -3 $O/zz12_pytest.log
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
  run six_waves maze256
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzfour.so run four_waves maze256
done | tee $O/zz12_maze256_six_waves_ab.txt
PYTHONPATH=.:tests timeout 400 python tests/soak_maze.py 300 > $O/zz12_soak_maze.txt 2>&1; echo "soak rc=$?"; tail -2 $O/zz12_soak_maze.txt
