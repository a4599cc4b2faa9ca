#!/bin/bash
# round 6, visit w: the pair copy in both phases (four 16-byte loads and no parity selects per window) against the one-phase
# copy with its third span fetched whole (six loads): parity, A/B at 64 x 64 and 256 x 256
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_maze.py tests/test_gpu_maze_agent.py tests/test_gpu_fullsize.py -x -q --timeout 600 > $O/w_pytest.log 2>&1; echo "rc=$?"; tail -3 $O/w_pytest.log
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
  run two_phases_4_loads maze64
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzonephase.so run one_phase_6_loads maze64
done | tee $O/w_maze_fetch_ab.txt
unset XV_LIB_PATH
run two_phases maze256 | tee -a $O/w_maze_fetch_ab.txt
PYTHONPATH=.:tests timeout 300 python tests/soak_maze.py 200 > $O/w_soak_maze.txt 2>&1; echo "soak rc=$?"; tail -2 $O/w_soak_maze.txt
