#!/bin/bash
# round 6, visit zz19: the per-column table at 80 instead of 96 bytes (the wall's texel row as an int): with it a HALF-pass frame
# chunk makes the rows workgroup 50,176 B (three to a CU); parity, then 128 / 112 columns per sub-pass against the whole pass
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_maze.py tests/test_gpu_maze_agent.py tests/test_gpu_fullsize.py -x -q --timeout 600 > $O/zz19_pytest.log 2>&1; echo "rc=$?"; tail -3 $O/zz19_pytest.log
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
  run half_128 maze256
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzn112.so run sub_112 maze256
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzwhole.so run whole_pass maze256
done | tee $O/zz19_maze256_half_ab.txt
