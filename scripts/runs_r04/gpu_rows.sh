# A/B: ray caster pixel phase with lanes = rows of one column (XV_MAZE_ROWS=1) against lanes = columns
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
XV_MAZE_ROWS=1 timeout 1500 python -m pytest tests/test_gpu_maze.py tests/test_gpu_fullsize.py -m gpu -q -x -k "maze or config_4" > gpurun_out/r04_rows_pytest.log 2>&1; echo "pytest rc=$? $(grep -h 'passed\|failed' gpurun_out/r04_rows_pytest.log | tail -1)"; grep -n "^FAILED\|^E   " gpurun_out/r04_rows_pytest.log | head -8
for rep in 1 2; do
for v in 0 1 2; do
  for fam in maze64 maze256; do
    XV_MAZE_ROWS=$v timeout 600 python scripts/bench_families.py --families $fam 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rows=$v', '$fam', {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
  done
done
done
