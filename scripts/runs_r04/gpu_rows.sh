# ray caster with the AUTO mapping (rows for fp32 and for exact beyond 128 x 128) at the final tree: tests and times
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_maze.py tests/test_gpu_fullsize.py tests/test_gpu_maze_agent.py -m gpu -q -x > gpurun_out/r04_rows_pytest.log 2>&1; echo "pytest rc=$? $(grep -h 'passed\|failed' gpurun_out/r04_rows_pytest.log | tail -1)"; grep -n "^FAILED\|^E   " gpurun_out/r04_rows_pytest.log | head -8
for rep in 1 2; do
  for fam in maze64 maze64_f32 maze256 maze256_f32; do
    timeout 600 python scripts/bench_families.py --families $fam 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$fam', {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
  done
done
PYTHONPATH=.:tests timeout 900 python tests/soak_maze.py 150 31 > gpurun_out/r04_soak4_maze.txt 2>&1; echo "soak rc=$?"; tail -1 gpurun_out/r04_soak4_maze.txt
