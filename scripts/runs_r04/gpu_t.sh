# A/B: the speculated exact texture filter (in-tree: 3 waves per SIMD, pair-interleaved copy) against the direct one and its
# own variants (2 waves per SIMD; row-major copy)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_maze.py tests/test_gpu_maze_agent.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/r04_t_pytest.log 2>&1; echo "pytest rc=$? $(grep -h 'passed\|failed' gpurun_out/r04_t_pytest.log | tail -1)"; grep -n "^FAILED\|^E  " gpurun_out/r04_t_pytest.log | head
for rep in 1 2; do
for v in intree; do
  if [ $v = intree ]; then unset XV_LIB_PATH; else export XV_LIB_PATH=scripts/devtools/_build/libxeno_$v.so; fi
  for fam in maze64 maze64_direct maze64_f32 maze256 maze256_direct; do
    if [ $v != intree ] && [ $fam != maze64 ] && [ $fam != maze256 ]; then continue; fi
    timeout 600 python scripts/bench_families.py --families $fam 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', '$fam', {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
  done
done
done
