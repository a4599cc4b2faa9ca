# A/B: speculated filter with float32 colour sums (in-tree) vs float64 sums (rcf64); soak of the in-tree one
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_maze.py tests/test_gpu_fullsize.py -m gpu -q -x -k "maze or config_4" > gpurun_out/r04_w_pytest.log 2>&1; echo "pytest rc=$? $(grep -h 'passed\|failed' gpurun_out/r04_w_pytest.log | tail -1)"
for rep in 1 2; do
for v in intree rcf64; do
  if [ $v = intree ]; then unset XV_LIB_PATH; else export XV_LIB_PATH=scripts/devtools/_build/libxeno_$v.so; fi
  for fam in maze64 maze256; do
    timeout 600 python scripts/bench_families.py --families $fam 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', '$fam', {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
  done
done
done
unset XV_LIB_PATH
PYTHONPATH=. timeout 900 python scripts/devtools/soak_spec_filter.py 180 > gpurun_out/r04_w_soak_spec_filter.txt 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r04_w_soak_spec_filter.txt
