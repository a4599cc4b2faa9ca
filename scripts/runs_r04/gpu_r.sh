# A/B: lanes per env of the bucket line read in the MDP step kernel (in-tree = 2 contiguous; variants 8, 4, 1, 2 interleaved)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_anymdp.py tests/test_gpu_anymdp_tok.py tests/test_gpu_mixed.py tests/test_gpu_capture.py -m gpu -q -x > gpurun_out/r04_r_pytest.log 2>&1; echo "pytest rc=$? $(grep -h 'passed\|failed' gpurun_out/r04_r_pytest.log | tail -1)"
for rep in 1 2; do
for v in intree s8 s4 s1 s2i; do
  if [ $v = intree ]; then unset XV_LIB_PATH; else export XV_LIB_PATH=scripts/devtools/_build/libxeno_$v.so; fi
  timeout 600 python bench.py --no-variants --no-families --sustain-seconds 0 2>/dev/null > gpurun_out/r04_r_bench_${v}_$rep.json
  python - gpurun_out/r04_r_bench_${v}_$rep.json $v <<PY
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
r = d["roofline"]
print(sys.argv[2], "value %.4e kernel us %.3f search %s" % (d["value"], r["avg_launch_us"], d["config"]["search"]))
PY
done
done
