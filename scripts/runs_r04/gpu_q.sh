# A/B: lanes per env of the cooperative token step (in-tree = 4 contiguous; variants 4 interleaved, 2, 1, 8)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in intree q4i d2c d2i u1 o8; do
  if [ $v = intree ]; then unset XV_LIB_PATH; else export XV_LIB_PATH=scripts/devtools/_build/libxeno_$v.so; fi
  timeout 900 python -m pytest tests/test_gpu_anymdp_tok.py -m gpu -q -x > gpurun_out/r04_q_pytest_$v.log 2>&1; echo "== $v pytest rc=$? $(grep -h 'passed\|failed' gpurun_out/r04_q_pytest_$v.log | tail -1)"
  for rep in 1 2; do
    timeout 600 python scripts/bench_families.py --families anymdp_tok,anymdp_tok_refdist 2>/dev/null > gpurun_out/r04_q_tok_${v}_$rep.json
    python - gpurun_out/r04_q_tok_${v}_$rep.json <<PY
import json, sys
for l in open(sys.argv[1]):
    if not l.startswith("{"): continue
    d = json.loads(l)
    if "us_per_step" in d: print("   synthetic (2,2):", d["us_per_step"])
    else: print("   refdist:", {k: round(v["auto"]["us_per_step"], 2) for k, v in d["variants"].items()})
PY
  done
done
