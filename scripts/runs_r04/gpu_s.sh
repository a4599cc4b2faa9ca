# A/B: restart-state observation lines read by the owner lane alone (in-tree) vs by lane pairs (tbase)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_anymdp_tok.py tests/test_gpu_mixed.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/r04_s_pytest.log 2>&1; echo "pytest rc=$? $(grep -h 'passed\|failed' gpurun_out/r04_s_pytest.log | tail -1)"
for rep in 1 2; do
for v in intree tbase; do
  if [ $v = intree ]; then unset XV_LIB_PATH; else export XV_LIB_PATH=scripts/devtools/_build/libxeno_$v.so; fi
  timeout 600 python scripts/bench_families.py --families anymdp_tok,anymdp_tok_refdist 2>/dev/null > gpurun_out/r04_s_tok_${v}_$rep.json
  python - gpurun_out/r04_s_tok_${v}_$rep.json $v <<PY
import json, sys
for l in open(sys.argv[1]):
    if not l.startswith("{"): continue
    d = json.loads(l)
    if "us_per_step" in d: print(sys.argv[2], "  synthetic (2,2):", d["us_per_step"])
    else: print(sys.argv[2], "  refdist:", {k: round(v["auto"]["us_per_step"], 2) for k, v in d["variants"].items()})
PY
done
done
