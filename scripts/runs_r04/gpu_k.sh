# round 4, trip k: cooperative token kernel with the uniforms sent behind the line requests — A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/scripts/devtools/_build/libxeno_nopre.so
timeout 900 python -m pytest tests/test_gpu_anymdp_tok.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2; do
  for L in pre nopre; do
    if [ $L = nopre ]; then export XV_LIB_PATH=$V; else unset XV_LIB_PATH; fi
    timeout 600 python scripts/bench_families.py --families anymdp_tok,anymdp_tok_refdist --steps 800 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l)
    if 'variants' in d: print('$L', {k: {a: round(b['us_per_step'],2) for a,b in v.items() if isinstance(b, dict) and 'us_per_step' in b} for k,v in d['variants'].items()})
    else: print('$L', d['us_per_step'])"
  done
done | tee gpurun_out/r04_k_ab_coop_pre.txt
