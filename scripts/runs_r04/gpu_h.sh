# round 4, trip h: A/B of the LDS slot exchange in the bucket search (headline kernel)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/scripts/devtools/_build/libxeno_ldsx.so
XV_LIB_PATH=$V timeout 900 python -m pytest tests/test_gpu_anymdp.py -x -q -m gpu -k "bucket or search or census or six_cut or overflow or config1 or golden" 2>&1 | tail -4
for i in 1 2 3; do
  for L in base ldsx; do
    if [ $L = ldsx ]; then export XV_LIB_PATH=$V; else unset XV_LIB_PATH; fi
    timeout 600 python bench.py --steps 2000 --warmup 200 --repeats 15 --no-cpu-baseline --no-families --no-variants --sustain-seconds 0 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('$L', d['config']['search'], round(d['roofline']['avg_launch_us'],3), 'us', '%.4g'%d['value'])"
  done
done | tee gpurun_out/r04_h_ab_lds_xchg.txt
unset XV_LIB_PATH
for L in base ldsx; do
  if [ $L = ldsx ]; then export XV_LIB_PATH=$V; else unset XV_LIB_PATH; fi
  timeout 300 python scripts/bench_families.py --families anymdp_refdist,mixed --steps 400 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('$L', d['family'], d.get('us_per_step', d.get('us_per_vector_step')))"
done | tee -a gpurun_out/r04_h_ab_lds_xchg.txt
