# round 4, trip i: A/B of the four-lane (32 B per lane) bucket search against the eight-lane one
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/scripts/devtools/_build/libxeno_oct.so
timeout 900 python -m pytest tests/test_gpu_anymdp.py tests/test_gpu_mixed.py tests/test_gpu_capture.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2 3; do
  for L in quad oct; do
    if [ $L = oct ]; then export XV_LIB_PATH=$V; else unset XV_LIB_PATH; fi
    timeout 600 python bench.py --steps 2000 --warmup 200 --repeats 15 --no-cpu-baseline --no-families --no-variants --sustain-seconds 0 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('$L', d['config']['search'], round(d['roofline']['avg_launch_us'],3), 'us', '%.4g'%d['value'])"
  done
done | tee gpurun_out/r04_i_ab_quad.txt
for L in quad oct; do
  if [ $L = oct ]; then export XV_LIB_PATH=$V; else unset XV_LIB_PATH; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families --no-variants --sustain-seconds 0 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('$L steps20', d['config']['search'], round(d['roofline']['avg_launch_us'],3), 'us', '%.4g'%d['value'])"
  timeout 300 python scripts/bench_families.py --families anymdp_refdist,mixed,python_loop --steps 400 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('$L', d['family'], d.get('us_per_step', d.get('us_per_vector_step')))"
done | tee -a gpurun_out/r04_i_ab_quad.txt
