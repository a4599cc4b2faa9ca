# A/B: fence search (S <= 112) through the generic line reader (in-tree) vs the 8-lane code (fold)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_anymdp.py tests/test_gpu_mixed.py tests/test_gpu_capture.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/r04_x_pytest.log 2>&1; echo "pytest rc=$? $(grep -h 'passed\|failed' gpurun_out/r04_x_pytest.log | tail -1)"
for rep in 1 2; do
for v in intree fold; do
  if [ $v = intree ]; then unset XV_LIB_PATH; else export XV_LIB_PATH=scripts/devtools/_build/libxeno_$v.so; fi
  timeout 600 python bench.py --no-variants --no-families --no-cpu-baseline --sustain-seconds 0 --search fence 2>/dev/null | python -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); r = d['roofline']
print('$v', '2a fence: value %.4e kernel us %.3f search %s' % (d['value'], r['avg_launch_us'], d['config']['search']))"
  timeout 600 python bench.py --no-variants --no-families --no-cpu-baseline --sustain-seconds 0 --search fence --tasks 1024 2>/dev/null | python -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); r = d['roofline']
print('$v', '2b fence: value %.4e kernel us %.3f search %s' % (d['value'], r['avg_launch_us'], d['config']['search']))"
  timeout 600 python scripts/bench_families.py --families anymdp_refdist 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', 'refdist', {k: round(x, 3) for k, x in d['us_per_step'].items()})"
done
done
