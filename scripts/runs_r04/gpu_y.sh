# the whole GPU suite + smoke at the final tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04_z_pytest_gpu.log 2>&1; echo "rc=$?"; grep -n "passed\|failed" gpurun_out/r04_z_pytest_gpu.log | tee gpurun_out/r04_z_pytest_gpu_tail.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); r = d['roofline']
print('steps20: value %.4e ms/step %.5f kernel us %.3f search %s primary %s' % (d['value'], d['ms_per_step'], r['avg_launch_us'], d['config']['search'], r['primary']))
for k, v in (d.get('families') or {}).items(): print('   ', k, v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('error'))
print('cpu_baseline', d.get('cpu_baseline'))"
