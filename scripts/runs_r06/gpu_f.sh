#!/bin/bash
# round 6, visit f: the whole GPU suite on the tree so far; ray-caster LDS chunk A/B; the Python loops (slab recycling);
# eight ranks sharing the one GPU (functional); floor probe
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 > $O/f_pytest_gpu.log 2>&1; echo "rc=$?"; tail -6 $O/f_pytest_gpu.log
echo "== maze HC A/B (64 x 64, rows per LDS chunk)"
for hc in 64 32 16; do
  XV_MAZE_HC=$hc timeout 300 python scripts/bench_families.py --families maze64,maze64_f32 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('HC $hc', d['filter'], {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
done | tee $O/f_maze_hc_ab.txt
echo "== python loops"
timeout 600 python scripts/bench_families.py --families python_loop --steps 2000 > $O/f_python_loop.jsonl 2> $O/f_python_loop.err; echo "rc=$?"; cat $O/f_python_loop.jsonl | cut -c1-2500; tail -3 $O/f_python_loop.err
echo "== floor probe"
timeout 300 python scripts/devtools/floor_probe.py $O/f_floor_probe.json | cut -c1-600
echo "== eight ranks sharing the one GPU (functional, not a measurement)"
export XV_BENCH_SHARE_GPU=1 XV_BENCH_FORCE_GATHER=1 MASTER_PORT=29517
timeout 900 python bench.py --gpus 8 --steps 128 --warmup 32 --repeats 3 --tasks 256 --envs 16384 --no-cpu-baseline --sustain-seconds 1 --long-steps 256 --long-repeats 2 > $O/f_bench_n8_shared.json 2> $O/f_bench_n8_shared.err; echo "rc=$?"; tail -5 $O/f_bench_n8_shared.err
grep '^{"metric"' $O/f_bench_n8_shared.json | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('n_gpus', 'value', 'rccl', 'rccl_ranks', 'transport', 'transport_note')}); print(d['config']['exchange'][:300]); print('with_allgather', d.get('with_allgather'))
print('long_call', {m: (d['long_call'][m].get('us_per_step'), d['long_call'][m].get('overlap_state')) for m in ('one_stream', 'overlapped', 'fused_rollout')} if d.get('long_call') else None)
print('families.mixed', json.dumps(d.get('families', {}).get('mixed'))[:900])
"
MASTER_PORT=29519 timeout 900 python bench.py --workload mixed --gpus 8 --steps 128 --warmup 32 --repeats 3 --no-cpu-baseline > $O/f_bench_mixed_n8_shared.json 2> $O/f_bench_mixed_n8_shared.err; echo "rc=$?"; tail -3 $O/f_bench_mixed_n8_shared.err
grep '^{"metric"' $O/f_bench_mixed_n8_shared.json | tail -1 | cut -c1-1500
