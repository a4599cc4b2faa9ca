#!/bin/bash
# round 6, visit g: token-step replay, python loops of LinDS and the mixed batch, soaks (alone and beside a second process),
# eight ranks with the gather over gloo on device records, the quickstart
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
echo "== chain tests (token replay, alternating env / sub-batches)"
timeout 900 python -m pytest tests/test_gpu_chains.py tests/test_gpu_mixed_shard.py tests/test_gpu_bench_line.py -x -q -s --timeout 300 > $O/g_pytest_chains.log 2>&1; echo "rc=$?"; tail -8 $O/g_pytest_chains.log
echo "== python loops"
timeout 600 python scripts/bench_families.py --families python_loop --steps 2000 > $O/g_python_loop.jsonl 2> $O/g_python_loop.err; echo "rc=$?"; cat $O/g_python_loop.jsonl | cut -c1-3000; tail -3 $O/g_python_loop.err
echo "== quickstart"
timeout 300 python examples/quickstart.py > $O/g_quickstart.txt 2>&1; echo "rc=$?"; tail -12 $O/g_quickstart.txt
echo "== soaks"
PYTHONPATH=.:tests timeout 300 python tests/soak_anymdp.py 200 > $O/g_soak_anymdp.txt 2>&1; echo "soak anymdp rc=$?"; tail -2 $O/g_soak_anymdp.txt | cut -c1-300
PYTHONPATH=.:tests timeout 300 python tests/soak_mixed.py 200 > $O/g_soak_mixed.txt 2>&1; echo "soak mixed rc=$?"; tail -2 $O/g_soak_mixed.txt | cut -c1-300
PYTHONPATH=.:tests timeout 200 python tests/soak_linds.py 100 > $O/g_soak_linds.txt 2>&1; echo "soak linds rc=$?"; tail -1 $O/g_soak_linds.txt | cut -c1-300
echo "== the mixed soak beside a second process that keeps the GPU busy"
python scripts/devtools/gpu_hog.py 150 > $O/g_hog.txt 2>&1 &
HOG=$!
sleep 5
PYTHONPATH=.:tests timeout 200 python tests/soak_mixed.py 120 > $O/g_soak_mixed_beside_hog.txt 2>&1; echo "soak mixed beside hog rc=$?"; tail -2 $O/g_soak_mixed_beside_hog.txt | cut -c1-300
kill $HOG 2>/dev/null; wait $HOG 2>/dev/null
echo "== eight ranks on one GPU, gloo backend, gather forced (device records through the pack kernels; functional)"
export XV_BENCH_SHARE_GPU=1 XV_BENCH_FORCE_GATHER=1 XV_BENCH_BACKEND=gloo MASTER_PORT=29521
timeout 900 python bench.py --gpus 8 --steps 128 --warmup 32 --repeats 3 --tasks 256 --envs 16384 --no-cpu-baseline --sustain-seconds 0 --long-steps 0 --transport torch > $O/g_bench_n8_gloo.json 2> $O/g_bench_n8_gloo.err; echo "rc=$?"; tail -2 $O/g_bench_n8_gloo.err | cut -c1-300
grep '^{"metric"' $O/g_bench_n8_gloo.json | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('n_gpus', 'value', 'rccl', 'transport', 'transport_note')}); print(d['config']['exchange'][:300]); print('with_allgather', d.get('with_allgather'))
print('families.mixed', json.dumps({k: (d.get('families') or {}).get('mixed', {}).get(k) for k in ('value', 'with_allgather', 'transport', 'error')})[:700])
"
