#!/bin/bash
# round 6, visit h: LinDS copy=True on slabs, MixedBatch.step_fused persistent path (tests + python loops); PMC traffic of the
# AnyMDP step kernels on the current source (2a)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
echo "== tests"
timeout 1200 python -m pytest tests/test_gpu_mixed.py tests/test_gpu_linds.py tests/test_gpu_capture.py tests/test_gpu_cartpole.py tests/test_gpu_fullsize.py -x -q --timeout 600 > $O/h_pytest.log 2>&1; echo "rc=$?"; tail -6 $O/h_pytest.log
echo "== python loops"
timeout 600 python scripts/bench_families.py --families python_loop --steps 2000 > $O/h_python_loop.jsonl 2> $O/h_python_loop.err; echo "rc=$?"; cat $O/h_python_loop.jsonl | cut -c1-3000; tail -3 $O/h_python_loop.err
echo "== PMC traffic 2a"
bash scripts/gpu_pmc.sh > $O/h_pmc.log 2>&1; tail -8 $O/h_pmc.log
