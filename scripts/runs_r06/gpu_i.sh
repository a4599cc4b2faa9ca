#!/bin/bash
# round 6, visit i: mixed copy=True on slabs (tests, python loops); PMC traffic incl. the fused roll-out kernel
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_mixed.py tests/test_gpu_linds.py tests/test_gpu_anymdp.py -x -q --timeout 600 > $O/i_pytest.log 2>&1; echo "rc=$?"; tail -4 $O/i_pytest.log
timeout 600 python scripts/bench_families.py --families python_loop --steps 2000 > $O/i_python_loop.jsonl 2> $O/i_python_loop.err; echo "rc=$?"; cat $O/i_python_loop.jsonl | cut -c1-3000
bash scripts/gpu_pmc.sh > $O/i_pmc.log 2>&1; tail -6 $O/i_pmc.log
