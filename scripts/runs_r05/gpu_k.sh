#!/bin/bash
# round 5, call K: device-side set_task, output slabs (copy=True), mazeworld_256 family, the whole GPU suite
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_k
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tables.py -x -q -s > $O/pytest_tables.txt 2>&1; echo "pytest tables rc=$?"; grep -i "build of\|passed\|failed\|Error" $O/pytest_tables.txt | tail -5
timeout 900 python -m pytest tests/test_gpu_anymdp.py -x -q -k "copy or slab or steps_and_done" > $O/pytest_copy.txt 2>&1; echo "pytest copy rc=$?"; tail -3 $O/pytest_copy.txt
timeout 600 python scripts/bench_families.py --families python_loop > $O/python_loop.jsonl 2> $O/python_loop.err; echo "python_loop rc=$?"; cut -c1-900 $O/python_loop.jsonl
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_all.txt 2>&1; echo "pytest all rc=$?"; tail -5 $O/pytest_gpu_all.txt
