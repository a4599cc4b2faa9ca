#!/bin/bash
# round 5, call L: set_task probe, python loop with slabs, tables tests, rest of the GPU suite from where K stopped
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_l
mkdir -p $O
timeout 600 python scripts/devtools/probe_set_task.py --tasks 1024 > $O/set_task_probe.json 2> $O/set_task_probe.err; echo "probe rc=$?"; cat $O/set_task_probe.json
timeout 600 python scripts/bench_families.py --families python_loop > $O/python_loop.jsonl 2> $O/python_loop.err; echo "python_loop rc=$?"; cut -c1-900 $O/python_loop.jsonl
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_all.txt 2>&1; echo "pytest all rc=$?"; tail -4 $O/pytest_gpu_all.txt | cut -c1-300
