#!/bin/bash
# round 5, call B: overlapped step_many (two streams + per-wave hand-off words): parity, then the A/B at 2a / 2b
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_b
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_chains.py -x -q -k "overlapped and small" > $O/pytest_overlap_small.txt 2>&1
echo "pytest overlap small rc=$?"; tail -5 $O/pytest_overlap_small.txt
timeout 900 python scripts/devtools/probe_chains.py --tag 2a --ks 1 --overlap > $O/overlap_2a.jsonl 2> $O/overlap_2a.err
echo "2a rc=$?"; cut -c1-500 $O/overlap_2a.jsonl; tail -3 $O/overlap_2a.err
timeout 600 python scripts/devtools/probe_chains.py --tag 2b --tasks 1024 --ks 1 --overlap > $O/overlap_2b.jsonl 2> $O/overlap_2b.err
echo "2b rc=$?"; cut -c1-500 $O/overlap_2b.jsonl
timeout 1500 python -m pytest tests/test_gpu_chains.py -x -q > $O/pytest_chains.txt 2>&1
echo "pytest chains rc=$?"; tail -8 $O/pytest_chains.txt
