#!/bin/bash
# round 6, visit d: the whole chains / mixed-shard files in ONE process (visit a stopped moving in the matmul neighbour test after 54 tests)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_chains.py tests/test_gpu_mixed_shard.py -x -q -s --timeout 240 -o faulthandler_timeout=200 > gpurun_out/d_pytest_chains.log 2>&1; echo "rc=$?"; tail -60 gpurun_out/d_pytest_chains.log
