#!/bin/bash
# round 6, visit c: the neighbour TESTS under pytest's faulthandler (visit a's run stopped moving in the matmul case; the probe of visit b did not)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== neighbour tests alone"
timeout 400 python -m pytest tests/test_gpu_chains.py -x -q -s -k "neighbour" -o faulthandler_timeout=150 > gpurun_out/c_pytest_neighbour.log 2>&1; echo "rc=$?"; tail -60 gpurun_out/c_pytest_neighbour.log
