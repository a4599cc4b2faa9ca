#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" gpurun_out/r_pytest_gpu.log | head -3
timeout 600 python examples/quickstart.py > gpurun_out/r_quickstart.log 2>&1; echo "quickstart rc=$?"; tail -6 gpurun_out/r_quickstart.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
