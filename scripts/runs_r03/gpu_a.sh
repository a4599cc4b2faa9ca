#!/bin/bash
# round 3, visit A: the re-laid-out LinDS kernel — parity first, then its step time beside the round-2 figure.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest linds"; timeout 900 python -m pytest tests/test_gpu_linds.py -x -q > gpurun_out/a_pytest_linds.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/a_pytest_linds.log
echo "== pytest all gpu"; timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/a_pytest_gpu.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/a_pytest_gpu.log
echo "== families linds"; timeout 600 python scripts/bench_families.py --families linds,mixed > gpurun_out/a_families.jsonl 2> gpurun_out/a_families.err; echo "rc=$?"; cut -c1-700 gpurun_out/a_families.jsonl; tail -3 gpurun_out/a_families.err
