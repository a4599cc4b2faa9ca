#!/bin/bash
# round 6, visit b: what stopped moving in the neighbour test of visit a; the remaining mixed tests
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for k in none matmul allgather; do
  echo "== neighbour $k"
  XV_PIPE_DEBUG=1 PROBE_DUMP_S=100 timeout 150 python scripts/devtools/probe_neighbour.py $k 8 > gpurun_out/b_neighbour_$k.log 2>&1; echo "rc=$?"
  tail -40 gpurun_out/b_neighbour_$k.log
done
echo "== mixed shard tests"
timeout 600 python -m pytest tests/test_gpu_mixed_shard.py -x -q > gpurun_out/b_pytest_mixed.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/b_pytest_mixed.log
