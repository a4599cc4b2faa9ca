#!/bin/bash
# round 2, call l: LinDS k-order change + fused rollout: parity and timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_linds.py tests/test_gpu_fullsize.py tests/test_gpu_mixed.py -x -q -m gpu > gpurun_out/pytest_l.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_l.log
tail -5 gpurun_out/pytest_l.log
timeout 600 python scripts/bench_families.py --families linds --steps 400 --warmup 50 > gpurun_out/fam_l.jsonl 2> gpurun_out/fam_l.err
cat gpurun_out/fam_l.jsonl
