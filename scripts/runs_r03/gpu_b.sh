#!/bin/bash
# round 3, visit B: counters of the LinDS MFMA step kernel
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
bash scripts/pmc_kernel.sh linds_r03a linds_step_mfma scripts/bench_families.py --families linds_mfma --steps 300 --warmup 30
cat gpurun_out/pmc_linds_r03a.json
