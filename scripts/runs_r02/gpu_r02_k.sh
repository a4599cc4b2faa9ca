#!/bin/bash
# full GPU suite after the sampler / cartpole / maze changes
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python scripts/bench_families.py --steps 400 --warmup 40 --families linds,cartpole,maze64,maze64_f32,maze256,maze256_f32,mixed > gpurun_out/r02_c_bench_families.jsonl 2> gpurun_out/fam_k.err; cut -c1-300 gpurun_out/r02_c_bench_families.jsonl
