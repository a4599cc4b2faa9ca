#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python scripts/bench_families.py --steps 400 --warmup 40 --families maze64_m1,maze64_m3,maze64_m9,cartpole > gpurun_out/fam_h.jsonl 2> gpurun_out/fam_h.err; echo rc=$?; cut -c1-360 gpurun_out/fam_h.jsonl; tail -5 gpurun_out/fam_h.err
