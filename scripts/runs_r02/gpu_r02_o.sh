#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cartpole.py tests/test_gpu_acrobot.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python scripts/bench_families.py --families cartpole,acrobot --steps 400 --warmup 40 > gpurun_out/fam_o.jsonl 2> gpurun_out/fam_o.err
cat gpurun_out/fam_o.jsonl; tail -3 gpurun_out/fam_o.err
