#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_J.json 2> gpurun_out/bench_J.err; cut -c1-200 gpurun_out/bench_J.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_J20.json 2> gpurun_out/bench_J20.err; cut -c1-200 gpurun_out/bench_J20.json
