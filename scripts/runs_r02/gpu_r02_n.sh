#!/bin/bash
mkdir -p gpurun_out
timeout 900 python scripts/bench_families.py --families teacher > gpurun_out/fam_n.jsonl 2> gpurun_out/fam_n.err
cat gpurun_out/fam_n.jsonl; tail -3 gpurun_out/fam_n.err
timeout 900 python -m pytest tests/test_gpu_maze_agent.py -x -q -m gpu 2>&1 | tail -3
