#!/bin/bash
# final family numbers of round 2 + config 2b with the bucket search
mkdir -p gpurun_out
timeout 1800 python scripts/bench_families.py --families linds,cartpole,acrobot,maze64,maze64_f32,maze256,mixed,anymdp_tok,teacher --steps 400 --warmup 40 > gpurun_out/r02_final_bench_families.jsonl 2> gpurun_out/fam_E.err
cut -c1-260 gpurun_out/r02_final_bench_families.jsonl
timeout 900 python bench.py --tasks 1024 --no-cpu-baseline > gpurun_out/bench_E_2b.json 2> gpurun_out/bench_E_2b.err; cut -c1-330 gpurun_out/bench_E_2b.json; python -c "
import json; d = json.load(open('gpurun_out/bench_E_2b.json')); print(d['config']['search'], d['roofline']['avg_launch_us'])"
