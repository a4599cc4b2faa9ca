#!/bin/bash
# counters of the acrobot step kernel (why 47 us for 1,024 waves?)
bash scripts/pmc_kernel.sh acrobot acrobot_step_kernel scripts/bench_families.py --families acrobot --steps 60 --warmup 5 2>&1 | tail -8
cat gpurun_out/pmc_acrobot.json
