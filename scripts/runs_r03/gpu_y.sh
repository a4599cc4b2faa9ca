#!/bin/bash
# the texture path of the ray caster in counters: exact and fp32 filter
set -u
export TMPDIR=/tmp
# (a TA_* group — TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum ... — never returned: 2 x 600 s of timeout; left out)
export PMC_EXTRA="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum|TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum"
bash scripts/pmc_kernel.sh rc_exact maze_raycast scripts/bench_families.py --families maze64 --steps 200 > gpurun_out/pmc_rc_exact.log 2>&1
bash scripts/pmc_kernel.sh rc_f32 maze_raycast scripts/bench_families.py --families maze64_f32 --steps 200 > gpurun_out/pmc_rc_f32.log 2>&1
tail -60 gpurun_out/pmc_rc_exact.log | grep -E "^void|   " | cut -c1-120
tail -60 gpurun_out/pmc_rc_f32.log | grep -E "^void|   " | cut -c1-120
