# counters of the speculated and the direct exact ray caster (64 x 64, 16,384 envs)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export XV_MAZE_STEPS=6
PMC_EXTRA="TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr" bash scripts/pmc_kernel.sh raycast_spec_r04 maze_raycast scripts/bench_families.py --families maze64 > gpurun_out/r04_u_pmc_spec.log 2>&1
tail -3 gpurun_out/r04_u_pmc_spec.log | cut -c1-300
PMC_EXTRA="TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr" bash scripts/pmc_kernel.sh raycast_direct_r04 maze_raycast scripts/bench_families.py --families maze64_direct > gpurun_out/r04_u_pmc_direct.log 2>&1
tail -3 gpurun_out/r04_u_pmc_direct.log | cut -c1-300
python - <<PY
import json
for t in ("spec", "direct"):
    d = json.load(open("gpurun_out/pmc_raycast_%s_r04.json" % t))
    for k, v in d["kernels"].items():
        print(t, k[:60], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})
PY
