# round 4, trip m: envs sweep with the annotated fractions; SQ counters of the headline kernel, the cooperative token kernel,
# LinDS and the fused mixed kernel at the final source
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python bench.py --sweep-envs 4096,16384,65536,131072 --steps 1000 --warmup 100 --sweep-out gpurun_out/r04_z_anymdp_envs_sweep.json 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
for r in d['rows']: print({k: (round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ('envs','search','us_per_step','frac','frac_traffic','frac_of_line_rate','frac_of_floor','frac_note','fused_rollout_us_per_step')})"
bash scripts/pmc_kernel.sh anymdp2a_r04_z anymdp_step bench.py --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline --no-families > gpurun_out/r04_z_pmc_sq_anymdp.log 2>&1; tail -2 gpurun_out/r04_z_pmc_sq_anymdp.log | cut -c1-200
bash scripts/pmc_kernel.sh tok_r04_z anymdp_tok_step_coop scripts/bench_families.py --families anymdp_tok --steps 300 > gpurun_out/r04_z_pmc_tok.log 2>&1
bash scripts/pmc_kernel.sh linds_r04_z linds_step_mfma scripts/bench_families.py --families linds_mfma --steps 300 --warmup 30 > gpurun_out/r04_z_pmc_linds.log 2>&1
bash scripts/pmc_kernel.sh mixed_r04_z mixed_step scripts/bench_families.py --families mixed --steps 200 > gpurun_out/r04_z_pmc_mixed.log 2>&1
python - <<PY
import json
for n in ("anymdp2a_r04_z", "tok_r04_z", "linds_r04_z", "mixed_r04_z"):
    try:
        d = json.load(open("gpurun_out/pmc_%s.json" % n))
        for k, v in d["kernels"].items():
            print(n, k[:60], {x: (round(v[x], 3) if isinstance(v[x], float) else v[x]) for x in ("hbm_bytes_per_launch_corrected", "SQ_INSTS_VALU_per_wave", "SQ_INSTS_LDS_per_wave", "SQ_WAVES", "SQ_WAIT_ANY_over_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU_over_WAVE_CYCLES", "dispatches") if x in v})
    except Exception as ex:
        print(n, "failed", ex)
PY
