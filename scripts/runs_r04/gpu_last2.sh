#!/bin/bash
# round 4, the record at the final tree: gpu_last.sh (whole GPU suite, smoke, quick tour, bench lines, rocprofv3 kernel stats and
# PMC traffic of the headline, token step, python loop), then every family (clean lines + rocprofv3 kernel stats), the envs
# sweep and the SQ / texture-path counters of the headline, token, mixed and ray-cast kernels.  -> gpurun_out/r04_z_*
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/runs_r04/gpu_last.sh
T=r04_z
echo "== rocprof families"
rm -rf gpurun_out/prof_fam
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fam -o fam -- python3 scripts/bench_families.py --families linds,cartpole,acrobot,maze64,maze64_direct,maze64_f32,mixed,anymdp_tok,anymdp_tok_refdist,anymdp_refdist --steps 200 --warmup 20 > gpurun_out/${T}_families_prof.jsonl 2> gpurun_out/${T}_families_prof.err; echo "rc=$?"
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_fam/**/fam_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
keep = [r for r in rows if any(k in r["Name"] for k in ("anymdp", "linds", "maze", "cartpole", "acrobot", "mixed"))]
with open("gpurun_out/${T}_kernel_stats_families.csv", "w", newline="") as o:
    w = csv.DictWriter(o, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
for r in keep:
    if "step" in r["Name"] or "raycast" in r["Name"] or "rollout" in r["Name"]:
        print("%-100s calls %6s avg %10.1f ns" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])))
PY
echo "== families clean"; timeout 1200 python scripts/bench_families.py --families linds,cartpole,acrobot,maze64,maze64_direct,maze64_f32,maze256,maze256_direct,maze256_f32,mixed,anymdp_refdist,teacher > gpurun_out/${T}_bench_families.jsonl 2> gpurun_out/${T}_families.err; echo "rc=$?"; cut -c1-420 gpurun_out/${T}_bench_families.jsonl
echo "== envs sweep"
timeout 1500 python bench.py --sweep-envs 4096,16384,65536,131072 --steps 960 --warmup 96 --sweep-out gpurun_out/${T}_anymdp_envs_sweep.json 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
for r in d['rows']: print({k: (round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ('envs','search','us_per_step','frac','frac_traffic','frac_of_line_rate','frac_of_floor','frac_note','fused_rollout_us_per_step')})"
echo "== SQ counters"
bash scripts/pmc_kernel.sh anymdp2a_${T} anymdp_step bench.py --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline --no-families > gpurun_out/${T}_pmc_sq_anymdp.log 2>&1; tail -2 gpurun_out/${T}_pmc_sq_anymdp.log | cut -c1-200
bash scripts/pmc_kernel.sh tok_${T} anymdp_tok_step_coop scripts/bench_families.py --families anymdp_tok --steps 300 > gpurun_out/${T}_pmc_tok.log 2>&1
bash scripts/pmc_kernel.sh mixed_${T} mixed_step scripts/bench_families.py --families mixed --steps 200 > gpurun_out/${T}_pmc_mixed.log 2>&1
export XV_MAZE_STEPS=6
PMC_EXTRA="TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr" bash scripts/pmc_kernel.sh raycast_spec_${T} maze_raycast scripts/bench_families.py --families maze64 > gpurun_out/${T}_pmc_raycast_spec.log 2>&1
PMC_EXTRA="TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr" bash scripts/pmc_kernel.sh raycast_direct_${T} maze_raycast scripts/bench_families.py --families maze64_direct > gpurun_out/${T}_pmc_raycast_direct.log 2>&1
unset XV_MAZE_STEPS
for k in anymdp2a tok mixed raycast_spec raycast_direct; do cp gpurun_out/pmc_${k}_${T}.json gpurun_out/${T}_pmc_sq_${k}.json 2>/dev/null; done
python3 - <<PY
import json
for k in ("anymdp2a", "tok", "mixed", "raycast_spec", "raycast_direct"):
    try:
        d = json.load(open("gpurun_out/r04_z_pmc_sq_%s.json" % k))
    except Exception as ex:
        print(k, "unreadable", ex); continue
    for n, v in d["kernels"].items():
        print(k, n[:70], {a: round(v[a], 2) for a in ("SQ_INSTS_VALU_per_wave", "SQ_INSTS_LDS_per_wave", "SQ_WAIT_INST_ANY_over_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU_over_WAVE_CYCLES", "hbm_bytes_per_launch_corrected") if a in v})
PY
