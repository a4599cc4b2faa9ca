#!/bin/bash
# round 3, the record: whole GPU suite, smoke, the bench line (default flags, the driver's flags, graph replay), rocprofv3
# kernel stats of bench.py and of every family, PMC traffic of the AnyMDP step kernel (keyed on its source hash) and
# counters of the LinDS / maze-move / mixed kernels, the two-ranks-on-one-GPU functional run.  Everything -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-final}
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest_gpu.log 2>&1; echo "rc=$?"; grep -n "passed\|failed\|Error" gpurun_out/${T}_pytest_gpu.log | head -5
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== bench default"; timeout 900 python bench.py --fused > gpurun_out/${T}_bench_2a.json 2> gpurun_out/${T}_bench_2a.err; echo "rc=$?"
echo "== bench driver flags"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_2a_steps20.json 2> gpurun_out/${T}_bench_2a_steps20.err; echo "rc=$?"
echo "== bench plain launches (2000 steps)"; timeout 900 python bench.py --graph off --no-cpu-baseline --no-families > gpurun_out/${T}_bench_2a_plain.json 2>/dev/null; echo "rc=$?"
echo "== bench 2b"; timeout 900 python bench.py --tasks 1024 --no-cpu-baseline --no-families --fused > gpurun_out/${T}_bench_2b.json 2>/dev/null; echo "rc=$?"
python - <<PY
import json
for f in ("bench_2a", "bench_2a_steps20", "bench_2a_plain", "bench_2b"):
    try:
        d = json.load(open("gpurun_out/${T}_%s.json" % f))
    except Exception as ex:
        print(f, "unreadable", ex); continue
    print("%-18s value %.4e ms/step %.5f kernel us %.3f frac %.3f launch: %s" % (f, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["config"]["launch"][-40:]), d.get("fused_rollout_env_steps_per_s_rank0"))
    for k, v in (d.get("families") or {}).items():
        print("   ", k, {a: v.get(a) for a in ("ms_per_step", "env_steps_per_s", "wall_s", "error")}, (v.get("roofline") or {}).get("frac"))
PY
echo "== rocprof bench"
rm -rf gpurun_out/prof_b
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b -o b -- python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-families > /dev/null 2> gpurun_out/${T}_prof_b.err; echo "rc=$?"
echo "== rocprof families"
rm -rf gpurun_out/prof_fam
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fam -o fam -- python3 scripts/bench_families.py --families linds,cartpole,acrobot,maze64,maze64_f32,mixed,anymdp_tok --steps 200 --warmup 20 > gpurun_out/${T}_families_prof.jsonl 2> gpurun_out/${T}_families_prof.err; echo "rc=$?"
python3 - <<PY
import csv, glob
for tag, pat, out in (("bench", "gpurun_out/prof_b/**/b_kernel_stats.csv", "gpurun_out/${T}_kernel_stats_anymdp_2a.csv"),
                      ("families", "gpurun_out/prof_fam/**/fam_kernel_stats.csv", "gpurun_out/${T}_kernel_stats_families.csv")):
    f = glob.glob(pat, recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if any(k in r["Name"] for k in ("anymdp", "linds", "maze", "cartpole", "acrobot", "mixed"))]
    with open(out, "w", newline="") as o:
        w = csv.DictWriter(o, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
    for r in keep:
        if "step" in r["Name"] or "raycast" in r["Name"] or "rollout" in r["Name"]:
            print("%-9s %-100s calls %6s avg %10.1f ns" % (tag, r["Name"][:100], r["Calls"], float(r["AverageNs"])))
PY
echo "== families clean"; timeout 900 python scripts/bench_families.py --families linds,cartpole,acrobot,maze64,maze64_f32,maze256,mixed,anymdp_tok,teacher > gpurun_out/${T}_bench_families.jsonl 2> gpurun_out/${T}_families.err; echo "rc=$?"; cut -c1-330 gpurun_out/${T}_bench_families.jsonl
echo "== PMC anymdp traffic"; bash scripts/gpu_pmc.sh > gpurun_out/${T}_pmc_anymdp.log 2>&1; tail -3 gpurun_out/${T}_pmc_anymdp.log | cut -c1-400
echo "== PMC linds"; bash scripts/pmc_kernel.sh linds_${T} linds_step_mfma scripts/bench_families.py --families linds_mfma --steps 300 --warmup 30 > gpurun_out/${T}_pmc_linds.log 2>&1
echo "== PMC maze move"; bash scripts/pmc_kernel.sh maze_m9_${T} maze_step9 scripts/bench_families.py --families maze64 --steps 200 > gpurun_out/${T}_pmc_maze.log 2>&1
echo "== PMC mixed"; bash scripts/pmc_kernel.sh mixed_${T} mixed_step scripts/bench_families.py --families mixed --steps 200 > gpurun_out/${T}_pmc_mixed.log 2>&1
python - <<PY
import json
for n in ("linds_${T}", "maze_m9_${T}", "mixed_${T}"):
    try:
        d = json.load(open("gpurun_out/pmc_%s.json" % n))
        for k, v in d["kernels"].items():
            print(n, k[:50], {x: (round(v[x], 3) if isinstance(v[x], float) else v[x]) for x in ("hbm_bytes_per_launch_corrected", "SQ_INSTS_VALU_per_wave", "SQ_WAVES", "SQ_WAIT_ANY_over_WAVE_CYCLES", "SQ_WAIT_INST_ANY_over_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU_over_WAVE_CYCLES") if x in v})
    except Exception as ex:
        print(n, "failed", ex)
PY
echo "== n2 functional (two ranks sharing the GPU)"
bash scripts/gpu_n2_functional.sh 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l[l.index('{'):]); print({k: d.get(k) for k in ('n_gpus', 'value', 'rccl', 'rccl_ranks', 'transport', 'transport_requested', 'transport_note', 'allgather_timeout')}, d['config']['exchange'][:160])
except Exception as ex:
    print('n2 line unreadable:', ex, l[:300])
" | tee gpurun_out/${T}_n2.txt
