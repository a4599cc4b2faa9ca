#!/bin/bash
# round 3, families of record after the last kernel changes (sorted maze move, acrobot wrap): full GPU suite, rocprofv3
# kernel stats of every family, clean bench_families lines, PMC of the maze move and acrobot kernels.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-finalc}
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest_gpu.log 2>&1; echo "rc=$?"; grep -h "passed\|failed" gpurun_out/${T}_pytest_gpu.log | tail -1
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== quickstart"; timeout 600 python examples/quickstart.py > gpurun_out/${T}_quickstart.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/${T}_quickstart.log | cut -c1-200
echo "== bench default"; timeout 900 python bench.py --fused > gpurun_out/${T}_bench_2a.json 2> gpurun_out/${T}_bench_2a.err; echo "rc=$?"
echo "== bench driver flags"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_2a_steps20.json 2> gpurun_out/${T}_bench_2a_steps20.err; echo "rc=$?"
python - <<PY
import json
for f in ("bench_2a", "bench_2a_steps20"):
    d = json.load(open("gpurun_out/${T}_%s.json" % f))
    r = d["roofline"]
    print("%-18s value %.4e ms/step %.5f kernel us %.3f frac %.3f traffic %s" % (f, d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"], r.get("traffic")))
    for k, v in (d.get("families") or {}).items():
        print("   ", k, {a: v.get(a) for a in ("ms_per_step", "env_steps_per_s", "error")}, (v.get("roofline") or {}).get("frac"))
PY
echo "== rocprof families"
rm -rf gpurun_out/prof_fam
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fam -o fam -- python3 scripts/bench_families.py --families linds,cartpole,acrobot,maze64,maze64_f32,mixed,anymdp_tok --steps 200 --warmup 20 > gpurun_out/${T}_families_prof.jsonl 2> gpurun_out/${T}_families_prof.err; echo "rc=$?"
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_fam/**/fam_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
keep = [r for r in rows if any(k in r["Name"] for k in ("anymdp", "linds", "maze", "cartpole", "acrobot", "mixed"))]
with open("gpurun_out/${T}_kernel_stats_families.csv", "w", newline="") as o:
    w = csv.DictWriter(o, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
for r in keep:
    if "step" in r["Name"] or "raycast" in r["Name"] or "rollout" in r["Name"] or "sort" in r["Name"]:
        print("%-100s calls %6s avg %10.1f ns" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])))
PY
echo "== families clean"; timeout 900 python scripts/bench_families.py --families linds,cartpole,acrobot,maze64,maze64_f32,maze256,mixed,anymdp_tok,teacher > gpurun_out/${T}_bench_families.jsonl 2> gpurun_out/${T}_families.err; echo "rc=$?"; cut -c1-300 gpurun_out/${T}_bench_families.jsonl
echo "== PMC maze move"; bash scripts/pmc_kernel.sh maze_m9_${T} maze_ scripts/bench_families.py --families maze64 --steps 200 > gpurun_out/${T}_pmc_maze.log 2>&1
python - <<PY
import json
d = json.load(open("gpurun_out/pmc_maze_m9_${T}.json"))
for k, v in d["kernels"].items():
    print(k[:60], {x: (round(v[x], 3) if isinstance(v[x], float) else v[x]) for x in ("hbm_bytes_per_launch_corrected", "SQ_INSTS_VALU_per_wave", "SQ_WAVES", "SQ_ACTIVE_INST_VALU_over_WAVE_CYCLES", "dispatches") if x in v})
PY
