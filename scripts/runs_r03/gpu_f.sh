#!/bin/bash
# round 3, visit F: whole GPU suite, the bench line with its `families` object (default flags and the driver's), kernel
# stats of bench.py and of the families under rocprofv3, counters of the LinDS step kernel
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/f_pytest_gpu.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/f_pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== bench default"; timeout 900 python bench.py > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err; echo "rc=$?"; cut -c1-400 gpurun_out/f_bench.json; tail -2 gpurun_out/f_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/f_bench.json"))
print("value %.3e ms/step %.5f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
for k, v in d.get("families", {}).items():
    print(k, json.dumps(v)[:600])
PY
echo "== bench driver flags"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/f_bench_steps20.json 2> gpurun_out/f_bench_steps20.err; echo "rc=$?"; cut -c1-300 gpurun_out/f_bench_steps20.json
echo "== rocprof families"
rm -rf gpurun_out/prof_fam
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fam -o fam -- python3 scripts/bench_families.py --families linds,cartpole,acrobot,maze64,maze64_f32,mixed,anymdp_tok --steps 200 --warmup 20 > gpurun_out/f_families_prof.jsonl 2> gpurun_out/f_families_prof.err
echo "rc=$?"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_fam/**/fam_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
keep = [r for r in rows if any(k in r["Name"] for k in ("anymdp", "linds", "maze", "cartpole", "acrobot"))]
with open("gpurun_out/f_kernel_stats_families.csv", "w", newline="") as o:
    w = csv.DictWriter(o, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
for r in keep:
    print("%-100s calls %6s avg %12.1f ns" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])))
PY
echo "== families clean"; timeout 900 python scripts/bench_families.py --families linds,cartpole,acrobot,maze64,maze64_f32,maze256,mixed,anymdp_tok,teacher > gpurun_out/f_families.jsonl 2> gpurun_out/f_families.err; echo "rc=$?"; cut -c1-420 gpurun_out/f_families.jsonl
echo "== PMC linds"
bash scripts/pmc_kernel.sh linds_r03f linds_step_mfma scripts/bench_families.py --families linds_mfma --steps 300 --warmup 30 > gpurun_out/f_pmc.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_linds_r03f.json"))
for k, v in d["kernels"].items():
    print(k, {x: v[x] for x in ("FETCH_SIZE", "WRITE_SIZE", "hbm_bytes_per_launch_corrected", "SQ_INSTS_VALU_per_wave", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY_over_WAVE_CYCLES", "SQ_WAIT_INST_ANY_over_WAVE_CYCLES") if x in v})
PY
echo "== n2 functional (two ranks sharing the GPU)"
bash scripts/gpu_n2_functional.sh
