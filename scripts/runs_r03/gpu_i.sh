#!/bin/bash
# round 3, visit I: kernel time and counters of the cooperative multi-token step
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_i
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_i -o i -- python3 scripts/bench_families.py --families anymdp_tok --steps 300 --warmup 30 > gpurun_out/i_tok.jsonl 2> gpurun_out/i_tok.err; echo "rc=$?"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_i/**/i_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "tok" in r["Name"]:
        print("%-100s calls %6s avg %12.1f ns" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])))
PY
bash scripts/pmc_kernel.sh tok_r03 anymdp_tok_step_coop scripts/bench_families.py --families anymdp_tok --steps 200 --warmup 20 > gpurun_out/i_pmc.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_tok_r03.json"))
for k, v in d["kernels"].items():
    print(k[:60], {x: (round(v[x], 3) if isinstance(v[x], float) else v[x]) for x in ("FETCH_SIZE", "WRITE_SIZE", "hbm_bytes_per_launch_corrected", "SQ_INSTS_VALU_per_wave", "SQ_INSTS_VMEM_RD_per_wave", "SQ_INSTS_LDS_per_wave", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY_over_WAVE_CYCLES", "SQ_WAIT_INST_ANY_over_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU_over_WAVE_CYCLES", "TCC_HIT_sum", "TCC_MISS_sum") if x in v})
PY
