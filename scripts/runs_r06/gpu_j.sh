#!/bin/bash
# round 6, visit j: the bench line with its own PMC traffic (child passes), at the driver's flags; the bench-line tests
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_bench_line.py -x -q --timeout 600 > $O/j_pytest.log 2>&1; echo "rc=$?"; tail -5 $O/j_pytest.log
S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/j_bench_steps20.json 2> $O/j_bench_steps20.err; echo "rc=$? wall $(( $(date +%s) - S )) s"; tail -3 $O/j_bench_steps20.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/j_bench_steps20.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"])
r = d["roofline"]
print({k: r[k] for k in ("bound", "achieved", "frac", "frac_survey_bytes", "traffic", "traffic_source", "traffic_committed_profile", "traffic_live")})
for m in ("one_stream", "overlapped", "fused_rollout"):
    row = d["long_call"][m]
    print(m, round(row["us_per_step"], 3), round(row["wall_us_per_step"], 3), "%.3e" % row["env_steps_per_s"], row["overlap_state"], {k: row["roofline"][k] for k in ("frac", "traffic", "frac_survey_bytes")})
print("sustain", d["sustain"], "ratio", d["long_call"].get("sustain_over_long_call"))
PY
