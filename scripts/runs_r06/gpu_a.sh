#!/bin/bash
# round 6, visit a: the replay of expired hand-offs (AnyMDP step_many, mixed step_many), gates on every stream, neighbour tests,
# and the bench line with the long_call block at the driver's flags
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== new chain / mixed tests"
timeout 1500 python -m pytest tests/test_gpu_chains.py tests/test_gpu_mixed_shard.py -x -q -s > gpurun_out/a_pytest_chains.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/a_pytest_chains.log
echo "== bench at the driver's flags"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/a_bench_steps20.json 2> gpurun_out/a_bench_steps20.err; echo "rc=$?"; tail -3 gpurun_out/a_bench_steps20.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/a_bench_steps20.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"])
print("roofline", {k: d["roofline"][k] for k in ("bound", "achieved", "frac", "frac_survey_bytes", "traffic", "basis")})
print("long_call", json.dumps(d.get("long_call"), indent=1)[:3000])
print("sustain", d.get("sustain"))
print("launch variants", d.get("search_variants", {}).get("launch"))
PY
