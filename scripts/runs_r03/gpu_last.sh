#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/last_pytest_gpu.log 2>&1; echo "rc=$?"; grep -h "passed\|failed" gpurun_out/last_pytest_gpu.log | tail -1
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== bench driver flags"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/last_bench_steps20.json 2> gpurun_out/last_bench_steps20.err; echo "rc=$?"
python - <<PY
import json
d = json.load(open("gpurun_out/last_bench_steps20.json"))
r = d["roofline"]
print("value %.4e ms/step %.5f kernel us %.3f frac %.3f traffic %s cpu %s" % (d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"], r.get("traffic"), d["cpu_baseline"]["value"]))
for k, v in (d.get("families") or {}).items():
    print("   ", k, {a: v.get(a) for a in ("ms_per_step", "env_steps_per_s", "error")}, (v.get("roofline") or {}).get("frac"))
PY
