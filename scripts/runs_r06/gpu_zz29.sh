#!/bin/bash
# round 6, visit zz29: the final tree — the whole GPU suite, smoke, the driver's bench command
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu --timeout 900 > $O/zz29_pytest.log 2>&1; echo "rc=$?"; grep -n "passed\|failed" $O/zz29_pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/zz29_bench_steps20.json 2> $O/zz29_bench_steps20.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/zz29_bench_steps20.json") if l.startswith("{")][-1])
lc = d.get("long_call", {})
print("value %.4g steps %d ms/step %.5f frac %.3f" % (d["value"], d["steps"], d["ms_per_step"], d["roofline"]["frac"]),
      {k: round(lc[k]["us_per_step"], 2) for k in ("one_stream", "overlapped", "fused_rollout") if k in lc},
      {k: (round(v["ms_per_step"], 4) if isinstance(v, dict) and "ms_per_step" in v else None) for k, v in d.get("families", {}).items()})
for k in ("mazeworld_64", "mazeworld_256"):
    r = d["families"][k]["roofline"]
    print(k, r["bound"], round(r["frac"], 3), r["valu_instructions_per_pixel"], r["source"])
PY
