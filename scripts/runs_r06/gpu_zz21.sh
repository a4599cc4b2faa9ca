#!/bin/bash
# round 6, visit zz21: the driver's command and the default command on the final tree (after the ray caster's fetch work)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
t0=$(date +%s.%N)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/zz21_bench_steps20.json 2> $O/zz21_bench_steps20.err; echo "rc=$? $(echo "$(date +%s.%N) - $t0" | bc) s"
python bench.py > $O/zz21_bench_default.json 2> $O/zz21_bench_default.err; echo "rc=$?"
python - <<'PY'
import json
for f in ("zz21_bench_steps20", "zz21_bench_default"):
    d = json.loads([l for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][-1])
    lc = d.get("long_call", {})
    print(f, "value %.4g steps %d ms/step %.5f frac %.3f" % (d["value"], d["steps"], d["ms_per_step"], d["roofline"]["frac"]),
          {k: round(lc[k]["us_per_step"], 2) for k in ("one_stream", "overlapped", "fused_rollout") if k in lc},
          {k: (round(v["ms_per_step"], 4) if isinstance(v, dict) and "ms_per_step" in v else None) for k, v in d.get("families", {}).items()})
PY
