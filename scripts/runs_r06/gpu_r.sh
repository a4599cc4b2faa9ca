#!/bin/bash
# round 6, visit r: last sanity on the committed tree — quickstart, smoke, the driver's bench command
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_r
mkdir -p $O
timeout 300 python examples/quickstart.py > $O/quickstart.txt 2>&1; echo "quickstart rc=$?"; tail -4 $O/quickstart.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt
S0=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_2a_steps20.json 2> $O/bench_2a_steps20.err; echo "bench rc=$? wall $(( $(date +%s) - S0 )) s"
python3 - $O/bench_2a_steps20.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
r = d["roofline"]
print("value %.4g ms/step %.5f frac %s traffic %s (%s)" % (d["value"], d["ms_per_step"], r["frac"], r["traffic"], r["traffic_source"][:40]))
for m in ("one_stream", "overlapped", "fused_rollout"):
    row = d["long_call"][m]
    print("  %-14s %.3f us %.4g env-steps/s state %s frac %s" % (m, row["us_per_step"], row["env_steps_per_s"], row["overlap_state"], row["roofline"]["frac"]))
f = d["families"]
print("  families:", {k: round(v.get("ms_per_step", 0) * 1e3, 2) for k, v in f.items()})
print("  python_loop:", f["python_loop"]["us_per_vector_step"])
PY
