#!/bin/bash
# round 5, call E: overlapped step_many with plain alternating launches for short calls: parity, A/B, burst timeline
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_e
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chains.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
for w in 2a 2b; do
  t=""; [ $w = 2b ] && t="--tasks 1024"
  timeout 600 python scripts/devtools/probe_chains.py --tag $w --ks 1 --overlap --repeats 7 $t > $O/overlap_$w.jsonl 2> $O/overlap_$w.err
  echo "$w rc=$?"
  python3 - $O/overlap_$w.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print("  %-8s us/step %.3f  short %.3f (min %.3f)  err %s state %s" % (d["how"], d["us_per_step"], d["short_us_per_step"], d["short_us_min"], d["device_error_flags"], d["overlap_state"]))
PY
done
bash scripts/runs_r05/gpu_d.sh > $O/d.log 2>&1; cp gpurun_out/r05_d/burst_timeline.txt $O/burst_timeline.txt; head -30 $O/burst_timeline.txt
