#!/bin/bash
# round 5, call C: overlapped step_many — what the 20-step burst pays for: fork event, launch order
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_c
mkdir -p $O
for v in base nofork sidefirst nofork_sidefirst; do
  unset XV_PIPE_NOFORK XV_PIPE_SIDE_FIRST
  case $v in nofork) export XV_PIPE_NOFORK=1;; sidefirst) export XV_PIPE_SIDE_FIRST=1;; nofork_sidefirst) export XV_PIPE_NOFORK=1 XV_PIPE_SIDE_FIRST=1;; esac
  timeout 600 python scripts/devtools/probe_chains.py --tag 2a_$v --ks 1 --overlap --repeats 5 > $O/overlap_2a_$v.jsonl 2> $O/overlap_2a_$v.err
  echo "$v rc=$?"
  python3 - $O/overlap_2a_$v.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print("  %-8s us/step %.3f  short %.3f (min %.3f)  err %s state %s" % (d["how"], d["us_per_step"], d["short_us_per_step"], d["short_us_min"], d["device_error_flags"], d["overlap_state"]))
PY
done
unset XV_PIPE_NOFORK XV_PIPE_SIDE_FIRST
timeout 900 python -m pytest tests/test_gpu_chains.py -x -q -k overlapped > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
