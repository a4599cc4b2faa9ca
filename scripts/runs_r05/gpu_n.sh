#!/bin/bash
# round 5, call N: 20-step bursts with both chains as branches of one graph (one hipGraphLaunch); maze tests after the revert
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_n
mkdir -p $O
for fq in unset 2 4; do
  unset DEBUG_HIP_FORCE_GRAPH_QUEUES
  [ $fq != unset ] && export DEBUG_HIP_FORCE_GRAPH_QUEUES=$fq
  XV_ANYMDP_PIPE_ONE_GRAPH=1 XV_ANYMDP_PIPE_MIN_STEPS=20 timeout 300 python scripts/devtools/probe_chains.py --tag onegraph_fq$fq --ks 1 --overlap --repeats 5 --steps 640 > $O/onegraph_fq$fq.jsonl 2> $O/onegraph_fq$fq.err
  echo "one graph, DEBUG_HIP_FORCE_GRAPH_QUEUES=$fq rc=$?"
  python3 - $O/onegraph_fq$fq.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print("  %-8s us/step %.3f  short %.3f (min %.3f)  err %s state %s" % (d["how"], d["us_per_step"], d["short_us_per_step"], d["short_us_min"], d["device_error_flags"], d["overlap_state"]))
PY
done
unset DEBUG_HIP_FORCE_GRAPH_QUEUES
timeout 900 python -m pytest tests/test_gpu_maze.py -x -q > $O/pytest_maze.txt 2>&1; echo "pytest maze rc=$?"; tail -2 $O/pytest_maze.txt
