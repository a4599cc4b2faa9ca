#!/bin/bash
# round 5, call F: 20-step bursts — graph packet capture on / off, overlap forced for short calls
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_f
mkdir -p $O
for cap in unset 0 1; do
  for min in 64 20; do
    unset DEBUG_CLR_GRAPH_PACKET_CAPTURE
    [ $cap != unset ] && export DEBUG_CLR_GRAPH_PACKET_CAPTURE=$cap
    export XV_ANYMDP_PIPE_MIN_STEPS=$min
    timeout 600 python scripts/devtools/probe_chains.py --tag cap${cap}_min$min --ks 1 --overlap --repeats 5 --steps 640 > $O/burst_cap${cap}_min$min.jsonl 2> $O/burst_cap${cap}_min$min.err
    echo "capture=$cap min_steps=$min rc=$?"
    python3 - $O/burst_cap${cap}_min$min.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print("  %-8s us/step %.3f  short %.3f (min %.3f)  err %s state %s" % (d["how"], d["us_per_step"], d["short_us_per_step"], d["short_us_min"], d["device_error_flags"], d["overlap_state"]))
PY
  done
done
