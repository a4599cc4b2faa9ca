#!/bin/bash
# round 5, call O: overlapped step_many, one poll in flight per wave vs two
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_o
mkdir -p $O
for v in intree poll2 intree poll2; do
  if [ $v = intree ]; then unset XV_LIB_PATH; else export XV_LIB_PATH=scripts/devtools/_build/libxeno_$v.so; fi
  for w in 2a 2b; do
    t=""; [ $w = 2b ] && t="--tasks 1024"
    timeout 300 python scripts/devtools/probe_chains.py --tag ${v}_$w --ks 1 --overlap --repeats 7 --short 0 $t 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    if d['how'] == 'overlap': print('$v $w overlap us/step %.3f (min %.3f) err %s' % (d['us_per_step'], d['us_min'], d['device_error_flags']))
"
  done
done | tee $O/poll_depth_ab.txt
