#!/bin/bash
set -u
export TMPDIR=/tmp
for cfg in "1 1" "2 1" "1 2" "2 2" "4 2" "2 4" "4 4"; do
  set -- $cfg
  XV_TOK_DACT=$1 XV_TOK_DOBS=$2 timeout 300 python scripts/bench_families.py --families anymdp_tok --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('d_act $1 d_obs $2', d['us_per_step'])"
done
XV_TOK_MODE=disabled timeout 300 python scripts/bench_families.py --families anymdp_tok --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('disabled 2 2', d['us_per_step'])"
