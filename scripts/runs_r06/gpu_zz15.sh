#!/bin/bash
# round 6, visit zz15: the final tree once more — GPU suite, smoke, maze lines
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/zz15_pytest.log 2>&1; echo "rc=$?"; grep -n "passed\|failed" $O/zz15_pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python scripts/bench_families.py --families maze64,maze256 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['workload'][-16:], d['filter'], {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
