#!/bin/bash
# round 3, visit N: AnyMDP outputs non-temporal? (A/B on the headline workload)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in default ant default ant; do
  if [ $v = default ]; then unset XV_LIB_PATH; else export XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_$v.so; fi
  timeout 600 python bench.py --no-cpu-baseline --no-families --graph off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v plain: value %.4e us/step %.3f' % (d['value'], d['ms_per_step']*1e3))"
  timeout 600 python bench.py --no-cpu-baseline --no-families 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v graph: value %.4e us/step %.3f' % (d['value'], d['ms_per_step']*1e3))"
done
