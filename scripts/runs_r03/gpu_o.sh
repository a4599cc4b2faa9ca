#!/bin/bash
# round 3, visit O: owner-lane results by selects instead of exec-masked branches (A/B on the headline workload)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
export XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_sel.so
timeout 900 python -m pytest tests/test_gpu_anymdp.py -x -q -k "not full_size" 2>&1 | tail -1
for v in default sel default sel; do
  if [ $v = default ]; then unset XV_LIB_PATH; else export XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_$v.so; fi
  timeout 600 python bench.py --no-cpu-baseline --no-families 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v graph: value %.4e us/step %.3f' % (d['value'], d['ms_per_step']*1e3))"
done
