#!/bin/bash
# A/B of the hand-off poll interval (scripts/devtools/build_variant.py sK anymdp.hip mixed.hip -DXV_HAND_POLL_SLEEP=K): 2a, 2b, mixed
for i in 1 2 3; do
  for v in s1 clk s2 s3 s4 s6 s8; do
    if [ $v = s1 ]; then unset XV_LIB_PATH; else export XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_$v.so; fi
    a=$(timeout 200 python bench.py --steps 2048 --warmup 256 --repeats 15 --no-cpu-baseline --no-families --no-variants --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f' % (d['ms_per_step']*1e3))")
    b=$(timeout 200 python bench.py --tasks 1024 --steps 2048 --warmup 256 --repeats 15 --no-cpu-baseline --no-families --no-variants --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f' % (d['ms_per_step']*1e3))")
    m=$(timeout 200 python bench.py --workload mixed --steps 2048 --warmup 256 --repeats 9 --no-cpu-baseline --no-allgather 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f' % (d['ms_per_step']*1e3))")
    echo "round $i $v 2a $a 2b $b mixed $m"
  done
done
