# timing probe (frames are wrong in the variants): what the ray caster costs when the lanes' filter windows share lines
cd $GRAFT_REPO_ROOT
for v in intree loc1 loc2; do
  if [ $v = intree ]; then unset XV_LIB_PATH; else export XV_LIB_PATH=scripts/devtools/_build/libxeno_$v.so; fi
  for fam in maze64 maze64_direct maze64_f32 maze256; do
    timeout 600 python scripts/bench_families.py --families $fam 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', '$fam', {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
  done
done
