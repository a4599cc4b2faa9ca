#!/bin/bash
# round 6, visit n (the overlapped xv_linds_step_many was removed after this run: DESIGN.md 4.1; the script is kept as the record of what ran):
# fills the device, the second cannot be resident with it: XV_LINDS_PIPE_FORCE=1 tries it all the same)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_linds.py tests/test_gpu_chains.py -x -q --timeout 300 -k "linds or overlap or slot or handle" > $O/n_pytest.log 2>&1; echo "rc=$?"; tail -5 $O/n_pytest.log
for force in 0 1; do
  if [ $force = 1 ]; then export XV_LINDS_PIPE_FORCE=1; else unset XV_LINDS_PIPE_FORCE; fi
  for rep in 1 2; do
  timeout 600 python scripts/bench_families.py --families linds_mfma --steps 2000 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('force=$force', {k: round(x, 2) for k, x in d['us_per_step'].items()}, d.get('overlap_note'))
"
  done
done | tee $O/n_linds_overlap.txt
unset XV_LINDS_PIPE_FORCE
python - <<'PY' | tee -a gpurun_out/n_linds_overlap.txt
# smaller batches: where two launches fit
import sys, time
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import torch
import bench_families as bf
from xenoverse_amd.linds import LinDSVecEnv
for n in (16384, 32768):
    tasks = bf.linds_tasks(n // 64)
    env = LinDSVecEnv(n, autoreset_mode="same_step", seed=1)
    env.set_task(tasks)
    env.reset()
    aP = torch.rand((8, n, 8), device=env.device) * 2 - 1
    ring = env.step_many(8, aP)
    out = {}
    for ov in (False, True):
        env.set_step_many_overlap(ov)
        env.step_many(2000, aP, out=ring)
        out[ov] = bf.timed(lambda: env.step_many(2000, aP, out=ring), 3, 1) / 2000
        torch.cuda.synchronize()
        st = env.step_many_overlap_state
    print("envs", n, "one stream %.2f us, overlapped %.2f us, state %d, flags %d" % (out[False], out[True], st, env.check_errors()))
    env.set_step_many_overlap(False)
    env.close()
PY
