"""Dev measurement: xv_anymdp_solve (device value iteration, one workgroup per task) on synthetic S=64, A=8 tasks."""
import sys
import time

import torch

sys.path.insert(0, '.')
from xenoverse_amd import _lib  # noqa: E402
from xenoverse_amd.anymdp import AnyMDPVecEnv, row_lines  # noqa: E402

n_task = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S, A = 64, 8
env = AnyMDPVecEnv(n_task, seed=1)
d = env.device
tab = dict(S=S, A=A, s0_max=4, rows=torch.empty((n_task, S, A, row_lines(S), 16), dtype=torch.float64, device=d),
           state_map=torch.empty((n_task, S), dtype=torch.int32, device=d), term_mask=torch.empty((n_task, 1), dtype=torch.int64, device=d),
           s0_cdf=torch.empty((n_task, 4), dtype=torch.float64, device=d), s0_ids=torch.empty((n_task, 4), dtype=torch.int32, device=d),
           max_steps=torch.empty(n_task, dtype=torch.int32, device=d))
_lib.check(env.lib.xv_anymdp_synth_tasks(env.engine.handle, 7, 0, n_task, S, A, 4, *[_lib.ptr(tab[k]) for k in ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")]))
env.set_task(tab)
env.solve(return_q=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
q, g, it = env.solve(return_q=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
its = it.float()
print("n_task %d: %.1f ms (%.1f us per task, %.0f tasks/s); sweeps mean %.0f max %d" % (n_task, dt * 1e3, dt / n_task * 1e6, n_task / dt, its.mean().item(), int(it.max())))
