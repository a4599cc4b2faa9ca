#!/usr/bin/env python3
"""Host-side cost of the Python VectorEnv.step() path (what a training loop calls) next to the C step loop."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xenoverse_amd import _lib  # noqa: E402
from xenoverse_amd.anymdp import AnyMDPVecEnv, row_lines  # noqa: E402

n, n_task, S, A = 65536, 1024, 64, 8
def run(copy):
    env = AnyMDPVecEnv(n, seed=1, autoreset_mode="same_step", copy=copy)
    d = env.device
    tab = dict(S=S, A=A, s0_max=4, rows=torch.empty((n_task, S, A, row_lines(S), 16), dtype=torch.float64, device=d),
               state_map=torch.empty((n_task, S), dtype=torch.int32, device=d),
               term_mask=torch.empty((n_task, 1), dtype=torch.int64, device=d),
               s0_cdf=torch.empty((n_task, 4), dtype=torch.float64, device=d),
               s0_ids=torch.empty((n_task, 4), dtype=torch.int32, device=d),
               max_steps=torch.empty(n_task, dtype=torch.int32, device=d))
    _lib.check(env.lib.xv_anymdp_synth_tasks(env.engine.handle, 7, 0, n_task, S, A, 4, *[_lib.ptr(tab[k]) for k in
               ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")]))
    env.set_task(tab)
    env.reset()
    a = torch.randint(0, A, (n,), device=d, dtype=torch.int32)
    for _ in range(50):
        env.step(a)
    torch.cuda.synchronize()
    K = 2000
    t0 = time.perf_counter()
    for _ in range(K):
        obs, r, te, tr, info = env.step(a)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("copy=%-5s python VectorEnv.step: %.1f us per vector step (%.2e env-steps/s)" % (copy, dt / K * 1e6, n * K / dt))
    t0 = time.perf_counter()
    for _ in range(K):
        obs, r, te, tr, info = env.step(a)
        a = (obs & 7).to(torch.int32)          # a trivial "policy" on the device
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("copy=%-5s with a device-side policy op: %.1f us per vector step" % (copy, dt / K * 1e6))
    env.close()


for c in (True, False):
    run(c)
