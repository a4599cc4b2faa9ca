"""Wall time of one Python-level VectorEnv.step() call (what a policy-in-the-loop user pays per vector step), per family,
copy=True (default) and copy=False, 65,536 envs (maze 16,384): the kernels take 5-9 us (maze 1.4 ms)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from xenoverse_amd import _lib
from xenoverse_amd.anymdp import AnyMDPVecEnv, row_lines
from xenoverse_amd.linds import LinDSVecEnv
from xenoverse_amd.metacontrol import CartPoleVecEnv, sample_cartpole, AcrobotVecEnv, sample_acrobot
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_families import linds_tasks


def wall(fn, n=300, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


n = 65536


def synth_tables(env, n_task=64, S=64, A=8):
    """synthetic tasks made on the device (xv_anymdp_synth_tasks, as bench.py does)"""
    d = env.device
    tab = dict(S=S, A=A, s0_max=4, rows=torch.empty((n_task, S, A, row_lines(S), 16), dtype=torch.float64, device=d),
               state_map=torch.empty((n_task, S), dtype=torch.int32, device=d),
               term_mask=torch.empty((n_task, 1), dtype=torch.int64, device=d),
               s0_cdf=torch.empty((n_task, 4), dtype=torch.float64, device=d),
               s0_ids=torch.empty((n_task, 4), dtype=torch.int32, device=d),
               max_steps=torch.empty(n_task, dtype=torch.int32, device=d))
    _lib.check(env.lib.xv_anymdp_synth_tasks(env.engine.handle, 7, 0, n_task, S, A, 4, *[_lib.ptr(tab[k]) for k in
               ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")]))
    return tab


for copy in (True, False):
    env = AnyMDPVecEnv(n, seed=1, autoreset_mode="same_step", copy=copy)
    env.set_task(synth_tables(env))
    env.reset()
    a = torch.randint(0, 8, (n,), device=env.device, dtype=torch.int32)
    print("anymdp   copy=%-5s %.1f us per step() call" % (copy, wall(lambda: env.step(a))), flush=True)
    env.close()
    env = LinDSVecEnv(n, seed=1, autoreset_mode="same_step", copy=copy)
    env.set_task(linds_tasks(1024))
    env.reset()
    a = torch.rand((n, 8), device=env.device) * 2 - 1
    print("linds    copy=%-5s %.1f us per step() call" % (copy, wall(lambda: env.step(a))), flush=True)
    env.close()
    env = CartPoleVecEnv(n, seed=1, autoreset_mode="same_step", frameskip=1, copy=copy)
    env.set_task([sample_cartpole(seed=k) for k in range(64)])
    env.reset()
    a = torch.randint(0, 2, (n,), device=env.device, dtype=torch.int32)
    print("cartpole copy=%-5s %.1f us per step() call" % (copy, wall(lambda: env.step(a))), flush=True)
    env.close()
    env = AcrobotVecEnv(n, seed=1, autoreset_mode="same_step", frameskip=1, copy=copy)
    env.set_task([sample_acrobot(seed=k) for k in range(64)])
    env.reset()
    a = torch.randint(0, 3, (n,), device=env.device, dtype=torch.int32)
    print("acrobot  copy=%-5s %.1f us per step() call" % (copy, wall(lambda: env.step(a))), flush=True)
    env.close()
