"""How much of the LinDS step time is the tail of waves that take the reset branch?  Steps 65,536 envs of config 3 with
max_steps 500 (3 % of the waves restart an env per step) and with max_steps 10^6 (none do), and counts terminations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xenoverse_amd import _lib
from xenoverse_amd.engine import AUTORESET
from xenoverse_amd.linds import LinDSVecEnv, LinearDSSampler

def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps

n_task, per = 1024, 64
n = n_task * per
base = [LinearDSSampler(32, 8, 8, seed=k) for k in range(64)]
for ms, scale in ((500, 1.0), (1000000, 1.0), (1000000, 0.0)):
    tasks = []
    for k in range(n_task):
        t = dict(base[k % 64]); t["max_steps"] = ms
        tasks.append(t)
    env = LinDSVecEnv(n, autoreset_mode="same_step", seed=1)
    env.set_task(tasks)
    env.reset()
    a = (torch.rand((n, 8), device=env.device) * 2 - 1) * scale
    ends = torch.zeros((), device=env.device)
    def step():
        _lib.check(env.lib.xv_linds_step(env._h, _lib.ptr(a), _lib.ptr(env._obs), _lib.ptr(env._reward),
                                         _lib.ptr(env._term), _lib.ptr(env._trunc), _lib.ptr(env._cmd),
                                         _lib.ptr(env._error), _lib.ptr(env._fobs), AUTORESET["same_step"]))
    us = timed(step, 400, 600)
    cnt = 0
    for _ in range(50):
        step(); cnt += int((env._term | env._trunc).sum())
    print("max_steps", ms, "action scale", scale, "us/step %.2f" % us, "episode ends per step %.1f" % (cnt / 50), flush=True)
    env.close()
