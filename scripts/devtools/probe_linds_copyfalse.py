import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xenoverse_amd.linds import LinDSVecEnv
from xenoverse_amd import _lib
from xenoverse_amd.engine import AUTORESET
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
for copy in (False, True, False):
    env = LinDSVecEnv(n, seed=1, autoreset_mode="same_step", copy=copy)
    env.set_task(linds_tasks(1024))
    env.reset()
    a = torch.rand((n, 8), device=env.device) * 2 - 1
    print("linds copy=%-5s step() %.1f us" % (copy, wall(lambda: env.step(a))), flush=True)
    aa = env._action(a)
    def raw():
        _lib.check(env.lib.xv_linds_step(env._h, _lib.ptr(aa), _lib.ptr(env._obs), _lib.ptr(env._reward), _lib.ptr(env._term),
                                         _lib.ptr(env._trunc), _lib.ptr(env._cmd), _lib.ptr(env._error), _lib.ptr(env._fobs),
                                         AUTORESET["same_step"]))
    print("   raw xv_linds_step, same buffers: %.1f us" % wall(raw), flush=True)
    print("   _steps_now: %.1f us" % wall(lambda: env._steps_now()), flush=True)
    print("   _infos: %.1f us" % wall(lambda: env._infos(True, fresh=True)), flush=True)
    print("   _ret: %.1f us" % wall(lambda: env._ret()), flush=True)
    env.close()
