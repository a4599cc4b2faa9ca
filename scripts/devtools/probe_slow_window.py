"""After a second env is created in one process some ~80 ms of step launches run 30x slower (265 instead of 8.6 us):
host-side or device-side?  Bursts of 50 xv_linds_step launches, wall clock and the engine's HIP events per burst."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xenoverse_amd.linds import LinDSVecEnv
from xenoverse_amd import _lib
from xenoverse_amd.engine import AUTORESET
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_families import linds_tasks

n = 65536
tasks = linds_tasks(1024)
for rep in range(3):
    env = LinDSVecEnv(n, seed=1, autoreset_mode="same_step", copy=False)
    env.set_task(tasks)
    env.reset()
    a = env._action(torch.rand((n, 8), device=env.device) * 2 - 1)

    def raw():
        _lib.check(env.lib.xv_linds_step(env._h, _lib.ptr(a), _lib.ptr(env._obs), _lib.ptr(env._reward), _lib.ptr(env._term),
                                         _lib.ptr(env._trunc), _lib.ptr(env._cmd), _lib.ptr(env._error), _lib.ptr(env._fobs),
                                         AUTORESET["same_step"]))
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    rows = []
    for burst in range(40):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.engine.event_record(0)
        for _ in range(50):
            raw()
        t1 = time.perf_counter()
        env.engine.event_record(1)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rows.append(((t0 - t_start) * 1e3, (t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6, env.engine.event_elapsed_ms() / 50 * 1e3))
    slow = [r for r in rows if r[3] > 20 or r[2] > 20]
    print("env %d: bursts with >20 us per step: %d of 40" % (rep, len(slow)), flush=True)
    for r in (slow[:6] or rows[:3]):
        print("   at %.1f ms: host launch %.1f us, wall %.1f us, events %.1f us per step" % r, flush=True)
    env.close()
