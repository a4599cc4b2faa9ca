import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from xenoverse_amd.linds import LinDSVecEnv, LinearDSSampler
from xenoverse_amd import _lib
from xenoverse_amd.engine import AUTORESET
n_task, per = 1024, 64; n = n_task*per
base = []
for k in range(64):
    t = LinearDSSampler(32, 8, 8, seed=k); t["max_steps"] = 500; base.append(t)
def run(label, static, inject):
    tasks = []
    for k in range(n_task):
        t = dict(base[k % 64])
        if static:
            t["target_type"] = "static_target"; t["command"] = np.zeros(8); t["target_delay"] = 0
        tasks.append(t)
    env = LinDSVecEnv(n, autoreset_mode="same_step", seed=1); env.set_task(tasks); env.set_path("mfma"); env.reset()
    a = torch.rand((n, 8), device=env.device)*2-1
    z = torch.randn((32, n), device=env.device); idx = torch.zeros(n, dtype=torch.int32, device=env.device)
    def step():
        if inject:
            _lib.check(env.lib.xv_linds_step_injected(env._h, _lib.ptr(a), _lib.ptr(z), _lib.ptr(idx), _lib.ptr(env._obs), _lib.ptr(env._reward), _lib.ptr(env._term), _lib.ptr(env._trunc), _lib.ptr(env._cmd), _lib.ptr(env._error), _lib.ptr(env._fobs), 2))
        else:
            _lib.check(env.lib.xv_linds_step(env._h, _lib.ptr(a), _lib.ptr(env._obs), _lib.ptr(env._reward), _lib.ptr(env._term), _lib.ptr(env._trunc), _lib.ptr(env._cmd), _lib.ptr(env._error), _lib.ptr(env._fobs), 2))
    for _ in range(30): step()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): step()
    e1.record(); torch.cuda.synchronize()
    print("%-28s %.2f us/step" % (label, e0.elapsed_time(e1)*1e3/300)); env.close()
run("fourier + philox", False, False)
run("static  + philox", True, False)
run("fourier + injected noise", False, True)
run("static  + injected noise", True, True)
