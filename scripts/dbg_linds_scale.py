import sys, numpy as np, torch
sys.path.insert(0, '.')
from xenoverse_amd.linds import LinDSVecEnv, LinearDSSampler
from xenoverse_amd import _lib
base = []
for k in range(64):
    t = LinearDSSampler(32, 8, 8, seed=k); t["max_steps"] = 500
    t["target_type"] = "static_target"; t["command"] = np.zeros(8); t["target_delay"] = 0
    base.append(t)
for n_task in (128, 256, 512, 1024, 2048, 4096):
    n = n_task*64
    env = LinDSVecEnv(n, autoreset_mode="same_step", seed=1); env.set_task([base[k % 64] for k in range(n_task)]); env.set_path("mfma"); env.reset()
    a = torch.rand((n, 8), device=env.device)*2-1
    z = torch.randn((32, n), device=env.device); idx = torch.zeros(n, dtype=torch.int32, device=env.device)
    def step():
        _lib.check(env.lib.xv_linds_step_injected(env._h, _lib.ptr(a), _lib.ptr(z), _lib.ptr(idx), _lib.ptr(env._obs), _lib.ptr(env._reward), _lib.ptr(env._term), _lib.ptr(env._trunc), _lib.ptr(env._cmd), _lib.ptr(env._error), _lib.ptr(env._fobs), 2))
    for _ in range(30): step()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): step()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)*1e3/300
    print("n_env %7d  %.2f us/step  %.2e env-steps/s" % (n, us, n/us*1e6)); env.close()
