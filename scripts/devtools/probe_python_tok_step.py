"""Dev probe (not shipped): what a Python-level step() of a POMDP / multi-token env costs per call (closed loop: the action of
step k + 1 is an elementwise function of the observation of step k), reference-distribution tasks, 65,536 envs."""
import sys
import torch
sys.path.insert(0, "scripts")
from bench_families import timed
from xenoverse_amd.anymdp import AnyMDPVecEnv
from xenoverse_amd.anymdp import device_sampler as ds

S, A, n_task, per = 64, 8, 1024, 64
n = n_task * per
for tt, do, da in (("POMDP", 1, 1), ("MTPOMDP", 2, 2)):
    t = ds.sample_tasks_device(n_task, S, A, seed=3, batch=4096, task_type=tt, observation_space=64, observation_tokens=do,
                               action_tokens=da)
    for copy in (True, False):
        env = AnyMDPVecEnv(n, seed=1, autoreset_mode="same_step", copy=copy)
        env.set_task(t, env_task_index=(torch.arange(n, device=env.device, dtype=torch.int32) // per).contiguous())
        obs, _ = env.reset()
        st = {"o": obs}

        def it():
            o = st["o"]
            a = (o % A).to(torch.int32)
            if da > 1:
                a = a.reshape(n, -1)[:, :1].expand(n, da).contiguous()
            st["o"] = env.step(a)[0]
        us = timed(it, 400, 20)
        print("%s (%d, %d) copy=%s kernel=%s: %.1f us per [policy -> step]" % (tt, da, do, copy, env.token_kernel, us), flush=True)
        env.close()
