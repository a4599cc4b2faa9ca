"""AnyMDP step on tasks of the REFERENCE's distribution (device sampler, banded sparse rows with zero-probability
next states) instead of the survey's synthetic dense bands: fence vs bucket search, 65,536 envs = 1,024 tasks x 64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xenoverse_amd.anymdp import AnyMDPVecEnv
from xenoverse_amd.anymdp import device_sampler as ds
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_families import timed

n_task, per, S, A = 1024, 64, 64, 8
n = n_task * per
t = ds.sample_tasks_device(n_task, S, A, seed=3, batch=4096)
print("sampled", t["stats"], flush=True)
env = AnyMDPVecEnv(n, seed=1, autoreset_mode="same_step")
env.set_task({k: t[k] for k in ("S", "A", "s0_max", "rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")},
             env_task_index=(torch.arange(n, device=env.device, dtype=torch.int32) // per).contiguous())
env.reset()
P = 32
acts = torch.randint(0, A, (P, n), device=env.device, dtype=torch.int32)
ring = env.step_many(P, acts)
for search in ("fence", "bucket", "bucket32", "bucket64"):
    nb = 16 if search == "bucket" else int(search[6:] or 16) if search.startswith("bucket") else 0
    if search.startswith("bucket"):
        env.set_search("bucket", n_bucket=nb)
    else:
        env.set_search(search)
    us = [timed(lambda: env.step_many(P, acts, out=ring), 30, 5) / P for _ in range(3)]
    print("%-9s %s us per step" % (search, ["%.2f" % u for u in us]), flush=True)
env.close()

# the multi-token / POMDP step on reference-distribution tasks (transition rows as above, sparse observation rows)
for tt, do, da in (("POMDP", 1, 1), ("MTPOMDP", 2, 2)):
    t = ds.sample_tasks_device(n_task, S, A, seed=3, batch=4096, task_type=tt, observation_space=64, observation_tokens=do,
                               action_tokens=da)
    env = AnyMDPVecEnv(n, seed=1, autoreset_mode="same_step")
    env.set_task(t, env_task_index=(torch.arange(n, device=env.device, dtype=torch.int32) // per).contiguous())
    env.reset()
    a = torch.randint(0, A, (n, da) if tt == "MTPOMDP" else (n,), device=env.device, dtype=torch.int32)
    for search in ("fence", "bucket"):
        if search == "bucket":
            env.set_search("bucket", n_bucket=16)
        else:
            env.set_search(search)
        us = [timed(lambda: env.step(a), 60, 10) for _ in range(3)]
        print("%-8s d_obs %d d_act %d %-7s %s us per step (python step() loop)" % (tt, do, da, search, ["%.2f" % u for u in us]), flush=True)
        from xenoverse_amd import _lib
        from xenoverse_amd.engine import AUTORESET
        aa = env._tok_action(a)

        def raw():
            _lib.check(env.lib.xv_anymdp_step_tokens(env._h, _lib.ptr(aa), _lib.ptr(env._tobs), _lib.ptr(env._reward),
                                                     _lib.ptr(env._reward_gt), _lib.ptr(env._term), _lib.ptr(env._trunc),
                                                     _lib.ptr(env._tfobs), AUTORESET["same_step"]))
        us = [timed(raw, 200, 20) for _ in range(3)]
        print("%-8s d_obs %d d_act %d %-7s %s us per step (xv_anymdp_step_tokens from a Python loop)" % (tt, do, da, search, ["%.2f" % u for u in us]), flush=True)
    env.close()
