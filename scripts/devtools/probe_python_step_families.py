"""Dev probe (not shipped): host cost of a Python-level step() per family, closed loop with a one-op policy, copy=True / False."""
import sys
import torch
sys.path.insert(0, "scripts")
from bench_families import timed, linds_tasks
from xenoverse_amd.linds import LinDSVecEnv
from xenoverse_amd.metacontrol import CartPoleVecEnv, AcrobotVecEnv, sample_cartpole, sample_acrobot

for copy in (True, False):
    env = LinDSVecEnv(65536, autoreset_mode="same_step", seed=1, copy=copy)
    env.set_task(linds_tasks(1024)); obs, _ = env.reset()
    st = {"o": obs}
    def it():
        st["o"] = env.step((st["o"][:, :8] * -0.3).clamp(-1, 1))[0]
    print("linds    copy=%s: %.1f us per [policy -> step]" % (copy, timed(it, 400, 20)), flush=True)
    env.close()
for copy in (True, False):
    env = CartPoleVecEnv(65536, frameskip=1, autoreset_mode="same_step", seed=1, copy=copy)
    env.set_task([sample_cartpole(seed=k) for k in range(1024)]); obs, _ = env.reset()
    st = {"o": obs}
    def it():
        st["o"] = env.step((st["o"][:, 2] > 0).to(torch.int32))[0]
    print("cartpole copy=%s: %.1f us per [policy -> step]" % (copy, timed(it, 400, 20)), flush=True)
    env.close()
for copy in (True, False):
    env = AcrobotVecEnv(65536, frameskip=1, autoreset_mode="same_step", seed=1, copy=copy)
    env.set_task([sample_acrobot(seed=k) for k in range(1024)]); obs, _ = env.reset()
    st = {"o": obs}
    def it():
        st["o"] = env.step((st["o"][:, 4] > 0).to(torch.int32))[0]
    print("acrobot  copy=%s: %.1f us per [policy -> step]" % (copy, timed(it, 400, 20)), flush=True)
    env.close()
