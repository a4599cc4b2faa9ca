import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle
from xenoverse_amd.linds import LinDSVecEnv, build_tables, pad_tables
from util import golden_files, load_linds_golden
FILES = golden_files("linds_")
tasks = [load_linds_golden(p)[1] for p in FILES[:4]]
for t in tasks: t["max_steps"] = min(int(t["max_steps"]), 40)
tab = pad_tables(build_tables(tasks))
env_task = np.repeat(np.arange(4, dtype=np.int32), 32)
n = len(env_task)
rng = np.random.RandomState(3)
env = LinDSVecEnv(n, autoreset_mode="disabled"); env.set_task(tasks, env_task_index=env_task)
ora = oracle.LinDSOracle(tab, env_task)
n_init = tab["ints"][env_task, 2]
idx0 = (rng.random_sample(n) * n_init).astype(np.int32)
env.reset_injected(idx0); ora.reset_injected(idx0)
for t in range(90):
    a = rng.uniform(-1.4, 1.4, (n, 8)).astype(np.float32)
    z = rng.standard_normal((tab["NS"], n)).astype(np.float32)
    idx = (rng.random_sample(n) * n_init).astype(np.int32)
    obs, r, term, trunc, info = env.step_injected(a, z, idx)
    o = ora.step_injected(a, z, idx, 0)
    x = env.get_state()[0].cpu().numpy()
    bad = np.nonzero((term.cpu().numpy().astype(np.uint8) != o["terminated"]) | (np.abs(x - ora.x).max(0) > 0) |
                     (np.abs(obs.cpu().numpy() - o["obs"]).max(1) > 0))[0]
    if len(bad):
        e = bad[0]
        print("step", t, "bad envs", bad[:10], "task", env_task[e])
        print("dev term", int(term[e]), "ora", o["terminated"][e], "err dev", float(info["error"][e]), "ora", o["error"][e])
        print("x diff", np.abs(x[:, e] - ora.x[:, e]).max(), "obs diff", np.abs(obs.cpu().numpy()[e] - o["obs"][e]).max())
        print("obs dev", obs.cpu().numpy()[e][:8]); print("obs ora", o["obs"][e][:8])
        print("cmd dev", info["command"].cpu().numpy()[e][:8]); print("cmd ora", o["cmd"][e][:8])
        break
    m = (o["terminated"] | o["truncated"]).astype(np.uint8)
    if m.any():
        env.reset_injected(idx, mask=m); ora.reset_injected(idx, mask=m)
else:
    print("no mismatch")
