import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_linds import *
from test_gpu_linds import _np
files = FILES[:4]
tasks = []
for f in files:
    t = load_linds_golden(f)[1]; t["max_steps"] = 23; tasks.append(t)
env_task = np.repeat(np.arange(len(tasks), dtype=np.int32), 64)
n, T = len(env_task), 70
acts = np.random.RandomState(3).uniform(-1.3, 1.3, (T, n, 8)).astype(np.float32)
recs = []
for fused in (False, True):
    env = LinDSVecEnv(n, autoreset_mode="same_step", seed=77, env_id_base=123)
    env.set_task(tasks, env_task_index=env_task)
    env.reset()
    if fused:
        a = env.rollout(acts)
        rec = {k: _np(a[k]) for k in a}
    else:
        rows = []
        for t in range(T):
            o, r, te, tr, info = env.step(acts[t])
            rows.append(dict(obs=_np(o), reward=_np(r), terminated=_np(te).astype(np.uint8), truncated=_np(tr).astype(np.uint8),
                             command=_np(info["command"]), error=_np(info["error"]), final_obs=_np(info["final_obs"])))
        rec = {k: np.stack([row[k] for row in rows]) for k in rows[0]}
    recs.append(rec); env.close()
for k in recs[0]:
    a, b = recs[0][k], recs[1][k]
    if a.ndim == 3: b = b[..., :a.shape[-1]]
    bad = np.argwhere(a != b)
    print(k, len(bad), bad[:3].tolist())
    if len(bad):
        i = tuple(bad[0]); print("   ", a[i], b[i])
k = "error"; bad = np.argwhere(recs[0][k] != recs[1][k])
if len(bad):
    t, e = bad[0]
    for kk in ("error", "reward", "terminated", "truncated"):
        print(kk, recs[0][kk][t, e], recs[1][kk][t, e])
    print("obs", recs[0]["obs"][t, e][:8], recs[1]["obs"][t, e][:8])
print("---- carry check ----")
outs = []
for split in (True, False):
    env = LinDSVecEnv(n, autoreset_mode="same_step", seed=77, env_id_base=123)
    env.set_task(tasks, env_task_index=env_task)
    env.reset()
    if split:
        a = env.rollout(acts[:1]); b = env.rollout(acts[1:2])
        obs = np.concatenate([_np(a["obs"]), _np(b["obs"])])
    else:
        obs = _np(env.rollout(acts[:2])["obs"])
    x = _np(env.get_state()[0])
    outs.append((obs, x)); env.close()
d = np.argwhere(outs[0][0] != outs[1][0]); print("obs diff split vs fused", len(d), d[:6].tolist())
d = np.argwhere(outs[0][1] != outs[1][1]); print("x diff", len(d), d[:6].tolist())
d = np.argwhere(outs[0][0] != recs[0]["obs"][:2]) if outs[0][0].shape == recs[0]["obs"][:2].shape else np.argwhere(outs[0][0][..., :8] != recs[0]["obs"][:2])
print("split rollout vs steps", len(d), d[:6].tolist())
