"""Randomised soak of the LinDS step kernels against the CPU oracle (not collected by pytest; run as a script on a GPU box,
`PYTHONPATH=.:tests python tests/soak_linds.py [seconds]`): tasks drawn by the package's seed-compatible sampler with random
dimensions, random env -> task maps and counts (partial tiles, one-env tasks), both kernels, the three auto-reset modes, injected
draws — state and observation bit for bit, command / error / reward within 1e-5 as the unit tests."""
import sys
import time

import numpy as np

import oracle
from xenoverse_amd.linds import LinDSVecEnv, build_tables, pad_tables
from xenoverse_amd.linds.task_sampler import LinearDSSampler
from util import close_rel

MODES = {"disabled": 0, "next_step": 1, "same_step": 2}


def _np(t):
    return t.detach().cpu().numpy()


class Mismatch(Exception):
    pass


def _check(ok, what):
    if not ok:
        raise Mismatch(what)


def soak(rng, seed):
    n_task = int(rng.randint(1, 7))
    ns = int(rng.choice([rng.randint(2, 17), rng.randint(17, 33)]))
    same_dims = rng.random_sample() < 0.5
    tasks = []
    for k in range(n_task):
        d = ns if same_dims else int(rng.randint(2, ns + 1))
        t = LinearDSSampler(state_dim=d, action_dim=int(rng.randint(1, 9)), observation_dim=int(rng.randint(1, 17)), seed=seed + k)
        t["max_steps"] = int(rng.randint(5, 40))
        tasks.append(t)
    tab = pad_tables(build_tables(tasks))
    counts = [int(rng.choice([1, rng.randint(1, 20), 64, rng.randint(20, 200)])) for _ in range(n_task)]
    env_task = np.repeat(np.arange(n_task, dtype=np.int32), counts)
    layout = str(rng.choice(["grouped", "shuffled"]))
    if layout == "shuffled":
        rng.shuffle(env_task)
    n = len(env_task)
    mode = str(rng.choice(list(MODES)))
    path = str(rng.choice(["mfma", "mfma", "scalar"]))
    env = LinDSVecEnv(n, autoreset_mode=mode, seed=seed)
    env.set_task(tasks, env_task_index=env_task)
    env.set_path(path)
    ora = oracle.LinDSOracle(tab, env_task)
    n_init = tab["ints"][env_task, 2]
    idx0 = (rng.random_sample(n) * n_init).astype(np.int32)
    obs, info = env.reset_injected(idx0)
    o0 = ora.reset_injected(idx0)
    no = _np(obs).shape[1]
    _check(np.array_equal(_np(obs), o0["obs"][:, :no]), "reset obs")
    T = int(rng.randint(10, 60))
    na = tab["NA"]
    for t in range(T):
        a = rng.uniform(-1.4, 1.4, (n, na)).astype(np.float32)
        z = rng.standard_normal((tab["NS"], n)).astype(np.float32)
        idx = (rng.random_sample(n) * n_init).astype(np.int32)
        d = env.step_injected(a, z, idx)
        o = ora.step_injected(a, z, idx, MODES[mode])
        obs, r, term, trunc, info = d
        _check(np.array_equal(_np(term).astype(np.uint8), o["terminated"]), "terminated")
        _check(np.array_equal(_np(trunc).astype(np.uint8), o["truncated"]), "truncated")
        _check(np.array_equal(_np(obs), o["obs"][:, :no]), "obs")
        _check(close_rel(_np(info["command"]), o["cmd"][:, :no], 1e-5, 2e-6), "command")
        _check(close_rel(_np(info["error"]), o["error"], 1e-5, 2e-6), "error")
        _check(close_rel(_np(r), o["reward"], 1e-5, 2e-6), "reward")
        x, st, nr = env.get_state()
        _check(np.array_equal(_np(x), ora.x) and np.array_equal(_np(st), ora.steps), "state / steps")
        _check(np.array_equal(_np(nr), ora.need_reset), "need_reset")
        done = (o["terminated"] | o["truncated"]).astype(bool)
        if mode == "same_step" and done.any():
            _check(np.array_equal(_np(info["final_obs"])[done], o["final_obs"][done][:, :no]), "final_obs")
        if mode == "disabled" and done.any():
            env.reset_injected(idx, mask=done.astype(np.uint8))
            ora.reset_injected(idx, mask=done.astype(np.uint8))
    env.close()
    return "linds tasks=%d ns<=%d (NS %d NA %d NO %d) envs=%d %s path=%s mode=%s steps=%d" % (
        n_task, ns, tab["NS"], tab["NA"], tab["NO"], n, layout, path, mode, T)


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    t_end = time.time() + budget
    master = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    n = 0
    while time.time() < t_end:
        seed = int(master.randint(1, 1 << 30))
        try:
            line = soak(np.random.RandomState(seed), seed)
        except Mismatch as ex:
            print("MISMATCH with seed %d: %s" % (seed, ex), flush=True)
            sys.exit(1)
        n += 1
        print("ok seed=%d %s" % (seed, line), flush=True)
    print("TOTAL %d configurations, 0 mismatches" % n)
