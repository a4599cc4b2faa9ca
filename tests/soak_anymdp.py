"""Randomised soak of the AnyMDP step kernels against the CPU oracle (not collected by pytest: run as a script on a GPU box,
`PYTHONPATH=.:tests python tests/soak_anymdp.py [seconds]`).  The unit tests pin chosen shapes; this draws shapes, env -> task
maps, bucket counts, searches and auto-reset modes at random and compares every output of every step, bit for bit where the
unit tests do.  One line per configuration; exit code 1 on the first mismatch (the configuration's seed is in the line)."""
import sys
import time

import numpy as np
import torch

import oracle
from xenoverse_amd import _lib
from xenoverse_amd.anymdp import AnyMDPVecEnv, to_blocked
from util import close_f32

MODES = {"disabled": 0, "next_step": 1, "same_step": 2}


def _np(t):
    return t.detach().cpu().numpy()


def _dev_tables(tab, dev="cuda:0"):
    out = dict(S=tab["S"], A=tab["A"], s0_max=tab["s0_max"])
    tab = dict(tab, rows=to_blocked(tab["cdf"], tab["rs"]))
    for k in ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps"):
        v = np.ascontiguousarray(tab[k])
        out[k] = torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev)
    return out


class Mismatch(Exception):
    pass


def _check(ok, what):
    if not ok:
        raise Mismatch(what)


def _compare(dev_out, ora_out, exact_reward, same_step):
    obs, r, term, trunc, info = dev_out
    o_obs, o_r, o_rgt, o_term, o_trunc, o_fobs = ora_out
    _check(np.array_equal(_np(obs), o_obs), "obs")
    _check(np.array_equal(_np(term).astype(np.uint8), o_term), "terminated")
    _check(np.array_equal(_np(trunc).astype(np.uint8), o_trunc), "truncated")
    _check(np.array_equal(_np(info["reward_gt"]), o_rgt), "reward_gt")
    _check(np.array_equal(_np(r), o_r) if exact_reward else close_f32(_np(r), o_r, rel=1e-5, abs_=2e-6), "reward")
    if same_step and "final_obs" in info:
        _check(np.array_equal(_np(info["final_obs"]), o_fobs), "final_obs")


def soak_mdp(rng, seed):
    S = int(rng.choice([rng.randint(3, 65), rng.randint(65, 130), rng.randint(130, 321)]))
    A = int(rng.randint(2, 9))
    n_task = int(rng.randint(1, 10))
    n_env = int(rng.choice([rng.randint(1, 70), rng.randint(70, 700), rng.randint(700, 4000)]))
    search = str(rng.choice(["fence", "bucket", "binary", "auto"]))
    nb = int(rng.choice([16, 32, 64]))
    mode = str(rng.choice(list(MODES)))
    big_obs = rng.random_sample() < 0.2
    tab = oracle.anymdp_synth(seed=seed, task_index_base=int(rng.randint(0, 1000)), n_task=n_task, S=S, A=A,
                              s0_max=int(rng.randint(1, min(4, S) + 1)))
    if big_obs:
        tab["state_map"] = tab["state_map"] * np.int32(101) + np.int32(300)
    tab["max_steps"][:] = rng.randint(3, 40, n_task)          # truncations happen
    env_task = rng.randint(0, n_task, n_env).astype(np.int32)
    env = AnyMDPVecEnv(n_env, autoreset_mode=mode, seed=seed, env_id_base=int(rng.randint(0, 1 << 20)))
    env.set_task(_dev_tables(tab), env_task_index=env_task)
    if search in ("bucket", "auto"):
        env.set_search(search, n_bucket=nb)
    else:
        env.set_search(search)
    ora = oracle.AnyMDPOracle(tab, env_task)
    tick = env.engine.tick
    obs, _ = env.reset()
    _check(np.array_equal(_np(obs), ora.reset(seed, env.engine.env_id_base, tick)), "reset obs")
    T = int(rng.randint(10, 50))
    for t in range(T):
        a = rng.randint(0, A, n_env).astype(np.int32)
        if t % 2 == 0:
            tick = env.engine.tick
            d = env.step(a)
            o = ora.step(seed, env.engine.env_id_base, tick, a, MODES[mode])
            _compare(d, o, False, mode == "same_step")
        else:
            u, z, ur = rng.random_sample(n_env), rng.standard_normal(n_env).astype(np.float32), rng.random_sample(n_env)
            k = rng.randint(0, n_env, 8)                      # exact CDF entries select the NEXT state
            rows = tab["cdf"][env_task[k], ora.state[k], a[k]]
            u[k] = np.minimum(rows[np.arange(8), rng.randint(0, S, 8)], np.nextafter(1.0, 0.0))
            d = env.step_injected(a, u, z, ur)
            o = ora.step_injected(a, u, z, ur, MODES[mode])
            _compare(d, o, True, mode == "same_step")
        s, st, nr = env.get_state()
        _check(np.array_equal(_np(s), ora.state) and np.array_equal(_np(st), ora.steps), "state / steps")
        if mode == "disabled" and o[3].any():
            ur2 = rng.random_sample(n_env)
            env.reset_injected(ur2, mask=o[3]); ora.reset_injected(ur2, mask=o[3])
    many = ""
    if mode != "disabled":      # a burst issued from C (step_many): plain, graph replay, overlapped launches, K chains
        P = int(rng.choice([2, 4, 6, 7, 8]))
        n_steps = int(rng.choice([rng.randint(1, 3 * P + 1), rng.randint(64, 100)]))
        how = str(rng.choice(["plain", "graph", "overlap", "chains"]))
        acts = rng.randint(0, A, (P, n_env)).astype(np.int32)
        env.set_step_many_graph(how != "plain")
        env.set_step_many_overlap(how == "overlap")
        chains = 1
        if how == "chains":
            chains = int(rng.choice([k for k in (2, 4, 8) if n_env % k == 0] or [1]))
        tick0 = env.engine.tick
        ring = env.step_many(n_steps, torch.as_tensor(acts, device=env.device), chains=chains,
                             how=str(rng.choice(["streams", "graph"])))
        torch.cuda.synchronize()
        last = {}
        for k in range(n_steps):
            last[k % P] = ora.step(seed, env.engine.env_id_base, tick0 + k, acts[k % P], MODES[mode])
        for slot, o in last.items():
            _check(np.array_equal(_np(ring["obs"][slot]), o[0]) and np.array_equal(_np(ring["terminated"][slot]), o[3]) and
                   np.array_equal(_np(ring["truncated"][slot]), o[4]), "step_many ring (%s)" % how)
        s, st, nr = env.get_state()
        _check(np.array_equal(_np(s), ora.state) and np.array_equal(_np(st), ora.steps), "state / steps after step_many (%s)" % how)
        _check(env.engine.tick == tick0 + n_steps, "tick after step_many")
        many = " many=%s/P%d/n%d/K%d/ov%d" % (how, P, n_steps, chains, env.step_many_overlap_state)
    flags = env.check_errors() if mode != "disabled" else 0
    eff = env.effective_search
    env.close()
    return "mdp S=%d A=%d tasks=%d envs=%d search=%s->%s nb=%d mode=%s big_obs=%d steps=%d flags=%d%s" % (
        S, A, n_task, n_env, search, eff, nb, mode, big_obs, T, flags, many)


def soak_tok(rng, seed):
    S = int(rng.choice([rng.randint(4, 65), rng.randint(65, 301)]))
    A = int(rng.randint(2, 7))
    n_task = int(rng.randint(1, 6))
    n_obs = int(rng.choice([rng.randint(2, 16), rng.randint(16, 257), rng.randint(257, 400)]))
    d_obs, d_act = int(rng.randint(1, 4)), int(rng.randint(1, 4))
    n_env = int(rng.choice([rng.randint(1, 70), rng.randint(70, 1500)]))
    search = str(rng.choice(["fence", "bucket", "bucket", "binary"]))
    nb = int(rng.choice([16, 32, 64]))
    mode = str(rng.choice(list(MODES)))
    sparse = rng.random_sample() < 0.5
    tab = oracle.anymdp_synth(seed=seed, task_index_base=int(rng.randint(0, 1000)), n_task=n_task, S=S, A=A, s0_max=3)
    tab["max_steps"][:] = rng.randint(3, 30, n_task)
    w = rng.random_sample((n_task, d_obs, S, n_obs)) * (rng.random_sample((n_task, d_obs, S, n_obs)) < (0.08 if sparse else 0.6))
    w[..., 0] += (w.sum(-1) == 0)
    obs_cdf = np.cumsum(w, -1)
    obs_cdf = obs_cdf / obs_cdf[..., -1:]
    env_task = rng.randint(0, n_task, n_env).astype(np.int32)
    env = AnyMDPVecEnv(n_env, autoreset_mode=mode, seed=seed)
    env.set_task(_dev_tables(tab), env_task_index=env_task)
    oc = torch.from_numpy(np.ascontiguousarray(obs_cdf)).cuda()
    _lib.check(env.lib.xv_anymdp_set_observation_model(env._h, n_obs, d_obs, d_act, _lib.ptr(oc)))
    env._tok = (d_obs, d_act); env.task_type = "MTPOMDP"
    env._tobs = torch.zeros((n_env, d_obs), dtype=torch.int32, device="cuda")
    env._tfobs = torch.full((n_env, d_obs), -1, dtype=torch.int32, device="cuda")
    if search == "bucket":
        env.set_search(search, n_bucket=nb)
    else:
        env.set_search(search)
    ora = oracle.AnyMDPTokOracle(tab, env_task, obs_cdf, d_act)
    ur0, uo0 = rng.random_sample(n_env), rng.random_sample((d_obs, n_env))
    _check(np.array_equal(_np(env.reset_tokens_injected(ur0, uo0)), ora.tok_reset_injected(ur0, uo0)), "reset obs")
    T = int(rng.randint(8, 40))
    for t in range(T):
        a = rng.randint(0, A, (n_env, d_act)).astype(np.int32)
        u, z = rng.random_sample((d_act, n_env)), rng.standard_normal((d_act, n_env)).astype(np.float32)
        uo, ur, uor = rng.random_sample((d_obs, n_env)), rng.random_sample(n_env), rng.random_sample((d_obs, n_env))
        k = rng.randint(0, n_env, 6)
        uo[0, k] = np.minimum(obs_cdf[env_task[k], 0, ora.state[k], rng.randint(0, n_obs, 6)], np.nextafter(1.0, 0))
        obs, r, term, trunc, info = env.step_tokens_injected(a, u, z, uo, ur, uor)
        o = ora.tok_step_injected(a, u, z, uo, ur, uor, MODES[mode])
        _check(np.array_equal(_np(obs), o[0]), "obs")
        _check(np.array_equal(_np(r), o[1]), "reward")
        _check(np.array_equal(_np(info["reward_gt"]), o[2]), "reward_gt")
        _check(np.array_equal(_np(term).astype(np.uint8), o[3]) and np.array_equal(_np(trunc).astype(np.uint8), o[4]), "flags")
        if mode == "same_step":
            _check(np.array_equal(_np(info["final_obs"]), o[5]), "final_obs")
        s, st, nr = env.get_state()
        _check(np.array_equal(_np(s), ora.state) and np.array_equal(_np(st), ora.steps) and
               np.array_equal(_np(nr), ora.need_reset), "state / steps / need_reset")
        if mode == "disabled" and o[3].any():
            ur2, uor2 = rng.random_sample(n_env), rng.random_sample((d_obs, n_env))
            env.reset_tokens_injected(ur2, uor2, mask=o[3]); ora.tok_reset_injected(ur2, uor2, mask=o[3])
    kern = env.token_kernel
    env.close()
    return "tok S=%d A=%d tasks=%d n_obs=%d d_obs=%d d_act=%d envs=%d search=%s (%s) nb=%d mode=%s sparse=%d steps=%d" % (
        S, A, n_task, n_obs, d_obs, d_act, n_env, search, kern, nb, mode, sparse, T)


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    t_end = time.time() + budget
    master = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
    n = 0
    while time.time() < t_end:
        seed = int(master.randint(1, 1 << 30))
        rng = np.random.RandomState(seed)
        fn = soak_tok if n % 3 == 2 else soak_mdp
        try:
            line = fn(rng, seed)
        except Mismatch as ex:
            print("MISMATCH in %s with seed %d: %s" % (fn.__name__, seed, ex), flush=True)
            sys.exit(1)
        n += 1
        print("ok seed=%d %s" % (seed, line), flush=True)
    print("TOTAL %d configurations, 0 mismatches" % n)
