"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

usage:  python oracle/gen_golden.py [anymdp] [linds] [maze] [cartpole] ...   (default: all available)

Each fixture holds inputs and the reference's outputs for one family.  The
reference is stochastic through numpy's *global* legacy RandomState and reseeds it from OS entropy at
every reset (anymdp_env.py:87), so every recorded call is preceded by `numpy.random.seed(seed_j)`; the
uniform / normal numbers the call consumed are then replayed from `RandomState(seed_j)` and stored next to
the outputs.  A restatement is correct iff, given those numbers, it reproduces the outputs.

Nothing from /root/reference is copied: fixtures are arrays (task tensors sampled by the reference's
sampler, actions, random numbers, results).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402
import sample_ref_tasks  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _task_arrays(task):
    out = dict(ns=np.int64(task["ns"]), na=np.int64(task["na"]), max_steps=np.float64(task["max_steps"]),
               state_mapping=np.asarray(task["state_mapping"], np.int64),
               s_0=np.atleast_1d(np.asarray(task["s_0"], np.int64)),
               s_0_prob=np.atleast_1d(np.asarray(task["s_0_prob"], np.float64)),
               s_e=np.asarray(task["s_e"], np.int64).reshape(-1),
               transition=np.asarray(task["transition"], np.float64),
               reward=np.asarray(task["reward"], np.float64),
               reward_noise=np.asarray(task["reward_noise"], np.float64))
    return out


def gen_anymdp_one(name, task, n_tuples=4096, n_traj=1536, seed0=1000):
    """G-A for one task: single_step tuples, reset draws, a full step()/reset() trajectory that crosses
    the max_steps truncation boundary, and transition_gt rows."""
    AnyMDPEnv, _ = _refimport.anymdp()
    import xenoverse.anymdp.anymdp_env as envmod

    env = AnyMDPEnv(max_steps=5000)
    env.set_task(task)
    n = len(task["state_mapping"])
    na = int(task["na"])
    s_e = set(int(x) for x in np.asarray(task["s_e"]).reshape(-1))
    non_term = np.array([s for s in range(n) if s not in s_e], dtype=np.int64)
    rng = np.random.RandomState(seed0)

    # the reference reseeds from OS entropy inside reset(): inject the seed at that very call site
    inject = {"seed": 0}
    envmod.pseudo_random_seed = lambda *a, **k: inject["seed"]

    def ref_reset(seed):
        inject["seed"] = int(seed)
        obs, info = env.reset()
        return obs, info

    # ---- reset draws -------------------------------------------------------------------------
    n_reset = 256
    reset_seed = np.arange(seed0 + 500000, seed0 + 500000 + n_reset, dtype=np.int64)
    reset_u = np.zeros(n_reset)
    reset_state = np.zeros(n_reset, np.int64)
    reset_obs = np.zeros(n_reset, np.int64)
    for j in range(n_reset):
        obs, info = ref_reset(reset_seed[j])
        assert info["steps"] == 0
        reset_u[j] = np.random.RandomState(int(reset_seed[j])).random_sample()
        reset_state[j] = env._state
        reset_obs[j] = obs

    # ---- single_step tuples ------------------------------------------------------------------
    ref_reset(1)
    ss_s = non_term[rng.randint(0, len(non_term), n_tuples)]
    ss_a = rng.randint(0, na, n_tuples).astype(np.int64)
    ss_seed = np.arange(seed0, seed0 + n_tuples, dtype=np.int64)
    ss_u = np.zeros(n_tuples)
    ss_z = np.zeros(n_tuples)
    ss_next = np.zeros(n_tuples, np.int64)
    ss_rgt = np.zeros(n_tuples)
    ss_r = np.zeros(n_tuples)
    ss_term = np.zeros(n_tuples, np.uint8)
    for j in range(n_tuples):
        env._state = int(ss_s[j])
        np.random.seed(int(ss_seed[j]))
        rgt, r, term = env.single_step(int(ss_a[j]))
        rs = np.random.RandomState(int(ss_seed[j]))
        ss_u[j] = rs.random_sample()
        ss_z[j] = rs.standard_normal()
        ss_next[j] = env._state
        ss_rgt[j] = rgt
        ss_r[j] = r
        ss_term[j] = term

    # ---- full trajectory through step()/reset(), manual reset on done (how the reference is driven,
    #      anymdp/test_utils.py:42-60), forced across the truncation boundary once ---------------
    tr_a = rng.randint(0, na, n_traj).astype(np.int64)
    tr_seed = np.arange(seed0 + 100000, seed0 + 100000 + n_traj, dtype=np.int64)
    tr_reset_seed = np.arange(seed0 + 200000, seed0 + 200000 + n_traj, dtype=np.int64)
    tr_u = np.zeros(n_traj); tr_z = np.zeros(n_traj); tr_ur = np.zeros(n_traj)
    tr_obs = np.zeros(n_traj, np.int64); tr_r = np.zeros(n_traj); tr_rgt = np.zeros(n_traj)
    tr_term = np.zeros(n_traj, np.uint8); tr_trunc = np.zeros(n_traj, np.uint8)
    tr_steps = np.zeros(n_traj, np.int64); tr_state = np.zeros(n_traj, np.int64)
    tr_reset_obs = np.full(n_traj, -1, np.int64)
    tr_tgt = np.zeros((n_traj, int(task["ns"])))
    tr_set_steps = np.full(n_traj, -1, np.int64)  # steps counter forced BEFORE step t (-1: untouched)
    obs0, _ = ref_reset(seed0 + 300000)
    init_u = np.random.RandomState(seed0 + 300000).random_sample()
    init_state = int(env._state)
    jump_at = n_traj // 3
    for t in range(n_traj):
        if t == jump_at:  # place the episode 3 steps before truncation
            env.steps = int(np.ceil(float(task["max_steps"]))) - 3
            tr_set_steps[t] = env.steps
        np.random.seed(int(tr_seed[t]))
        obs, r, term, trunc, info = env.step(int(tr_a[t]))
        rs = np.random.RandomState(int(tr_seed[t]))
        tr_u[t] = rs.random_sample(); tr_z[t] = rs.standard_normal()
        tr_obs[t] = obs; tr_r[t] = r; tr_rgt[t] = info["reward_gt"]
        tr_term[t] = term; tr_trunc[t] = trunc; tr_steps[t] = info["steps"]; tr_state[t] = env._state
        tr_tgt[t] = info["transition_gt"]
        tr_ur[t] = np.random.RandomState(int(tr_reset_seed[t])).random_sample()
        if term or trunc:
            o, _ = ref_reset(tr_reset_seed[t])
            tr_reset_obs[t] = o

    out = _task_arrays(task)
    out.update(reset_u=reset_u, reset_state=reset_state, reset_obs=reset_obs,
               ss_s=ss_s, ss_a=ss_a, ss_u=ss_u, ss_z=ss_z, ss_next=ss_next, ss_rgt=ss_rgt, ss_r=ss_r,
               ss_term=ss_term,
               tr_a=tr_a, tr_u=tr_u, tr_z=tr_z, tr_ur=tr_ur, tr_obs=tr_obs, tr_r=tr_r, tr_rgt=tr_rgt,
               tr_term=tr_term, tr_trunc=tr_trunc, tr_steps=tr_steps, tr_state=tr_state,
               tr_reset_obs=tr_reset_obs, tr_tgt=tr_tgt[:256], tr_set_steps=tr_set_steps,
               init_u=np.float64(init_u), init_state=np.int64(init_state), init_obs=np.int64(obs0))
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;",
          "episodes ended:", int((tr_term | tr_trunc).sum()), "truncations:", int(tr_trunc.sum()))


def gen_anymdp():
    for seed in range(4):
        task = sample_ref_tasks.get(16, 4, seed)
        gen_anymdp_one("anymdp_16x4_seed%d" % seed, task, seed0=1000 + 7919 * seed)
    p = sample_ref_tasks.cache_path(64, 8, 1)
    if os.path.exists(p):
        gen_anymdp_one("anymdp_64x8_seed1", sample_ref_tasks.get(64, 8, 1), seed0=77000)
    else:
        print("skip 64x8: task cache not ready (run oracle/sample_ref_tasks.py 64 8 1; ~9 min)")


FAMILIES = {"anymdp": gen_anymdp}

if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or list(FAMILIES)
    for f in which:
        FAMILIES[f]()
