"""AnyMDPTaskSampler / AnyPOMDPTaskSampler / MultiTokensAnyPOMDPTaskSampler — procedurally generated tabular
tasks with the reference's signatures and dict schema (xenoverse/anymdp/task_sampler.py:15-118; generative
model in task_sampler_utils.py:11-256; acceptance test in solver.py:57-148).

This is a re-statement of the generative model, not of the reference's random stream: a seed gives a different
(equally distributed in kind) task than the reference's, deterministically.  What is kept exactly: the keys,
dtypes and shapes of the task dict, every structural constraint the env relies on
(rows of terminal states all-zero, other rows sum to 1, s_0 and s_e disjoint, banded transitions), and the
acceptance rule (value gap between the optimal and the uniform-random policy >= 2 at every start state; Gini
impurity > 0.70 and normalised entropy > 0.35 of the long-run occupancy of the greedy policy).

Why it exists: the reference sampler spends its time in a triple Python loop of damped Gauss-Seidel value
iteration (518 s for one 64x8 task without numba); here value iteration is a batched matrix
recursion (same fixed point: the Bellman operator is a contraction), ~0.1 s per 64x8 task on one core.
"""
import numpy as np
import scipy.sparse as sp

EPS = 1e-10


def value_iteration(T, R, gamma, greedy=True, tol=1.0e-4, max_iter=20000):
    """Q[s,a] = sum_s' T[s,a,s'] (gamma * V(s') + R[s,a,s']),  V = max_a Q (greedy) or mean_a Q (uniform policy).
    Iterated to an rms update <= tol (the reference's stopping rule, solver.py:64-81)."""
    ns, na, _ = T.shape
    ER = np.einsum("san,san->sa", T, R)
    Q = np.zeros((ns, na))
    for _ in range(max_iter):
        V = Q.max(1) if greedy else Q.mean(1)
        Qn = ER + gamma * (T @ V)
        diff = np.sqrt(np.mean((Qn - Q) ** 2))
        Q = Qn
        if diff <= tol:
            break
    return Q


def occupancy_stats(s_0, s_0_prob, s_e, T, Q, K):
    """Gini impurity and normalised entropy of the 2^K-step occupancy under the greedy policy, with terminal
    states wired back to the start distribution (solver.py:84-103)."""
    ns = T.shape[0]
    a_max = Q.argmax(1)
    P = T[np.arange(ns), a_max, :].copy()
    for s in s_e:
        P[s, :] = 0.0
        P[s, s_0] = s_0_prob
    for _ in range(K):
        P = P @ P
    gini, ent = [], []
    for s in s_0:
        p = P[s] + 1.0e-12
        gini.append(1.0 - np.sum(p * p))
        ent.append(-np.sum(p * np.log(p)) / np.log(ns))
    return min(gini), min(ent)


def check_task(task):
    """the acceptance test of the reference sampler (solver.py:105-148)"""
    T, R = task["transition"], task["reward"]
    ns = T.shape[0]
    gamma = 2.0 ** (-1.0 / ns)
    q_opt = value_iteration(T, R, gamma, greedy=True)
    q_rnd = value_iteration(T, R, gamma, greedy=False)
    scale = (1.0 - gamma) * task["max_steps"]
    for s in np.atleast_1d(task["s_0"]):
        if (q_opt[s].max() - q_rnd[s].max()) * scale < 2.0:
            return False
    K = int(np.log2(task["max_steps"])) + 1
    gini, ent = occupancy_stats(np.atleast_1d(task["s_0"]), np.atleast_1d(task["s_0_prob"]),
                                list(task["s_e"]), T, q_opt, K)
    row_err = (T.sum(-1) - 1.0) ** 2
    if len(task["s_e"]) > 0:
        row_err[list(task["s_e"])] = 0.0
    if (row_err >= 1.0e-6).any():
        return False
    return gini > 0.70 and ent > 0.35


def _fourier_potential(rng, ns):
    """random smooth potential over the state index: a few sin/cos terms (RandomFourier, utils/random_nn.py:346-368)"""
    base = 0.0 if rng.random_sample() < 0.5 else float(np.clip(rng.exponential(1.0), 0.20, 5.0))
    box = max(rng.uniform(-base, base), 0.0)
    n_items = rng.randint(1, 4)
    x = np.arange(ns) / (2.0 * ns)
    pot = np.zeros(ns)
    orders = [0.0] + [rng.randint(1, 6) + rng.normal() for _ in range(n_items)]
    for o in orders:
        c = rng.normal(size=2) * rng.exponential(scale=box / np.sqrt(n_items), size=2) if box > 0 else np.zeros(2)
        pot += c[0] * np.sin(o * x) + c[1] * np.cos(o * x)
    return pot


def _sample_structure(rng, ns, na, s0_range=3):
    """start states, terminal states and the banded transition tensor (task_sampler_utils.py:65-175)"""
    p0 = np.zeros(s0_range)
    while p0.sum() < EPS:
        p0 = np.clip(rng.normal(size=s0_range), 0, None)
    s_0 = np.where(p0 > EPS)[0]
    s_0_prob = p0[s_0] / p0[s_0].sum()

    p_pit = max(rng.uniform(-0.20, 0.40), 0.0)          # at most 40 % pitfalls
    while True:
        pit = rng.random_sample(ns) < p_pit
        if pit.sum() < ns * p_pit + 1:
            break
    pit[s_0] = False
    final_terminate = bool(rng.random_sample() < 0.3)   # the last state is a terminal goal
    pit[-1] = final_terminate
    s_e = np.where(pit)[0].tolist()
    term = set(s_e)

    max_leap = max(2, ns // 4 + 1)
    max_back = max(2, ns // 2 + 1)
    ss = np.zeros((ns, ns))
    lo = np.zeros(ns, int)
    hi = np.zeros(ns, int)
    for s in range(ns):
        if s in term:
            continue
        f_min = max(0, s - max_back)
        f_max = max(0, s - 1, f_min + 1)
        t_max = min(ns, s + max_leap)
        t_min = min(ns - 1, s + 1, t_max - 1)
        s_from = rng.randint(f_min, f_max)
        s_to = rng.randint(t_min, t_max)
        fwd = []
        while s_to < ns:                                 # widen until two non-terminal forward states are inside
            fwd = [j for j in range(s + 1, s_to) if j not in term]
            if len(fwd) > 1:
                break
            s_to += 1
        else:
            fwd = [j for j in range(s + 1, s_to) if j not in term]
        if final_terminate:
            fwd.append(ns - 1)
        need = fwd if len(fwd) > 1 else list(range(ns))
        while ss[s][need].sum() < 1.0e-3 or (ss[s] > 1.0e-3).sum() < 2:
            ss[s, s_from:s_to] = np.clip(rng.normal(size=s_to - s_from), 0.10, 1.0)
        ss[s, s] /= 2.0                                   # damp self loops; none at the last state
        if s == ns - 1:
            ss[s, s] = 0.0
        ss[s] /= ss[s].sum()
        lo[s], hi[s] = s_from, s_to

    T = np.zeros((ns, na, ns))
    for s in range(ns):
        if s in term:
            continue
        w = hi[s] - lo[s]
        centre = rng.uniform(lo[s] - 1, hi[s], size=na)
        d2 = (centre[:, None] - np.arange(lo[s], hi[s])[None, :]) ** 2
        sigma = float(np.clip(rng.exponential(1.0), 0.20, 1.6))
        ap = np.exp(-d2 / sigma ** 2)
        col = ap.sum(0)
        for j in np.where(col < EPS)[0]:
            ap[np.argmin(d2[:, j]), j] = 1.0
        ap = ap / ap.sum(0)
        T[s, :, lo[s]:hi[s]] = ap * ss[s:s + 1, lo[s]:hi[s]]
        T[s] = T[s] / T[s].sum(-1, keepdims=True)
        assert w > 0
    return s_0, s_0_prob, s_e, final_terminate, T


def sample_mdp(rng, ns, na, max_steps, max_try=5):
    """one candidate MDP, or None if its terminal rewards cannot be repaired (task_sampler_utils.py:177-256)"""
    s_0, s_0_prob, s_e, final_terminate, T = _sample_structure(rng, ns, na)
    pot = _fourier_potential(rng, ns)
    r_pot = pot[:, None, None] - pot[None, None, :]
    # position reward: cumulative positive-part normal, centred; silent at terminals
    base = rng.exponential(0.2)
    pdf = np.clip(rng.normal(size=ns), 0.0, None)
    pdf[-1] += 0.20
    cdf = np.cumsum(pdf * base)
    r_pos = cdf - rng.uniform(0.1 * cdf[-1], 0.9 * cdf[-1])
    n_pos = np.clip(rng.uniform(-0.30, 0.30, size=ns), 0.0, None) * base
    r_pos[s_e] = 0.0
    n_pos[s_e] = 0.0
    # sparse state-action cost
    cbase = float(np.clip(rng.exponential(0.05), 0.0, 0.10))
    mask = (rng.uniform(-0.7, 0.3, size=(ns, na)) > 0).astype(float)
    r_sa = cbase * rng.normal(size=(ns, na)) * mask
    n_sa = 0.30 * cbase * np.clip(rng.normal(size=(ns, na)), 0, None) * mask
    if final_terminate:
        r_step = min(rng.normal(), 0.0) * 0.01
    elif len(s_e) > 0:
        r_step = max(rng.normal(), 0.0) * 0.01
    else:
        r_step = 0.0
    raw = r_pot + r_pos[None, None, :] + r_sa[:, :, None] + r_step
    noise = np.broadcast_to(n_pos[None, None, :] + n_sa[:, :, None], (ns, na, ns)).copy()

    term_reward = np.zeros(ns)
    term_reward[-1] = 1.0
    pitfalls = [s for s in s_e if not (final_terminate and s == ns - 1)]
    last_valid = ns - 2 if final_terminate else ns - 1
    non_pit = [i for i in range(ns) if i not in set(s_e)]
    gamma = 0.99
    for _ in range(max_try):
        Q = value_iteration(T, raw + term_reward[None, None, :], gamma)
        V = Q.max(1)
        pitgain = term_reward.min() - V[non_pit].min() + 1.0
        goalfall = V[s_0].max() - V[last_valid] + rng.uniform(2.0, 5.0)
        if pitgain <= 0 and goalfall <= 0:
            break
        if pitgain > 0:
            term_reward[pitfalls] -= pitgain + rng.uniform(1.0, 10.0)
        if goalfall > 0:
            dv = max(2.0 * goalfall, rng.uniform(1.0, 10.0))
            term_reward[-1] += dv if final_terminate else (1.0 - gamma) * dv
    else:
        return None
    return {"s_0": np.asarray(s_0), "s_0_prob": np.asarray(s_0_prob), "s_e": np.asarray(s_e, dtype=np.int64),
            "final_goal_terminate": final_terminate, "transition": T, "reward": raw + term_reward[None, None, :],
            "reward_noise": noise}


def sample_bandit(rng, na):
    """single-state task (state_space == 1)"""
    base = float(np.clip(rng.exponential(1.0), 0.05, 2.0))
    nbase = max(rng.uniform(-0.30, 0.30), 0.0)
    return {"s_0": np.array([0]), "s_0_prob": np.array([1.0]), "s_e": np.array([], np.int64),
            "final_goal_terminate": False, "transition": np.ones((1, na, 1)),
            "reward": (base * rng.normal(size=(1, na, 1))), "reward_noise": nbase * base * np.ones((1, na, 1))}


def AnyMDPTaskSampler(state_space=64, action_space=5, min_state_space=None, seed=None, verbose=False):
    rng = np.random.RandomState(seed)
    assert (state_space >= 8 or state_space == 1), "State Space must be at least 8 or 1 (Multi-armed Bandit)!"
    if state_space < 2:
        max_steps = 1
    else:
        lower = max(4.0 * state_space, 100)
        upper = max(min(8.0 * state_space, 500), lower + 1)
        max_steps = rng.uniform(lower, upper)
    if min_state_space is None:
        real = state_space
    else:
        min_state_space = min(min_state_space, state_space)
        assert (min_state_space >= 8), "Minimum State Space must be at least 8!"
        real = rng.randint(min_state_space, state_space + 1)
    task = {"ns": state_space, "na": action_space, "max_steps": max_steps,
            "state_mapping": rng.permutation(state_space)[:real], "task_type": "MDP"}
    tries = 0
    while True:
        tries += 1
        if real == 1:
            task.update(sample_bandit(rng, action_space))
            break
        res = sample_mdp(rng, real, action_space, max_steps)
        if res is not None:
            task.update(res)
            if check_task(task):
                break
        if verbose and tries % 10 == 0:
            print("AnyMDPTaskSampler: %d candidates rejected so far" % tries)
    return task


def GarnetTaskSampler(state_space=128, action_space=5, min_state_space=None, b=2, sigma=0.1, seed=None, verbose=False):
    """Garnet MDPs (reference task_sampler.py:120-160, task_sampler_utils.py:274-313): every (s, a) row puts a random
    partition of 1 on `b` distinct next states, rewards are N(0, sigma), no terminal states, s_0 = {0}, no reward
    noise.  The draws follow the reference's order on a RandomState(seed), so a seeded task equals the reference's
    (checked in the build container and by tests/golden/garnet_8x2_seed3.npz).  b >= 2 (the reference's b = 1 path
    raises a NameError)."""
    rng = np.random.RandomState(seed)
    assert (state_space >= 8 or state_space == 1), "State Space must be at least 8 or 1 (Multi-armed Bandit)!"
    if b < 2 or b > state_space:
        raise ValueError("b must satisfy 2 <= b <= state_space")
    if state_space < 2:
        max_steps = 1
    else:
        lower = max(4.0 * state_space, 100)
        upper = max(min(8.0 * state_space, 500), lower + 1)
        max_steps = rng.uniform(lower, upper)
    if min_state_space is None:
        real = state_space
    else:
        min_state_space = min(min_state_space, state_space)
        assert (min_state_space >= 8), "Minimum State Space must be at least 8!"
        real = rng.randint(min_state_space, state_space + 1)
    task = {"ns": state_space, "na": action_space, "max_steps": max_steps,
            "state_mapping": rng.permutation(state_space)[:real], "task_type": "MDP"}
    assert real >= 8, "ns must be at least 8 for MDP"
    transition = np.zeros((real, action_space, real))
    arr = np.arange(real)
    for i in range(real):
        for j in range(action_space):
            sample = rng.choice(arr, size=b, replace=False)
            cuts = rng.random_sample(b - 1)
            cuts.sort()
            cuts = np.concatenate(([0], cuts, [1]))
            transition[i, j, sample] = cuts[1:] - cuts[:-1]
    reward = rng.normal(size=(real, action_space, real)) * sigma + 0.0
    task.update({"s_0": np.array([0]), "s_0_prob": np.array([1.0]), "s_e": np.array([], dtype=int),
                 "transition": transition, "reward": reward, "reward_noise": np.zeros((real, action_space, real)),
                 "final_goal_terminate": False})
    return task


def _obs_matrix(rng, n_states, n_obs, density, maximum_distribution):
    density = min(density, maximum_distribution / n_obs)
    m = sp.random(n_states, n_obs, density=density, format="csr", random_state=rng).toarray()
    for i in range(n_states):
        if m[i].sum() == 0:
            m[i][rng.randint(n_obs)] = 1
        m[i] /= m[i].sum()
    return m


def AnyPOMDPTaskSampler(state_space=64, action_space=5, min_state_space=None, observation_space=64, density=0.20,
                        maximum_distribution=4, seed=None, verbose=False):
    task = AnyMDPTaskSampler(state_space, action_space, min_state_space, seed, verbose)
    rng = np.random.RandomState(None if seed is None else seed + 1000003)
    task["no"] = observation_space
    task["task_type"] = "POMDP"
    task["observation_transition"] = _obs_matrix(rng, task["state_mapping"].shape[0], observation_space, density,
                                                 maximum_distribution)
    return task


def MultiTokensAnyPOMDPTaskSampler(state_space=256, action_space=5, min_state_space=None, observation_space=64,
                                   observation_tokens=4, action_tokens=2, density=0.20, maximum_distribution=4,
                                   seed=None, verbose=False):
    task = AnyMDPTaskSampler(state_space, action_space, min_state_space, seed, verbose)
    rng = np.random.RandomState(None if seed is None else seed + 1000003)
    task.update(no=observation_space, do=observation_tokens, da=action_tokens, task_type="MTPOMDP")
    task["observation_transition"] = [_obs_matrix(rng, task["state_mapping"].shape[0], observation_space, density,
                                                  maximum_distribution) for _ in range(observation_tokens)]
    return task


def sample_batch(n, sampler=None, seed=None, **kwargs):
    """n tasks from `sampler` (default AnyMDPTaskSampler; task k uses seed + k) as ONE dict of stacked arrays — the
    struct-of-arrays tables `AnyMDPVecEnv.set_task` uploads as they are (anymdp.tables.build_tables)."""
    from .tables import build_tables
    sampler = AnyMDPTaskSampler if sampler is None else sampler
    base = np.random.SeedSequence(seed).generate_state(1)[0] if seed is None else int(seed)
    return build_tables([sampler(seed=base + k, **kwargs) for k in range(n)])
