"""AnyMDPTaskSampler / AnyPOMDPTaskSampler / MultiTokensAnyPOMDPTaskSampler / GarnetTaskSampler — procedurally
generated tabular tasks with the reference's signatures, dict schema AND random stream
(xenoverse/anymdp/task_sampler.py:15-160; generative model task_sampler_utils.py:11-313; acceptance solver.py:57-148).

Seed compatibility.  The reference draws from NumPy's global legacy generator after `random.seed(seed)`; a task is a
function of that stream, of the order in which the model consumes it, and — through the repair loop, which feeds value
iteration results back into the rewards — of the exact arithmetic of `update_value_matrix`.  This module consumes a
`numpy.random.RandomState(seed)` (the same MT19937 + legacy distributions) in the reference's order, evaluates the
array expressions with the same NumPy calls, and runs the value iterations in `libxeno_hip.so`'s host function
`xv_anymdp_value_iteration_gs` (C++, the reference's damped Gauss-Seidel in its order of operations, ~1000x the
speed of the interpreted loop: 64x8 tasks in seconds instead of ~9 minutes).  Result: `AnyMDPTaskSampler(16, 4,
seed=k)`, `(64, 8, seed=1)` and the POMDP variants equal the reference's tasks bit for bit (tests/golden/anymdp_*,
tests/test_host_samplers.py) on a host whose NumPy takes the same SIMD paths as the one that wrote the fixtures
(`numpy.exp` differs in the last bit between NumPy's AVX512 and scalar kernels: the reference itself is only
reproducible per machine class).  "The reference" here is the reference run as plain Python (its `@njit` functions
interpreted — numba is not installable in the build image — which is how the fixtures were made): NumPy reduces the
`np.mean` calls of `update_value_matrix` pairwise, a numba-compiled run reduces them sequentially, and the last bits of
those means can differ.  `set_vi_summation("numba")` selects the sequential order (not pinned by any fixture).

Throughput sampling of fresh tasks by the thousand is a different job: `device_sampler.sample_batch_device` runs the
same generative model with counter-based randomness on the GPU (not stream-compatible, distribution-tested).
"""

import numpy as np
import scipy.sparse as sp

EPS = 1e-10


def set_vi_summation(order="numpy"):
    """order of the np.mean reductions inside the C++ Gauss-Seidel value iteration: "numpy" (pairwise; the interpreted
    reference, pinned by the fixtures; default) or "numba" (one sequential loop, as numba compiles np.mean; unpinned)"""
    from .. import _lib
    _lib.check(_lib.load().xv_anymdp_value_iteration_set_summation({"numpy": 0, "numba": 1}[order]))


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


def gauss_seidel_values(T, R, gamma, greedy=True, start=None):
    """`update_value_matrix(T, R, gamma, start)` of the reference (solver.py:57-82), bit for bit, through the C-ABI
    host function -> float64[ns, na]"""
    from .. import _lib
    T = np.ascontiguousarray(T, np.float64)
    R = np.ascontiguousarray(R, np.float64)
    ns, na, _ = T.shape
    vm = np.zeros((ns, na)) if start is None else np.array(start, np.float64, order="C", copy=True)
    _lib.check(_lib.load().xv_anymdp_value_iteration_gs(T.ctypes.data, R.ctypes.data, ns, na, float(gamma),
                                                        1 if greedy else 0, vm.ctypes.data, None))
    return vm


def reference_acceptance(task):
    """check_valuefunction (solver.py:105-148) with the reference's own arithmetic: Gauss-Seidel values for the greedy
    and the uniform policy, value gap >= 2 at every start state, long-run occupancy statistics by repeated squaring"""
    T = np.copy(task["transition"])
    R = np.copy(task["reward"])
    ns, na, _ = T.shape
    gamma = np.power(2, -1.0 / ns)
    vm_opt = gauss_seidel_values(T, R, gamma, True)
    vm_rnd = gauss_seidel_values(T, R, gamma, False)
    avg_opt = vm_opt * (1.0 - gamma) * task["max_steps"]
    avg_rnd = vm_rnd * (1.0 - gamma) * task["max_steps"]
    for s in task["s_0"]:
        if np.max(avg_opt[s]) - np.max(avg_rnd[s]) < 2.0:
            return False
    K = int(np.log2(task["max_steps"])) + 1
    greedy = np.argmax(vm_opt, axis=1)
    P = np.copy(T[np.arange(ns)[:, None], greedy[:, None], np.arange(ns)])
    for s in task["s_e"]:
        P[s, task["s_0"]] = task["s_0_prob"]          # terminal states lead straight back to the start states
    for _ in range(K):
        P = np.matmul(P, P)
    gini, ent = [], []
    for s in task["s_0"]:
        p = P[s] + 1.0e-12
        gini.append(1.0 - np.sum(p * p))
        ent.append(-np.sum(p * np.log(p)) / np.log(ns))
    err = (np.sum(T, axis=-1) - 1.0) ** 2
    if len(task["s_e"]) > 0:
        err[task["s_e"]] = 0.0
    if (err >= 1.0e-6).any():
        return False
    return bool(np.min(gini) > 0.70 and np.min(ent) > 0.35)


class _ReferenceStream(object):
    """The generative model of task_sampler_utils.py consuming a legacy RandomState in the reference's order.  Each
    method is one stage; the comments give the draws it makes (that order IS the compatibility contract)."""

    def __init__(self, rng, ns, na):
        self.rng, self.ns, self.na = rng, ns, na

    # -- stage 1: start states, terminal states (task_sampler_utils.py:65-93) ------------------------------------
    def start_and_terminal_states(self, s0_range=3):
        rng, ns = self.rng, self.ns
        w = 0.0
        while np.sum(w) < EPS:                                   # normal(size=3) until some mass is positive
            w = np.clip(rng.normal(loc=0, scale=1, size=(s0_range)), 0, None)
        s_0 = np.where(w > EPS)[0]
        s_0_prob = w[s_0]
        s_0_prob = s_0_prob / np.sum(s_0_prob)
        p_pit = np.clip(rng.uniform(-0.20, 0.40), 0.0, None)     # uniform: pitfall density, at most 40 %
        while True:                                              # choice(size=ns) until not more pitfalls than expected
            pit = rng.choice([0, 1], size=ns, p=[1 - p_pit, p_pit])
            if np.sum(pit) < ns * p_pit + 1:
                break
        pit[s_0] = 0
        goal_terminates = bool(rng.random_sample() < 0.3)        # random(): the last state is a terminal goal
        pit[-1] = 1 if goal_terminates else 0
        self.s_0, self.s_0_prob = s_0, s_0_prob
        self.s_e = np.where(pit == 1)[0].tolist()
        self.goal_terminates = goal_terminates

    # -- stage 2: banded state-to-state kernel (:95-151) -----------------------------------------------------------
    def band_kernel(self):
        rng, ns = self.rng, self.ns
        term = set(self.s_e)
        fwd_max = max(2, ns // 4 + 1)
        back_max = max(2, ns // 2 + 1)
        ss = np.zeros((ns, ns), dtype=float)
        lo = np.zeros(ns, dtype=int)
        hi = np.zeros(ns, dtype=int)
        for s in range(ns):
            if s in term:
                continue
            a_lo = max(0, s - back_max)
            a_hi = max(0, s - 1, a_lo + 1)
            b_hi = min(ns, s + fwd_max)
            b_lo = min(ns - 1, s + 1, b_hi - 1)
            first = rng.randint(a_lo, a_hi)                      # randint: first state of the band
            last = rng.randint(b_lo, b_hi)                       # randint: end of the band (exclusive), always < ns
            while last < ns:                                     # widen until two live states lie ahead inside it
                ahead = [j for j in range(s + 1, last) if j not in term]
                if len(ahead) > 1:
                    break
                last += 1
            # (when the band ran into the end, `ahead` is the list seen at last = ns - 1: the reference's loop leaves
            #  it that way, and the draw condition below depends on it)
            lo[s], hi[s] = first, last
            if self.goal_terminates:
                ahead.append(ns - 1)
            must = ahead if len(ahead) > 1 else slice(None)
            while np.sum(ss[s][must]) < 1.0e-3 or np.where(ss[s] > 1.0e-3)[0].size < 2:
                ss[s, first:last] = np.clip(rng.normal(size=(last - first)), 0.10, 1.0)   # normal(size=band)
            ss[s, s] /= 2.0                                      # damp the self loop; none at the last state
            if s == ns - 1:
                ss[s, s] = 0
            ss[s] = ss[s] / np.sum(ss[s])
        self.ss, self.lo, self.hi = ss, lo, hi

    # -- stage 3: split every band over the actions (:154-175) ------------------------------------------------------
    def action_kernel(self):
        rng, ns, na = self.rng, self.ns, self.na
        term = set(self.s_e)
        T = np.zeros((ns, na, ns), dtype=float)
        for s in range(ns):
            if s in term:
                continue
            lo, hi = self.lo[s], self.hi[s]
            centre = rng.uniform(lo - 1, hi, size=na)            # uniform(size=na): where each action aims
            d2 = (centre[:, None] - np.arange(lo, hi)[None, :]) ** 2
            width = np.clip(rng.exponential(1.0), 0.20, 1.6)     # exponential: how sharply
            aff = np.exp(-d2 / width ** 2)
            col = np.sum(aff, axis=0)
            for i in np.where(col < EPS)[0]:                     # a next state no action reaches: give it to the nearest
                aff[np.argmin(d2[:, i]), i] = 1.0
            aff = aff / np.sum(a=aff, axis=0)
            T[s, :, lo:hi] = aff * self.ss[s:s + 1, lo:hi]
            T[s] = T[s] / np.sum(T[s], axis=-1, keepdims=True)
        self.T = T

    # -- stage 4: reward components (:11-63, :193-207) ----------------------------------------------------------------
    def potential(self):
        rng, ns = self.rng, self.ns
        if rng.random_sample() < 0.5:                            # random(): half of the tasks have no shaping
            base = 0
        else:
            base = np.clip(rng.exponential(1.0), 0.20, 5.0)      # exponential
        box = max(rng.uniform(-base, base), 0.0)                 # uniform
        # RandomFourier(ndim=1, max_order=5, max_item=3, max_steps=2 ns, box_size=box), utils/random_nn.py:346-359
        n_items = rng.randint(1, 3 + 1)                          # randint
        terms = [(0, rng.normal(size=(1, 2)) * rng.exponential(scale=box / np.sqrt(n_items), size=(1, 2)))]
        for _ in range(n_items):                                 # per term: randint + normal, normal(1,2), exponential(1,2)
            order = rng.randint(1, 5 + 1) + rng.normal(scale=1.0)
            terms.append((order, rng.normal(size=(1, 2)) * rng.exponential(scale=box / np.sqrt(n_items), size=(1, 2))))
        pot = []
        for s in range(ns):
            x = s / (ns * 2)
            y = 0
            for order, c in terms:
                y += c[:, 0] * np.sin(order * x) + c[:, 1] * np.cos(order * x)
            pot.append(y[0])
        pot = np.array(pot)
        return pot[:, None, None] - pot[None, None, :]

    def position_reward(self):
        rng, ns = self.rng, self.ns
        base = rng.exponential(0.2)                              # exponential
        pdf = np.clip(rng.normal(size=(ns,)), 0.0, None)         # normal(size=ns)
        pdf[-1] += 0.20
        pdf *= base
        cdf = np.cumsum(pdf)
        r = cdf - rng.uniform(0.1 * cdf[-1], 0.9 * cdf[-1])      # uniform: zero crossing
        noise = np.clip(rng.uniform(-0.30, 0.30, size=r.shape), 0.0, None) * base   # uniform(size=ns)
        r[self.s_e] = 0.0
        noise[self.s_e] = 0.0
        return r[None, None, :], noise[None, None, :]

    def state_action_cost(self):
        rng, ns, na = self.rng, self.ns, self.na
        base = np.clip(rng.exponential(0.05), 0.0, 0.10)         # exponential
        on = (rng.uniform(-0.7, 0.3, size=(ns, na)) > 0).astype(float)               # uniform(ns, na): 30 % of the pairs
        r = base * rng.normal(size=(ns, na)) * on                # normal(ns, na)
        noise = 0.30 * base * np.clip(rng.normal(size=(ns, na)), 0, None) * on       # normal(ns, na)
        return r[:, :, None], noise[:, :, None]

    # -- stage 5: terminal rewards repaired against the value function (:209-256) ---------------------------------------
    def candidate(self, max_try=5):
        """one candidate task (sample_mdp), or None when five repairs do not put the pitfalls below and the goal above"""
        rng, ns = self.rng, self.ns
        self.start_and_terminal_states()
        self.band_kernel()
        self.action_kernel()
        r_pot = self.potential()
        r_pos, n_pos = self.position_reward()
        r_sa, n_sa = self.state_action_cost()
        if self.goal_terminates:
            r_step = min(rng.normal(), 0.0) * 0.01               # normal: a cost per step when there is a goal to reach
        elif len(self.s_e) > 0:
            r_step = max(rng.normal(), 0.0) * 0.01               # normal: a survival bonus when there are pitfalls
        else:
            r_step = 0.0
        raw = r_pot + r_pos + r_sa + r_step
        noise = n_pos + n_sa
        bonus = np.zeros(ns, dtype=float)
        bonus[-1] = 1.0
        gamma = 0.99
        pits = list(self.s_e)
        if self.goal_terminates:
            last_live = ns - 2
            pits.remove(ns - 1)
        else:
            last_live = ns - 1
        live = [i for i in range(ns) if i not in self.s_e]
        vm = np.zeros((ns, self.na), dtype=float)
        tries = 0
        while tries < max_try:
            vm = gauss_seidel_values(self.T, raw + bonus[None, None, :], gamma, True, start=vm)
            v = np.max(vm, axis=-1)
            pit_gap = np.min(bonus) - np.min(v[live]) + 1.0
            goal_gap = np.max(v[self.s_0]) - v[last_live] + rng.uniform(2.0, 5.0)   # uniform: margin (always drawn)
            if pit_gap <= 0 and goal_gap <= 0:
                break
            if pit_gap > 0.0:
                bonus[pits] -= pit_gap + rng.uniform(1.0, 10.0)                     # uniform
            if goal_gap > 0.0:
                lift = max(2.0 * goal_gap, rng.uniform(1.0, 10.0))                  # uniform
                bonus[-1] += lift if self.goal_terminates else (1.0 - gamma) * lift
            tries += 1
        if tries >= max_try:
            return None
        return {"s_0": np.copy(self.s_0), "s_0_prob": np.copy(self.s_0_prob), "s_e": np.copy(self.s_e),
                "transition": np.copy(self.T), "final_goal_terminate": self.goal_terminates,
                "reward": raw + bonus[None, None, :], "reward_noise": np.copy(noise)}


def _bandit(rng, na):
    """single-state task (task_sampler_utils.py:258-272): exponential, uniform, uniform(1, na, 1) until spread"""
    base = np.clip(rng.exponential(1.0), 0.05, 2.0)
    noise_base = np.clip(rng.uniform(-0.30, 0.30), 0.0, None)
    while True:
        reward = rng.uniform(0.5 * base, base, size=(1, na, 1))
        if np.std(reward) > 0.01:
            break
    return {"transition": np.ones((1, na, 1), dtype=float), "reward": reward, "reward_noise": noise_base * reward,
            "s_0": np.array([0]), "s_e": np.array([]), "s_0_prob": np.array([1.0])}


def _task_head(rng, state_space, action_space, min_state_space):
    """max_steps, the active-state subset and the permutation (task_sampler.py:26-50): uniform, [randint], permutation"""
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
    return {"ns": state_space, "na": action_space, "max_steps": max_steps,
            "state_mapping": rng.permutation(state_space)[:real], "task_type": "MDP"}, real


def _mdp_task(rng, state_space, action_space, min_state_space, verbose=False):
    task, real = _task_head(rng, state_space, action_space, min_state_space)
    tries = 0
    while True:
        tries += 1
        if real == 1:
            task.update(_bandit(rng, action_space))
            break
        assert real >= 8, "ns must be at least 8 for MDP"
        res = _ReferenceStream(rng, real, action_space).candidate()
        if res is not None:
            task.update(res)
            if reference_acceptance(task):
                break
        elif verbose:
            print("Failed to generate valid MDP, trying again...")
    if verbose:
        print("AnyMDPTaskSampler: accepted candidate %d" % tries)
    return task


def AnyMDPTaskSampler(state_space=64, action_space=5, min_state_space=None, seed=None, verbose=False):
    """task_sampler.py:15-65.  seed=None draws a seed from OS entropy (the reference: clock + stdlib random)."""
    return _mdp_task(np.random.RandomState(seed), state_space, action_space, min_state_space, verbose)


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
    """task_sampler.py:81-86: a sparse random emission matrix, rows without mass get one random observation"""
    density = min(density, maximum_distribution / n_obs)
    m = sp.random(n_states, n_obs, density=density, format="csr", random_state=rng).toarray()
    for i in range(n_states):
        if np.sum(m[i]) == 0:
            m[i][rng.randint(n_obs)] = 1
        m[i] /= np.sum(m[i])
    return m


def AnyPOMDPTaskSampler(state_space=64, action_space=5, min_state_space=None, observation_space=64, density=0.20,
                        maximum_distribution=4, seed=None, verbose=False):
    """task_sampler.py:67-88: the MDP, then the emission matrix from the SAME stream"""
    rng = np.random.RandomState(seed)
    task = _mdp_task(rng, state_space, action_space, min_state_space, verbose)
    task["no"] = observation_space
    task["task_type"] = "POMDP"
    task["observation_transition"] = _obs_matrix(rng, task["state_mapping"].shape[0], observation_space, density,
                                                 maximum_distribution)
    return task


def MultiTokensAnyPOMDPTaskSampler(state_space=256, action_space=5, min_state_space=None, observation_space=64,
                                   observation_tokens=4, action_tokens=2, density=0.20, maximum_distribution=4,
                                   seed=None, verbose=False):
    """task_sampler.py:90-118"""
    rng = np.random.RandomState(seed)
    task = _mdp_task(rng, state_space, action_space, min_state_space, verbose)
    task.update(no=observation_space, do=observation_tokens, da=action_tokens, task_type="MTPOMDP")
    mats = []
    for _ in range(observation_tokens):
        density = min(density, maximum_distribution / observation_space)
        mats.append(_obs_matrix(rng, task["state_mapping"].shape[0], observation_space, density, maximum_distribution))
    task["observation_transition"] = mats
    return task


def sample_batch(n, sampler=None, seed=None, **kwargs):
    """n tasks from `sampler` (default AnyMDPTaskSampler; task k uses seed + k) as ONE dict of stacked arrays — the
    struct-of-arrays tables `AnyMDPVecEnv.set_task` uploads as they are (anymdp.tables.build_tables)."""
    from .tables import build_tables
    sampler = AnyMDPTaskSampler if sampler is None else sampler
    base = np.random.SeedSequence(seed).generate_state(1)[0] if seed is None else int(seed)
    return build_tables([sampler(seed=base + k, **kwargs) for k in range(n)])
