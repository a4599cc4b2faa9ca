"""LinearDSSampler / LinearDSSamplerRandomDim — randomised LTI control tasks with the reference's signature, dict schema
and draw order (xenoverse/linds/task_sampler.py:60-154, building blocks :12-58; RandomFourier
xenoverse/utils/random_nn.py:346-368; weights_and_biases :45-53).

The reference seeds NumPy with `timestamp + system random + seed` (pseudo_random_seed, utils/random_nn.py:9-16), so its
sampler is not reproducible even with a seed.  What IS fixed is the function from NumPy's stream to the task; this
module consumes a `RandomState(seed)` in the reference's order, so `LinearDSSampler(16, 8, 8, seed=k)` equals what the
reference returns when its generator is in the state `numpy.random.seed(k)` leaves it in
(tests/golden/sampler_reflinds.npz, written with the reference's seeding function pinned to exactly that).
One documented deviation: the reference's initial-state rejection loop (:108-132) has no bound; its acceptance
probability depends on the task (a few per cent for some (16, 8, 8) tasks, ~5e-4 at state_dim = 32, ~0 at (32, 8, 16),
SURVEY.md §7), so there it can run for minutes or forever.  Here the loop follows the reference for up to 20,000
draws (tens of microseconds each); only then are the initial states of the last draw shrunk toward the origin until
they pass, so the sampler terminates for every dimension.
"""
import numpy as np


class RandomFourier(object):
    """y(t) = sum_k c_k[:,0] sin(w_k t/max_steps) + c_k[:,1] cos(w_k t/max_steps); w_0 = 0, 1..max_item further terms"""

    def __init__(self, ndim, max_order=16, max_item=5, max_steps=1000, box_size=2, rng=None):
        rng = rng if rng is not None else np.random
        n_items = rng.randint(1, max_item + 1)
        self.coeffs = [(0, rng.normal(size=(ndim, 2)) * rng.exponential(scale=box_size / np.sqrt(n_items), size=(ndim, 2)))]
        self.max_steps = max_steps
        for _ in range(n_items):
            order = rng.randint(1, max_order + 1) + rng.normal(scale=1.0)
            factor = rng.normal(size=(ndim, 2)) * rng.exponential(scale=box_size / np.sqrt(n_items), size=(ndim, 2))
            self.coeffs.append((order, factor))

    def __call__(self, t):
        x = t / self.max_steps
        y = 0
        for order, coeff in self.coeffs:
            y = y + coeff[:, 0] * np.sin(order * x) + coeff[:, 1] * np.cos(order * x)
        return y


def _weights_and_biases(rng, n_in, n_out, need_bias=False):
    w = rng.normal(0, np.sqrt(2.0 / (n_in + n_out)), size=(n_out, n_in)) * 3      # xavier_normal_init(gain=3)
    b = 0.1 * rng.normal(size=[n_out]) if need_bias else np.zeros(shape=[n_out])
    return w, b


def _sample_variants(rng, ns, na, no):
    AB, X = _weights_and_biases(rng, ns + na, ns, need_bias=True)
    C, Y = _weights_and_biases(rng, ns, no, need_bias=False)
    A = AB[:, :ns] * rng.choice([0.01, 0.02, 0.05, 0.1, 0.2])
    B = AB[:, ns:]
    X = X * rng.choice([0.0, 0.05, 0.1])
    trim = int(rng.choice(3))          # choice over the three trimming rules (:52-53)
    if trim == 0:      # banded
        width = rng.randint(2, max(ns // 2, 3) + 1)
        if width < ns:
            i, j = np.indices((ns, ns))
            A = np.where(np.abs(i - j) > width, 0.0, A)
    elif trim == 1:    # triangular
        width = rng.randint(-1, max(ns // 4, 2) + 1)
        if width < ns:
            i, j = np.indices((ns, ns))
            A = np.where(j < i + width, 0.0, A)
    return A, B, C, X, Y


MAX_DRAWS = 20000      # rejected (initial states, command) draws before the shrinking fallback


def LinearDSSampler(state_dim=16, action_dim=8, observation_dim=8, seed=None, verbose=False):
    rng = np.random.RandomState(seed)
    task = dict(state_dim=state_dim, observation_dim=observation_dim, action_dim=action_dim)
    task["max_steps"] = int(rng.randint(100, 1000))
    while True:
        A, B, C, X, Y = _sample_variants(rng, state_dim, action_dim, observation_dim)
        if (np.linalg.matrix_rank(B) > min(action_dim, state_dim) - 1 and
                np.linalg.matrix_rank(C) > min(observation_dim, state_dim) - 1):
            break
    task.update(ld_A=A, ld_B=B, ld_C=C, ld_X=X, ld_Y=Y)
    task["action_cost"] = max(rng.uniform(-1.0, 1.0) * rng.exponential(0.05), 0.0)
    task["reward_base"] = rng.exponential(0.10)
    task["terminate_punish"] = rng.exponential(scale=5.0) * rng.choice([0, 1, 1])
    task["reward_factor"] = rng.exponential(scale=0.50)
    eps = min(rng.uniform(0.2, 1.2), 1.0)
    tv = np.zeros((observation_dim,))
    while np.sum(tv) < 0.5:
        tv = rng.binomial(1, eps, size=(observation_dim,))
    task["target_valid"] = tv
    task["target_type"] = str(rng.choice(["dynamic_target", "dynamic_target", "static_target"]))

    def close_enough(x0, cmd):
        return not (np.linalg.norm((cmd - C @ x0 - Y) * tv) > 3.0 or np.linalg.norm(x0) > 10.0)

    for attempt in range(MAX_DRAWS + 1):   # the reference re-draws everything below until all initial states pass
        born_loc = int(max(rng.exponential(scale=1.0), 1))                 # exponential: how many initial states
        states = [rng.randn(state_dim) for _ in range(born_loc)]          # randn(ns) each
        task["noise_drift"] = np.clip(rng.uniform(-0.02, 0.02), 0.0, 0.02)   # uniform
        if task["target_type"] == "static_target":
            task["command"] = rng.randn(observation_dim) * rng.choice([0, 1])   # randn(no), choice
            task["target_delay"] = 0
            cmd = task["command"]
        else:
            task["command"] = RandomFourier(observation_dim, rng=rng)
            task["target_delay"] = max(rng.randint(-10, 30), 0)                  # randint
            cmd = task["command"](-task["target_delay"])
        if attempt >= MAX_DRAWS:     # the deviation: shrink toward the origin instead of rejecting forever
            for k in range(len(states)):
                for _ in range(24):
                    if close_enough(states[k], cmd):
                        break
                    states[k] = 0.7 * states[k]
        if all(close_enough(x0, cmd) for x0 in states):
            break
    else:   # last resort: start on the least-squares pre-image of the command
        M = tv[:, None] * C
        x_ls = np.linalg.pinv(M) @ (tv * (cmd - Y))
        x_ls *= min(1.0, 9.0 / max(np.linalg.norm(x_ls), 1e-9))
        states = [x_ls + 0.01 * rng.randn(state_dim) for _ in range(born_loc)]
    task["initial_states"] = states
    return task


def LinearDSSamplerRandomDim(max_state_dim=16, max_observation_dim=16, max_action_dim=8, seed=None, verbose=False):
    assert max_state_dim >= 2, "max_state_dim should be at least 2"
    assert max_action_dim >= 1, "max_action_dim should be at least 1"
    rng = np.random.RandomState(seed)
    state_dim = int(rng.randint(1, max_state_dim + 1))
    min_action_dim = max(1, (state_dim + 1) // 2)
    max_action_dim = max(min(max_action_dim, state_dim * 3 // 2), min_action_dim)
    min_observation_dim = max(1, state_dim // 4)
    max_observation_dim = max(min(max_observation_dim, state_dim * 3 // 2), min_observation_dim)
    action_dim = int(rng.randint(min_action_dim, max_action_dim + 1))
    observation_dim = int(rng.randint(min_observation_dim, max_observation_dim + 1))
    return LinearDSSampler(state_dim, action_dim, observation_dim, seed=None if seed is None else seed + 1)


def sample_batch(n, sampler=None, seed=None, dt=0.1, pad_observation_dim=16, pad_action_dim=8, pad_command_dim=16,
                 **kwargs):
    """n tasks from `sampler` (default LinearDSSampler; task k uses seed + k) as ONE dict of stacked arrays — the
    tables `LinDSVecEnv.set_task` uploads as they are (linds.tables.build_tables, ZOH-discretised on the host)."""
    from .tables import build_tables
    sampler = LinearDSSampler if sampler is None else sampler
    base = np.random.SeedSequence(seed).generate_state(1)[0] if seed is None else int(seed)
    return build_tables([sampler(seed=base + k, **kwargs) for k in range(n)], dt=dt,
                        pad_observation_dim=pad_observation_dim, pad_action_dim=pad_action_dim,
                        pad_command_dim=pad_command_dim)
