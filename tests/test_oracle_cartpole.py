"""CartPole oracle: PARITY UNPINNED — gymnasium (whose CartPoleEnv.step the reference subclasses,
metacontrol/random_cartpole.py:11,57) is neither vendored in the reference nor installed here, and the
reference's tests hold no vectors for it.  The restatement is checked against an independent fp64 NumPy
statement of the published gymnasium 1.x equations and against hand-computed known answers."""
import numpy as np

import oracle


def gym_cartpole_step_f64(state, action, gravity, masscart, masspole, length):
    """gymnasium.envs.classic_control.CartPoleEnv.step (Euler), public 1.x source equations, fp64."""
    x, x_dot, theta, theta_dot = state
    force = 10.0 if action == 1 else -10.0
    polemass_length, total_mass = masspole * length, masspole + masscart
    costheta, sintheta = np.cos(theta), np.sin(theta)
    temp = (force + polemass_length * theta_dot ** 2 * sintheta) / total_mass
    thetaacc = (gravity * sintheta - costheta * temp) / (length * (4.0 / 3.0 - masspole * costheta ** 2 / total_mass))
    xacc = temp - polemass_length * thetaacc * costheta / total_mass
    x, x_dot = x + 0.02 * x_dot, x_dot + 0.02 * xacc
    theta, theta_dot = theta + 0.02 * theta_dot, theta_dot + 0.02 * thetaacc
    term = bool(x < -2.4 or x > 2.4 or theta < -12 * 2 * np.pi / 360 or theta > 12 * 2 * np.pi / 360)
    return np.array([x, x_dot, theta, theta_dot]), 1.0, term


def test_known_answer_default_cartpole_from_rest():
    # classic CartPole-v1 constants, zero state, push right: temp = 10/1.1; thetaacc = -temp/(0.5*(4/3-0.1/1.1))
    o = oracle.CartPoleOracle([[9.8, 1.0, 0.1, 0.5]], [0], frameskip=1)
    o.need_reset[:] = 0
    out = o.step_injected([1], np.zeros((4, 1)), 0)
    temp = 10.0 / 1.1
    thacc = -temp / (0.5 * (4.0 / 3.0 - 0.1 / 1.1))
    xacc = temp - 0.05 * thacc / 1.1
    assert np.allclose(out["obs"][0], [0.0, 0.02 * xacc, 0.0, 0.02 * thacc], rtol=1e-6, atol=0)
    assert out["reward"][0] == 1.0 and not out["terminated"][0] and not out["truncated"][0]


def test_matches_fp64_equations_per_step_and_termination():
    rng = np.random.RandomState(0)
    n = 512
    params = np.stack([rng.uniform(1, 11, n), rng.uniform(0.5, 2, n), rng.uniform(0.05, 0.2, n), rng.uniform(0.25, 1, n)], 1)
    o = oracle.CartPoleOracle(params, np.arange(n), frameskip=1)
    u0 = rng.random_sample((4, n))
    o.reset_injected(u0)
    n_term = 0
    for t in range(200):
        a = rng.randint(0, 2, n)
        before = o.state.astype(np.float64).copy()
        out = o.step_injected(a, rng.random_sample((4, n)), 0)
        for i in range(0, n, 37):
            s, r, term = gym_cartpole_step_f64(before[:, i], a[i], *params[i])
            # the oracle computes in float64 like gymnasium: the state agrees to rounding, the observation is its
            # float32 cast, the termination flag is the same decision
            assert np.allclose(o.state[:, i] if not term else s, s, rtol=1e-13, atol=1e-15)
            assert np.array_equal(out["obs"][i], s.astype(np.float32)) or np.allclose(out["obs"][i], s, rtol=2e-7, atol=1e-9)
            assert bool(out["terminated"][i]) == term
        done = out["terminated"].astype(bool)
        n_term += int(done.sum())
        if done.any():
            o.reset_injected(rng.random_sample((4, n)), mask=done.astype(np.uint8))
    assert n_term > 100


def test_frameskip_accumulates_reward_and_stops_at_termination():
    o = oracle.CartPoleOracle([[9.8, 1.0, 0.1, 0.5]], [0], frameskip=5)
    o.need_reset[:] = 0
    out = o.step_injected([1], np.zeros((4, 1)), 0)
    assert out["reward"][0] == 5.0
    o.state[:, 0] = [2.39, 3.0, 0.0, 0.0]      # leaves the track on the first sub-step
    out = o.step_injected([1], np.zeros((4, 1)), 0)
    assert out["terminated"][0] == 1 and out["reward"][0] == 1.0   # random_cartpole.py:56-60 breaks on done


def test_reset_distribution_and_modes():
    o = oracle.CartPoleOracle([[9.8, 1.0, 0.1, 0.5]], np.zeros(4, np.int32), frameskip=1, max_steps=3)
    u = np.array([[0.0, 0.5, 1.0 - 2 ** -24, 0.25]] * 4, np.float32).T.copy()   # [component][env]
    obs = o.reset_injected(u)
    assert np.allclose(obs[0], [-0.45, 0.0, 0.13 * (1 - 2 ** -23), -0.5])   # uniform(-1,1)*scale, :70
    for k in range(3):
        out = o.step_injected(np.ones(4, np.int32), np.full((4, 4), 0.5, np.float32), 2)
    assert out["truncated"].all() and np.all(o.steps == 0) and np.allclose(out["obs"], 0.0)
    assert np.any(out["final_obs"] != 0)
