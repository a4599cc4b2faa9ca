"""CPU: the Acrobot oracle against (i) vectors produced by the reference's OWN code — RandomAcrobotEnv._dsdt,
._terminal and the reset-state formula (tests/golden/acrobot_dsdt.npz, oracle/gen_golden.py acrobot) — and (ii) a
plain-numpy reading of gymnasium's public AcrobotEnv.step / rk4 / wrap / bound (third-party, not installed:
that part of the parity is unpinned and this is a cross-check of two restatements, not a pin)."""
import os

import numpy as np

import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "acrobot_dsdt.npz")


def test_dsdt_and_terminal_match_the_reference_functions():
    g = np.load(GOLD)
    got = oracle.acrobot_dsdt(g["params"], g["y"])
    # theta-independent entries and most others are bit-identical; pow(x, 2) of Python floats in the reference vs
    # x*x here may differ in the last bit, which the 1e-13 relative bound absorbs
    assert np.array_equal(got[:, :2], g["dsdt"][:, :2]) and np.all(got[:, 4] == 0.0)
    assert np.allclose(got, g["dsdt"], rtol=1e-13, atol=1e-13)
    frac_exact = np.mean(got == g["dsdt"])
    assert frac_exact > 0.95, frac_exact
    assert np.array_equal(oracle.acrobot_terminal(g["params"], g["y"][:, :4]), g["terminal"])


def test_reset_state_typing():
    """uniform(-1,1,4).astype(float32) * scale: float32 product for a scalar scale, float64 for a list"""
    g = np.load(GOLD)
    u = g["reset_u"]
    n = len(u)
    for scale, key in ((0.10, "reset_scalar"), (g["scale_vec"], "reset_vector")):
        o = oracle.AcrobotOracle(np.ones((1, 7)), np.zeros(n, np.int32), reset_scale=scale)
        o.reset_injected(u.T)
        assert np.array_equal(o.state.T, g[key]), key


def _np_step(prm, s, torque):
    """gymnasium AcrobotEnv.step physics, numpy, written from its public source"""
    def dsdt(y):
        return oracle.acrobot_dsdt(prm, y)[0]
    dt = 0.2
    dt2 = dt / 2.0
    y0 = np.append(s, torque)
    k1 = dsdt(y0); k2 = dsdt(y0 + dt2 * k1); k3 = dsdt(y0 + dt2 * k2); k4 = dsdt(y0 + dt * k3)
    ns = (y0 + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4))[:4]

    def wrap(x, m, M):
        diff = M - m
        while x > M:
            x = x - diff
        while x < m:
            x = x + diff
        return x
    ns[0] = wrap(ns[0], -np.pi, np.pi); ns[1] = wrap(ns[1], -np.pi, np.pi)
    ns[2] = min(max(ns[2], -4 * np.pi), 4 * np.pi); ns[3] = min(max(ns[3], -9 * np.pi), 9 * np.pi)
    return ns


def test_step_against_numpy_reading_of_gymnasium():
    rng = np.random.RandomState(0)
    g = np.load(GOLD)
    n, fs = 24, 3
    prm = g["params"][:n]
    o = oracle.AcrobotOracle(prm, np.arange(n, dtype=np.int32), frameskip=fs, max_steps=40)
    o.reset_injected(rng.random_sample((4, n)))
    ended = 0
    for t in range(120):
        a = rng.randint(0, 3, n).astype(np.int32)
        before = o.state.copy()
        nr_before = o.need_reset.copy()
        r = o.step_injected(a, rng.random_sample((4, n)), 1)      # NEXT_STEP: state after the step stays visible
        for i in range(n):
            if nr_before[i]:
                assert o.steps[i] == 0 and r["reward"][i] == 0     # this call was the env's reset
                continue
            s = before[:, i].copy()
            rew, term = 0.0, False
            for _ in range(fs):
                s = _np_step(prm[i], s, float(a[i] - 1))
                term = bool(-np.cos(s[0]) - np.cos(s[1] + s[0]) > prm[i][0])
                rew += 0.0 if term else -1.0
                if term:
                    break
            assert np.array_equal(o.state[:, i], s), (t, i)
            assert r["reward"][i] == np.float32(rew) and bool(r["terminated"][i]) == term
            exp_obs = np.array([np.cos(s[0]), np.sin(s[0]), np.cos(s[1]), np.sin(s[1]), s[2], s[3]], np.float32)
            assert np.array_equal(r["obs"][i], exp_obs)
        ended += int((r["terminated"] | r["truncated"]).sum())
    assert ended > 10
