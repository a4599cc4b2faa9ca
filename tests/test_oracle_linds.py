"""Pin oracle/xeno_oracle.c (LinDS, fp32) to the reference's own outputs (tests/golden/linds_*.npz, made by
oracle/gen_golden.py from /root/reference/xenoverse/linds/linds_env.py, fp64).  Tolerance: north_star's 1e-5
relative for float dynamics, checked PER STEP from the reference's own state (fp32 vs fp64 trajectories of a
dynamical system drift apart over hundreds of steps; a second test bounds that drift)."""
import numpy as np
import pytest

import oracle
from xenoverse_amd.linds.tables import build_dynamics_matrices, build_tables
from util import close_rel, golden_files, load_linds_golden

FILES = golden_files("linds_")
DISABLED, NEXT_STEP, SAME_STEP = 0, 1, 2


def test_golden_present():
    assert len(FILES) >= 5


@pytest.mark.parametrize("path", FILES)
def test_host_zoh_discretisation_matches_reference_expm(path):
    g, task = load_linds_golden(path)
    phi, gam, xt = build_dynamics_matrices(task["ld_A"], task["ld_B"], task["ld_X"], float(g["dt"]))
    assert np.max(np.abs(phi - g["ref_phi"])) < 1e-13
    assert np.max(np.abs(gam - g["ref_gamma"])) < 1e-13
    assert np.max(np.abs(xt - g["ref_xt"])) < 1e-15


def _before_states(g):
    """state and step counter BEFORE step t of the golden trajectory"""
    T = len(g["tr_reward"])
    xb = np.zeros_like(g["tr_x"])
    sb = np.zeros(T, np.int64)
    x, s = g["initial_states"][int(g["init_idx"])], 0
    for t in range(T):
        xb[t], sb[t] = x, s
        if g["tr_term"][t] or g["tr_trunc"][t]:
            x, s = g["initial_states"][int(g["tr_reset_idx"][t])], 0
        else:
            x, s = g["tr_x"][t], g["tr_steps"][t]
    return xb, sb


@pytest.mark.parametrize("path", FILES)
def test_every_step_from_reference_state(path):
    g, task = load_linds_golden(path)
    tab = build_tables([task])
    T, ns = g["tr_x"].shape
    xb, sb = _before_states(g)
    o = oracle.LinDSOracle(tab, np.zeros(T, np.int32))       # one env per recorded step
    o.x[:ns, :] = xb.T.astype(np.float32)
    o.steps[:] = sb
    o.need_reset[:] = 0
    z = np.zeros((tab["NS"], T), np.float32)
    z[:ns] = g["tr_z"].T
    out = o.step_injected(g["tr_action"], z, np.zeros(T, np.int32), DISABLED)
    assert np.array_equal(out["terminated"], g["tr_term"])       # flags: exact
    assert np.array_equal(out["truncated"], g["tr_trunc"])
    assert np.array_equal(o.steps, g["tr_steps"])
    assert close_rel(o.x[:ns].T, g["tr_x"])                      # float dynamics: 1e-5 rel
    assert close_rel(out["obs"], g["tr_obs"])
    assert close_rel(out["cmd"], g["tr_cmd"])
    assert close_rel(out["error"], g["tr_error"])
    assert close_rel(out["reward"], g["tr_reward"])
    assert g["tr_term"].sum() + g["tr_trunc"].sum() > 0


@pytest.mark.parametrize("path", FILES)
def test_reset_outputs(path):
    g, task = load_linds_golden(path)
    tab = build_tables([task])
    done = np.nonzero(g["tr_term"] | g["tr_trunc"])[0]
    idx = np.concatenate([[int(g["init_idx"])], g["tr_reset_idx"][done]]).astype(np.int32)
    o = oracle.LinDSOracle(tab, np.zeros(len(idx), np.int32))
    out = o.reset_injected(idx)
    ref_obs = np.concatenate([g["init_obs"][None], g["tr_reset_obs"][done]])
    ref_cmd = np.concatenate([g["init_cmd"][None], g["tr_reset_cmd"][done]])
    ref_err = np.concatenate([[float(g["init_err"])], g["tr_reset_err"][done]])
    assert close_rel(out["obs"], ref_obs) and close_rel(out["cmd"], ref_cmd) and close_rel(out["error"], ref_err)
    assert np.all(o.steps == 0)


@pytest.mark.parametrize("path", FILES)
def test_free_running_fp32_trajectory_stays_close(path):
    """the whole trajectory in fp32 without re-injecting the reference state: bounded drift, same episode ends"""
    g, task = load_linds_golden(path)
    tab = build_tables([task])
    T, ns = g["tr_x"].shape
    o = oracle.LinDSOracle(tab, np.zeros(1, np.int32))
    o.reset_injected([int(g["init_idx"])])
    worst = 0.0
    for t in range(T):
        z = np.zeros((tab["NS"], 1), np.float32)
        z[:ns, 0] = g["tr_z"][t]
        nxt = max(int(g["tr_reset_idx"][t]), 0)
        out = o.step_injected(g["tr_action"][t:t + 1], z, [nxt], DISABLED)
        assert out["terminated"][0] == g["tr_term"][t] and out["truncated"][0] == g["tr_trunc"][t]
        worst = max(worst, float(np.max(np.abs(o.x[:ns, 0] - g["tr_x"][t]) / (np.abs(g["tr_x"][t]) + 1.0))))
        if g["tr_term"][t] or g["tr_trunc"][t]:
            o.reset_injected([nxt])
    assert worst < 2e-4


def test_autoreset_modes():
    g, task = load_linds_golden(FILES[1])
    tab = build_tables([task])
    for mode in (NEXT_STEP, SAME_STEP):
        o = oracle.LinDSOracle(tab, np.zeros(1, np.int32))
        o.reset_injected([int(g["init_idx"])])
        T, ns = g["tr_x"].shape
        t = 0
        while True:
            z = np.zeros((tab["NS"], 1), np.float32); z[:ns, 0] = g["tr_z"][t]
            nxt = max(int(g["tr_reset_idx"][t]), 0)
            out = o.step_injected(g["tr_action"][t:t + 1], z, [nxt], mode)
            if out["terminated"][0] or out["truncated"][0]:
                break
            t += 1
        if mode == SAME_STEP:
            assert close_rel(out["final_obs"][0], g["tr_obs"][t]) and close_rel(out["obs"][0], g["tr_reset_obs"][t])
            assert o.steps[0] == 0
        else:
            assert close_rel(out["obs"][0], g["tr_obs"][t]) and o.need_reset[0] == 1
            out2 = o.step_injected(np.zeros((1, 8), np.float32), np.zeros((tab["NS"], 1), np.float32), [nxt], mode)
            assert close_rel(out2["obs"][0], g["tr_reset_obs"][t]) and out2["reward"][0] == 0
            assert not out2["terminated"][0] and o.steps[0] == 0 and o.need_reset[0] == 0
