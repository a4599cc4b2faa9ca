"""Pin the POMDP / multi-token POMDP restatement to the reference (tests/golden/anymdptok_*.npz)."""
import numpy as np
import pytest

import oracle
from xenoverse_amd.anymdp import build_obs_tables, build_tables
from util import close_f32, golden_files, load_anymdp_tok_golden

FILES = golden_files("anymdptok_")


def test_golden_present():
    assert len(FILES) >= 3


def make(g, task, n=1):
    tab = build_tables([task])
    obs_cdf, n_obs, d_obs, d_act = build_obs_tables([task], tab["S"])
    return oracle.AnyMDPTokOracle(tab, np.zeros(n, np.int32), obs_cdf, d_act), d_obs, d_act


@pytest.mark.parametrize("path", FILES)
def test_trajectory_matches_reference(path):
    g, task = load_anymdp_tok_golden(path)
    o, d_obs, d_act = make(g, task)
    obs0 = o.tok_reset_injected([float(g["init_ur"])], g["init_uo"].reshape(d_obs, 1))
    assert np.array_equal(obs0[0], g["init_obs"]) and o.state[0] == g["init_state"]
    T = len(g["tr_r"])
    for t in range(T):
        if g["tr_set_steps"][t] >= 0:
            o.steps[0] = g["tr_set_steps"][t]
        obs, r, rgt, term, trunc, _ = o.tok_step_injected(
            g["tr_a"][t], g["tr_u"][t].reshape(d_act, 1), g["tr_z"][t].astype(np.float32).reshape(d_act, 1),
            g["tr_uo"][t].reshape(d_obs, 1), [0.0], np.zeros((d_obs, 1)), 0)
        assert np.array_equal(obs[0], g["tr_obs"][t]) and o.state[0] == g["tr_state"][t]
        assert term[0] == g["tr_term"][t] and trunc[0] == g["tr_trunc"][t] and o.steps[0] == g["tr_steps"][t]
        assert close_f32(r, g["tr_r"][t:t + 1]) and close_f32(rgt, g["tr_rgt"][t:t + 1])
        if term[0] or trunc[0]:
            ro = o.tok_reset_injected([g["tr_ur"][t]], g["tr_uor"][t].reshape(d_obs, 1))
            assert np.array_equal(ro[0], g["tr_reset_obs"][t])
    assert o.err_flags == 0


@pytest.mark.parametrize("path", FILES)
def test_same_step_autoreset(path):
    g, task = load_anymdp_tok_golden(path)
    o, d_obs, d_act = make(g, task)
    o.tok_reset_injected([float(g["init_ur"])], g["init_uo"].reshape(d_obs, 1))
    for t in range(len(g["tr_r"])):
        if g["tr_set_steps"][t] >= 0:
            o.steps[0] = g["tr_set_steps"][t]
        obs, r, rgt, term, trunc, fobs = o.tok_step_injected(
            g["tr_a"][t], g["tr_u"][t].reshape(d_act, 1), g["tr_z"][t].astype(np.float32).reshape(d_act, 1),
            g["tr_uo"][t].reshape(d_obs, 1), [g["tr_ur"][t]], g["tr_uor"][t].reshape(d_obs, 1), 2)
        if g["tr_term"][t] or g["tr_trunc"][t]:
            assert np.array_equal(obs[0], g["tr_reset_obs"][t]) and np.array_equal(fobs[0], g["tr_obs"][t])
        else:
            assert np.array_equal(obs[0], g["tr_obs"][t]) and np.all(fobs[0] == -1)
