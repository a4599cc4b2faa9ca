"""Pin oracle/xeno_oracle.c (AnyMDP) to the reference's own outputs (tests/golden/anymdp_*.npz, made by
oracle/gen_golden.py from /root/reference/xenoverse/anymdp/anymdp_env.py)."""
import numpy as np
import pytest

import oracle
from xenoverse_amd.anymdp.tables import build_tables, row_cdf
from util import golden_files, load_anymdp_golden, close_f32

FILES = golden_files("anymdp_")
DISABLED, NEXT_STEP, SAME_STEP = 0, 1, 2


def test_golden_present():
    assert len(FILES) >= 4


@pytest.mark.parametrize("path", FILES)
def test_single_step_tuples(path):
    g, task = load_anymdp_golden(path)
    tab = build_tables([task])
    n = len(g["ss_s"])
    o = oracle.AnyMDPOracle(tab, np.zeros(n, np.int32))
    o.state[:] = g["ss_s"]
    o.steps[:] = 0
    o.need_reset[:] = 0
    obs, r, rgt, term, trunc, _ = o.step_injected(g["ss_a"], g["ss_u"], g["ss_z"].astype(np.float32),
                                                   np.zeros(n), DISABLED)
    assert np.array_equal(o.state, g["ss_next"])                       # integer path: bit-exact
    assert np.array_equal(obs, g["state_mapping"][g["ss_next"]])
    assert np.array_equal(term, g["ss_term"])
    assert np.array_equal(rgt, g["ss_rgt"].astype(np.float32))         # table lookup of an f32-rounded table
    assert close_f32(r, g["ss_r"])                                     # float path: 1e-5 rel
    assert o.err_flags == 0


@pytest.mark.parametrize("path", FILES)
def test_choice_equals_upper_bound_on_host_cdf(path):
    """numpy.random.choice(n, p=row) == searchsorted(cumsum(row)/cumsum(row)[-1], u, 'right')"""
    g, task = load_anymdp_golden(path)
    c = row_cdf(task["transition"])
    rows = c[g["ss_s"], g["ss_a"]]
    idx = np.array([np.searchsorted(rows[i], g["ss_u"][i], side="right") for i in range(len(rows))])
    assert np.array_equal(idx, g["ss_next"])


@pytest.mark.parametrize("path", FILES)
def test_reset_draws(path):
    g, task = load_anymdp_golden(path)
    tab = build_tables([task])
    n = len(g["reset_u"])
    o = oracle.AnyMDPOracle(tab, np.zeros(n, np.int32))
    obs = o.reset_injected(g["reset_u"])
    assert np.array_equal(o.state, g["reset_state"])
    assert np.array_equal(obs, g["reset_obs"])
    assert np.all(o.steps == 0) and np.all(o.need_reset == 0)


def _replay(g, task, mode):
    tab = build_tables([task])
    o = oracle.AnyMDPOracle(tab, np.zeros(1, np.int32))
    obs0 = o.reset_injected(np.array([float(g["init_u"])]))
    assert o.state[0] == g["init_state"] and obs0[0] == g["init_obs"]
    T = len(g["tr_a"])
    out = {k: [] for k in ("obs", "r", "rgt", "term", "trunc", "steps", "state", "fobs", "tgt")}
    for t in range(T):
        if g["tr_set_steps"][t] >= 0:
            o.steps[0] = g["tr_set_steps"][t]
        a = np.array([g["tr_a"][t]], np.int32)
        obs, r, rgt, term, trunc, fobs = o.step_injected(
            a, [g["tr_u"][t]], np.array([g["tr_z"][t]], np.float32), [g["tr_ur"][t]], mode)
        steps_after = int(o.steps[0])
        if mode == DISABLED:
            out["tgt"].append(o.transition_gt(a)[0].copy())
            out["steps"].append(steps_after)
            out["state"].append(int(o.state[0]))
            if term[0] or trunc[0]:   # the caller resets, as the reference's rollout loops do
                ro = o.reset_injected(np.array([g["tr_ur"][t]]))
                assert ro[0] == g["tr_reset_obs"][t]
        out["obs"].append(int(obs[0])); out["r"].append(float(r[0])); out["rgt"].append(float(rgt[0]))
        out["term"].append(int(term[0])); out["trunc"].append(int(trunc[0])); out["fobs"].append(int(fobs[0]))
    assert o.err_flags == 0
    return {k: np.array(v) for k, v in out.items()}


@pytest.mark.parametrize("path", FILES)
def test_trajectory_disabled_mode_matches_reference(path):
    g, task = load_anymdp_golden(path)
    out = _replay(g, task, DISABLED)
    assert np.array_equal(out["obs"], g["tr_obs"])
    assert np.array_equal(out["state"], g["tr_state"])
    assert np.array_equal(out["steps"], g["tr_steps"])
    assert np.array_equal(out["term"], g["tr_term"])
    assert np.array_equal(out["trunc"], g["tr_trunc"])
    assert g["tr_trunc"].sum() >= 1          # the fixture crosses the max_steps boundary
    assert np.array_equal(out["rgt"].astype(np.float32), g["tr_rgt"].astype(np.float32))
    assert close_f32(out["r"], g["tr_r"])
    # info["transition_gt"]: differences of an fp64 CDF vs the pmf itself
    k = len(g["tr_tgt"])
    assert np.max(np.abs(out["tgt"][:k] - g["tr_tgt"])) < 1e-12


@pytest.mark.parametrize("path", FILES)
def test_trajectory_same_step_autoreset(path):
    g, task = load_anymdp_golden(path)
    out = _replay(g, task, SAME_STEP)
    done = (g["tr_term"] | g["tr_trunc"]).astype(bool)
    assert np.array_equal(out["term"], g["tr_term"]) and np.array_equal(out["trunc"], g["tr_trunc"])
    assert np.array_equal(out["obs"][~done], g["tr_obs"][~done])
    assert np.array_equal(out["obs"][done], g["tr_reset_obs"][done])     # reset obs returned in the same step
    assert np.array_equal(out["fobs"][done], g["tr_obs"][done])          # terminal obs kept in final_obs
    assert np.all(out["fobs"][~done] == -1)


def test_next_step_autoreset_semantics():
    g, task = load_anymdp_golden(FILES[0])
    tab = build_tables([task])
    o = oracle.AnyMDPOracle(tab, np.zeros(1, np.int32))
    o.reset_injected(np.array([float(g["init_u"])]))
    t = 0
    while True:
        a = np.array([g["tr_a"][t]], np.int32)
        obs, r, rgt, term, trunc, _ = o.step_injected(a, [g["tr_u"][t]], np.array([g["tr_z"][t]], np.float32),
                                                      [g["tr_ur"][t]], NEXT_STEP)
        if term[0] or trunc[0]:
            break
        t += 1
    assert obs[0] == g["tr_obs"][t] and o.need_reset[0] == 1
    # the following call ignores its action and returns the reset observation, reward 0, flags False
    obs2, r2, rgt2, term2, trunc2, _ = o.step_injected(np.array([0], np.int32), [0.5], np.zeros(1, np.float32),
                                                       [g["tr_ur"][t]], NEXT_STEP)
    assert obs2[0] == g["tr_reset_obs"][t] and r2[0] == 0 and not term2[0] and not trunc2[0]
    assert o.steps[0] == 0 and o.need_reset[0] == 0


def test_error_flags():
    g, task = load_anymdp_golden(FILES[0])
    tab = build_tables([task])
    o = oracle.AnyMDPOracle(tab, np.zeros(2, np.int32))
    o.reset_injected(np.array([0.1, 0.1]))
    o.step_injected(np.array([0, tab["A"]], np.int32), [0.5, 0.5], np.zeros(2, np.float32), [0.5, 0.5], DISABLED)
    assert o.err_flags & 1                       # action out of range (anymdp_env.py:97)
    s_term = int(task["s_e"][0])
    o2 = oracle.AnyMDPOracle(tab, np.zeros(1, np.int32))
    o2.state[0] = s_term; o2.need_reset[0] = 0
    obs, r, rgt, term, trunc, _ = o2.step_injected(np.array([0], np.int32), [0.5], np.zeros(1, np.float32), [0.5], DISABLED)
    assert o2.err_flags & 2 and term[0] == 1 and o2.state[0] == s_term   # anymdp_env.py:95-96
