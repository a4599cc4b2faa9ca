"""GPU parity of the POMDP / multi-token POMDP path: reference goldens and seeded batches vs the oracle."""
import numpy as np
import pytest

import oracle
from xenoverse_amd.anymdp import AnyMDPVecEnv, build_obs_tables, build_tables
from util import close_f32, golden_files, load_anymdp_tok_golden

pytestmark = pytest.mark.gpu
FILES = golden_files("anymdptok_")
MODES = {"disabled": 0, "next_step": 1, "same_step": 2}


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("path", FILES)
def test_golden_trajectory(path):
    g, task = load_anymdp_tok_golden(path)
    mt = bool(g["is_mt"])
    d_obs, d_act = int(g["do"]), int(g["da"])
    env = AnyMDPVecEnv(1, autoreset_mode="disabled")
    env.set_task(task)
    assert env.task_type == ("MTPOMDP" if mt else "POMDP")
    o0 = env.reset_tokens_injected([float(g["init_ur"])], g["init_uo"].reshape(d_obs, 1))
    assert np.array_equal(np.atleast_1d(_np(o0)[0]), g["init_obs"])
    T = min(len(g["tr_r"]), 400)
    T = max(T, int(np.argmax(g["tr_trunc"])) + 2)
    for t in range(T):
        if g["tr_set_steps"][t] >= 0:
            env.set_state(steps=[int(g["tr_set_steps"][t])])
        a = g["tr_a"][t].reshape(1, d_act)
        obs, r, term, trunc, info = env.step_tokens_injected(
            a, g["tr_u"][t].reshape(d_act, 1), g["tr_z"][t].reshape(d_act, 1), g["tr_uo"][t].reshape(d_obs, 1),
            [0.0], np.zeros((d_obs, 1)))
        assert np.array_equal(np.atleast_1d(_np(obs)[0]), g["tr_obs"][t])
        assert bool(term[0]) == bool(g["tr_term"][t]) and bool(trunc[0]) == bool(g["tr_trunc"][t])
        assert int(info["steps"][0]) == g["tr_steps"][t] and int(env.inner_state[0]) == g["tr_state"][t]
        assert close_f32(_np(r), g["tr_r"][t:t + 1]) and close_f32(_np(info["reward_gt"]), g["tr_rgt"][t:t + 1])
        if term[0] or trunc[0]:
            ro = env.reset_tokens_injected([g["tr_ur"][t]], g["tr_uor"][t].reshape(d_obs, 1))
            assert np.array_equal(np.atleast_1d(_np(ro)[0]), g["tr_reset_obs"][t])
    env.close()


@pytest.mark.parametrize("mode", ["disabled", "next_step", "same_step"])
def test_batch_vs_oracle_injected_and_free_running(mode):
    tasks = [load_anymdp_tok_golden(p)[1] for p in FILES if "mtpomdp" in p]
    tab = build_tables(tasks)
    obs_cdf, n_obs, d_obs, d_act = build_obs_tables(tasks, tab["S"])
    n = 200
    env_task = (np.arange(n) % len(tasks)).astype(np.int32)
    seed, base = 77, 1000
    env = AnyMDPVecEnv(n, autoreset_mode=mode, seed=seed, env_id_base=base)
    env.set_task(tasks, env_task_index=env_task)
    ora = oracle.AnyMDPTokOracle(tab, env_task, obs_cdf, d_act)
    tick = env.engine.tick
    o0, _ = env.reset()
    assert np.array_equal(_np(o0), ora.tok_reset(seed, base, tick))
    rng = np.random.RandomState(9)
    ended = 0
    for t in range(160):
        a = rng.randint(0, tab["A"], (n, d_act)).astype(np.int32)
        if t % 2:
            tick = env.engine.tick
            obs, r, term, trunc, info = env.step(a)
            o = ora.tok_step(seed, base, tick, a, MODES[mode])
            assert close_f32(_np(r), o[1], 1e-5, 4e-6)
        else:
            u, z = rng.random_sample((d_act, n)), rng.standard_normal((d_act, n)).astype(np.float32)
            uo, ur, uor = rng.random_sample((d_obs, n)), rng.random_sample(n), rng.random_sample((d_obs, n))
            obs, r, term, trunc, info = env.step_tokens_injected(a, u, z, uo, ur, uor)
            o = ora.tok_step_injected(a, u, z, uo, ur, uor, MODES[mode])
            assert np.array_equal(_np(r), o[1])
        assert np.array_equal(_np(obs), o[0]) and np.array_equal(_np(info["reward_gt"]), o[2])
        assert np.array_equal(_np(term).astype(np.uint8), o[3]) and np.array_equal(_np(trunc).astype(np.uint8), o[4])
        if mode == "same_step":
            assert np.array_equal(_np(info["final_obs"]), o[5])
        s, st, nr = env.get_state()
        assert np.array_equal(_np(s), ora.state) and np.array_equal(_np(st), ora.steps) and np.array_equal(_np(nr), ora.need_reset)
        done = (o[3] | o[4]).astype(bool)
        ended += int(done.sum())
        if mode == "disabled" and o[3].any():
            ur2, uor2 = rng.random_sample(n), rng.random_sample((d_obs, n))
            env.reset_tokens_injected(ur2, uor2, mask=o[3]); ora.tok_reset_injected(ur2, uor2, mask=o[3])
    assert ended > 50
    env.close()
