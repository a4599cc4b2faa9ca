"""GPU parity of the POMDP / multi-token POMDP path: reference goldens and seeded batches vs the oracle."""
import numpy as np
import pytest
import torch

import oracle
from xenoverse_amd.anymdp import AnyMDPVecEnv, build_obs_tables, build_tables, to_blocked
from util import close_f32, golden_files, load_anymdp_tok_golden

pytestmark = pytest.mark.gpu
FILES = golden_files("anymdptok_")
MODES = {"disabled": 0, "next_step": 1, "same_step": 2}


def _dev_tables(tab, dev="cuda:0"):
    out = dict(S=tab["S"], A=tab["A"], s0_max=tab["s0_max"])
    tab = dict(tab, rows=to_blocked(tab["cdf"], tab["rs"]))
    for k in ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps"):
        v = np.ascontiguousarray(tab[k])
        out[k] = torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev)
    return out


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


@pytest.mark.parametrize("search", ["fence", "bucket"])
@pytest.mark.parametrize("mode", ["disabled", "next_step", "same_step"])
def test_batch_vs_oracle_injected_and_free_running(mode, search):
    """reference-sampled multi-token tasks; search = bucket runs the cooperative kernel (transition and observation
    bucket lines), fence the per-lane one"""
    tasks = [load_anymdp_tok_golden(p)[1] for p in FILES if "mtpomdp" in p]
    tab = build_tables(tasks)
    obs_cdf, n_obs, d_obs, d_act = build_obs_tables(tasks, tab["S"])
    n = 200
    env_task = (np.arange(n) % len(tasks)).astype(np.int32)
    seed, base = 77, 1000
    env = AnyMDPVecEnv(n, autoreset_mode=mode, seed=seed, env_id_base=base)
    env.set_task(tasks, env_task_index=env_task)
    env.set_search(search, n_bucket=16) if search == "bucket" else env.set_search(search)
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
        # info["steps"] and the done mask: written by the step launch itself on the free-running path (xv_anymdp_step_tokens_info)
        assert np.array_equal(_np(info["steps"]), ora.steps)
        if mode == "same_step":
            assert np.array_equal(_np(info["_final_obs"]), done)
        ended += int(done.sum())
        if mode == "disabled" and o[3].any():
            ur2, uor2 = rng.random_sample(n), rng.random_sample((d_obs, n))
            env.reset_tokens_injected(ur2, uor2, mask=o[3]); ora.tok_reset_injected(ur2, uor2, mask=o[3])
    assert ended > 50
    env.close()


@pytest.mark.parametrize("n_obs,d_obs,d_act,search", [(64, 3, 2, "fence"), (22, 2, 3, "fence"), (15, 2, 2, "fence"),
                                                     (64, 3, 2, "binary"), (64, 3, 2, "bucket"), (22, 2, 3, "bucket"),
                                                     (15, 1, 1, "bucket"), (200, 2, 2, "bucket"), (256, 2, 1, "bucket"),
                                                     (300, 2, 2, "bucket"), (64, 2, 2, "sparse-bucket"),
                                                     (40, 2, 2, "bucket-S300"), (40, 1, 1, "bucket-S300")])
def test_synthetic_observation_models_vs_oracle(n_obs, d_obs, d_act, search):
    """S=64, A=8 synthetic tasks with random observation models of several shapes (n_obs 64 / 22 / 15, up to 3
    tokens) against the oracle: injected draws incl. exact CDF entries, all three auto-reset modes in turn"""
    wide = search == "bucket-S300"          # S > 256: the transition lines' six-cut packing (16-bit ids) in the cooperative kernel
    S, A, n_task = (300, 4, 3) if wide else (64, 8, 6)
    tab = oracle.anymdp_synth(seed=17, task_index_base=0, n_task=n_task, S=S, A=A, s0_max=3)
    rng = np.random.RandomState(n_obs)
    sparse = search == "sparse-bucket"      # the reference's observation rows: a few live symbols, exact zeros between them
    search = "bucket" if sparse or wide else search
    w = rng.random_sample((n_task, d_obs, S, n_obs)) * (rng.random_sample((n_task, d_obs, S, n_obs)) < (0.08 if sparse else 0.5))
    if sparse:
        w[..., 0] += (w.sum(-1) == 0)
    else:
        w += 1e-3
    obs_cdf = np.cumsum(w, -1)
    obs_cdf = obs_cdf / obs_cdf[..., -1:]
    n = 333
    env_task = (np.arange(n) * 7 % n_task).astype(np.int32)
    for mode in ("same_step", "next_step", "disabled"):
        env = AnyMDPVecEnv(n, autoreset_mode=mode, seed=5)
        dev = {k: v for k, v in _dev_tables(tab).items()}
        env.set_task(dev, env_task_index=env_task)
        oc = torch.from_numpy(np.ascontiguousarray(obs_cdf)).cuda()
        from xenoverse_amd import _lib
        _lib.check(env.lib.xv_anymdp_set_observation_model(env._h, n_obs, d_obs, d_act, _lib.ptr(oc)))
        env._tok = (d_obs, d_act); env.task_type = "MTPOMDP"
        env._tobs = torch.zeros((n, d_obs), dtype=torch.int32, device="cuda")
        env._tfobs = torch.full((n, d_obs), -1, dtype=torch.int32, device="cuda")
        env.set_search(search)
        if search == "bucket":      # the cooperative kernel serves when the symbol ids fit the lines' bytes (n_obs <= 256)
            assert env.token_kernel == ("cooperative" if n_obs <= 256 else "per-lane")
            cen = env.bucket_census()
            assert cen["format"] == (2 if wide else 1)
            assert (cen["obs_lines"] > 0) == (n_obs <= 256)
            if sparse:
                assert cen["obs_lines_dirty"] == 0 and cen["obs_p_fallback"] == 0.0
        else:
            assert env.token_kernel == "per-lane"
        ora = oracle.AnyMDPTokOracle(tab, env_task, obs_cdf, d_act)
        ur0, uo0 = rng.random_sample(n), rng.random_sample((d_obs, n))
        assert np.array_equal(_np(env.reset_tokens_injected(ur0, uo0)), ora.tok_reset_injected(ur0, uo0))
        ended = 0
        for t in range(40):
            a = rng.randint(0, A, (n, d_act)).astype(np.int32)
            u, z = rng.random_sample((d_act, n)), rng.standard_normal((d_act, n)).astype(np.float32)
            uo, ur, uor = rng.random_sample((d_obs, n)), rng.random_sample(n), rng.random_sample((d_obs, n))
            k = rng.randint(0, n, 6)       # draws that hit a stored observation-CDF entry exactly
            uo[0, k] = np.minimum(obs_cdf[env_task[k], 0, ora.state[k], rng.randint(0, n_obs, 6)], np.nextafter(1.0, 0))
            obs, r, term, trunc, info = env.step_tokens_injected(a, u, z, uo, ur, uor)
            o = ora.tok_step_injected(a, u, z, uo, ur, uor, MODES[mode])
            assert np.array_equal(_np(obs), o[0]) and np.array_equal(_np(r), o[1])
            assert np.array_equal(_np(info["reward_gt"]), o[2])
            assert np.array_equal(_np(term).astype(np.uint8), o[3]) and np.array_equal(_np(trunc).astype(np.uint8), o[4])
            if mode == "same_step":
                assert np.array_equal(_np(info["final_obs"]), o[5])
            s, st, nr = env.get_state()
            assert np.array_equal(_np(s), ora.state) and np.array_equal(_np(st), ora.steps)
            assert np.array_equal(_np(nr), ora.need_reset)
            ended += int((o[3] | o[4]).sum())
            if mode == "disabled" and o[3].any():
                ur2, uor2 = rng.random_sample(n), rng.random_sample((d_obs, n))
                env.reset_tokens_injected(ur2, uor2, mask=o[3]); ora.tok_reset_injected(ur2, uor2, mask=o[3])
        assert ended > 30
        env.close()


@pytest.mark.parametrize("search,task_type", [("fence", "MTPOMDP"), ("bucket", "MTPOMDP"), ("fence", "POMDP")])
def test_step_tokens_many_equals_single_steps(search, task_type):
    """xv_anymdp_step_tokens_many (token steps issued from C over ring buffers) == the same number of step() calls: every
    output of the last ring cycle and the engine state, free-running draws (same seed, same ticks)"""
    S, A, n_task, n_obs = 64, 8, 6, 22
    d_obs, d_act = (2, 2) if task_type == "MTPOMDP" else (1, 1)
    tab = oracle.anymdp_synth(seed=17, task_index_base=0, n_task=n_task, S=S, A=A, s0_max=3)
    rng = np.random.RandomState(4)
    w = rng.random_sample((n_task, d_obs, S, n_obs)) + 1e-3
    oc = np.cumsum(w, -1); oc /= oc[..., -1:]; oc[..., -1] = 1.0
    n, P, K = 500, 6, 3 * 6 + 4
    env_task = (np.arange(n) * 5 % n_task).astype(np.int32)
    acts = rng.randint(0, A, (P, n, d_act)).astype(np.int32)
    res = []
    for many in (True, False):
        env = AnyMDPVecEnv(n, autoreset_mode="same_step", seed=9)
        env.set_task(dict(_dev_tables(tab), obs_cdf=np.ascontiguousarray(oc), n_obs=n_obs, d_obs=d_obs, d_act=d_act, task_type=task_type),
                     env_task_index=env_task)
        env.set_search(search)
        env.reset()
        if many:
            ring = env.step_tokens_many(K, acts if d_act > 1 else acts[:, :, 0])
            rec = {k: _np(v).copy() for k, v in ring.items()}
        else:
            rows = [None] * P
            for k in range(K):
                a = acts[k % P] if d_act > 1 else acts[k % P][:, 0]
                o, r, te, tr, info = env.step(a)
                rows[k % P] = dict(obs=_np(o).reshape(n, d_obs), reward=_np(r), reward_gt=_np(info["reward_gt"]),
                                   terminated=_np(te).astype(np.uint8), truncated=_np(tr).astype(np.uint8),
                                   final_obs=_np(info["final_obs"]).reshape(n, d_obs))
            rec = {k: np.stack([row[k] for row in rows]) for k in rows[0]}
        s, st, nr = env.get_state()
        rec["state"], rec["steps"], rec["tick"] = _np(s), _np(st), np.array([env.engine.tick])
        assert env.check_errors() == 0
        res.append(rec)
        env.close()
    done = (res[1]["terminated"] | res[1]["truncated"]).astype(bool)
    for k in res[0]:
        if k == "final_obs":      # written for finished envs (the per-lane kernel leaves the other rows alone)
            assert np.array_equal(res[0][k][done], res[1][k][done]), k
        else:
            assert np.array_equal(res[0][k], res[1][k]), k
    assert done.sum() > 20


@pytest.mark.parametrize("mode", ["same_step", "next_step", "disabled"])
@pytest.mark.parametrize("kind", ["pomdp", "mtpomdp"])
def test_copy_false_token_steps_equal_fresh_tensor_steps(kind, mode):
    """copy=False (persistent outputs, cached pointers and views, one launch per step) returns what copy=True returns — same
    seed, same actions — and info["steps"] / the done mask come from the step launch in both"""
    tasks = [load_anymdp_tok_golden(p)[1] for p in FILES if ("mtpomdp" in p) == (kind == "mtpomdp")]
    n = 150
    env_task = (np.arange(n) % len(tasks)).astype(np.int32)
    envs = []
    for copy in (True, False):
        env = AnyMDPVecEnv(n, autoreset_mode=mode, seed=21, copy=copy)
        env.set_task(tasks, env_task_index=env_task)
        env.set_search("bucket", n_bucket=16)
        envs.append(env)
    o0 = [_np(e.reset()[0]).copy() for e in envs]
    assert np.array_equal(o0[0], o0[1])
    A = int(envs[0].na)
    rng = np.random.RandomState(2)
    d_act = envs[0]._tok[1]
    for t in range(60):
        a = rng.randint(0, A, (n, d_act)).astype(np.int32)
        outs = [e.step(a if d_act > 1 else a[:, 0]) for e in envs]
        got = [[_np(x).copy() for x in o[:4]] + [_np(o[4]["steps"]).copy(), _np(o[4]["reward_gt"]).copy()] for o in outs]
        for x, y in zip(*got):
            assert np.array_equal(x, y)
        if mode == "same_step":
            assert np.array_equal(_np(outs[0][4]["final_obs"]), _np(outs[1][4]["final_obs"]))
            assert np.array_equal(_np(outs[0][4]["_final_obs"]), _np(outs[1][4]["_final_obs"]))
            assert np.array_equal(_np(outs[1][4]["_final_obs"]), got[1][2] | got[1][3])
        if mode == "disabled":
            term = got[0][2].astype(bool)
            if term.any():
                for e in envs:
                    e.reset_tokens_injected(np.full(n, 0.25), np.full((e._tok[0], n), 0.5), mask=term.astype(np.uint8))
    for e in envs:
        e.close()
