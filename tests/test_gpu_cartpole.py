"""GPU parity: HIP CartPole (C-ABI) vs the CPU oracle on seeded batches (fp64 state and update as gymnasium keeps them;
device and host sincos differ in the last bit -> state to 1e-12, float32 observations to 1e-6) and vs the fp64 statement
of gymnasium's equations."""
import numpy as np
import pytest

import oracle
from xenoverse_amd.metacontrol import CartPoleVecEnv, sample_cartpole
from test_oracle_cartpole import gym_cartpole_step_f64

pytestmark = pytest.mark.gpu
MODES = {"disabled": 0, "next_step": 1, "same_step": 2}


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("mode", ["disabled", "next_step", "same_step"])
@pytest.mark.parametrize("frameskip", [1, 5])
def test_batch_vs_oracle(mode, frameskip):
    n, n_task = 1000, 50
    tasks = [sample_cartpole(seed=k) for k in range(n_task)]
    params = np.array([[t["gravity"], t["masscart"], t["masspole"], t["length"]] for t in tasks], np.float64)
    env_task = (np.arange(n) % n_task).astype(np.int32)
    env = CartPoleVecEnv(n, frameskip=frameskip, autoreset_mode=mode, max_steps=60)
    env.set_task(tasks, env_task_index=env_task)
    ora = oracle.CartPoleOracle(params, env_task, frameskip=frameskip, max_steps=60)
    rng = np.random.RandomState(1)
    u0 = rng.random_sample((4, n))
    assert np.array_equal(_np(env.reset_injected(u0)), ora.reset_injected(u0))
    ended = 0
    for t in range(150):
        a = rng.randint(0, 2, n).astype(np.int32)
        u = rng.random_sample((4, n))
        obs, r, term, trunc, info = env.step_injected(a, u)
        o = ora.step_injected(a, u, MODES[mode])
        assert np.allclose(_np(obs), o["obs"], rtol=1e-6, atol=1e-7)
        near = (np.abs(np.abs(ora.state[0]) - 2.4) < 1e-9) | (np.abs(np.abs(ora.state[2]) - 0.20943951023931953) < 1e-9)
        assert np.array_equal(_np(term).astype(np.uint8)[~near], o["terminated"][~near])
        assert np.array_equal(_np(trunc).astype(np.uint8), o["truncated"])
        assert np.array_equal(_np(r)[~near], o["reward"][~near])
        s, st, nr = env.get_state()
        ora.state[:] = _np(s); ora.steps[:] = _np(st); ora.need_reset[:] = _np(nr)   # re-sync: no compounding
        done = (o["terminated"] | o["truncated"]).astype(bool)
        ended += int(done.sum())
        if mode == "disabled" and done.any():
            ur = rng.random_sample((4, n))
            env.reset_injected(ur, mask=done.astype(np.uint8)); ora.reset_injected(ur, mask=done.astype(np.uint8))
    assert ended > 500
    assert env.check_errors() == 0
    env.close()


def test_free_running_and_fp64_equations():
    n = 256
    tasks = [sample_cartpole(seed=100 + k) for k in range(n)]
    params = np.array([[t["gravity"], t["masscart"], t["masspole"], t["length"]] for t in tasks], np.float64)
    seed, base = 31, 1 << 20
    env = CartPoleVecEnv(n, frameskip=1, seed=seed, env_id_base=base, autoreset_mode="same_step")
    env.set_task(tasks)
    ora = oracle.CartPoleOracle(params, np.arange(n), frameskip=1)
    tick = env.engine.tick
    obs, _ = env.reset()
    assert np.array_equal(_np(obs), ora.reset(seed, base, tick))
    rng = np.random.RandomState(2)
    for t in range(100):
        a = rng.randint(0, 2, n).astype(np.int32)
        before = _np(env.get_state()[0]).astype(np.float64)
        tick = env.engine.tick
        obs, r, term, trunc, info = env.step(a)
        if "_final_obs" in info:      # the done mask comes from the step launch (xv_*_step_info)
            assert np.array_equal(_np(info["_final_obs"]), _np(term) | _np(trunc))
        o = ora.step(seed, base, tick, a, 2)
        assert np.allclose(_np(obs), o["obs"], rtol=1e-6, atol=1e-7)
        fo = _np(info["final_obs"])
        for i in range(0, n, 29):
            s64, _, term64 = gym_cartpole_step_f64(before[:, i], a[i], *params[i])
            got = fo[i] if bool(term[i]) else _np(obs)[i]
            assert np.allclose(got, s64.astype(np.float32), rtol=1e-6, atol=1e-7) and bool(term[i]) == bool(term64)
        s, st, nr = env.get_state()
        ora.state[:] = _np(s); ora.steps[:] = _np(st)
    env.close()


@pytest.mark.parametrize("mode", ["same_step", "next_step"])
def test_fused_rollout_equals_single_steps(mode):
    """xv_cartpole_rollout: T steps in one launch = T calls of xv_cartpole_step bit for bit, through episode ends"""
    n, T = 1000, 120
    tasks = [sample_cartpole(seed=k) for k in range(50)]
    env_task = np.arange(n) % 50
    acts = np.random.RandomState(1).randint(0, 2, (T, n)).astype(np.int32)
    recs = []
    for fused in (False, True):
        env = CartPoleVecEnv(n, frameskip=2, seed=9, env_id_base=40, autoreset_mode=mode, max_steps=30)
        env.set_task(tasks, env_task_index=env_task)
        env.reset()
        if fused:
            a = env.rollout(acts[:50]); b = env.rollout(acts[50:])
            rec = {k: np.concatenate([_np(a[k]), _np(b[k])]) for k in a}
        else:
            rows = []
            for t in range(T):
                o, r, te, tr, info = env.step(acts[t])
                rows.append(dict(obs=_np(o), reward=_np(r), terminated=_np(te).astype(np.uint8),
                                 truncated=_np(tr).astype(np.uint8), final_obs=_np(info["final_obs"]) if "final_obs" in info
                                 else None))
            rec = {k: np.stack([row[k] for row in rows]) for k in rows[0] if rows[0][k] is not None}
        st = env.get_state()
        rec.update(state=_np(st[0]), steps=_np(st[1]), tick=np.int64(env.engine.tick))
        recs.append(rec)
        env.close()
    assert recs[0]["terminated"].sum() > 0 and recs[0]["truncated"].sum() > 0
    for k in recs[0]:
        assert np.array_equal(recs[0][k], recs[1][k]), k


@pytest.mark.parametrize("family", ["cartpole", "acrobot"])
@pytest.mark.parametrize("mode", ["same_step", "next_step"])
def test_copy_false_steps_equal_fresh_tensor_steps(family, mode):
    """copy=False (persistent outputs, cached pointers and views) returns what copy=True returns: same seed, same actions"""
    from xenoverse_amd.metacontrol import AcrobotVecEnv, sample_acrobot
    n = 300
    envs = []
    for copy in (True, False):
        if family == "cartpole":
            env = CartPoleVecEnv(n, frameskip=1, autoreset_mode=mode, max_steps=40, seed=5, copy=copy)
            env.set_task([sample_cartpole(seed=k) for k in range(10)], env_task_index=(np.arange(n) % 10).astype(np.int32))
        else:
            env = AcrobotVecEnv(n, frameskip=1, autoreset_mode=mode, max_steps=40, seed=5, copy=copy)
            env.set_task([sample_acrobot(seed=k) for k in range(10)], env_task_index=(np.arange(n) % 10).astype(np.int32))
        envs.append(env)
    o = [_np(e.reset()[0]).copy() for e in envs]
    assert np.array_equal(o[0], o[1])
    rng = np.random.RandomState(3)
    ended = 0
    for t in range(120):
        a = rng.randint(0, 2, n).astype(np.int32)
        outs = [e.step(a) for e in envs]
        got = [[_np(x).copy() for x in out[:4]] for out in outs]
        for x, y in zip(*got):
            assert np.array_equal(x, y)
        if mode == "same_step":
            done = got[1][2] | got[1][3]
            assert np.array_equal(_np(outs[1][4]["_final_obs"]), done) and np.array_equal(_np(outs[0][4]["_final_obs"]), done)
            assert np.array_equal(_np(outs[0][4]["final_obs"])[done], _np(outs[1][4]["final_obs"])[done])
            ended += int(done.sum())
    assert mode != "same_step" or ended > 100
    for e in envs:
        e.close()
