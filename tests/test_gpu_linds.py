"""GPU parity: the HIP LinDS path (through the C-ABI) vs the reference's golden trajectories (1e-5 rel, per
step from the reference's own state) and vs the CPU oracle on seeded batches (state/observation path
bit-exact: same fp32 fmaf chains in the same order; command/reward within 1e-5 because sinf differs)."""
import numpy as np
import pytest
import torch

import oracle
from xenoverse_amd.linds import LinDSVecEnv, build_tables, pad_tables
from util import close_rel, golden_files, load_linds_golden
from test_oracle_linds import _before_states

pytestmark = pytest.mark.gpu
FILES = golden_files("linds_")
MODES = {"disabled": 0, "next_step": 1, "same_step": 2}


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("kernel", ["mfma", "scalar"])
@pytest.mark.parametrize("path", FILES)
def test_golden_every_step_from_reference_state(path, kernel):
    g, task = load_linds_golden(path)
    T, ns = g["tr_x"].shape
    xb, sb = _before_states(g)
    env = LinDSVecEnv(T, autoreset_mode="disabled")
    env.set_task(task)
    env.set_path(kernel)
    env.set_state(x=xb.T.astype(np.float32), steps=sb, need_reset=np.zeros(T))
    obs, r, term, trunc, info = env.step_injected(g["tr_action"], g["tr_z"].T, np.zeros(T))
    x, st, _ = env.get_state()
    assert np.array_equal(_np(term).astype(np.uint8), g["tr_term"])
    assert np.array_equal(_np(trunc).astype(np.uint8), g["tr_trunc"])
    assert np.array_equal(_np(st), g["tr_steps"])
    assert close_rel(_np(x)[:ns].T, g["tr_x"])
    assert close_rel(_np(obs), g["tr_obs"])
    assert close_rel(_np(info["command"]), g["tr_cmd"])
    assert close_rel(_np(info["error"]), g["tr_error"])
    assert close_rel(_np(r), g["tr_reward"])
    assert env.check_errors() == 0
    env.close()


@pytest.mark.parametrize("path", FILES[:2])
def test_golden_reset(path):
    g, task = load_linds_golden(path)
    done = np.nonzero(g["tr_term"] | g["tr_trunc"])[0]
    idx = np.concatenate([[int(g["init_idx"])], g["tr_reset_idx"][done]]).astype(np.int32)
    env = LinDSVecEnv(len(idx), autoreset_mode="disabled")
    env.set_task(task)
    obs, info = env.reset_injected(idx)
    assert close_rel(_np(obs), np.concatenate([g["init_obs"][None], g["tr_reset_obs"][done]]))
    assert close_rel(_np(info["command"]), np.concatenate([g["init_cmd"][None], g["tr_reset_cmd"][done]]))
    assert close_rel(_np(info["error"]), np.concatenate([[float(g["init_err"])], g["tr_reset_err"][done]]))
    env.close()


def _batch(n_per_task, files):
    tasks = [load_linds_golden(p)[1] for p in files]
    for t in tasks:
        t["max_steps"] = min(int(t["max_steps"]), 40)     # make episodes end inside the test
    tab = pad_tables(build_tables(tasks))
    env_task = np.repeat(np.arange(len(tasks), dtype=np.int32), n_per_task)
    return tasks, tab, env_task


def _cmp(dev, ora, exact_state=True):
    obs, r, term, trunc, info = dev
    assert np.array_equal(_np(term).astype(np.uint8), ora["terminated"])
    assert np.array_equal(_np(trunc).astype(np.uint8), ora["truncated"])
    no = _np(obs).shape[1]
    if exact_state:
        assert np.array_equal(_np(obs), ora["obs"][:, :no])
    else:
        assert close_rel(_np(obs), ora["obs"][:, :no])
    assert close_rel(_np(info["command"]), ora["cmd"][:, :no], 1e-5, 2e-6)
    assert close_rel(_np(info["error"]), ora["error"], 1e-5, 2e-6)
    assert close_rel(_np(r), ora["reward"], 1e-5, 2e-6)


@pytest.mark.parametrize("mode", ["disabled", "next_step", "same_step"])
@pytest.mark.parametrize("layout,path", [("grouped64", "mfma"), ("grouped64", "scalar"), ("grouped32", "mfma"),
                                         ("grouped32", "scalar"), ("ragged", "mfma"), ("mixed", "mfma"),
                                         ("mixed", "scalar"), ("odd_counts", "mfma")])
def test_batch_injected_vs_oracle(mode, layout, path):
    tasks, tab, env_task = _batch(64 if layout not in ("grouped32", "ragged") else 32, FILES[:4])
    rng = np.random.RandomState(3)
    if layout == "mixed":
        rng.shuffle(env_task)              # lanes of one wave hold different tasks: the waterfall path
    if layout == "ragged":
        env_task = env_task[:-13]          # last tile is partial
    if layout == "odd_counts":             # tasks with 1 / 33 / 64 / 7 envs, interleaved: padded tiles in the slot layout
        env_task = np.array([0] * 1 + [1] * 33 + [2] * 64 + [3] * 7, np.int32)
        rng.shuffle(env_task)
    n = len(env_task)
    env = LinDSVecEnv(n, autoreset_mode=mode)
    env.set_task(tasks, env_task_index=env_task)
    env.set_path(path)
    ora = oracle.LinDSOracle(tab, env_task)
    n_init = tab["ints"][env_task, 2]
    idx0 = (rng.random_sample(n) * n_init).astype(np.int32)
    obs, info = env.reset_injected(idx0)
    o0 = ora.reset_injected(idx0)
    assert np.array_equal(_np(obs), o0["obs"][:, :16])
    ended = 0
    for t in range(90):
        a = rng.uniform(-1.4, 1.4, (n, 8)).astype(np.float32)
        z = rng.standard_normal((tab["NS"], n)).astype(np.float32)
        idx = (rng.random_sample(n) * n_init).astype(np.int32)
        d = env.step_injected(a, z, idx)
        o = ora.step_injected(a, z, idx, MODES[mode])
        _cmp(d, o)
        x, st, nr = env.get_state()
        assert np.array_equal(_np(x), ora.x) and np.array_equal(_np(st), ora.steps)
        assert np.array_equal(_np(nr), ora.need_reset)
        ended += int((o["terminated"] | o["truncated"]).sum())
        if mode == "same_step":
            done = (o["terminated"] | o["truncated"]).astype(bool)
            assert np.array_equal(_np(d[4]["final_obs"])[done], o["final_obs"][done][:, :16])
        if mode == "disabled":
            m = (o["terminated"] | o["truncated"]).astype(np.uint8)
            if m.any():
                env.reset_injected(idx, mask=m)
                ora.reset_injected(idx, mask=m)
    assert ended > 20
    env.close()


@pytest.mark.parametrize("path", ["mfma", "scalar"])
def test_command_table_equals_direct_evaluation(path):
    """the per-task table of get_inner_cmd values (built at set_task) gives bit-identical steps to evaluating the
    Fourier terms in the kernel, also past the end of the table (auto-reset disabled, envs stepped on after
    truncation) and right after resets"""
    tasks, tab, env_task = _batch(64, FILES[:4])
    n = len(env_task)
    rng = np.random.RandomState(8)
    acts = rng.uniform(-0.3, 0.3, (70, n, 8)).astype(np.float32)
    zs = rng.standard_normal((70, tab["NS"], n)).astype(np.float32)
    n_init = tab["ints"][env_task, 2]
    idxs = (rng.random_sample((70, n)) * n_init).astype(np.int32)
    outs = []
    for table in (True, False):
        env = LinDSVecEnv(n, autoreset_mode="disabled")
        env.set_task(tasks, env_task_index=env_task)
        env.set_path(path)
        env.set_command_table(table)
        obs, info = env.reset_injected(idxs[0])
        rec = [_np(obs), _np(info["command"])]
        for t in range(70):            # max_steps is 40: the last 30 steps run past the table
            o, r, te, tr, info = env.step_injected(acts[t], zs[t], idxs[t])
            rec += [_np(o), _np(r), _np(info["command"]), _np(info["error"]), _np(te), _np(tr)]
            if t == 20:
                m = np.zeros(n, np.uint8); m[::3] = 1
                env.reset_injected(idxs[t], mask=m)
        assert int(_np(env.get_state()[1]).max()) > 45
        outs.append(rec)
        env.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_free_running_philox_vs_oracle():
    tasks, tab, env_task = _batch(64, FILES[:4])
    n = len(env_task)
    seed, base = 987654321, 5000
    env = LinDSVecEnv(n, autoreset_mode="same_step", seed=seed, env_id_base=base)
    env.set_task(tasks, env_task_index=env_task)
    ora = oracle.LinDSOracle(tab, env_task)
    tick = env.engine.tick
    obs, info = env.reset()
    o0 = ora.reset(seed, base, tick)
    assert np.array_equal(_np(obs), o0["obs"][:, :16])
    rng = np.random.RandomState(5)
    for t in range(60):
        a = rng.uniform(-1.2, 1.2, (n, 8)).astype(np.float32)
        tick = env.engine.tick
        d = env.step(a)
        o = ora.step(seed, base, tick, a, 2)
        _cmp(d, o, exact_state=False)      # Box-Muller: device logf/sincospif vs host double -> 1e-5
        x, st, _ = env.get_state()
        assert close_rel(_np(x), ora.x, 1e-5, 2e-6) and np.array_equal(_np(st), ora.steps)
        ora.x[:] = _np(x)                  # re-sync so tolerance does not compound over the trajectory
    env.close()


@pytest.mark.parametrize("path", ["mfma", "scalar"])
def test_state_dim_32_task(path):
    g, task = load_linds_golden([f for f in FILES if "32x8x8" in f][0])
    tab = pad_tables(build_tables([task]))
    assert tab["NS"] == 32
    n = 128
    env = LinDSVecEnv(n, autoreset_mode="same_step")
    env.set_task(task)
    env.set_path(path)
    ora = oracle.LinDSOracle(tab, np.zeros(n, np.int32))
    rng = np.random.RandomState(0)
    idx = rng.randint(0, 3, n).astype(np.int32)
    env.reset_injected(idx); ora.reset_injected(idx)
    for t in range(50):
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        z = rng.standard_normal((32, n)).astype(np.float32)
        idx = rng.randint(0, 3, n).astype(np.int32)
        _cmp(env.step_injected(a, z, idx), ora.step_injected(a, z, idx, 2))
    assert np.array_equal(_np(env.get_state()[0]), ora.x)
    env.close()


def test_misuse_messages():
    g, task = load_linds_golden(FILES[0])
    env = LinDSVecEnv(4)
    with pytest.raises(Exception, match="Must call \"set_task\" first"):
        env.reset()
    env.set_task(task)
    with pytest.raises(Exception, match="before doing any actions"):
        env.step(np.zeros((4, 8), np.float32))
    env.reset()
    with pytest.raises(AssertionError, match="Action shape mismatch"):
        env.step(np.zeros((4, 5), np.float32))
    env.close()


def test_get_future_inner_cmds():
    """linds_env.py:171-183: the queue of tracked commands, cmd(steps - delay + k) * target_valid"""
    g, task = load_linds_golden(FILES[0])          # dynamic target with a delay
    env = LinDSVecEnv(3, autoreset_mode="disabled")
    env.set_task(task)
    env.reset_injected([int(g["init_idx"])] * 3)
    for t in range(5):
        env.step_injected(np.tile(g["tr_action"][t], (3, 1)), np.tile(g["tr_z"][t][:, None], (1, 3)), [0, 0, 0])
    K = 7
    fut = env.get_future_inner_cmds(K)
    d, valid = int(task["target_delay"]), np.asarray(task["target_valid"], np.float64)
    ref = np.stack([task["command"](5 - d + k) * valid for k in range(K)])
    assert np.allclose(fut[0, :, :8], ref, rtol=1e-5, atol=1e-5) and np.allclose(fut[0], fut[2])
    env.close()


def test_state_accessors_under_the_slot_layout():
    """an interleaved env -> task map makes the engine keep its state in slot order (tiles of 32 per task); get_state /
    set_state / `state` speak the caller's env order, and a step after set_state equals the oracle's from that state"""
    tasks, tab, env_task = _batch(40, FILES[:4])
    rng = np.random.RandomState(12)
    rng.shuffle(env_task)
    n = len(env_task)
    env = LinDSVecEnv(n, autoreset_mode="disabled")
    env.set_task(tasks, env_task_index=env_task)
    ora = oracle.LinDSOracle(tab, env_task)
    idx = np.zeros(n, np.int32)
    env.reset_injected(idx); ora.reset_injected(idx)
    x0 = rng.uniform(-0.2, 0.2, (tab["NS"], n)).astype(np.float32)
    st0 = rng.randint(0, 20, n).astype(np.int32)
    env.set_state(x=x0, steps=st0, need_reset=np.zeros(n, np.uint8))
    ora.x[:] = x0; ora.steps[:] = st0; ora.need_reset[:] = 0
    x, st, nr = env.get_state()
    assert np.array_equal(_np(x), x0) and np.array_equal(_np(st), st0) and not _np(nr).any()
    assert np.array_equal(_np(env.state), x0.T)
    for path in ("mfma", "scalar"):
        env.set_path(path)
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        z = rng.standard_normal((tab["NS"], n)).astype(np.float32)
        _cmp(env.step_injected(a, z, idx), ora.step_injected(a, z, idx, 0))
        x, st, nr = env.get_state()
        assert np.array_equal(_np(x), ora.x) and np.array_equal(_np(st), ora.steps)
    env.close()


@pytest.mark.parametrize("layout", ["grouped64", "mixed_dims", "scattered"])
def test_fused_rollout_equals_single_steps(layout):
    """xv_linds_rollout: T steps in one launch (state resident in registers) = T calls of xv_linds_step, bit for bit,
    through episode ends (short max_steps -> SAME_STEP resets inside the launch) and under the slot layout"""
    if layout == "mixed_dims":
        files = FILES[:2] + [f for f in FILES if "32x8x8" in f][:1]
    else:
        files = FILES[:4]
    tasks = []
    for f in files:
        t = load_linds_golden(f)[1]
        t["max_steps"] = 23
        tasks.append(t)
    if layout == "scattered":
        env_task = np.random.RandomState(2).randint(0, len(tasks), 1000).astype(np.int32)
    else:
        env_task = np.repeat(np.arange(len(tasks), dtype=np.int32), 64)
    n, T = len(env_task), 70
    acts = np.random.RandomState(3).uniform(-1.3, 1.3, (T, n, 8)).astype(np.float32)
    recs = []
    for fused in (False, True):
        env = LinDSVecEnv(n, autoreset_mode="same_step", seed=77, env_id_base=123)
        env.set_task(tasks, env_task_index=env_task)
        env.reset()
        if fused:
            a = env.rollout(acts[:30])
            b = env.rollout(acts[30:])          # a second launch continues where the first stopped
            rec = {k: np.concatenate([_np(a[k]), _np(b[k])]) for k in a}
        else:
            rows = []
            for t in range(T):
                o, r, te, tr, info = env.step(acts[t])
                rows.append(dict(obs=_np(o), reward=_np(r), terminated=_np(te).astype(np.uint8),
                                 truncated=_np(tr).astype(np.uint8), command=_np(info["command"]),
                                 error=_np(info["error"]), final_obs=_np(info["final_obs"])))
            rec = {k: np.stack([row[k] for row in rows]) for k in rows[0]}
        x, st, nr = env.get_state()
        rec.update(x=_np(x), steps=_np(st), tick=np.int64(env.engine.tick))
        recs.append(rec)
        env.close()
    assert recs[0]["truncated"].sum() > 0
    for k in recs[0]:
        a, b = recs[0][k], recs[1][k]
        if k in ("obs", "command", "final_obs"):
            b = b[..., :a.shape[-1]]
        assert np.array_equal(a, b), k


def test_fused_rollout_vs_oracle():
    tasks, tab, env_task = _batch(64, FILES[:3])
    n, T = len(env_task), 25
    seed, base = 4242, 900
    env = LinDSVecEnv(n, autoreset_mode="same_step", seed=seed, env_id_base=base)
    env.set_task(tasks, env_task_index=env_task)
    ora = oracle.LinDSOracle(tab, env_task)
    tick = env.engine.tick
    env.reset(); ora.reset(seed, base, tick)
    x, st, _ = env.get_state()
    ora.x[:] = _np(x)
    acts = np.random.RandomState(8).uniform(-1, 1, (T, n, 8)).astype(np.float32)
    tick = env.engine.tick
    out = env.rollout(acts)
    for t in range(T):
        o = ora.step(seed, base, tick + t, acts[t], 2)
        assert close_rel(_np(out["obs"][t]), o["obs"], 2e-4, 2e-5), t     # noise tolerance compounds over T (no re-sync)
        assert np.array_equal(_np(out["truncated"][t]), o["truncated"])
    env.close()


def test_step_many_equals_single_steps():
    """xv_linds_step_many: K steps issued from C over ring buffers == K calls of step() (outputs of the last ring cycle,
    state, counters, engine tick); final_obs rows of finished envs only"""
    tasks, tab, env_task = _batch(64, FILES[:3])
    n, P, K = len(env_task), 8, 21
    acts = np.random.RandomState(11).uniform(-1.2, 1.2, (P, n, 8)).astype(np.float32)
    recs = []
    for many in (False, True):
        env = LinDSVecEnv(n, autoreset_mode="same_step", seed=5, env_id_base=64)
        env.set_task(tasks, env_task_index=env_task)
        env.reset()
        extra = np.random.RandomState(12).uniform(-1, 1, (n, 8)).astype(np.float32)
        if many:      # xv_linds_step_many is a plain launch loop in C over the ring slots (no graph: HISTORY.md 5.1)
            ring = env.step_many(K, acts)                 # 2 ring cycles + 5 steps
            env.step(extra)                               # an ordinary step in between moves the engine tick
            ring = env.step_many(2 * P, acts, out=ring)   # the same ring buffers again
            rec = {k: _np(v) for k, v in ring.items()}
        else:
            rows = [None] * P
            seq = [acts[k % P] for k in range(K)] + [extra] + [acts[k % P] for k in range(2 * P)]
            for q, a in enumerate(seq):
                o, r, te, tr, info = env.step(a)
                if q == K:
                    continue
                k = q if q < K else q - K - 1
                rows[k % P] = dict(obs=_np(o), reward=_np(r), terminated=_np(te).astype(np.uint8),
                                   truncated=_np(tr).astype(np.uint8), command=_np(info["command"]), error=_np(info["error"]),
                                   final_obs=_np(info["final_obs"]))
            rec = {k: np.stack([row[k] for row in rows]) for k in rows[0]}
        x, st, nr = env.get_state()
        rec.update(x=_np(x), steps=_np(st), tick=np.int64(env.engine.tick))
        recs.append(rec)
        env.close()
    done = (recs[0]["terminated"] | recs[0]["truncated"]).astype(bool)
    assert done.sum() > 0
    for k in recs[0]:
        a, b = recs[0][k], recs[1][k]
        if k in ("obs", "command", "final_obs"):
            b = b[..., :a.shape[-1]]
        if k == "final_obs":            # the ring's rows are written by finished envs only (earlier cycles may linger)
            a, b = a[done], b[done]
        assert np.array_equal(a, b), k


def test_process_noise_generator_is_standard_normal_on_the_device():
    """the device's 16 + 16-bit Box-Muller (csrc/philox.h: xv_box_muller16), sampled through the C-ABI and independent of
    the oracle's restatement: from x = 0 with a zero action and X = 0 one step leaves x' = noise_drift * dt * z, so the state
    IS the noise.  16.8 million draws: mean, variance, kurtosis and tail mass against N(0, 1) and against the exact moments of
    the generator's 65,536 x 65,536 grid (tests/test_host_normals.py): variance 0.99990, |z| <= 4.71."""
    from math import erfc, sqrt
    from xenoverse_amd.linds import LinearDSSampler
    n, nd = 65536, 0.02
    tasks = []
    for k in range(4):
        t = LinearDSSampler(16, 8, 8, seed=k)
        t["ld_X"] = np.zeros_like(np.asarray(t["ld_X"]))
        t["initial_states"] = [np.zeros(16)]
        t["noise_drift"] = nd
        t["max_steps"] = 10000
        tasks.append(t)
    env = LinDSVecEnv(n, autoreset_mode="disabled", seed=77)
    env.set_task(tasks)
    env.reset()
    a = torch.zeros((n, 8), device=env.device)
    zero = torch.zeros((16, n), device=env.device)
    scale = np.float32(nd) * np.float32(0.1)
    acc = []
    for it in range(16):
        env.set_state(x=zero, steps=torch.zeros(n, dtype=torch.int32, device=env.device))
        env.step(a)
        x, _, _ = env.get_state()
        acc.append((x.double() / float(scale)).cpu().numpy().ravel())
    z = np.concatenate(acc)
    N = z.size
    assert N == 16 * 16 * n
    assert abs(z.mean()) < 5.0 / np.sqrt(N)
    var = z.var()
    assert abs(var - 0.99990) < 5.0 * np.sqrt(2.0 / N) + 2e-5, var
    kurt = np.mean((z - z.mean()) ** 4) / var ** 2
    assert abs(kurt - 2.9987) < 5.0 * np.sqrt(24.0 / N) + 1e-3, kurt
    assert np.abs(z).max() <= 4.7097 * (1 + 1e-5)
    for t_, tol in ((1.0, 0.005), (2.0, 0.01), (3.0, 0.03)):
        p = float(np.mean(np.abs(z) > t_))
        assert abs(p - erfc(t_ / sqrt(2.0))) <= tol * erfc(t_ / sqrt(2.0)), (t_, p)
    # the draws of different envs, components and steps are not the same numbers
    assert len(np.unique(z[:1 << 20])) > 0.98 * (1 << 20)
    env.close()


@pytest.mark.parametrize("copy", [True, False])
def test_step_writes_steps_and_done_mask_from_the_same_launch(copy):
    """xv_linds_step_info: info["steps"] and the `_final_obs` mask come out of the step kernel — equal to the env's counters
    (xv_linds_get_state) and to terminated | truncated; the scalar test kernel (two more launches) gives the same infos"""
    tasks, tab, env_task = _batch(64, FILES[:3])
    n = len(env_task)
    acts = np.random.RandomState(3).uniform(-1.3, 1.3, (80, n, 8)).astype(np.float32)
    recs = []
    for path in ("mfma", "scalar"):
        env = LinDSVecEnv(n, autoreset_mode="same_step", seed=5, copy=copy)
        env.set_task(tasks, env_task_index=env_task)
        env.set_path(path)
        env.reset()
        rec, ended = [], 0
        for t in range(80):
            o, r, te, tr, info = env.step(acts[t])
            _, st, _ = env.get_state()
            assert np.array_equal(_np(info["steps"]), _np(st))
            assert np.array_equal(_np(info["_final_obs"]), _np(te) | _np(tr))
            rec.append((_np(info["steps"]).copy(), _np(info["_final_obs"]).copy(), _np(o).copy()))
            ended += int((_np(te) | _np(tr)).sum())
        assert ended > 5 and env.check_errors() == 0
        recs.append(rec)
        env.close()
    for a, b in zip(*recs):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
