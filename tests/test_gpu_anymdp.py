"""GPU parity: the HIP AnyMDP path (through the C-ABI) vs the reference's golden outputs and vs the CPU
oracle on the same seeded inputs.  Integer paths (state, obs, flags, step counters) bit-exact; rewards are
the same fp32 fmaf on both sides (bit-exact vs the oracle) and within 1e-5 rel of the fp64 reference."""
import numpy as np
import pytest
import torch

import oracle
from xenoverse_amd.anymdp import AnyMDPVecEnv, build_tables, row_lines, to_blocked
from util import close_f32, golden_files, load_anymdp_golden

pytestmark = pytest.mark.gpu
FILES = golden_files("anymdp_")
MODES = {"disabled": 0, "next_step": 1, "same_step": 2}


def _np(t):
    return t.detach().cpu().numpy()


def _dev_tables(tab, dev="cuda:0"):
    out = dict(S=tab["S"], A=tab["A"], s0_max=tab["s0_max"])
    tab = dict(tab, rows=to_blocked(tab["cdf"], tab["rs"]))
    for k in ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps"):
        v = np.ascontiguousarray(tab[k])
        if v.dtype == np.uint64:
            v = v.view(np.int64)
        out[k] = torch.from_numpy(v).to(dev)
    return out


# ---------------------------------------------------------------------------------------------------
# golden vectors produced by the reference itself
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", FILES)
def test_golden_single_step_tuples(path):
    g, task = load_anymdp_golden(path)
    n = len(g["ss_s"])
    env = AnyMDPVecEnv(n, autoreset_mode="disabled")
    env.set_task(task)
    env.set_state(inner_state=g["ss_s"], steps=np.zeros(n), need_reset=np.zeros(n))
    obs, r, term, trunc, info = env.step_injected(g["ss_a"], g["ss_u"], g["ss_z"], np.zeros(n))
    assert np.array_equal(_np(env.inner_state), g["ss_next"])
    assert np.array_equal(_np(obs), g["state_mapping"][g["ss_next"]])
    assert np.array_equal(_np(term).astype(np.uint8), g["ss_term"])
    assert np.array_equal(_np(info["reward_gt"]), g["ss_rgt"].astype(np.float32))
    assert close_f32(_np(r), g["ss_r"])
    assert env.check_errors() == 0
    env.close()


@pytest.mark.parametrize("path", FILES)
def test_golden_reset_draws(path):
    g, task = load_anymdp_golden(path)
    n = len(g["reset_u"])
    env = AnyMDPVecEnv(n, autoreset_mode="disabled")
    env.set_task(task)
    obs = env.reset_injected(g["reset_u"])
    assert np.array_equal(_np(obs), g["reset_obs"])
    assert np.array_equal(_np(env.inner_state), g["reset_state"])
    env.close()


@pytest.mark.parametrize("path", FILES[:2])
def test_golden_trajectory_with_truncation_and_transition_gt(path):
    """one env driven exactly like the reference's rollout loop (manual reset on done)"""
    g, task = load_anymdp_golden(path)
    env = AnyMDPVecEnv(1, autoreset_mode="disabled", with_transition_gt=True)
    env.set_task(task)
    o0 = env.reset_injected([float(g["init_u"])])
    assert int(o0[0]) == int(g["init_obs"])
    T = 600 if len(g["tr_a"]) > 600 else len(g["tr_a"])
    T = max(T, int(np.argmax(g["tr_trunc"])) + 2)
    for t in range(T):
        if g["tr_set_steps"][t] >= 0:
            env.set_state(steps=[int(g["tr_set_steps"][t])])
        obs, r, term, trunc, info = env.step_injected([g["tr_a"][t]], [g["tr_u"][t]], [g["tr_z"][t]], [0.0])
        assert int(obs[0]) == g["tr_obs"][t]
        assert bool(term[0]) == bool(g["tr_term"][t]) and bool(trunc[0]) == bool(g["tr_trunc"][t])
        assert int(info["steps"][0]) == g["tr_steps"][t]
        assert close_f32(_np(r), g["tr_r"][t:t + 1])
        if t < len(g["tr_tgt"]):
            assert np.max(np.abs(_np(info["transition_gt"])[0] - g["tr_tgt"][t])) < 1e-12
        if term[0] or trunc[0]:
            ro = env.reset_injected([g["tr_ur"][t]])
            assert int(ro[0]) == g["tr_reset_obs"][t]
    assert g["tr_trunc"][:T].sum() >= 1
    env.close()


# ---------------------------------------------------------------------------------------------------
# seeded batches vs the oracle — BASELINE config 1 (S=16, A=4, 128 envs, 4 reference-sampled tasks)
# ---------------------------------------------------------------------------------------------------
def _config1():
    tasks = [load_anymdp_golden(p)[1] for p in golden_files("anymdp_16x4")[:4]]
    tab = build_tables(tasks)
    env_task = np.repeat(np.arange(4, dtype=np.int32), 32)
    return tasks, tab, env_task


def _compare_step(dev_out, ora_out, exact_reward=True):
    obs, r, term, trunc, info = dev_out
    o_obs, o_r, o_rgt, o_term, o_trunc, o_fobs = ora_out
    assert np.array_equal(_np(obs), o_obs)
    assert np.array_equal(_np(term).astype(np.uint8), o_term)
    assert np.array_equal(_np(trunc).astype(np.uint8), o_trunc)
    assert np.array_equal(_np(info["reward_gt"]), o_rgt)
    if exact_reward:
        assert np.array_equal(_np(r), o_r)
    else:
        assert close_f32(_np(r), o_r, rel=1e-5, abs_=2e-6)
    if "final_obs" in info:
        assert np.array_equal(_np(info["final_obs"]), o_fobs)


@pytest.mark.parametrize("mode", ["disabled", "next_step", "same_step"])
def test_config1_injected_vs_oracle(mode):
    tasks, tab, env_task = _config1()
    n = len(env_task)
    env = AnyMDPVecEnv(n, autoreset_mode=mode)
    env.set_task(tasks, env_task_index=env_task)
    ora = oracle.AnyMDPOracle(tab, env_task)
    rng = np.random.RandomState(7)
    u0 = rng.random_sample(n)
    assert np.array_equal(_np(env.reset_injected(u0)), ora.reset_injected(u0))
    ended = 0
    for t in range(300):
        a = rng.randint(0, tab["A"], n).astype(np.int32)
        u, z, ur = rng.random_sample(n), rng.standard_normal(n).astype(np.float32), rng.random_sample(n)
        d = env.step_injected(a, u, z, ur)
        o = ora.step_injected(a, u, z, ur, MODES[mode])
        _compare_step(d, o)
        s, st, nr = env.get_state()
        assert np.array_equal(_np(s), ora.state) and np.array_equal(_np(st), ora.steps)
        assert np.array_equal(_np(nr), ora.need_reset)
        done = (o[3] | o[4]).astype(bool)
        ended += int(done.sum())
        if mode == "disabled" and done.any():
            # truncated-only envs keep stepping in the reference; terminated ones must be reset by the caller
            m = o[3].astype(np.uint8)
            if m.any():
                ur2 = rng.random_sample(n)
                assert np.array_equal(_np(env.reset_injected(ur2, mask=m))[m.astype(bool)],
                                      ora.reset_injected(ur2, mask=m)[m.astype(bool)])
    assert ended > 100
    assert env.check_errors() == ora.err_flags == 0
    env.close()


@pytest.mark.parametrize("mode", ["next_step", "same_step"])
def test_config1_free_running_philox_vs_oracle(mode):
    tasks, tab, env_task = _config1()
    n = len(env_task)
    seed, base = 0xC0FFEE1234, 1 << 33
    env = AnyMDPVecEnv(n, autoreset_mode=mode, seed=seed, env_id_base=base)
    env.set_task(tasks, env_task_index=env_task)
    ora = oracle.AnyMDPOracle(tab, env_task)
    env.engine.tick = (1 << 32) - 3          # crosses the 32-bit tick boundary
    tick = env.engine.tick
    obs, _ = env.reset()
    assert np.array_equal(_np(obs), ora.reset(seed, base, tick))
    rng = np.random.RandomState(11)
    for t in range(200):
        a = rng.randint(0, tab["A"], n).astype(np.int32)
        tick = env.engine.tick
        d = env.step(a)
        o = ora.step(seed, base, tick, a, MODES[mode])
        _compare_step(d, o, exact_reward=False)
    env.close()


# ---------------------------------------------------------------------------------------------------
# S = 64 wave-cooperative kernel (headline shape), synthetic tasks, ragged env count
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("search", ["fence", "binary", "bucket"])
@pytest.mark.parametrize("n_env,n_task", [(2048, 32), (1000, 8), (64, 64), (37, 3)])
def test_s64_wave_kernel_vs_oracle(n_env, n_task, search):
    tab = oracle.anymdp_synth(seed=99, task_index_base=5, n_task=n_task, S=64, A=8, s0_max=4)
    env_task = (np.arange(n_env) * n_task // n_env).astype(np.int32)
    rng = np.random.RandomState(n_env)
    rng.shuffle(env_task)                     # lanes of one wave mix tasks
    seed = 4242
    env = AnyMDPVecEnv(n_env, autoreset_mode="same_step", seed=seed)
    env.set_task(_dev_tables(tab), env_task_index=env_task)
    env.set_search(search)
    ora = oracle.AnyMDPOracle(tab, env_task)
    tick = env.engine.tick
    obs, _ = env.reset()
    assert np.array_equal(_np(obs), ora.reset(seed, 0, tick))
    for t in range(60):
        a = rng.randint(0, 8, n_env).astype(np.int32)
        if t % 2 == 0:
            tick = env.engine.tick
            d = env.step(a)
            o = ora.step(seed, 0, tick, a, 2)
            _compare_step(d, o, exact_reward=False)
        else:
            u, z, ur = rng.random_sample(n_env), rng.standard_normal(n_env).astype(np.float32), rng.random_sample(n_env)
            # exercise exact CDF boundaries: u equal to a stored CDF entry must select the NEXT state
            k = rng.randint(0, n_env, 8)
            rows = tab["cdf"][env_task[k], ora.state[k], a[k]]
            u[k] = np.minimum(rows[np.arange(8), rng.randint(0, 64, 8)], np.nextafter(1.0, 0.0))
            d = env.step_injected(a, u, z, ur)
            o = ora.step_injected(a, u, z, ur, 2)
            _compare_step(d, o)
        s, st, _ = env.get_state()
        assert np.array_equal(_np(s), ora.state) and np.array_equal(_np(st), ora.steps)
    assert env.check_errors() == 0
    env.close()


@pytest.mark.parametrize("S,A,n_task,obs_offset", [(16, 4, 6, 0), (100, 5, 3, 0), (256, 3, 2, 0), (8, 2, 5, 0),
                                                    (7, 2, 3, 0), (112, 3, 2, 0), (113, 3, 2, 0), (128, 5, 2, 0),
                                                    (224, 2, 2, 0), (225, 2, 2, 0),
                                                    (300, 4, 2, 0), (336, 2, 2, 0), (337, 3, 2, 0), (449, 2, 2, 0), (512, 2, 2, 100),
                                                    (64, 8, 4, 65000), (64, 8, 4, 70000)])
def test_generic_kernel_other_sizes_vs_oracle(S, A, n_task, obs_offset):
    """block-count boundaries of the fence path (S = 7: one block; S = 112: 16 blocks, one per fence entry; S = 113
    .. 224: two blocks per fence entry; S = 225 .. 336: three; .. 448: four; .. 512: five — the reference's samplers take
    any state_space, task_sampler.py:15-19,90-100) and observation ids at the 16-bit metadata limit (ids < 65536 stay on
    the fence path, larger ones fall back to the per-lane binary search); every case is also run with search = bucket
    (where the fence layout exists) and search = binary"""
    tab = oracle.anymdp_synth(seed=5, task_index_base=0, n_task=n_task, S=S, A=A, s0_max=3)
    tab["state_map"] = tab["state_map"] + np.int32(obs_offset)
    n_env = 50 * n_task
    env_task = np.repeat(np.arange(n_task, dtype=np.int32), 50)
    env = AnyMDPVecEnv(n_env, autoreset_mode="same_step")
    env.set_task(_dev_tables(tab), env_task_index=env_task)
    expect_fast = obs_offset + S <= 65536
    if expect_fast:
        env.set_search("fence")
    else:
        with pytest.raises(Exception, match="FENCE needs"):
            env.set_search("fence")
    ora = oracle.AnyMDPOracle(tab, env_task)
    rng = np.random.RandomState(S)
    u0 = rng.random_sample(n_env)
    assert np.array_equal(_np(env.reset_injected(u0)), ora.reset_injected(u0))
    for t in range(80):
        if t == 25 and expect_fast:
            env.set_search("bucket", n_bucket=16)   # ... on the bucket lines (9-bit first-index field for S > 256)
        if t == 50:
            env.set_search("binary")       # the same states continue on the per-lane path
        a = rng.randint(0, A, n_env).astype(np.int32)
        u, z, ur = rng.random_sample(n_env), rng.standard_normal(n_env).astype(np.float32), rng.random_sample(n_env)
        if t % 3 == 0:                     # exact CDF entries, incl. the fence values at block-group edges
            k = rng.randint(0, n_env, 8)
            rows = tab["cdf"][env_task[k], ora.state[k], a[k]]
            u[k] = np.minimum(rows[np.arange(8), rng.randint(0, S, 8)], np.nextafter(1.0, 0.0))
        _compare_step(env.step_injected(a, u, z, ur), ora.step_injected(a, u, z, ur, 2))
    env.close()


@pytest.mark.parametrize("S,A,n_task,s0_max", [(64, 8, 16, 4), (16, 4, 9, 3), (100, 5, 4, 2), (256, 3, 2, 8)])
def test_device_synth_generator_bit_exact(S, A, n_task, s0_max):
    import ctypes as C
    from xenoverse_amd import Engine, _lib
    eng = Engine("cuda:0", seed=1)
    ref = oracle.anymdp_synth(seed=31337, task_index_base=1000, n_task=n_task, S=S, A=A, s0_max=s0_max)
    d = "cuda:0"
    words = (S + 63) // 64
    t = dict(rows=torch.zeros((n_task, S, A, row_lines(S), 16), dtype=torch.float64, device=d),
             state_map=torch.empty((n_task, S), dtype=torch.int32, device=d),
             term_mask=torch.empty((n_task, words), dtype=torch.int64, device=d),
             s0_cdf=torch.empty((n_task, s0_max), dtype=torch.float64, device=d),
             s0_ids=torch.empty((n_task, s0_max), dtype=torch.int32, device=d),
             max_steps=torch.empty(n_task, dtype=torch.int32, device=d))
    _lib.check(eng.lib.xv_anymdp_synth_tasks(eng.handle, 31337, 1000, n_task, S, A, s0_max, *[_lib.ptr(t[k]) for k in
               ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")]))
    eng.sync()
    ref = dict(ref, rows=to_blocked(ref["cdf"], ref["rs"]))
    for k in t:
        got = _np(t[k])
        exp = ref[k].view(np.int64) if ref[k].dtype == np.uint64 else ref[k]
        if k == "rows":   # bit-exact entries (incl. padding), float pairs viewed as 8-byte words; fence line and
            # block metadata are completed by xv_anymdp_create, not by the generator
            assert np.array_equal(got[..., 1:, :14].view(np.int64), exp[..., 1:, :14].view(np.int64)), k
        else:
            assert np.array_equal(got, exp), k
    from xenoverse_amd.anymdp import from_blocked
    c, _ = from_blocked(_np(t["rows"]), S)
    # structural properties of every row: monotone CDF ending in exactly 1.0
    assert np.all(np.diff(c, axis=-1) >= 0) and np.all(c[..., -1] == 1.0)
    eng.close()


@pytest.mark.parametrize("search", ["fence", "binary", "bucket"])
def test_golden_64x8_tuples_all_search_modes(search):
    g, task = load_anymdp_golden([f for f in FILES if "64x8" in f][0])
    n = len(g["ss_s"])
    env = AnyMDPVecEnv(n, autoreset_mode="disabled")
    env.set_task(task)
    env.set_search(search)
    env.set_state(inner_state=g["ss_s"], steps=np.zeros(n), need_reset=np.zeros(n))
    obs, r, term, trunc, info = env.step_injected(g["ss_a"], g["ss_u"], g["ss_z"], np.zeros(n))
    assert np.array_equal(_np(env.inner_state), g["ss_next"])
    assert np.array_equal(_np(term).astype(np.uint8), g["ss_term"])
    assert close_f32(_np(r), g["ss_r"])
    env.close()


@pytest.mark.parametrize("search", ["fence", "binary", "bucket"])
def test_fused_rollout_equals_stepwise(search):
    tab = oracle.anymdp_synth(seed=3, task_index_base=0, n_task=16, S=64, A=8, s0_max=4)
    n_env, T = 1024, 48
    env_task = np.repeat(np.arange(16, dtype=np.int32), 64)
    acts = np.random.RandomState(0).randint(0, 8, (T, n_env)).astype(np.int32)
    outs = []
    for fused in (False, True):
        env = AnyMDPVecEnv(n_env, autoreset_mode="same_step", seed=77)
        env.set_task(_dev_tables(tab), env_task_index=env_task)
        env.set_search(search)
        env.reset()
        if fused:
            o = env.rollout(acts)
            outs.append({k: _np(v) for k, v in o.items()})
        else:
            rec = {k: [] for k in ("obs", "reward", "reward_gt", "terminated", "truncated", "final_obs")}
            for t in range(T):
                obs, r, term, trunc, info = env.step(acts[t])
                rec["obs"].append(_np(obs)); rec["reward"].append(_np(r)); rec["reward_gt"].append(_np(info["reward_gt"]))
                rec["terminated"].append(_np(term).astype(np.uint8)); rec["truncated"].append(_np(trunc).astype(np.uint8))
                rec["final_obs"].append(_np(info["final_obs"]))
            outs.append({k: np.stack(v) for k, v in rec.items()})
        s, st, _ = env.get_state()
        outs[-1]["state"], outs[-1]["steps"] = _np(s), _np(st)
        env.close()
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert outs[0]["terminated"].sum() > 1000


def test_sharding_invariance_env_id_base():
    """an env's trajectory depends on (seed, GLOBAL env id) only: a shard reproduces its slice of the batch"""
    tab = oracle.anymdp_synth(seed=8, task_index_base=0, n_task=8, S=64, A=8, s0_max=4)
    n_env, T = 512, 20
    env_task = np.repeat(np.arange(8, dtype=np.int32), 64)
    acts = np.random.RandomState(1).randint(0, 8, (T, n_env)).astype(np.int32)

    def run(lo, hi):
        env = AnyMDPVecEnv(hi - lo, autoreset_mode="same_step", seed=2024, env_id_base=lo)
        env.set_task(_dev_tables(tab), env_task_index=env_task[lo:hi])
        env.reset()
        o = env.rollout(acts[:, lo:hi])
        res = {k: _np(v) for k, v in o.items()}
        env.close()
        return res

    full = run(0, n_env)
    for lo, hi in ((0, 256), (256, 512)):
        part = run(lo, hi)
        for k in full:
            assert np.array_equal(full[k][:, lo:hi], part[k]), k


def test_error_flags_and_misuse_messages():
    _, task = load_anymdp_golden(FILES[0])
    env = AnyMDPVecEnv(2, autoreset_mode="disabled")
    with pytest.raises(Exception, match="Must call \"set_task\" first"):
        env.reset()
    env.set_task(task)
    with pytest.raises(Exception, match="before doing any actions"):
        env.step([0, 0])
    env.reset_injected([0.1, 0.1])
    env.step_injected([0, task["na"]], [0.5, 0.5], [0.0, 0.0], [0.5, 0.5])
    assert env.check_errors() & 1                    # action out of range
    s_term = int(task["s_e"][0])
    env.set_state(inner_state=[s_term, s_term])
    obs, r, term, trunc, _ = env.step_injected([0, 0], [0.5, 0.5], [0.0, 0.0], [0.5, 0.5])
    assert env.check_errors() & 2 and bool(term[0])   # stepping a terminated env
    assert np.all(_np(env.inner_state) == s_term)
    env.close()


def test_device_philox_kat():
    from xenoverse_amd import Engine
    eng = Engine("cuda:0")
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kats:
        c = torch.from_numpy(np.array(ctr, np.uint32).view(np.int32)).reshape(1, 4).cuda()
        k = torch.from_numpy(np.array(key, np.uint32).view(np.int32)).cuda()
        out = _np(eng.philox(c, k)).view(np.uint32)[0]
        assert tuple(int(x) for x in out) == exp
    rng = np.random.RandomState(0)
    ctrs = rng.randint(0, 2**32, (1000, 4), dtype=np.uint64).astype(np.uint32)
    key = np.array([123456789, 987654321], np.uint32)
    out = _np(eng.philox(torch.from_numpy(ctrs.view(np.int32)).cuda(), torch.from_numpy(key.view(np.int32)).cuda()))
    assert np.array_equal(out.view(np.uint32), oracle.philox4x32_10(ctrs, key))
    eng.close()


# ---------------------------------------------------------------------------------------------------
# BASELINE full size (config 2a): 65,536 envs, one synthetic task per env = 44 GiB of rows in HBM.
# Too large for the oracle as a whole -> size-independent properties over all envs + an oracle check of every 8th env
# (8,192 envs x 32 steps; the synthetic generator is a pure function of (seed, task index), so the CPU rebuilds just
# those tasks — 4 GiB — and the oracle draws with each env's global id).
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("search", ["fence", "bucket"])
def test_full_size_config_2a_properties_and_subset_vs_oracle(search):
    import ctypes as C
    from xenoverse_amd import _lib
    free, total = torch.cuda.mem_get_info()
    if free < (52 if search == "fence" else 120) * 2**30:
        pytest.skip("needs ~46 GiB of free HBM (+64 GiB of bucket lines)")
    n_env, S, A, T = 65536, 64, 8, 32
    seed_tab, seed = 1235, 1234
    env = AnyMDPVecEnv(n_env, seed=seed, autoreset_mode="same_step")
    d = env.device
    t = dict(S=S, A=A, s0_max=4,
             rows=torch.empty((n_env, S, A, row_lines(S), 16), dtype=torch.float64, device=d),
             state_map=torch.empty((n_env, S), dtype=torch.int32, device=d),
             term_mask=torch.empty((n_env, 1), dtype=torch.int64, device=d),
             s0_cdf=torch.empty((n_env, 4), dtype=torch.float64, device=d),
             s0_ids=torch.empty((n_env, 4), dtype=torch.int32, device=d),
             max_steps=torch.empty(n_env, dtype=torch.int32, device=d))
    _lib.check(env.lib.xv_anymdp_synth_tasks(env.engine.handle, seed_tab, 0, n_env, S, A, 4, *[_lib.ptr(t[k]) for k in
               ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")]))
    env.set_task(t, env_task_index=torch.arange(n_env, dtype=torch.int32, device=d))
    env.set_search(search, n_bucket=16) if search == "bucket" else env.set_search(search)
    # oracle for a scattered subset: every 8th env (8,192 envs: some lanes of every wave of the launch), tables rebuilt on
    # the CPU per task index (4 GiB), draws keyed by each env's GLOBAL id
    stride, first = 8, 5
    sub = np.arange(first, n_env, stride)
    tab = oracle.anymdp_synth(seed=seed_tab, task_index_base=first, n_task=len(sub), S=S, A=A, s0_max=4, task_stride=stride)
    ora = oracle.AnyMDPOracle(tab, np.arange(len(sub), dtype=np.int32), gid_stride=stride)
    sub_t = torch.from_numpy(sub).to(d)
    tick = env.engine.tick
    obs, _ = env.reset()
    assert np.array_equal(ora.reset(seed, first, tick), _np(obs[sub_t]))
    g = torch.Generator(device=d); g.manual_seed(5)
    term_mask = t["term_mask"][:, 0]
    n_done = n_sub_done = 0
    for step in range(T):
        a = torch.randint(0, A, (n_env,), generator=g, device=d, dtype=torch.int32)
        tick = env.engine.tick
        s_before, st_before, _ = env.get_state()
        obs, r, term, trunc, info = env.step(a)
        s_after, st_after, _ = env.get_state()
        done = term | trunc
        # -- properties over all 65,536 envs --
        assert int(obs.min()) >= 0 and int(obs.max()) < S
        fo = info["final_obs"]
        inv = torch.empty_like(t["state_map"]); inv.scatter_(1, t["state_map"].long(), torch.arange(S, device=d, dtype=torch.int32).expand(n_env, S))
        s_next = torch.where(done, inv.gather(1, fo.clamp(min=0).long()[:, None])[:, 0], s_after)   # inner next state
        assert torch.equal(((term_mask >> s_next.long()) & 1).bool(), term)               # terminated <=> s' in s_e
        assert torch.equal(trunc, (st_before + 1) >= t["max_steps"])                       # truncation rule
        assert torch.equal(st_after, torch.where(done, torch.zeros_like(st_after), st_before + 1))
        assert bool(((s_after[done] >= 0) & (s_after[done] <= 2)).all())                   # resets land in s_0 = {0,1,2}
        assert torch.equal(obs, t["state_map"].gather(1, s_after.long()[:, None])[:, 0])   # obs = state_mapping[state]
        lo = torch.clamp(s_before - 33, min=0); hi = torch.clamp(s_before + 17, max=S)    # band of the synthetic rows
        assert bool(((s_next >= lo) & (s_next < hi)).all())
        assert bool(torch.isfinite(r).all())
        n_done += int(done.sum())
        # -- subset vs oracle, draw for draw --
        eo, er, ergt, eterm, etrunc, efo = ora.step(seed, first, tick, _np(a[sub_t]), 2)
        assert np.array_equal(eo, _np(obs[sub_t])) and np.array_equal(efo, _np(fo[sub_t]))
        assert np.array_equal(eterm, _np(term[sub_t]).astype(np.uint8)) and np.array_equal(etrunc, _np(trunc[sub_t]).astype(np.uint8))
        assert np.array_equal(ergt, _np(info["reward_gt"][sub_t])) and close_f32(_np(r[sub_t]), er)
        assert np.array_equal(ora.state, _np(s_after[sub_t])) and np.array_equal(ora.steps, _np(st_after[sub_t]))
        n_sub_done += int((eterm | etrunc).sum())
    assert n_done > 65536 and n_sub_done > 8192 and env.check_errors() == 0
    # -- the same envs stepped from C: overlapped launches (two streams, hand-off through the env records) against the
    #    one-stream path, every ring slot of 4 cycles + 9 steps, the env records and the tick, bit for bit; the scattered
    #    subset of the last ring against the oracle --
    P = 32
    acts = torch.randint(0, A, (P, n_env), generator=g, device=d, dtype=torch.int32)
    env.set_step_many_graph("on")
    snap = []
    for overlap in (False, True):
        env.reset(seed=77)
        env.set_step_many_overlap(overlap)
        tick0 = env.engine.tick
        ring = env.step_many(4 * P + 9, acts)
        torch.cuda.synchronize()
        assert env.step_many_overlap_state == (1 if overlap else 0)
        snap.append(([v.clone() for v in ring.values()], [x.clone() for x in env.get_state()], env.engine.tick))
    for x, y in zip(snap[0][0] + snap[0][1], snap[1][0] + snap[1][1]):
        assert torch.equal(x, y)
    assert snap[0][2] == snap[1][2] == tick0 + 4 * P + 9
    ora.reset(seed, first, (77 & 0xFFFFFFFF) << 24)
    for k in range(4 * P + 9):
        o = ora.step(seed, first, tick0 + k, _np(acts[k % P][sub_t]), 2)
    last = (4 * P + 8) % P
    assert np.array_equal(o[0], _np(snap[1][0][0][last][sub_t])) and np.array_equal(ora.state, _np(snap[1][1][0][sub_t]))
    assert env.check_errors() == 0
    env.close()


def test_gt_transition_and_reward_accessors():
    """get_gt_transition / get_gt_reward (anymdp_env.py:161-165): tensors re-indexed by observation id"""
    g, task = load_anymdp_golden(FILES[0])
    env = AnyMDPVecEnv(4)
    env.set_task(task)
    sm = g["state_mapping"]
    T_obs = np.zeros_like(g["transition"]); R_obs = np.zeros_like(g["reward"])
    for i, si in enumerate(sm):
        T_obs[si][:, sm] = g["transition"][i]
        R_obs[si][:, sm] = g["reward"][i]
    assert np.max(np.abs(env.get_gt_transition() - T_obs)) < 1e-12
    assert np.allclose(env.get_gt_reward(), R_obs, rtol=1e-6, atol=1e-6)
    env.close()


def test_teacher_rollout_on_device():
    """fused rollout driven by the optimal-policy table: actions are argmax_a Q[state]; equals stepping with them"""
    from xenoverse_amd.anymdp.teacher import optimal_policy_table
    tasks = [load_anymdp_golden(p)[1] for p in golden_files("anymdp_16x4")[:4]]
    greedy = optimal_policy_table(tasks)
    n, T = 128, 40
    env_task = np.repeat(np.arange(4, dtype=np.int32), 32)
    outs = []
    for fused in (True, False):
        env = AnyMDPVecEnv(n, seed=3, autoreset_mode="same_step")
        env.set_task(tasks, env_task_index=env_task)
        if fused:
            env.set_search("bucket")        # the teacher roll-out through the bucket lines, the steps through the fence
        env.reset()
        if fused:
            s0 = _np(env.inner_state).copy()
            o = env.rollout_teacher(T, greedy, epsilon=0.0)
            outs.append({k: _np(v) for k, v in o.items()})
            assert np.array_equal(outs[0]["action"][0], greedy[env_task, s0])
        else:
            rec = {k: [] for k in ("action", "obs", "reward", "terminated")}
            for t in range(T):
                a = greedy[env_task, _np(env.inner_state)].astype(np.int32)
                obs, r, term, trunc, info = env.step(a)
                rec["action"].append(a); rec["obs"].append(_np(obs)); rec["reward"].append(_np(r))
                rec["terminated"].append(_np(term).astype(np.uint8))
            outs.append({k: np.stack(v) for k, v in rec.items()})
        env.close()
    for k in ("action", "obs", "reward", "terminated"):
        assert np.array_equal(outs[0][k], outs[1][k]), k
    # ... and the oracle's restatement of the teacher rollout (greedy and epsilon-greedy), draw for draw
    from xenoverse_amd.anymdp import build_tables
    tab = build_tables(tasks)
    for eps in (0.0, 0.3):
        env = AnyMDPVecEnv(n, seed=11, env_id_base=4000, autoreset_mode="same_step")
        env.set_task(tasks, env_task_index=env_task)
        ora = oracle.AnyMDPOracle(tab, env_task)
        tick = env.engine.tick
        obs, _ = env.reset()
        assert np.array_equal(_np(obs), ora.reset(11, 4000, tick))
        tick = env.engine.tick
        dev = env.rollout_teacher(T, greedy, epsilon=eps)
        ref = ora.rollout_teacher(11, 4000, tick, T, greedy, epsilon=eps)
        for k in ("action", "obs", "terminated", "truncated", "final_obs"):
            assert np.array_equal(_np(dev[k]), ref[k]), (eps, k)
        assert np.allclose(_np(dev["reward"]), ref["reward"], rtol=1e-5, atol=2e-6)
        env.close()
    # the teacher is better than random: compare average reward_gt per step
    env = AnyMDPVecEnv(n, seed=3, autoreset_mode="same_step"); env.set_task(tasks, env_task_index=env_task); env.reset()
    opt = float(env.rollout_teacher(200, greedy, 0.0)["reward_gt"].mean())
    rnd = float(env.rollout_teacher(200, greedy, 1.0)["reward_gt"].mean())
    assert opt > rnd
    env.close()


# ---------------------------------------------------------------------------------------------------
# adversarial CDF rows: plateaus (zero-probability states) across block boundaries, all mass on the first / last
# state, u exactly on / one ulp around every stored entry.  Expected: numpy.searchsorted(cdf, u, 'right').
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("search", ["fence", "binary", "bucket"])
@pytest.mark.parametrize("S", [7, 8, 20, 64, 130, 250])
def test_adversarial_rows_match_searchsorted(S, search):
    rng = np.random.RandomState(S)
    A = 2
    pdfs = []
    e0 = np.zeros(S); e0[0] = 1.0; pdfs.append(e0)                       # everything on the first state
    eL = np.zeros(S); eL[S - 1] = 1.0; pdfs.append(eL)                   # ... on the last
    pdfs.append(np.full(S, 1.0 / S))
    for lo, hi in ((5, 9), (6, 7), (0, 6), (S - 3, S), (13, 15)):       # plateaus straddling 7-entry block edges
        p = rng.random_sample(S) + 0.01
        p[max(0, min(lo, S - 1)):min(hi, S)] = 0.0
        if p.sum() == 0:
            p[0] = 1.0
        pdfs.append(p / p.sum())
    p = np.zeros(S); p[::max(1, S // 3)] = 1.0; pdfs.append(p / p.sum())  # three spikes
    while len(pdfs) < S * A:
        p = rng.random_sample(S) * (rng.random_sample(S) < 0.3)
        if p.sum() == 0:
            p[rng.randint(S)] = 1.0
        pdfs.append(p / p.sum())
    T = np.array(pdfs[:S * A]).reshape(S, A, S)
    cdf = np.cumsum(T, -1)
    cdf = cdf / cdf[..., -1:]
    rs = rng.standard_normal((1, S, A, S, 2)).astype(np.float32)
    tab = dict(S=S, A=A, s0_max=1, cdf=cdf[None], rs=rs, state_map=rng.permutation(S).astype(np.int32)[None],
               term_mask=np.zeros((1, (S + 63) // 64), np.uint64), s0_cdf=np.ones((1, 1)),
               s0_ids=np.zeros((1, 1), np.int32),
               max_steps=np.array([10 ** 6], np.int32))
    # every (row, probe): u on each stored entry, one ulp below / above, 0, tiny, just below 1
    probes, rows_sa = [], []
    for s in range(S):
        for a in range(A):
            c = cdf[s, a]
            us = np.concatenate([c, np.nextafter(c, 0.0), np.nextafter(c, 2.0), [0.0, 5e-324, 1e-300, 0.5]])
            us = us[(us >= 0.0) & (us < 1.0)]
            probes.append(us); rows_sa.append(np.tile([[s, a]], (len(us), 1)))
    u = np.concatenate(probes)
    sa = np.concatenate(rows_sa)
    n = len(u)
    env = AnyMDPVecEnv(n, autoreset_mode="disabled")
    env.set_task(_dev_tables(tab), env_task_index=np.zeros(n, np.int32))
    env.set_search(search)
    env.reset_injected(np.zeros(n))
    env.set_state(inner_state=sa[:, 0].astype(np.int32))
    obs, r, term, trunc, info = env.step_injected(sa[:, 1].astype(np.int32), u, np.zeros(n, np.float32), np.zeros(n))
    exp = np.array([min(int(np.searchsorted(cdf[s, a], uu, side="right")), S - 1) for (s, a), uu in zip(sa, u)])
    got = _np(env.inner_state)
    bad = np.nonzero(got != exp)[0]
    assert len(bad) == 0, (len(bad), [(int(sa[b, 0]), int(sa[b, 1]), float(u[b]), int(got[b]), int(exp[b])) for b in bad[:12]])
    assert np.array_equal(_np(obs), tab["state_map"][0][exp])
    assert np.array_equal(_np(info["reward_gt"]), rs[0, sa[:, 0], sa[:, 1], exp, 0])
    assert env.check_errors() == 0
    env.close()


def test_copy_false_returns_views_of_alternating_output_sets():
    """copy=False: same values as copy=True; what a step returned stays intact through the next step and is
    overwritten by the one after (two output sets used alternately)"""
    tab = oracle.anymdp_synth(seed=3, task_index_base=0, n_task=4, S=64, A=8, s0_max=4)
    n = 256
    outs = {}
    for copy in (True, False):
        env = AnyMDPVecEnv(n, seed=9, autoreset_mode="same_step", copy=copy)
        env.set_task(_dev_tables(tab))
        env.reset()
        rng = np.random.RandomState(0)
        rec, kept = [], []
        for t in range(12):
            o = env.step(rng.randint(0, 8, n).astype(np.int32))
            rec.append([_np(o[0]), _np(o[1]), _np(o[2]), _np(o[3]), _np(o[4]["steps"]), _np(o[4]["reward_gt"]),
                        _np(o[4]["final_obs"]), _np(o[4]["_final_obs"])])
            kept.append(o)
            if t >= 1:      # the previous step's tensors are still what they were
                prev = kept[t - 1]
                assert np.array_equal(_np(prev[0]), rec[t - 1][0]) and np.array_equal(_np(prev[1]), rec[t - 1][1])
            if t >= 2 and not copy:
                assert kept[t - 2][0].data_ptr() == o[0].data_ptr()          # the set is reused two steps later
        assert o[2].dtype == torch.bool and o[4]["_final_obs"].dtype == torch.bool
        outs[copy] = rec
        env.close()
    for a, b in zip(outs[True], outs[False]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("search", ["fence", "binary", "bucket"])
def test_step_many_graph_replay_equals_plain_launches(search):
    """whole ring cycles of step_many are replayed from a hipGraph whose kernels read the launch tick from device
    memory: identical to plain launches, also with a remainder, with ordinary steps in between (the device tick must
    follow the engine's) and after switching to another set of ring buffers (graph rebuilt)"""
    tab = oracle.anymdp_synth(seed=12, task_index_base=0, n_task=8, S=64, A=8, s0_max=4)
    n, P = 512, 8
    rng = np.random.RandomState(5)
    acts_np = rng.randint(0, 8, (P, n)).astype(np.int32)
    acts2 = rng.randint(0, 8, (P, n)).astype(np.int32)
    single = rng.randint(0, 8, n).astype(np.int32)
    res = []
    for graph in (True, False):
        env = AnyMDPVecEnv(n, seed=77, autoreset_mode="same_step")
        env.set_task(_dev_tables(tab))
        env.set_search(search)
        env.set_step_many_graph(graph)
        env.reset()
        rec = []
        acts = torch.as_tensor(acts_np, device=env.device)        # device tensors: the marshalled argument list is cached
        ring = env.step_many(3 * P + 5, acts)                    # 3 cycles + remainder
        rec.append({k: _np(v).copy() for k, v in ring.items()})
        o = env.step(single)                                      # an ordinary step moves the engine tick
        rec.append({"obs": _np(o[0]), "reward": _np(o[1])})
        ring = env.step_many(2 * P, acts, out=ring)               # same arrays: cached graph, tick re-synchronised
        rec.append({k: _np(v).copy() for k, v in ring.items()})
        ring2 = env.step_many(P, acts2)                           # other arrays: graph rebuilt
        rec.append({k: _np(v).copy() for k, v in ring2.items()})
        for _ in range(3):                                        # single cycles: the two tick words alternate
            ring2 = env.step_many(P, acts2, out=ring2)
            rec.append({k: _np(v).copy() for k, v in ring2.items()})
        s, st, _ = env.get_state()
        rec.append({"state": _np(s), "steps": _np(st)})
        assert env.check_errors() == 0
        res.append(rec)
        env.close()
    for a, b in zip(*res):
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    assert res[0][0]["terminated"].sum() > 50


def test_engine_timing_events_bracket_a_burst_of_steps():
    """xv_engine_event_*: two events on the engine's stream; elapsed time of a 64-step burst is positive and below the
    host's wall clock around the same region"""
    import time
    tab = oracle.anymdp_synth(seed=3, task_index_base=0, n_task=4, S=64, A=8, s0_max=4)
    env = AnyMDPVecEnv(1024, seed=1, autoreset_mode="same_step")
    env.set_task(_dev_tables(tab))
    env.reset()
    acts = torch.randint(0, 8, (8, 1024), device=env.device, dtype=torch.int32)
    ring = env.step_many(8, acts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.engine.event_record(0)
    env.step_many(64, acts, out=ring)
    env.engine.event_record(1)
    while not env.engine.event_done(1):
        pass
    wall_ms = (time.perf_counter() - t0) * 1e3
    assert env.engine.event_done(0)
    ms = env.engine.event_elapsed_ms()
    assert 0.0 < ms <= wall_ms
    env.close()


@pytest.mark.parametrize("case", ["golden16", "synth64", "synth100", "sampled64"])
def test_device_value_iteration_vs_oracle_and_numpy(case):
    """xv_anymdp_solve (register path for S <= 64, S*A <= 512; strided path otherwise) against the oracle's restatement
    (xo_anymdp_solve: same tables, same sweeps) and the host value iteration: sweep counts equal, Q to 1e-9, greedy
    actions equal wherever the best two Q values are not within 1e-7 of each other"""
    from xenoverse_amd.anymdp.task_sampler import value_iteration
    from xenoverse_amd.anymdp import from_blocked
    if case == "golden16":
        tasks = [load_anymdp_golden(p)[1] for p in FILES if "16x4" in p]
        tab = build_tables(tasks)
    elif case == "sampled64":
        tab = build_tables([load_anymdp_golden(p)[1] for p in FILES if "64x8" in p])
    else:
        S, A = (64, 8) if case == "synth64" else (100, 5)
        tab = oracle.anymdp_synth(seed=21, task_index_base=0, n_task=6, S=S, A=A, s0_max=3)
    S, A = tab["S"], tab["A"]
    n_task = len(tab["max_steps"])
    env = AnyMDPVecEnv(n_task * 2)
    env.set_task(_dev_tables(tab) if "rows" not in tab else tab)
    q, g, it = env.solve(gamma=0.99)
    q, g, it = _np(q), _np(g), _np(it)
    cdf, rs = from_blocked(_np(env._tab["rows"]), S)
    otab = dict(S=S, A=A, s0_max=int(tab["s0_max"]), cdf=cdf, rs=rs, state_map=_np(env._tab["state_map"]),
                term_mask=_np(env._tab["term_mask"]).view(np.uint64), s0_cdf=_np(env._tab["s0_cdf"]),
                s0_ids=_np(env._tab["s0_ids"]), max_steps=_np(env._tab["max_steps"]))
    qo, go, ito = oracle.AnyMDPOracle(otab, np.zeros(1, np.int32)).solve(0.99)
    assert np.max(np.abs(it.astype(np.int64) - ito)) <= 1 and np.allclose(q, qo, rtol=1e-9, atol=1e-9)
    srt_o = np.sort(qo, -1)
    clear_o = (srt_o[..., -1] - srt_o[..., -2]) > 1e-7
    assert np.array_equal(g[clear_o], go[clear_o])
    for t in range(n_task):
        T = np.diff(np.concatenate([np.zeros((S, A, 1)), cdf[t]], -1), axis=-1)
        term = np.array([(int(np.asarray(tab["term_mask"]).view(np.uint64)[t][s >> 6]) >> (s & 63)) & 1 for s in range(S)], bool)
        T[term] = 0.0
        Qh = value_iteration(T, rs[t][..., 0].astype(np.float64), 0.99)
        assert np.allclose(q[t], Qh, rtol=1e-9, atol=1e-9), np.abs(q[t] - Qh).max()
        srt = np.sort(Qh, 1)
        clear = (srt[:, -1] - srt[:, -2]) > 1e-7
        assert np.array_equal(g[t][clear], Qh.argmax(1)[clear])
    assert it.min() > 10 and it.max() < 20000
    # the teacher rollout solves on the device when no table is given
    env.reset()
    out = env.rollout_teacher(8)
    st = _np(env.get_state()[0])
    assert out["action"].shape == (8, n_task * 2) and st.shape == (n_task * 2,)
    env.close()


def test_bandit_tasks_terminate_every_step():
    from xenoverse_amd.anymdp import AnyMDPTaskSampler
    tasks = [AnyMDPTaskSampler(1, 4, seed=k) for k in range(3)]
    n = 96
    env = AnyMDPVecEnv(n, autoreset_mode="same_step", seed=2)
    env.set_task(tasks)
    obs, _ = env.reset()
    exp_obs = np.repeat([int(t["state_mapping"][0]) for t in tasks], n // 3)
    assert np.array_equal(_np(obs), exp_obs)
    rng = np.random.RandomState(0)
    for _ in range(5):
        a = rng.randint(0, 4, n).astype(np.int32)
        obs, r, term, trunc, info = env.step(a)
        assert _np(term).all() and np.array_equal(_np(obs), exp_obs) and np.array_equal(_np(info["final_obs"]), exp_obs)
        rgt = np.array([tasks[i // (n // 3)]["reward"][0, a[i], 0] for i in range(n)], np.float32)
        assert np.array_equal(_np(info["reward_gt"]), rgt)
    assert env.check_errors() == 0
    env.close()


@pytest.mark.parametrize("n_bucket", [16, 32, 64])
def test_bucket_search_with_rows_that_overflow_a_line(n_bucket):
    """bucket search: rows built so that one bucket of probability 1 / n_bucket holds far more than 7 next states (a
    run of tiny probabilities) force the per-wave fall-back to the fence path; others resolve in the one line.  Injected
    uniforms incl. exact CDF entries and bucket edges: identical to the oracle and to the fence search."""
    S, A, n_task, n_env = 100, 3, 5, 1500
    rng = np.random.RandomState(n_bucket)
    T = np.zeros((n_task, S, A, S))
    for t in range(n_task):
        for s_ in range(S):
            for a in range(A):
                w = rng.uniform(0.2, 1.0, S) * (rng.random_sample(S) < 0.3)
                w[rng.randint(0, S)] += 1.0
                if (s_ + a) % 2 == 0:                       # a run of 30 states of probability ~1e-4 each
                    lo = rng.randint(0, S - 30)
                    w[lo:lo + 30] = 1e-4 * rng.uniform(0.5, 1.5, 30) * w.sum()
                T[t, s_, a] = w / w.sum()
    cdf = np.cumsum(T, axis=-1); cdf /= cdf[..., -1:]
    rs = rng.standard_normal((n_task, S, A, S, 2)).astype(np.float32)
    tab = dict(S=S, A=A, s0_max=2, cdf=cdf, rs=rs, state_map=np.tile(np.arange(S, dtype=np.int32), (n_task, 1)),
               term_mask=np.zeros((n_task, 2), np.uint64), s0_cdf=np.tile(np.array([0.5, 1.0]), (n_task, 1)),
               s0_ids=np.tile(np.array([0, 1], np.int32), (n_task, 1)), max_steps=np.full(n_task, 1000, np.int32))
    env_task = rng.randint(0, n_task, n_env).astype(np.int32)
    env = AnyMDPVecEnv(n_env, autoreset_mode="same_step", seed=1)
    env.set_task(_dev_tables(tab), env_task_index=env_task)
    env.set_search("bucket", n_bucket=n_bucket)
    ora = oracle.AnyMDPOracle(tab, env_task)
    ur0 = rng.random_sample(n_env)
    env.reset_injected(ur0); ora.reset_injected(ur0)
    for t in range(40):
        a = rng.randint(0, A, n_env).astype(np.int32)
        u, z, ur = rng.random_sample(n_env), rng.standard_normal(n_env).astype(np.float32), rng.random_sample(n_env)
        k = rng.randint(0, n_env, 64)
        rows = cdf[env_task[k], ora.state[k], a[k]]
        u[k[:32]] = np.minimum(rows[np.arange(32), rng.randint(0, S, 32)], np.nextafter(1.0, 0.0))   # exact CDF entries
        u[k[32:]] = rng.randint(0, n_bucket, 32) / n_bucket                                           # exact bucket edges
        d = env.step_injected(a, u, z, ur)
        o = ora.step_injected(a, u, z, ur, 2)
        _compare_step(d, o)
        s_, st, _ = env.get_state()
        assert np.array_equal(_np(s_), ora.state)
    assert env.check_errors() == 0
    env.close()


@pytest.mark.parametrize("S", [20, 21, 64])
def test_rows_whose_cdf_ends_below_one_clamp_alike_in_every_search(S):
    """caller-supplied rows whose last CDF entry stays below 1 (not what numpy.random.choice builds, but what the C-ABI
    accepts): for u >= cdf[S-1] the draw clamps to s' = S-1 and takes THAT entry's reward pair — in the per-lane search,
    on the fence path (padding entries of the last block) and in the bucket lines alike, as the oracle does."""
    A, n_task, n_env = 3, 4, 1024
    rng = np.random.RandomState(S)
    T = rng.uniform(0.1, 1.0, (n_task, S, A, S)) * (rng.random_sample((n_task, S, A, S)) < 0.5)
    T[..., 0] += 0.05
    cdf = np.cumsum(T, axis=-1)
    cdf /= cdf[..., -1:]
    cdf *= rng.uniform(0.90, 1.0, (n_task, S, A, 1))          # rows end at 0.90 .. 1.0
    rs = rng.standard_normal((n_task, S, A, S, 2)).astype(np.float32)
    tab = dict(S=S, A=A, s0_max=2, cdf=cdf, rs=rs, state_map=np.tile(np.arange(S, dtype=np.int32)[::-1], (n_task, 1)),
               term_mask=np.zeros((n_task, 1), np.uint64), s0_cdf=np.tile(np.array([0.5, 1.0]), (n_task, 1)),
               s0_ids=np.tile(np.array([0, 1], np.int32), (n_task, 1)), max_steps=np.full(n_task, 1000, np.int32))
    env_task = rng.randint(0, n_task, n_env).astype(np.int32)
    for search in ("binary", "fence", "bucket"):
        env = AnyMDPVecEnv(n_env, autoreset_mode="same_step", seed=1)
        env.set_task(_dev_tables(tab), env_task_index=env_task)
        env.set_search(search, n_bucket=16) if search == "bucket" else env.set_search(search)
        ora = oracle.AnyMDPOracle(tab, env_task)
        r2 = np.random.RandomState(3)
        ur0 = r2.random_sample(n_env)
        env.reset_injected(ur0); ora.reset_injected(ur0)
        beyond = 0
        for t in range(12):
            a = r2.randint(0, A, n_env).astype(np.int32)
            u, z, ur = r2.random_sample(n_env), r2.standard_normal(n_env).astype(np.float32), r2.random_sample(n_env)
            last = cdf[env_task, ora.state, a, -1]
            k = r2.random_sample(n_env) < 0.5                  # half of the envs draw at or beyond the row's last entry
            u[k] = np.minimum(last[k] + r2.uniform(0.0, 0.05, k.sum()) * (r2.random_sample(k.sum()) < 0.8),
                              np.nextafter(1.0, 0.0))
            beyond += int((u >= last).sum())
            d = env.step_injected(a, u, z, ur)
            o = ora.step_injected(a, u, z, ur, 2)
            _compare_step(d, o)
            s_, _, _ = env.get_state()
            assert np.array_equal(_np(s_), ora.state)
        assert beyond > 2000 and env.check_errors() == 0
        env.close()


def test_rebuilding_the_bucket_lines_drops_the_cached_step_many_graph():
    """the step_many graph bakes the bucket lines' address and count into its kernel nodes: rebuilding the lines with
    another n_bucket (or dropping them) between two replays must not replay the old graph"""
    tab = oracle.anymdp_synth(seed=12, task_index_base=0, n_task=8, S=64, A=8, s0_max=4)
    n, P = 512, 8
    acts = np.random.RandomState(5).randint(0, 8, (P, n)).astype(np.int32)
    res = []
    for variant in ("rebuild", "fence"):
        env = AnyMDPVecEnv(n, seed=77, autoreset_mode="same_step")
        env.set_task(_dev_tables(tab))
        env.set_step_many_graph(True)
        env.reset()
        rec = []
        if variant == "rebuild":
            env.set_search("bucket", n_bucket=32)
        else:
            env.set_search("fence")
        ring = env.step_many(2 * P, acts)
        rec.append({k: _np(v).copy() for k, v in ring.items()})
        if variant == "rebuild":
            env.set_search("bucket", n_bucket=64)          # frees the 32-bucket lines the graph was built on
            torch.empty(1 << 26, dtype=torch.uint8, device="cuda:0").fill_(0xFF)   # scribble over freed memory
        ring = env.step_many(2 * P, acts, out=ring)
        rec.append({k: _np(v).copy() for k, v in ring.items()})
        if variant == "rebuild":
            assert env.lib.xv_anymdp_build_buckets(env._h, 0) == 0    # lines dropped: search falls back to AUTO
            env._n_bucket = 0
        ring = env.step_many(2 * P, acts, out=ring)
        rec.append({k: _np(v).copy() for k, v in ring.items()})
        assert env.check_errors() == 0
        res.append(rec)
        env.close()
    for a, b in zip(*res):
        for k in a:
            assert np.array_equal(a[k], b[k]), k


# ---------------------------------------------------------------------------------------------------
# round 4: bucket lines as chosen cuts (csrc/anymdp_cutline.h), their census, and what AUTO makes of it
# ---------------------------------------------------------------------------------------------------
def _host_cut_census(cdf_rows, nbk, K):
    """p_fallback of the same rows from the host build of anymdp_cutline.h (tests/native/cutline_host.cpp)"""
    import ctypes as C
    import os
    import subprocess
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(tempfile.mkdtemp(), "libcut.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so,
                           os.path.join(here, "native", "cutline_host.cpp")])
    lib = C.CDLL(so)
    rows = np.ascontiguousarray(cdf_rows, np.float64)
    out, mass = np.zeros(5, np.int64), np.zeros(1, np.float64)
    lib.cutline_check(rows.ctypes.data_as(C.c_void_p), C.c_int(rows.shape[0]), C.c_int(rows.shape[1]), C.c_int(nbk), C.c_int(K),
                      C.c_int(0), out.ctypes.data_as(C.c_void_p), mass.ctypes.data_as(C.c_void_p))
    assert out[0] == 0 and out[4] == 0
    return float(mass[0]) / rows.shape[0], int(out[3])


def test_bucket_census_of_reference_rows_and_auto_takes_the_bucket_search():
    """the golden 64x8 task of the reference's sampler: its skewed rows left 2.7e-2 of the draws to the fence search with
    consecutive-entry lines; the cut lines leave < 5e-7, the device census equals the host build of the same function,
    AUTO therefore runs the bucket search, and the golden tuples come out of it"""
    path = [p for p in FILES if "64x8" in p][0]
    g, task = load_anymdp_golden(path)
    n = len(g["ss_s"])
    dflt = AnyMDPVecEnv(n, autoreset_mode="disabled")      # the default: 1 MB of lines is within the 1-GiB budget of "auto"
    dflt.set_task(task)
    assert dflt.effective_search == "bucket" and dflt.bucket_census()["built"] == 1
    dflt.close()
    env = AnyMDPVecEnv(n, autoreset_mode="disabled", bucket_lines="off")
    env.set_task(task)
    assert env.effective_search == "fence" and env.bucket_census()["built"] == 0
    cen = env.probe_buckets(16)
    assert cen["built"] == 0 and cen["format"] == 1 and cen["cuts_per_line"] == 7 and cen["n_bucket"] == 16
    assert cen["lines"] == 64 * 8 * 16 and 0 < cen["p_fallback"] < 5e-7 and cen["auto_uses_bucket"] == 1
    assert abs(cen["fallbacks_per_launch"] - cen["p_fallback"] * n) < 1e-12
    assert env.effective_search == "fence"               # a probe allocates nothing and changes nothing
    T = g["transition"]
    live = np.setdiff1d(np.arange(64), g["s_e"])
    c = np.cumsum(T[live].reshape(-1, 64), -1)
    p_host, dirty_host = _host_cut_census(c / c[:, -1:], 16, 7)
    assert cen["lines_dirty"] == dirty_host and cen["live_rows"] == len(live) * 8
    assert abs(cen["p_fallback"] - p_host) <= 1e-9 + 1e-6 * p_host      # the device sums in fixed point, rounded up per wave
    env.set_search("auto", n_bucket=16)
    assert env.effective_search == "bucket" and env.bucket_census()["built"] == 1
    assert env.bucket_census()["p_fallback"] == cen["p_fallback"]
    env.set_state(inner_state=g["ss_s"], steps=np.zeros(n), need_reset=np.zeros(n))
    obs, r, term, trunc, info = env.step_injected(g["ss_a"], g["ss_u"], g["ss_z"], np.zeros(n))
    assert np.array_equal(_np(env.inner_state), g["ss_next"])
    assert np.array_equal(_np(obs), g["state_mapping"][g["ss_next"]])
    assert np.array_equal(_np(term).astype(np.uint8), g["ss_term"])
    assert np.array_equal(_np(info["reward_gt"]), g["ss_rgt"].astype(np.float32))
    assert close_f32(_np(r), g["ss_r"])
    env.set_search("fence")
    assert env.effective_search == "fence"
    env.set_search("binary")
    assert env.effective_search == "binary"
    assert env.check_errors() == 0
    env.close()


def test_auto_keeps_the_fence_search_when_the_census_says_lines_overflow():
    """uniform rows over 200 next states: 12.5 states per bucket of 1 / 16, far more than a line lists — the census says
    so, AUTO stays on the fence search although the lines are built, and an explicit "bucket" still gives equal results"""
    S, A, n_task, n_env = 200, 2, 3, 768
    rng = np.random.RandomState(4)
    T = np.ones((n_task, S, A, S)) * rng.uniform(0.9, 1.1, (n_task, S, A, S))
    cdf = np.cumsum(T, -1)
    cdf /= cdf[..., -1:]
    rs = rng.standard_normal((n_task, S, A, S, 2)).astype(np.float32)
    tab = dict(S=S, A=A, s0_max=2, cdf=cdf, rs=rs, state_map=np.tile(np.arange(S, dtype=np.int32), (n_task, 1)),
               term_mask=np.zeros((n_task, 4), np.uint64), s0_cdf=np.tile(np.array([0.5, 1.0]), (n_task, 1)),
               s0_ids=np.tile(np.array([0, 1], np.int32), (n_task, 1)), max_steps=np.full(n_task, 1000, np.int32))
    env_task = rng.randint(0, n_task, n_env).astype(np.int32)
    env = AnyMDPVecEnv(n_env, autoreset_mode="same_step", seed=1)      # (bucket_lines="auto": the census declines, nothing built)
    env.set_task(_dev_tables(tab), env_task_index=env_task)
    assert env.effective_search == "fence" and env.bucket_census()["built"] == 0
    cen = env.probe_buckets(16)
    assert cen["p_fallback"] > 0.2 and cen["auto_uses_bucket"] == 0 and cen["lines_dirty"] == cen["lines"]
    env.set_search("auto", n_bucket=16)                 # the census says no: nothing is built
    assert env.effective_search == "fence" and env.bucket_census()["built"] == 0
    env.set_search("bucket", n_bucket=16)
    assert env.effective_search == "bucket"
    env.set_search("auto")                              # lines exist now, AUTO still declines them
    assert env.effective_search == "fence" and env.bucket_census()["built"] == 1
    env.set_search("bucket", n_bucket=16)
    ora = oracle.AnyMDPOracle(tab, env_task)
    ur0 = rng.random_sample(n_env)
    env.reset_injected(ur0); ora.reset_injected(ur0)
    for t in range(10):
        a = rng.randint(0, A, n_env).astype(np.int32)
        u, z, ur = rng.random_sample(n_env), rng.standard_normal(n_env).astype(np.float32), rng.random_sample(n_env)
        _compare_step(env.step_injected(a, u, z, ur), ora.step_injected(a, u, z, ur, 2))
    assert env.check_errors() == 0
    env.close()


@pytest.mark.parametrize("S,big_obs", [(64, True), (300, False), (256, False), (257, True)])
def test_bucket_lines_in_the_six_cut_packing(S, big_obs):
    """observation ids above 255 or S > 256 take the wide metadata packing (6 cuts per line, format 2); S = 256 with small
    ids is the largest task of the 7-cut packing.  Skewed rows (a few heavy states among runs of tiny ones), injected draws
    on exact CDF entries and bucket edges, against the oracle."""
    A, n_task, n_env = 3, 4, 1024
    rng = np.random.RandomState(S + int(big_obs))
    T = np.exp(-rng.uniform(0, 60, (n_task, S, A, S))) * (rng.random_sample((n_task, S, A, S)) < 0.4)
    T[..., 0] += 1e-30
    cdf = np.cumsum(T, -1)
    cdf /= cdf[..., -1:]
    rs = rng.standard_normal((n_task, S, A, S, 2)).astype(np.float32)
    sm = np.stack([rng.permutation(S) for _ in range(n_task)]).astype(np.int32)
    if big_obs:
        sm = sm * 97 + 300                                # ids up to ~25,000
    tm = np.zeros((n_task, (S + 63) // 64), np.uint64)
    for t in range(n_task):
        for s_ in rng.choice(np.arange(2, S), 5, replace=False):
            tm[t, s_ // 64] |= np.uint64(1) << np.uint64(s_ % 64)
    tab = dict(S=S, A=A, s0_max=2, cdf=cdf, rs=rs, state_map=sm, term_mask=tm, s0_cdf=np.tile(np.array([0.5, 1.0]), (n_task, 1)),
               s0_ids=np.tile(np.array([0, 1], np.int32), (n_task, 1)), max_steps=np.full(n_task, 1000, np.int32),
               obs_space=np.full(n_task, int(sm.max()) + 1))
    env_task = rng.randint(0, n_task, n_env).astype(np.int32)
    env = AnyMDPVecEnv(n_env, autoreset_mode="same_step", seed=1)
    env.set_task(_dev_tables(tab), env_task_index=env_task)
    env.set_search("bucket", n_bucket=16)
    cen = env.bucket_census()
    assert cen["format"] == (2 if (big_obs or S > 256) else 1) and cen["cuts_per_line"] == (6 if cen["format"] == 2 else 7)
    assert cen["p_fallback"] < 0.05       # (S = 300: ~120 live states per row, many of middling weight)
    ora = oracle.AnyMDPOracle(tab, env_task)
    ur0 = rng.random_sample(n_env)
    env.reset_injected(ur0); ora.reset_injected(ur0)
    for t in range(24):
        a = rng.randint(0, A, n_env).astype(np.int32)
        u, z, ur = rng.random_sample(n_env), rng.standard_normal(n_env).astype(np.float32), rng.random_sample(n_env)
        k = rng.randint(0, n_env, 64)
        rows = cdf[env_task[k], ora.state[k], a[k]]
        u[k[:32]] = np.minimum(rows[np.arange(32), rng.randint(0, S, 32)], np.nextafter(1.0, 0.0))
        u[k[32:]] = rng.randint(0, 16, 32) / 16
        _compare_step(env.step_injected(a, u, z, ur), ora.step_injected(a, u, z, ur, 2))
        s_, _, _ = env.get_state()
        assert np.array_equal(_np(s_), ora.state)
    tick = env.engine.tick
    for t in range(8):                                    # free-running draws as well (Philox on the device)
        a = rng.randint(0, A, n_env).astype(np.int32)
        d = env.step(a)
        _compare_step(d, ora.step(1, 0, tick + t, a, 2), exact_reward=False)   # the normal is device log / cos vs libm
    assert env.check_errors() == 0
    env.close()


def test_default_path_builds_bucket_lines_within_an_eighth_of_the_free_memory():
    """AnyMDPVecEnv(bucket_lines="auto"), the default: 2,048 tasks of 64 x 8 need 2 GiB of lines — more than the 1 GiB that is
    always allowed, within an eighth of the free memory of an otherwise empty GPU: set_task builds them and AUTO runs the
    one-line search; with the cap taken away (or little memory free) the same call stays on the fence search"""
    from xenoverse_amd import _lib
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * 2**30:
        pytest.skip("needs an (almost) empty GPU")
    n_task, S, A, n_env = 2048, 64, 8, 4096
    got = {}
    for cap in (16 << 30, 1 << 30):
        env = AnyMDPVecEnv(n_env, seed=3, autoreset_mode="same_step")
        env.AUTO_BUCKET_CAP = cap
        d = env.device
        t = dict(S=S, A=A, s0_max=4,
                 rows=torch.empty((n_task, S, A, row_lines(S), 16), dtype=torch.float64, device=d),
                 state_map=torch.empty((n_task, S), dtype=torch.int32, device=d),
                 term_mask=torch.empty((n_task, 1), dtype=torch.int64, device=d),
                 s0_cdf=torch.empty((n_task, 4), dtype=torch.float64, device=d),
                 s0_ids=torch.empty((n_task, 4), dtype=torch.int32, device=d),
                 max_steps=torch.empty(n_task, dtype=torch.int32, device=d))
        _lib.check(env.lib.xv_anymdp_synth_tasks(env.engine.handle, 77, 0, n_task, S, A, 4, *[_lib.ptr(t[k]) for k in
                   ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")]))
        env.set_task(t, env_task_index=(torch.arange(n_env, dtype=torch.int32, device=d) // 2))
        got[cap] = (env.effective_search, env.bucket_census()["built"])
        obs, _ = env.reset()
        out = env.step(torch.zeros(n_env, dtype=torch.int32, device=d))
        got[(cap, "obs")] = _np(out[0]).copy()
        assert env.check_errors() == 0
        env.close()
        del t
        torch.cuda.empty_cache()
    assert got[16 << 30] == ("bucket", 1) and got[1 << 30][0] == "fence"
    assert np.array_equal(got[(16 << 30, "obs")], got[(1 << 30, "obs")])      # same seed, same draws: the searches agree


@pytest.mark.parametrize("copy", [True, False])
@pytest.mark.parametrize("mode", ["same_step", "next_step", "disabled"])
def test_step_writes_steps_and_done_mask_from_the_same_launch(copy, mode):
    """xv_anymdp_step_info: info["steps"] and the `_final_obs` mask come out of the step kernel itself — equal to the env's
    counters (xv_anymdp_get_state) and to terminated | truncated, in every auto-reset mode, copy=True and copy=False"""
    tasks, tab, env_task = _config1()
    n = len(env_task)
    env = AnyMDPVecEnv(n, autoreset_mode=mode, seed=4, copy=copy)
    env.set_task(tasks, env_task_index=env_task)
    ora = oracle.AnyMDPOracle(tab, env_task)
    tick = env.engine.tick
    obs, _ = env.reset()
    assert np.array_equal(_np(obs), ora.reset(4, 0, tick))
    rng = np.random.RandomState(1)
    ended = 0
    for t in range(120):
        a = rng.randint(0, 4, n).astype(np.int32)
        tick = env.engine.tick
        o, r, te, tr, info = env.step(a)
        eo = ora.step(4, 0, tick, a, MODES[mode])
        assert np.array_equal(_np(o), eo[0]) and np.array_equal(_np(te).astype(np.uint8), eo[3])
        _, st, _ = env.get_state()
        assert np.array_equal(_np(info["steps"]), _np(st)) and np.array_equal(_np(st), ora.steps)
        if mode == "same_step":
            assert np.array_equal(_np(info["_final_obs"]), _np(te) | _np(tr))
            assert np.array_equal(_np(info["final_obs"]), eo[5])
        ended += int((eo[3] | eo[4]).sum())
        if mode == "disabled" and (eo[3] | eo[4]).any():
            m = (eo[3] | eo[4]).astype(bool)
            tick = env.engine.tick
            env.reset(options={"reset_mask": m})
            ora.reset(4, 0, tick, mask=m)
    assert ended > 20 and env.check_errors() == 0
    env.close()


def test_copy_true_hands_out_tensors_that_are_never_written_again():
    """copy=True (the default): outputs come from slabs made for 64 steps at once (vector.OutputSlabs) — every step's tensors
    are fresh memory, stay what they were over the following steps (three slabs later) and equal the copy=False values"""
    tab = oracle.anymdp_synth(seed=9, task_index_base=0, n_task=8, S=64, A=8, s0_max=4)
    n, T = 1024, 150
    acts = np.random.RandomState(4).randint(0, 8, (T, n)).astype(np.int32)
    ref = []
    env = AnyMDPVecEnv(n, seed=3, copy=False)
    env.set_task(_dev_tables(tab))
    env.reset()
    for t in range(T):
        o = env.step(acts[t])
        ref.append([_np(x).copy() for x in o[:4]] + [_np(o[4][k]).copy() for k in ("steps", "reward_gt", "final_obs", "_final_obs")])
    env.close()
    env = AnyMDPVecEnv(n, seed=3)
    env.set_task(_dev_tables(tab))
    obs0, _ = env.reset()
    kept, ptrs = [], set()
    for t in range(T):
        o = env.step(acts[t])
        assert o[2].dtype == torch.bool and o[3].dtype == torch.bool and o[4]["_final_obs"].dtype == torch.bool
        kept.append(list(o[:4]) + [o[4][k] for k in ("steps", "reward_gt", "final_obs", "_final_obs")])
        ptrs.add(o[0].data_ptr())
    assert len(ptrs) == T                                   # a new tensor every step
    obs_m, _ = env.reset(options={"reset_mask": np.arange(n) % 2 == 0})      # a masked reset keeps the other envs' last observation
    assert np.array_equal(_np(obs_m)[1::2], ref[-1][0][1::2])
    for t in range(T):
        for x, y in zip(kept[t], ref[t]):
            assert np.array_equal(_np(x), y), t
    env.close()
