"""GPU parity at BASELINE.json's full sizes, every env of the batch against the CPU oracle.

  config 3  linds ns=32 na=8 no=8, 65,536 envs = 1,024 tasks x 64   (MFMA and scalar step kernels)
  config 4  mazeworld 15x15, 16,384 envs = 256 tasks x 64, 64x64 frames (move + rules + ray-cast)
  config 5  the per-GPU share of the 262,144-env mixed batch over 8 GPUs: 16,384 anymdp (2b) + 8,192 linds +
            8,192 cartpole through MixedBatch, as rank 0 and as rank 5 (global env ids / task indices of that rank)

(config 2a is in test_gpu_anymdp.py: 44 GiB of tables do not fit the oracle, properties + scattered subset.)
The oracle steps whole batches in 0.1-0.4 s, so nothing is sub-sampled: integers and flags are compared with
array_equal over all envs, floats with the north-star tolerance (1e-5 rel).  Where device and host libm differ in
the last bit (logf/sincosf of Box-Muller, sinf of the Fourier target) a threshold decision (error > 10,
|x| > 2.4, ...) can in principle flip for an env that sits on the threshold: such envs must be provably within
1e-4 of it, and are re-synchronised.  State is copied device -> oracle after every step so that last-bit
differences do not compound (the per-step-from-the-same-state comparison the golden tests use as well).
"""
import numpy as np
import pytest
import torch

import oracle
from util import close_rel, frame_mismatch

pytestmark = pytest.mark.gpu
NT = 8   # oracle threads


def _np(t):
    return t.detach().cpu().numpy()


# ---------------------------------------------------------------------------------------------------
# LinDS
# ---------------------------------------------------------------------------------------------------
def _linds_tasks(n_task, seed0=0, n_distinct=64):
    from xenoverse_amd.linds import LinearDSSampler
    base = []
    for k in range(n_distinct):
        t = LinearDSSampler(32, 8, 8, seed=seed0 + k)
        t["max_steps"] = 500 if k % 4 else 14 + k // 4       # a quarter of the tasks truncate inside the test
        base.append(t)
    return [base[k % n_distinct] for k in range(n_task)]


def _linds_compare(dev, o, tag):
    """dev: the VecEnv 5-tuple; o: oracle dict.  Returns the number of finished episodes."""
    obs, r, term, trunc, info = dev
    no = _np(obs).shape[1]
    te, tr = _np(term).astype(np.uint8), _np(trunc).astype(np.uint8)
    assert np.array_equal(tr, o["truncated"]), tag
    flip = te != o["terminated"]
    if flip.any():       # only an env sitting on a threshold may flip (error > 10 or |y| > 20)
        done_o = (o["terminated"] | o["truncated"]).astype(bool)
        y = np.where(done_o[:, None], o["final_obs"], o["obs"])
        scale = np.sqrt((y.astype(np.float64) ** 2).sum(1))
        near = (np.abs(o["error"] - 10.0) < 1e-3) | (np.abs(scale - 20.0) < 2e-3)
        assert near[flip].all(), (tag, int(flip.sum()))
        assert flip.sum() <= 4, (tag, int(flip.sum()))
    ok = ~flip
    assert close_rel(_np(obs)[ok], o["obs"][ok][:, :no], 1e-5, 2e-6), tag
    assert close_rel(_np(info["command"])[ok], o["cmd"][ok][:, :no], 1e-5, 2e-6), tag
    assert close_rel(_np(info["error"])[ok], o["error"][ok], 1e-5, 2e-6), tag
    assert close_rel(_np(r)[ok], o["reward"][ok], 1e-5, 2e-5), tag
    done = (te | tr).astype(bool)
    fo = _np(info["final_obs"])
    assert close_rel(fo[done & ok], o["final_obs"][done & ok][:, :no], 1e-5, 2e-6), tag
    assert np.isfinite(_np(obs)).all() and np.isfinite(_np(r)).all()
    return int(done.sum())


def _linds_resync(env, ora):
    x, st, nr = env.get_state()
    ora.x[:] = _np(x); ora.steps[:] = _np(st); ora.need_reset[:] = _np(nr)


@pytest.mark.parametrize("path", ["mfma", "scalar"])
def test_config3_linds_65536_envs_all_vs_oracle(path):
    from xenoverse_amd.linds import LinDSVecEnv, build_tables, pad_tables
    n_task, per = 1024, 64
    n = n_task * per
    tasks = _linds_tasks(n_task)
    tab = pad_tables(build_tables(tasks))
    assert tab["NS"] == 32
    env_task = np.repeat(np.arange(n_task, dtype=np.int32), per)
    seed, base = 20250703, 3 * n
    env = LinDSVecEnv(n, autoreset_mode="same_step", seed=seed, env_id_base=base)
    env.set_task(tasks, env_task_index=env_task)
    env.set_path(path)
    ora = oracle.LinDSOracle(tab, env_task)
    tick = env.engine.tick
    obs, info = env.reset()
    o0 = ora.reset(seed, base, tick)
    assert np.array_equal(_np(obs), o0["obs"][:, :16])                 # reset: table copies, bit-exact
    assert close_rel(_np(info["error"]), o0["error"], 1e-5, 2e-6)
    x, st, nr = env.get_state()
    assert np.array_equal(_np(x), ora.x) and not _np(st).any()
    rng = np.random.RandomState(11)
    n_init = tab["ints"][env_task, 2]
    ended = 0
    # leg 1: injected noise -> the state / observation path is the same fp32 fmaf chain: bit-exact over all envs
    for t in range(10):
        a = rng.uniform(-1.3, 1.3, (n, 8)).astype(np.float32)
        z = rng.standard_normal((32, n)).astype(np.float32)
        idx = (rng.random_sample(n) * n_init).astype(np.int32)
        d = env.step_injected(a, z, idx)
        o = ora.step_injected(a, z, idx, 2)
        ended += _linds_compare(d, o, ("injected", t))
        same = _np(d[2]).astype(np.uint8) == o["terminated"]
        assert np.array_equal(_np(d[0])[same], o["obs"][same][:, :16]), t
        x, st, nr = env.get_state()
        assert np.array_equal(_np(x)[:, same], ora.x[:, same]) and np.array_equal(_np(st)[same], ora.steps[same])
        _linds_resync(env, ora)
    # leg 2: free-running Philox draws keyed by the global env id
    for t in range(14):
        a = rng.uniform(-1.3, 1.3, (n, 8)).astype(np.float32)
        tick = env.engine.tick
        d = env.step(a)
        o = ora.step(seed, base, tick, a, 2, n_threads=NT)
        ended += _linds_compare(d, o, ("free", t))
        x, st, nr = env.get_state()
        same = _np(d[2]).astype(np.uint8) == o["terminated"]
        assert close_rel(_np(x)[:, same], ora.x[:, same], 1e-5, 2e-6) and np.array_equal(_np(st)[same], ora.steps[same])
        _linds_resync(env, ora)
    assert ended > 10000, ended          # the short tasks truncated and restarted
    assert env.check_errors() == 0
    env.close()


# ---------------------------------------------------------------------------------------------------
# MazeWorld
# ---------------------------------------------------------------------------------------------------
def test_config4_maze_16384_envs_state_and_frames_vs_oracle():
    from xenoverse_amd.mazeworld import (DEFAULT_ACTION_SPACE_16, MazeTaskSampler, MazeWorldVecEnv, build_tables,
                                         make_texture_library)
    n_task, per, res = 256, 64, (64, 64)
    n = n_task * per
    tex = make_texture_library(8, 4, 4, seed=0)
    tasks = [MazeTaskSampler(n_range=(15, 16), seed=k, n_wall_textures=8, n_ground_textures=4, n_ceiling_textures=4)
             for k in range(n_task)]
    tab = build_tables(tasks)
    env_task = np.repeat(np.arange(n_task, dtype=np.int32), per)
    max_steps = 9
    env = MazeWorldVecEnv(n, resolution=res, textures=tex, autoreset_mode="same_step", max_steps=max_steps,
                          action_space_type="Discrete16")
    env.set_task(tasks, env_task_index=env_task)
    ora = oracle.MazeOracle(tab, tex, env_task, resolution=res, max_steps=max_steps)
    table = np.array(DEFAULT_ACTION_SPACE_16, np.float64)
    f0, info0 = env.reset()
    ora.reset()
    fo, co = ora.render(n_threads=NT)
    frac, worst = frame_mismatch(_np(f0), fo)
    assert frac <= 0.005 and worst <= 1, (frac, worst)
    assert np.array_equal(_np(info0["command"]), co)
    rng = np.random.RandomState(4)
    contact = 0
    n_trunc = 0
    for t in range(14):
        a = rng.randint(0, 16, n).astype(np.int32)
        a[::7] = 11 if t < 6 else a[::7]        # a share of the envs keeps walking straight into walls
        frames, r, term, trunc, info = env.step(a)
        ro, teo, tro = ora.step(table[a], 2)
        st = env.get_state()
        assert np.max(np.abs(_np(st["pos"]) - ora.pos)) < 1e-9 and np.max(np.abs(_np(st["ori"]) - ora.ori)) < 1e-9
        assert np.array_equal(_np(st["grid"]), ora.grid) and np.array_equal(_np(st["steps"]), ora.steps)
        assert np.array_equal(_np(st["cmd_idx"]), ora.cmd_idx) and np.array_equal(_np(st["cmd_age"]), ora.cmd_age)
        assert np.max(np.abs(_np(st["collision"]) - ora.collision)) < 1e-9
        assert np.array_equal(_np(term).astype(np.uint8), teo) and np.array_equal(_np(trunc).astype(np.uint8), tro)
        assert np.array_equal(_np(r), ro)
        contact += int((ora.collision > 0).sum())
        n_trunc += int(tro.sum())
        ora.pos[:] = _np(st["pos"]); ora.ori[:] = _np(st["ori"])      # no compounding of last-bit differences
        if t in (0, 5, 8, 13):            # every frame of the batch (t = 8: the SAME_STEP reset frames)
            fo, co = ora.render(n_threads=NT)
            frac, worst = frame_mismatch(_np(frames), fo)
            assert frac <= 0.005 and worst <= 1, (t, frac, worst)
            per_env = np.abs(_np(frames).astype(np.int16) - fo.astype(np.int16)).reshape(n, -1).max(1)
            assert (per_env > 0).mean() <= 0.05, (t, float((per_env > 0).mean()))
            assert np.array_equal(_np(info["command"]), co)
    assert contact > 1000 and n_trunc >= n          # wall contact happened; every env truncated (step 9)
    assert env.check_errors() == 0
    env.close()


def test_config4_maze_teachers_16384_envs_vs_oracle():
    """SURVEY 8(f)3 at config-4 size: the device SmartSLAMAgent decides for all 16,384 envs and drives them through
    episode ends; every env's exposure map, memory, path head and action against the oracle agent, every step"""
    from xenoverse_amd.mazeworld import (DEFAULT_ACTION_SPACE_16, MazeTaskSampler, MazeWorldVecEnv, SmartSLAMAgent,
                                         build_tables, make_texture_library)
    n_task, per, res = 256, 64, (64, 64)
    n = n_task * per
    tex = make_texture_library(8, 4, 4, seed=0)
    tasks = [MazeTaskSampler(n_range=(15, 16), seed=k, n_wall_textures=8, n_ground_textures=4, n_ceiling_textures=4)
             for k in range(n_task)]
    tab = build_tables(tasks)
    env_task = np.repeat(np.arange(n_task, dtype=np.int32), per)
    seed, base, max_steps = 11, 1 << 30, 8
    env = MazeWorldVecEnv(n, resolution=res, textures=tex, autoreset_mode="same_step", max_steps=max_steps,
                          action_space_type="Discrete16", seed=seed, env_id_base=base)
    env.set_task(tasks, env_task_index=env_task)
    ora = oracle.MazeOracle(tab, tex, env_task, resolution=res, max_steps=max_steps)
    table = np.array(DEFAULT_ACTION_SPACE_16, np.float64)
    env.reset(); ora.reset()
    agent = SmartSLAMAgent(maze_env=env)
    oag = oracle.MazeAgentOracle(ora, table)
    wrong = 0
    for t in range(12):
        tick = env.engine.tick
        a = _np(agent.step())
        s = {k: _np(v) for k, v in agent.inspect().items()}
        ex = ora.expose(seed, base, tick)
        assert np.array_equal(s["exposed"], ex), t
        ao = oag.act(ex)
        assert np.array_equal(s["mask"], oag.mask), t
        assert np.array_equal(s["path"], oag.path), t
        wrong += int((a != ao).sum())
        env.step(a)
        ora.step(table[a], 2)
        st = env.get_state()
        assert np.array_equal(_np(st["steps"]), ora.steps) and np.array_equal(_np(st["grid"]), ora.grid)
        ora.pos[:] = _np(st["pos"]); ora.ori[:] = _np(st["ori"])
    assert wrong <= n * 12 // 100000, wrong      # an action is an argmin over fp64 costs: ties at the last bit may flip
    assert int(ora.steps.max()) < max_steps      # every env restarted once, with a fresh agent memory
    agent.close(); env.close()


def test_fused_rollouts_at_full_size_equal_single_steps():
    """xv_linds_rollout / xv_cartpole_rollout at 65,536 envs: one launch of T steps = T launches, bit for bit"""
    from xenoverse_amd.linds import LinDSVecEnv
    from xenoverse_amd.metacontrol import CartPoleVecEnv, sample_cartpole
    n, T = 65536, 24
    ltasks = _linds_tasks(1024)
    ctasks = [sample_cartpole(seed=k) for k in range(1024)]
    g = torch.Generator(device="cuda").manual_seed(5)
    la = torch.rand((T, n, 8), generator=g, device="cuda") * 2.4 - 1.2
    ca = torch.randint(0, 2, (T, n), generator=g, device="cuda", dtype=torch.int32)
    for cls, tasks, acts, kw in ((LinDSVecEnv, ltasks, la, {}), (CartPoleVecEnv, ctasks, ca, dict(frameskip=1, max_steps=17))):
        outs = []
        for fused in (False, True):
            env = cls(n, autoreset_mode="same_step", seed=3, env_id_base=1 << 33, **kw)
            env.set_task(tasks)
            env.reset()
            if fused:
                r = env.rollout(acts)
                rec = [r["obs"], r["reward"], r["terminated"], r["truncated"]]
            else:
                rows = [env.step(acts[t]) for t in range(T)]
                rec = [torch.stack([row[k] for row in rows]) for k in range(4)]
                rec[2] = rec[2].to(torch.uint8); rec[3] = rec[3].to(torch.uint8)
            rec.append(env.get_state()[0].clone())
            outs.append(rec)
            env.close()
        assert int(outs[0][3].sum()) > 0
        for a, b in zip(*outs):
            if a.ndim == 3 and a.shape[-1] != b.shape[-1]:
                b = b[..., :a.shape[-1]]
            assert torch.equal(a, b), cls.__name__


# ---------------------------------------------------------------------------------------------------
# config 5: the mixed batch's per-GPU share, as a given rank of 8
# ---------------------------------------------------------------------------------------------------
def _cartpole_compare(dev, o, ora_state_before, tag):
    obs, r, term, trunc, info = dev
    assert np.allclose(_np(obs), o["obs"], rtol=1e-5, atol=1e-6), tag
    near = (np.abs(np.abs(ora_state_before[0]) - 2.4) < 1e-4) | (np.abs(np.abs(ora_state_before[2]) - 0.20943951) < 1e-5)
    assert near.sum() <= 8
    assert np.array_equal(_np(term).astype(np.uint8)[~near], o["terminated"][~near]), tag
    assert np.array_equal(_np(trunc).astype(np.uint8), o["truncated"]), tag
    assert np.array_equal(_np(r)[~near], o["reward"][~near]), tag
    return int((o["terminated"] | o["truncated"]).sum())


@pytest.mark.parametrize("rank", [0, 5])
def test_config5_mixed_share_of_rank_vs_oracles(rank):
    from xenoverse_amd import _lib
    from xenoverse_amd.anymdp import AnyMDPVecEnv, row_lines
    from xenoverse_amd.distributed import shard_range
    from xenoverse_amd.linds import LinDSVecEnv, build_tables, pad_tables
    from xenoverse_amd.metacontrol import CartPoleVecEnv, sample_cartpole
    from xenoverse_amd.mixed import MixedBatch
    world = 8
    tot = dict(a=131072, l=65536, c=65536)                    # 262,144 envs over 8 GPUs
    rng_ = {k: shard_range(v, rank, world) for k, v in tot.items()}
    na, nl, nc = (rng_[k][1] - rng_[k][0] for k in "alc")
    assert (na, nl, nc) == (16384, 8192, 8192)
    S, A, seed, seed_tab = 64, 8, 77, 4321
    mb = MixedBatch("cuda:0", seed=seed, streams="shared")
    ea = mb.add("a", AnyMDPVecEnv, na, env_id_base=rng_["a"][0])
    el = mb.add("l", LinDSVecEnv, nl, env_id_base=rng_["l"][0])
    ec = mb.add("c", CartPoleVecEnv, nc, env_id_base=rng_["c"][0], frameskip=1, max_steps=40)
    d = ea.device
    # anymdp 2b: 256 tasks x 64 envs, the rank's own task indices (synthetic generator = f(seed, task index))
    nt_a = na // 64
    ta = dict(S=S, A=A, s0_max=4, rows=torch.empty((nt_a, S, A, row_lines(S), 16), dtype=torch.float64, device=d),
              state_map=torch.empty((nt_a, S), dtype=torch.int32, device=d),
              term_mask=torch.empty((nt_a, 1), dtype=torch.int64, device=d),
              s0_cdf=torch.empty((nt_a, 4), dtype=torch.float64, device=d),
              s0_ids=torch.empty((nt_a, 4), dtype=torch.int32, device=d),
              max_steps=torch.empty(nt_a, dtype=torch.int32, device=d))
    _lib.check(ea.lib.xv_anymdp_synth_tasks(ea.engine.handle, seed_tab, rank * nt_a, nt_a, S, A, 4, *[_lib.ptr(ta[k]) for k in
               ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")]))
    ea.engine.sync()
    et_a = np.repeat(np.arange(nt_a, dtype=np.int32), 64)
    ora_a = oracle.AnyMDPOracle(oracle.anymdp_synth(seed=seed_tab, task_index_base=rank * nt_a, n_task=nt_a, S=S, A=A,
                                                    s0_max=4), et_a)
    # linds (32, 8, 8): 128 tasks x 64 envs
    ltasks = _linds_tasks(nl // 64, seed0=100 * rank, n_distinct=32)
    ltab = pad_tables(build_tables(ltasks))
    et_l = np.repeat(np.arange(nl // 64, dtype=np.int32), 64)
    ora_l = oracle.LinDSOracle(ltab, et_l)
    # cartpole: 1,024 tasks, neighbours on different tasks
    ctasks = [sample_cartpole(seed=1000 * rank + k) for k in range(1024)]
    params = np.array([[t["gravity"], t["masscart"], t["masspole"], t["length"]] for t in ctasks], np.float64)
    et_c = (np.arange(nc) % 1024).astype(np.int32)
    ora_c = oracle.CartPoleOracle(params, et_c, frameskip=1, max_steps=40)
    mb.set_task({"a": (ta, et_a), "l": (ltasks, et_l), "c": (ctasks, et_c)})

    ticks = {k: mb.envs[k].engine.tick for k in "alc"}
    r0 = mb.reset()
    assert np.array_equal(_np(r0["a"][0]), ora_a.reset(seed, rng_["a"][0], ticks["a"]))
    assert np.array_equal(_np(r0["l"][0]), ora_l.reset(seed, rng_["l"][0], ticks["l"])["obs"][:, :16])
    assert np.array_equal(_np(r0["c"][0]), ora_c.reset(seed, rng_["c"][0], ticks["c"]))
    rng = np.random.RandomState(rank)
    ended = dict(a=0, l=0, c=0)
    for t in range(48):
        acts = dict(a=rng.randint(0, A, na).astype(np.int32), l=rng.uniform(-1.2, 1.2, (nl, 8)).astype(np.float32),
                    c=rng.randint(0, 2, nc).astype(np.int32))
        ticks = {k: mb.envs[k].engine.tick for k in "alc"}
        out = mb.step(acts)
        # anymdp: integers exact, reward 1e-5
        o, r, te, tr, info = out["a"]
        eo, er, ergt, ete, etr, efo = ora_a.step(seed, rng_["a"][0], ticks["a"], acts["a"], 2, n_threads=NT)
        assert np.array_equal(_np(o), eo) and np.array_equal(_np(te).astype(np.uint8), ete)
        assert np.array_equal(_np(tr).astype(np.uint8), etr) and np.array_equal(_np(info["final_obs"]), efo)
        assert close_rel(_np(r), er, 1e-5, 2e-6) and close_rel(_np(info["reward_gt"]), ergt, 1e-5, 2e-6)
        s, st, nr = ea.get_state()
        assert np.array_equal(_np(s), ora_a.state) and np.array_equal(_np(st), ora_a.steps)
        ended["a"] += int((ete | etr).sum())
        # linds
        ol = ora_l.step(seed, rng_["l"][0], ticks["l"], acts["l"], 2, n_threads=NT)
        ended["l"] += _linds_compare(out["l"], ol, ("mixed linds", t))
        _linds_resync(el, ora_l)
        # cartpole
        before = ora_c.state.copy()
        oc = ora_c.step(seed, rng_["c"][0], ticks["c"], acts["c"], 2)
        ended["c"] += _cartpole_compare(out["c"], oc, before, ("mixed cartpole", t))
        s, st, nr = ec.get_state()
        ora_c.state[:] = _np(s); ora_c.steps[:] = _np(st); ora_c.need_reset[:] = _np(nr)
    assert ended["a"] > 50000 and ended["l"] > 3000 and ended["c"] > 8000, ended
    for k in "alc":
        assert mb.envs[k].check_errors() == 0
    mb.close()


def test_step_then_masked_reset_keeps_returned_tensors():
    """copy=True: tensors returned by step() are never written again — also not by a later (masked) reset or by
    render_frames (the launches write into private clones), and the entries a masked reset skips keep their values"""
    from xenoverse_amd.anymdp import AnyMDPVecEnv, to_blocked
    from xenoverse_amd.linds import LinDSVecEnv, LinearDSSampler
    from xenoverse_amd.mazeworld import MazeTaskSampler, MazeWorldVecEnv, make_texture_library
    from xenoverse_amd.metacontrol import CartPoleVecEnv, sample_cartpole
    n = 256
    mask = np.zeros(n, np.uint8); mask[::3] = 1
    mt = torch.from_numpy(mask).cuda().bool()
    # anymdp
    tab = oracle.anymdp_synth(seed=3, task_index_base=0, n_task=4, S=64, A=8, s0_max=4)
    tab["rows"] = to_blocked(tab["cdf"], tab["rs"])
    env = AnyMDPVecEnv(n, autoreset_mode="disabled", seed=1)
    env.set_task(tab)
    env.reset()
    obs = env.step(np.zeros(n, np.int32))[0]
    keep = obs.clone()
    o2, _ = env.reset(options={"reset_mask": mask})
    assert torch.equal(obs, keep) and torch.equal(o2[~mt], keep[~mt])
    assert bool((env.inner_state[mt] <= 2).all())            # the masked envs restarted in s_0 = {0, 1, 2}
    env.close()
    # anymdp, multi-token POMDP: step() hands its fresh output set out without copies
    d_obs, d_act, n_obs = 2, 2, 16
    w = np.random.RandomState(1).rand(4, d_obs, 64, n_obs) + 0.05
    oc = np.cumsum(w, -1); oc /= oc[..., -1:]; oc[..., -1] = 1.0
    tab2 = dict(tab, obs_cdf=np.ascontiguousarray(oc), n_obs=n_obs, d_obs=d_obs, d_act=d_act, task_type="MTPOMDP")
    env = AnyMDPVecEnv(n, autoreset_mode="disabled", seed=1)
    env.set_task(tab2)
    env.reset()
    a = np.random.RandomState(2).randint(0, 8, (n, d_act)).astype(np.int32)
    o1, r1, te1, tr1, i1 = env.step(a)
    keep = [x.clone() for x in (o1, r1, te1, tr1, i1["reward_gt"])]
    o2, r2, te2, tr2, i2 = env.step(a)                         # the next step writes another set
    o3, _ = env.reset(options={"reset_mask": mask})            # and a masked reset a private clone
    for x, k in zip((o1, r1, te1, tr1, i1["reward_gt"]), keep):
        assert torch.equal(x, k)
    assert o1.data_ptr() != o2.data_ptr() and torch.equal(o3[~mt], o2[~mt])
    env.close()
    # linds
    env = LinDSVecEnv(n, autoreset_mode="disabled", seed=1)
    env.set_task([LinearDSSampler(16, 8, 8, seed=k) for k in range(4)])
    env.reset()
    obs, r, te, tr, info = env.step(np.zeros((n, 8), np.float32))
    keep = [obs.clone(), info["command"].clone(), info["error"].clone()]
    o2, i2 = env.reset(options={"reset_mask": mask})
    assert torch.equal(obs, keep[0]) and torch.equal(info["command"], keep[1]) and torch.equal(info["error"], keep[2])
    assert torch.equal(o2[~mt], keep[0][~mt]) and not torch.equal(o2[mt], keep[0][mt])
    env.close()
    # cartpole
    env = CartPoleVecEnv(n, autoreset_mode="disabled", seed=1, frameskip=1)
    env.set_task([sample_cartpole(seed=k) for k in range(4)])
    env.reset()
    obs = env.step(np.zeros(n, np.int32))[0]
    keep = obs.clone()
    o2, _ = env.reset(options={"reset_mask": mask})
    assert torch.equal(obs, keep) and torch.equal(o2[~mt], keep[~mt]) and not torch.equal(o2[mt], keep[mt])
    env.close()
    # maze: reset and render_frames
    env = MazeWorldVecEnv(16, resolution=(32, 32), textures=make_texture_library(8, 4, 4, seed=0),
                          autoreset_mode="disabled", seed=1)
    env.set_task([MazeTaskSampler(n_range=(9, 10), seed=k, n_wall_textures=8, n_ground_textures=4,
                                  n_ceiling_textures=4) for k in range(2)])
    f0, _ = env.reset()
    frames = env.step(np.full(16, 3, np.int32))[0]
    keep = frames.clone()
    m16 = np.zeros(16, np.uint8); m16[::2] = 1
    f2, _ = env.reset(options={"reset_mask": m16})
    assert torch.equal(frames, keep)
    assert torch.equal(f2[1::2], keep[1::2]) and torch.equal(f2[::2], f0[::2])
    f3 = env.render_frames()
    assert torch.equal(f2[1::2], keep[1::2]) and torch.equal(frames, keep) and torch.equal(f3, f2)
    env.close()


# ---------------------------------------------------------------------------------------------------
# AnyMDP on tasks of the REFERENCE's distribution at full size (round 4): 65,536 envs = 1,024 device-sampled tasks x 64, the
# search AUTO picks after the census of the bucket lines — every env against the oracle, free-running draws
# ---------------------------------------------------------------------------------------------------
def _refdist_tables(task_type="MDP", **kw):
    from xenoverse_amd.anymdp import from_blocked
    from xenoverse_amd.anymdp.device_sampler import sample_tasks_device
    tab = sample_tasks_device(1024, 64, 8, seed=3, batch=4096, task_type=task_type, **kw)
    cdf, rs = from_blocked(_np(tab["rows"]), 64)
    host = dict(S=64, A=8, s0_max=tab["s0_max"], cdf=cdf, rs=rs, state_map=_np(tab["state_map"]), s0_cdf=_np(tab["s0_cdf"]),
                s0_ids=_np(tab["s0_ids"]), max_steps=_np(tab["max_steps"]), term_mask=_np(tab["term_mask"]).view(np.uint64))
    return tab, host


def test_reference_distribution_tasks_65536_envs_auto_search_vs_oracle():
    from xenoverse_amd.anymdp import AnyMDPVecEnv
    tab, host = _refdist_tables()
    n, seed = 65536, 11
    env_task = np.repeat(np.arange(1024, dtype=np.int32), 64)
    env = AnyMDPVecEnv(n, seed=seed, autoreset_mode="same_step")
    env.set_task({k: tab[k] for k in ("S", "A", "s0_max", "rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")},
                 env_task_index=env_task)
    env.set_search("auto", n_bucket=16)
    cen = env.bucket_census()
    assert env.effective_search == "bucket" and cen["built"] == 1 and cen["lines_dirty"] > 0      # skewed rows: some lines lump runs
    assert cen["p_fallback"] < 5e-6 and cen["fallbacks_per_launch"] <= cen["auto_limit"]
    ora = oracle.AnyMDPOracle(host, env_task)
    tick = env.engine.tick
    obs, _ = env.reset()
    assert np.array_equal(_np(obs), ora.reset(seed, 0, tick))
    rng = np.random.RandomState(0)
    done = 0
    for t in range(48):
        a = rng.randint(0, 8, n).astype(np.int32)
        tick = env.engine.tick
        o, r, te, tr, info = env.step(a)
        eo, er, ergt, ete, etr, efo = ora.step(seed, 0, tick, a, 2, n_threads=NT)
        assert np.array_equal(_np(o), eo) and np.array_equal(_np(te).astype(np.uint8), ete)
        assert np.array_equal(_np(tr).astype(np.uint8), etr) and np.array_equal(_np(info["reward_gt"]), ergt)
        assert np.array_equal(_np(info["final_obs"]), efo) and np.array_equal(_np(info["steps"]), ora.steps)
        assert np.allclose(_np(r), er, rtol=1e-5, atol=2e-6)
        s_, _, _ = env.get_state()
        assert np.array_equal(_np(s_), ora.state)
        done += int((ete | etr).sum())
    assert done > 10000 and env.check_errors() == 0
    # the same steps on the fence search give the same trajectory (states were compared above): one more step on each
    env.set_search("fence")
    a = rng.randint(0, 8, n).astype(np.int32)
    tick = env.engine.tick
    o = env.step(a)
    eo = ora.step(seed, 0, tick, a, 2, n_threads=NT)
    assert np.array_equal(_np(o[0]), eo[0])
    env.close()


def test_reference_distribution_multi_token_tasks_65536_envs_cooperative_kernel_vs_oracle():
    from xenoverse_amd.anymdp import AnyMDPVecEnv
    tab, host = _refdist_tables("MTPOMDP", observation_space=64, observation_tokens=2, action_tokens=2)
    n, seed = 65536, 12
    env_task = np.repeat(np.arange(1024, dtype=np.int32), 64)
    env = AnyMDPVecEnv(n, seed=seed, autoreset_mode="same_step")
    env.set_task(tab, env_task_index=env_task)
    env.set_search("auto", n_bucket=16)
    assert env.effective_search == "bucket" and env.token_kernel == "cooperative"
    assert env.bucket_census()["obs_lines"] == 1024 * 2 * 64 * 16
    ora = oracle.AnyMDPTokOracle(host, env_task, _np(tab["obs_cdf"]), 2)
    tick = env.engine.tick
    o0, _ = env.reset()
    assert np.array_equal(_np(o0), ora.tok_reset(seed, 0, tick))
    rng = np.random.RandomState(1)
    done = 0
    for t in range(24):
        a = rng.randint(0, 8, (n, 2)).astype(np.int32)
        tick = env.engine.tick
        obs, r, term, trunc, info = env.step(a)
        o = ora.tok_step(seed, 0, tick, a, 2)
        assert np.array_equal(_np(obs), o[0]) and np.array_equal(_np(term).astype(np.uint8), o[3])
        assert np.array_equal(_np(trunc).astype(np.uint8), o[4]) and np.array_equal(_np(info["reward_gt"]), o[2])
        assert np.allclose(_np(r), o[1], rtol=1e-5, atol=4e-6)
        done += int((o[3] | o[4]).sum())
    assert done > 5000 and env.check_errors() == 0
    env.close()
