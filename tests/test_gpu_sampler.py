"""The device task sampler (xv_anymdp_sample_tasks, csrc/anymdp_sampler.hip) against its CPU restatement
(oracle/xeno_oracle_sampler.c part 2: same counter-based draws, same formulas), against the reference's population
(tests/golden/sampler_refpop_16x4.npz: 128 tasks of the reference's own sampler with its candidate counts) and end
to end: accepted tasks go straight into the step engine and step like the same tasks uploaded from the host."""
import os

import numpy as np
import pytest
import torch

import oracle
from util import GOLD

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _engine():
    from xenoverse_amd.engine import Engine
    return Engine("cuda:0")


@pytest.mark.parametrize("S,A,n", [(16, 4, 192), (8, 2, 64), (24, 5, 48), (32, 8, 40), (64, 8, 24), (64, 5, 16), (40, 12, 12),
                                   (96, 4, 10), (128, 5, 8), (256, 5, 4), (64, 12, 8), (200, 20, 3)])
def test_candidates_equal_the_oracle(S, A, n):
    """every candidate, accepted or not: status, start / terminal states, bands, sweep counts exact; tensors to 1e-9"""
    from xenoverse_amd.anymdp import device_sampler as ds
    eng = _engine()
    seed, base = 20251003 + S, 1000
    r = ds.sample_candidates(eng, seed, base, n, S, A, s0_max=4, tables=True, dense=True, info=True)
    eng.sync()
    st = _np(r["status"])
    T, R, Nz = _np(r["transition"]), _np(r["reward"]), _np(r["reward_noise"])
    n_acc = 0
    for i in range(n):
        o = oracle.anymdp_sample_candidate(seed, base + i, S, A)
        inf = r["info"][i]
        assert st[i] == o["status"] == inf["status"], (i, st[i], o["status"])
        n0 = int(inf["n_s0"])
        assert np.array_equal(inf["s0"][:n0], o["s_0"]) and np.allclose(inf["s0_prob"][:n0], o["s_0_prob"], rtol=1e-12, atol=0)
        assert np.array_equal(np.nonzero(inf["s_e"][:S])[0], o["s_e"]) and bool(inf["goal"]) == o["goal"]
        assert np.array_equal(inf["band_lo"][:S], o["band_lo"]) and np.array_equal(inf["band_hi"][:S], o["band_hi"])
        assert np.array_equal(inf["state_map"][:S], o["state_mapping"])
        assert abs(inf["max_steps"] - o["max_steps"]) <= 1e-12 * o["max_steps"]
        assert inf["repair_rounds"] == o["repair_rounds"]
        assert np.max(np.abs(inf["sweeps"].astype(np.int64) - o["sweeps"])) <= 1, (inf["sweeps"], o["sweeps"])
        assert np.allclose(T[i], o["transition"], rtol=1e-9, atol=1e-13)
        assert np.allclose(R[i], o["reward"], rtol=1e-9, atol=1e-9)
        assert np.allclose(Nz[i], o["reward_noise"], rtol=1e-9, atol=1e-12)
        if o["status"] in (0, 3):
            assert abs(inf["gini"] - o["gini"]) < 1e-7 and abs(inf["ent"] - o["ent"]) < 1e-7
        if o["status"] in (0, 2, 3):
            assert abs(inf["gap_min"] - o["gap_min"]) < 1e-6 * max(1.0, abs(o["gap_min"]))
        n_acc += st[i] == 0
    if n >= 12:
        assert n_acc >= 1 and (st != 0).sum() >= 1        # both outcomes occurred
    eng.close()


def test_accepted_tables_equal_the_host_table_builder():
    """the row records / start tables the kernel emits for an accepted candidate are what build_tables makes from the
    same task's dense tensors (so the step engine cannot tell a device-sampled task from an uploaded one)"""
    from xenoverse_amd.anymdp import build_tables
    from xenoverse_amd.anymdp import device_sampler as ds
    eng = _engine()
    S, A, n = 16, 4, 96
    r = ds.sample_candidates(eng, 5, 0, n, S, A, s0_max=4, tables=True, dense=True, info=True)
    eng.sync()
    st = _np(r["status"])
    acc = np.nonzero(st == 0)[0]
    assert len(acc) >= 8
    for i in acc[:12]:
        task = ds.task_dict_from_dense(_np(r["transition"][i]), _np(r["reward"][i]), _np(r["reward_noise"][i]), r["info"][i], S, A)
        ref = build_tables([task], s0_max=4)
        assert np.array_equal(_np(r["rows"][i]), ref["rows"][0])
        assert np.array_equal(_np(r["state_map"][i]), ref["state_map"][0])
        assert np.array_equal(_np(r["term_mask"][i]).view(np.uint64), ref["term_mask"][0])
        assert np.array_equal(_np(r["s0_ids"][i]), ref["s0_ids"][0]) and np.array_equal(_np(r["s0_cdf"][i]), ref["s0_cdf"][0])
        assert int(r["max_steps"][i]) == int(ref["max_steps"][0])
    eng.close()


def _pop_stats(T, s_e_mask, goal):
    live = ~s_e_mask.astype(bool)
    bw, nnz = [], []
    for k in range(len(T)):
        rows = T[k][live[k]]
        nz = rows > 0
        bw.append(np.mean([np.ptp(np.nonzero(x.any(0))[0]) + 1 for x in nz]))
        nnz.append(nz.sum(-1).mean())
    return dict(pit_frac=s_e_mask.mean(), goal=np.mean(goal), band=np.mean(bw), nnz=np.mean(nnz))


def test_population_matches_the_reference_sampler():
    """distribution test against 128 tasks of the reference's own sampler (and the 452 candidates it needed): acceptance
    rate, pitfall density, goal share, band width and non-zeros per row of the device sampler's accepted tasks lie
    within sampling error of the reference population"""
    from xenoverse_amd.anymdp import device_sampler as ds
    g = np.load(os.path.join(GOLD, "sampler_refpop_16x4.npz"))
    ref = _pop_stats(g["transition"], g["s_e_mask"], g["goal"])
    n_ref, cand_ref = len(g["seed"]), int(g["n_cand"].sum())
    eng = _engine()
    n = 4096
    r = ds.sample_candidates(eng, 99, 0, n, 16, 4, tables=False, dense=True, info=True)
    eng.sync()
    st = _np(r["status"])
    acc = st == 0
    inf = r["info"]
    dev = _pop_stats(_np(r["transition"])[acc], inf["s_e"][acc][:, :16], inf["goal"][acc])
    rate_dev, rate_ref = acc.mean(), n_ref / cand_ref
    se = np.sqrt(rate_dev * (1 - rate_dev) / cand_ref)
    assert abs(rate_dev - rate_ref) < 4 * se, (rate_dev, rate_ref)
    none_dev, none_ref = (st == 1).mean(), g["n_none"].sum() / cand_ref
    assert abs(none_dev - none_ref) < 4 * np.sqrt(none_dev * (1 - none_dev) / cand_ref), (none_dev, none_ref)
    # per-task statistics: compare means with the spread of the reference sample
    T_ref = g["transition"]
    live = ~g["s_e_mask"].astype(bool)
    per_task_band = np.array([np.mean([np.ptp(np.nonzero(x)[0]) + 1 for x in (T_ref[k][live[k]] > 0).any(1)]) for k in range(n_ref)])
    per_task_nnz = np.array([(T_ref[k][live[k]] > 0).sum(-1).mean() for k in range(n_ref)])
    assert abs(dev["band"] - ref["band"]) < 4 * per_task_band.std() / np.sqrt(n_ref), (dev["band"], ref["band"])
    assert abs(dev["nnz"] - ref["nnz"]) < 4 * per_task_nnz.std() / np.sqrt(n_ref), (dev["nnz"], ref["nnz"])
    pf = g["s_e_mask"].mean(1)
    assert abs(dev["pit_frac"] - ref["pit_frac"]) < 4 * pf.std() / np.sqrt(n_ref), (dev["pit_frac"], ref["pit_frac"])
    assert abs(dev["goal"] - ref["goal"]) < 4 * np.sqrt(0.25 / n_ref)
    eng.close()


def test_population_at_128_states_matches_the_seed_compatible_host_sampler():
    """128 x 5 (the size of the reference's Garnet default; its multi-token default is 256): the interpreted reference cannot
    produce tasks of this size here, so the population is the seed-compatible host sampler's (oracle/gen_hostpop.py: bit-equal
    to the reference wherever the reference could be run).  Band width, non-zeros per row, pitfall density, goal share and
    start-state count of the device sampler's accepted tasks within 4 standard errors of that population; and the tables go
    into the step engine and step."""
    from xenoverse_amd.anymdp import AnyMDPVecEnv
    from xenoverse_amd.anymdp import device_sampler as ds
    g = np.load(os.path.join(GOLD, "sampler_hostpop_128x5.npz"))
    n_task = 48
    out = ds.sample_tasks_device(n_task, 128, 5, seed=11, batch=96, dense=True)
    assert out["stats"]["accepted"] == n_task and tuple(out["term_mask"].shape) == (n_task, 2)
    T = _np(out["transition"])
    tm = _np(out["term_mask"]).view(np.uint64)
    se = np.array([[(int(tm[k, j >> 6]) >> (j & 63)) & 1 for j in range(128)] for k in range(n_task)], bool)
    dev = dict(band=[], nnz=[], pit_frac=se.mean(1), goal=se[:, 127].astype(float))
    for k in range(n_task):
        nz = T[k][~se[k]] > 0
        dev["band"].append(np.mean([np.ptp(np.nonzero(x.any(0))[0]) + 1 for x in nz]))
        dev["nnz"].append(nz.sum(-1).mean())
        assert np.allclose(T[k][~se[k]].sum(-1), 1.0, atol=1e-12) and not T[k][se[k]].any()
    for key in ("band", "nnz", "pit_frac", "goal"):
        a, b = np.asarray(dev[key], float), g[key]
        se_ = np.sqrt(a.var() / len(a) + b.var() / len(b)) + 1e-12
        assert abs(a.mean() - b.mean()) < 4 * se_, (key, a.mean(), b.mean(), se_)
    env = AnyMDPVecEnv(n_task * 16, seed=2, autoreset_mode="same_step")
    env.set_task({k: out[k] for k in ("S", "A", "s0_max", "rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")})
    obs, _ = env.reset()
    done = 0
    for t in range(64):
        o = env.step(torch.randint(0, 5, (n_task * 16,), device=env.device, dtype=torch.int32))
        done += int((o[2] | o[3]).sum())
    assert env.check_errors() == 0 and int(o[0].max()) < 128
    env.close()


def test_sampled_tasks_step_in_the_engine_like_uploaded_ones():
    """sample_tasks_device -> set_task (device tables, no host round trip) -> steps equal the oracle's on the same
    tasks rebuilt from the dense tensors; the k-th task does not depend on the batch size used to find it"""
    from xenoverse_amd.anymdp import AnyMDPVecEnv, build_tables
    from xenoverse_amd.anymdp import device_sampler as ds
    S, A, n_task = 16, 4, 24
    a = ds.sample_tasks_device(n_task, S, A, seed=3, batch=64, dense=True)
    b = ds.sample_tasks_device(n_task, S, A, seed=3, batch=200)
    assert a["stats"]["accepted"] == n_task and torch.equal(a["rows"], b["rows"]) and torch.equal(a["max_steps"], b["max_steps"])
    n_env = n_task * 8
    env_task = np.repeat(np.arange(n_task, dtype=np.int32), 8)
    seed = 77
    env = AnyMDPVecEnv(n_env, seed=seed, autoreset_mode="same_step")
    env.set_task({k: a[k] for k in ("S", "A", "s0_max", "rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")},
                 env_task_index=env_task)
    from xenoverse_amd.anymdp.tables import from_blocked
    cdf, rs = from_blocked(_np(a["rows"]), S)
    tab = dict(S=S, A=A, s0_max=4, cdf=cdf, rs=rs, state_map=_np(a["state_map"]), term_mask=_np(a["term_mask"]).view(np.uint64),
               s0_cdf=_np(a["s0_cdf"]), s0_ids=_np(a["s0_ids"]), max_steps=_np(a["max_steps"]))
    ora = oracle.AnyMDPOracle(tab, env_task)
    tick = env.engine.tick
    obs, _ = env.reset()
    assert np.array_equal(_np(obs), ora.reset(seed, 0, tick))
    rng = np.random.RandomState(0)
    done = 0
    for t in range(200):
        act = rng.randint(0, A, n_env).astype(np.int32)
        tick = env.engine.tick
        o, r, te, tr, info = env.step(act)
        eo, er, ergt, ete, etr, efo = ora.step(seed, 0, tick, act, 2)
        assert np.array_equal(_np(o), eo) and np.array_equal(_np(te).astype(np.uint8), ete)
        assert np.array_equal(_np(tr).astype(np.uint8), etr) and np.allclose(_np(r), er, rtol=1e-5, atol=2e-6)
        done += int((ete | etr).sum())
    assert done > 50 and env.check_errors() == 0
    env.close()


def test_throughput_is_reported():
    """a smoke-sized throughput sample (the measured numbers are in profiles/ and DESIGN.md)"""
    from xenoverse_amd.anymdp import device_sampler as ds
    out = ds.sample_tasks_device(64, 64, 8, seed=1, batch=512)
    s = out["stats"]
    assert s["accepted"] == 64 and s["candidates"] >= 64 and out["rows"].shape[0] == 64
    assert sum(s["status"].values()) == s["candidates"]


@pytest.mark.parametrize("S,n_obs,d_obs,density", [(64, 64, 2, 0.20), (16, 22, 1, 0.20), (64, 200, 4, 0.05), (300, 64, 1, 0.20), (8, 5, 3, 0.0)])
def test_device_observation_model_sampler_equals_its_oracle(S, n_obs, d_obs, density):
    """xv_anymdp_sample_observation_model (AnyPOMDPTaskSampler / MultiTokensAnyPOMDPTaskSampler's observation matrices,
    task_sampler.py:78-87, :103-117, on the device) against xo_anymdp_sample_observation_model: the same tables bit for
    bit (the k chosen cells — a key-threshold bisection on the device, a sort in the oracle — values, fixed rows, CDFs),
    and independent of how the task range is split over launches"""
    import oracle
    from xenoverse_amd import Engine
    from xenoverse_amd.anymdp.device_sampler import sample_observation_model_device
    eng = Engine("cuda:0")
    n_task = 9
    dev = sample_observation_model_device(eng, 21, n_task, S, n_obs, d_obs, density=density)
    eng.sync()
    ref = oracle.anymdp_sample_observation_model(21, 0, n_task, S, n_obs, d_obs, density=density)
    assert np.array_equal(dev.cpu().numpy(), ref)
    part = sample_observation_model_device(eng, 21, 4, S, n_obs, d_obs, density=density, task_base=3)
    eng.sync()
    assert np.array_equal(part.cpu().numpy(), ref[3:7])
    eng.close()


def test_sample_tasks_device_emits_multi_token_pomdp_tables_that_step():
    """sample_tasks_device(task_type="MTPOMDP"): MDP tables from the device task sampler + observation models from the
    device observation sampler go into AnyMDPVecEnv.set_task as they are and step on the cooperative kernel; the first
    steps are checked against the oracle built from the same tables"""
    import oracle
    from xenoverse_amd.anymdp import AnyMDPVecEnv, from_blocked
    from xenoverse_amd.anymdp.device_sampler import sample_tasks_device
    tab = sample_tasks_device(6, state_space=16, action_space=4, seed=5, task_type="MTPOMDP", observation_space=12,
                              observation_tokens=3, action_tokens=2)
    assert tab["task_type"] == "MTPOMDP" and tuple(tab["obs_cdf"].shape) == (6, 3, 16, 12)
    n = 6 * 20
    env = AnyMDPVecEnv(n, seed=3, autoreset_mode="same_step")
    env.set_task(tab)
    assert env.task_type == "MTPOMDP" and (env.no, env.do, env.da) == (12, 3, 2)
    env.set_search("bucket", n_bucket=16)
    cdf, rs = from_blocked(tab["rows"].cpu().numpy(), 16)
    host = dict(S=16, A=4, s0_max=tab["s0_max"], cdf=cdf, rs=rs, **{k: tab[k].cpu().numpy() for k in
                ("state_map", "s0_cdf", "s0_ids", "max_steps")}, term_mask=tab["term_mask"].cpu().numpy().view(np.uint64))
    env_task = np.repeat(np.arange(6, dtype=np.int32), 20)
    ora = oracle.AnyMDPTokOracle(host, env_task, tab["obs_cdf"].cpu().numpy(), 2)
    tick = env.engine.tick
    o0, _ = env.reset()
    assert np.array_equal(o0.cpu().numpy(), ora.tok_reset(3, 0, tick))
    rng = np.random.RandomState(0)
    for t in range(30):
        a = rng.randint(0, 4, (n, 2)).astype(np.int32)
        tick = env.engine.tick
        obs, r, term, trunc, info = env.step(a)
        o = ora.tok_step(3, 0, tick, a, 2)
        assert np.array_equal(obs.cpu().numpy(), o[0]) and np.array_equal(term.cpu().numpy().astype(np.uint8), o[3])
        assert np.array_equal(trunc.cpu().numpy().astype(np.uint8), o[4])
    assert env.check_errors() == 0
    env.close()
