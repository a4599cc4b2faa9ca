"""Closed-loop stepping from a captured graph (xenoverse_amd/capture.py, xv_engine_set_device_tick): the loop the reference
runs per env object — action = policy(obs); obs, r, done = env.step(action) (anymdp/test_utils.py:45-57) — captured once
in a torch.cuda.graph and replayed gives, bit for bit, the trajectory of the same calls issued eagerly with the host
tick: AnyMDP, LinDS, CartPole and the fused mixed batch, 256 steps each, unroll 1 and 4."""
import numpy as np
import pytest
import torch

import oracle
from xenoverse_amd.anymdp import AnyMDPVecEnv, to_blocked
from xenoverse_amd.linds import LinDSVecEnv, LinearDSSampler
from xenoverse_amd.metacontrol import CartPoleVecEnv, sample_cartpole
from xenoverse_amd.mixed import MixedBatch

pytestmark = pytest.mark.gpu
T = 256


def _np(t):
    return t.detach().cpu().numpy().copy()


def _anymdp_tables(n_task, seed=21):
    tab = oracle.anymdp_synth(seed=seed, task_index_base=0, n_task=n_task, S=64, A=8, s0_max=4)
    tab["rows"] = to_blocked(tab["cdf"], tab["rs"])
    out = dict(S=64, A=8, s0_max=4)
    for k in ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps"):
        v = np.ascontiguousarray(tab[k])
        out[k] = torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).cuda()
    return out


def _pol_a(obs):
    return (obs * 7 + 3) % 8


def _pol_l(obs):
    return (obs[:, :8] * -0.3).clamp(-1.5, 1.5)


def _pol_c(obs):
    return (obs[:, 2] + 0.3 * obs[:, 3] > 0).to(torch.int32)


def _make(family, copy, n=1024):
    if family == "anymdp":
        env = AnyMDPVecEnv(n, seed=9, copy=copy)
        env.set_task(_anymdp_tables(n // 64))
        return env, _pol_a, ("reward_gt", "final_obs")
    if family == "linds":
        env = LinDSVecEnv(n, seed=9, copy=copy)
        env.set_task([LinearDSSampler(16, 8, 8, seed=k) for k in range(n // 64)])
        return env, _pol_l, ("command", "error", "final_obs")
    if family == "pomdp":      # token steps (xv_anymdp_step_tokens_info) on reference-sampled POMDP tasks, cooperative kernel
        from util import golden_files, load_anymdp_tok_golden
        tasks = [load_anymdp_tok_golden(p)[1] for p in golden_files("anymdptok_") if "mtpomdp" not in p]
        env = AnyMDPVecEnv(n, seed=9, copy=copy)
        env.set_task(tasks, env_task_index=(np.arange(n) % len(tasks)).astype(np.int32))
        env.set_search("bucket", n_bucket=16)
        na = int(env.na)
        return env, (lambda obs: (obs * 5 + 1) % na), ("reward_gt", "final_obs", "steps")
    env = CartPoleVecEnv(n, seed=9, frameskip=1, copy=copy)
    env.set_task([sample_cartpole(seed=k) for k in range(16)])
    return env, _pol_c, ("final_obs",)


def _record(out, keys, finished_only=("final_obs",)):
    obs, r, te, tr, info = out
    rec = dict(obs=_np(obs), reward=_np(r), term=_np(te), trunc=_np(tr))
    done = rec["term"] | rec["trunc"]
    for k in keys:
        v = _np(info[k])
        if k in finished_only:        # rows of envs that did not finish hold what an earlier episode end left there
            v = v[done]
        rec[k] = v
    return rec


@pytest.mark.parametrize("unroll", [1, 4])
@pytest.mark.parametrize("family", ["anymdp", "linds", "cartpole", "pomdp"])
def test_captured_loop_equals_eager_calls(family, unroll):
    # eager, host tick, fresh tensors from every call
    env, pol, keys = _make(family, copy=True)
    obs, _ = env.reset()
    ref = []
    for t in range(T + 1):
        out = env.step(pol(obs))
        obs = out[0]
        ref.append(_record(out, keys))
    assert env.check_errors() == 0
    env.close()
    # captured: one eager warm-up iteration (a real step), then T steps from T / unroll graph launches
    env, pol, keys = _make(family, copy=False)
    obs, _ = env.reset()
    tick0 = env.engine.tick
    loop = env.capture(pol, obs, unroll=unroll, warmup=1)
    assert env.engine.device_tick and env.engine.tick == tick0 + 1
    got = [_record(loop.out, keys)]                  # the warm-up step
    for k in range(T // unroll):
        out = loop.replay()
        got.append(_record(out, keys))
    assert env.engine.tick == tick0 + 1 + T          # the device word moved with every replayed step
    assert env.check_errors() == 0
    for k, g in enumerate(got):
        r = ref[0] if k == 0 else ref[k * unroll]
        for name in g:
            assert np.array_equal(g[name], r[name]), (family, unroll, k, name)
    loop.close()
    env.close()


def test_device_tick_mode_steps_eagerly_as_the_host_tick_does():
    """the tick word in device memory, no capture: reset, step, masked reset and step_many draw what the host tick draws"""
    recs = []
    for dev in (False, True):
        env = AnyMDPVecEnv(512, seed=3, autoreset_mode="disabled")
        env.set_task(_anymdp_tables(8, seed=5))
        if dev:
            env.engine.set_device_tick(True)
        obs, _ = env.reset()
        rec = [_np(obs)]
        for t in range(20):
            o = env.step(_pol_a(obs))
            obs = o[0]
            done = (o[2] | o[3])
            if bool(done.any()):
                obs2, _ = env.reset(options={"reset_mask": done})
                obs = torch.where(done, obs2, obs)
            rec.append(_np(obs)); rec.append(_np(o[1]))
        assert env.engine.tick >= 21
        recs.append(rec)
        env.close()
    for a, b in zip(*recs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("unroll", [1, 4])
def test_captured_mixed_batch_equals_eager_fused_steps(unroll):
    def build(copy):
        mb = MixedBatch("cuda:0", seed=5)
        mb.add("a", AnyMDPVecEnv, 512, copy=copy)
        mb.add("l", LinDSVecEnv, 256, copy=copy)
        mb.add("c", CartPoleVecEnv, 256, frameskip=1, copy=copy)
        mb.set_task({"a": _anymdp_tables(8), "l": [LinearDSSampler(16, 8, 8, seed=k) for k in range(4)],
                     "c": [sample_cartpole(seed=k) for k in range(256)]})
        return mb

    def pol(obs):
        return {"a": _pol_a(obs["a"]), "l": _pol_l(obs["l"]), "c": _pol_c(obs["c"])}

    keys = {"a": ("reward_gt", "final_obs"), "l": ("command", "error", "final_obs"), "c": ("final_obs",)}
    mb = build(True)
    obs = {k: v[0] for k, v in mb.reset().items()}
    ref = []
    for t in range(T + 1):
        out = mb.step_fused(pol(obs))
        obs = {k: v[0] for k, v in out.items()}
        ref.append({k: _record(out[k], keys[k]) for k in "alc"})
    mb.close()
    mb = build(False)
    obs = {k: v[0] for k, v in mb.reset().items()}
    loop = mb.capture(pol, obs, unroll=unroll, warmup=1)
    got = [{k: _record(loop.out[1][k], keys[k]) for k in "alc"}]
    for k in range(T // unroll):
        out = loop.replay()
        got.append({f: _record(out[1][f], keys[f]) for f in "alc"})
    for k, g in enumerate(got):
        r = ref[0] if k == 0 else ref[k * unroll]
        for f in "alc":
            for name in g[f]:
                assert np.array_equal(g[f][name], r[f][name]), (unroll, k, f, name)
    for e in mb.envs.values():
        assert e.check_errors() == 0
    loop.close()
    mb.close()


def test_step_fused_falls_back_to_three_launches_without_touching_the_buffers():
    """handles xv_mixed_step has no instantiation for (AnyMDP on the per-lane search): step_fused gives what step() gives"""
    res = []
    for fused in (False, True):
        mb = MixedBatch("cuda:0", seed=5)
        mb.add("a", AnyMDPVecEnv, 512)
        mb.add("l", LinDSVecEnv, 256)
        mb.add("c", CartPoleVecEnv, 256, frameskip=1)
        mb.set_task({"a": _anymdp_tables(8), "l": [LinearDSSampler(16, 8, 8, seed=k) for k in range(4)],
                     "c": [sample_cartpole(seed=k) for k in range(256)]})
        mb.envs["a"].set_search("binary")
        obs = {k: v[0] for k, v in mb.reset().items()}
        rec = []
        for t in range(6):
            acts = {"a": _pol_a(obs["a"]), "l": _pol_l(obs["l"]), "c": _pol_c(obs["c"])}
            out = mb.step_fused(acts) if fused else mb.step(acts)
            obs = {k: v[0] for k, v in out.items()}
            rec.append({k: (_np(out[k][0]), _np(out[k][1])) for k in "alc"})
        res.append(rec)
        mb.close()
    for a, b in zip(*res):
        for k in "alc":
            assert np.array_equal(a[k][0], b[k][0]) and np.array_equal(a[k][1], b[k][1])


@pytest.mark.parametrize("dev_tick", [False, True])
def test_fused_mixed_step_with_one_shared_engine_equals_three_separate_steps(dev_tick):
    """handles that SHARE an engine (one tick word): xv_mixed_step must hand the families the ticks T, T + 1, T + 2 that
    three separate step calls take — also with the tick in device memory, where one advance per call used to give all three
    the same tick"""
    from xenoverse_amd.engine import Engine
    res = []
    for fused in (False, True):
        eng = Engine("cuda:0", seed=5)
        mb = MixedBatch("cuda:0", seed=5)
        mb.envs = {"a": AnyMDPVecEnv(512, engine=eng), "l": LinDSVecEnv(256, engine=eng),
                   "c": CartPoleVecEnv(256, frameskip=1, engine=eng)}
        mb.streams = {k: eng.torch_stream for k in mb.envs}
        mb.envs["a"].set_task(_anymdp_tables(8))
        mb.envs["l"].set_task([LinearDSSampler(16, 8, 8, seed=k) for k in range(4)])
        mb.envs["c"].set_task([sample_cartpole(seed=k) for k in range(256)])
        if dev_tick:
            eng.set_device_tick(True)
        obs = {k: e.reset()[0] for k, e in mb.envs.items()}
        rec = []
        for t in range(12):
            acts = {"a": _pol_a(obs["a"]), "l": _pol_l(obs["l"]), "c": _pol_c(obs["c"])}
            if fused:
                out = mb.step_fused(acts)
            else:
                out = {k: mb.envs[k].step(acts[k]) for k in "alc"}      # anymdp, linds, cartpole: the order of the fused call
            obs = {k: v[0] for k, v in out.items()}
            rec.append({k: (_np(out[k][0]), _np(out[k][1]), _np(out[k][2])) for k in "alc"})
        assert eng.tick == 3 + 3 * 12
        res.append(rec)
        for e in mb.envs.values():
            e.close()
        eng.close()
    for a, b in zip(*res):
        for k in "alc":
            for x, y in zip(a[k], b[k]):
                assert np.array_equal(x, y), k
