"""BASELINE config 5 (scaled down): anymdp + linds + cartpole stepped concurrently on separate streams; every
family's trajectory equals its standalone run bit for bit, and a rank's shard equals its slice of the batch."""
import numpy as np
import pytest
import torch

import oracle
from xenoverse_amd.anymdp import AnyMDPVecEnv, to_blocked
from xenoverse_amd.distributed import shard_range
from xenoverse_amd.linds import LinDSVecEnv, LinearDSSampler
from xenoverse_amd.metacontrol import CartPoleVecEnv, sample_cartpole
from xenoverse_amd.mixed import MixedBatch

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _anymdp_tables(n_task):
    tab = oracle.anymdp_synth(seed=21, task_index_base=0, n_task=n_task, S=64, A=8, s0_max=4)
    tab["rows"] = to_blocked(tab["cdf"], tab["rs"])
    out = dict(S=64, A=8, s0_max=4)
    for k in ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps"):
        v = np.ascontiguousarray(tab[k])
        out[k] = torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).cuda()
    return out


def _run(n_a, n_l, n_c, lo_frac=(0.0, 1.0), mixed=True, T=12, seed=5, streams="separate"):
    """step the three families for T steps; returns per-family stacked observations / rewards"""
    full = dict(a=512, l=256, c=256)
    sl = {k: (int(lo_frac[0] * v), int(lo_frac[1] * v)) for k, v in full.items()}
    tabs = _anymdp_tables(full["a"] // 64)
    ltasks = [LinearDSSampler(16, 8, 8, seed=k) for k in range(full["l"] // 64)]
    ctasks = [sample_cartpole(seed=k) for k in range(full["c"])]
    et_a = np.repeat(np.arange(full["a"] // 64, dtype=np.int32), 64)[sl["a"][0]:sl["a"][1]]
    et_l = np.repeat(np.arange(full["l"] // 64, dtype=np.int32), 64)[sl["l"][0]:sl["l"][1]]
    et_c = np.arange(full["c"], dtype=np.int32)[sl["c"][0]:sl["c"][1]]
    rng = np.random.RandomState(0)
    acts = dict(a=rng.randint(0, 8, (T, full["a"])).astype(np.int32),
                l=rng.uniform(-1, 1, (T, full["l"], 8)).astype(np.float32),
                c=rng.randint(0, 2, (T, full["c"])).astype(np.int32))
    rec = {k: [] for k in "alc"}
    if mixed:
        mb = MixedBatch("cuda:0", seed=seed, streams=streams)
        mb.add("a", AnyMDPVecEnv, len(et_a), env_id_base=sl["a"][0])
        mb.add("l", LinDSVecEnv, len(et_l), env_id_base=sl["l"][0])
        mb.add("c", CartPoleVecEnv, len(et_c), env_id_base=sl["c"][0], frameskip=1)
        mb.set_task({"a": (tabs, et_a), "l": (ltasks, et_l), "c": (ctasks, et_c)})
        mb.reset()
        for t in range(T):
            out = mb.step({k: acts[k][t, sl[k][0]:sl[k][1]] for k in "alc"})
            for k in "alc":
                rec[k].append((_np(out[k][0]), _np(out[k][1])))
        mb.close()
    else:
        envs = dict(a=AnyMDPVecEnv(len(et_a), seed=seed, env_id_base=sl["a"][0]),
                    l=LinDSVecEnv(len(et_l), seed=seed, env_id_base=sl["l"][0]),
                    c=CartPoleVecEnv(len(et_c), seed=seed, env_id_base=sl["c"][0], frameskip=1))
        envs["a"].set_task(tabs, env_task_index=et_a); envs["l"].set_task(ltasks, env_task_index=et_l)
        envs["c"].set_task(ctasks, env_task_index=et_c)
        for k in "alc":
            envs[k].reset()
            for t in range(T):
                o = envs[k].step(acts[k][t, sl[k][0]:sl[k][1]])
                rec[k].append((_np(o[0]), _np(o[1])))
            envs[k].close()
    return {k: (np.stack([x[0] for x in v]), np.stack([x[1] for x in v])) for k, v in rec.items()}


@pytest.mark.parametrize("streams", ["separate", "shared"])
def test_mixed_equals_standalone(streams):
    m = _run(0, 0, 0, mixed=True, streams=streams)
    s = _run(0, 0, 0, mixed=False)
    for k in "alc":
        assert np.array_equal(m[k][0], s[k][0]) and np.array_equal(m[k][1], s[k][1]), k


def test_rank_shards_reproduce_the_unsharded_batch():
    full = _run(0, 0, 0, mixed=True)
    for r in range(2):
        part = _run(0, 0, 0, lo_frac=(r * 0.5, (r + 1) * 0.5), mixed=True)
        for k, n in (("a", 512), ("l", 256), ("c", 256)):
            lo, hi = shard_range(n, r, 2)
            assert np.array_equal(full[k][0][:, lo:hi], part[k][0]), (k, r)
            assert np.array_equal(full[k][1][:, lo:hi], part[k][1]), (k, r)


@pytest.mark.parametrize("family", ["linds", "cartpole", "acrobot", "maze"])
def test_copy_false_gives_the_same_values(family):
    """copy=False returns views of the engine-owned output buffers (no per-step device copies): same values"""
    from xenoverse_amd.metacontrol import AcrobotVecEnv, sample_acrobot
    from xenoverse_amd.mazeworld import MazeTaskSampler, MazeWorldVecEnv, make_texture_library
    rng = np.random.RandomState(0)
    recs = []
    for copy in (True, False):
        if family == "linds":
            env = LinDSVecEnv(128, seed=4, copy=copy)
            env.set_task([LinearDSSampler(16, 8, 8, seed=k) for k in range(2)])
            acts = rng.uniform(-1, 1, (6, 128, 8)).astype(np.float32) if copy else acts
        elif family == "cartpole":
            env = CartPoleVecEnv(128, seed=4, frameskip=1, copy=copy)
            env.set_task([sample_cartpole(seed=k) for k in range(4)])
            acts = rng.randint(0, 2, (6, 128)).astype(np.int32) if copy else acts
        elif family == "acrobot":
            env = AcrobotVecEnv(128, seed=4, frameskip=1, copy=copy)
            env.set_task([sample_acrobot(seed=k) for k in range(4)])
            acts = rng.randint(0, 3, (6, 128)).astype(np.int32) if copy else acts
        else:
            env = MazeWorldVecEnv(16, seed=4, resolution=(32, 32), textures=make_texture_library(8, 4, 4, seed=0),
                                  max_steps=4, copy=copy)
            env.set_task([MazeTaskSampler(n_range=(9, 10), seed=k, n_wall_textures=8, n_ground_textures=4,
                                          n_ceiling_textures=4) for k in range(2)])
            acts = rng.randint(0, 16, (6, 16)).astype(np.int32) if copy else acts
        obs, info = env.reset()
        rec = [obs.detach().cpu().numpy().copy()]
        for t in range(6):
            o = env.step(acts[t])
            rec += [o[k].detach().cpu().numpy().copy() for k in range(4)]
            rec += [v.detach().cpu().numpy().copy() for k, v in sorted(o[4].items())]
            assert o[2].dtype == torch.bool
        recs.append(rec)
        env.close()
    assert len(recs[0]) == len(recs[1])
    for a, b in zip(*recs):
        assert np.array_equal(a, b)


def test_rollout_record_packing_on_device_matches_host_format():
    """xv_pack_rollout / xv_unpack_rollout: one 8-byte record per env-step, the same bits as the torch (CPU) packer"""
    from xenoverse_amd.distributed import REC_BYTES, pack_records, unpack_records
    g = torch.Generator().manual_seed(1)
    T, N = 7, 1000
    obs = torch.randint(0, 65536, (T, N), generator=g, dtype=torch.int32)
    act = torch.randint(0, 256, (T, N), generator=g, dtype=torch.int32)
    rew = torch.randn((T, N), generator=g)
    te = (torch.rand((T, N), generator=g) < 0.3).to(torch.uint8)
    tr = (torch.rand((T, N), generator=g) < 0.3).to(torch.uint8)
    host = pack_records(obs, act, rew, te, tr)
    dev = pack_records(obs.cuda(), act.cuda(), rew.cuda(), te.cuda(), tr.cuda())
    assert dev.shape == (T, N, REC_BYTES) and torch.equal(dev.cpu(), host)
    o, a, r, t1, t2 = unpack_records(dev)
    assert torch.equal(o.cpu(), obs) and torch.equal(a.cpu(), act) and torch.equal(r.cpu(), rew)
    assert torch.equal(t1.cpu(), te) and torch.equal(t2.cpu(), tr)


def test_rollout_allgather_over_rccl_through_the_c_abi():
    """xv_rccl_unique_id / xv_rccl_comm_create / xv_rollout_allgather on the GPU of this box: a one-rank communicator
    (what a 1-GPU box can host) gathers a packed rollout chunk on the side stream — RCCL itself executes; ranks > 1 use
    the same calls with the id carried by a TCP store (covered on CPU by the gloo tests of the torch transport)"""
    from xenoverse_amd.distributed import REC_BYTES, RolloutGather, pack_records, unpack_records
    T, N = 32, 4096
    g = torch.Generator(device="cuda").manual_seed(3)
    obs = torch.randint(0, 64, (T, N), generator=g, device="cuda", dtype=torch.int32)
    act = torch.randint(0, 8, (T, N), generator=g, device="cuda", dtype=torch.int32)
    rew = torch.randn((T, N), generator=g, device="cuda")
    te = (torch.rand((T, N), generator=g, device="cuda") < 0.2).to(torch.uint8)
    tr = (torch.rand((T, N), generator=g, device="cuda") < 0.1).to(torch.uint8)
    rg = RolloutGather((T, N, REC_BYTES), device="cuda:0", transport="rccl", rank=0, world=1)
    assert rg.comm.count() == 1            # ncclCommCount through the C-ABI: what bench.py prints as `rccl_ranks`
    for _ in range(3):
        pack_records(obs, act, rew, te, tr, out=rg.local)
        rg.launch()
        out = rg.wait()
    torch.cuda.synchronize()
    assert out.shape == (1, T, N, REC_BYTES) and torch.equal(out[0], rg.local)
    o2, a2, r2, te2, tr2 = unpack_records(out[0])
    assert torch.equal(o2, obs) and torch.equal(a2, act) and torch.equal(r2, rew) and torch.equal(te2, te) and torch.equal(tr2, tr)
    rg.close()


@pytest.mark.parametrize("search", ["fence", "bucket"])
@pytest.mark.parametrize("mode", ["same_step", "next_step"])
@pytest.mark.parametrize("copy", [True, False])
def test_fused_mixed_step_equals_separate_steps(mode, search, copy):
    """xv_mixed_step: ONE launch for an AnyMDP + LinDS + CartPole batch (the families' step bodies share a grid) — every
    output, every state and every engine tick equals stepping the three families one after the other; ragged env counts
    (partial last workgroups / tiles) included"""
    import oracle
    from xenoverse_amd.anymdp import AnyMDPVecEnv, to_blocked
    from xenoverse_amd.linds import LinDSVecEnv, LinearDSSampler
    from xenoverse_amd.metacontrol import CartPoleVecEnv, sample_cartpole
    from xenoverse_amd.mixed import MixedBatch
    na, nl, nc, S, A = 1000, 17 * 16 + 5, 700, 64, 8
    tab = oracle.anymdp_synth(seed=3, task_index_base=0, n_task=10, S=S, A=A, s0_max=4)
    dev = dict(S=S, A=A, s0_max=4)
    tab["rows"] = to_blocked(tab["cdf"], tab["rs"])
    for k in ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps"):
        v = np.ascontiguousarray(tab[k])
        dev[k] = torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).cuda()
    ltasks = [LinearDSSampler(16, 8, 8, seed=k) for k in range(3)]
    for t in ltasks:
        t["max_steps"] = 30
    ctasks = [sample_cartpole(seed=k) for k in range(5)]
    a_task = (np.arange(na) % 10).astype(np.int32)
    l_task = np.sort(np.arange(nl) % 3).astype(np.int32)
    rng = np.random.RandomState(1)
    T = 40
    acts = dict(a=rng.randint(0, A, (T, na)).astype(np.int32), l=rng.uniform(-1.3, 1.3, (T, nl, 8)).astype(np.float32),
                c=rng.randint(0, 2, (T, nc)).astype(np.int32))
    recs = []
    for fused in (False, True):
        mb = MixedBatch("cuda:0", seed=11)
        # copy=False: step_fused takes its persistent path (structs and views made once, steps / done masks from the ONE launch)
        ea = mb.add("a", AnyMDPVecEnv, na, autoreset_mode=mode, copy=copy)
        el = mb.add("l", LinDSVecEnv, nl, autoreset_mode=mode, copy=copy)
        ec = mb.add("c", CartPoleVecEnv, nc, frameskip=1, autoreset_mode=mode, copy=copy)
        mb.set_task({"a": (dev, a_task), "l": (ltasks, l_task), "c": ctasks})
        ea.set_search(search, n_bucket=16) if search == "bucket" else ea.set_search(search)
        mb.reset()
        rec = []
        for t in range(T):
            out = (mb.step_fused if fused else mb.step)({k: v[t] for k, v in acts.items()})
            for name in ("a", "l", "c"):
                o, r, te, tr, info = out[name]
                rec += [_np(o).copy(), _np(r).copy(), _np(te).copy(), _np(tr).copy()]
                # the LinDS final_obs rows of unfinished envs are unspecified with copy=False: compare them under the mask
                for k in sorted(info):
                    if not torch.is_tensor(info[k]):
                        continue
                    v = _np(info[k]).copy()
                    if k == "final_obs" and not copy and "_final_obs" in info:
                        v = v[_np(info["_final_obs"]).astype(bool)]
                    rec.append(v)
        rec += [_np(x) for x in ea.get_state()] + [_np(x) for x in el.get_state()] + [_np(x) for x in ec.get_state()]
        rec += [np.int64(ea.engine.tick), np.int64(el.engine.tick), np.int64(ec.engine.tick)]
        assert ea.check_errors() == 0
        recs.append(rec)
        mb.close()
    assert len(recs[0]) == len(recs[1])
    for x, y in zip(*recs):
        assert np.array_equal(x, y)


def test_create_and_close_return_device_memory():
    """every family handle frees what it allocated (tables' engine-side copies, bucket lines, compaction lists, ...): after
    a warm-up round, 12 rounds of create -> set_task -> reset -> step -> close leave the free device memory where it was"""
    from xenoverse_amd.anymdp import AnyMDPVecEnv, to_blocked
    from xenoverse_amd.linds import LinDSVecEnv, LinearDSSampler
    from xenoverse_amd.mazeworld import MazeTaskSampler, MazeWorldVecEnv, make_texture_library
    from xenoverse_amd.metacontrol import AcrobotVecEnv, CartPoleVecEnv, sample_acrobot, sample_cartpole
    tab = oracle.anymdp_synth(seed=3, task_index_base=0, n_task=16, S=64, A=8, s0_max=4)
    tab["rows"] = to_blocked(tab["cdf"], tab["rs"])
    lt = [LinearDSSampler(16, 8, 8, seed=k) for k in range(4)]
    mz = [MazeTaskSampler(n_range=(9, 10), seed=k, n_wall_textures=8, n_ground_textures=4, n_ceiling_textures=4) for k in range(2)]
    tex = make_texture_library(8, 4, 4, seed=0)

    def round_():
        e = AnyMDPVecEnv(12288, seed=1, autoreset_mode="same_step")
        e.set_task(tab); e.set_search("bucket", n_bucket=16); e.reset()
        e.step(np.zeros(12288, np.int32)); e.step_many(8, torch.zeros((8, 12288), dtype=torch.int32, device=e.device)); e.close()
        e = LinDSVecEnv(4096, seed=1, autoreset_mode="same_step")
        e.set_task(lt); e.reset(); e.step(np.zeros((4096, 8), np.float32)); e.close()
        e = MazeWorldVecEnv(12288, resolution=(16, 16), textures=tex, autoreset_mode="same_step", seed=1)
        e.set_task(mz); e.reset(); e.step(np.zeros(12288, np.int32)); e.close()      # 12,288 envs: the sorted move kernel
        e = CartPoleVecEnv(4096, seed=1, autoreset_mode="same_step")
        e.set_task([sample_cartpole(seed=k) for k in range(4)]); e.reset(); e.step(np.zeros(4096, np.int32)); e.close()
        e = AcrobotVecEnv(4096, seed=1, autoreset_mode="same_step")
        e.set_task([sample_acrobot(seed=k) for k in range(4)]); e.reset(); e.step(np.zeros(4096, np.int32)); e.close()

    import gc
    round_()
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(12):
        round_()
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, (free0 - free1) / 2**20      # a leak of one handle's tables per round would be >> 64 MiB
