"""BASELINE configs[4] across ranks (scaled down, one GPU): every rank's share of the mixed batch, stepped on its own and
exchanged through the chunk format, reproduces its slice of the UNSHARDED batch bit for bit — the property that lets the
8-GPU run be correct by construction (env_id_base = the share's start, tasks named by global index)."""
import numpy as np
import pytest
import torch

from xenoverse_amd.distributed import MixedChunk, RolloutGather, pack_records_f32, unpack_records_f32
from xenoverse_amd.mixed_shard import MixedShare

pytestmark = pytest.mark.gpu

A_KEYS = ("obs", "action", "reward", "terminated", "truncated")
L_KEYS = ("obs", "reward", "terminated", "truncated")
C_KEYS = ("obs", "reward", "terminated", "truncated", "action")


def _np(t):
    return t.detach().cpu().numpy()


def _run_share(rank, world, tot, T, n_steps, acts, seed=11):
    sh = MixedShare(rank, world, *tot, T=T, seed=seed, linds_ns=16)
    lo = sh.lo
    sh.set_actions(acts["a"][:, lo["anymdp"]:lo["anymdp"] + sh.n["anymdp"]],
                   acts["l"][:, lo["linds"]:lo["linds"] + sh.n["linds"]],
                   acts["c"][:, lo["cartpole"]:lo["cartpole"] + sh.n["cartpole"]])
    sh.reset()
    sh.step_many(n_steps)
    torch.cuda.synchronize()
    assert sh.check_errors() == 0
    return sh


@pytest.mark.parametrize("world", [2, 4])
def test_rank_shares_exchange_to_the_unsharded_batch(world):
    tot, T, n_steps = (2048, 1024, 1024), 8, 19            # 19 steps: the ring holds steps 11 .. 18 (slots of k % 8)
    rng = np.random.RandomState(3)
    acts = dict(a=rng.randint(0, 8, (T, tot[0])).astype(np.int32), l=rng.uniform(-1.2, 1.2, (T, tot[1], 8)).astype(np.float32),
                c=rng.randint(0, 2, (T, tot[2])).astype(np.int32))
    whole = _run_share(0, 1, tot, T, n_steps, acts)
    assert whole.fused
    ref = {f: {k: v.clone() for k, v in d.items()} for f, d in whole.rings_for_pack().items()}
    whole.close()
    chunk = MixedChunk(T, *tot, world)
    gathered = torch.zeros((world, chunk.bytes_per_rank), dtype=torch.uint8, device="cuda")
    for r in range(world):
        sh = _run_share(r, world, tot, T, n_steps, acts)
        assert sh.chunk.bytes_per_rank == chunk.bytes_per_rank
        # the rank's own rings are its slice of the unsharded ones
        mine = sh.rings_for_pack()
        for f in ("anymdp", "linds", "cartpole"):
            lo, hi = chunk.share[f][r]
            for k, v in mine[f].items():
                assert torch.equal(v, ref[f][k][:, lo:hi]), (f, k, r)
        # pack (HIP kernels) and send through the C-ABI's collective on a one-rank communicator (what one GPU can host)
        g = RolloutGather((chunk.bytes_per_rank,), device="cuda", transport="rccl", rank=0, world=1)
        sh.pack(g.local)
        g.launch()
        gathered[r].copy_(g.wait()[0])
        torch.cuda.synchronize()
        g.close()
        sh.close()
    got = chunk.unpack(gathered)
    for f, keys in (("anymdp", A_KEYS), ("linds", L_KEYS), ("cartpole", C_KEYS)):
        for j, k in enumerate(keys):
            assert torch.equal(got[f][j], ref[f][k]), (f, k)
    assert int(ref["anymdp"]["terminated"].sum()) > 100 and int(ref["cartpole"]["terminated"].sum()) > 50


def test_device_float_records_match_the_host_format():
    """xv_pack_rollout_f32 / xv_unpack_rollout_f32: the same bits as the torch (CPU) packer, both record widths"""
    g = torch.Generator().manual_seed(2)
    for D, with_action in ((16, False), (4, True)):
        T, N = 7, 1001
        obs = torch.randn((T, N, D), generator=g)
        rew = torch.randn((T, N), generator=g)
        te = (torch.rand((T, N), generator=g) < 0.3).to(torch.uint8)
        tr = (torch.rand((T, N), generator=g) < 0.3).to(torch.uint8)
        act = torch.randint(0, 1 << 24, (T, N), generator=g, dtype=torch.int32) if with_action else None
        host = pack_records_f32(obs, rew, te, tr, act)
        dev = pack_records_f32(obs.cuda(), rew.cuda(), te.cuda(), tr.cuda(), None if act is None else act.cuda())
        assert torch.equal(dev.cpu(), host)
        out = unpack_records_f32(dev, D, with_action=with_action)
        assert torch.equal(out[0].cpu(), obs) and torch.equal(out[1].cpu(), rew)
        assert torch.equal(out[2].cpu(), te) and torch.equal(out[3].cpu(), tr)
        if with_action:
            assert torch.equal(out[4].cpu(), act)


def test_shares_must_be_whole_tasks():
    with pytest.raises(ValueError):
        MixedShare(0, 3, 2048, 1024, 1024, T=4)          # 2048 / 3 is not a multiple of 64


@pytest.mark.parametrize("mode,T,n_steps", [("same_step", 8, 203), ("next_step", 16, 128), ("disabled", 2, 64)])
def test_overlapped_step_many_equals_the_ordinary_one(mode, T, n_steps):
    """Two HIP streams, the launch of step k + 1 under step k, hand-off per wave (AnyMDP: tag in the env record; LinDS and
    CartPole: a word per wave): rings, states, step counters and later steps equal the one-stream loop's bit for bit"""
    tot = (4096, 2048, 1024 + 8)                              # a partial last CartPole wave
    rng = np.random.RandomState(5)
    acts = dict(a=rng.randint(0, 8, (T, tot[0])).astype(np.int32), l=rng.uniform(-1.2, 1.2, (T, tot[1], 8)).astype(np.float32),
                c=rng.randint(0, 2, (T, tot[2])).astype(np.int32))
    res = []
    for overlap in (False, True):
        sh = MixedShare(0, 1, *tot, T=T, seed=21, linds_ns=16, autoreset_mode=mode)
        sh.set_actions(acts["a"], acts["l"], acts["c"])
        if overlap:
            sh.set_overlap(True)
        sh.reset()
        rec = []
        for n in (n_steps, 3, n_steps):                      # a long call, a short one (ordinary loop), a long one again
            sh.step_many(n)
            torch.cuda.synchronize()
            if overlap:
                assert sh.overlap_state == (1 if n >= 64 else 0), (n, sh.overlap_state)
            rec.append({k: v.clone() for k, v in sh.ring.items()})
            st = {}
            for f, e in (("a", sh.ea), ("l", sh.el), ("c", sh.ec)):
                for name, v in zip(("state", "steps", "need_reset"), e.get_state()):
                    st[f + "_" + name] = torch.as_tensor(v).clone()
            rec.append(st)
        flags = sh.check_errors()                            # "disabled": stepping terminated envs is flagged, on both paths
        assert flags == (0 if mode != "disabled" else flags & 2)
        rec.append(dict(flags=torch.tensor(flags)))
        if overlap:
            sh.set_overlap(False)
        sh.close()
        res.append(rec)
    for i, (p, q) in enumerate(zip(*res)):
        for k in p:
            assert torch.equal(torch.as_tensor(p[k]), torch.as_tensor(q[k])), (i, k)
    last = res[0][-3]
    if mode != "disabled":
        assert int(last["at"].sum()) > 0 and int(last["ct"].sum()) > 0


def test_overlapped_mixed_steps_with_a_held_up_host(monkeypatch):
    """as tests/test_gpu_chains.py: 3.5 s between the two launches of a cycle; the cycle gate keeps the bounded hand-off waits
    from expiring"""
    monkeypatch.setenv("XV_PIPE_TEST_STALL_MS", "3500")
    sh = MixedShare(0, 1, 1024, 512, 512, T=8, seed=4, linds_ns=16)
    sh.random_actions(3)
    sh.set_overlap(True)
    sh.reset()
    sh.step_many(80)
    torch.cuda.synchronize()
    assert sh.overlap_state == 1 and sh.check_errors() == 0
    sh.set_overlap(False)
    sh.close()


@pytest.mark.parametrize("mode,T,n_steps", [("same_step", 8, 203), ("next_step", 16, 128)])
def test_an_expired_hand_off_of_the_mixed_step_is_repaired_by_the_replay(monkeypatch, mode, T, n_steps):
    """XV_PIPE_TEST_FAIL=1 leaves behind what an expired hand-off does (the flag plus a spurious error bit, wrong AnyMDP records,
    LinDS tiles and CartPole states, wrong ring contents) between the join of an overlapped xv_mixed_step_many and its replay
    kernel: the three families are restored from the call's snapshot and the call is re-run in one launch — rings, states,
    counters and flags equal the one-stream loop's; the state word says -2"""
    tot = (4096, 2048, 1024 + 8)
    rng = np.random.RandomState(5)
    acts = dict(a=rng.randint(0, 8, (T, tot[0])).astype(np.int32), l=rng.uniform(-1.2, 1.2, (T, tot[1], 8)).astype(np.float32),
                c=rng.randint(0, 2, (T, tot[2])).astype(np.int32))
    res = []
    for overlap in (False, True):
        if overlap:
            monkeypatch.setenv("XV_PIPE_TEST_FAIL", "1")
            monkeypatch.setenv("XV_PIPE_NO_BACKOFF", "1")      # every long call is overlapped, fails and is replayed
        sh = MixedShare(0, 1, *tot, T=T, seed=21, linds_ns=16, autoreset_mode=mode)
        sh.set_actions(acts["a"], acts["l"], acts["c"])
        if overlap:
            sh.set_overlap(True)
        sh.reset()
        rec = []
        for n in (n_steps, 3, n_steps):
            sh.step_many(n)
            torch.cuda.synchronize()
            if overlap:
                assert sh.overlap_state == (-2 if n >= 64 else 0), (n, sh.overlap_state)
            rec.append({k: v.clone() for k, v in sh.ring.items()})
            st = {}
            for f, e in (("a", sh.ea), ("l", sh.el), ("c", sh.ec)):
                for name, v in zip(("state", "steps", "need_reset"), e.get_state()):
                    st[f + "_" + name] = torch.as_tensor(v).clone()
            rec.append(st)
        assert sh.check_errors() == 0
        if overlap:
            sh.set_overlap(False)
        sh.close()
        res.append(rec)
    for i, (p, q) in enumerate(zip(*res)):
        for k in p:
            assert torch.equal(torch.as_tensor(p[k]), torch.as_tensor(q[k])), (i, k)


def test_run_mixed_beside_a_handle_that_holds_the_overlap_switch():
    """bench.py at N > 1 runs the mixed workload (`families.mixed`) while its AnyMDP env exists (round-5 advisor finding: the
    env held the device's one overlap slot and the block was always an error).  run_mixed must cope with a taken slot — one
    stream, said so in the line — and overlap once the slot is free; bench.py releases it around the call."""
    import argparse
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import bench_mixed
    import oracle
    from xenoverse_amd.anymdp import AnyMDPVecEnv, to_blocked
    tab = oracle.anymdp_synth(seed=12, task_index_base=0, n_task=4, S=64, A=8, s0_max=4)
    dev = dict(S=64, A=8, s0_max=4)
    tab["rows"] = to_blocked(tab["cdf"], tab["rs"])
    for k in ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps"):
        v = np.ascontiguousarray(tab[k])
        dev[k] = torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).cuda()
    env = AnyMDPVecEnv(256, seed=1)
    env.set_task(dev)
    env.set_step_many_overlap(True)                      # holds the device's slot, as bench.py's headline env does
    args = argparse.Namespace(period=8, seed=3, overlap="auto", no_allgather=True, transport="torch", warmup=16, steps=64, repeats=2)
    dinfo = {"note": None, "rccl": None, "rccl_ranks": None, "backend": None}
    out = bench_mixed.run_mixed(args, torch, None, dinfo, 0, 1, 0, None, scale=16)
    assert out["value"] > 0 and out["config"]["overlap"] is False and "overlap not taken" in out["config"]["overlap_note"]
    assert out["config"]["device_error_flags"] == 0
    env.set_step_many_overlap(False)                     # what bench.py does before families.mixed
    out = bench_mixed.run_mixed(args, torch, None, dinfo, 0, 1, 0, None, scale=16)
    assert out["config"]["overlap"] is True and out["config"]["overlap_note"] is None and out["config"]["device_error_flags"] == 0
    env.set_step_many_overlap(True)                      # the share released the slot when it was closed
    env.close()
