"""N > 1 path on CPU: world_size-2 `gloo` processes exercise the sharding helpers and the rollout all-gather
(the same code runs over RCCL/xGMI with backend "nccl" on the GPUs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from xenoverse_amd.distributed import (REC_BYTES, MixedChunk, RolloutGather, pack_records, pack_records_f32, shard_env_task,
                                       shard_range, unpack_records, unpack_records_f32)


def test_shard_range_partitions_exactly():
    for n in (1, 7, 8, 65536, 65537, 262144):
        for world in (1, 2, 3, 8):
            ends = [shard_range(n, r, world) for r in range(world)]
            assert ends[0][0] == 0 and ends[-1][1] == n
            assert all(ends[r][1] == ends[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in ends]
            assert max(sizes) - min(sizes) <= 1


def test_shard_env_task_renumbers_tasks():
    env_task = np.repeat(np.arange(10), 7)
    for world in (1, 2, 4):
        seen = []
        for r in range(world):
            lo, hi, local, ids = shard_env_task(env_task, r, world)
            assert np.array_equal(ids[local], env_task[lo:hi])
            assert local.min() == 0 and local.max() == len(ids) - 1
            seen.append(np.arange(lo, hi))
        assert np.array_equal(np.concatenate(seen), np.arange(len(env_task)))


def test_record_packing_roundtrip():
    T, N = 5, 33
    g = torch.Generator().manual_seed(0)
    obs = torch.randint(0, 64, (T, N), generator=g, dtype=torch.int32)
    act = torch.randint(0, 8, (T, N), generator=g, dtype=torch.int32)
    rew = torch.randn((T, N), generator=g)
    te = (torch.rand((T, N), generator=g) < 0.1).to(torch.uint8)
    tr = (torch.rand((T, N), generator=g) < 0.1).to(torch.uint8)
    rec = pack_records(obs, act, rew, te, tr)
    assert rec.shape == (T, N, REC_BYTES) and rec.dtype == torch.uint8
    o2, a2, r2, te2, tr2 = unpack_records(rec)
    assert torch.equal(o2, obs) and torch.equal(a2, act) and torch.equal(r2, rew)
    assert torch.equal(te2, te) and torch.equal(tr2, tr)


def test_float_record_packing_roundtrip():
    """LinDS / CartPole records: D fp32 observation words + reward + flag word (terminated, truncated, action)"""
    g = torch.Generator().manual_seed(3)
    for D, with_action in ((16, False), (4, True)):
        T, N = 5, 37
        obs = torch.randn((T, N, D), generator=g)
        obs[0, 0, 0] = float("inf"); obs[1, 2, 1] = -0.0
        rew = torch.randn((T, N), generator=g)
        te = (torch.rand((T, N), generator=g) < 0.3).to(torch.uint8)
        tr = (torch.rand((T, N), generator=g) < 0.3).to(torch.uint8)
        act = torch.randint(0, 1 << 24, (T, N), generator=g, dtype=torch.int32) if with_action else None
        rec = pack_records_f32(obs, rew, te, tr, act)
        assert rec.shape == (T, N, 4 * (D + 2)) and rec.dtype == torch.uint8
        out = unpack_records_f32(rec, D, with_action=with_action)
        assert torch.equal(out[0].view(torch.int32), obs.view(torch.int32)) and torch.equal(out[1], rew)
        assert torch.equal(out[2], te) and torch.equal(out[3], tr)
        if with_action:
            assert torch.equal(out[4], act)


@pytest.mark.parametrize("world,tot", [(8, (131072, 65536, 65536)), (2, (1024, 512, 520)), (3, (200, 100, 50))])
def test_mixed_chunk_layout_and_pack_unpack(world, tot):
    """config 5's chunk: three record blocks per rank, padded to the largest share; packing every rank's share and
    unpacking the concatenation gives back the unsharded families in global env order"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import bench_mixed
    T = 3
    if tot[0] > 4096:       # the real sizes: layout arithmetic only
        ch = MixedChunk(32, *tot, world)
        assert ch.bytes_per_rank == 32 * (16384 * 8 + 8192 * 72 + 8192 * 24) == 29360128
        assert all(ch.n_local("anymdp", r) == 16384 and ch.share["linds"][r] == (8192 * r, 8192 * (r + 1)) for r in range(world))
        return
    ch = MixedChunk(T, *tot, world)
    assert ch.bytes_per_rank % 16 == 0
    gathered = torch.zeros((world, ch.bytes_per_rank), dtype=torch.uint8)
    for r in range(world):
        fab = bench_mixed.fabricate(torch, ch, T, {f: ch.share[f][r] for f in ch.rec})
        ch.pack(r, fab, gathered[r])
    got = ch.unpack(gathered)
    ref = bench_mixed.fabricate(torch, ch, T, {"anymdp": (0, tot[0]), "linds": (0, tot[1]), "cartpole": (0, tot[2])})
    for j, k in enumerate(("obs", "action", "reward", "terminated", "truncated")):
        assert torch.equal(got["anymdp"][j], ref["anymdp"][k]), k
    for j, k in enumerate(("obs", "reward", "terminated", "truncated")):
        assert torch.equal(got["linds"][j], ref["linds"][k]), k
    for j, k in enumerate(("obs", "reward", "terminated", "truncated", "action")):
        assert torch.equal(got["cartpole"][j], ref["cartpole"][k]), k


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, T, q, uneven=None):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n_total, rank, world)
        n = hi - lo
        # each rank fabricates the records of ITS envs from the global env id, so the gathered batch is checkable
        gid = torch.arange(lo, hi, dtype=torch.int32)
        obs = (gid[None, :] * 3 + torch.arange(T, dtype=torch.int32)[:, None]) % 64
        act = (gid[None, :] + torch.arange(T, dtype=torch.int32)[:, None]) % 8
        rew = gid[None, :].float() * 0.5 + torch.arange(T)[:, None].float()
        te = ((gid[None, :] + torch.arange(T, dtype=torch.int32)[:, None]) % 5 == 0).to(torch.uint8)
        tr = torch.zeros((T, n), dtype=torch.uint8)
        g = RolloutGather((T, n, REC_BYTES), device="cpu")
        pack_records(obs, act, rew, te, tr, out=g.local)
        g.launch()
        out = g.wait()
        # every rank sees every shard; rank order == env order
        full = torch.cat([out[r] for r in range(world)], dim=1)
        o2, a2, r2, te2, _ = unpack_records(full)
        gid_all = torch.arange(0, n_total, dtype=torch.int32)
        ok = bool(torch.equal(o2, (gid_all[None, :] * 3 + torch.arange(T, dtype=torch.int32)[:, None]) % 64)
                  and torch.equal(a2, (gid_all[None, :] + torch.arange(T, dtype=torch.int32)[:, None]) % 8)
                  and torch.equal(r2, gid_all[None, :].float() * 0.5 + torch.arange(T)[:, None].float())
                  and torch.equal(te2, ((gid_all[None, :] + torch.arange(T, dtype=torch.int32)[:, None]) % 5 == 0).to(torch.uint8)))
        # the store that carries a RCCL unique id: the process group's own (no second port is opened)
        from xenoverse_amd.distributed import _rendezvous_store
        st = _rendezvous_store(rank, world, None, None, None, 30)
        if rank == 0:
            st.set("id_test", b"\x07" * 128)
        ok = ok and bytes(st.get("id_test")) == b"\x07" * 128
        if uneven is not None:
            # uneven shares travel as a MixedChunk (every rank's blocks padded to the largest share: an all-gather moves equal
            # byte counts); every rank unpacks the whole batch and checks it against the fabricated global one
            import sys
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
            import bench_mixed
            ch = MixedChunk(T, *uneven, world)
            g2 = RolloutGather((ch.bytes_per_rank,), device="cpu")
            ch.pack(rank, bench_mixed.fabricate(torch, ch, T, {f: ch.share[f][rank] for f in ch.rec}), g2.local)
            g2.launch()
            got = ch.unpack(g2.wait())
            ref = bench_mixed.fabricate(torch, ch, T, {"anymdp": (0, uneven[0]), "linds": (0, uneven[1]), "cartpole": (0, uneven[2])})
            ok = ok and all(torch.equal(got["anymdp"][j], ref["anymdp"][k]) for j, k in
                            enumerate(("obs", "action", "reward", "terminated", "truncated")))
            ok = ok and all(torch.equal(got["linds"][j], ref["linds"][k]) for j, k in enumerate(("obs", "reward", "terminated", "truncated")))
            ok = ok and all(torch.equal(got["cartpole"][j], ref["cartpole"][k]) for j, k in
                            enumerate(("obs", "reward", "terminated", "truncated", "action")))
            ok = ok and len({ch.n_local("anymdp", r) for r in range(world)}) > 1      # the shares really differ
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, ok, float(t[0])))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_world_size_2_gloo_allgather_of_rollout_chunks():
    world, n_total, T = 2, 64, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in res) == [0, 1]
    assert all(ok for _, ok, _ in res)
    assert all(tmax == 2.0 for _, _, tmax in res)


def _run_bench_selftest(extra_env, extra_args, timeout=240, world=2):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="1", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "64",
           "--warmup", "32", "--repeats", "3", "--envs", "512", "--exchange-selftest"] + extra_args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    return r, [json.loads(ln) for ln in lines]


@pytest.mark.timeout(300)
def test_bench_py_two_ranks_end_to_end_on_gloo():
    """bench.py --gpus 2 through torch.distributed.run: process group, barriers, MAX over ranks, pack -> all-gather ->
    unpack of every rollout chunk and the JSON line — on fabricated CPU records (no GPU here), checked on every rank"""
    r, out = _run_bench_selftest({}, ["--transport", "rccl"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(out) == 1                      # rank 0 only
    d = out[0]
    assert d["n_gpus"] == 2 and d["selftest"] == "ok" and d["ranks_seen"] == 2 and d["repeats"] == 3
    # --transport rccl is the C-ABI's collective over device buffers: on CPU records the torch transport serves, and says so
    assert d["transport_requested"] == "rccl" and d["transport"] == "torch" and "torch used on CPU" in d["transport_note"]
    assert "families" not in d
    assert d["allgather_timeout"] is False and d["rccl"] is False and d["with_allgather"]["value"] > 0
    assert d["value"] is None and "NOT a measurement" in d["mode"] and d["cpu_baseline"] is None


@pytest.mark.timeout(300)
def test_bench_py_mixed_workload_two_ranks_on_gloo():
    """bench.py --workload mixed --gpus 2 (BASELINE configs[4]): sharding of the three families, chunk layout, pack ->
    all-gather -> unpack of all three record formats, checked against the fabricated global batch on every rank"""
    r, out = _run_bench_selftest({}, ["--workload", "mixed"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(out) == 1
    d = out[0]
    assert d["n_gpus"] == 2 and d["selftest"] == "ok" and d["ranks_seen"] == 2
    assert d["config"]["envs_total"] == {"anymdp": 32768, "linds": 16384, "cartpole": 16384}
    assert d["config"]["chunk_bytes_per_rank"] == 29360128 and d["transport"] == "torch"
    assert d["with_allgather"]["gathered_slice_equals_local_rings"] == "ok" and d["value"] is None


@pytest.mark.timeout(300)
def test_bench_py_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` the way the driver runs N = 1 (no torch.distributed.run in front): the parent starts the
    ranks as one fresh child process and leaves with its code; two ranks are seen, one JSON line comes out"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "64", "--warmup", "32", "--repeats", "3",
           "--envs", "512", "--exchange-selftest"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(out) == 1
    assert out[0]["n_gpus"] == 2 and out[0]["ranks_seen"] == 2 and out[0]["selftest"] == "ok"
    assert "without a launcher" in r.stderr
    # the child's failure is the parent's: a stalled rank makes the whole command leave non-zero
    r = subprocess.run(cmd + ["--gather-timeout", "6", "--transport", "torch"], capture_output=True, text=True, timeout=240,
                       env=dict(env, XV_BENCH_TEST_STALL="1"), cwd=root)
    assert r.returncode != 0


@pytest.mark.timeout(300)
def test_bench_py_watchdog_exits_nonzero_and_keeps_the_pass1_line():
    """a rank that never joins the all-gather pass: every rank leaves with a non-zero code, rank 0 still prints the
    pass-1 line, flagged machine-readably"""
    r, out = _run_bench_selftest({"XV_BENCH_TEST_STALL": "1"}, ["--gather-timeout", "6", "--transport", "torch"])
    assert r.returncode != 0
    assert len(out) == 1 and out[0]["allgather_timeout"] is True
    assert out[0]["transport"] == "torch" and out[0]["transport_requested"] == "torch" and out[0]["transport_note"] is None
    assert "did not finish within 6 s" in out[0]["config"]["exchange"]


# ---- eight ranks before there is an eight-GPU node (round-5 review, item 3) -------------------------------------------------
@pytest.mark.timeout(200)
def test_world_size_8_gloo_allgather_uneven_shares_and_the_id_store():
    """eight gloo processes: every rank checks every shard of the gathered AnyMDP chunk; UNEVEN shares (1,003 / 501 / 250 envs of
    the three families over eight ranks) travel as a MixedChunk and every rank rebuilds the whole batch; the store that would
    carry the RCCL unique id hands rank 0's 128 bytes to all eight; MAX over ranks"""
    world, n_total, T = 8, 1024, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, T, q, (1003, 501, 250))) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in res) == list(range(8))
    assert all(ok for _, ok, _ in res)
    assert all(tmax == 8.0 for _, _, tmax in res)


@pytest.mark.timeout(400)
@pytest.mark.parametrize("workload", ["anymdp", "mixed"])
def test_bench_py_eight_ranks_end_to_end_on_gloo(workload):
    """bench.py --gpus 8 through torch.distributed.run for both workloads: eight ranks rendezvous, every rank packs its share,
    all-gathers, unpacks and checks EVERY shard against the fabricated global batch; one JSON line"""
    r, out = _run_bench_selftest({}, ["--workload", workload], timeout=380, world=8)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(out) == 1
    d = out[0]
    assert d["n_gpus"] == 8 and d["selftest"] == "ok" and d["ranks_seen"] == 8
    if workload == "mixed":
        assert d["config"]["envs_total"] == {"anymdp": 131072, "linds": 65536, "cartpole": 65536}      # BASELINE configs[4]: 262,144 envs
        assert d["with_allgather"]["gathered_slice_equals_local_rings"] == "ok" and d["with_allgather"]["ranks"] == 8
    else:
        assert d["with_allgather"]["value"] > 0 and d["allgather_timeout"] is False


@pytest.mark.timeout(400)
def test_bench_py_watchdog_with_rank_5_of_8_stalling():
    """rank 5 of eight never joins the all-gather pass: every rank leaves non-zero within the deadline, rank 0 keeps the pass-1
    line, flagged"""
    r, out = _run_bench_selftest({"XV_BENCH_TEST_STALL": "1", "XV_BENCH_TEST_STALL_RANK": "5"},
                                 ["--gather-timeout", "8", "--transport", "torch"], timeout=380, world=8)
    assert r.returncode != 0
    assert len(out) == 1 and out[0]["allgather_timeout"] is True and out[0]["n_gpus"] == 8
    assert "did not finish within 8 s" in out[0]["config"]["exchange"]
