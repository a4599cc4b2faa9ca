#!/usr/bin/env python3
"""BASELINE.json configs[4] end to end: the mixed task batch (anymdp + linds + metacontrol) sharded over the N GPUs of a node.

Per GPU 16,384 anymdp (2b: 256 synthetic tasks x 64) + 8,192 linds (128 tasks x 64, ns = 32) + 8,192 cartpole (1,024 tasks x 8)
— 262,144 envs at N = 8 (weak scaling: the per-GPU share is fixed).  A "step" is one vector step of a
rank's whole share = ONE launch (xv_mixed_step_many: the three families' step bodies in one grid).  Stepping needs no
collective; the exchange is the all-gather of each finished T = 32-step rollout chunk of all three families
(xenoverse_amd.distributed.MixedChunk, 29 MB per rank) on a side stream, overlapped with the next chunk's stepping.

Called by bench.py: `python bench.py --workload mixed --gpus N ...` prints this module's line as THE line; the default
(anymdp) workload at N > 1 carries it as `families.mixed`.  `--exchange-selftest` runs the control flow (layout, pack ->
all-gather -> unpack, MAX over ranks) on fabricated CPU records over gloo: a functional test, not a measurement.
"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PER_GPU = {"anymdp": 16384, "linds": 8192, "cartpole": 8192}
HBM_PEAK_GBS = 8000.0


def _median(xs):
    s = sorted(xs)
    n = len(s)
    return s[n // 2] if n % 2 else 0.5 * (s[n // 2 - 1] + s[n // 2])


def fabricate(torch, chunk, T, lo_hi):
    """records that are functions of (global env id, step) — the gathered batch can be checked on every rank"""
    tt = torch.arange(T, dtype=torch.int32)[:, None]
    out = {}
    for f, (lo, hi) in lo_hi.items():
        gid = torch.arange(lo, hi, dtype=torch.int32)[None, :]
        te = ((gid + tt) % 5 == 0).to(torch.uint8)
        tr = ((gid + 2 * tt) % 7 == 0).to(torch.uint8)
        rew = gid.float() * 0.25 + tt.float()
        if f == "anymdp":
            out[f] = dict(obs=(gid * 3 + tt) % 64, action=(gid + tt) % 8, reward=rew, terminated=te, truncated=tr)
        else:
            D = chunk.LINDS_DIM if f == "linds" else chunk.CART_DIM
            k = torch.arange(D, dtype=torch.float32)[None, None, :]
            out[f] = dict(obs=(gid.float()[:, :, None] + 0.5 * tt.float()[:, :, None] + 0.125 * k).contiguous(), reward=rew,
                          terminated=te, truncated=tr, action=((gid + tt) % 2).to(torch.int32))
    return out


def cpu_baseline_mixed(seconds, seed, cores):
    """the C oracle (oracle/xeno_oracle.c: the restatement of the three families' reference step) on ONE GPU's share of the
    batch, `cores` OpenMP threads for anymdp and linds (the cartpole oracle is scalar): a bounded sample of vector steps"""
    import numpy as np
    import oracle
    from xenoverse_amd.linds import build_tables, pad_tables
    from xenoverse_amd.metacontrol import sample_cartpole
    from xenoverse_amd.mixed_shard import linds_task
    na, nl, nc = PER_GPU["anymdp"], PER_GPU["linds"], PER_GPU["cartpole"]
    t0 = time.perf_counter()
    ora_a = oracle.AnyMDPOracle(oracle.anymdp_synth(seed=7, task_index_base=0, n_task=na // 64, S=64, A=8, s0_max=4),
                                np.repeat(np.arange(na // 64, dtype=np.int32), 64))
    ora_l = oracle.LinDSOracle(pad_tables(build_tables([linds_task(g) for g in range(nl // 64)])),
                               np.repeat(np.arange(nl // 64, dtype=np.int32), 64))
    ct = [sample_cartpole(seed=g) for g in range(nc // 8)]
    ora_c = oracle.CartPoleOracle(np.array([[t["gravity"], t["masscart"], t["masspole"], t["length"]] for t in ct], np.float64),
                                  np.repeat(np.arange(nc // 8, dtype=np.int32), 8), frameskip=1)
    ora_a.reset(seed, 0, 0); ora_l.reset(seed, 0, 0); ora_c.reset(seed, 0, 0)
    rng = np.random.RandomState(0)
    aa = rng.randint(0, 8, (16, na)).astype(np.int32)
    al = rng.uniform(-1, 1, (16, nl, 8)).astype(np.float32)
    ac = rng.randint(0, 2, (16, nc)).astype(np.int32)
    setup = time.perf_counter() - t0
    k, t0 = 0, time.perf_counter()
    while True:
        for _ in range(4):
            ora_a.step(seed, 0, 10 + k, aa[k % 16], 2, n_threads=cores)
            ora_l.step(seed, 0, 10 + k, al[k % 16], 2, n_threads=cores)
            ora_c.step(seed, 0, 10 + k, ac[k % 16], 2)
            k += 1
        if time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": (na + nl + nc) * k / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "oracle/xeno_oracle.c: one GPU's share (%d anymdp + %d linds + %d cartpole envs), %d vector steps in %.1f s "
                      "(OpenMP %d threads for anymdp / linds, cartpole scalar; set-up %.1f s)" % (na, nl, nc, k, dt, cores, setup)}


def run_mixed(args, torch, dist, dinfo, rank, world, local, wd, selftest=False, scale=1):
    """-> the JSON line (dict) on every rank (rank 0 prints it).  scale: divides the per-GPU share (tests).
    The share and the gather are closed on every way out (an exception included)."""
    held = {}
    try:
        return _run_mixed(args, torch, dist, dinfo, rank, world, local, wd, selftest, scale, held)
    finally:
        for k in ("gather", "share"):
            obj = held.pop(k, None)
            if obj is not None:
                try:
                    obj.close()
                except Exception:
                    pass


def _run_mixed(args, torch, dist, dinfo, rank, world, local, wd, selftest, scale, held):
    from xenoverse_amd.distributed import MixedChunk, RolloutGather
    T = int(args.period)
    overlap_note = None
    per = {f: max(64, n // scale) for f, n in PER_GPU.items()}
    tot = {f: n * world for f, n in per.items()}
    gpu = not selftest
    dev = torch.device("cuda", local) if gpu else torch.device("cpu")
    share = None
    if gpu:
        from xenoverse_amd.mixed_shard import MixedShare
        share = MixedShare(rank, world, tot["anymdp"], tot["linds"], tot["cartpole"], T=T, seed=args.seed, device=str(dev))
        held["share"] = share
        chunk = share.chunk
        if getattr(args, "overlap", "auto") != "off":
            try:
                share.set_overlap(True)       # calls of >= 64 steps: two streams, the launch of step k + 1 under step k
            except Exception as ex:           # e.g. another handle on this device holds the overlap switch: one stream, same results
                overlap_note = "overlap not taken: %r" % (ex,)
        share.random_actions(args.seed + 17 * rank)
        share.reset()
        share.step_many(T)
        torch.cuda.synchronize()
    else:
        chunk = MixedChunk(T, tot["anymdp"], tot["linds"], tot["cartpole"], world)
        fab = fabricate(torch, chunk, T, {f: chunk.share[f][rank] for f in per})

    gather, gather_note, transport = None, "none", None
    want_gather = not args.no_allgather and (world > 1 or gpu) and dinfo.get("note") is None
    if want_gather:
        try:
            if gpu and args.transport in ("auto", "rccl"):
                if wd is not None:
                    wd.arm("RCCL communicator set-up (mixed)")
                gather = RolloutGather((chunk.bytes_per_rank,), device=dev, transport="rccl", rank=rank, world=world)
                transport = "rccl"
                if wd is not None:
                    wd.cancel()
            elif world > 1:
                gather = RolloutGather((chunk.bytes_per_rank,), device=dev)
                transport = "torch"
        except Exception as ex:
            gather, gather_note = None, "all_gather unavailable: %r" % (ex,)
            if world > 1 and transport is None:
                try:
                    gather = RolloutGather((chunk.bytes_per_rank,), device=dev)
                    transport = "torch"
                except Exception as ex2:
                    gather_note += " / %r" % (ex2,)
        if gather is not None:
            held["gather"] = gather
            gather_note = "all_gather of %d-step chunks of the three families, %d B per rank (%s, side stream)" % (
                T, chunk.bytes_per_rank, "xv_rollout_allgather: ncclAllGather over librccl" if transport == "rccl"
                else "torch.distributed " + str(dinfo.get("backend")))

    def run(k_steps, with_gather):
        if not with_gather:
            if gpu and k_steps > 0:
                share.step_many(k_steps)
            return
        done = 0
        while done < k_steps:
            n = min(T, k_steps - done)
            if gpu:
                share.step_many(n)
            done += n
            if n == T:
                gather.wait()                    # the previous chunk has left before its buffer is repacked
                if gpu:
                    share.pack(gather.local)
                else:
                    chunk.pack(rank, fab, gather.local)
                gather.launch()
        gather.wait()

    def barrier():
        if gpu:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if gpu:
            torch.cuda.synchronize()

    def timed_pass(with_gather, repeats):
        gc.collect()
        gc.disable()
        try:
            run(args.warmup, with_gather)
            walls, evs = [], []
            for _ in range(repeats):
                barrier()
                t0 = time.perf_counter()
                if gpu:
                    share.ea.engine.event_record(0)
                run(args.steps, with_gather)
                if gpu:
                    share.ea.engine.event_record(1)
                barrier()
                walls.append(time.perf_counter() - t0)
                evs.append(share.ea.engine.event_elapsed_ms() if gpu else walls[-1] * 1e3)
        finally:
            gc.enable()
        if os.environ.get("XV_BENCH_DEBUG_WALLS"):
            sys.stderr.write("walls us/step %s\n" % ([round(w / args.steps * 1e6, 2) for w in walls],))
        tt = torch.tensor([walls, evs], dtype=torch.float64)
        if dist is not None:
            if dist.get_backend() == "nccl":
                tt = tt.to(dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tt = tt.cpu()
        return _median(tt[0].tolist()), _median(tt[1].tolist())

    R = max(1, args.repeats)
    wall, ev_ms = timed_pass(False, R)
    errs = share.check_errors() if gpu else 0
    overlapped = bool(gpu and share.overlap_state == 1)
    wall_g, check = None, None
    if gather is not None:
        if wd is not None:
            wd.arm("the all-gather pass (mixed)")
        try:
            wall_g, _ = timed_pass(True, max(1, min(R, 5)))
            # every rank sees every shard: rank order == env order of each family
            run(T, True)
            got = chunk.unpack(gather.wait())
            if gpu:
                torch.cuda.synchronize()
                mine = share.rings_for_pack()
                ok = True
                for f, keys in (("anymdp", ("obs", "action", "reward", "terminated", "truncated")),
                                ("linds", ("obs", "reward", "terminated", "truncated")),
                                ("cartpole", ("obs", "reward", "terminated", "truncated", "action"))):
                    lo, hi = chunk.share[f][rank]
                    for j, k in enumerate(keys):
                        ok = ok and bool(torch.equal(got[f][j][:, lo:hi], mine[f][k]))
                check = "ok" if ok else "MISMATCH"
            else:
                ref = fabricate(torch, chunk, T, {f: (0, tot[f]) for f in per})
                ok = all(torch.equal(got["anymdp"][j], ref["anymdp"][k]) for j, k in
                         enumerate(("obs", "action", "reward", "terminated", "truncated")))
                ok = ok and all(torch.equal(got["linds"][j], ref["linds"][k]) for j, k in
                                enumerate(("obs", "reward", "terminated", "truncated")))
                ok = ok and all(torch.equal(got["cartpole"][j], ref["cartpole"][k]) for j, k in
                                enumerate(("obs", "reward", "terminated", "truncated", "action")))
                check = "ok" if ok else "MISMATCH"
            if dist is not None:
                flag = torch.tensor([1.0 if check == "ok" else 0.0])
                if dist.get_backend() == "nccl":
                    flag = flag.to(dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                check = "ok" if float(flag.item()) == 1.0 else "MISMATCH"
        except Exception as ex:
            gather_note += "; failed: %r" % (ex,)
        if wd is not None:
            wd.cancel()

    n_rank = sum(per.values())
    total = world * n_rank * args.steps
    us = ev_ms * 1e3 / args.steps
    algo = share.algorithmic_bytes_per_vector_step() if gpu else 562 * per["anymdp"] + 432 * per["linds"] + 74 * per["cartpole"]
    out = {
        "metric": "env-steps/sec (whole node), mixed task batch (anymdp + linds + metacontrol), %d envs per GPU" % n_rank,
        "value": None if selftest else total / wall, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "repeats": R, "ms_per_step": None if selftest else wall * 1e3 / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64/f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4]: mixed task batch, %d envs over %d GPU(s) = per GPU %d anymdp (S=64 A=8, %d shared "
                               "synthetic tasks x 64) + %d linds (ns=32 na=8 no=8, %d tasks x 64) + %d cartpole (%d tasks x 8); "
                               "SAME_STEP auto-reset, random actions"
                               % (world * n_rank, world, per["anymdp"], per["anymdp"] // 64, per["linds"], per["linds"] // 64,
                                  per["cartpole"], per["cartpole"] // 8),
                   "envs_per_gpu": per, "envs_total": {f: n * world for f, n in per.items()},
                   "launch": ("one fused kernel per vector step of a rank's share (xv_mixed_step_many)" +
                              ("; consecutive steps alternate between two HIP streams and overlap, every wave takes its envs over "
                               "from the same wave of the step before (xv_anymdp_set_step_many_overlap)" if overlapped else ""))
                             if (share is None or share.fused)
                             else "three launches per vector step (no fused instantiation for these handles)",
                   "overlap": overlapped, "overlap_requested": getattr(args, "overlap", "auto"), "overlap_note": overlap_note,
                   "sharding": "contiguous ranges per family (shard_range), env_id_base = the range's start, tasks named by global "
                               "index: no data-path collective",
                   "exchange": gather_note, "chunk_bytes_per_rank": chunk.bytes_per_rank, "chunk_steps": T,
                   "device_error_flags": errs},
        "roofline": None if selftest else {
            "bound": "hbm", "achieved": algo / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": algo / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "frac_wall": algo / (wall / args.steps) / 1e9 / HBM_PEAK_GBS,
            "traffic": None, "kernel": "mixed_step_kernel (anymdp + linds + cartpole step bodies in one grid)",
            "avg_launch_us": us, "algorithmic_bytes_per_launch": algo,
            "avg_launch_us_note": "overlapped launches: time per launch in steady state = timed region / launches" if overlapped else None,
            "note": "a 13 MB vector step: launch-latency bound (an empty launch is 2.7 us)"},
        "rccl": dinfo.get("rccl"), "rccl_ranks": (gather.comm.count() if transport == "rccl" else dinfo.get("rccl_ranks")),
        "transport": transport,
    }
    if wall_g is not None:
        chunks = args.steps // T
        out["with_allgather"] = {"value": None if selftest else total / wall_g, "unit": "env-steps/s",
                                 "gathered_GB_per_s_per_rank": chunks * chunk.bytes_per_rank * max(world - 1, 1) / wall_g / 1e9,
                                 "ranks": world, "gathered_slice_equals_local_rings": check,
                                 "note": "one-rank communicator: the pack kernels and the collective's launch cost, no link traffic"
                                         if world == 1 else None}
    if selftest:
        out["mode"] = "exchange-selftest: fabricated CPU records, no stepping — NOT a measurement"
        out["selftest"] = check
    return out
