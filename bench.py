#!/usr/bin/env python3
"""bench.py — env-steps/sec of the AnyMDP hot path (BASELINE.json metric) on N MI355X GPUs of one node.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload = BASELINE.json configs[1]: anymdp |S|=64, |A|=8, 65,536 envs per GPU, synthetic tasks generated on
the device.  Default task sharing is "distinct" (2a: one task per env, 44 GiB of
tables per GPU, every table read misses every cache); --tasks 1024 gives "shared" (2b).
A "step" is one vector step of all envs of a rank = one launch of the step kernel; K steps are K back-to-back
launches (xv_anymdp_step_many), auto-reset SAME_STEP, actions pre-generated on the device.

One JSON line on rank 0: metric/value/unit/... as the driver's contract says, plus
  roofline      HBM bound for the step kernel (algorithmic 562 B per env-step)
  cpu_baseline  the C oracle (a port of the reference's step()) on the host cores, bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = {8: 8 * 64 + 50}   # w*S + 50, w = 8 (fp64 CDF), S = 64  -> 562 B
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--tasks", type=int, default=0, help="tasks per GPU (0 = one per env, config 2a)")
    ap.add_argument("--period", type=int, default=32, help="rollout-chunk ring length T")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--gather-timeout", type=int, default=120, help="N>1: seconds allowed for the all-gather pass")
    ap.add_argument("--no-allgather", action="store_true",
                    help="N>1: skip the RCCL all-gather of rollout chunks (pure replicas)")
    ap.add_argument("--search", default="auto", choices=["auto", "binary", "fence"])
    ap.add_argument("--fused", action="store_true", help="also time the fused T-step rollout kernel")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="xv_anymdp_step_many: replay ring cycles from a hipGraph (auto: only for small batches)")
    return ap.parse_args()


def make_tables(eng, torch, _lib, n_task, task_base, seed, S=64, A=8, s0_max=4):
    d = eng.device
    words = (S + 63) // 64
    from xenoverse_amd.anymdp import row_lines
    t = dict(S=S, A=A, s0_max=s0_max,
             rows=torch.empty((n_task, S, A, row_lines(S), 16), dtype=torch.float64, device=d),
             state_map=torch.empty((n_task, S), dtype=torch.int32, device=d),
             term_mask=torch.empty((n_task, words), dtype=torch.int64, device=d),
             s0_cdf=torch.empty((n_task, s0_max), dtype=torch.float64, device=d),
             s0_ids=torch.empty((n_task, s0_max), dtype=torch.int32, device=d),
             max_steps=torch.empty(n_task, dtype=torch.int32, device=d))
    _lib.check(eng.lib.xv_anymdp_synth_tasks(
        eng.handle, seed, task_base, n_task, S, A, s0_max,
        *[_lib.ptr(t[k]) for k in ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")]))
    eng.sync()
    return t


def cpu_baseline(seconds, seed):
    """The CPU oracle (a C port of the reference's step(), pinned to the reference's golden vectors) on the
    host cores: same workload shape (S=64, A=8, one synthetic task per env, random actions, SAME_STEP
    auto-reset), on a bounded sample of envs, all host threads."""
    import numpy as np
    import oracle
    n_task, per = 2048, 32        # 1 GiB of flat tables; 32 envs per task so that a vector step is >= 0.1 ms of work
    n_env = n_task * per          # per thread and the OpenMP fork/join does not dominate
    cores = max(1, min(os.cpu_count() or 1, oracle.lib().xo_max_threads()))
    tab = oracle.anymdp_synth(seed=seed, task_index_base=0, n_task=n_task, S=64, A=8, s0_max=4)
    env_task = (np.arange(n_env, dtype=np.int32) % n_task).astype(np.int32)   # neighbours use different tasks
    ora = oracle.AnyMDPOracle(tab, env_task)
    ora.reset(seed, 0, 0)
    rng = np.random.RandomState(0)
    acts = rng.randint(0, 8, (64, n_env)).astype(np.int32)
    for k in range(5):
        ora.step(seed, 0, 1 + k, acts[k % 64], 2, n_threads=cores)
    t0 = time.perf_counter()
    k = 0
    while True:
        for _ in range(10):
            ora.step(seed, 0, 100 + k, acts[k % 64], 2, n_threads=cores)
            k += 1
        if time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    out = {"value": n_env * k / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
           "sample": "oracle/xeno_oracle.c step (OpenMP, %d threads), %d envs over %d distinct S=64,A=8 tasks "
                     "(1 GiB of tables), %d vector steps in %.1f s" % (cores, n_env, n_task, k, dt)}
    # secondary line: the reference's own execution style — one env per Python object, one
    # step() per call, NumPy global RNG — on ONE core, same task shape, bounded to a few seconds
    try:
        from oracle import py_ref_style
        sub = {kk: (v[:64] if hasattr(v, "shape") else v) for kk, v in tab.items()}
        v, n_py = py_ref_style.time_python_loop(sub, min(3.0, seconds))
        out["python_loop"] = {"value": v, "unit": "env-steps/s", "cores": 1, "kind": "port",
                              "sample": "oracle/py_ref_style.py: %d env objects stepped one by one (the reference's "
                                        "style), S=64,A=8, one core" % n_py}
    except Exception as ex:   # never lose the bench line to the secondary baseline
        out["python_loop"] = {"error": repr(ex)}
    return out


def pmc_traffic(n_env, n_task, search):
    """HBM bytes per launch of the step kernel from the committed PMC run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate passes, gfx950 x2 read correction; scripts/gpu_pmc.sh -> profiles/*pmc_traffic*.json).  Counters cannot
    be read from inside this process, so the figure is the profiled one for the same workload, or null."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            d = json.load(open(f))
            k = d.get("bench_key", {})
            want = "2a" if n_task == n_env else "2b"
            if k.get("workload") == want and k.get("search") in (search, "auto") and k.get("envs_per_gpu") == n_env:
                for name, v in d["kernels"].items():
                    if "step_kernel" in name:
                        return v["traffic_bytes_per_launch_corrected"], os.path.basename(f)
        except Exception:
            pass
    return None, None


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
    if os.environ.get("XV_BENCH_SHARE_GPU"):   # functional test of the N>1 path on a 1-GPU box (not a measurement)
        local = 0
    torch.cuda.set_device(local)
    dist = None
    dist_note = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("XV_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
                probe = torch.ones(1, device="cuda:%d" % local)
                dist.all_reduce(probe)                          # fail here, not inside the timed region
                torch.cuda.synchronize()
            else:
                dist.init_process_group(backend)
        except Exception as ex:   # keep the scaling measurement alive: barrier and MAX over ranks through gloo
            dist_note = "RCCL unavailable (%r): gloo used for the barrier and the MAX over ranks, no all-gather" % (ex,)
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
                dist.init_process_group("gloo")
            except Exception as ex2:
                sys.exit("bench.py: no usable torch.distributed backend: %r / %r" % (ex, ex2))

    from xenoverse_amd import _lib
    from xenoverse_amd.anymdp import AnyMDPVecEnv

    n_env = args.envs
    n_task = args.tasks if args.tasks > 0 else n_env
    S, A = 64, 8
    env = AnyMDPVecEnv(n_env, device="cuda:%d" % local, seed=args.seed, env_id_base=rank * n_env,
                       autoreset_mode="same_step")
    tab = make_tables(env.engine, torch, _lib, n_task, rank * n_task, args.seed + 1, S, A)
    per = n_env // n_task
    env_task = (torch.arange(n_env, device=env.device, dtype=torch.int32) // per).contiguous()
    env.set_task(tab, env_task_index=env_task)
    env.set_search(args.search)
    env.set_step_many_graph(args.graph)
    search = {"auto": "fence"}.get(args.search, args.search)
    P = args.period
    g = torch.Generator(device=env.device)
    g.manual_seed(args.seed + 17 * rank)
    actions = torch.randint(0, A, (P, n_env), generator=g, device=env.device, dtype=torch.int32)
    env.reset()
    ring = env.step_many(1, actions)   # allocates the [P, N] output ring

    # exchange step: all-gather of each finished T-step rollout chunk (14-B records) over RCCL,
    # on a side stream so that it overlaps the next chunk's stepping.  Stepping itself needs no collective.
    # the all-gather runs on device buffers: RCCL only (gloo would stage 16 MB per rank and chunk through the host)
    do_gather = world > 1 and not args.no_allgather and dist_note is None and \
        (dist.get_backend() == "nccl" or bool(os.environ.get("XV_BENCH_FORCE_GATHER")))   # the latter: watchdog tests
    gather = None
    gather_note = dist_note or "none"
    if do_gather:
        try:
            from xenoverse_amd.distributed import REC_BYTES, RolloutGather, pack_records
            gather = RolloutGather((P, n_env, REC_BYTES), device=env.device)
            gather_note = "all_gather of %d-step rollout chunks, %d B/record (RCCL, side stream)" % (P, REC_BYTES)
        except Exception as ex:   # never lose the measurement to a collective set-up problem
            gather = None
            gather_note = "all_gather unavailable: %r" % (ex,)

    def run(k_steps, with_gather=False):
        done = 0
        while done < k_steps:
            n = min(P, k_steps - done)
            env.step_many(n, actions, out=ring)
            done += n
            if with_gather and n == P:
                gather.wait()        # the previous chunk must have left before its buffer is repacked
                pack_records(ring["obs"], actions, ring["reward"], ring["terminated"], ring["truncated"],
                             out=gather.local)
                gather.launch()
        if with_gather:
            gather.wait()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_pass(with_gather):
        run(args.warmup, with_gather)
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        run(args.steps, with_gather)
        e1.record()
        barrier()
        w = time.perf_counter() - t0
        ms = e0.elapsed_time(e1)          # HIP events on the stream the step kernels were launched on
        if dist is not None:              # MAX over ranks
            tt = torch.tensor([w, ms], dtype=torch.float64,
                              device=env.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            w, ms = float(tt[0]), float(tt[1])
        return w, ms

    # pass 1 (the reported value): sharded stepping, no data-path collective — envs are independent
    wall, ev_ms = timed_pass(False)
    # pass 2 (N > 1): the same with every finished rollout chunk all-gathered to all ranks, overlapped.  A watchdog
    # guards it: should the collective stall, rank 0 still prints the pass-1 line (the measurement) and every rank exits.
    wall_g = None
    state = {"emit": None}
    if gather is not None:
        import threading

        def _bail():
            if rank == 0 and state["emit"] is not None:
                state["emit"]("all_gather pass did not finish within %d s: skipped" % args.gather_timeout)
            os._exit(0)
        state["pass1"] = (wall, ev_ms)
        watchdog = threading.Timer(args.gather_timeout, _bail)
        watchdog.daemon = True
    errs = env.check_errors()

    fused = None
    if args.fused:
        T = P
        env.rollout(actions)
        torch.cuda.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(1, args.steps // T)
        f0.record()
        for _ in range(reps):
            env.rollout(actions, out=ring)
        f1.record()
        torch.cuda.synchronize()
        fused = n_env * T * reps / (f0.elapsed_time(f1) * 1e-3)

    def report(extra_note=None):
        if rank != 0 or state.get("done"):
            return
        state["done"] = True
        total_steps = world * n_env * args.steps
        kern_us = ev_ms * 1e3 / args.steps
        algo = ALGO_BYTES_PER_ENV_STEP[8] * n_env
        achieved = algo / (kern_us * 1e-6) / 1e9
        traffic, traffic_src = pmc_traffic(n_env, n_task, search)
        out = {
            "metric": "env-steps/sec (whole node), anymdp |S|=64 |A|=8, 65k envs/GPU",
            "value": total_steps / wall, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "anymdp S=64 A=8, %d envs/GPU, %d tasks/GPU (%s), fp64 CDF rows, "
                                   "SAME_STEP auto-reset, random actions"
                                   % (n_env, n_task, "2a distinct: one task per env" if n_task == n_env
                                      else "2b shared"),
                       "envs_per_gpu": n_env, "tasks_per_gpu": n_task, "S": S, "A": A,
                       "table_gib_per_gpu": round(n_task * S * A * (1 + (S + 6) // 7) * 128 / 2**30, 2),
                       "launch": "one step kernel per vector step (xv_anymdp_step_many)",
                       "search": search,
                       "exchange": gather_note if extra_note is None else gather_note + "; " + extra_note,
                       "device_error_flags": errs},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "anymdp_step_kernel<false, %d, false, false>  (INJECT, blocks per fence entry | "
                                   "0 = binary search, ROLLOUT, TICKDEV)" % (1 if search == "fence" else 0),
                         "avg_launch_us": kern_us,
                         "algorithmic_bytes_per_launch": algo},
        }
        if fused is not None:
            out["fused_rollout_env_steps_per_s_rank0"] = fused
        if wall_g is not None:
            chunks = args.steps // P
            out["with_allgather"] = {"value": total_steps / wall_g, "unit": "env-steps/s",
                                     "gathered_GB_per_s_per_rank": chunks * P * n_env * REC_BYTES * (world - 1) / wall_g / 1e9}
        if not args.no_cpu_baseline and world == 1:      # the CPU line is measured at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.seed)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if gather is not None:      # pass 2, guarded
        state["emit"] = report
        watchdog.start()
        try:
            wall_g, _ = timed_pass(True)
        except Exception as ex:
            gather_note += "; failed: %r" % (ex,)
        watchdog.cancel()
    report()
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
