#!/usr/bin/env python3
"""A/B of xv_anymdp_step_many_chains on BASELINE config 2 (65,536 envs, S=64, A=8): K chains x {streams, one graph} against
the one-chain step_many, same workload, same process.  One JSON line per variant on stdout.

  python scripts/devtools/probe_chains.py [--tasks 0] [--steps 2000] [--period 32] [--ks 1,2,4,8] [--hows streams,graph]
                                          [--search auto] [--repeats 7] [--short 20]

GPU_MAX_HW_QUEUES / DEBUG_HIP_FORCE_GRAPH_QUEUES are read by the HIP runtime at start-up: set them in the environment of
the process (scripts/runs_r05/gpu_a.sh runs this file once per setting).
"""
import argparse
import gc
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--tasks", type=int, default=0)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--short", type=int, default=20, help="also time bursts of this many steps (the driver's --steps); 0 = off")
    ap.add_argument("--period", type=int, default=32)
    ap.add_argument("--ks", default="1,2,4,8")
    ap.add_argument("--hows", default="streams,graph")
    ap.add_argument("--search", default="auto")
    ap.add_argument("--repeats", type=int, default=7)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--tag", default="")
    ap.add_argument("--overlap", action="store_true", help="also time chains=1 with the overlapped step_many")
    args = ap.parse_args()

    import torch
    import bench
    from xenoverse_amd import _lib
    from xenoverse_amd.anymdp import AnyMDPVecEnv

    n_env = args.envs
    n_task = args.tasks if args.tasks > 0 else n_env
    S, A = 64, 8
    env = AnyMDPVecEnv(n_env, device="cuda:0", seed=args.seed, autoreset_mode="same_step", bucket_lines="off")
    tab = bench.make_tables(env.engine, torch, _lib, n_task, 0, args.seed + 1, S, A)
    env_task = (torch.arange(n_env, device=env.device, dtype=torch.int32) // (n_env // n_task)).contiguous()
    env.set_task(tab, env_task_index=env_task)
    if args.search == "auto":
        env.set_search("auto", n_bucket=16)
    elif args.search == "bucket":
        env.set_search("bucket", n_bucket=16)
    else:
        env.set_search(args.search)
    env.set_step_many_graph("on")
    g = torch.Generator(device=env.device)
    g.manual_seed(args.seed)
    env.reset()
    meta = {"envs": n_env, "tasks": n_task, "search": env.effective_search, "tag": args.tag,
            "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
            "DEBUG_HIP_FORCE_GRAPH_QUEUES": os.environ.get("DEBUG_HIP_FORCE_GRAPH_QUEUES")}

    def timed(n_steps, P, K, how, reps):
        actions = torch.randint(0, A, (P, n_env), generator=g, device=env.device, dtype=torch.int32)
        ring = env.step_many(P, actions, chains=K, how=how)
        env.step_many(max(P, min(n_steps, 200)), actions, out=ring, chains=K, how=how)
        torch.cuda.synchronize()
        us = []
        gc.collect()
        gc.disable()
        try:
            for _ in range(reps):
                torch.cuda.synchronize()
                env.engine.event_record(0)
                env.step_many(n_steps, actions, out=ring, chains=K, how=how)
                env.engine.event_record(1)
                torch.cuda.synchronize()
                us.append(env.engine.event_elapsed_ms() * 1e3 / n_steps)
        finally:
            gc.enable()
        us.sort()
        return us[len(us) // 2], us[0], us[-1]

    variants = [(K, how, False) for K in [int(x) for x in args.ks.split(",")] for how in (args.hows.split(",") if K > 1 else ["streams"])]
    if args.overlap:
        variants.append((1, "overlap", True))
        variants.append((1, "streams", False))      # and once more without, after it
    for K, how, ov in variants:
        if True:
            row = dict(meta, chains=K, how=how)
            env.set_step_many_overlap(ov)
            if ov:
                how = "streams"
            try:
                med, lo, hi = timed(args.steps, args.period, K, how, args.repeats)
                row.update(steps=args.steps, period=args.period, us_per_step=med, us_min=lo, us_max=hi,
                           env_steps_per_s=n_env / (med * 1e-6))
                if args.short > 0:
                    m2, l2, h2 = timed(args.short, min(args.period, args.short), K, how, max(args.repeats, 15))
                    row.update(short_steps=args.short, short_us_per_step=m2, short_us_min=l2,
                               short_env_steps_per_s=n_env / (m2 * 1e-6))
                row["device_error_flags"] = env.check_errors()
                row["overlap_state"] = env.step_many_overlap_state
            except Exception as ex:
                row["error"] = repr(ex)
            print(json.dumps(row), flush=True)
    env.close()


if __name__ == "__main__":
    main()
