#!/usr/bin/env python3
"""Experiment: K sub-batch views, EACH stepping with the overlapped step_many (2 streams per view): do phase-shifted
sub-batches smooth the line-request bursts?  65,536 envs, config 2a / 2b, long runs (the host needs ~15 us per graph launch).

  python scripts/devtools/probe_views_overlap.py [--tasks 0] [--steps 19200] [--ks 1,2,4]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--tasks", type=int, default=0)
    ap.add_argument("--steps", type=int, default=19200)
    ap.add_argument("--period", type=int, default=32)
    ap.add_argument("--ks", default="1,2,4")
    a = ap.parse_args()
    import torch
    import bench
    from xenoverse_amd import _lib
    from xenoverse_amd.anymdp import AnyMDPVecEnv
    n, S, A, P = a.envs, 64, 8, a.period
    n_task = a.tasks if a.tasks > 0 else n
    env = AnyMDPVecEnv(n, seed=1234, autoreset_mode="same_step", bucket_lines="off")
    tab = bench.make_tables(env.engine, torch, _lib, n_task, 0, 1235, S, A)
    env.set_task(tab, env_task_index=(torch.arange(n, device=env.device, dtype=torch.int32) // (n // n_task)).contiguous())
    env.set_search("auto", n_bucket=16)
    env.set_step_many_graph("on")
    env.reset()
    for K in [int(x) for x in a.ks.split(",")]:
        subs = [env] if K == 1 else env.split(K)
        rings, acts = [], []
        for s in subs:
            s.set_step_many_graph("on")
            s.set_step_many_overlap(True)
            with torch.cuda.stream(s.stream if K > 1 else torch.cuda.current_stream()):
                ac = torch.randint(0, A, (P, s.num_envs), device=env.device, dtype=torch.int32)
                acts.append(ac)
                rings.append(s.step_many(4 * P, ac))
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s, ac, r in zip(subs, acts, rings):
                s.step_many(a.steps, ac, out=r)
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        errs = env.check_errors()
        print(json.dumps({"views": K, "envs": n, "tasks": n_task, "steps": a.steps, "us_per_vector_step": best * 1e6 / a.steps,
                          "env_steps_per_s": n * a.steps / best, "host_issue_ms": t_issue * 1e3, "wall_ms": best * 1e3,
                          "overlap_state": [s.step_many_overlap_state for s in subs], "device_error_flags": errs}), flush=True)
    env.close()


if __name__ == "__main__":
    main()
