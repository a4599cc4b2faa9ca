#!/usr/bin/env python3
"""Host build (tables.build_tables) against device build (tables.build_tables_device, xv_anymdp_build_rows) of AnyMDP task
tables from raw reference-format task dicts: seconds per task and host memory, 64 x 8 tasks.

  python scripts/devtools/probe_set_task.py [--tasks 1024] [--distinct 16]
"""
import argparse
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tasks", type=int, default=1024)
    ap.add_argument("--distinct", type=int, default=16)
    a = ap.parse_args()
    import numpy as np
    from xenoverse_amd.anymdp import AnyMDPTaskSampler, build_tables
    from xenoverse_amd.anymdp.tables import build_tables_device
    from xenoverse_amd.engine import Engine
    base = [AnyMDPTaskSampler(64, 8, seed=k) for k in range(a.distinct)]
    # every task its own arrays (as a loaded task set has them): copies of the distinct ones
    tasks = [dict(base[k % a.distinct], transition=base[k % a.distinct]["transition"].copy(),
                  reward=base[k % a.distinct]["reward"].copy(), reward_noise=base[k % a.distinct]["reward_noise"].copy())
             for k in range(a.tasks)]
    eng = Engine("cuda:0")
    build_tables_device(tasks[:64], eng)
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    t0 = time.perf_counter(); d = build_tables_device(tasks, eng); t_dev = time.perf_counter() - t0
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    t0 = time.perf_counter(); h = build_tables(tasks); t_host = time.perf_counter() - t0
    rss2 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    import torch
    same = bool(torch.equal(d["rows"].cpu(), torch.from_numpy(h["rows"])))
    print(json.dumps({"tasks": a.tasks, "S": 64, "A": 8, "device_build_s": t_dev, "host_build_s": t_host,
                      "device_ms_per_task": t_dev * 1e3 / a.tasks, "host_ms_per_task": t_host * 1e3 / a.tasks,
                      "speedup": t_host / t_dev, "rows_bit_equal": same,
                      "max_rss_growth_MiB": {"device_build": (rss1 - rss0) / 1024.0, "host_build": (rss2 - rss1) / 1024.0},
                      "cpus": os.cpu_count()}))
    eng.close()


if __name__ == "__main__":
    main()
