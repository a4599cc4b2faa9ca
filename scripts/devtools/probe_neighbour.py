#!/usr/bin/env python3
"""Overlapped step_many at 65,536 envs beside a busy neighbour of the SAME process (round-5 review, weak 4): per-call wall time and
overlap state with a torch matmul loop or an all-gather loop on another stream; faulthandler shows where the threads are if
something stops moving.  usage: probe_neighbour.py matmul|allgather|none [calls]"""
import faulthandler
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch

import oracle
from xenoverse_amd.anymdp import AnyMDPVecEnv, to_blocked


def dev_tables(tab, dev="cuda:0"):
    out = dict(S=tab["S"], A=tab["A"], s0_max=tab["s0_max"])
    tab = dict(tab, rows=to_blocked(tab["cdf"], tab["rs"]))
    for k in ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps"):
        v = np.ascontiguousarray(tab[k])
        if v.dtype == np.uint64:
            v = v.view(np.int64)
        out[k] = torch.from_numpy(v).to(dev)
    return out


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "matmul"
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    faulthandler.dump_traceback_later(int(os.environ.get("PROBE_DUMP_S", "90")), exit=True)
    tab = oracle.anymdp_synth(seed=21, task_index_base=0, n_task=1024, S=64, A=8, s0_max=4)
    n, P = 65536, 32
    env = AnyMDPVecEnv(n, seed=77, autoreset_mode="same_step", bucket_lines="off")
    env.set_task(dev_tables(tab))
    env.set_search("bucket", n_bucket=16)
    env.set_step_many_graph(True)
    env.reset()
    acts = torch.randint(0, 8, (P, n), device=env.device, dtype=torch.int32)
    ring = env.step_many(P, acts)
    torch.cuda.synchronize()
    stop = threading.Event()
    side = torch.cuda.Stream()
    count = [0]
    gather = None
    if kind == "allgather":
        from xenoverse_amd.distributed import RolloutGather
        gather = RolloutGather((32, 65536, 14), device="cuda", transport="rccl", rank=0, world=1)

    def busy():
        torch.cuda.set_device(0)
        with torch.cuda.stream(side):
            if kind == "matmul":
                a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
                b = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
                while not stop.is_set():
                    for _ in range(8):
                        a = torch.tanh(a @ b) * 0.5
                    side.synchronize()
                    count[0] += 8
            elif kind == "allgather":
                while not stop.is_set():
                    gather.launch()
                    gather.wait()
                    side.synchronize()
                    count[0] += 1
    th = None
    if kind != "none":
        th = threading.Thread(target=busy, daemon=True)
        th.start()
        while count[0] == 0 and th.is_alive():
            time.sleep(0.01)
    print("neighbour %s running: %d" % (kind, count[0]), flush=True)
    for overlap in (False, True):
        env.set_step_many_overlap(overlap)
        for c in range(calls):
            c0 = count[0]
            t0 = time.perf_counter()
            env.step_many(20 * P, acts, out=ring)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print("overlap %d call %2d: issue %.2f ms, done %.2f ms = %.2f us/step, state %d, flags %d, neighbour iterations %d"
                  % (overlap, c, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e6 / (20 * P), env.step_many_overlap_state,
                     env.check_errors(), count[0] - c0), flush=True)
    stop.set()
    if th is not None:
        th.join(timeout=30)
    env.set_step_many_overlap(False)
    env.close()
    if gather is not None:
        gather.close()
    print("done", flush=True)


if __name__ == "__main__":
    main()
