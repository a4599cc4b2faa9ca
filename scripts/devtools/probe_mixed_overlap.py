"""Which family's hand-off chain bounds the overlapped mixed step?  Times MixedShare.step_many with one family at its
config-5 size and the other two at one task, with and without the overlap.  (GPU; devtool, not a test.)"""
import sys
import time

import torch

sys.path.insert(0, ".")
from xenoverse_amd.mixed_shard import MixedShare  # noqa: E402


def timed(sh, n, all_times=None):
    sh.step_many(n)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        sh.step_many(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if all_times is not None:
            all_times.append(round(dt / n * 1e6, 2))
        best = min(best, dt)
    return best / n * 1e6


if __name__ == "__main__":
    combos = [(16384, 8192, 8192), (16384, 64, 8), (64, 8192, 8), (64, 64, 8192), (64, 64, 8)]
    if "--linds" in sys.argv:       # what would an overlapped xv_linds_step_many give at config 3's size?
        combos = [(64, 65536, 8), (64, 16384, 8), (64, 32768, 8), (64, 131072, 8)]
    gather = None
    if "--gather" in sys.argv:      # does a live RCCL communicator (its streams and queues) change the picture?
        from xenoverse_amd.distributed import RolloutGather
        gather = RolloutGather((1 << 20,), device="cuda", transport="rccl", rank=0, world=1)
        gather.launch(); gather.wait()
        torch.cuda.synchronize()
        combos = combos[:2]
    for tot in combos:
        row = []
        times = []
        for ov in (False, True):
            sh = MixedShare(0, 1, *tot, T=32, seed=3)
            sh.random_actions(5)
            if ov:
                sh.set_overlap(True)
            sh.reset()
            us = timed(sh, 2048, times)
            row.append((us, sh.overlap_state, sh.check_errors()))
            if ov:
                sh.set_overlap(False)
            sh.close()
        print("envs", tot, "one stream %.2f us  overlapped %.2f us (state %d, flags %d)" % (row[0][0], row[1][0], row[1][1], row[1][2]), times, flush=True)
