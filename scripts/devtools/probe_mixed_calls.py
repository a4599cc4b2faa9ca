"""Per-call times of MixedShare.step_many in the order bench_mixed.py issues things (devtool): is a slow overlapped call a
property of the call sequence (a short non-overlapped call first, an RCCL communicator created in between, events)?"""
import gc
import sys
import time

import torch

sys.path.insert(0, ".")
from xenoverse_amd.mixed_shard import MixedShare  # noqa: E402


def calls(sh, n, reps):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sh.step_many(n)
        torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) / n * 1e6, 2))
    return out


if __name__ == "__main__":
    for variant in sys.argv[1:] or ["plain"]:
        sh = MixedShare(0, 1, 16384, 8192, 8192, T=32, seed=3)
        sh.set_overlap(True)
        sh.random_actions(5)
        sh.reset()
        if "short" in variant:
            sh.step_many(32)
            torch.cuda.synchronize()
        g = None
        if "gather" in variant:
            from xenoverse_amd.distributed import RolloutGather
            g = RolloutGather((sh.chunk.bytes_per_rank,), device="cuda", transport="rccl", rank=0, world=1)
        if "gc" in variant:
            gc.collect(); gc.disable()
        if "warm" in variant:
            sh.step_many(256)
        if "event" in variant:
            sh.ea.engine.event_record(0)
        print(variant, calls(sh, 2048, 8), "state", sh.overlap_state, "flags", sh.check_errors(), flush=True)
        gc.enable()
        if g is not None:
            g.close()
        sh.set_overlap(False)
        sh.close()
