"""Would stepping a LinDS batch as K independent sub-batches on K streams pay (devtool)?  K LinDSVecEnv of 65,536 / K envs, each
on a stream of its own, their xv_linds_step_many calls issued back to back from one thread, against one env of 65,536."""
import sys
import time

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "scripts")
from bench_families import linds_tasks  # noqa: E402
from xenoverse_amd.engine import Engine  # noqa: E402
from xenoverse_amd.linds import LinDSVecEnv  # noqa: E402


def build(n, stream, seed):
    with torch.cuda.stream(stream):
        eng = Engine("cuda:0", seed=seed)
        env = LinDSVecEnv(n, autoreset_mode="same_step", engine=eng)
        env.set_task(linds_tasks(n // 64))
        env.reset()
        a = torch.rand((8, n, 8), device="cuda") * 2 - 1
        ring = env.step_many(8, a)
    return env, a, ring


if __name__ == "__main__":
    N, steps = 65536, 2048
    for K in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(K)]
        parts = [build(N // K, streams[j], 3 + j) for j in range(K)]
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for j, (env, a, ring) in enumerate(parts):
                with torch.cuda.stream(streams[j]):
                    env.step_many(steps, a, out=ring)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("K = %d sub-batches of %d envs: %.2f us per 65,536-env vector step" % (K, N // K, best / steps * 1e6), flush=True)
        for env, _, _ in parts:
            env.close()
