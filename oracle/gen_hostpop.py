"""tests/golden/sampler_hostpop_128x5.npz: per-task statistics of 32 tasks AnyMDPTaskSampler(128, 5, seed=k) — the
reference's GarnetTaskSampler / multi-token default sizes are out of reach of the interpreted reference in this container
(518 s per 64x8 task; hours at 128), so the population comes from the SEED-COMPATIBLE HOST SAMPLER
(xenoverse_amd/anymdp/task_sampler.py), which returns the reference's own task bit for bit wherever the reference could be
run (16x4 seeds 0-3 and 100-227, 64x8 seed 1: tests/test_host_samplers.py).  The device sampler's accepted tasks at 128x5
are compared with these statistics (tests/test_gpu_sampler.py).   python oracle/gen_hostpop.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stats(task):
    T = np.asarray(task["transition"])
    S = T.shape[0]
    se = np.zeros(S, bool)
    se[np.asarray(task["s_e"], np.int64)] = True
    rows = T[~se]
    nz = rows > 0
    band = np.mean([np.ptp(np.nonzero(x.any(0))[0]) + 1 for x in nz])
    return dict(band=band, nnz=nz.sum(-1).mean(), pit_frac=se.mean(), goal=float(se[S - 1]), max_steps=float(task["max_steps"]),
                n_s0=len(task["s_0"]), live_p_min=float(rows[nz].min()))


def main(S=128, A=5, n=32):
    from xenoverse_amd.anymdp.task_sampler import AnyMDPTaskSampler
    out = {}
    t0 = time.time()
    rows = [stats(AnyMDPTaskSampler(S, A, seed=1000 + k)) for k in range(n)]
    for k in rows[0]:
        out[k] = np.array([r[k] for r in rows])
    out["seed"] = np.arange(1000, 1000 + n)
    path = os.path.join(ROOT, "tests", "golden", "sampler_hostpop_%dx%d.npz" % (S, A))
    np.savez_compressed(path, **out)
    print(path, "%.0f s" % (time.time() - t0), {k: float(np.mean(v)) for k, v in out.items() if k != "seed"})


if __name__ == "__main__":
    main()
