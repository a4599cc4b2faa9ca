"""Sample AnyMDP tasks with the REFERENCE sampler (container only) and cache them as pickles.

`AnyMDPTaskSampler(64, 8, seed=1)` takes ~9 minutes without numba, so the sampled task
dicts are cached under oracle/_cache/ (git-ignored, gpurun-ignored) and reused by oracle/gen_golden.py.
usage: python oracle/sample_ref_tasks.py NS NA SEED [pomdp|mtpomdp]
"""
import os
import pickle
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _refimport  # noqa: E402

CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_cache")


def cache_path(ns, na, seed, kind="mdp"):
    return os.path.join(CACHE, "anymdp_%s_%d_%d_seed%d.pkl" % (kind, ns, na, seed))


def get(ns, na, seed, kind="mdp"):
    p = cache_path(ns, na, seed, kind)
    if os.path.exists(p):
        with open(p, "rb") as f:
            return pickle.load(f)
    _, ts = _refimport.anymdp()
    t = time.time()
    if kind == "mdp":
        task = ts.AnyMDPTaskSampler(ns, na, seed=seed)
    elif kind == "pomdp":
        task = ts.AnyPOMDPTaskSampler(ns, na, observation_space=ns, seed=seed)
    elif kind == "mtpomdp":
        task = ts.MultiTokensAnyPOMDPTaskSampler(ns, na, observation_space=ns, seed=seed)
    else:
        raise ValueError(kind)
    os.makedirs(CACHE, exist_ok=True)
    with open(p, "wb") as f:
        pickle.dump(task, f)
    print("sampled %s in %.1fs" % (p, time.time() - t), flush=True)
    return task


if __name__ == "__main__":
    ns, na, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    kind = sys.argv[4] if len(sys.argv) > 4 else "mdp"
    get(ns, na, seed, kind)
