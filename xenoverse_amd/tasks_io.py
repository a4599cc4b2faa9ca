"""Task (de)serialisation.

The reference stores tasks with `dump_task / load_task` = pickle of the task dict (xenoverse/utils/tools.py:62-68).
Those functions are kept (same names, same behaviour) so existing task files load, and a batch of tasks can also
be written as one `.npz` of stacked arrays — the struct-of-arrays form the device tables are built from — which
needs no pickle and no reference classes to read back (a RandomFourier command is stored as its orders/coeffs).
"""
import pickle

import numpy as np


def dump_task(file, task):
    with open(file, "wb") as f:
        pickle.dump(task, f)


def load_task(file):
    with open(file, "rb") as f:
        return pickle.load(f)


def save_task_batch(file, family, tasks):
    """family in {"anymdp", "linds", "mazeworld", "cartpole"}: stores the device-table arrays of the batch"""
    if family == "anymdp":
        from .anymdp import build_obs_tables, build_tables
        tab = build_tables(tasks)
        arrs = {k: v for k, v in tab.items() if isinstance(v, np.ndarray) and k not in ("cdf", "rs")}
        meta = dict(S=tab["S"], A=tab["A"], s0_max=tab["s0_max"], task_type=tasks[0].get("task_type", "MDP"))
        if meta["task_type"] != "MDP":
            cdf, n_obs, d_obs, d_act = build_obs_tables(tasks, tab["S"])
            arrs["obs_cdf"] = cdf
            meta.update(n_obs=n_obs, d_obs=d_obs, d_act=d_act)
    elif family == "linds":
        from .linds import build_tables
        tab = build_tables(tasks)
        arrs = {k: v for k, v in tab.items() if isinstance(v, np.ndarray)}
        meta = {k: tab[k] for k in ("NS", "NA", "NO", "NI", "dt")}
    elif family == "mazeworld":
        from .mazeworld import build_tables
        tab = build_tables(tasks)
        arrs = {k: v for k, v in tab.items() if isinstance(v, np.ndarray)}
        meta = {k: tab[k] for k in ("NG", "n_cmd")}
    elif family == "cartpole":
        arrs = {"params": np.array([[t["gravity"], t["masscart"], t["masspole"], t["length"]] for t in tasks], np.float32)}
        meta = {}
    else:
        raise ValueError(family)
    np.savez_compressed(file, __family__=np.array(family), **{"meta_" + k: np.array(v) for k, v in meta.items()}, **arrs)


def load_task_batch(file):
    """-> (family, tables dict) ready for `<Family>VecEnv.set_task(tables)`"""
    z = np.load(file, allow_pickle=False)
    family = str(z["__family__"])
    tab = {}
    for k in z.files:
        if k == "__family__":
            continue
        if k.startswith("meta_"):
            v = z[k]
            tab[k[5:]] = v.item() if v.shape == () else v
        else:
            tab[k] = z[k]
    if family == "cartpole":
        return family, [dict(gravity=float(p[0]), masscart=float(p[1]), masspole=float(p[2]), length=float(p[3]))
                        for p in tab["params"]]
    return family, tab
