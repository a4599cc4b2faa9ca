"""Shared helpers for the parity tests (golden loading, task-dict rebuilding)."""
import glob
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLD, prefix + "*.npz")))


def load_anymdp_golden(path):
    g = dict(np.load(path, allow_pickle=False))
    task = dict(ns=int(g["ns"]), na=int(g["na"]), max_steps=float(g["max_steps"]),
                state_mapping=g["state_mapping"], task_type="MDP", s_0=g["s_0"], s_0_prob=g["s_0_prob"],
                s_e=g["s_e"], transition=g["transition"], reward=g["reward"],
                reward_noise=g["reward_noise"])
    return g, task


def close_f32(a, b, rel=1e-5, abs_=1e-6):
    """north_star tolerance for float dynamics: 1e-5 relative (plus a float32 absolute floor)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= rel * np.abs(b) + abs_)
