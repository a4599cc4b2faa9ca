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


class FourierCommand(object):
    """Duck-type of the reference's RandomFourier (utils/random_nn.py:346-368): `.coeffs`, `.max_steps`."""

    def __init__(self, orders, coeffs, max_steps):
        self.coeffs = [(float(o), np.asarray(c, np.float64)) for o, c in zip(orders, coeffs)]
        self.max_steps = float(max_steps)

    def __call__(self, t):
        x = t / self.max_steps
        y = 0
        for order, coeff in self.coeffs:
            y = y + coeff[:, 0] * np.sin(order * x) + coeff[:, 1] * np.cos(order * x)
        return y


def load_linds_golden(path):
    g = dict(np.load(path, allow_pickle=False))
    dyn = bool(g["is_dynamic"])
    task = dict(state_dim=int(g["state_dim"]), observation_dim=int(g["observation_dim"]),
                action_dim=int(g["action_dim"]), max_steps=int(g["max_steps"]), ld_A=g["ld_A"], ld_B=g["ld_B"],
                ld_C=g["ld_C"], ld_X=g["ld_X"], ld_Y=g["ld_Y"], action_cost=float(g["action_cost"]),
                reward_base=float(g["reward_base"]), terminate_punish=float(g["terminate_punish"]),
                reward_factor=float(g["reward_factor"]), target_valid=g["target_valid"],
                target_type="dynamic_target" if dyn else "static_target",
                initial_states=[x for x in g["initial_states"]], noise_drift=float(g["noise_drift"]),
                target_delay=int(g["target_delay"]))
    task["command"] = (FourierCommand(g["four_orders"], g["four_coeffs"], g["four_period"]) if dyn
                       else g["command"])
    return g, task


def close_rel(a, b, rel=1e-5, abs_=1e-5):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= rel * np.abs(b) + abs_))


def load_maze_golden(path):
    g = dict(np.load(path, allow_pickle=False))
    task = dict(start=tuple(int(x) for x in g["start"]), cell_walls=g["cell_walls"], cell_texts=g["cell_texts"],
                cell_size=float(g["cell_size"]), ground_text=int(g["ground_text"]), ceiling_text=int(g["ceiling_text"]),
                step_reward=float(g["step_reward"]), goal_reward=float(g["goal_reward"]),
                collision_reward=float(g["collision_reward"]), wall_height=float(g["wall_height"]),
                agent_height=float(g["agent_height"]), fol_angle=float(g["fol_angle"]),
                commands_sequence=g["commands_sequence"],
                landmarks_coordinates=[tuple(int(v) for v in x) for x in g["landmarks_coordinates"]],
                cell_landmarks=g["cell_landmarks"])
    return g, task


def frame_mismatch(a, b):
    """fraction of channel values that differ, and the largest absolute difference"""
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    return float(np.mean(d > 0)), int(d.max())


def load_anymdp_tok_golden(path):
    g = dict(np.load(path, allow_pickle=False))
    mt = bool(g["is_mt"])
    task = dict(ns=int(g["ns"]), na=int(g["na"]), max_steps=float(g["max_steps"]), state_mapping=g["state_mapping"],
                task_type="MTPOMDP" if mt else "POMDP", s_0=g["s_0"], s_0_prob=g["s_0_prob"], s_e=g["s_e"],
                transition=g["transition"], reward=g["reward"], reward_noise=g["reward_noise"], no=int(g["no"]))
    if mt:
        task.update(do=int(g["do"]), da=int(g["da"]), observation_transition=[m for m in g["observation_transition"]])
    else:
        task["observation_transition"] = g["observation_transition"][0]
    return g, task
