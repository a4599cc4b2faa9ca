"""Teacher policies for AnyMDP: the ground-truth optimal policy of the reference's AnyMDPSolverOpt
(xenoverse/anymdp/anymdp_solver_opt.py: value iteration on the true tensors with gamma = 0.99, then
argmax_a Q[inner_state]) as a per-task table the device rollout kernel reads (`AnyMDPVecEnv.rollout_teacher`)."""
import numpy as np

from .task_sampler import value_iteration


def optimal_q(task, gamma=0.99):
    return value_iteration(np.asarray(task["transition"], np.float64), np.asarray(task["reward"], np.float64), gamma)


def optimal_policy_table(tasks, S=None, gamma=0.99):
    """uint8[n_task, S]: greedy action per inner state (padded rows: action 0)"""
    if isinstance(tasks, dict):
        tasks = [tasks]
    S = S or max(np.asarray(t["transition"]).shape[0] for t in tasks)
    out = np.zeros((len(tasks), S), np.uint8)
    for i, t in enumerate(tasks):
        q = optimal_q(t, gamma)
        out[i, :q.shape[0]] = q.argmax(1)
    return out
