"""Throughput task sampling on the GPU: `sample_tasks_device` returns n accepted AnyMDP tasks as device tables that
`AnyMDPVecEnv.set_task` takes as they are (no host round trip), generated, repaired and accepted by
`xv_anymdp_sample_tasks` (csrc/anymdp_sampler.hip), one workgroup per candidate.

Same generative model and acceptance test as the reference's AnyMDPTaskSampler (xenoverse/anymdp/task_sampler.py:15-65,
task_sampler_utils.py:65-256, solver.py:84-148) with counter-based randomness: task k of (seed, state_space,
action_space) is the k-th ACCEPTED candidate in candidate order — a pure function of those arguments, independent of
the batch size used to find it — but not the task NumPy's stream would give for that seed (for that, the host sampler
`AnyMDPTaskSampler(seed=...)`).  Supported sizes: 8 <= state_space <= 64 with state_space * action_space <= 512 (a
candidate's transition tensor lives in registers) and, round 4, state_space <= 256 with state_space * action_space <= 4096
(rows in a transposed global scratch: the defaults of the reference's Garnet (128) and multi-token (256) samplers).
"""
import ctypes as C
import time

import numpy as np
import torch

from .. import _lib
from ..engine import Engine
from .tables import row_lines

STATUS = ("accepted", "unrepairable", "value_gap", "occupancy", "no_convergence")

INFO_DTYPE = np.dtype([("status", "<i4"), ("goal", "<i4"), ("n_s0", "<i4"), ("repair_rounds", "<i4"), ("s0", "<i4", 4),
                       ("sweeps", "<i4", 8), ("band_lo", "<i4", 256), ("band_hi", "<i4", 256), ("state_map", "<i4", 256),
                       ("s_e", "u1", 256), ("max_steps", "<f8"), ("gini", "<f8"), ("ent", "<f8"), ("gap_min", "<f8"),
                       ("s0_prob", "<f8", 4)], align=True)     # xv_anymdp_cand_info (include/xeno.h)


def sample_candidates(engine, seed, cand_base, n_cand, S, A, s0_max=4, tables=True, dense=False, info=False):
    """One launch over candidates [cand_base, cand_base + n_cand).  -> dict with `status` int32[n_cand] (device) and,
    as requested, the table slots (written where status == 0), dense fp64 tensors and the per-candidate info records."""
    d = engine.device
    out = {"status": torch.empty(n_cand, dtype=torch.int32, device=d)}
    if tables:
        out.update(rows=torch.empty((n_cand, S, A, row_lines(S), 16), dtype=torch.float64, device=d),
                   state_map=torch.empty((n_cand, S), dtype=torch.int32, device=d),
                   term_mask=torch.empty((n_cand, (S + 63) // 64), dtype=torch.int64, device=d),
                   s0_cdf=torch.empty((n_cand, s0_max), dtype=torch.float64, device=d),
                   s0_ids=torch.empty((n_cand, s0_max), dtype=torch.int32, device=d),
                   max_steps=torch.empty(n_cand, dtype=torch.int32, device=d))
    if dense:
        for k in ("transition", "reward", "reward_noise"):
            out[k] = torch.empty((n_cand, S, A, S), dtype=torch.float64, device=d)
    if info:
        out["info_raw"] = torch.zeros((n_cand, INFO_DTYPE.itemsize), dtype=torch.uint8, device=d)
    p = lambda k: _lib.ptr(out.get(k))
    _lib.check(engine.lib.xv_anymdp_sample_tasks(
        engine.handle, int(seed) & (2**64 - 1), int(cand_base), int(n_cand), S, A, s0_max, p("rows"), p("state_map"),
        p("term_mask"), p("s0_cdf"), p("s0_ids"), p("max_steps"), p("transition"), p("reward"), p("reward_noise"),
        p("info_raw"), p("status")))
    if info:
        engine.sync()
        out["info"] = out.pop("info_raw").cpu().numpy().view(INFO_DTYPE).reshape(n_cand)
    return out


def sample_observation_model_device(engine, seed, n_tasks, state_space, observation_space=64, observation_tokens=1,
                                    density=0.20, maximum_distribution=4, task_base=0):
    """Observation models of AnyPOMDPTaskSampler / MultiTokensAnyPOMDPTaskSampler (task_sampler.py:78-87, :103-117) for
    n_tasks tasks on the device -> obs_cdf float64[n_tasks, observation_tokens, state_space, observation_space] (device):
    sparse random rows (exactly round(density * S * n_obs) cells per matrix), empty rows fixed, normalised, as CDFs."""
    out = torch.empty((n_tasks, observation_tokens, state_space, observation_space), dtype=torch.float64, device=engine.device)
    _lib.check(engine.lib.xv_anymdp_sample_observation_model(
        engine.handle, int(seed) & (2**64 - 1), int(task_base), int(n_tasks), int(state_space), int(observation_space),
        int(observation_tokens), float(density), float(maximum_distribution), _lib.ptr(out)))
    return out


def sample_tasks_device(n_tasks, state_space=64, action_space=5, seed=0, device="cuda:0", engine=None, batch=None,
                        s0_max=4, dense=False, max_candidates=None, task_type="MDP", observation_space=64,
                        observation_tokens=4, action_tokens=2, density=0.20, maximum_distribution=4):
    """n_tasks accepted tasks -> dict of device tensors (keys as anymdp.tables.build_tables: S, A, s0_max, rows,
    state_map, term_mask, s0_cdf, s0_ids, max_steps; with dense=True also transition / reward / reward_noise fp64 and
    max_steps_real) + "stats" (candidates tried, status histogram, seconds).
    task_type "POMDP" / "MTPOMDP" (the reference's AnyPOMDPTaskSampler / MultiTokensAnyPOMDPTaskSampler, whose arguments
    observation_space, observation_tokens, action_tokens, density, maximum_distribution are taken as they are) adds the
    observation model of every task, sampled on the device too: obs_cdf, n_obs, d_obs, d_act, task_type — the dict
    `AnyMDPVecEnv.set_task` takes as it is."""
    if task_type not in ("MDP", "POMDP", "MTPOMDP"):
        raise NotImplementedError(f"Unknown task type: {task_type}")
    S, A = int(state_space), int(action_space)
    own = engine is None
    eng = Engine(device) if own else engine
    try:
        if batch is None:
            per = S * A * row_lines(S) * 128 + (3 * S * A * S * 8 if dense else 0)
            if S > 64 or S * A > 512:      # the candidate's scratch: transposed transition rows + two S x S matrices
                per += ((S * A + 63) // 64 * 64 * S + 2 * S * S) * 8
            batch = int(max(64, min(8192, (6 << 30) // per, 4 * n_tasks + 64)))
        keys = ["rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps"] + \
            (["transition", "reward", "reward_noise"] if dense else [])
        got = {k: [] for k in keys}
        hist = np.zeros(len(STATUS), np.int64)
        n_acc, base, t0 = 0, 0, time.perf_counter()
        while n_acc < n_tasks:
            if max_candidates is not None and base >= max_candidates:
                raise RuntimeError("only %d of %d tasks accepted within %d candidates" % (n_acc, n_tasks, base))
            r = sample_candidates(eng, seed, base, batch, S, A, s0_max, tables=True, dense=dense)
            st = r["status"]
            acc = torch.nonzero(st == 0).flatten()[: n_tasks - n_acc]
            last = int(acc[-1]) + 1 if len(acc) and n_acc + len(acc) >= n_tasks else batch
            hist += np.bincount(st[:last].cpu().numpy(), minlength=len(STATUS))[: len(STATUS)]
            for k in keys:
                got[k].append(r[k].index_select(0, acc))
            n_acc += int(len(acc))
            base += batch
        out = dict(S=S, A=A, s0_max=s0_max, **{k: torch.cat(v) for k, v in got.items()})
        if task_type != "MDP":
            d_obs, d_act = (int(observation_tokens), int(action_tokens)) if task_type == "MTPOMDP" else (1, 1)
            out.update(task_type=task_type, n_obs=int(observation_space), d_obs=d_obs, d_act=d_act,
                       obs_cdf=sample_observation_model_device(eng, seed, n_tasks, S, observation_space, d_obs, density,
                                                               maximum_distribution))
        eng.sync()
        out["stats"] = dict(candidates=int(hist.sum()), accepted=int(hist[0]), seconds=time.perf_counter() - t0,
                            status={STATUS[i]: int(hist[i]) for i in range(len(STATUS))}, batch=batch)
        return out
    finally:
        if own:
            eng.close()


def task_dict_from_dense(transition, reward, reward_noise, info_row, S, A):
    """one candidate's dense tensors + info record -> the reference's task dict"""
    n0 = int(info_row["n_s0"])
    return dict(ns=S, na=A, max_steps=float(info_row["max_steps"]), state_mapping=np.array(info_row["state_map"][:S], np.int64),
                task_type="MDP", s_0=np.array(info_row["s0"][:n0], np.int64), s_0_prob=np.array(info_row["s0_prob"][:n0]),
                s_e=np.nonzero(info_row["s_e"][:S])[0].astype(np.int64), transition=np.asarray(transition, np.float64),
                reward=np.asarray(reward, np.float64), reward_noise=np.asarray(reward_noise, np.float64),
                final_goal_terminate=bool(info_row["goal"]))
