"""AnyMDP task dicts -> struct-of-arrays device tables (the host half of `set_task`).

Reference: `AnyMDPEnv.set_task` (xenoverse/anymdp/anymdp_env.py:32-79) copies every key of the task dict
onto the env object and validates it; `single_step` (:99-100) then calls `numpy.random.choice(n, p=row)`
each step, which internally forms `cdf = cumsum(row); cdf /= cdf[-1]` and searches it.  Here that CDF is
formed ONCE per task, on the host, in fp64 with numpy's own cumsum/divide so that every bit equals what
`choice` would compute, and laid out for the device (layout: include/xeno.h, "AnyMDP": row records of 128-byte
lines — a fence line and blocks of 7 next states carrying CDF entry, {reward, noise} pair, observation id and
terminal flag).

Task dict schema: ns, na, max_steps (float), state_mapping int[n], task_type,
s_0 int[k], s_0_prob f64[k], s_e int[m], transition/reward/reward_noise f64[n,na,n].
"""
import math

import numpy as np

S_MAX = 512
A_MAX = 64


def validate_task(task):
    """The checks of AnyMDPEnv.set_task (anymdp_env.py:48-76), same exception types and messages."""
    ttype = task.get("task_type", "MDP")
    if ttype not in ("MDP", "POMDP", "MTPOMDP"):
        raise NotImplementedError(f"Unknown task type: {ttype}")
    T = np.asarray(task["transition"])
    R = np.asarray(task["reward"])
    assert T.shape == R.shape
    assert T.shape[0] == len(task["state_mapping"]) and T.shape[1] == task["na"]
    assert task["ns"] > 0, "State space must be at least 1"
    assert task["na"] > 1, "Action space must be at least 2"
    s_e = np.asarray(task["s_e"], dtype=np.int64).reshape(-1)
    err = (np.sum(T, axis=-1) - 1.0) ** 2
    if len(s_e) > 0:
        err[s_e] = 0.0
    if (err >= 1.0e-6).any():
        raise Exception(f"Transition Matrix Sum != 1 at {np.where(err >= 1.0e-6)}")
    inter = np.intersect1d(np.asarray(task["s_0"]), s_e)
    if len(inter) > 0:
        raise Exception(f"State {inter} is {task['s_0']} and {task['s_e']}")


def row_cdf(T):
    """cumsum(row)/cumsum(row)[-1] along the last axis, exactly as numpy.random.choice forms it
    (numpy/random/mtrand.pyx `choice`: `cdf = p.cumsum(); cdf /= cdf[-1]`).  All-zero rows (terminal
    states in the reference) become 1.0: they are never sampled from."""
    c = np.cumsum(np.asarray(T, dtype=np.float64), axis=-1)
    last = c[..., -1:]
    zero = last == 0.0
    with np.errstate(invalid="ignore", divide="ignore"):
        c = c / np.where(zero, 1.0, last)
    c = np.where(zero, 1.0, c)
    return c


BLK = 7          # next states per 128-byte block (include/xeno.h, "rows")


def row_blocks(S):
    """Blocks per row record: ceil(S/7), rounded up to a multiple of G = ceil(blocks/16) — a fence entry names G whole
    blocks, so the 16-entry fence line covers any S <= 512 (G = 1 up to S = 112, 2 up to 224, 3 up to 336, 4 up to 448, 5 beyond)."""
    nb = (S + BLK - 1) // BLK
    g = (nb + 15) // 16
    return (nb + g - 1) // g * g


def row_lines(S):
    """Lines of 128 bytes per row record: the fence line + row_blocks(S) blocks (XV_ANYMDP_ROW_LINES)."""
    return 1 + row_blocks(S)


def to_blocked(cdf, rs):
    """Flat per-row arrays -> the device's row records (include/xeno.h, "rows").

    cdf float64[..., S], rs float32[..., S, 2]  ->  float64[..., 1 + NB, 16] with NB = row_blocks(S): line 1+b holds
    7 entries of {cdf (8 B), reward, noise (4 B each)} for next states 7b..7b+6; entries past S hold cdf 2.0
    (never <= u) and a zero pair.  Line 0 (fence) and the last 16 bytes of every block (observation ids and
    terminal flags) are left zero: xv_anymdp_create completes them on the device."""
    cdf = np.asarray(cdf, np.float64)
    rs = np.asarray(rs, np.float32)
    S = cdf.shape[-1]
    NB = row_blocks(S)
    lead = cdf.shape[:-1]
    c = np.full(lead + (NB * BLK,), 2.0, np.float64)
    c[..., :S] = cdf
    r = np.zeros(lead + (NB * BLK, 2), np.float32)
    r[..., :S, :] = rs
    out = np.zeros(lead + (1 + NB, 16), np.float64)
    ent = out[..., 1:, :14].reshape(lead + (NB, BLK, 2))     # a view: 7 entries x {cdf, pair}
    ent[..., 0] = c.reshape(lead + (NB, BLK))
    ent[..., 1] = np.ascontiguousarray(r.reshape(lead + (NB, BLK, 2))).view(np.float64).reshape(lead + (NB, BLK))
    return out


def from_blocked(rows, S):
    """Inverse of to_blocked: -> (cdf float64[..., S], rs float32[..., S, 2]); fence and metadata are ignored."""
    rows = np.ascontiguousarray(rows, np.float64)
    lead = rows.shape[:-2]
    NB = rows.shape[-2] - 1
    ent = rows[..., 1:, :14].reshape(lead + (NB, BLK, 2))
    cdf = np.ascontiguousarray(ent[..., 0]).reshape(lead + (NB * BLK,))[..., :S]
    rs = np.ascontiguousarray(ent[..., 1]).view(np.float32).reshape(lead + (NB * BLK, 2))[..., :S, :]
    return np.ascontiguousarray(cdf), np.ascontiguousarray(rs)


def build_tables(tasks, s0_max=None, validate=True):
    """Stack a list of reference task dicts into the device layout.  All tasks must share `na`; tasks
    with fewer active states than the largest are padded (CDF 1.0, reward 0) — padding is unreachable
    because the last real CDF entry is already 1.0 > u."""
    if isinstance(tasks, dict):
        tasks = [tasks]
    if len(tasks) == 0:
        raise ValueError("empty task list")
    if validate:
        for t in tasks:
            validate_task(t)
    A = int(tasks[0]["na"])
    for t in tasks:
        if int(t["na"]) != A:
            raise ValueError("all tasks of one batch must share the action space size `na`")
    n_list = [int(np.asarray(t["transition"]).shape[0]) for t in tasks]
    # single-state tasks (multi-armed bandits, state_space = 1): the reference terminates every step (`ns < 2`,
    # anymdp_env.py:107-108).  They are embedded in two inner states: 0 = the state, 1 = an absorbing terminal copy
    # of it with the same observation id; every action moves 0 -> 1 and carries R[0, a, 0].
    S = max(max(n_list), 2)
    if not (2 <= S <= S_MAX) or not (2 <= A <= A_MAX):
        raise ValueError(f"unsupported sizes S={S}, A={A} (need 2<=S<={S_MAX}, 2<=A<={A_MAX})")
    k_max = max(len(np.atleast_1d(t["s_0"])) for t in tasks)
    if s0_max is None:
        s0_max = k_max
    if s0_max < k_max:
        raise ValueError("s0_max smaller than the longest s_0 list")
    n_task = len(tasks)
    words = (S + 63) // 64
    cdf = np.ones((n_task, S, A, S), np.float64)
    rs = np.zeros((n_task, S, A, S, 2), np.float32)
    state_map = np.zeros((n_task, S), np.int32)
    term_mask = np.zeros((n_task, words), np.uint64)
    s0_cdf = np.ones((n_task, s0_max), np.float64)
    s0_ids = np.zeros((n_task, s0_max), np.int32)
    max_steps = np.zeros(n_task, np.int32)
    obs_space = np.zeros(n_task, np.int32)
    for i, t in enumerate(tasks):
        n = n_list[i]
        if n == 1:
            cdf[i, 0, :, 0] = 0.0          # cdf = [0, 1, 1, ...]: upper_bound(u) = 1 for every u in [0, 1)
            rs[i, 0, :, 1, 0] = np.asarray(t["reward"], np.float64)[0, :, 0]
            rs[i, 0, :, 1, 1] = np.asarray(t["reward_noise"], np.float64)[0, :, 0]
            state_map[i, :2] = int(np.asarray(t["state_mapping"], np.int64).reshape(-1)[0])
            term_mask[i, 0] |= np.uint64(2)
        else:
            cdf[i, :n, :, :n] = row_cdf(t["transition"])
            rs[i, :n, :, :n, 0] = np.asarray(t["reward"], np.float64)
            rs[i, :n, :, :n, 1] = np.asarray(t["reward_noise"], np.float64)
            state_map[i, :n] = np.asarray(t["state_mapping"], np.int64)
            for s in np.asarray(t["s_e"], np.int64).reshape(-1):
                term_mask[i, int(s) >> 6] |= np.uint64(1) << np.uint64(int(s) & 63)
        s0 = np.atleast_1d(np.asarray(t["s_0"], np.int64))
        p0 = np.atleast_1d(np.asarray(t["s_0_prob"], np.float64))
        c0 = np.cumsum(p0)
        c0 = c0 / c0[-1]
        s0_cdf[i, :len(s0)] = c0
        s0_ids[i, :len(s0)] = s0
        s0_ids[i, len(s0):] = s0[-1]
        # `truncated = steps >= max_steps` with integer steps and a real max_steps (anymdp_env.py:114)
        max_steps[i] = int(min(math.ceil(float(t["max_steps"])), 2**31 - 1))
        obs_space[i] = int(t["ns"])
    return dict(S=S, A=A, s0_max=int(s0_max), cdf=cdf, rs=rs, rows=to_blocked(cdf, rs),
                state_map=state_map, term_mask=term_mask,
                s0_cdf=s0_cdf, s0_ids=s0_ids, max_steps=max_steps, obs_space=obs_space)


def build_obs_tables(tasks, S):
    """POMDP / MTPOMDP observation model -> (obs_cdf float64[n_task, d_obs, S, n_obs], n_obs, d_obs, d_act).
    task["observation_transition"]: f64[n, no] (POMDP, task_sampler.py:78-87) or a list of `do` such matrices
    (MTPOMDP, :103-117); CDF rows formed as numpy.random.choice forms them (anymdp_env.py:150-157)."""
    if isinstance(tasks, dict):
        tasks = [tasks]
    ttype = tasks[0]["task_type"]
    if any(t["task_type"] != ttype for t in tasks):
        raise ValueError("all tasks of one batch must share the task_type")
    if ttype == "MTPOMDP":
        d_obs, d_act = int(tasks[0]["do"]), int(tasks[0]["da"])
    elif ttype == "POMDP":
        d_obs, d_act = 1, 1
    else:
        raise ValueError("task_type %r has no observation model" % (ttype,))
    n_obs = int(tasks[0]["no"])
    out = np.ones((len(tasks), d_obs, S, n_obs), np.float64)
    for i, t in enumerate(tasks):
        mats = t["observation_transition"] if ttype == "MTPOMDP" else [t["observation_transition"]]
        if len(mats) != d_obs or int(t["no"]) != n_obs or (ttype == "MTPOMDP" and int(t["da"]) != d_act):
            raise ValueError("all tasks of one batch must share no / do / da")
        for k, m in enumerate(mats):
            m = np.asarray(m, np.float64)
            out[i, k, :m.shape[0], :] = row_cdf(m)
    return out, n_obs, d_obs, d_act


def device_buildable(tasks):
    """can build_tables_device serve this task list?  One shape for all tasks, at least two states (single-state bandits are
    embedded in two inner states by the host builder), sizes within the engine's limits"""
    if isinstance(tasks, dict) or len(tasks) == 0:
        return False
    sh = np.shape(tasks[0]["transition"])
    if len(sh) != 3 or sh[0] != sh[2] or sh[0] < 2 or sh[0] > S_MAX or not (2 <= sh[1] <= A_MAX):
        return False
    return all(np.shape(t["transition"]) == sh and np.shape(t["reward"]) == sh and np.shape(t["reward_noise"]) == sh and
               int(t["na"]) == sh[1] for t in tasks)


_STAGING = {}      # (device, bytes) -> two pinned staging buffers, kept: pinning 2 x 32 MB costs more than building 100 tasks


def build_tables_device(tasks, engine, chunk_bytes=32 << 20, s0_max=None, validate=True):
    """build_tables with the heavy half on the device (xv_anymdp_build_rows): the transition / reward / reward_noise tensors
    go up in chunks through pinned memory and the row records — CDF as numpy.random.choice forms it, fp32 reward pairs,
    block layout — are written where they will be read; host memory stays O(chunk) beside the caller's task dicts and a
    64 x 8 task takes ~0.1 ms instead of 7.5.  Same rows as build_tables bit for bit (tests/test_gpu_tables.py), same
    checks (anymdp_env.py:48-76; the row-sum check is made by the kernel), same exception types.
    -> dict as build_tables returns, with `rows` a device tensor and without the flat `cdf` / `rs` arrays."""
    import ctypes as C

    import torch

    from .. import _lib
    if not device_buildable(tasks):
        raise ValueError("build_tables_device needs a list of tasks of one shape with at least two states")
    n_task = len(tasks)
    S, A = int(np.shape(tasks[0]["transition"])[0]), int(np.shape(tasks[0]["transition"])[1])
    k_max = max(len(np.atleast_1d(t["s_0"])) for t in tasks)
    s0_max = k_max if s0_max is None else s0_max
    if s0_max < k_max:
        raise ValueError("s0_max smaller than the longest s_0 list")
    words = (S + 63) // 64
    state_map = np.zeros((n_task, S), np.int32)
    term_mask = np.zeros((n_task, words), np.uint64)
    s0_cdf = np.ones((n_task, s0_max), np.float64)
    s0_ids = np.zeros((n_task, s0_max), np.int32)
    max_steps = np.zeros(n_task, np.int32)
    obs_space = np.zeros(n_task, np.int32)
    for i, t in enumerate(tasks):      # the small per-task vectors and the cheap checks of set_task (anymdp_env.py:48-63,74-76)
        if validate:
            ttype = t.get("task_type", "MDP")
            if ttype not in ("MDP", "POMDP", "MTPOMDP"):
                raise NotImplementedError(f"Unknown task type: {ttype}")
            assert np.shape(t["transition"]) == np.shape(t["reward"])
            assert S == len(t["state_mapping"]) and A == t["na"]
            assert t["ns"] > 0, "State space must be at least 1"
            assert t["na"] > 1, "Action space must be at least 2"
        s_e = np.asarray(t["s_e"], np.int64).reshape(-1)
        if validate:
            inter = np.intersect1d(np.asarray(t["s_0"]), s_e)
            if len(inter) > 0:
                raise Exception(f"State {inter} is {t['s_0']} and {t['s_e']}")
        state_map[i] = np.asarray(t["state_mapping"], np.int64)
        for s in s_e:
            term_mask[i, int(s) >> 6] |= np.uint64(1) << np.uint64(int(s) & 63)
        s0 = np.atleast_1d(np.asarray(t["s_0"], np.int64))
        c0 = np.cumsum(np.atleast_1d(np.asarray(t["s_0_prob"], np.float64)))
        c0 = c0 / c0[-1]
        s0_cdf[i, :len(s0)] = c0
        s0_ids[i, :len(s0)] = s0
        s0_ids[i, len(s0):] = s0[-1]
        max_steps[i] = int(min(math.ceil(float(t["max_steps"])), 2**31 - 1))
        obs_space[i] = int(t["ns"])
    dev = engine.device
    lib = engine.lib
    rows = torch.empty((n_task, S, A, row_lines(S), 16), dtype=torch.float64, device=dev)
    tm_dev = torch.from_numpy(term_mask.view(np.int64)).to(dev)
    bad = torch.full((1,), -1, dtype=torch.int64, device=dev)          # ~0 as an unsigned word
    per_task = 3 * S * A * S * 8
    chunk = max(1, min(n_task, int(chunk_bytes) // per_task))
    # two pinned staging buffers: the host fills one while the other's copy and kernel run
    key = (str(dev), 3 * chunk * S * A * S)
    if key not in _STAGING:
        _STAGING.clear()
        _STAGING[key] = [torch.empty(key[1], dtype=torch.float64).pin_memory() for _ in range(2)]
    pins = [p.view(3, chunk, S, A, S) for p in _STAGING[key]]
    devs = [torch.empty((3, chunk, S, A, S), dtype=torch.float64, device=dev) for _ in range(2)]
    evs = [None, None]
    st = engine.torch_stream
    with torch.cuda.stream(st):
        for ci, t0 in enumerate(range(0, n_task, chunk)):
            t1 = min(t0 + chunk, n_task)
            b = ci & 1
            if evs[b] is not None:
                evs[b].synchronize()          # the copy that last read this staging buffer has finished
            pv = pins[b].numpy()
            for k, t in enumerate(tasks[t0:t1]):
                pv[0, k] = t["transition"]
                pv[1, k] = t["reward"]
                pv[2, k] = t["reward_noise"]
            devs[b][:, :t1 - t0].copy_(pins[b][:, :t1 - t0], non_blocking=True)
            evs[b] = torch.cuda.Event()
            evs[b].record(st)
            d = devs[b]
            _lib.check(lib.xv_anymdp_build_rows(engine.handle, t1 - t0, S, A, C.c_void_p(d[0].data_ptr()), C.c_void_p(d[1].data_ptr()),
                                                C.c_void_p(d[2].data_ptr()), C.c_void_p(tm_dev[t0:].data_ptr()),
                                                C.c_void_p(rows[t0:].data_ptr()), C.c_void_p(bad.data_ptr()), t0 * S * A))
    engine.sync()          # (the staging buffers are free again: the next call may fill them)
    if validate:      # the kernel made the row-sum check on the way (anymdp_env.py:66-71)
        w = int(bad.item())
        if w != -1:
            t_, rem = divmod(w, S * A)
            raise Exception(f"Transition Matrix Sum != 1 at (task {t_}, state {rem // A}, action {rem % A})")
    return dict(S=S, A=A, s0_max=int(s0_max), rows=rows, state_map=state_map, term_mask=term_mask, s0_cdf=s0_cdf, s0_ids=s0_ids,
                max_steps=max_steps, obs_space=obs_space)
