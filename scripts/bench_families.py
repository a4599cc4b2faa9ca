#!/usr/bin/env python3
"""Per-family measurements beside bench.py's AnyMDP headline: BASELINE.json configs 3 (linds), 4 (mazeworld) and
the CartPole leg of config 5, one GPU.  One JSON line per family: env-steps/s, average launch time of each
kernel (HIP events on the launch stream) and the HBM roofline fraction from the algorithmic bytes of
SURVEY.md §8(d): linds 432 B (+96 B amortised matrices), cartpole 74 B, maze 3*W*H + 64 B.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK = 8000.0


def timed(fn, steps, warmup):
    """us per call from HIP events around `steps` launches issued by a Python loop.  The collector is held off meanwhile:
    a full collection of this process's heap takes ~75 ms (scripts/devtools/probe_slow_window.py), during which the GPU
    starves and a 9-us step reads as 265 us — the 'erratic' family lines of earlier rounds."""
    import gc
    was = gc.isenabled() and not os.environ.get("XV_BENCH_KEEP_GC")
    if was:
        gc.collect()      # before the warm-up: tens of idle milliseconds right in front of the timed launches would let
        gc.disable()      # the GPU clock down
    try:
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
    finally:
        if was:
            gc.enable()
    return e0.elapsed_time(e1) * 1e3 / steps      # us per call


_LINDS_TASKS = {}


def linds_tasks(n_task, distinct=16):
    """`distinct` tasks from the sampler (ns = 32 needs ~0.2 s of rejection sampling each), tiled to n_task: the tables
    are per task index anyway, so every task index owns its own copy of the matrices in HBM"""
    from xenoverse_amd.linds import LinearDSSampler
    if distinct not in _LINDS_TASKS:
        ts = []
        for k in range(distinct):
            t = LinearDSSampler(32, 8, 8, seed=k)
            t["max_steps"] = 500
            ts.append(t)
        _LINDS_TASKS[distinct] = ts
    ts = _LINDS_TASKS[distinct]
    return [ts[k % distinct] for k in range(n_task)]


def bench_linds(args, paths=("mfma", "scalar"), rollout=True):
    from xenoverse_amd.linds import LinDSVecEnv
    n_task, per = 1024, 64
    n = n_task * per
    t0 = time.time()
    tasks = linds_tasks(n_task)
    out = {}
    for path in paths:
        env = LinDSVecEnv(n, autoreset_mode="same_step", seed=1)
        env.set_task(tasks)
        env.set_path(path)
        if os.environ.get("XV_LINDS_AB_CMD_TABLE_OFF"):      # A/B (scripts/runs_r06/gpu_e.sh): commands evaluated, not looked up
            env.set_command_table(False)
        env.reset()
        a = torch.rand((n, 8), device=env.device) * 2 - 1
        from xenoverse_amd import _lib
        from xenoverse_amd.engine import AUTORESET

        def step():
            _lib.check(env.lib.xv_linds_step(env._h, _lib.ptr(a), _lib.ptr(env._obs), _lib.ptr(env._reward),
                                             _lib.ptr(env._term), _lib.ptr(env._trunc), _lib.ptr(env._cmd),
                                             _lib.ptr(env._error), _lib.ptr(env._fobs), AUTORESET["same_step"]))
        us = timed(step, args.steps, args.warmup)
        out[path] = us
        if path == "mfma":      # the same launches issued from C (xv_linds_step_many): no Python between the kernels
            aP = torch.rand((8, n, 8), device=env.device) * 2 - 1
            ring = env.step_many(8, aP)
            k = max(8, args.steps // 8 * 8)
            out["mfma, launches from C"] = timed(lambda: env.step_many(k, aP, out=ring), 3, 1) / k
        if path == "mfma" and rollout:      # fused roll-out: T steps per launch, state resident in registers
            T = 64
            aT = torch.rand((T, n, 8), device=env.device) * 2 - 1
            ro = env.rollout(aT)

            def roll():
                env.rollout(aT, out=ro)
            out["rollout_T64_per_step"] = timed(roll, max(args.steps // T, 5), 2) / T
        env.close()
    algo = 432 * n
    best = min(out[p] for p in out if not p.startswith("rollout"))
    return {"family": "linds", "workload": "ns=32 na=8 no=8 (pads 16/8/16), 65,536 envs = 1,024 tasks x 64", "kernel": "linds_step_mfma_kernel<32, 8, 16, false>",
            "env_steps_per_s": n / (best * 1e-6), "us_per_step": out, "dtype": "f32",
            "roofline": {"bound": "hbm", "achieved": algo / (best * 1e-6) / 1e9, "peak": HBM_PEAK, "unit": "GB/s",
                         "frac": algo / (best * 1e-6) / 1e9 / HBM_PEAK, "algorithmic_bytes_per_env_step": 432},
            "mfma_flops_per_env_step": 3072, "achieved_tflops": 3072 * n / (out["mfma"] * 1e-6) / 1e12,
            "setup_s": round(time.time() - t0, 1)}


def bench_linds_sweep(args):
    """step time of the LinDS matrix kernel against the batch size (64 envs per task): fixed cost vs cost per byte"""
    from xenoverse_amd import _lib
    from xenoverse_amd.engine import AUTORESET
    from xenoverse_amd.linds import LinDSVecEnv
    rows = []
    for n in (4096, 16384, 32768, 65536, 131072, 262144):
        env = LinDSVecEnv(n, autoreset_mode="same_step", seed=1)
        env.set_task(linds_tasks(n // 64))
        env.reset()
        a = torch.rand((n, 8), device=env.device) * 2 - 1

        def step():
            _lib.check(env.lib.xv_linds_step(env._h, _lib.ptr(a), _lib.ptr(env._obs), _lib.ptr(env._reward),
                                             _lib.ptr(env._term), _lib.ptr(env._trunc), _lib.ptr(env._cmd),
                                             _lib.ptr(env._error), _lib.ptr(env._fobs), AUTORESET["same_step"]))
        us = timed(step, args.steps, args.warmup)
        rows.append({"envs": n, "us_per_step": us, "env_steps_per_s": n / (us * 1e-6), "frac_432B": 432 * n / (us * 1e-6) / 1e9 / HBM_PEAK})
        env.close()
    return {"family": "linds sweep", "lib": os.environ.get("XV_LIB_PATH", "default"), "rows": rows}


def bench_cartpole(args):
    from xenoverse_amd.metacontrol import CartPoleVecEnv, sample_cartpole
    from xenoverse_amd import _lib
    from xenoverse_amd.engine import AUTORESET
    n = 65536
    env = CartPoleVecEnv(n, frameskip=1, autoreset_mode="same_step", seed=1)
    env.set_task([sample_cartpole(seed=k) for k in range(1024)])
    env.reset()
    a = torch.randint(0, 2, (n,), device=env.device, dtype=torch.int32)

    def step():
        _lib.check(env.lib.xv_cartpole_step(env._h, _lib.ptr(a), _lib.ptr(env._obs), _lib.ptr(env._reward),
                                            _lib.ptr(env._term), _lib.ptr(env._trunc), _lib.ptr(env._fobs),
                                            AUTORESET["same_step"]))
    us = timed(step, args.steps, args.warmup)
    T = 64
    aT = torch.randint(0, 2, (T, n), device=env.device, dtype=torch.int32)
    ro = env.rollout(aT)
    us_roll = timed(lambda: env.rollout(aT, out=ro), max(args.steps // T, 5), 2) / T
    env.close()
    algo = 74 * n
    return {"family": "cartpole", "workload": "65,536 envs, 1,024 tasks, frameskip 1", "dtype": "f32",
            "env_steps_per_s": n / (us * 1e-6), "us_per_step": us, "rollout_T64_us_per_step": us_roll,
            "rollout_env_steps_per_s": n / (us_roll * 1e-6),
            "roofline": {"bound": "hbm", "achieved": algo / (us * 1e-6) / 1e9, "peak": HBM_PEAK, "unit": "GB/s",
                         "frac": algo / (us * 1e-6) / 1e9 / HBM_PEAK, "frac_rollout": algo / (us_roll * 1e-6) / 1e9 / HBM_PEAK,
                         "algorithmic_bytes_per_env_step": 74,
                         "note": "4.8 MB per launch: launch-latency bound, not bandwidth bound"}}


def bench_acrobot(args):
    from xenoverse_amd.metacontrol import AcrobotVecEnv, sample_acrobot
    from xenoverse_amd import _lib
    from xenoverse_amd.engine import AUTORESET
    n = 65536
    out = {}
    for fs in (1, 5):
        env = AcrobotVecEnv(n, frameskip=fs, autoreset_mode="same_step", seed=1, max_steps=500)
        env.set_task([sample_acrobot(seed=k) for k in range(1024)])
        env.reset()
        a = torch.randint(0, 3, (n,), device=env.device, dtype=torch.int32)

        def step():
            _lib.check(env.lib.xv_acrobot_step(env._h, _lib.ptr(a), _lib.ptr(env._obs), _lib.ptr(env._reward),
                                               _lib.ptr(env._term), _lib.ptr(env._trunc), _lib.ptr(env._fobs),
                                               AUTORESET["same_step"]))
        out[fs] = timed(step, args.steps, args.warmup)
        if fs == 1:
            T = 64
            aT = torch.randint(0, 3, (T, n), device=env.device, dtype=torch.int32)
            ro = env.rollout(aT)
            us_roll = timed(lambda: env.rollout(aT, out=ro), max(args.steps // T, 5), 2) / T
        env.close()
    algo = (64 + 4 + 24 + 4 + 2 + 24 + 56) * n      # fp64 state r/w, action, obs, reward, flags, final_obs, task params
    us = out[1]
    return {"family": "acrobot", "workload": "65,536 envs, 1,024 tasks, rk4 in fp64", "dtype": "f64",
            "env_steps_per_s": n / (us * 1e-6), "us_per_step": {"frameskip 1": out[1], "frameskip 5": out[5], "rollout_T64 frameskip 1": us_roll},
            "rollout_env_steps_per_s": n / (us_roll * 1e-6),
            "roofline": {"bound": "hbm", "achieved": algo / (us * 1e-6) / 1e9, "peak": HBM_PEAK, "unit": "GB/s",
                         "frac": algo / (us * 1e-6) / 1e9 / HBM_PEAK, "algorithmic_bytes_per_env_step": algo // n,
                         "note": "11 MB per launch; 4 dsdt evaluations (8 fp64 sin/cos) per sub-step: latency / "
                                 "VALU bound, not bandwidth bound"}}


def bench_anymdp_tok(args):
    """multi-token POMDP path (anymdp_env.py:116-128,148-157): d_act transition draws + d_obs observation draws per
    step, per-lane binary searches.  65,536 envs, 1,024 synthetic tasks (S=64, A=8), n_obs=64, d_obs=2, d_act=2."""
    import ctypes as C
    from xenoverse_amd import _lib
    from xenoverse_amd.anymdp import AnyMDPVecEnv, row_lines
    from xenoverse_amd.engine import AUTORESET
    n, n_task, S, A, n_obs, d_obs, d_act = 65536, 1024, 64, 8, 64, 2, 2
    d_obs = int(os.environ.get("XV_TOK_DOBS", d_obs))      # decomposition runs (scripts/runs_r03): other token counts
    d_act = int(os.environ.get("XV_TOK_DACT", d_act))
    mode = os.environ.get("XV_TOK_MODE", "same_step")
    env = AnyMDPVecEnv(n, seed=1, autoreset_mode=mode)
    d = env.device
    tab = dict(S=S, A=A, s0_max=4, rows=torch.empty((n_task, S, A, row_lines(S), 16), dtype=torch.float64, device=d),
               state_map=torch.empty((n_task, S), dtype=torch.int32, device=d),
               term_mask=torch.empty((n_task, 1), dtype=torch.int64, device=d),
               s0_cdf=torch.empty((n_task, 4), dtype=torch.float64, device=d),
               s0_ids=torch.empty((n_task, 4), dtype=torch.int32, device=d),
               max_steps=torch.empty(n_task, dtype=torch.int32, device=d))
    _lib.check(env.lib.xv_anymdp_synth_tasks(env.engine.handle, 7, 0, n_task, S, A, 4, *[_lib.ptr(tab[k]) for k in
               ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")]))
    env.set_task(tab)
    w = torch.rand((n_task, d_obs, S, n_obs), dtype=torch.float64, device=d) + 0.05
    obs_cdf = torch.cumsum(w, -1)
    obs_cdf = (obs_cdf / obs_cdf[..., -1:]).contiguous()
    obs_cdf[..., -1] = 1.0
    _lib.check(env.lib.xv_anymdp_set_observation_model(env._h, n_obs, d_obs, d_act, _lib.ptr(obs_cdf)))
    tobs = torch.zeros((n, d_obs), dtype=torch.int32, device=d)
    tfobs = torch.zeros((n, d_obs), dtype=torch.int32, device=d)
    _lib.check(env.lib.xv_anymdp_reset_tokens(env._h, None, _lib.ptr(tobs)))
    a = torch.randint(0, A, (n, d_act), device=d, dtype=torch.int32)

    def step():
        _lib.check(env.lib.xv_anymdp_step_tokens(env._h, _lib.ptr(a), _lib.ptr(tobs), _lib.ptr(env._reward),
                                                 _lib.ptr(env._reward_gt), _lib.ptr(env._term), _lib.ptr(env._trunc),
                                                 _lib.ptr(tfobs), AUTORESET[mode]))
    env.set_search("fence")                    # the per-lane kernel (binary searches of the rows)
    us = timed(step, args.steps, args.warmup)
    env.set_search("bucket", n_bucket=16)      # the cooperative kernel on transition + observation bucket lines
    us_b = timed(step, args.steps, args.warmup)
    env.close()
    return {"family": "anymdp multi-token POMDP", "workload": "65,536 envs, 1,024 tasks, S=64 A=8 n_obs=64 d_obs=%d d_act=%d%s" % (d_obs, d_act, "" if mode == "same_step" else ", " + mode),
            "dtype": "f64", "env_steps_per_s": n / (min(us, us_b) * 1e-6), "us_per_step": {"binary search": us, "bucket lines": us_b},
            "note": "per-lane searches (general path); 2 transition + 2 observation draws per env-step"}


def bench_anymdp_tok_refdist(args, n_task=1024, per=64):
    """the POMDP / multi-token step on tasks of the REFERENCE's distribution (device sampler: skewed transition rows, sparse
    observation rows), 65,536 envs, token steps issued from C: the per-lane kernel (search = fence) against what AUTO picks
    once the bucket lines are offered (the cooperative kernel on transition + observation cut lines)"""
    from xenoverse_amd.anymdp import AnyMDPVecEnv
    from xenoverse_amd.anymdp import device_sampler as ds
    S, A = 64, 8
    n = n_task * per
    out = {}
    for tt, do, da in (("POMDP", 1, 1), ("MTPOMDP", 2, 2)):
        t = ds.sample_tasks_device(n_task, S, A, seed=3, batch=4096, task_type=tt, observation_space=64, observation_tokens=do,
                                   action_tokens=da)
        env = AnyMDPVecEnv(n, seed=1, autoreset_mode="same_step")
        env.set_task(t, env_task_index=(torch.arange(n, device=env.device, dtype=torch.int32) // per).contiguous())
        env.reset()
        P = 8
        a = torch.randint(0, A, (P, n, da), device=env.device, dtype=torch.int32)
        ring = env.step_tokens_many(P, a)
        k = max(P, args.steps // P * P)
        row = {}
        for name in ("fence", "auto"):
            env.set_search(name, n_bucket=16) if name == "auto" else env.set_search(name)
            us = min(timed(lambda: env.step_tokens_many(k, a, out=ring), 3, 1) / k for _ in range(2))
            row[name] = {"us_per_step": us, "kernel": env.token_kernel, "search": env.effective_search}
        # the same steps with consecutive launches overlapped (two streams, hand-off through the env records: the record
        # leaves before the observation stage) — xv_anymdp_set_step_many_overlap
        env.set_step_many_overlap(True)
        us = min(timed(lambda: env.step_tokens_many(k, a, out=ring), 3, 1) / k for _ in range(2))
        row["auto, overlapped"] = {"us_per_step": us, "kernel": env.token_kernel, "taken": env.step_many_overlap_state == 1}
        env.set_step_many_overlap(False)
        cen = env.bucket_census()
        row["census"] = {k2: cen[k2] for k2 in ("p_fallback", "fallbacks_per_launch", "auto_uses_bucket", "obs_lines", "obs_lines_dirty",
                                               "obs_p_fallback")}
        row["device_error_flags"] = env.check_errors()
        out["%s d_obs=%d d_act=%d" % (tt, do, da)] = row
        env.close()
    return {"family": "anymdp_tok_refdist", "workload": "65,536 envs = 1,024 device-sampled reference-distribution tasks x 64, S=64 A=8 "
            "n_obs=64, token steps from C (xv_anymdp_step_tokens_many)", "dtype": "f64", "variants": out}


def bench_mixed(args, variants=("three streams", "one stream", "one launch")):
    """BASELINE.json config 5, the per-GPU share: 16,384 anymdp (2b: 256 tasks x 64) + 8,192 linds (128 tasks x 64)
    + 8,192 cartpole, one launch per family per vector step, families on separate HIP streams (xenoverse_amd.mixed)
    vs the same launches serialised on one stream."""
    from xenoverse_amd import Engine, _lib
    from xenoverse_amd.anymdp import AnyMDPVecEnv, row_lines
    from xenoverse_amd.engine import AUTORESET
    from xenoverse_amd.linds import LinDSVecEnv, LinearDSSampler
    from xenoverse_amd.metacontrol import CartPoleVecEnv, sample_cartpole
    from xenoverse_amd.mixed import MixedBatch
    na, nl, nc, S, A = 16384, 8192, 8192, 64, 8
    ltasks = linds_tasks(nl // 64)
    ctasks = [sample_cartpole(seed=k) for k in range(1024)]
    res = {}
    # order: the fused launch first — once extra HIP streams exist in the process (the three-streams variant), launches on the
    # default stream carry the legacy stream's implicit synchronisation and the later variants measure that
    for label, mixed in (("one launch", False), ("one stream", False), ("three streams", True)):
        if label not in variants:
            continue
        if mixed:
            mb = MixedBatch("cuda:0", seed=3, streams="separate")
            ea = mb.add("a", AnyMDPVecEnv, na)
            el = mb.add("l", LinDSVecEnv, nl)
            ec = mb.add("c", CartPoleVecEnv, nc, frameskip=1)
        else:
            ea, el, ec = AnyMDPVecEnv(na, seed=3), LinDSVecEnv(nl, seed=3), CartPoleVecEnv(nc, seed=3, frameskip=1)
        d = ea.device
        n_task = na // 64
        tab = dict(S=S, A=A, s0_max=4, rows=torch.empty((n_task, S, A, row_lines(S), 16), dtype=torch.float64, device=d),
                   state_map=torch.empty((n_task, S), dtype=torch.int32, device=d),
                   term_mask=torch.empty((n_task, 1), dtype=torch.int64, device=d),
                   s0_cdf=torch.empty((n_task, 4), dtype=torch.float64, device=d),
                   s0_ids=torch.empty((n_task, 4), dtype=torch.int32, device=d),
                   max_steps=torch.empty(n_task, dtype=torch.int32, device=d))
        _lib.check(ea.lib.xv_anymdp_synth_tasks(ea.engine.handle, 7, 0, n_task, S, A, 4, *[_lib.ptr(tab[k]) for k in
                   ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")]))
        ea.engine.sync()
        if mixed:
            mb.set_task({"a": tab, "l": ltasks, "c": ctasks})
            mb.reset()
        else:
            ea.set_task(tab); el.set_task(ltasks); ec.set_task(ctasks)
            ea.reset(); el.reset(); ec.reset()
        torch.cuda.synchronize()
        aa = torch.randint(0, A, (na,), device=d, dtype=torch.int32)
        al = torch.rand((nl, 8), device=d) * 2 - 1
        ac = torch.randint(0, 2, (nc,), device=d, dtype=torch.int32)
        mode = AUTORESET["same_step"]

        def launch_all():
            _lib.check(ea.lib.xv_anymdp_step(ea._h, _lib.ptr(aa), _lib.ptr(ea._obs), _lib.ptr(ea._reward), _lib.ptr(ea._reward_gt),
                                             _lib.ptr(ea._term), _lib.ptr(ea._trunc), _lib.ptr(ea._final_obs), mode))
            _lib.check(el.lib.xv_linds_step(el._h, _lib.ptr(al), _lib.ptr(el._obs), _lib.ptr(el._reward), _lib.ptr(el._term),
                                            _lib.ptr(el._trunc), _lib.ptr(el._cmd), _lib.ptr(el._error), _lib.ptr(el._fobs), mode))
            _lib.check(ec.lib.xv_cartpole_step(ec._h, _lib.ptr(ac), _lib.ptr(ec._obs), _lib.ptr(ec._reward), _lib.ptr(ec._term),
                                               _lib.ptr(ec._trunc), _lib.ptr(ec._fobs), mode))

        def step():
            launch_all()
            if mixed:      # the caller's stream waits for the three family streams, and they wait for it (next actions)
                mb.sync()
                for st in mb.streams.values():
                    st.wait_stream(torch.cuda.current_stream())
        if label == "one launch":      # xv_mixed_step_many: the three families' step bodies in ONE grid, launches issued from C
            import ctypes as C
            from xenoverse_amd.mixed import _MixedIO
            P = 32
            ring = dict(aa=torch.randint(0, A, (P, na), device=d, dtype=torch.int32), ao=torch.zeros((P, na), device=d, dtype=torch.int32),
                        ar=torch.zeros((P, na), device=d), ag=torch.zeros((P, na), device=d),
                        at=torch.zeros((P, na), device=d, dtype=torch.uint8), au=torch.zeros((P, na), device=d, dtype=torch.uint8),
                        af=torch.zeros((P, na), device=d, dtype=torch.int32),
                        la=torch.rand((P, nl, 8), device=d) * 2 - 1, lo=torch.zeros((P, nl, 16), device=d), lr=torch.zeros((P, nl), device=d),
                        lt=torch.zeros((P, nl), device=d, dtype=torch.uint8), lu=torch.zeros((P, nl), device=d, dtype=torch.uint8),
                        lc=torch.zeros((P, nl, 16), device=d), le=torch.zeros((P, nl), device=d), lf=torch.zeros((P, nl, 16), device=d),
                        ca=torch.randint(0, 2, (P, nc), device=d, dtype=torch.int32), co=torch.zeros((P, nc, 4), device=d),
                        cr=torch.zeros((P, nc), device=d), ct=torch.zeros((P, nc), device=d, dtype=torch.uint8),
                        cu=torch.zeros((P, nc), device=d, dtype=torch.uint8), cf=torch.zeros((P, nc, 4), device=d))
            io = _MixedIO(*[_lib.ptr(ring[k]) for k in ("aa", "ao", "ar", "ag", "at", "au", "af", "la", "lo", "lr", "lt", "lu", "lc",
                                                         "le", "lf", "ca", "co", "cr", "ct", "cu", "cf")])
            k = max(P, args.steps // P * P)

            def many():
                _lib.check(ea.lib.xv_mixed_step_many(ea._h, el._h, ec._h, C.byref(io), k, P, mode))
            res[label] = timed(many, 3, 1) / k
            # the same calls with consecutive steps alternating between two streams (overlapped xv_mixed_step_many)
            ea.set_step_many_overlap(True)
            t_ov = timed(many, 3, 1) / k
            if int(ea.lib.xv_mixed_step_many_overlap_state(ea._h)) == 1:
                res[label + ", overlapped"] = t_ov
            ea.set_step_many_overlap(False)
        else:
            res[label] = timed(step, args.steps, args.warmup)
        (mb.close() if mixed else [e.close() for e in (ea, el, ec)])
    n = na + nl + nc
    best = min(res.values())
    algo = 562 * na + 432 * nl + 74 * nc
    return {"family": "mixed (config 5, per-GPU share)", "workload": "16,384 anymdp(2b) + 8,192 linds(32,8,8) + 8,192 cartpole",
            "env_steps_per_s": n / (best * 1e-6), "us_per_vector_step": res, "dtype": "f64/f32",
            "roofline": {"bound": "hbm", "achieved": algo / (best * 1e-6) / 1e9, "peak": HBM_PEAK, "unit": "GB/s",
                         "frac": algo / (best * 1e-6) / 1e9 / HBM_PEAK, "algorithmic_bytes_per_vector_step": algo,
                         "note": "a 13 MB vector step: launch-latency bound; one fused launch instead of three; overlapped: "
                                 "consecutive launches on two streams, per-wave hand-off (xv_mixed_step_many with the overlap on)"}}


def _synth_anymdp_env(n, n_task, copy=True, seed=1234):
    """65,536-env style AnyMDP batch on synthetic tasks generated on the device (as bench.py's headline, shared tasks)"""
    from xenoverse_amd import _lib
    from xenoverse_amd.anymdp import AnyMDPVecEnv, row_lines
    S, A, s0_max = 64, 8, 4
    env = AnyMDPVecEnv(n, seed=seed, autoreset_mode="same_step", copy=copy)
    d = env.device
    t = dict(S=S, A=A, s0_max=s0_max,
             rows=torch.empty((n_task, S, A, row_lines(S), 16), dtype=torch.float64, device=d),
             state_map=torch.empty((n_task, S), dtype=torch.int32, device=d),
             term_mask=torch.empty((n_task, 1), dtype=torch.int64, device=d),
             s0_cdf=torch.empty((n_task, s0_max), dtype=torch.float64, device=d),
             s0_ids=torch.empty((n_task, s0_max), dtype=torch.int32, device=d),
             max_steps=torch.empty(n_task, dtype=torch.int32, device=d))
    _lib.check(env.lib.xv_anymdp_synth_tasks(env.engine.handle, seed + 1, 0, n_task, S, A, s0_max,
                                             *[_lib.ptr(t[k]) for k in ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")]))
    env.set_task(t, env_task_index=(torch.arange(n, device=d, dtype=torch.int32) // (n // n_task)).contiguous())
    return env


def bench_anymdp_refdist(args, n_task=1024, per=64):
    """the AnyMDP step on tasks of the REFERENCE's distribution (task_sampler_utils.py:65-175: banded rows with a few large
    probabilities among many tiny ones, sampled on the device by xv_anymdp_sample_tasks) instead of the survey's synthetic
    dense bands: 65,536 envs = 1,024 tasks x 64; what AUTO picks, and each search timed"""
    from xenoverse_amd.anymdp import AnyMDPVecEnv
    from xenoverse_amd.anymdp import device_sampler as ds
    S, A = 64, 8
    n = n_task * per
    t0 = time.time()
    t = ds.sample_tasks_device(n_task, S, A, seed=3, batch=4096)
    env = AnyMDPVecEnv(n, seed=1, autoreset_mode="same_step")
    env.set_task({k: t[k] for k in ("S", "A", "s0_max", "rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")},
                 env_task_index=(torch.arange(n, device=env.device, dtype=torch.int32) // per).contiguous())
    env.reset()
    P = 32
    acts = torch.randint(0, A, (P, n), device=env.device, dtype=torch.int32)
    ring = env.step_many(P, acts)
    env.set_search("auto", n_bucket=16)
    auto = env.effective_search
    cen = env.bucket_census() if auto == "bucket" else env.probe_buckets(16)
    k = max(P, args.steps // P * P)
    out = {}
    for name in ("auto", "fence", "bucket"):
        env.set_search(name, n_bucket=16) if name != "fence" else env.set_search("fence")
        out[name] = min(timed(lambda: env.step_many(k, acts, out=ring), 3, 1) / k for _ in range(2))
    # the AUTO search with consecutive launches overlapped (xv_anymdp_set_step_many_overlap; calls of >= 64 steps)
    env.set_search("auto", n_bucket=16)
    env.set_step_many_graph("on")
    env.set_step_many_overlap(True)
    out["auto, overlapped"] = min(timed(lambda: env.step_many(k, acts, out=ring), 3, 1) / k for _ in range(2))
    overlapped = env.step_many_overlap_state == 1
    errs = env.check_errors()
    env.close()
    us = min(out["auto"], out["auto, overlapped"])
    algo = (8 * S + 50) * n
    return {"family": "anymdp_refdist", "workload": "anymdp S=64 A=8, 65,536 envs = 1,024 device-sampled reference-distribution "
            "tasks x 64 (xv_anymdp_sample_tasks), launches from C", "dtype": "f64", "auto_search": auto,
            "us_per_step": out, "overlap_taken": overlapped, "env_steps_per_s": n / (us * 1e-6),
            "bucket_census": {k2: cen[k2] for k2 in ("n_bucket", "format", "cuts_per_line", "lines", "lines_dirty", "p_fallback",
                                                    "fallbacks_per_launch", "auto_limit", "auto_uses_bucket")},
            # 1,024 shared tasks: the rows are Infinity-Cache / L2 resident.  `achieved` prices SURVEY 8(d)'s 562 B per env-step as
            # if they were HBM reads, which they are not: no fraction of the HBM peak is claimed (a fraction above 1 is a wrong
            # bound, not a fast kernel)
            "roofline": {"bound": "cache", "achieved": algo / (us * 1e-6) / 1e9, "peak": None, "unit": "GB/s", "frac": None,
                         "algorithmic_bytes_per_env_step": 8 * S + 50,
                         "frac_survey_bytes_of_hbm_peak": algo / (us * 1e-6) / 1e9 / HBM_PEAK,
                         "note": "1,024 shared tasks: the rows are Infinity-Cache resident; the 562-byte count would price them as HBM "
                                 "reads (frac_survey_bytes_of_hbm_peak, above 1), so `frac` is null"},
            "device_error_flags": errs, "setup_s": round(time.time() - t0, 1)}


def bench_python_loop(args, n=65536, n_task=1024):
    """the loop a training script runs — actions = policy(obs) on the device, env.step(actions) — per vector step of 65,536
    AnyMDP envs: issued from Python call by call (copy=True and copy=False), and replayed from a captured graph
    (xenoverse_amd/capture.py: device tick, one hipGraphLaunch per `unroll` iterations)"""
    def policy(obs):
        return torch.bitwise_and(obs, 7)      # ONE elementwise op: a graph node of its own, like any policy kernel

    out = {}
    steps = max(200, args.steps)
    for copy in (True, False):
        env = _synth_anymdp_env(n, n_task, copy=copy)
        obs, _ = env.reset()
        st = {"obs": obs}

        def it():
            st["obs"] = env.step(policy(st["obs"]))[0]
        out["eager copy=%s" % copy] = timed(it, steps, 20)
        if not copy:      # without the infos that need a launch of their own (`steps`, the `_final_obs` mask)
            env.lean_infos = True
            out["eager copy=False lean_infos"] = timed(it, steps, 20)
        env.close()
    for unroll in (1, 8, 32):
        env = _synth_anymdp_env(n, n_task, copy=False)
        obs, _ = env.reset()
        loop = env.capture(policy, obs, unroll=unroll, warmup=2)
        k = max(1, steps // unroll)
        out["captured unroll=%d" % unroll] = min(timed(lambda: loop.replay(k), 3, 1) / (k * unroll) for _ in range(2))
        errs = env.check_errors()
        loop.close()
        env.close()
    best = min(v for k2, v in out.items() if k2.startswith("captured"))
    more = {}
    for name, fn in (("linds", _python_loop_linds), ("mixed", _python_loop_mixed)):
        try:
            more[name] = fn(steps)
        except Exception as ex:      # never lose the AnyMDP figures to a sibling
            more[name] = {"error": repr(ex)}
    return {"family": "python_loop", "workload": "anymdp S=64 A=8, 65,536 envs over 1,024 synthetic tasks; policy = one "
            "elementwise torch op on the device; closed loop policy -> step", "dtype": "f64", "us_per_vector_step": out,
            "env_steps_per_s": n / (best * 1e-6), "device_error_flags": errs, "other_families": more}


def _python_loop_linds(steps, n=65536):
    """the same closed loop for config 3 (LinDS ns = 32, 65,536 envs): policy = one elementwise op on the observation"""
    from xenoverse_amd.linds import LinDSVecEnv
    tasks = linds_tasks(n // 64)

    def policy(obs):
        return torch.tanh(obs[:, :8])      # action_dim 8; the observation comes padded to 16 columns (linds_env.py:83-91)
    out = {}
    for copy in (True, False):
        env = LinDSVecEnv(n, autoreset_mode="same_step", seed=1, copy=copy)
        env.set_task(tasks)
        obs, _ = env.reset()
        st = {"obs": obs}

        def it():
            st["obs"] = env.step(policy(st["obs"]))[0]
        out["eager copy=%s" % copy] = timed(it, steps, 20)
        env.close()
    env = LinDSVecEnv(n, autoreset_mode="same_step", seed=1, copy=False)
    env.set_task(tasks)
    obs, _ = env.reset()
    loop = env.capture(policy, obs, unroll=8, warmup=2)
    k = max(1, steps // 8)
    out["captured unroll=8"] = min(timed(lambda: loop.replay(k), 3, 1) / (k * 8) for _ in range(2))
    errs = env.check_errors()
    loop.close()
    env.close()
    return {"workload": "linds ns=32 na=8 no=8, 65,536 envs = 1,024 tasks x 64; policy = tanh(obs)", "us_per_vector_step": out,
            "env_steps_per_s": n / (min(out.values()) * 1e-6), "device_error_flags": errs}


def _python_loop_mixed(steps):
    """the same closed loop for config 5's per-GPU share (16,384 anymdp + 8,192 linds + 8,192 cartpole): MixedBatch.step_fused
    (one launch for the three families) behind three elementwise policy ops"""
    from xenoverse_amd import _lib
    from xenoverse_amd.anymdp import AnyMDPVecEnv, row_lines
    from xenoverse_amd.linds import LinDSVecEnv
    from xenoverse_amd.metacontrol import CartPoleVecEnv, sample_cartpole
    from xenoverse_amd.mixed import MixedBatch
    na, nl, nc, S, A = 16384, 8192, 8192, 64, 8

    def build(copy):
        mb = MixedBatch("cuda:0", seed=3, streams="shared")
        ea = mb.add("a", AnyMDPVecEnv, na, copy=copy)
        mb.add("l", LinDSVecEnv, nl, copy=copy)
        mb.add("c", CartPoleVecEnv, nc, frameskip=1, copy=copy)
        d = ea.device
        n_task = na // 64
        tab = dict(S=S, A=A, s0_max=4, rows=torch.empty((n_task, S, A, row_lines(S), 16), dtype=torch.float64, device=d),
                   state_map=torch.empty((n_task, S), dtype=torch.int32, device=d),
                   term_mask=torch.empty((n_task, 1), dtype=torch.int64, device=d),
                   s0_cdf=torch.empty((n_task, 4), dtype=torch.float64, device=d),
                   s0_ids=torch.empty((n_task, 4), dtype=torch.int32, device=d),
                   max_steps=torch.empty(n_task, dtype=torch.int32, device=d))
        _lib.check(ea.lib.xv_anymdp_synth_tasks(ea.engine.handle, 7, 0, n_task, S, A, 4, *[_lib.ptr(tab[k]) for k in
                   ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")]))
        ea.engine.sync()
        mb.set_task({"a": tab, "l": linds_tasks(nl // 64), "c": [sample_cartpole(seed=k) for k in range(1024)]})
        r = mb.reset()
        return mb, {k: v[0] for k, v in r.items()}

    def policy(obs):
        return {"a": torch.bitwise_and(obs["a"], 7), "l": torch.tanh(obs["l"][:, :8]), "c": (obs["c"][:, 2] > 0).to(torch.int32)}
    out = {}
    for copy in (True, False):
        mb, obs = build(copy)
        st = {"obs": obs}

        def it():
            r = mb.step_fused(policy(st["obs"]))
            st["obs"] = {k: v[0] for k, v in r.items()}
        out["eager step_fused copy=%s" % copy] = timed(it, steps, 20)
        mb.close()
    mb, obs = build(False)
    loop = mb.capture(policy, obs, unroll=8, warmup=2)
    k = max(1, steps // 8)
    out["captured unroll=8"] = min(timed(lambda: loop.replay(k), 3, 1) / (k * 8) for _ in range(2))
    errs = 0
    for e in mb.envs.values():
        errs |= e.check_errors()
    loop.close()
    mb.close()
    n = na + nl + nc
    return {"workload": "config 5's per-GPU share: 16,384 anymdp + 8,192 linds + 8,192 cartpole, MixedBatch.step_fused (one launch); "
            "policy = three elementwise ops", "us_per_vector_step": out, "env_steps_per_s": n / (min(out.values()) * 1e-6),
            "device_error_flags": errs}


def _maze_valu_roofline(r):
    v = r.get("valu_issue")
    if not v:
        return {"bound": "valu_issue", "frac": None, "achieved": None, "peak": None, "unit": "G wave-instr/s",
                "model": "no committed counter profile of this ray caster: no fraction claimed"}
    return {"bound": "valu_issue", "frac": v["frac"], "frac_at_measured_clock": v["frac_at_measured_clock"], "achieved": v["achieved"],
            "peak": v["peak"], "unit": v["unit"], "model": v["model"], "valu_instructions_per_pixel": v["valu_instructions_per_pixel"],
            "source": v["source"]}


def quick_families(steps=200, warmup=20):
    """the `families` object of bench.py's JSON line: BASELINE.json configs 3 (linds), 4 (mazeworld, 64x64 frames) and the
    per-GPU share of config 5 (mixed), each timed with HIP events over back-to-back launches; a few seconds in all"""
    a = argparse.Namespace(steps=steps, warmup=warmup)
    out = {}

    def guard(name, fn):
        t0 = time.time()
        try:
            r = fn()
        except Exception as ex:      # a family must never cost the headline line
            r = {"error": repr(ex)}
        r["wall_s"] = round(time.time() - t0, 1)
        out[name] = r

    def linds():
        r = bench_linds(a, paths=("mfma",), rollout=False)
        us = min(r["us_per_step"].values())
        return {"config": "BASELINE configs[2]: " + r["workload"], "ms_per_step": us * 1e-3, "env_steps_per_s": r["env_steps_per_s"],
                "us_per_step": r["us_per_step"], "dtype": "f32",
                "roofline": {"bound": "hbm", "frac": r["roofline"]["frac"], "algorithmic_bytes": 432 * 65536, "peak": HBM_PEAK,
                             "unit": "GB/s", "achieved": r["roofline"]["achieved"], "kernel": r["kernel"],
                             "mfma_frac_of_157_TF": r["achieved_tflops"] / 157.3}}

    def maze():
        r = bench_maze(a, 64)
        us = r["us_per_step"]["step (both)"]
        return {"config": "BASELINE configs[3]: " + r["workload"], "ms_per_step": us * 1e-3, "env_steps_per_s": r["env_steps_per_s"],
                "us_per_step": r["us_per_step"], "dtype": r["dtype"],
                # the ray caster (90 % of the step) is bound by fp64 VALU issue and texture requests — the 16-tap filter with the
                # reference's bytes — not by HBM: `bound` / `frac` are that kernel's; the HBM view of the same step is kept beside it
                "roofline": dict(_maze_valu_roofline(r),
                                 kernel="maze_raycast_kernel (exact filter, speculated with float32 colour sums, rows of a column per wave; + maze_step9_kernel for the move)",
                                 hbm={"frac": r["roofline"]["frac"], "algorithmic_bytes": (3 * 64 * 64 + 64) * 16384, "peak": HBM_PEAK,
                                      "unit": "GB/s", "achieved": r["roofline"]["achieved"]},
                                 note=r["roofline"]["note"])}

    def maze256():      # the registered default resolution (mazeworld/__init__.py:27): 196,672 B per env-step
        r = bench_maze(a, 256)
        us = r["us_per_step"]["step (both)"]
        return {"config": "BASELINE configs[3] at the registered 256 x 256: " + r["workload"], "ms_per_step": us * 1e-3,
                "env_steps_per_s": r["env_steps_per_s"], "us_per_step": r["us_per_step"], "dtype": r["dtype"],
                "roofline": dict(_maze_valu_roofline(r),
                                 kernel="maze_raycast_kernel (exact filter, speculated with float32 colour sums, rows of a column per wave; + maze_step9_kernel)",
                                 hbm={"frac": r["roofline"]["frac"], "algorithmic_bytes": (3 * 256 * 256 + 64) * 16384,
                                      "peak": HBM_PEAK, "unit": "GB/s", "achieved": r["roofline"]["achieved"]},
                                 note=r["roofline"]["note"])}

    def mixed():
        r = bench_mixed(a, variants=("one stream", "one launch"))
        us = min(r["us_per_vector_step"].values())
        return {"config": "BASELINE configs[4], one GPU's share: " + r["workload"], "ms_per_step": us * 1e-3,
                "env_steps_per_s": r["env_steps_per_s"], "dtype": r["dtype"],
                "roofline": {"bound": "hbm", "frac": r["roofline"]["frac"], "algorithmic_bytes": r["roofline"]["algorithmic_bytes_per_vector_step"],
                             "peak": HBM_PEAK, "unit": "GB/s", "achieved": r["roofline"]["achieved"],
                             "kernel": "mixed_step_kernel (anymdp + linds + cartpole step bodies in one grid, xv_mixed_step_many)",
                             "us_per_vector_step": r["us_per_vector_step"], "note": r["roofline"]["note"]}}
    def refdist():
        r = bench_anymdp_refdist(a)
        return {"config": r["workload"], "ms_per_step": min(r["us_per_step"]["auto"], r["us_per_step"]["auto, overlapped"]) * 1e-3,
                "env_steps_per_s": r["env_steps_per_s"],
                "auto_search": r["auto_search"], "us_per_step": r["us_per_step"], "bucket_census": r["bucket_census"],
                "dtype": "f64", "roofline": r["roofline"], "device_error_flags": r["device_error_flags"]}

    def pyloop():
        r = bench_python_loop(a)
        us = min(v for k2, v in r["us_per_vector_step"].items() if k2.startswith("captured"))
        return {"config": r["workload"], "ms_per_step": us * 1e-3, "env_steps_per_s": r["env_steps_per_s"],
                "us_per_vector_step": r["us_per_vector_step"], "dtype": "f64", "device_error_flags": r["device_error_flags"],
                "other_families": r.get("other_families")}
    def tok():      # the POMDP / multi-token step on reference-distribution tasks (cooperative kernel on AUTO)
        r = bench_anymdp_tok_refdist(a)
        v = r["variants"]
        us = {k: {"auto": v[k]["auto"]["us_per_step"], "auto, overlapped": v[k]["auto, overlapped"]["us_per_step"],
                  "per-lane (fence search)": v[k]["fence"]["us_per_step"], "kernel": v[k]["auto"]["kernel"]} for k in v}
        one = min(v["POMDP d_obs=1 d_act=1"]["auto"]["us_per_step"], v["POMDP d_obs=1 d_act=1"]["auto, overlapped"]["us_per_step"])
        # priced in random 128-byte lines: (1, 1) reads 2 dependent lines per env-step, (2, 2) three stages of 4.4 lines
        return {"config": r["workload"], "ms_per_step": one * 1e-3, "env_steps_per_s": 65536 / (one * 1e-6), "us_per_step": us,
                "dtype": "f64", "note": "ms_per_step / env_steps_per_s: the single-token POMDP; (2, 2) beside it in us_per_step",
                "device_error_flags": max(v[k]["device_error_flags"] for k in v)}
    guard("linds", linds)
    guard("mazeworld_64", maze)
    guard("mazeworld_256", maze256)
    guard("mixed_share", mixed)
    guard("anymdp_refdist", refdist)
    guard("anymdp_tok_refdist", tok)
    guard("python_loop", pyloop)
    return out



_MAZE_TASKS = {}


def maze_tasks(n_task):
    """BASELINE config 4's mazes (15 x 15, seeds 0 .. n_task - 1), sampled once per process (11 ms each)"""
    from xenoverse_amd.mazeworld import MazeTaskSampler
    if n_task not in _MAZE_TASKS:
        _MAZE_TASKS[n_task] = [MazeTaskSampler(n_range=(15, 16), seed=k, n_wall_textures=8, n_ground_textures=4,
                                               n_ceiling_textures=4) for k in range(n_task)]
    return _MAZE_TASKS[n_task]


def bench_maze(args, res, precision="exact", move_kernel="auto"):
    from xenoverse_amd.mazeworld import MazeWorldVecEnv, make_texture_library
    from xenoverse_amd import _lib
    from xenoverse_amd.engine import AUTORESET
    n_task, per = 256, 64
    n = n_task * per
    tasks = maze_tasks(n_task)
    env = MazeWorldVecEnv(n, resolution=(res, res), textures=make_texture_library(8, 4, 4, seed=0),
                          autoreset_mode="same_step", action_space_type="Discrete16", precision=precision)
    env.set_task(tasks)
    env.set_move_kernel(move_kernel)
    if os.environ.get("XV_MAZE_MAPPING"):      # A/B of the ray caster's lane -> pixel mappings (scripts/runs_r05/gpu_m.sh)
        env.set_raycast_mapping(os.environ["XV_MAZE_MAPPING"])
    env.reset()
    # BASELINE config 4: actions uniform over Discrete16, drawn anew for every step (a ring of 64 pre-generated action sets);
    # 32 untimed steps first, so that the batch is past the reset transient (every agent at its start cell's centre) —
    # what a frame costs depends on what the agents see.  Enough timed launches for the GPU's clock to settle: 5 steps
    # (7 ms) read 1.38 or 1.54 ms per step depending on what ran (or idled) just before.
    ring = torch.randint(0, 16, (64, n), device=env.device, dtype=torch.int32)
    clock = [0]

    def action_ptr():
        clock[0] += 1
        return _lib.ptr(ring[clock[0] % 64])
    steps = max(24 if res <= 64 else 10, args.steps // (40 if res <= 64 else 400))
    steps = int(os.environ.get("XV_MAZE_STEPS", steps))

    def move():
        _lib.check(env.lib.xv_maze_step(env._h, action_ptr(), 1, None, _lib.ptr(env._reward), _lib.ptr(env._term),
                                        _lib.ptr(env._trunc), None, None, AUTORESET["same_step"]))

    def render():
        _lib.check(env.lib.xv_maze_render(env._h, _lib.ptr(env._frames), _lib.ptr(env._cmd_rgb)))

    def full():
        _lib.check(env.lib.xv_maze_step(env._h, action_ptr(), 1, _lib.ptr(env._frames), _lib.ptr(env._reward),
                                        _lib.ptr(env._term), _lib.ptr(env._trunc), _lib.ptr(env._cmd_rgb), None,
                                        AUTORESET["same_step"]))
    for _ in range(32 if res <= 64 else 8):
        move()
    us_move = timed(move, steps, 4)
    us_render = timed(render, steps, 4)
    us_full = timed(full, steps, 4)
    env.close()
    algo = (3 * res * res + 64) * n
    # The ray caster is bound by VALU ISSUE (every class of instruction; gfx950 issues v_fma_f64 at the rate of v_fma_f32), not by
    # HBM and not by the fp64 pipe.  Instructions per pixel come from the committed counters of this filter at this
    # resolution (rocprofv3 --pmc SQ_INSTS_VALU / SQ_WAVES: profiles/*pmc_raycast_<filter>_<res>.json, scripts/pmc_kernel.sh);
    # peak = 1,024 SIMDs x one wave instruction per 4 cycles at the 2.4-GHz peak clock (the counters' own GRBM_GUI_ACTIVE gives
    # 2.25 GHz during the kernel: `frac_at_measured_clock`)
    ipp, ipp_src = maze_valu_per_pixel("spec32" if precision == "exact" else ("f32" if precision == "f32" else None), res, n)
    valu_peak = 1024 * 2.4e9 / 4
    valu = None
    if ipp is not None:
        wave_instr = ipp * res * res * n / 64.0
        valu = {"achieved": wave_instr / (us_render * 1e-6) / 1e9, "peak": valu_peak / 1e9, "unit": "G wave-instr/s",
                "frac": wave_instr / (us_render * 1e-6) / valu_peak, "frac_at_measured_clock": wave_instr / (us_render * 1e-6) / (1024 * 2.25e9 / 4),
                "valu_instructions_per_pixel": ipp, "source": ipp_src,
                "model": "SQ_INSTS_VALU per pixel (committed counters of this kernel) x pixels / 64 lanes / ray-cast time, against "
                         "1,024 SIMDs x 1 wave instruction / 4 cycles x 2.4 GHz"}
    return {"family": "mazeworld", "filter": precision, "move_kernel": move_kernel,
            "workload": "15x15 mazes, 16,384 envs = 256 tasks x 64, %dx%d frames" % (res, res),
            "dtype": "f64 pose, f32/f64 ray-caster, u8 frames", "env_steps_per_s": n / (us_full * 1e-6),
            "us_per_step": {"move+rules": us_move, "raycast": us_render, "step (both)": us_full},
            "roofline": {"bound": "hbm", "achieved": algo / (us_full * 1e-6) / 1e9, "peak": HBM_PEAK, "unit": "GB/s",
                         "frac": algo / (us_full * 1e-6) / 1e9 / HBM_PEAK,
                         "algorithmic_bytes_per_env_step": 3 * res * res + 64,
                         "note": "VALU issue (every instruction class), not HBM, bounds the ray caster: see valu_issue"},
            "valu_issue": valu}


def maze_valu_per_pixel(tag, res, n):
    """-> (VALU instructions per pixel, file) from the newest committed counter profile of the ray caster with this filter at this
    resolution (profiles/*pmc_raycast_<tag>_<res>.json), or (None, None)"""
    import glob
    if tag is None:
        return None, None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def visit_order(path):      # r06_m < r06_zz5 < r06_zz28: numbers compare as numbers (a plain sort puts zz5 behind zz28)
        import re
        return [int(t) if t.isdigit() else t for t in re.findall(r"\d+|\D+", os.path.basename(path))]
    for f in sorted(glob.glob(os.path.join(root, "profiles", "*pmc_raycast_%s_%d.json" % (tag, res))), key=visit_order, reverse=True):
        try:
            d = json.load(open(f))
            for name, v in d["kernels"].items():
                if "maze_raycast" in name and "SQ_INSTS_VALU" in v:
                    return float(v["SQ_INSTS_VALU"]) * 64.0 / (float(res) * res * n), os.path.basename(f)
        except Exception:
            pass
    return None, None


def bench_teacher(args, res=64):
    """the maze teachers deciding for 16,384 envs on the device, and the teacher-driven roll-out"""
    from xenoverse_amd.mazeworld import (MazeTaskSampler, MazeWorldVecEnv, OracleAgent, SmartSLAMAgent,
                                         make_texture_library)
    from xenoverse_amd import _lib
    from xenoverse_amd.engine import AUTORESET
    n_task, per = 256, 64
    n = n_task * per
    tasks = [MazeTaskSampler(n_range=(15, 16), seed=k, n_wall_textures=8, n_ground_textures=4, n_ceiling_textures=4)
             for k in range(n_task)]
    out = {}
    for name, cls in (("smart_slam", SmartSLAMAgent), ("oracle", OracleAgent)):
        env = MazeWorldVecEnv(n, resolution=(res, res), textures=make_texture_library(8, 4, 4, seed=0),
                              autoreset_mode="same_step", action_space_type="Discrete16", max_steps=2000)
        env.set_task(tasks)
        env.reset()
        agent = cls(maze_env=env)
        act = agent._action

        def decide():
            _lib.check(env.lib.xv_maze_agent_act(agent._h, None, _lib.ptr(act)))

        def full():
            decide()
            _lib.check(env.lib.xv_maze_step(env._h, _lib.ptr(act), 1, _lib.ptr(env._frames), _lib.ptr(env._reward),
                                            _lib.ptr(env._term), _lib.ptr(env._trunc), _lib.ptr(env._cmd_rgb), None,
                                            AUTORESET["same_step"]))
        for _ in range(60):        # let the memories fill before timing
            full()
        us_dec = timed(decide, 20, 2)
        us_full = timed(full, 20, 2)
        out[name] = {"decide_us": us_dec, "decide+step_us": us_full, "env_steps_per_s": n / (us_full * 1e-6)}
        agent.close(); env.close()
    return {"family": "mazeworld teachers", "workload": "15x15 mazes, 16,384 envs = 256 tasks x 64, %dx%d frames, "
            "one decision per env-step" % (res, res), "dtype": "f64 cost maps, bitmap memories", "agents": out}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--families", default="linds,cartpole,acrobot,maze64,maze256")
    args = ap.parse_args()
    for f in args.families.split(","):
        if f == "linds":
            r = bench_linds(args)
        elif f == "linds_sweep":
            r = bench_linds_sweep(args)
        elif f == "linds_mfma":
            r = bench_linds(args, paths=("mfma",), rollout=False)
        elif f == "cartpole":
            r = bench_cartpole(args)
        elif f == "acrobot":
            r = bench_acrobot(args)
        elif f == "mixed":
            r = bench_mixed(args)
        elif f == "anymdp_tok":
            r = bench_anymdp_tok(args)
        elif f == "anymdp_refdist":
            r = bench_anymdp_refdist(args)
        elif f == "anymdp_tok_refdist":
            r = bench_anymdp_tok_refdist(args)
        elif f == "python_loop":
            r = bench_python_loop(args)
        elif f == "teacher":
            r = bench_teacher(args)
        elif f.startswith("maze"):
            tok = f[4:].split("_")
            mv = {"m1": "lane_per_env", "m3": "three_lanes", "m9": "nine_lanes"}
            r = bench_maze(args, int(tok[0]), "f32" if "f32" in tok else ("exact_direct" if "direct" in tok else "exact"),
                           next((mv[t] for t in tok if t in mv), "auto"))
        else:
            continue
        print(json.dumps(r), flush=True)
