"""oracle — CPU restatement of the Xenoverse env-step hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.  The product
(xenoverse_amd) never imports it and has no CPU fallback: without libxeno_hip.so it raises.

Contents
  xeno_oracle.c/.h   plain-C scalar restatement of each step function, citing the reference file:line
  Makefile           builds oracle/_build/libxeno_oracle.so (gcc, -ffp-contract=off)
  gen_golden.py      imports /root/reference (build container only) and writes tests/golden/*.npz
  stubs/             stand-ins for numba/gymnasium/gym/pygame so that the Python reference imports here

Pinning: tests/test_oracle_*.py check this restatement against every fixture under tests/golden/, which
were produced by running the reference itself (see gen_golden.py).  There is no pure-CPU product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libxeno_oracle.so")
_lib = None


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("xeno_oracle.c", "xeno_oracle_sampler.c", "xeno_oracle.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.xo_u53.restype = C.c_double
        _lib.xo_upper_bound.restype = C.c_int
        _lib.xo_max_threads.restype = C.c_int
        _lib.xo_np_pairwise_sum.restype = C.c_double
        _lib.xo_update_value_matrix.restype = C.c_int
    return _lib


def _p(a, t=None):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def philox4x32_10(ctr, key):
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32).reshape(-1, 4)
    key = np.ascontiguousarray(key, dtype=np.uint32).reshape(2)
    out = np.empty_like(ctr)
    for i in range(ctr.shape[0]):
        lib().xo_philox4x32_10(_p(ctr[i:i + 1]), _p(key), _p(out[i:i + 1]))
    return out


def u53(a, b):
    return lib().xo_u53(C.c_uint32(int(a)), C.c_uint32(int(b)))


def env_draw_sub(seed, gid, tick, purpose, sub):
    out = np.empty(4, dtype=np.uint32)
    lib().xo_env_draw_sub(C.c_uint64(int(seed)), C.c_uint64(int(gid)), C.c_uint64(int(tick)), C.c_uint32(int(purpose)),
                          C.c_uint32(int(sub)), _p(out))
    return out


def env_draw(seed, gid, tick, purpose):
    out = np.empty(4, dtype=np.uint32)
    lib().xo_env_draw(C.c_uint64(seed), C.c_uint64(gid), C.c_uint64(tick), C.c_uint32(purpose), _p(out))
    return out


class _AnyMDPStruct(C.Structure):
    _fields_ = [("n_env", C.c_int), ("n_task", C.c_int), ("S", C.c_int), ("A", C.c_int),
                ("s0_max", C.c_int),
                ("cdf", C.c_void_p), ("rs", C.c_void_p), ("state_map", C.c_void_p),
                ("term_mask", C.c_void_p), ("s0_cdf", C.c_void_p), ("s0_ids", C.c_void_p),
                ("max_steps", C.c_void_p), ("env_task", C.c_void_p),
                ("state", C.c_void_p), ("steps", C.c_void_p), ("need_reset", C.c_void_p),
                ("err_flags", C.c_uint32), ("gid_stride", C.c_uint32)]


class AnyMDPOracle(object):
    """Batched CPU AnyMDP over the device table layout (include/xeno.h).  `tables` is the dict made by
    xenoverse_amd.anymdp.tables.build_tables (numpy arrays)."""

    def __init__(self, tables, env_task, gid_stride=1):
        """gid_stride: free-running draws of env i use global env id gid_base + i * gid_stride (a scattered subset of a
        larger device batch)"""
        t = tables
        self.S, self.A, self.s0_max = int(t["S"]), int(t["A"]), int(t["s0_max"])
        self.n_task = int(t["cdf"].shape[0])
        self.env_task = np.ascontiguousarray(env_task, dtype=np.int32)
        self.n_env = int(self.env_task.shape[0])
        self._keep = dict(
            cdf=np.ascontiguousarray(t["cdf"], dtype=np.float64),
            rs=np.ascontiguousarray(t["rs"], dtype=np.float32),
            state_map=np.ascontiguousarray(t["state_map"], dtype=np.int32),
            term_mask=np.ascontiguousarray(t["term_mask"], dtype=np.uint64),
            s0_cdf=np.ascontiguousarray(t["s0_cdf"], dtype=np.float64),
            s0_ids=np.ascontiguousarray(t["s0_ids"], dtype=np.int32),
            max_steps=np.ascontiguousarray(t["max_steps"], dtype=np.int32))
        self.state = np.zeros(self.n_env, dtype=np.int32)
        self.steps = np.zeros(self.n_env, dtype=np.int32)
        self.need_reset = np.ones(self.n_env, dtype=np.uint8)
        k = self._keep
        self._h = _AnyMDPStruct(self.n_env, self.n_task, self.S, self.A, self.s0_max,
                                _p(k["cdf"]), _p(k["rs"]), _p(k["state_map"]), _p(k["term_mask"]),
                                _p(k["s0_cdf"]), _p(k["s0_ids"]), _p(k["max_steps"]), _p(self.env_task),
                                _p(self.state), _p(self.steps), _p(self.need_reset), 0, int(gid_stride))

    @property
    def err_flags(self):
        return int(self._h.err_flags)

    def _outs(self):
        n = self.n_env
        return (np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.float32),
                np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.full(n, -1, np.int32))

    def reset_injected(self, u, mask=None):
        obs = np.full(self.n_env, -1, np.int32)
        u = np.ascontiguousarray(u, np.float64)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().xo_anymdp_reset_injected(C.byref(self._h), _p(m), _p(u), _p(obs))
        return obs

    def reset(self, seed, gid_base, tick, mask=None):
        obs = np.full(self.n_env, -1, np.int32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().xo_anymdp_reset(C.byref(self._h), C.c_uint64(seed), C.c_uint64(gid_base), C.c_uint64(tick),
                              _p(m), _p(obs))
        return obs

    def step_injected(self, action, u, z, u_reset, mode):
        obs, rew, rgt, term, trunc, fobs = self._outs()
        a = np.ascontiguousarray(action, np.int32)
        u = np.ascontiguousarray(u, np.float64)
        z = np.ascontiguousarray(z, np.float32)
        ur = np.ascontiguousarray(u_reset, np.float64)
        lib().xo_anymdp_step_injected(C.byref(self._h), _p(a), _p(u), _p(z), _p(ur), _p(obs), _p(rew),
                                      _p(rgt), _p(term), _p(trunc), _p(fobs), C.c_int(mode))
        return obs, rew, rgt, term, trunc, fobs

    def step(self, seed, gid_base, tick, action, mode, n_threads=0):
        obs, rew, rgt, term, trunc, fobs = self._outs()
        a = np.ascontiguousarray(action, np.int32)
        if n_threads and n_threads > 1:
            lib().xo_anymdp_step_mt(C.byref(self._h), C.c_uint64(seed), C.c_uint64(gid_base),
                                    C.c_uint64(tick), _p(a), _p(obs), _p(rew), _p(rgt), _p(term),
                                    _p(trunc), _p(fobs), C.c_int(mode), C.c_int(n_threads))
        else:
            lib().xo_anymdp_step(C.byref(self._h), C.c_uint64(seed), C.c_uint64(gid_base),
                                 C.c_uint64(tick), _p(a), _p(obs), _p(rew), _p(rgt), _p(term), _p(trunc),
                                 _p(fobs), C.c_int(mode))
        return obs, rew, rgt, term, trunc, fobs

    def solve(self, gamma=0.99, tol=1.0e-4, max_iter=20000):
        """the ground-truth teacher's table as xv_anymdp_solve computes it -> (Q f64[n_task,S,A], greedy u8[n_task,S], sweeps)"""
        q = np.zeros((self.n_task, self.S, self.A))
        g = np.zeros((self.n_task, self.S), np.uint8)
        it = np.zeros(self.n_task, np.int32)
        lib().xo_anymdp_solve(C.byref(self._h), C.c_double(gamma), C.c_double(tol), C.c_int(max_iter), _p(q), _p(g), _p(it))
        return q, g, it

    def rollout_teacher(self, seed, gid_base, tick0, T, greedy, epsilon=0.0):
        """xv_anymdp_rollout_teacher restated: per step the action is greedy[task, inner state] (the policy of
        AnyMDPSolverOpt, anymdp_solver_opt.py:38-51), replaced by a uniform action with probability epsilon (draw
        purpose 2 of the step's tick: word 0 -> 24-bit uniform, word 1 -> action), then one SAME_STEP step."""
        greedy = np.asarray(greedy)
        keys = ("action", "obs", "reward", "reward_gt", "terminated", "truncated", "final_obs")
        rec = {k: [] for k in keys}
        for ts in range(T):
            a = greedy[self.env_task, self.state].astype(np.int32)
            if epsilon > 0.0:
                for i in range(self.n_env):
                    w = env_draw(seed, gid_base + i, tick0 + ts, 2)
                    if np.float32(int(w[0]) >> 8) * np.float32(1.0 / 16777216.0) < np.float32(epsilon):
                        a[i] = int(w[1]) % self.A
            o = self.step(seed, gid_base, tick0 + ts, a, 2)
            for k, v in zip(keys, (a,) + o):
                rec[k].append(np.array(v, copy=True))
        return {k: np.stack(v) for k, v in rec.items()}

    def transition_gt(self, action):
        a = np.ascontiguousarray(action, np.int32)
        out = np.zeros((self.n_env, self.S), np.float64)
        lib().xo_anymdp_transition_gt(C.byref(self._h), _p(a), _p(out))
        return out


def anymdp_sample_observation_model(seed, task_base, n_task, S, n_obs, d_obs, density=0.20, maximum_distribution=4):
    """xv_anymdp_sample_observation_model restated (same draws): obs_cdf float64[n_task, d_obs, S, n_obs]"""
    out = np.empty((n_task, d_obs, S, n_obs), np.float64)
    lib().xo_anymdp_sample_observation_model(C.c_uint64(seed), C.c_int64(task_base), C.c_int(n_task), C.c_int(S), C.c_int(n_obs),
                                             C.c_int(d_obs), C.c_double(density), C.c_double(maximum_distribution), _p(out))
    return out


def anymdp_synth(seed, task_index_base, n_task, S, A, s0_max, task_stride=1):
    """Synthetic task tables (same bits as the device generator xv_anymdp_synth_tasks) of the tasks
    task_index_base + k * task_stride."""
    words = (S + 63) // 64
    t = dict(S=S, A=A, s0_max=s0_max,
             cdf=np.empty((n_task, S, A, S), np.float64), rs=np.empty((n_task, S, A, S, 2), np.float32),
             state_map=np.empty((n_task, S), np.int32), term_mask=np.empty((n_task, words), np.uint64),
             s0_cdf=np.empty((n_task, s0_max), np.float64), s0_ids=np.empty((n_task, s0_max), np.int32),
             max_steps=np.empty(n_task, np.int32))
    lib().xo_anymdp_synth_strided(C.c_uint64(seed), C.c_int64(task_index_base), C.c_int64(task_stride), C.c_int(n_task), C.c_int(S),
                          C.c_int(A), C.c_int(s0_max), _p(t["cdf"]), _p(t["rs"]), _p(t["state_map"]),
                          _p(t["term_mask"]), _p(t["s0_cdf"]), _p(t["s0_ids"]), _p(t["max_steps"]))
    return t


# ---------------------------------------------------------------------------------------------------
# LinDS
# ---------------------------------------------------------------------------------------------------
class _LinDSStruct(C.Structure):
    _fields_ = [("n_env", C.c_int), ("n_task", C.c_int), ("NS", C.c_int), ("NA", C.c_int), ("NO", C.c_int),
                ("NI", C.c_int)] + [(k, C.c_void_p) for k in
                                    ("phiT", "gamT", "cT", "xt", "y0", "valid", "cmd0", "four_coef", "four_omega",
                                     "four_period", "scal", "ints", "init", "env_task", "x", "steps",
                                     "need_reset")] + [("err_flags", C.c_uint32)]


class LinDSOracle(object):
    """Batched CPU LinDS over the device table layout (xenoverse_amd.linds.tables.build_tables)."""
    _F32 = ("phiT", "gamT", "cT", "xt", "y0", "valid", "cmd0", "four_coef", "scal", "init")

    def __init__(self, tables, env_task):
        t = tables
        self.NS, self.NA, self.NO, self.NI = int(t["NS"]), int(t["NA"]), int(t["NO"]), int(t["NI"])
        self.env_task = np.ascontiguousarray(env_task, np.int32)
        self.n_env = len(self.env_task)
        self.n_task = t["phiT"].shape[0]
        k = {n: np.ascontiguousarray(t[n], np.float32) for n in self._F32}
        k["four_omega"] = np.ascontiguousarray(t["four_omega"], np.float64)
        k["four_period"] = np.ascontiguousarray(t["four_period"], np.float64)
        k["ints"] = np.ascontiguousarray(t["ints"], np.int32)
        self._keep = k
        self.x = np.zeros((self.NS, self.n_env), np.float32)
        self.steps = np.zeros(self.n_env, np.int32)
        self.need_reset = np.ones(self.n_env, np.uint8)
        self._h = _LinDSStruct(self.n_env, self.n_task, self.NS, self.NA, self.NO, self.NI,
                               *[_p(k[n]) for n in ("phiT", "gamT", "cT", "xt", "y0", "valid", "cmd0", "four_coef",
                                                    "four_omega", "four_period", "scal", "ints", "init")],
                               _p(self.env_task), _p(self.x), _p(self.steps), _p(self.need_reset), 0)

    @property
    def err_flags(self):
        return int(self._h.err_flags)

    def _outs(self):
        n, no = self.n_env, self.NO
        return dict(obs=np.zeros((n, no), np.float32), reward=np.zeros(n, np.float32),
                    terminated=np.zeros(n, np.uint8), truncated=np.zeros(n, np.uint8),
                    cmd=np.zeros((n, no), np.float32), error=np.zeros(n, np.float32),
                    final_obs=np.zeros((n, no), np.float32))

    def cmd(self, task, t):
        out = np.zeros(self.NO, np.float32)
        lib().xo_linds_cmd(C.byref(self._h), C.c_int(task), C.c_int(t), _p(out))
        return out

    def reset_injected(self, init_index, mask=None):
        o = self._outs()
        idx = np.ascontiguousarray(init_index, np.int32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().xo_linds_reset_injected(C.byref(self._h), _p(m), _p(idx), _p(o["obs"]), _p(o["cmd"]), _p(o["error"]))
        return o

    def reset(self, seed, gid_base, tick, mask=None):
        o = self._outs()
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().xo_linds_reset(C.byref(self._h), C.c_uint64(seed), C.c_uint64(gid_base), C.c_uint64(tick), _p(m),
                             _p(o["obs"]), _p(o["cmd"]), _p(o["error"]))
        return o

    def step_injected(self, action, z, init_index, mode):
        o = self._outs()
        a = np.ascontiguousarray(action, np.float32).reshape(self.n_env, self.NA)
        z = np.ascontiguousarray(z, np.float32).reshape(self.NS, self.n_env)
        idx = np.ascontiguousarray(init_index, np.int32)
        lib().xo_linds_step_injected(C.byref(self._h), _p(a), _p(z), _p(idx), _p(o["obs"]), _p(o["reward"]),
                                     _p(o["terminated"]), _p(o["truncated"]), _p(o["cmd"]), _p(o["error"]),
                                     _p(o["final_obs"]), C.c_int(mode))
        return o

    def step(self, seed, gid_base, tick, action, mode, n_threads=1):
        o = self._outs()
        a = np.ascontiguousarray(action, np.float32).reshape(self.n_env, self.NA)
        lib().xo_linds_step(C.byref(self._h), C.c_uint64(seed), C.c_uint64(gid_base), C.c_uint64(tick), _p(a),
                            _p(o["obs"]), _p(o["reward"]), _p(o["terminated"]), _p(o["truncated"]), _p(o["cmd"]),
                            _p(o["error"]), _p(o["final_obs"]), C.c_int(mode), C.c_int(n_threads))
        return o


def linds_yorder(NS):
    ord_ = np.zeros(32, np.int32)
    n = lib().xo_linds_yorder(C.c_int(NS), _p(ord_))
    return ord_[:n].copy()


# ---------------------------------------------------------------------------------------------------
# CartPole
# ---------------------------------------------------------------------------------------------------
class _CartPoleStruct(C.Structure):
    _fields_ = [("n_env", C.c_int), ("n_task", C.c_int), ("frameskip", C.c_int), ("max_steps", C.c_int),
                ("params", C.c_void_p), ("reset_scale", C.c_void_p), ("env_task", C.c_void_p),
                ("state", C.c_void_p), ("steps", C.c_void_p), ("need_reset", C.c_void_p),
                ("err_flags", C.c_uint32)]


class CartPoleOracle(object):
    def __init__(self, params, env_task, frameskip=1, max_steps=0, reset_scale=(0.45, 0.90, 0.13, 1.0)):
        self.params = np.ascontiguousarray(params, np.float64).reshape(-1, 4)
        self.env_task = np.ascontiguousarray(env_task, np.int32)
        self.n_env = len(self.env_task)
        self.scale = np.ascontiguousarray(reset_scale, np.float64)
        self.state = np.zeros((4, self.n_env), np.float64)
        self.steps = np.zeros(self.n_env, np.int32)
        self.need_reset = np.ones(self.n_env, np.uint8)
        self._h = _CartPoleStruct(self.n_env, len(self.params), frameskip, max_steps, _p(self.params),
                                  _p(self.scale), _p(self.env_task), _p(self.state), _p(self.steps),
                                  _p(self.need_reset), 0)

    def _outs(self):
        n = self.n_env
        return dict(obs=np.zeros((n, 4), np.float32), reward=np.zeros(n, np.float32),
                    terminated=np.zeros(n, np.uint8), truncated=np.zeros(n, np.uint8),
                    final_obs=np.zeros((n, 4), np.float32))

    def reset_injected(self, u, mask=None):
        obs = np.zeros((self.n_env, 4), np.float32)
        u = np.ascontiguousarray(u, np.float64).reshape(4, self.n_env)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().xo_cartpole_reset_injected(C.byref(self._h), _p(m), _p(u), _p(obs))
        return obs

    def reset(self, seed, gid_base, tick, mask=None):
        obs = np.zeros((self.n_env, 4), np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().xo_cartpole_reset(C.byref(self._h), C.c_uint64(seed), C.c_uint64(gid_base), C.c_uint64(tick), _p(m), _p(obs))
        return obs

    def step_injected(self, action, u_reset, mode):
        o = self._outs()
        a = np.ascontiguousarray(action, np.int32)
        u = np.ascontiguousarray(u_reset, np.float64).reshape(4, self.n_env)
        lib().xo_cartpole_step_injected(C.byref(self._h), _p(a), _p(u), _p(o["obs"]), _p(o["reward"]),
                                        _p(o["terminated"]), _p(o["truncated"]), _p(o["final_obs"]), C.c_int(mode))
        return o

    def step(self, seed, gid_base, tick, action, mode):
        o = self._outs()
        a = np.ascontiguousarray(action, np.int32)
        lib().xo_cartpole_step(C.byref(self._h), C.c_uint64(seed), C.c_uint64(gid_base), C.c_uint64(tick), _p(a),
                               _p(o["obs"]), _p(o["reward"]), _p(o["terminated"]), _p(o["truncated"]),
                               _p(o["final_obs"]), C.c_int(mode))
        return o


# ---------------------------------------------------------------------------------------------------
# Acrobot
# ---------------------------------------------------------------------------------------------------
class _AcrobotStruct(C.Structure):
    _fields_ = [("n_env", C.c_int), ("n_task", C.c_int), ("frameskip", C.c_int), ("max_steps", C.c_int),
                ("scale_is_vector", C.c_int), ("params", C.c_void_p), ("reset_scale", C.c_void_p),
                ("env_task", C.c_void_p), ("state", C.c_void_p), ("fresh", C.c_void_p), ("steps", C.c_void_p),
                ("need_reset", C.c_void_p), ("err_flags", C.c_uint32)]


def acrobot_dsdt(params, y):
    """RandomAcrobotEnv._dsdt for rows of params[n][7], y[n][5] -> [n][5]"""
    params = np.ascontiguousarray(params, np.float64).reshape(-1, 7)
    y = np.ascontiguousarray(y, np.float64).reshape(-1, 5)
    out = np.zeros_like(y)
    f = lib().xo_acrobot_dsdt
    for i in range(len(y)):
        f(_p(params[i]), _p(y[i]), _p(out[i]))
    return out


def acrobot_terminal(params, s):
    params = np.ascontiguousarray(params, np.float64).reshape(-1, 7)
    s = np.ascontiguousarray(s, np.float64).reshape(-1, 4)
    f = lib().xo_acrobot_terminal
    f.restype = C.c_int
    return np.array([f(_p(params[i]), _p(s[i])) for i in range(len(s))], np.uint8)


class AcrobotOracle(object):
    def __init__(self, params, env_task, frameskip=1, max_steps=0, reset_scale=0.10):
        self.params = np.ascontiguousarray(params, np.float64).reshape(-1, 7)
        self.env_task = np.ascontiguousarray(env_task, np.int32)
        self.n_env = len(self.env_task)
        vec = not np.isscalar(reset_scale)
        self.scale = np.ascontiguousarray(reset_scale if vec else [reset_scale] * 4, np.float64)
        self.state = np.zeros((4, self.n_env), np.float64)
        self.fresh = np.zeros(self.n_env, np.uint8)
        self.steps = np.zeros(self.n_env, np.int32)
        self.need_reset = np.ones(self.n_env, np.uint8)
        self._h = _AcrobotStruct(self.n_env, len(self.params), frameskip, max_steps, int(vec), _p(self.params),
                                 _p(self.scale), _p(self.env_task), _p(self.state), _p(self.fresh), _p(self.steps),
                                 _p(self.need_reset), 0)

    @property
    def err_flags(self):
        return int(self._h.err_flags)

    def _outs(self):
        n = self.n_env
        return dict(obs=np.zeros((n, 6), np.float32), reward=np.zeros(n, np.float32),
                    terminated=np.zeros(n, np.uint8), truncated=np.zeros(n, np.uint8),
                    final_obs=np.zeros((n, 6), np.float32))

    def reset_injected(self, u, mask=None):
        obs = np.zeros((self.n_env, 6), np.float32)
        u = np.ascontiguousarray(u, np.float64).reshape(4, self.n_env)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().xo_acrobot_reset_injected(C.byref(self._h), _p(m), _p(u), _p(obs))
        return obs

    def reset(self, seed, gid_base, tick, mask=None):
        obs = np.zeros((self.n_env, 6), np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().xo_acrobot_reset(C.byref(self._h), C.c_uint64(seed), C.c_uint64(gid_base), C.c_uint64(tick), _p(m), _p(obs))
        return obs

    def step_injected(self, action, u_reset, mode):
        o = self._outs()
        a = np.ascontiguousarray(action, np.int32)
        u = np.ascontiguousarray(u_reset, np.float64).reshape(4, self.n_env)
        lib().xo_acrobot_step_injected(C.byref(self._h), _p(a), _p(u), _p(o["obs"]), _p(o["reward"]),
                                       _p(o["terminated"]), _p(o["truncated"]), _p(o["final_obs"]), C.c_int(mode))
        return o

    def step(self, seed, gid_base, tick, action, mode):
        o = self._outs()
        a = np.ascontiguousarray(action, np.int32)
        lib().xo_acrobot_step(C.byref(self._h), C.c_uint64(seed), C.c_uint64(gid_base), C.c_uint64(tick), _p(a),
                              _p(o["obs"]), _p(o["reward"]), _p(o["terminated"]), _p(o["truncated"]),
                              _p(o["final_obs"]), C.c_int(mode))
        return o


# ---------------------------------------------------------------------------------------------------
# MazeWorld
# ---------------------------------------------------------------------------------------------------
class _MazeStruct(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("n_env", "n_task", "NG", "n_cmd", "max_steps", "W", "H",
                                       "command_in_observation")] + \
               [("collision_dist", C.c_double), ("visibility", C.c_double)] + \
               [(k, C.c_void_p) for k in ("walls", "texts", "landmarks", "ints", "dbl", "commands", "lm_coord",
                                          "tex_walls", "tex_grounds", "tex_ceilings", "env_task", "pos", "ori",
                                          "grid", "steps", "cmd_idx", "cmd_age", "need_reset", "collision")]


class MazeOracle(object):
    """tables: xenoverse_amd.mazeworld.tables.build_tables; textures: dict(walls, grounds, ceilings) float32"""

    def __init__(self, tables, textures, env_task, resolution=(32, 32), max_steps=5000, visibility_3D=12.0,
                 collision_dist=0.20, command_in_observation=False):
        t = tables
        self.env_task = np.ascontiguousarray(env_task, np.int32)
        self.n_env = n = len(self.env_task)
        self.W, self.H = int(resolution[0]), int(resolution[1])
        self._keep = dict(walls=np.ascontiguousarray(t["walls"], np.int8),
                          texts=np.ascontiguousarray(t["texts"], np.int32),
                          landmarks=np.ascontiguousarray(t["landmarks"], np.int8),
                          ints=np.ascontiguousarray(t["ints"], np.int32), dbl=np.ascontiguousarray(t["dbl"], np.float64),
                          commands=np.ascontiguousarray(t["commands"], np.int32),
                          lm_coord=np.ascontiguousarray(t["lm_coord"], np.int32),
                          tw=np.ascontiguousarray(textures["walls"], np.float32),
                          tg=np.ascontiguousarray(textures["grounds"], np.float32),
                          tc=np.ascontiguousarray(textures["ceilings"], np.float32))
        k = self._keep
        self.pos = np.zeros((2, n), np.float64); self.ori = np.zeros(n, np.float64)
        self.grid = np.zeros((2, n), np.int32); self.steps = np.zeros(n, np.int32)
        self.cmd_idx = np.zeros(n, np.int32); self.cmd_age = np.zeros(n, np.int32)
        self.need_reset = np.ones(n, np.uint8); self.collision = np.zeros(n, np.float64)
        self._h = _MazeStruct(n, k["walls"].shape[0], int(t["NG"]), int(t["n_cmd"]), int(max_steps), self.W, self.H,
                              int(bool(command_in_observation)), float(collision_dist), float(visibility_3D),
                              _p(k["walls"]), _p(k["texts"]), _p(k["landmarks"]), _p(k["ints"]), _p(k["dbl"]),
                              _p(k["commands"]), _p(k["lm_coord"]), _p(k["tw"]), _p(k["tg"]), _p(k["tc"]),
                              _p(self.env_task), _p(self.pos), _p(self.ori), _p(self.grid), _p(self.steps),
                              _p(self.cmd_idx), _p(self.cmd_age), _p(self.need_reset), _p(self.collision))

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().xo_maze_reset(C.byref(self._h), _p(m))

    def step(self, action, mode):
        a = np.ascontiguousarray(action, np.float64).reshape(self.n_env, 2)
        r = np.zeros(self.n_env, np.float32); te = np.zeros(self.n_env, np.uint8); tr = np.zeros(self.n_env, np.uint8)
        lib().xo_maze_step(C.byref(self._h), _p(a), _p(r), _p(te), _p(tr), C.c_int(mode))
        return r, te, tr

    def render(self, n_threads=1, typing="stub"):
        """typing "stub": the reference's source as plain Python under NumPy 2 (the pinned one); "numba": DDA and
        wall-column geometry in float64, as numba types the same source (unpinned)"""
        f = np.zeros((self.n_env, self.W, self.H, 3), np.uint8)
        c = np.zeros((self.n_env, 3), np.float32)
        lib().xo_maze_render_typed(C.byref(self._h), _p(f), _p(c), C.c_int(n_threads), C.c_int(typing == "numba"))
        return f, c

    def expose(self, seed, gid_base, tick, prob=0.05):
        """maze_core._cell_exposed of the present pose: uint8[n_env, NG, NG] (ray_caster_utils.py:250-255)"""
        NG = self._h.NG
        ex = np.zeros((self.n_env, NG, NG), np.uint8)
        lib().xo_maze_expose(C.byref(self._h), C.c_uint64(seed), C.c_uint64(gid_base), C.c_uint64(tick),
                             C.c_double(prob), _p(ex))
        return ex


class _MazeAgentStruct(C.Structure):
    _fields_ = [("env", C.c_void_p), ("stm_size", C.c_int), ("oracle_agent", C.c_int), ("keep_ratio", C.c_double),
                ("na", C.c_int), ("actions", C.c_void_p), ("stm", C.c_void_p), ("stm_len", C.c_void_p),
                ("ltm", C.c_void_p), ("mask", C.c_void_p), ("cost", C.c_void_p), ("path", C.c_void_p)]


class MazeAgentOracle(object):
    """SmartSLAMAgent / OracleAgent over a MazeOracle's envs (xeno_oracle_agent.c)"""
    STM_MAX = 8

    def __init__(self, maze, actions, short_term_memory_size=3, memory_keep_ratio=1.0, oracle_agent=False):
        self.maze = maze
        n, NG = maze.n_env, maze._h.NG
        self.NG = NG
        self.actions = np.ascontiguousarray(actions, np.float64).reshape(-1, 2)
        self.stm = np.zeros((n, self.STM_MAX, NG, NG), np.uint8); self.stm_len = np.zeros(n, np.int32)
        self.ltm = np.zeros((n, NG, NG), np.uint8); self.mask = np.zeros((n, NG, NG), np.uint8)
        self.cost = np.zeros((n, NG, NG), np.float64); self.path = np.zeros((n, 5), np.int32)
        assert short_term_memory_size < self.STM_MAX
        self._h = _MazeAgentStruct(C.cast(C.pointer(maze._h), C.c_void_p), int(short_term_memory_size),
                                   int(bool(oracle_agent)), float(memory_keep_ratio), len(self.actions),
                                   _p(self.actions), _p(self.stm), _p(self.stm_len), _p(self.ltm), _p(self.mask),
                                   _p(self.cost), _p(self.path))

    def act(self, exposed, u_keep=None):
        ex = np.ascontiguousarray(exposed, np.uint8)
        assert ex.shape == (self.maze.n_env, self.NG, self.NG)
        uk = None if u_keep is None else np.ascontiguousarray(u_keep, np.float64)
        a = np.zeros(self.maze.n_env, np.int32)
        lib().xo_maze_agent_act(C.byref(self._h), _p(ex), _p(uk), _p(a))
        return a


def maze_search_action(ori, targ1, targ2, actions):
    a = np.ascontiguousarray(actions, np.float64).reshape(-1, 2)
    t1 = np.ascontiguousarray(targ1, np.float64)
    t2 = None if targ2 is None else np.ascontiguousarray(targ2, np.float64)
    f = lib().xo_maze_search_action
    f.restype = C.c_int
    return int(f(C.c_double(ori), _p(t1), _p(t2), _p(a), C.c_int(len(a))))


# ---------------------------------------------------------------------------------------------------
# AnyMDP POMDP / MTPOMDP
# ---------------------------------------------------------------------------------------------------
class _AnyMDPTokStruct(C.Structure):
    _fields_ = [("m", C.c_void_p), ("n_obs", C.c_int), ("d_obs", C.c_int), ("d_act", C.c_int), ("obs_cdf", C.c_void_p)]


class AnyMDPTokOracle(AnyMDPOracle):
    """POMDP / multi-token POMDP on top of AnyMDPOracle.  obs_cdf float64[n_task, d_obs, S, n_obs]."""

    def __init__(self, tables, env_task, obs_cdf, d_act):
        super().__init__(tables, env_task)
        self.obs_cdf = np.ascontiguousarray(obs_cdf, np.float64)
        _, self.d_obs, _, self.n_obs = self.obs_cdf.shape
        self.d_act = int(d_act)
        self._t = _AnyMDPTokStruct(C.cast(C.pointer(self._h), C.c_void_p), self.n_obs, self.d_obs, self.d_act,
                                   _p(self.obs_cdf))

    def _touts(self):
        n = self.n_env
        return (np.zeros((n, self.d_obs), np.int32), np.zeros(n, np.float32), np.zeros(n, np.float32),
                np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.full((n, self.d_obs), -1, np.int32))

    def tok_reset_injected(self, u_reset, u_obs_reset, mask=None):
        obs = np.full((self.n_env, self.d_obs), -1, np.int32)
        ur = np.ascontiguousarray(u_reset, np.float64)
        uo = np.ascontiguousarray(u_obs_reset, np.float64).reshape(self.d_obs, self.n_env)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().xo_anymdp_tok_reset_injected(C.byref(self._t), _p(m), _p(ur), _p(uo), _p(obs))
        return obs

    def tok_reset(self, seed, gid_base, tick, mask=None):
        obs = np.full((self.n_env, self.d_obs), -1, np.int32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().xo_anymdp_tok_reset(C.byref(self._t), C.c_uint64(seed), C.c_uint64(gid_base), C.c_uint64(tick), _p(m), _p(obs))
        return obs

    def tok_step_injected(self, action, u, z, u_obs, u_reset, u_obs_reset, mode):
        obs, rew, rgt, term, trunc, fobs = self._touts()
        n = self.n_env
        a = np.ascontiguousarray(action, np.int32).reshape(n, self.d_act)
        u = np.ascontiguousarray(u, np.float64).reshape(self.d_act, n)
        z = np.ascontiguousarray(z, np.float32).reshape(self.d_act, n)
        uo = np.ascontiguousarray(u_obs, np.float64).reshape(self.d_obs, n)
        ur = np.ascontiguousarray(u_reset, np.float64)
        uor = np.ascontiguousarray(u_obs_reset, np.float64).reshape(self.d_obs, n)
        lib().xo_anymdp_tok_step_injected(C.byref(self._t), _p(a), _p(u), _p(z), _p(uo), _p(ur), _p(uor), _p(obs),
                                          _p(rew), _p(rgt), _p(term), _p(trunc), _p(fobs), C.c_int(mode))
        return obs, rew, rgt, term, trunc, fobs

    def tok_step(self, seed, gid_base, tick, action, mode):
        obs, rew, rgt, term, trunc, fobs = self._touts()
        a = np.ascontiguousarray(action, np.int32).reshape(self.n_env, self.d_act)
        lib().xo_anymdp_tok_step(C.byref(self._t), C.c_uint64(seed), C.c_uint64(gid_base), C.c_uint64(tick), _p(a),
                                 _p(obs), _p(rew), _p(rgt), _p(term), _p(trunc), _p(fobs), C.c_int(mode))
        return obs, rew, rgt, term, trunc, fobs


# ---------------------------------------------------------------------------------------------------
# AnyMDP task sampler arithmetic (xeno_oracle_sampler.c)
# ---------------------------------------------------------------------------------------------------
def np_pairwise_sum(a):
    a = np.ascontiguousarray(a, np.float64).reshape(-1)
    return lib().xo_np_pairwise_sum(_p(a), C.c_int64(a.size))


def update_value_matrix(t_mat, r_mat, gamma, vm, is_greedy=True):
    """solver.py:57-82 in the reference's order of operations -> (value matrix, sweeps)"""
    t = np.ascontiguousarray(t_mat, np.float64)
    r = np.ascontiguousarray(r_mat, np.float64)
    ns, na, _ = t.shape
    out = np.array(vm, np.float64, order="C", copy=True).reshape(ns, na)
    it = lib().xo_update_value_matrix(_p(t), _p(r), C.c_int(ns), C.c_int(na), C.c_double(float(gamma)), _p(out),
                                      C.c_int(1 if is_greedy else 0))
    return out, int(it)


class CandInfo(C.Structure):
    _fields_ = [("status", C.c_int32), ("goal", C.c_int32), ("n_s0", C.c_int32), ("repair_rounds", C.c_int32),
                ("s0", C.c_int32 * 4), ("sweeps", C.c_int32 * 8), ("band_lo", C.c_int32 * 256),
                ("band_hi", C.c_int32 * 256), ("state_map", C.c_int32 * 256), ("s_e", C.c_uint8 * 256),
                ("max_steps", C.c_double), ("gini", C.c_double), ("ent", C.c_double), ("gap_min", C.c_double),
                ("s0_prob", C.c_double * 4)]


def anymdp_sample_candidate(seed, cand, ns, na):
    """one candidate of the device task sampler, restated on the CPU -> dict (status 0 = accepted)"""
    T = np.zeros((ns, na, ns)); R = np.zeros((ns, na, ns)); noise = np.zeros((ns, na, ns))
    info = CandInfo()
    f = lib().xo_anymdp_sample_candidate
    f.restype = C.c_int
    st = f(C.c_uint64(seed), C.c_uint64(cand), C.c_int(ns), C.c_int(na), _p(T), _p(R), _p(noise), C.byref(info))
    n0 = info.n_s0
    return dict(status=int(st), transition=T, reward=R, reward_noise=noise, max_steps=info.max_steps, goal=bool(info.goal),
                s_0=np.array(info.s0[:n0], np.int64), s_0_prob=np.array(info.s0_prob[:n0]),
                s_e=np.nonzero(np.frombuffer(info.s_e, np.uint8)[:ns])[0], state_mapping=np.array(info.state_map[:ns], np.int64),
                band_lo=np.array(info.band_lo[:ns]), band_hi=np.array(info.band_hi[:ns]), sweeps=np.array(info.sweeps[:]),
                repair_rounds=info.repair_rounds, gini=info.gini, ent=info.ent, gap_min=info.gap_min)
