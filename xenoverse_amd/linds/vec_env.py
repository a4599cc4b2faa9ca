"""LinDSVecEnv — N randomised LTI control environments stepped per kernel launch on one MI355X.

Mirrors the reference's per-env interface (xenoverse/linds/linds_env.py: LinearDSEnv.__init__ :16-38,
set_task :40-65, reset :108-131, step :133-169, get_future_inner_cmds :171-183, state :185-187) behind the
gymnasium VectorEnv surface.  Same task dicts, same info keys (`steps`, `command`,
`command_type` on reset, `error`), same exception messages for misuse.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..engine import AUTORESET
from ..spaces import Box
from ..vector import OutputSlabs, VectorEnv
from .tables import build_tables

_F32 = ("phiT", "gamT", "cT", "xt", "y0", "valid", "cmd0", "four_coef", "scal", "init")
_F64 = ("four_omega", "four_period")


class _Tables(C.Structure):   # xv_linds_tables (include/xeno.h)
    _fields_ = [(k, C.c_void_p) for k in ("phiT", "gamT", "cT", "xt", "y0", "valid", "cmd0", "four_coef",
                                          "four_omega", "four_period", "scal", "ints", "init")]


def _pad_to(n, choices):
    for c in choices:
        if n <= c:
            return c
    raise ValueError("dimension %d exceeds the supported maximum %d" % (n, choices[-1]))


def pad_tables(tab):
    """Zero-pad the batch dims to the kernel's compiled sizes (NS in {16,32}, NA in {8,16}, NO in {16,32}).
    Zero padding is exact: fmaf(0, x, acc) == acc."""
    NS, NA, NO = _pad_to(tab["NS"], (16, 32)), _pad_to(tab["NA"], (8, 16)), _pad_to(tab["NO"], (16, 32))
    if (NS, NA, NO) == (tab["NS"], tab["NA"], tab["NO"]):
        return tab
    n = tab["phiT"].shape[0]
    out = dict(tab, NS=NS, NA=NA, NO=NO)

    def z(shape, dt=np.float32):
        return np.zeros(shape, dt)
    s, a, o = tab["NS"], tab["NA"], tab["NO"]
    out["phiT"] = z((n, NS, NS)); out["phiT"][:, :s, :s] = tab["phiT"]
    out["gamT"] = z((n, NA, NS)); out["gamT"][:, :a, :s] = tab["gamT"]
    out["cT"] = z((n, NS, NO)); out["cT"][:, :s, :o] = tab["cT"]
    out["xt"] = z((n, NS)); out["xt"][:, :s] = tab["xt"]
    out["y0"] = z((n, NO)); out["y0"][:, :o] = tab["y0"]
    out["valid"] = z((n, NO)); out["valid"][:, :o] = tab["valid"]
    out["cmd0"] = z((n, NO)); out["cmd0"][:, :o] = tab["cmd0"]
    out["four_coef"] = z(tab["four_coef"].shape[:2] + (NO, 2)); out["four_coef"][:, :, :o] = tab["four_coef"]
    out["init"] = z((n, tab["NI"], NS)); out["init"][:, :, :s] = tab["init"]
    return out


class LinDSVecEnv(VectorEnv):
    def __init__(self, num_envs, dt=0.1, max_steps=1000, pad_observation_dim=16, pad_command_dim=16,
                 pad_action_dim=8, device="cuda:0", seed=0, env_id_base=0, autoreset_mode="same_step",
                 to_numpy=False, engine=None, copy=True):
        super().__init__(num_envs, device=device, seed=seed, env_id_base=env_id_base,
                         autoreset_mode=autoreset_mode, to_numpy=to_numpy, engine=engine, copy=copy)
        self.dt = dt
        self.max_steps = max_steps
        self.pad_observation_dim = pad_observation_dim
        self.pad_command_dim = pad_command_dim
        self.pad_action_dim = pad_action_dim
        self._set_spaces(Box(-np.inf, np.inf, shape=(pad_observation_dim,), dtype=np.float32),
                         Box(-1, 1, shape=(pad_action_dim,), dtype=np.float32))
        self._h = None

    def set_task(self, tasks, env_task_index=None):
        """tasks: one reference task dict, a list of them, or prebuilt tables (linds.tables.build_tables)."""
        if isinstance(tasks, dict) and "phiT" in tasks:
            tab = tasks
        else:
            tab = build_tables(tasks, dt=self.dt, pad_observation_dim=self.pad_observation_dim,
                               pad_action_dim=self.pad_action_dim, pad_command_dim=self.pad_command_dim)
        self.user_dims = (int(tab["NA"]), int(tab["NO"]))
        tab = pad_tables(tab)
        d = self.device
        dev = {}
        for k in _F32:
            dev[k] = torch.as_tensor(np.ascontiguousarray(tab[k], np.float32)).to(d) if not torch.is_tensor(tab[k]) \
                else tab[k].to(d, torch.float32).contiguous()
        for k in _F64:
            dev[k] = torch.as_tensor(np.ascontiguousarray(tab[k], np.float64)).to(d) if not torch.is_tensor(tab[k]) \
                else tab[k].to(d, torch.float64).contiguous()
        dev["ints"] = torch.as_tensor(np.ascontiguousarray(tab["ints"], np.int32)).to(d) \
            if not torch.is_tensor(tab["ints"]) else tab["ints"].to(d, torch.int32).contiguous()
        n_task = int(dev["phiT"].shape[0])
        if env_task_index is None:
            if self.num_envs % n_task != 0:
                raise ValueError("num_envs is not a multiple of the task count; pass env_task_index")
            env_task = torch.arange(self.num_envs, device=d, dtype=torch.int32) // (self.num_envs // n_task)
        else:
            env_task = self._dev(env_task_index, torch.int32)
            if env_task.shape != (self.num_envs,) or int(env_task.min()) < 0 or int(env_task.max()) >= n_task:
                raise ValueError("env_task_index must be (num_envs,) with entries in [0, n_task)")
        dev["env_task"] = env_task.contiguous()
        self.NS, self.NA, self.NO, self.NI = int(tab["NS"]), int(tab["NA"]), int(tab["NO"]), int(tab["NI"])
        if self._h is not None:
            self.lib.xv_linds_destroy(self._h)
            self._h = None
        ct = _Tables(*[_lib.ptr(dev[k]) for k in ("phiT", "gamT", "cT", "xt", "y0", "valid", "cmd0", "four_coef",
                                                   "four_omega", "four_period", "scal", "ints", "init")])
        h = C.c_void_p()
        _lib.check(self.lib.xv_linds_create(self.engine.handle, self.num_envs, n_task, self.NS, self.NA, self.NO,
                                            self.NI, C.byref(ct), _lib.ptr(dev["env_task"]), C.byref(h)))
        self._h = h
        self._tab = dev
        self.n_task = n_task
        n = self.num_envs
        self._step_cache = None      # (copy=False steps cache pointers and views of the buffers made here)
        self._slabs = None           # copy=True: output sets of 64 steps per allocation (vector.OutputSlabs), made at the first step
        self._obs = torch.zeros((n, self.NO), dtype=torch.float32, device=d)
        self._cmd = torch.zeros((n, self.NO), dtype=torch.float32, device=d)
        self._fobs = torch.zeros((n, self.NO), dtype=torch.float32, device=d)
        self._reward = torch.zeros(n, dtype=torch.float32, device=d)
        self._error = torch.zeros(n, dtype=torch.float32, device=d)
        self._term = torch.zeros(n, dtype=torch.uint8, device=d)
        self._trunc = torch.zeros(n, dtype=torch.uint8, device=d)
        self._steps = torch.zeros(n, dtype=torch.int32, device=d)
        self._done = torch.zeros(n, dtype=torch.uint8, device=d)
        self._path_name = "auto"
        self._command_type = np.where(tab["ints"][:, 3] > 0, "dynamic_target", "static_target") \
            if not torch.is_tensor(tab["ints"]) else None
        self.task_set = True
        self.need_reset = True

    PATH = {"auto": 0, "mfma": 1, "scalar": 2}

    def set_path(self, path):
        """Select the step kernel ("auto", "mfma", "scalar"); results are identical (include/xeno.h)."""
        _lib.check(self.lib.xv_linds_set_path(self._h, self.PATH[path]))
        self._path_name = path

    def set_command_table(self, enable):
        """Use the per-task table of get_inner_cmd values built at set_task (default) or evaluate the Fourier
        terms in every step; the results are identical."""
        _lib.check(self.lib.xv_linds_set_command_table(self._h, 1 if enable else 0))

    # -- helpers ------------------------------------------------------------------------------------
    def _user_obs(self, t):
        return t[:, :self.user_dims[1]]

    def _action(self, actions):
        a = self._dev(actions, torch.float32)
        na_user = self.user_dims[0]
        if a.shape != (self.num_envs, na_user):
            # reference: assert numpy.shape(action) == (self.pad_action_dim,) (linds_env.py:137)
            raise AssertionError(f"Action shape mismatch: expected {(self.num_envs, na_user)}, got {tuple(a.shape)}")
        if na_user != self.NA:
            p = torch.zeros((self.num_envs, self.NA), dtype=torch.float32, device=self.device)
            p[:, :na_user] = a
            a = p
        return a.contiguous()

    def _steps_now(self):
        self._renew("_steps")      # the launch writes every entry: a fresh buffer, handed out as it is (no copy)
        _lib.check(self.lib.xv_linds_get_state(self._h, None, _lib.ptr(self._steps), None))
        return self._steps if (self.copy and not self.to_numpy) else self._steps.clone()

    def _infos(self, with_final, fresh=False):
        o = self._of if fresh else self._o      # fresh: the buffers were renewed before the launch (step paths)
        infos = {"steps": self._out(self._steps_now()), "command": o(self._user_obs(self._cmd)),
                 "error": o(self._error)}
        if with_final and self.autoreset_mode == "same_step":
            infos["final_obs"] = o(self._user_obs(self._fobs))
            infos["_final_obs"] = self._out((self._term | self._trunc).view(torch.bool))
        return infos

    # -- API ----------------------------------------------------------------------------------------
    def reset(self, *, seed=None, options=None):
        self._require_task()
        if seed is not None:
            self.engine.tick = (int(seed) & 0xFFFFFFFF) << 24
        mask = None
        if options is not None and options.get("reset_mask") is not None:
            mask = self._dev(options["reset_mask"], torch.uint8)
        self._detach("_obs", "_cmd", "_error")
        _lib.check(self.lib.xv_linds_reset(self._h, _lib.ptr(mask), _lib.ptr(self._obs), _lib.ptr(self._cmd),
                                           _lib.ptr(self._error)))
        self.need_reset = False
        infos = self._infos(False)
        if self._command_type is not None:
            infos["command_type"] = self._command_type[self._tab["env_task"].cpu().numpy()]
        return self._o(self._user_obs(self._obs)), infos

    def reset_injected(self, init_index, mask=None):
        self._require_task()
        idx = self._dev(init_index, torch.int32)
        m = None if mask is None else self._dev(mask, torch.uint8)
        self._detach("_obs", "_cmd", "_error")
        _lib.check(self.lib.xv_linds_reset_injected(self._h, _lib.ptr(m), _lib.ptr(idx), _lib.ptr(self._obs),
                                                    _lib.ptr(self._cmd), _lib.ptr(self._error)))
        self.need_reset = False
        return self._o(self._user_obs(self._obs)), self._infos(False)

    def _check_step(self):
        if (not self.task_set) or self.need_reset:
            raise Exception("Must \"set_task\" and \"reset\" before doing any actions")   # linds_env.py:134-135

    _STEP_OUTPUTS = ("_obs", "_reward", "_term", "_trunc", "_cmd", "_error")   # all fully written by a step

    def _make_step_cache(self):
        return dict(key=(self._obs.data_ptr(), self._fobs.data_ptr(), self._done.data_ptr()),
                    args=tuple(C.c_void_p(t.data_ptr()) for t in (self._obs, self._reward, self._term, self._trunc, self._cmd,
                                                                  self._error, self._fobs, self._steps, self._done)),
                    ints=tuple(t.data_ptr() for t in (self._obs, self._reward, self._term, self._trunc, self._cmd,
                                                      self._error, self._fobs, self._steps, self._done)),
                    obs=self._user_obs(self._obs), cmd=self._user_obs(self._cmd), fobs=self._user_obs(self._fobs),
                    term_b=self._term.view(torch.bool), trunc_b=self._trunc.view(torch.bool),
                    done_b=self._done.view(torch.bool))

    def _fresh_final_obs(self):
        """the step writes final_obs rows of FINISHED envs only (64 B per env-step that ~93 % of the envs do not need):
        copy=True / to_numpy hand out zero rows elsewhere, as before; with copy=False the rows of unfinished envs keep
        whatever an earlier episode end left there — read them under info["_final_obs"]"""
        if self.copy and not self.to_numpy:
            self._fobs = torch.zeros_like(self._fobs)
        elif self.to_numpy:
            self._fobs.zero_()

    def _ret(self):
        if self.lean_infos:      # captured loops: nothing that needs a launch of its own (no `steps`, no `_final_obs` mask)
            infos = {"command": self._of(self._user_obs(self._cmd)), "error": self._of(self._error)}
            if self.autoreset_mode == "same_step":
                infos["final_obs"] = self._of(self._user_obs(self._fobs))
        else:
            infos = self._infos(True, fresh=True)
        return (self._of(self._user_obs(self._obs)), self._of(self._reward),
                self._obf(self._term), self._obf(self._trunc), infos)

    def step(self, actions):
        self._check_step()
        a = self._action(actions)
        slab = self.copy and not self.to_numpy and self._path_name != "scalar"
        if not slab:
            self._renew(*self._STEP_OUTPUTS)
            self._fresh_final_obs()
        if self._path_name == "scalar":      # the test kernel: steps and the done mask by a launch / an op of their own
            _lib.check(self.lib.xv_linds_step(self._h, _lib.ptr(a), _lib.ptr(self._obs), _lib.ptr(self._reward),
                                              _lib.ptr(self._term), _lib.ptr(self._trunc), _lib.ptr(self._cmd),
                                              _lib.ptr(self._error), _lib.ptr(self._fobs),
                                              AUTORESET[self.autoreset_mode]))
            return self._ret()
        if not self.copy and not self.to_numpy:      # persistent outputs: pointers and views are made once
            c = self._step_cache
            if c is None or c["key"] != (self._obs.data_ptr(), self._fobs.data_ptr(), self._done.data_ptr()):
                c = self._step_cache = self._make_step_cache()
            f = _lib.fast()
            if f is not None:
                rc = f.icall(self._fn_step_info(), self._h.value, a.data_ptr(), *c["ints"], AUTORESET[self.autoreset_mode])
                if rc:
                    _lib.check(rc)
            else:
                _lib.check(self.lib.xv_linds_step_info(self._h, C.c_void_p(a.data_ptr()), *c["args"], AUTORESET[self.autoreset_mode]))
            infos = {"steps": self._steps, "command": c["cmd"], "error": self._error}
            if self.autoreset_mode == "same_step":
                infos["final_obs"] = c["fobs"]
                infos["_final_obs"] = c["done_b"]
            return c["obs"], self._reward, c["term_b"], c["trunc_b"], infos
        if slab:
            # copy=True without copies, allocations or new tensor objects: the step writes every output of every env into the next
            # set of a slab made for 64 steps (vector.OutputSlabs; recycled once nobody can reach it); final_obs — written for
            # finished envs only — is zero-filled once per slab, so unfinished envs read zero rows as before
            t, p, pi = self._slab_next3()
            f = _lib.fast()
            if f is not None:      # the CPython trampoline (csrc/xvfast.c): the same call without ctypes' marshalling
                rc = f.icall(self._fn_step_info(), self._h.value, a.data_ptr(), *pi, AUTORESET[self.autoreset_mode])
                if rc:
                    _lib.check(rc)
            else:
                _lib.check(self.lib.xv_linds_step_info(self._h, C.c_void_p(a.data_ptr()), *p, AUTORESET[self.autoreset_mode]))
            self._obs, self._cmd, self._error = t["obs"], t["cmd"], t["error"]      # (reset() and the accessors read these)
            infos = {"steps": t["steps"], "command": t["cmd_u"], "error": t["error"]}
            if self.autoreset_mode == "same_step":
                infos["final_obs"] = t["fobs_u"]
                infos["_final_obs"] = t["done"]
            return t["obs_u"], t["reward"], t["term"], t["trunc"], infos
        # ONE launch: the step kernel writes info["steps"] and the terminated | truncated mask itself (xv_linds_step_info)
        self._renew("_steps", "_done")
        _lib.check(self.lib.xv_linds_step_info(self._h, _lib.ptr(a), _lib.ptr(self._obs), _lib.ptr(self._reward),
                                               _lib.ptr(self._term), _lib.ptr(self._trunc), _lib.ptr(self._cmd),
                                               _lib.ptr(self._error), _lib.ptr(self._fobs), _lib.ptr(self._steps),
                                               _lib.ptr(self._done), AUTORESET[self.autoreset_mode]))
        infos = {"steps": self._of(self._steps), "command": self._of(self._user_obs(self._cmd)), "error": self._of(self._error)}
        if self.autoreset_mode == "same_step":
            infos["final_obs"] = self._of(self._user_obs(self._fobs))
            infos["_final_obs"] = self._obf(self._done)
        return (self._of(self._user_obs(self._obs)), self._of(self._reward), self._obf(self._term), self._obf(self._trunc), infos)

    _FN_STEP_INFO = [None]

    def _fn_step_info(self):
        if self._FN_STEP_INFO[0] is None:
            self._FN_STEP_INFO[0] = _lib.fn_address("xv_linds_step_info")
        return self._FN_STEP_INFO[0]

    def _slab_next(self):
        return self._slab_next3()[:2]

    def _slab_next3(self):
        """copy=True: the next output set of a 64-step slab -> (dict of tensors incl. the user's columns obs_u / cmd_u / fobs_u,
        pointers in the order obs, reward, term, trunc, cmd, error, fobs, steps, done — as ctypes objects and as ints)"""
        if self._slabs is None:
            f32, u8, i32, NO = torch.float32, torch.uint8, torch.int32, self.NO
            uo = self.user_dims[1]
            cut = (lambda t: t[:, :uo]) if uo != NO else (lambda t: t)
            self._slabs = OutputSlabs([("obs", f32, (NO,)), ("cmd", f32, (NO,)), ("fobs", f32, (NO,)), ("reward", f32, ()),
                                       ("error", f32, ()), ("steps", i32, ()), ("term", u8, ()), ("trunc", u8, ()), ("done", u8, ())],
                                      self.num_envs, self.device, K=64, as_bool=("term", "trunc", "done"),
                                      order=("obs", "reward", "term", "trunc", "cmd", "error", "fobs", "steps", "done"),
                                      derived={"obs_u": ("obs", cut), "cmd_u": ("cmd", cut), "fobs_u": ("fobs", cut)},
                                      zero_on_refill=("fobs",))
        return self._slabs.next3()

    def step_injected(self, actions, z, init_index):
        """Parity hook: z float[NS, N] standard normals (process noise), init_index int[N] (used on reset)."""
        self._check_step()
        a = self._action(actions)
        z = self._dev(z, torch.float32)
        if z.shape[0] != self.NS:
            p = torch.zeros((self.NS, self.num_envs), dtype=torch.float32, device=self.device)
            p[:z.shape[0]] = z
            z = p
        idx = self._dev(init_index, torch.int32)
        self._renew(*self._STEP_OUTPUTS)
        self._fresh_final_obs()
        _lib.check(self.lib.xv_linds_step_injected(
            self._h, _lib.ptr(a), _lib.ptr(z.contiguous()), _lib.ptr(idx), _lib.ptr(self._obs),
            _lib.ptr(self._reward), _lib.ptr(self._term), _lib.ptr(self._trunc), _lib.ptr(self._cmd),
            _lib.ptr(self._error), _lib.ptr(self._fobs), AUTORESET[self.autoreset_mode]))
        return self._ret()

    def step_many(self, n_steps, actions, out=None):
        """n_steps vector steps issued from C (xv_linds_step_many): actions float32[P, N, na] is a ring of P action sets,
        step k uses slot k % P and writes slot k % P of the returned dict of [P, N(, NO)] device tensors.  Equals n_steps
        calls of step(); `final_obs` rows are written for finished envs only."""
        self._check_step()
        a = self._dev(actions, torch.float32)
        P = int(a.shape[0])
        na_user = self.user_dims[0]
        if a.shape != (P, self.num_envs, na_user):
            raise AssertionError(f"Action shape mismatch: expected {(P, self.num_envs, na_user)}, got {tuple(a.shape)}")
        if na_user != self.NA:
            p = torch.zeros((P, self.num_envs, self.NA), dtype=torch.float32, device=self.device)
            p[..., :na_user] = a
            a = p
        a = a.contiguous()
        d, n = self.device, self.num_envs
        if out is None:
            out = dict(obs=torch.empty((P, n, self.NO), dtype=torch.float32, device=d),
                       reward=torch.empty((P, n), dtype=torch.float32, device=d),
                       terminated=torch.empty((P, n), dtype=torch.uint8, device=d),
                       truncated=torch.empty((P, n), dtype=torch.uint8, device=d),
                       command=torch.empty((P, n, self.NO), dtype=torch.float32, device=d),
                       error=torch.empty((P, n), dtype=torch.float32, device=d),
                       final_obs=torch.zeros((P, n, self.NO), dtype=torch.float32, device=d))
        _lib.check(self.lib.xv_linds_step_many(self._h, int(n_steps), P, _lib.ptr(a), _lib.ptr(out["obs"]),
                                               _lib.ptr(out["reward"]), _lib.ptr(out["terminated"]),
                                               _lib.ptr(out["truncated"]), _lib.ptr(out["command"]), _lib.ptr(out["error"]),
                                               _lib.ptr(out.get("final_obs")), AUTORESET[self.autoreset_mode]))
        return out

    def rollout(self, actions, out=None, with_info=True):
        """Fused open-loop roll-out: actions float32[T, N, na] -> dict of [T, N(, NO)] device tensors, one launch with the
        state resident in registers between the steps.  Equals T calls of step() with SAME_STEP auto-reset, bit for bit."""
        self._check_step()
        a = self._dev(actions, torch.float32)
        T = int(a.shape[0])
        na_user = self.user_dims[0]
        if a.shape != (T, self.num_envs, na_user):
            raise AssertionError(f"Action shape mismatch: expected {(T, self.num_envs, na_user)}, got {tuple(a.shape)}")
        if na_user != self.NA:
            p = torch.zeros((T, self.num_envs, self.NA), dtype=torch.float32, device=self.device)
            p[..., :na_user] = a
            a = p
        a = a.contiguous()
        d, n = self.device, self.num_envs
        if out is None:
            out = dict(obs=torch.empty((T, n, self.NO), dtype=torch.float32, device=d),
                       reward=torch.empty((T, n), dtype=torch.float32, device=d),
                       terminated=torch.empty((T, n), dtype=torch.uint8, device=d),
                       truncated=torch.empty((T, n), dtype=torch.uint8, device=d))
            if with_info:
                out.update(command=torch.empty((T, n, self.NO), dtype=torch.float32, device=d),
                           error=torch.empty((T, n), dtype=torch.float32, device=d),
                           final_obs=torch.zeros((T, n, self.NO), dtype=torch.float32, device=d))   # written for
                # finished envs only; a caller-supplied `out` keeps its other rows
        _lib.check(self.lib.xv_linds_rollout(self._h, T, _lib.ptr(a), _lib.ptr(out["obs"]), _lib.ptr(out["reward"]),
                                             _lib.ptr(out["terminated"]), _lib.ptr(out["truncated"]),
                                             _lib.ptr(out.get("command")), _lib.ptr(out.get("error")),
                                             _lib.ptr(out.get("final_obs"))))
        return out

    @property
    def state(self):
        """env.state (linds_env.py:185-187): float[N, NS]"""
        x = torch.empty((self.NS, self.num_envs), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.xv_linds_get_state(self._h, _lib.ptr(x), None, None))
        return self._out(x.t().contiguous())

    def get_future_inner_cmds(self, K):
        """the next K inner commands of every env, starting with the one tracked at the next step
        (linds_env.py:171-183): float32[N, K, NO] on the host; command(t) = cmd_fn(t) * target_valid."""
        st = self._steps_now().cpu().numpy()
        t = self._tab
        task = t["env_task"].cpu().numpy()
        ints = t["ints"].cpu().numpy()[task]
        delay, nf = ints[:, 1], ints[:, 3]
        tt = st[:, None] - delay[:, None] + np.arange(K)[None, :]                     # [N, K]
        om = t["four_omega"].cpu().numpy()[task]                                       # [N, KMAX]
        per = t["four_period"].cpu().numpy()[task]
        co = t["four_coef"].cpu().numpy()[task].astype(np.float64)                     # [N, KMAX, NO, 2]
        ang = om[:, None, :] * (tt[:, :, None] / per[:, None, None])                   # [N, K, KMAX]
        live = (np.arange(om.shape[1])[None, :] < nf[:, None])[:, None, :, None]
        dyn = ((np.sin(ang)[..., None] * co[:, None, :, :, 0] + np.cos(ang)[..., None] * co[:, None, :, :, 1]) * live).sum(2)
        static = t["cmd0"].cpu().numpy()[task][:, None, :].astype(np.float64)
        cmd = np.where((nf > 0)[:, None, None], dyn, static) * t["valid"].cpu().numpy()[task][:, None, :]
        return cmd[..., :self.user_dims[1]].astype(np.float32)

    def get_state(self):
        x = torch.empty((self.NS, self.num_envs), dtype=torch.float32, device=self.device)
        st = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        nr = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.xv_linds_get_state(self._h, _lib.ptr(x), _lib.ptr(st), _lib.ptr(nr)))
        return x, st, nr

    def set_state(self, x=None, steps=None, need_reset=None):
        """x float[NS_user or NS, N] component-major"""
        xs = None
        if x is not None:
            xs = self._dev(x, torch.float32)
            if xs.shape[0] != self.NS:
                p = torch.zeros((self.NS, self.num_envs), dtype=torch.float32, device=self.device)
                p[:xs.shape[0]] = xs
                xs = p
            xs = xs.contiguous()
        st = None if steps is None else self._dev(steps, torch.int32)
        nr = None if need_reset is None else self._dev(need_reset, torch.uint8)
        _lib.check(self.lib.xv_linds_set_state(self._h, _lib.ptr(xs), _lib.ptr(st), _lib.ptr(nr)))
        self.engine.sync()
        self.need_reset = False

    def close_extras(self, **kwargs):
        if self._h is not None:
            self.lib.xv_linds_destroy(self._h)
            self._h = None
        self._tab = None
