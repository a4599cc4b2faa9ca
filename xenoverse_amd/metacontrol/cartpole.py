"""CartPoleVecEnv — N domain-randomised CartPoles per kernel launch.

Mirrors xenoverse/metacontrol/random_cartpole.py: sample_cartpole :13-29, RandomCartPoleEnv.__init__ :33-44
(frameskip, reset_bounds_scale), set_task :46-50, step :52-61, reset :63-75; registered id
`random-cartpole-v0` uses frameskip=1, reset_bounds_scale=[0.45, 0.90, 0.13, 1.0] (metacontrol/__init__.py:20-26).
The physics is gymnasium's CartPoleEnv (not installed here): parity unpinned, see include/xeno.h.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..engine import AUTORESET
from ..spaces import Box, Discrete
from ..vector import VectorEnv


def _versatile(setting, default_range, default_value, rng):
    # utils/tools.py:47-54
    if isinstance(setting, (tuple, list)):
        assert len(setting) == 2, f"Setting must be a tuple or list of length 2, got {len(setting)}"
        return float(rng.uniform(setting[0], setting[1]))
    if setting:
        return float(rng.uniform(default_range[0], default_range[1]))
    return default_value


def sample_cartpole(gravity_scope=True, masscart_scope=True, masspole_scope=True, length_scope=True, seed=None):
    """Same ranges and keys as the reference sampler (random_cartpole.py:13-29).  The reference is
    time-seeded and not reproducible; `seed` (optional) makes this one reproducible."""
    rng = np.random.RandomState(seed)
    return {"gravity": _versatile(gravity_scope, (1, 11), 9.8, rng),
            "masscart": _versatile(masscart_scope, (0.5, 2.0), 1.0, rng),
            "masspole": _versatile(masspole_scope, (0.05, 0.20), 0.1, rng),
            "length": _versatile(length_scope, (0.25, 1.0), 0.5, rng)}


class CartPoleVecEnv(VectorEnv):
    def __init__(self, num_envs, frameskip=5, reset_bounds_scale=(0.45, 0.90, 0.13, 1.0), max_steps=0,
                 device="cuda:0", seed=0, env_id_base=0, autoreset_mode="same_step", to_numpy=False,
                 engine=None, copy=True):
        super().__init__(num_envs, device=device, seed=seed, env_id_base=env_id_base,
                         autoreset_mode=autoreset_mode, to_numpy=to_numpy, engine=engine, copy=copy)
        self.frameskip = int(frameskip)
        rs = list(reset_bounds_scale)
        assert len(rs) == 4, "reset_bounds_scale should be a list of 4 elements"
        self.reset_bounds_scale = np.asarray(rs, np.float64)
        self.max_steps = int(max_steps)
        hi = np.array([4.8, np.finfo(np.float32).max, 0.41887903, np.finfo(np.float32).max], np.float32)
        self._set_spaces(Box(-hi, hi, dtype=np.float32), Discrete(2))
        self._h = None

    def set_task(self, tasks, env_task_index=None):
        if isinstance(tasks, dict):
            tasks = [tasks]
        params = np.array([[t["gravity"], t["masscart"], t["masspole"], t["length"]] for t in tasks], np.float64)
        d = self.device
        n_task = len(tasks)
        if env_task_index is None:
            if self.num_envs % n_task != 0:
                raise ValueError("num_envs is not a multiple of the task count; pass env_task_index")
            env_task = torch.arange(self.num_envs, device=d, dtype=torch.int32) // (self.num_envs // n_task)
        else:
            env_task = self._dev(env_task_index, torch.int32)
            if env_task.shape != (self.num_envs,) or int(env_task.min()) < 0 or int(env_task.max()) >= n_task:
                raise ValueError("env_task_index must be (num_envs,) with entries in [0, n_task)")
        self._tab = dict(params=torch.from_numpy(params).to(d), scale=torch.from_numpy(self.reset_bounds_scale).to(d),
                         env_task=env_task.contiguous())
        if self._h is not None:
            self.lib.xv_cartpole_destroy(self._h)
        h = C.c_void_p()
        _lib.check(self.lib.xv_cartpole_create(self.engine.handle, self.num_envs, n_task, self.frameskip,
                                               self.max_steps, _lib.ptr(self._tab["params"]),
                                               _lib.ptr(self._tab["scale"]), _lib.ptr(self._tab["env_task"]),
                                               C.byref(h)))
        self._h = h
        n = self.num_envs
        self._obs = torch.zeros((n, 4), dtype=torch.float32, device=d)
        self._fobs = torch.zeros((n, 4), dtype=torch.float32, device=d)
        self._reward = torch.zeros(n, dtype=torch.float32, device=d)
        self._term = torch.zeros(n, dtype=torch.uint8, device=d)
        self._trunc = torch.zeros(n, dtype=torch.uint8, device=d)
        self._done = torch.zeros(n, dtype=torch.uint8, device=d)       # terminated | truncated, written by the step launch
        self._step_cache = None      # (copy=False steps cache pointers and views of these buffers)
        self.task_set = True
        self.need_reset = True

    def reset(self, *, seed=None, options=None):
        self._require_task()
        if seed is not None:
            self.engine.tick = (int(seed) & 0xFFFFFFFF) << 24
        mask = None
        if options is not None and options.get("reset_mask") is not None:
            mask = self._dev(options["reset_mask"], torch.uint8)
        self._detach("_obs")
        _lib.check(self.lib.xv_cartpole_reset(self._h, _lib.ptr(mask), _lib.ptr(self._obs)))
        self.need_reset = False
        return self._o(self._obs), {}

    def reset_injected(self, u, mask=None):
        self._require_task()
        u = self._dev(u, torch.float64)
        m = None if mask is None else self._dev(mask, torch.uint8)
        self._detach("_obs")
        _lib.check(self.lib.xv_cartpole_reset_injected(self._h, _lib.ptr(m), _lib.ptr(u), _lib.ptr(self._obs)))
        self.need_reset = False
        return self._o(self._obs)

    def _ret(self, from_launch=False):
        """from_launch: the step wrote `_done` itself (xv_cartpole_step_info)"""
        infos = {}
        if self.autoreset_mode == "same_step":
            infos["final_obs"] = self._of(self._fobs)
            if not self.lean_infos:       # flags are 0 / 1 bytes: a view, no conversion
                infos["_final_obs"] = self._obf(self._done) if from_launch else \
                    self._out((self._term | self._trunc).view(torch.bool))
        return (self._of(self._obs), self._of(self._reward), self._obf(self._term),
                self._obf(self._trunc), infos)

    def step(self, actions):
        if (not self.task_set) or self.need_reset:
            raise Exception("Must \"set_task\" and \"reset\" before doing any actions")
        a = self._dev(actions, torch.int32)
        assert a.shape == (self.num_envs,)
        if not self.copy and not self.to_numpy:      # persistent outputs: pointers and views are made once
            c = self._step_cache
            if c is None or c["key"] != (self._obs.data_ptr(), self._fobs.data_ptr(), self._done.data_ptr()):
                c = self._step_cache = dict(
                    key=(self._obs.data_ptr(), self._fobs.data_ptr(), self._done.data_ptr()),
                    args=tuple(C.c_void_p(t.data_ptr()) for t in (self._obs, self._reward, self._term, self._trunc, self._fobs,
                                                                  self._done)),
                    term_b=self._term.view(torch.bool), trunc_b=self._trunc.view(torch.bool), done_b=self._done.view(torch.bool))
            _lib.check(self.lib.xv_cartpole_step_info(self._h, C.c_void_p(a.data_ptr()), *c["args"], AUTORESET[self.autoreset_mode]))
            infos = {}
            if self.autoreset_mode == "same_step":
                infos["final_obs"] = self._fobs
                if not getattr(self, "lean_infos", False):
                    infos["_final_obs"] = c["done_b"]
            return self._obs, self._reward, c["term_b"], c["trunc_b"], infos
        self._renew("_obs", "_reward", "_term", "_trunc", "_fobs", "_done")      # all fully written by the step
        _lib.check(self.lib.xv_cartpole_step_info(self._h, _lib.ptr(a), _lib.ptr(self._obs), _lib.ptr(self._reward),
                   _lib.ptr(self._term), _lib.ptr(self._trunc), _lib.ptr(self._fobs), _lib.ptr(self._done),
                   AUTORESET[self.autoreset_mode]))      # one launch: the done mask comes from the step kernel
        return self._ret(from_launch=True)

    def rollout(self, actions, out=None):
        """Fused open-loop roll-out: actions int32[T, N] -> dict of [T, N(, 4)] device tensors from one launch with the
        state in registers between the steps; equals T calls of step() bit for bit."""
        if (not self.task_set) or self.need_reset:
            raise Exception("Must \"set_task\" and \"reset\" before doing any actions")
        a = self._dev(actions, torch.int32).contiguous()
        T, n, d = int(a.shape[0]), self.num_envs, self.device
        assert a.shape == (T, n)
        if out is None:
            out = dict(obs=torch.empty((T, n, 4), dtype=torch.float32, device=d),
                       reward=torch.empty((T, n), dtype=torch.float32, device=d),
                       terminated=torch.empty((T, n), dtype=torch.uint8, device=d),
                       truncated=torch.empty((T, n), dtype=torch.uint8, device=d),
                       final_obs=torch.empty((T, n, 4), dtype=torch.float32, device=d))
        _lib.check(self.lib.xv_cartpole_rollout(self._h, T, _lib.ptr(a), _lib.ptr(out["obs"]), _lib.ptr(out["reward"]),
                                          _lib.ptr(out["terminated"]), _lib.ptr(out["truncated"]),
                                          _lib.ptr(out.get("final_obs")), AUTORESET[self.autoreset_mode]))
        return out

    def step_injected(self, actions, u_reset):
        a = self._dev(actions, torch.int32)
        u = self._dev(u_reset, torch.float64)
        self._renew("_obs", "_reward", "_term", "_trunc", "_fobs")
        _lib.check(self.lib.xv_cartpole_step_injected(self._h, _lib.ptr(a), _lib.ptr(u), _lib.ptr(self._obs),
                                                      _lib.ptr(self._reward), _lib.ptr(self._term),
                                                      _lib.ptr(self._trunc), _lib.ptr(self._fobs),
                                                      AUTORESET[self.autoreset_mode]))
        return self._ret()

    def get_state(self):
        n, d = self.num_envs, self.device
        s = torch.empty((4, n), dtype=torch.float64, device=d)
        st = torch.empty(n, dtype=torch.int32, device=d)
        nr = torch.empty(n, dtype=torch.uint8, device=d)
        _lib.check(self.lib.xv_cartpole_get_state(self._h, _lib.ptr(s), _lib.ptr(st), _lib.ptr(nr)))
        return s, st, nr

    def set_state(self, state=None, steps=None, need_reset=None):
        s = None if state is None else self._dev(state, torch.float64)
        st = None if steps is None else self._dev(steps, torch.int32)
        nr = None if need_reset is None else self._dev(need_reset, torch.uint8)
        _lib.check(self.lib.xv_cartpole_set_state(self._h, _lib.ptr(s), _lib.ptr(st), _lib.ptr(nr)))
        self.engine.sync()
        self.need_reset = False

    def close_extras(self, **kwargs):
        if self._h is not None:
            self.lib.xv_cartpole_destroy(self._h)
            self._h = None
