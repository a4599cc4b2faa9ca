"""AnyMDPVecEnv — N AnyMDP environments stepped per kernel launch on one MI355X.

Mirrors the reference's per-env interface (xenoverse/anymdp/anymdp_env.py: AnyMDPEnv.set_task :32-79,
reset :81-90, step :112-132, properties :134-165) behind gymnasium's VectorEnv surface.  Same task dicts,
same exception types/messages for misuse, same info keys (`steps`, `reward_gt`, optional `transition_gt`).
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..engine import AUTORESET
from ..spaces import Discrete, MultiDiscrete
from ..vector import OutputSlabs, VectorEnv
from .tables import build_obs_tables, build_tables, build_tables_device, device_buildable

_TABLE_KEYS = ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")
_TABLE_DTYPES = dict(rows=torch.float64, state_map=torch.int32, term_mask=torch.int64,
                     s0_cdf=torch.float64, s0_ids=torch.int32, max_steps=torch.int32)


class AnyMDPVecEnv(VectorEnv):
    # bucket_lines="auto": the lines are built at set_task when they take no more than 1 GiB, or no more than an eighth of the
    # device memory that is free at that moment, up to 16 GiB (an MI355X has 288 GB: 16,384 tasks of 64 x 8 on an empty one)
    AUTO_BUCKET_BYTES = 1 << 30
    AUTO_BUCKET_SHARE = 0.125
    AUTO_BUCKET_CAP = 16 << 30

    def __init__(self, num_envs, max_steps=5000, device="cuda:0", seed=0, env_id_base=0,
                 autoreset_mode="same_step", to_numpy=False, engine=None, with_transition_gt=False, copy=True,
                 bucket_lines="auto", device_tables=True):
        """`max_steps` is kept for signature parity with AnyMDPEnv(max_steps); as in the reference it is
        overridden by each task's own `max_steps` at set_task (anymdp_env.py:23-34).

        copy (as gymnasium's SyncVectorEnv(copy=...)): True returns fresh tensors from every step().  False returns
        views of two engine-owned output sets used alternately — what step() returned stays valid until the step
        after the next one — and takes the per-step host cost from ~50 us (clones, bool conversions) to a launch, a
        16-KB copy and one fused op (`scripts/bench_python_step.py`).

        bucket_lines: "auto" (default) — at set_task the engine takes the census of 16 bucket lines per row and builds them
        when its AUTO rule would use them AND they take at most 1 GiB or an eighth of the free device memory up to 16 GiB
        (n_task * S * A * 2 KiB: 1,024 ... 16,384 tasks of 64 x 8): a step then reads one table line instead of two, same results; "off" — never without set_search(..., n_bucket=);
        an int — that many buckets per row whatever the size (as set_search("auto", n_bucket=int))."""
        super().__init__(num_envs, device=device, seed=seed, env_id_base=env_id_base,
                         autoreset_mode=autoreset_mode, to_numpy=to_numpy, engine=engine, copy=copy)
        self.max_steps = max_steps
        self.with_transition_gt = bool(with_transition_gt)
        self.copy = bool(copy)
        self.bucket_lines = bucket_lines
        self.device_tables = bool(device_tables)      # set_task forms the row records of raw task dicts on the device
        self._ring = None
        self._tok_cache = None
        self._set_spaces(Discrete(1), Discrete(1))   # placeholders until set_task, as in the reference
        self._h = None
        self._many_cache = None
        self._tab = None
        self._views = []            # sub-batch envs made by split(): they borrow this env's tables
        self._parent = None

    # ---- set_task ---------------------------------------------------------------------------------
    def set_task(self, tasks, env_task_index=None):
        """tasks: one reference task dict, a list of them, or a dict of prebuilt tables (numpy arrays or
        device tensors, keys as xenoverse_amd.anymdp.tables.build_tables returns).  env_task_index[i] is
        the task of env i (default: envs split evenly and contiguously over tasks)."""
        obs_model = None
        table_type = None
        if isinstance(tasks, dict) and "rows" in tasks:
            tab = tasks
            if tab.get("obs_cdf") is not None:      # prebuilt POMDP / MTPOMDP tables (device_sampler.sample_tasks_device)
                obs_model = (tab["obs_cdf"], int(tab["n_obs"]), int(tab["d_obs"]), int(tab["d_act"]))
                table_type = tab.get("task_type", "MTPOMDP" if (obs_model[2] > 1 or obs_model[3] > 1) else "POMDP")
        else:
            if isinstance(tasks, dict):
                tasks = [tasks]
            ttype = tasks[0].get("task_type", "MDP")
            if ttype not in ("MDP", "POMDP", "MTPOMDP"):
                raise NotImplementedError(f"Unknown task type: {ttype}")   # anymdp_env.py:45-46
            # raw task tensors of one shape: the row records are formed on the device (xv_anymdp_build_rows), bit for bit
            # what the host builder forms and ~50x faster; anything else (ragged shapes, bandits) on the host
            tab = build_tables_device(tasks, self.engine) if (self.device_tables and device_buildable(tasks)) else build_tables(tasks)
            if ttype != "MDP":
                obs_model = build_obs_tables(tasks, tab["S"])
        self.task_type = "MDP" if obs_model is None else (table_type or tasks[0]["task_type"])
        dev = {}
        for k in _TABLE_KEYS:
            v = tab[k]
            if torch.is_tensor(v):
                dev[k] = v.to(self.device).contiguous()
            else:
                v = np.ascontiguousarray(v)
                if v.dtype == np.uint64:
                    v = v.view(np.int64)
                dev[k] = torch.from_numpy(v).to(self.device)
            assert dev[k].dtype == _TABLE_DTYPES[k], (k, dev[k].dtype)
        S, A, s0_max = int(tab["S"]), int(tab["A"]), int(tab["s0_max"])
        n_task = int(dev["max_steps"].shape[0])
        from .tables import row_lines
        want = dict(rows=(n_task, S, A, row_lines(S), 16), state_map=(n_task, S), term_mask=(n_task, (S + 63) // 64),
                    s0_cdf=(n_task, s0_max), s0_ids=(n_task, s0_max), max_steps=(n_task,))
        for k, shp in want.items():      # the device reads these extents: a short table is an out-of-bounds read
            if tuple(dev[k].shape) != shp:
                raise ValueError("table %r has shape %s, expected %s" % (k, tuple(dev[k].shape), shp))
        if env_task_index is None:
            if self.num_envs % n_task != 0:
                raise ValueError("num_envs (%d) is not a multiple of the task count (%d); pass "
                                 "env_task_index" % (self.num_envs, n_task))
            per = self.num_envs // n_task
            env_task = torch.arange(self.num_envs, device=self.device, dtype=torch.int32) // per
        else:
            env_task = self._dev(env_task_index, torch.int32)
            if env_task.shape != (self.num_envs,):
                raise ValueError("env_task_index must have shape (num_envs,)")
            lo, hi = int(env_task.min()), int(env_task.max())
            if lo < 0 or hi >= n_task:
                raise ValueError("env_task_index out of range")
        dev["env_task"] = env_task.contiguous()
        if self._parent is not None:
            raise ValueError("set_task on a sub-batch made by split(): set the task on the env it was split from")
        self._drop_views()
        if self._h is not None:
            self.lib.xv_anymdp_destroy(self._h)
            self._h = None
        h = C.c_void_p()
        _lib.check(self.lib.xv_anymdp_create(
            self.engine.handle, self.num_envs, n_task, S, A, s0_max,
            _lib.ptr(dev["rows"]), _lib.ptr(dev["state_map"]),
            _lib.ptr(dev["term_mask"]), _lib.ptr(dev["s0_cdf"]), _lib.ptr(dev["s0_ids"]),
            _lib.ptr(dev["max_steps"]), _lib.ptr(dev["env_task"]), C.byref(h)))
        self._h = h
        self._many_cache = None
        self._n_bucket = 0
        self._tab = dev       # keeps the borrowed device tables alive
        self.S, self.A, self.s0_max, self.n_task = S, A, s0_max, n_task
        ns = int(np.max(tab["obs_space"])) if "obs_space" in tab else S
        self.ns, self.na = ns, A
        self._set_spaces(Discrete(ns), Discrete(A))
        self._tok = None
        if obs_model is not None:      # POMDP / MTPOMDP (anymdp_env.py:39-44)
            obs_cdf, n_obs, d_obs, d_act = obs_model
            self._tab["obs_cdf"] = obs_cdf.to(self.device, torch.float64).contiguous() if torch.is_tensor(obs_cdf) else \
                torch.from_numpy(np.ascontiguousarray(obs_cdf, np.float64)).to(self.device)
            if tuple(self._tab["obs_cdf"].shape) != (n_task, d_obs, S, n_obs):
                raise ValueError("obs_cdf has shape %s, expected %s" % (tuple(self._tab["obs_cdf"].shape), (n_task, d_obs, S, n_obs)))
            _lib.check(self.lib.xv_anymdp_set_observation_model(self._h, n_obs, d_obs, d_act,
                                                                _lib.ptr(self._tab["obs_cdf"])))
            self._tok = (d_obs, d_act)
            self.no, self.do, self.da = n_obs, d_obs, d_act
        self._make_buffers()
        self.task_set = True
        self.need_reset = True
        if self.bucket_lines != "off":      # memory for speed, within a small budget unless the caller named a bucket count
            nb = 16 if self.bucket_lines == "auto" else int(self.bucket_lines)
            budget = self.AUTO_BUCKET_BYTES
            # transition lines + the observation lines a POMDP / multi-token task adds beside them (same 128-byte lines)
            need = n_task * S * A * nb * 128 + (n_task * self._tok[0] * S * nb * 128 if self._tok is not None else 0)
            if self.bucket_lines == "auto" and need > budget:
                free, _ = torch.cuda.mem_get_info(self.device)
                budget = max(budget, min(self.AUTO_BUCKET_CAP, int(free * self.AUTO_BUCKET_SHARE)))
            self.bucket_budget = {"bytes_needed": need, "budget": budget, "within": self.bucket_lines != "auto" or need <= budget}
            if self.bucket_budget["within"]:
                try:
                    self.set_search("auto", n_bucket=nb)
                except _lib.XenoError as ex:
                    # tables the fence layout does not serve (s0_max > 4, ...): the per-lane search.  Anything else — a HIP
                    # error, no memory — is the caller's to see
                    if ex.code != _lib.XV_ERR_UNSUPPORTED:
                        raise

    def _make_buffers(self):
        """spaces for the task type and the output buffers step() / reset() write (set_task, and a view made by split())"""
        n, d, S, A = self.num_envs, self.device, self.S, self.A
        if self._tok is not None:
            d_obs, d_act = self._tok
            if self.task_type == "MTPOMDP":
                self._set_spaces(MultiDiscrete([self.no] * d_obs), MultiDiscrete([A] * d_act))
            else:
                self._set_spaces(Discrete(self.no), Discrete(A))
            self._tobs = torch.zeros((n, d_obs), dtype=torch.int32, device=d)
            self._tfobs = torch.full((n, d_obs), -1, dtype=torch.int32, device=d)
        self._tok_cache = None          # (copy=False token steps cache pointers and views of the buffers made here)
        self._slabs = None              # copy=True: output sets of 64 steps per allocation (made at the first step)
        self._obs = torch.zeros(n, dtype=torch.int32, device=d)
        self._reward = torch.zeros(n, dtype=torch.float32, device=d)
        self._reward_gt = torch.zeros(n, dtype=torch.float32, device=d)
        self._term = torch.zeros(n, dtype=torch.uint8, device=d)
        self._trunc = torch.zeros(n, dtype=torch.uint8, device=d)
        self._final_obs = torch.full((n,), -1, dtype=torch.int32, device=d)
        self._steps = torch.zeros(n, dtype=torch.int32, device=d)
        self._done = torch.zeros(n, dtype=torch.uint8, device=d)
        self._tgt = torch.zeros((n, S), dtype=torch.float64, device=d) if self.with_transition_gt else None
        self._ring = None
        if (not self.copy) and (not self.to_numpy) and self._tok is None and not self.with_transition_gt:
            self._ring, self._ring_pos = [], 0
            for _ in range(2):
                b = dict(obs=torch.zeros(n, dtype=torch.int32, device=d), reward=torch.zeros(n, dtype=torch.float32, device=d),
                         reward_gt=torch.zeros(n, dtype=torch.float32, device=d),
                         term=torch.zeros(n, dtype=torch.uint8, device=d), trunc=torch.zeros(n, dtype=torch.uint8, device=d),
                         final_obs=torch.full((n,), -1, dtype=torch.int32, device=d),
                         steps=torch.zeros(n, dtype=torch.int32, device=d), done=torch.zeros(n, dtype=torch.uint8, device=d))
                b["term_b"], b["trunc_b"], b["done_b"] = (b[k].view(torch.bool) for k in ("term", "trunc", "done"))
                b["args"] = tuple(C.c_void_p(b[k].data_ptr()) for k in
                                  ("obs", "reward", "reward_gt", "term", "trunc", "final_obs"))
                b["steps_p"] = C.c_void_p(b["steps"].data_ptr())
                b["done_p"] = C.c_void_p(b["done"].data_ptr())
                b["ints"] = tuple(x.value for x in b["args"]) + (b["steps"].data_ptr(), b["done"].data_ptr())
                self._ring.append(b)

    SEARCH = {"auto": 0, "binary": 1, "fence": 3, "bucket": 4}
    _SEARCH_NAME = {1: "binary", 3: "fence", 4: "bucket"}

    def set_search(self, mode, n_bucket=None):
        """Select how the categorical draw searches the CDF row (results are identical; see xeno.h).
        "bucket": builds n_task * S * A * n_bucket * 128 bytes of bucket lines (once; default 32 buckets) and makes a step
                  one table line in one dependent level; raises if they do not fit.
        "auto":   the engine decides per launch: the bucket search when its lines are built AND their census expects no more
                  draws per launch that a line cannot answer than `auto_limit` (0.5; 0.2 for cache-resident tables), else the fence search (two dependent lines), else
                  the per-lane binary search.  With n_bucket given, the lines are built first when — and only when —
                  the census (taken without allocating anything) says AUTO would use them and they fit the free memory.
        `effective_search` names what runs; `bucket_census()` has the numbers."""
        if self._parent is not None:
            raise ValueError("set_search on a sub-batch made by split(): it follows the env it was split from")
        if n_bucket is not None and getattr(self, "_n_bucket", 0) != n_bucket:
            self._drop_views()          # they borrow the lines this call may rebuild
        if mode == "bucket":
            n_bucket = 32 if n_bucket is None else n_bucket
            if getattr(self, "_n_bucket", 0) != n_bucket:
                _lib.check(self.lib.xv_anymdp_build_buckets(self._h, int(n_bucket)))
                self._n_bucket = n_bucket
        elif mode == "auto" and n_bucket is not None and getattr(self, "_n_bucket", 0) != n_bucket:
            cen = self.probe_buckets(n_bucket)
            free, _ = torch.cuda.mem_get_info(self.device)
            if cen["auto_uses_bucket"] and cen["bytes"] + (4 << 30) <= free:
                _lib.check(self.lib.xv_anymdp_build_buckets(self._h, int(n_bucket)))
                self._n_bucket = n_bucket
        _lib.check(self.lib.xv_anymdp_set_search(self._h, self.SEARCH[mode]))
        for v in self._views:
            _lib.check(self.lib.xv_anymdp_set_search(v._h, self.SEARCH[mode]))
        self._many_cache = None

    def probe_buckets(self, n_bucket=16):
        """census of the bucket lines these rows WOULD get (nothing allocated): xv_anymdp_probe_buckets -> dict"""
        cen = _lib.BucketCensus()
        _lib.check(self.lib.xv_anymdp_probe_buckets(self._h, int(n_bucket), C.byref(cen)))
        return cen.as_dict()

    def bucket_census(self):
        """census of the bucket lines that are built (all zero when none are)"""
        cen = _lib.BucketCensus()
        _lib.check(self.lib.xv_anymdp_bucket_census_get(self._h, C.byref(cen)))
        return cen.as_dict()

    @property
    def effective_search(self):
        """"binary" | "fence" | "bucket": what a step launches now (how AUTO resolves)"""
        return self._SEARCH_NAME[int(self.lib.xv_anymdp_effective_search(self._h))]

    @property
    def token_kernel(self):
        """POMDP / multi-token tasks: "cooperative" (bucket lines) or "per-lane": the kernel step() launches now"""
        return "cooperative" if int(self.lib.xv_anymdp_token_kernel(self._h)) == 1 else "per-lane"

    # ---- reset ------------------------------------------------------------------------------------
    def reset(self, *, seed=None, options=None):
        """gymnasium VectorEnv.reset.  options={"reset_mask": bool[N]} resets a subset (autoreset
        DISABLED); seed (int) re-keys this engine's Philox stream position deterministically."""
        self._require_task()
        if seed is not None:
            self.engine.tick = (int(seed) & 0xFFFFFFFF) << 24
        mask = None
        if options is not None and options.get("reset_mask") is not None:
            mask = self._dev(options["reset_mask"], torch.uint8)
        self._detach("_obs")
        if self._tok is not None:
            self._detach("_tobs")      # a masked reset writes part of it: never into the tensor the last step handed out
            _lib.check(self.lib.xv_anymdp_reset_tokens(self._h, _lib.ptr(mask), _lib.ptr(self._tobs)))
            self.need_reset = False
            self._sync_views()
            return self._tok_obs(self._tobs), {"steps": self._out(self._get_steps())}
        _lib.check(self.lib.xv_anymdp_reset(self._h, _lib.ptr(mask), _lib.ptr(self._obs)))
        self.need_reset = False
        self._sync_views()
        return self._out(self._obs.clone()), {"steps": self._out(self._get_steps())}

    def _sync_views(self):
        """after a reset of this env: its sub-batches (split) continue from this env's tick"""
        if self._views:
            t = self.engine.tick
            for v in self._views:
                v.engine.tick = t
                v.need_reset = False

    # ---- POMDP / MTPOMDP helpers ---------------------------------------------------------------------
    def _tok_obs(self, t):
        t = t.clone()
        return self._out(t[:, 0] if self.task_type == "POMDP" else t)

    def _tok_action(self, actions):
        a = self._dev(actions, torch.int32)
        d_act = self._tok[1]
        if a.dim() == 1 and d_act == 1:
            a = a[:, None]
        if a.shape != (self.num_envs, d_act):
            raise AssertionError(f"Action {tuple(a.shape)} is out of range")   # anymdp_env.py:117
        return a.contiguous()

    def _tok_make_cache(self):
        pomdp = self.task_type == "POMDP"
        return dict(key=(self._tobs.data_ptr(), self._tfobs.data_ptr()),
                    args=tuple(C.c_void_p(t.data_ptr()) for t in (self._tobs, self._reward, self._reward_gt, self._term,
                                                                  self._trunc, self._tfobs, self._steps, self._done)),
                    obs=self._tobs[:, 0] if pomdp else self._tobs, fobs=self._tfobs[:, 0] if pomdp else self._tfobs,
                    term_b=self._term.view(torch.bool), trunc_b=self._trunc.view(torch.bool),
                    done_b=self._done.view(torch.bool))

    def _tok_fresh(self):
        """before a token step: the step kernels write observation, rewards, flags and final_obs (-1 where the env goes on) of
        EVERY env, so fresh buffers are swapped in and handed out without copies (the previous call's tensors are never
        written again)"""
        self._renew("_tobs", "_reward", "_reward_gt", "_term", "_trunc", "_tfobs")

    def _tok_ret(self, from_launch=False):
        """from_launch: the step wrote `_steps` and `_done` itself (xv_anymdp_step_tokens_info)"""
        infos = {"steps": self._of(self._steps) if from_launch else self._out(self._get_steps()),
                 "reward_gt": self._of(self._reward_gt)}
        if self.autoreset_mode == "same_step":
            infos["final_obs"] = self._of(self._tfobs[:, 0] if self.task_type == "POMDP" else self._tfobs)
            infos["_final_obs"] = self._obf(self._done) if from_launch else \
                self._out((self._term | self._trunc).view(torch.bool))
        obs = self._tobs[:, 0] if self.task_type == "POMDP" else self._tobs
        return (self._of(obs), self._of(self._reward), self._obf(self._term), self._obf(self._trunc), infos)

    def reset_tokens_injected(self, u_reset, u_obs_reset, mask=None):
        self._require_task()
        ur = self._dev(u_reset, torch.float64)
        uo = self._dev(u_obs_reset, torch.float64)
        m = None if mask is None else self._dev(mask, torch.uint8)
        self._detach("_tobs")
        _lib.check(self.lib.xv_anymdp_reset_tokens_injected(self._h, _lib.ptr(m), _lib.ptr(ur), _lib.ptr(uo),
                                                            _lib.ptr(self._tobs)))
        self.need_reset = False
        return self._tok_obs(self._tobs)

    def step_tokens_injected(self, actions, u, z, u_obs, u_reset, u_obs_reset):
        """Parity hook for POMDP / MTPOMDP: u, z [d_act, N]; u_obs, u_obs_reset [d_obs, N]; u_reset [N]."""
        self._check_step()
        a = self._tok_action(actions)
        args = [self._dev(u, torch.float64), self._dev(z, torch.float32), self._dev(u_obs, torch.float64),
                self._dev(u_reset, torch.float64), self._dev(u_obs_reset, torch.float64)]
        self._tok_fresh()
        _lib.check(self.lib.xv_anymdp_step_tokens_injected(
            self._h, _lib.ptr(a), *[_lib.ptr(x) for x in args], _lib.ptr(self._tobs), _lib.ptr(self._reward),
            _lib.ptr(self._reward_gt), _lib.ptr(self._term), _lib.ptr(self._trunc), _lib.ptr(self._tfobs),
            AUTORESET[self.autoreset_mode]))
        return self._tok_ret()

    def reset_injected(self, u, mask=None):
        """Parity hook: the initial-state uniform is supplied per env (fp64 in [0,1))."""
        self._require_task()
        u = self._dev(u, torch.float64)
        m = None if mask is None else self._dev(mask, torch.uint8)
        self._detach("_obs")
        _lib.check(self.lib.xv_anymdp_reset_injected(self._h, _lib.ptr(m), _lib.ptr(u), _lib.ptr(self._obs)))
        self.need_reset = False
        return self._out(self._obs.clone())

    # ---- step -------------------------------------------------------------------------------------
    def _check_step(self):
        if (not self.task_set) or self.need_reset:
            # reference message, anymdp_env.py:93-94
            raise Exception("Must \"set_task\" and \"reset\" before doing any actions")
        if self._parent is not None or self._views:
            self._sync_tick()

    def _sync_tick(self):
        """An env and its sub-batches (split) step the SAME envs through different handles, each with a launch tick of its own;
        an env must never draw twice at one tick (Philox(seed, env id, tick): the second draw would repeat the first).  Rule:
        a launch of the env starts at max(its tick, its sub-batches' ticks); a launch of a sub-batch at max(its tick, the
        env's tick).  Sub-batches stepped in lockstep therefore keep one common tick (and equal the env stepped whole, bit
        for bit), and whichever side steps next continues past everything the other side has drawn."""
        eng = self.engine
        if eng.device_tick:      # a captured loop owns the word (set refused): the capture keeps to one handle
            return
        if self._parent is not None:
            floor = self._parent.engine.tick
        else:
            floor = max(v.engine.tick for v in self._views)
        if floor > eng.tick:
            eng.tick = floor

    def _infos(self, actions):
        infos = {"steps": self._out(self._get_steps()), "reward_gt": self._of(self._reward_gt)}
        if self.autoreset_mode == "same_step":
            infos["final_obs"] = self._of(self._final_obs)
            infos["_final_obs"] = self._out((self._term | self._trunc).view(torch.bool))
        if self.with_transition_gt:
            _lib.check(self.lib.xv_anymdp_transition_gt(self._h, _lib.ptr(actions), _lib.ptr(self._tgt)))
            infos["transition_gt"] = self._out(self._tgt.clone())
        return infos

    def step(self, actions):
        self._check_step()
        if self._tok is not None:
            a = self._tok_action(actions)
            if not self.copy and not self.to_numpy:     # persistent outputs: pointers and views are made once
                c = self._tok_cache
                if c is None or c["key"] != (self._tobs.data_ptr(), self._tfobs.data_ptr()):
                    c = self._tok_cache = self._tok_make_cache()
                _lib.check(self.lib.xv_anymdp_step_tokens_info(self._h, C.c_void_p(a.data_ptr()), *c["args"],
                                                               AUTORESET[self.autoreset_mode]))
                infos = {"steps": self._steps, "reward_gt": self._reward_gt}
                if self.autoreset_mode == "same_step":
                    infos["final_obs"] = c["fobs"]
                    infos["_final_obs"] = c["done_b"]
                return c["obs"], self._reward, c["term_b"], c["trunc_b"], infos
            self._tok_fresh()
            self._renew("_steps", "_done")
            # ONE launch: the token step writes info["steps"] and the terminated | truncated mask itself
            _lib.check(self.lib.xv_anymdp_step_tokens_info(
                self._h, _lib.ptr(a), _lib.ptr(self._tobs), _lib.ptr(self._reward), _lib.ptr(self._reward_gt),
                _lib.ptr(self._term), _lib.ptr(self._trunc), _lib.ptr(self._tfobs), _lib.ptr(self._steps), _lib.ptr(self._done),
                AUTORESET[self.autoreset_mode]))
            return self._tok_ret(from_launch=True)
        a = self._dev(actions, torch.int32)
        if a.shape != (self.num_envs,):
            raise AssertionError(f"Action {tuple(a.shape)} is out of range")
        if self._ring is not None:      # copy=False: outputs are views of the output set this step writes
            b = self._ring[self._ring_pos]
            if not self._holding:           # a captured loop (capture.py) writes one output set: its policy reads it back
                self._ring_pos ^= 1
            lib, mode = self.lib, AUTORESET[self.autoreset_mode]
            # ONE launch: the step kernel writes info["steps"] and the terminated | truncated mask itself (xv_anymdp_step_info)
            f = _lib.fast()
            if f is not None:
                rc = f.icall(self._fn_step_info(), self._h.value, a.data_ptr(), *b["ints"], mode)
                if rc:
                    _lib.check(rc)
            else:
                _lib.check(lib.xv_anymdp_step_info(self._h, C.c_void_p(a.data_ptr()), *b["args"], b["steps_p"], b["done_p"], mode))
            infos = {"steps": b["steps"], "reward_gt": b["reward_gt"]}
            if mode == 2:
                infos["final_obs"] = b["final_obs"]
                infos["_final_obs"] = b["done_b"]
            return b["obs"], b["reward"], b["term_b"], b["trunc_b"], infos
        if self.copy and not self.to_numpy and not self.with_transition_gt:
            # copy=True without copies and without allocations: the step writes every output for every env into the next set of
            # a slab made for 64 steps at once (vector.OutputSlabs); what earlier steps handed out is never written again
            t, p, pi = self._slab_next3()
            mode = AUTORESET[self.autoreset_mode]
            f = _lib.fast()
            if f is not None:      # the CPython trampoline (csrc/xvfast.c): the same call without ctypes' marshalling
                rc = f.icall(self._fn_step_info(), self._h.value, a.data_ptr(), *pi, mode)
                if rc:
                    _lib.check(rc)
            else:
                _lib.check(self.lib.xv_anymdp_step_info(self._h, C.c_void_p(a.data_ptr()), *p, mode))
            self._obs = t["obs"]      # (reset() and the accessors read the latest observation from here)
            infos = {"steps": t["steps"], "reward_gt": t["reward_gt"]}
            if mode == 2:
                infos["final_obs"] = t["final_obs"]
                infos["_final_obs"] = t["done"]
            return t["obs"], t["reward"], t["term"], t["trunc"], infos
        # copy=True without copies: the step writes every output for every env, so fresh buffers are swapped in
        self._renew("_obs", "_reward", "_reward_gt", "_term", "_trunc", "_final_obs", "_steps", "_done")
        _lib.check(self.lib.xv_anymdp_step_info(
            self._h, _lib.ptr(a), _lib.ptr(self._obs), _lib.ptr(self._reward), _lib.ptr(self._reward_gt),
            _lib.ptr(self._term), _lib.ptr(self._trunc), _lib.ptr(self._final_obs), _lib.ptr(self._steps), _lib.ptr(self._done),
            AUTORESET[self.autoreset_mode]))
        infos = {"steps": self._of(self._steps), "reward_gt": self._of(self._reward_gt)}
        if self.autoreset_mode == "same_step":
            infos["final_obs"] = self._of(self._final_obs)
            infos["_final_obs"] = self._obf(self._done)
        if self.with_transition_gt:
            _lib.check(self.lib.xv_anymdp_transition_gt(self._h, _lib.ptr(a), _lib.ptr(self._tgt)))
            infos["transition_gt"] = self._out(self._tgt.clone())
        return (self._of(self._obs), self._of(self._reward), self._obf(self._term), self._obf(self._trunc), infos)

    _FN_STEP_INFO = [None]

    def _fn_step_info(self):
        if self._FN_STEP_INFO[0] is None:
            self._FN_STEP_INFO[0] = _lib.fn_address("xv_anymdp_step_info")
        return self._FN_STEP_INFO[0]

    def _slab_next(self):
        return self._slab_next3()[:2]

    def _slab_next3(self):
        """copy=True: the next output set of a 64-step slab (vector.OutputSlabs) -> (dict of tensors, pointers in the order
        obs, reward, reward_gt, term, trunc, final_obs, steps, done — as ctypes objects and as ints)"""
        if self._slabs is None:
            i32, f32, u8 = torch.int32, torch.float32, torch.uint8
            self._slabs = OutputSlabs([("obs", i32, ()), ("final_obs", i32, ()), ("steps", i32, ()), ("reward", f32, ()),
                                       ("reward_gt", f32, ()), ("term", u8, ()), ("trunc", u8, ()), ("done", u8, ())],
                                      self.num_envs, self.device, K=64, as_bool=("term", "trunc", "done"),
                                      order=("obs", "reward", "reward_gt", "term", "trunc", "final_obs", "steps", "done"))
        return self._slabs.next3()

    def step_injected(self, actions, u, z, u_reset):
        """Parity hook (C-ABI xv_anymdp_step_injected): random inputs supplied per env."""
        self._check_step()
        a = self._dev(actions, torch.int32)
        u = self._dev(u, torch.float64)
        z = self._dev(z, torch.float32)
        ur = self._dev(u_reset, torch.float64)
        self._renew("_obs", "_reward", "_reward_gt", "_term", "_trunc", "_final_obs")
        _lib.check(self.lib.xv_anymdp_step_injected(
            self._h, _lib.ptr(a), _lib.ptr(u), _lib.ptr(z), _lib.ptr(ur), _lib.ptr(self._obs),
            _lib.ptr(self._reward), _lib.ptr(self._reward_gt), _lib.ptr(self._term), _lib.ptr(self._trunc),
            _lib.ptr(self._final_obs), AUTORESET[self.autoreset_mode]))
        return (self._of(self._obs), self._of(self._reward), self._obf(self._term), self._obf(self._trunc),
                self._infos(a))

    def rollout(self, actions, out=None):
        """Fused open-loop rollout: actions int32[T, N] -> dict of [T, N] device tensors, one launch.
        Equals T calls of step() with SAME_STEP auto-reset, bit for bit."""
        self._check_step()
        a = self._dev(actions, torch.int32)
        T = int(a.shape[0])
        assert a.shape == (T, self.num_envs)
        d = self.device
        if out is None:
            out = dict(obs=torch.empty((T, self.num_envs), dtype=torch.int32, device=d),
                       reward=torch.empty((T, self.num_envs), dtype=torch.float32, device=d),
                       reward_gt=torch.empty((T, self.num_envs), dtype=torch.float32, device=d),
                       terminated=torch.empty((T, self.num_envs), dtype=torch.uint8, device=d),
                       truncated=torch.empty((T, self.num_envs), dtype=torch.uint8, device=d),
                       final_obs=torch.empty((T, self.num_envs), dtype=torch.int32, device=d))
        _lib.check(self.lib.xv_anymdp_rollout(
            self._h, T, _lib.ptr(a), _lib.ptr(out["obs"]), _lib.ptr(out["reward"]), _lib.ptr(out["reward_gt"]),
            _lib.ptr(out["terminated"]), _lib.ptr(out["truncated"]), _lib.ptr(out.get("final_obs"))))
        return out

    def solve(self, gamma=0.99, tol=1.0e-4, max_iter=20000, return_q=True):
        """Value iteration for every task of the batch on the device (one workgroup per task; the ground-truth
        teacher of AnyMDPSolverOpt, anymdp_solver_opt.py:30-51).  -> (Q float64[n_task, S, A] or None,
        greedy uint8[n_task, S], sweeps int32[n_task])"""
        self._require_task()
        d = self.device
        q = torch.empty((self.n_task, self.S, self.A), dtype=torch.float64, device=d) if return_q else None
        g = torch.empty((self.n_task, self.S), dtype=torch.uint8, device=d)
        it = torch.empty(self.n_task, dtype=torch.int32, device=d)
        _lib.check(self.lib.xv_anymdp_solve(self._h, float(gamma), float(tol), int(max_iter), _lib.ptr(q), _lib.ptr(g),
                                            _lib.ptr(it)))
        return q, g, it

    def rollout_teacher(self, T, greedy=None, epsilon=0.0, gamma=0.99):
        """Fused T-step rollout driven on the device by a teacher: greedy uint8[n_task, S] (e.g.
        `teacher.optimal_policy_table(tasks)`; None: solved on the device with `solve(gamma)` and cached),
        epsilon-greedy.  -> dict of [T, N] device tensors incl. "action"."""
        self._check_step()
        if greedy is None:
            if getattr(self, "_greedy_cache", None) is None or self._greedy_cache[0] != (id(self._tab), gamma):
                self._greedy_cache = ((id(self._tab), gamma), self.solve(gamma, return_q=False)[1])
            greedy = self._greedy_cache[1]
        g = self._dev(greedy, torch.uint8)
        assert g.shape == (self.n_task, self.S)
        d, n = self.device, self.num_envs
        out = dict(action=torch.empty((T, n), dtype=torch.int32, device=d),
                   obs=torch.empty((T, n), dtype=torch.int32, device=d),
                   reward=torch.empty((T, n), dtype=torch.float32, device=d),
                   reward_gt=torch.empty((T, n), dtype=torch.float32, device=d),
                   terminated=torch.empty((T, n), dtype=torch.uint8, device=d),
                   truncated=torch.empty((T, n), dtype=torch.uint8, device=d),
                   final_obs=torch.empty((T, n), dtype=torch.int32, device=d))
        _lib.check(self.lib.xv_anymdp_rollout_teacher(
            self._h, int(T), _lib.ptr(g), float(epsilon), _lib.ptr(out["action"]), _lib.ptr(out["obs"]),
            _lib.ptr(out["reward"]), _lib.ptr(out["reward_gt"]), _lib.ptr(out["terminated"]),
            _lib.ptr(out["truncated"]), _lib.ptr(out["final_obs"])))
        self.engine.sync()
        return out

    _MANY_KEYS = ("obs", "reward", "reward_gt", "terminated", "truncated", "final_obs")

    def step_many(self, n_steps, actions, out=None, chains=1, how="streams"):
        """n_steps back-to-back step launches issued from C.  actions int32[P, N] is cycled with period P;
        `out` holds [P, N] ring buffers (allocated when None) — slot k % P receives step k.

        chains = K > 1: the envs are stepped as K independent chains of N / K envs (split(K); xv_anymdp_step_many_chains) —
        step k of a chain waits for step k - 1 of that chain only, so one chain's launch gap is covered by the others'
        table lines in flight.  how: "streams" (each chain on its own stream, its own cycle graph) or "graph" (one graph
        with K branches).  Outputs, env states and the tick afterwards equal chains=1 bit for bit."""
        self._check_step()
        c = self._many_cache
        if c is not None and out is not None and actions is c[0] and all(out.get(k) is t for k, t in zip(self._MANY_KEYS, c[1])):
            args = c[2]     # the same tensors as in the last call: the marshalled argument list is reused
        else:
            a = self._dev(actions, torch.int32)
            P = int(a.shape[0])
            assert a.shape == (P, self.num_envs)
            d = self.device
            if out is None:
                out = dict(obs=torch.empty((P, self.num_envs), dtype=torch.int32, device=d),
                           reward=torch.empty((P, self.num_envs), dtype=torch.float32, device=d),
                           reward_gt=torch.empty((P, self.num_envs), dtype=torch.float32, device=d),
                           terminated=torch.empty((P, self.num_envs), dtype=torch.uint8, device=d),
                           truncated=torch.empty((P, self.num_envs), dtype=torch.uint8, device=d),
                           final_obs=torch.empty((P, self.num_envs), dtype=torch.int32, device=d))
            for k in self._MANY_KEYS[:5]:
                assert tuple(out[k].shape) == (P, self.num_envs) and out[k].device == d, k
            args = (P, _lib.ptr(a), _lib.ptr(out["obs"]), _lib.ptr(out["reward"]), _lib.ptr(out["reward_gt"]),
                    _lib.ptr(out["terminated"]), _lib.ptr(out["truncated"]), _lib.ptr(out.get("final_obs")))
            # `a` is kept alive with the cache entry when _dev had to convert `actions`
            self._many_cache = (actions, tuple(out.get(k) for k in self._MANY_KEYS), args, a) if a is actions else None
        if chains > 1:
            views = self.split(chains)
            arr = self._chain_handles
            _lib.check(self.lib.xv_anymdp_step_many_chains(self._h, arr, len(views), {"streams": 0, "graph": 1}[how],
                                                           int(n_steps), *args, AUTORESET[self.autoreset_mode]))
            return out
        _lib.check(self.lib.xv_anymdp_step_many(self._h, int(n_steps), *args, AUTORESET[self.autoreset_mode]))
        return out

    # ---- sub-batches (xv_anymdp_view) ------------------------------------------------------------------------
    def split(self, K):
        """-> K AnyMDPVecEnv over contiguous sub-batches of this env's envs (K | num_envs), each with a HIP stream of its
        own (`sub.stream`).  The reference steps env objects one at a time (anymdp/test_utils.py:42-60): sub-batches of a
        vector step are independent, so [policy(sub k) -> step(sub k)] of one sub-batch can overlap another's.  A sub-batch
        is a full VectorEnv (step, reset, step_many, rollout, get_state ...); it shares tables and env states with this env:
        stepping sub k IS stepping envs [k N / K, (k + 1) N / K) of this env, with the draws this env would make at the same
        tick (every sub-batch starts at this env's tick of the moment).  Stepping the env and its sub-batches in turn is safe:
        each side starts a launch past every tick the other has used (_sync_tick), so no env draws twice at one tick.  The
        views are cached per K and dropped by set_task / set_search(n_bucket=...)."""
        self._require_task()
        K = int(K)
        if self._parent is not None:
            raise ValueError("split() of a sub-batch")
        if self._views and len(self._views) == K:
            self._sync_tick()      # (the cached views continue from wherever this env and they have got to)
            t = self.engine.tick
            for v in self._views:
                if v.engine.tick < t and not v.engine.device_tick:
                    v.engine.tick = t
            return self._views
        if K < 1 or K > 16 or self.num_envs % K != 0:
            raise ValueError("split(K): K must divide num_envs and be at most 16")
        self._drop_views()
        n = self.num_envs // K
        tick = self.engine.tick
        views = []
        for c in range(K):
            views.append(AnyMDPSubBatch._make_view(self, c * n, n, tick))
        self._views = views
        self._chain_handles = (C.c_void_p * K)(*[v._h for v in views])
        return views

    @classmethod
    def _make_view(cls, parent, lo, n, tick):
        from ..engine import Engine
        stream = torch.cuda.Stream(device=parent.device)
        eng = Engine(parent.device, seed=parent.engine.seed, env_id_base=parent.engine.env_id_base + lo, stream=stream)
        eng.tick = tick
        v = cls(n, max_steps=parent.max_steps, autoreset_mode=parent.autoreset_mode, to_numpy=parent.to_numpy, engine=eng,
                with_transition_gt=parent.with_transition_gt, copy=parent.copy, bucket_lines="off")
        v._own_engine = True
        v._parent, v.view_lo, v.stream = parent, lo, stream
        h = C.c_void_p()
        _lib.check(parent.lib.xv_anymdp_view(parent._h, eng.handle, int(lo), int(n), C.byref(h)))
        v._h = h
        v._tab = dict(parent._tab)               # keeps the borrowed tables alive
        v._tab["env_task"] = parent._tab["env_task"][lo:lo + n]
        for k in ("S", "A", "s0_max", "n_task", "ns", "na", "task_type", "_tok", "_n_bucket"):
            setattr(v, k, getattr(parent, k, None))
        for k in ("no", "do", "da"):
            if hasattr(parent, k):
                setattr(v, k, getattr(parent, k))
        v._set_spaces(Discrete(parent.ns), Discrete(parent.A))
        with torch.cuda.stream(stream):
            v._make_buffers()
        stream.wait_stream(torch.cuda.current_stream(parent.device))      # behind whatever built / reset the parent
        v.task_set = True
        v.need_reset = parent.need_reset
        return v

    def _drop_views(self):
        for v in self._views:
            v.close()
        self._views = []
        self._chain_handles = None

    def check_errors(self, clear=True):
        """device error bits of this env and of its sub-batches (a view's launches report to its own engine)"""
        f = self.engine.error_flags(clear)
        for v in self._views:
            f |= v.engine.error_flags(clear)
        return f

    def step_tokens_many(self, n_steps, actions, out=None):
        """POMDP / multi-token tasks: n_steps token steps issued from C (xv_anymdp_step_tokens_many).  actions int32[P, N,
        d_act] is cycled with period P; `out` holds ring buffers (allocated when None): obs / final_obs [P, N, d_obs],
        reward, reward_gt, terminated, truncated [P, N] — slot k % P receives step k.  Equals n_steps calls of step()."""
        self._check_step()
        assert self._tok is not None, "step_tokens_many needs a POMDP / MTPOMDP task"
        d_obs, d_act = self._tok
        a = self._dev(actions, torch.int32)
        if a.dim() == 2 and d_act == 1:
            a = a[:, :, None]
        P = int(a.shape[0])
        assert a.shape == (P, self.num_envs, d_act)
        a = a.contiguous()
        d, n = self.device, self.num_envs
        if out is None:
            out = dict(obs=torch.empty((P, n, d_obs), dtype=torch.int32, device=d),
                       reward=torch.empty((P, n), dtype=torch.float32, device=d),
                       reward_gt=torch.empty((P, n), dtype=torch.float32, device=d),
                       terminated=torch.empty((P, n), dtype=torch.uint8, device=d),
                       truncated=torch.empty((P, n), dtype=torch.uint8, device=d),
                       final_obs=torch.full((P, n, d_obs), -1, dtype=torch.int32, device=d))
        _lib.check(self.lib.xv_anymdp_step_tokens_many(
            self._h, int(n_steps), P, _lib.ptr(a), _lib.ptr(out["obs"]), _lib.ptr(out["reward"]), _lib.ptr(out["reward_gt"]),
            _lib.ptr(out["terminated"]), _lib.ptr(out["truncated"]), _lib.ptr(out.get("final_obs")),
            AUTORESET[self.autoreset_mode]))
        return out

    def set_step_many_graph(self, mode):
        """step_many replays whole ring cycles from a hipGraph (True / "on"), issues plain launches (False / "off")
        or decides ("auto", the default: graph up to 8,192 envs and for calls of at most 128 steps); same results
        either way."""
        m = {"off": 0, "on": 1, "auto": 2}.get(mode, 1 if mode is True else (0 if mode is False else mode))
        _lib.check(self.lib.xv_anymdp_set_step_many_graph(self._h, int(m)))

    def set_step_many_overlap(self, on=True):
        """step_many issues consecutive vector steps in turn on two or three HIP streams with no dependency between them; each
        wave of step k + 1 takes its envs over from the same wave of step k through a hand-off word (xeno.h:
        xv_anymdp_set_step_many_overlap).  Same results; whole cycles of an even ring period only."""
        _lib.check(self.lib.xv_anymdp_set_step_many_overlap(self._h, 1 if on else 0))

    @property
    def step_many_overlap_state(self):
        """1: the last step_many overlapped its cycles, 0: it did not, -1: the overlapped path failed on this env, -2: it
        overlapped, a hand-off expired and the call was replayed on one stream (results are right; read after a sync)"""
        return int(self.lib.xv_anymdp_step_many_overlap_state(self._h))

    # ---- accessors (anymdp_env.py:134-165) ----------------------------------------------------------
    def _get_steps(self):
        self._renew("_steps")
        _lib.check(self.lib.xv_anymdp_get_state(self._h, None, _lib.ptr(self._steps), None))
        return self._steps if (self.copy and not self.to_numpy) else self._steps.clone()

    @property
    def inner_state(self):
        s = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.xv_anymdp_get_state(self._h, _lib.ptr(s), None, None))
        return self._out(s)

    @property
    def state(self):
        """observation-space id of each env's state: state_mapping[_state] (anymdp_env.py:134-136)"""
        s = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.xv_anymdp_get_state(self._h, _lib.ptr(s), None, None))
        idx = self._tab["env_task"].long() * self.S + s.long()
        return self._out(self._tab["state_map"].reshape(-1)[idx])

    def get_gt_transition(self, task=0):
        """transition_obs of one task: the transition tensor re-indexed by observation ids
        (anymdp_env.py:12-20,61-63,161-162) — what the reference's solvers read.  float64[ns, na, ns] (host)."""
        from .tables import from_blocked
        cdf, _ = from_blocked(self._tab["rows"][task].cpu().numpy(), self.S)
        T = np.diff(np.concatenate([np.zeros(cdf.shape[:-1] + (1,)), cdf], -1), axis=-1)
        tm = int(self._tab["term_mask"][task, 0].item()) if self.S <= 64 else None
        sm = self._tab["state_map"][task].cpu().numpy()
        out = np.zeros((self.ns, self.A, self.ns))
        for i in range(self.S):
            if tm is not None and (tm >> i) & 1:
                continue        # rows of terminal states are all-zero in the reference
            out[sm[i]][:, sm] = T[i]
        return out

    def get_gt_reward(self, task=0):
        """reward_obs of one task (anymdp_env.py:164-165), fp32 table values as float64[ns, na, ns] (host)"""
        from .tables import from_blocked
        _, rs = from_blocked(self._tab["rows"][task].cpu().numpy(), self.S)
        sm = self._tab["state_map"][task].cpu().numpy()
        out = np.zeros((self.ns, self.A, self.ns))
        for i in range(self.S):
            out[sm[i]][:, sm] = rs[i, :, :, 0]
        return out

    def get_state(self):
        n, d = self.num_envs, self.device
        s = torch.empty(n, dtype=torch.int32, device=d)
        st = torch.empty(n, dtype=torch.int32, device=d)
        nr = torch.empty(n, dtype=torch.uint8, device=d)
        _lib.check(self.lib.xv_anymdp_get_state(self._h, _lib.ptr(s), _lib.ptr(st), _lib.ptr(nr)))
        return s, st, nr

    def set_state(self, inner_state=None, steps=None, need_reset=None):
        s = None if inner_state is None else self._dev(inner_state, torch.int32)
        st = None if steps is None else self._dev(steps, torch.int32)
        nr = None if need_reset is None else self._dev(need_reset, torch.uint8)
        _lib.check(self.lib.xv_anymdp_set_state(self._h, _lib.ptr(s), _lib.ptr(st), _lib.ptr(nr)))
        self.engine.sync()   # the temporaries above must outlive the copies
        self.need_reset = False

    def close_extras(self, **kwargs):
        self._drop_views()
        if self._h is not None:
            self.lib.xv_anymdp_destroy(self._h)
            self._h = None
        self._tab = None
        self._parent = None


class AnyMDPSubBatch(AnyMDPVecEnv):
    """A sub-batch of an AnyMDPVecEnv (AnyMDPVecEnv.split): the same surface, launched on `self.stream`.  Calls that
    launch run under `torch.cuda.stream(self.stream)`, so the tensors they allocate belong to that stream and torch ops
    the caller issues on the results inside the same `with` are ordered behind the step; to read a result from another
    stream, `other.wait_stream(sub.stream)` first (or `sub.stream.synchronize()`)."""

    def _on_stream(name):      # noqa: N805 (a decorator factory evaluated in the class body)
        def call(self, *a, **kw):
            with torch.cuda.stream(self.stream):
                return getattr(AnyMDPVecEnv, name)(self, *a, **kw)
        call.__name__ = name
        call.__doc__ = getattr(AnyMDPVecEnv, name).__doc__
        return call

    for _n in ("reset", "step", "step_injected", "reset_injected", "rollout", "rollout_teacher", "step_many",
               "step_tokens_many", "step_tokens_injected", "reset_tokens_injected", "get_state", "set_state", "solve"):
        locals()[_n] = _on_stream(_n)
    del _n, _on_stream

    @property
    def inner_state(self):
        with torch.cuda.stream(self.stream):
            r = AnyMDPVecEnv.inner_state.fget(self)
        self.stream.synchronize()
        return r

    @property
    def state(self):
        with torch.cuda.stream(self.stream):
            r = AnyMDPVecEnv.state.fget(self)
        self.stream.synchronize()
        return r
