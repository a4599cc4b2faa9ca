"""VectorEnv base: the gymnasium >= 1.0 `VectorEnv` surface over a device
engine.  gymnasium's class is generic over the array type; these envs return torch tensors that live on
the GPU (`to_numpy=True` converts for NumPy consumers)."""
import math

import torch

from . import _lib
from .engine import AUTORESET, Engine
from .spaces import batch_space


try:      # gymnasium >= 1.0 present: be a gymnasium.vector.VectorEnv (isinstance checks of wrappers / trainers hold)
    from gymnasium.vector import VectorEnv as _GymVectorEnv
except Exception:      # not installed: the same surface on a plain class
    _GymVectorEnv = object


class OutputSlabs(object):
    """Fresh output tensors for every step() at the cost of none: the outputs of K consecutive steps are carved out of ONE
    allocation per dtype (K x fields x N), the per-step tensors and their device pointers are made when the slab is, and a
    step takes the next precomputed set.  What a step hands out is never written again WHILE ANYBODY CAN SEE IT — copy=True
    semantics, as gymnasium's SyncVectorEnv(copy=True) — and a tensor the caller keeps keeps its slab alive (torch counts
    references on the storage).  (One torch.empty_like per output per step cost ~10 us of the 22-us eager step of 65,536
    envs.)

    Round 6: slabs are RECYCLED, with their tensor objects.  Up to `keep` slabs stay cached; when a slab is used up, the oldest
    cached one that nobody outside this object can reach any more is handed out again — the very same Python tensor objects,
    so a step creates and destroys no object at all (14.2 -> ~10 us per eager copy=True step of 65,536 AnyMDP envs).  "Nobody
    can reach it" is checked, not assumed, once per slab (K steps):
      * every cached tensor object has exactly the Python references this cache holds (sys.getrefcount): no caller variable,
        list or dict holds one of them;
      * every cached tensor's TensorImpl has one owner (Tensor._use_count() == 1): no C++ holder such as a DLPack capsule;
      * every slab storage is shared by exactly the tensors made here (torch._C._storage_Use_Count): no view, .detach(),
        .numpy() or reshaped alias of a handed-out tensor is alive.
    A slab that fails any of the three is left alone (and dropped from the cache when `keep` newer ones exist: it lives on
    for as long as its holders do).

    fields: [(name, torch dtype, trailing shape)], uint8 fields named in `as_bool` are handed out as bool views."""

    def __init__(self, fields, n, device, K=64, as_bool=(), order=None, keep=3, derived=None, zero_on_refill=()):
        """order: field names in the order the C call takes its pointers — next() then returns them as a ready tuple
        derived: {name: (field, fn)} — further views made once per slab, e.g. the user's columns of a padded observation;
        zero_on_refill: fields whose whole [K, ...] block is zeroed when a slab is (re)issued — for outputs a step writes only
        in part (LinDS final_obs: rows of finished envs), ONE fill per K steps instead of one per step"""
        self.fields, self.n, self.device, self.K = list(fields), int(n), device, int(K)
        self.derived = dict(derived or {})
        self.zero_on_refill = tuple(zero_on_refill)
        self.as_bool = set(as_bool)
        self.order = list(order) if order is not None else [f[0] for f in self.fields]
        self.keep = max(1, int(keep))
        self._sets, self._pos = [], 0
        self._ring = []          # cached slabs, oldest first; the last one is the slab in use
        self.made = 0            # slabs allocated so far (tests, diagnostics)
        self.recycled = 0        # slabs handed out again

    def _make(self):
        import ctypes as C
        by = {}
        for name, dt, tail in self.fields:
            by.setdefault(dt, []).append((name, tuple(tail)))
        per_step = [dict() for _ in range(self.K)]
        ptrs = [dict() for _ in range(self.K)]
        storages, zero_blocks = [], []
        for dt, fl in by.items():
            # [field][step][n * tail]: two unbind() calls hand out every per-step tensor (slicing K x fields views one by one
            # in Python cost more than the allocations it was meant to save)
            width = max(self.n * math.prod(t) for _, t in fl)
            slab = torch.empty((len(fl), self.K, width), dtype=dt, device=self.device)
            storages.append(slab.untyped_storage())
            base, esz = slab.data_ptr(), slab.element_size()
            fields = (slab.view(torch.bool) if dt == torch.uint8 else slab).unbind(0)
            plain = slab.unbind(0)
            for f, (name, tail) in enumerate(fl):
                src = fields[f] if name in self.as_bool else plain[f]
                sz = self.n * math.prod(tail)
                if sz != width:
                    src = src[:, :sz]
                if name in self.zero_on_refill:
                    zero_blocks.append(src)
                if tail:
                    src = src.view((self.K, self.n) + tail)
                rows = src.unbind(0)
                for k in range(self.K):
                    per_step[k][name] = rows[k]
                    ptrs[k][name] = C.c_void_p(base + ((f * self.K + k) * width) * esz)
        for name, (field, fn) in self.derived.items():
            for k in range(self.K):
                per_step[k][name] = fn(per_step[k][field])
        rec = {"zero": zero_blocks,
               "sets": [(per_step[k], tuple(ptrs[k][name] for name in self.order), tuple(ptrs[k][name].value for name in self.order))
                        for k in range(self.K)],
               "flat": [t for d in per_step for t in d.values()] + per_step, "tensors": [t for d in per_step for t in d.values()],
               "storages": storages}      # (`flat` counts the per-step dicts too: next() hands them out)
        self.made += 1
        return rec

    @staticmethod
    def _counts(rec):
        import sys
        return (sum(map(sys.getrefcount, rec["flat"])), sum(map(torch.Tensor._use_count, rec["tensors"])),
                tuple(torch._C._storage_Use_Count(st._cdata) for st in rec["storages"]))

    def _refill(self):
        # the oldest cached slab nobody outside can reach, else a new one.  (The slab just used up is the newest: the
        # caller usually still holds its last outputs, so it is looked at last.)
        pick = None
        for i, rec in enumerate(self._ring):
            if self._counts(rec) == rec["idle"]:
                pick = self._ring.pop(i)
                self.recycled += 1
                break
        if pick is None:
            pick = self._make()
            pick["idle"] = None
        self._ring.append(pick)
        for blk in pick["zero"]:
            blk.zero_()
        if pick["idle"] is None:      # (taken with every temporary of _make gone: the slab is referenced exactly as when it idles)
            pick["idle"] = self._counts(pick)
        while len(self._ring) > self.keep:
            self._ring.pop(0)
        self._sets = pick["sets"]
        self._pos = 0

    def next(self):
        """-> (dict name -> tensor, tuple of ctypes pointers in `order`) of a set no earlier step has written — or one whose
        earlier contents nobody can reach any more"""
        return self.next3()[:2]

    def next3(self):
        """next() plus the same pointers as plain ints (for _xvfast.icall)"""
        if self._pos >= len(self._sets):
            self._refill()
        s = self._sets[self._pos]
        self._pos += 1
        return s


class VectorEnv(_GymVectorEnv):
    metadata = {"autoreset_mode": "same_step"}
    render_mode = None
    spec = None

    def __init__(self, num_envs, device="cuda:0", seed=0, env_id_base=0, autoreset_mode="same_step",
                 to_numpy=False, engine=None, copy=True):
        if autoreset_mode not in AUTORESET:
            raise ValueError("autoreset_mode must be one of %s" % sorted(AUTORESET))
        self.num_envs = int(num_envs)
        self.engine = engine if engine is not None else Engine(device, seed=seed, env_id_base=env_id_base)
        self._own_engine = engine is None
        self.device = self.engine.device
        self.lib = self.engine.lib
        self.autoreset_mode = autoreset_mode
        self.metadata = dict(type(self).metadata, autoreset_mode=autoreset_mode)
        self.to_numpy = bool(to_numpy)
        # copy (as gymnasium's SyncVectorEnv(copy=...)): True returns fresh tensors from every call; False returns
        # views of engine-owned output buffers, valid until the next step()/reset() (no per-step device copies)
        self.copy = bool(copy)
        # lean_infos: step() returns only the infos that cost no extra launch (captured loops: xenoverse_amd/capture.py)
        self.lean_infos = False
        self._holding = False
        self.closed = False
        self.task_set = False
        self.single_observation_space = None
        self.single_action_space = None
        self.observation_space = None
        self.action_space = None

    # -- helpers ------------------------------------------------------------------------------------
    def _set_spaces(self, single_obs, single_act):
        self.single_observation_space = single_obs
        self.single_action_space = single_act
        self.observation_space = batch_space(single_obs, self.num_envs)
        self.action_space = batch_space(single_act, self.num_envs)

    def _dev(self, x, dtype):
        """Bring `x` (tensor / ndarray / list) to a contiguous device tensor of `dtype` without copying
        when it already is one."""
        if torch.is_tensor(x):
            if x.device != self.device or x.dtype != dtype:
                x = x.to(device=self.device, dtype=dtype)
            return x.contiguous()
        return torch.as_tensor(x).to(device=self.device, dtype=dtype).contiguous()

    def _out(self, t):
        return t.cpu().numpy() if self.to_numpy else t

    def _o(self, t):
        """an output buffer -> what the caller receives (a copy unless copy=False)"""
        return self._out(t.clone() if (self.copy and not self.to_numpy) else t)

    def _renew(self, *names):
        """copy=True without copies: before a launch that fully overwrites these output buffers, swap in fresh
        ones — the tensors handed out by the previous call are then never written again (use with _of / _obf)"""
        if self.copy and not self.to_numpy:
            for name in names:
                t = getattr(self, name)
                if t is not None:
                    setattr(self, name, torch.empty_like(t))

    def _detach(self, *names):
        """copy=True before a launch that overwrites output buffers only in part (masked reset, render of the
        current state): the launch writes into private clones, so tensors handed out earlier keep their values
        and the entries the launch does not touch keep theirs"""
        if self.copy and not self.to_numpy:
            for name in names:
                t = getattr(self, name, None)
                if t is not None:
                    setattr(self, name, t.clone())

    def _of(self, t):
        """an output buffer that was renewed before the launch (or copy=False): handed out as it is"""
        return self._out(t)

    def _obf(self, t):
        return self._out(t.bool() if self.to_numpy else t.view(torch.bool))

    def _ob(self, t):
        """a uint8 0/1 flag buffer -> bool tensor (zero-copy view when copy=False)"""
        return self._out(t.bool() if (self.copy or self.to_numpy) else t.view(torch.bool))

    def _require_task(self):
        if not self.task_set:
            # reference: raise Exception("Must call \"set_task\" first") (anymdp_env.py:83, linds_env.py:110)
            raise Exception("Must call \"set_task\" first")

    def _capture_hold(self, on):
        """while a captured loop exists: pins whatever the family alternates between step() calls (output sets) and,
        when asked, leaves out the infos that need a launch of their own"""
        self._holding = bool(on)
        self.lean_infos = self._lean_wanted if on else self._lean_saved

    AUTO_UNROLL = 8      # one hipGraphLaunch costs the host about what it replaces; 8 iterations per launch: 15 -> 7.3 us per step

    def capture(self, policy_fn, obs, unroll=1, warmup=1, lean=True):
        """[policy_fn(obs) -> actions; step(actions)] captured in a torch.cuda.graph -> CapturedLoop (capture.py): one
        graph launch per `unroll` vector steps, same trajectory as the eager calls.  Needs copy=False, to_numpy=False.
        unroll="auto": AUTO_UNROLL iterations per graph launch (`loop.unroll` says how many steps one replay() makes).
        `warmup` eager iterations run first and are real steps.  lean: step() skips the infos that need a launch of
        their own (AnyMDP / LinDS `steps`, the `_final_obs` mask) while the loop exists."""
        from .capture import CapturedLoop
        if self.copy or self.to_numpy:
            raise ValueError("capture() needs an env built with copy=False and to_numpy=False (fixed output buffers)")
        self._lean_saved, self._lean_wanted = self.lean_infos, bool(lean) or self.lean_infos
        if unroll == "auto":
            unroll = self.AUTO_UNROLL
        return CapturedLoop(self.step, [self.engine], policy_fn, obs, unroll=unroll, warmup=warmup, device=self.device,
                            hold=self._capture_hold)

    def check_errors(self, clear=True):
        """Device-side range errors are sticky bits, read on demand (one stream sync)."""
        return self.engine.error_flags(clear)

    def close(self, **kwargs):
        if self.closed:
            return
        self.close_extras(**kwargs)
        if self._own_engine:
            self.engine.close()
        self.closed = True

    def close_extras(self, **kwargs):
        pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


XenoError = _lib.XenoError
