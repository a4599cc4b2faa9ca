"""One rank's share of a MIXED task batch (BASELINE.json configs[4]: anymdp + linds + metacontrol over the GPUs of a node)
and its exchange step.

The reference has no such driver — it steps one env object per call (anymdp/anymdp_env.py:112-132, linds/linds_env.py:133-169,
metacontrol/random_cartpole.py:52-61); what is mirrored is the per-family step, and what is added is the partitioning of
SURVEY.md 8(e): every family's global env range is cut into contiguous per-rank ranges (`distributed.shard_range`), a rank's
engines carry `env_id_base = lo` of their family, so an env draws the same Philox numbers whatever the rank count, and tasks
are named by GLOBAL task index (task g of a family is the same task on every layout).  Stepping needs no collective; the one
exchange is the all-gather of a T-step rollout chunk of all three families (`distributed.MixedChunk`: AnyMDP 8-byte records,
LinDS {obs f32[16], reward, flags}, CartPole {obs f32[4], reward, flags | action}).
"""
import ctypes as C

import torch

from . import _lib
from .distributed import MixedChunk
from .engine import AUTORESET
from .mixed import _MixedIO

ENVS_PER_TASK = {"anymdp": 64, "linds": 64, "cartpole": 8}      # 2b sharing for anymdp / linds

_LINDS_CACHE = {}


def linds_task(g, ns=32, distinct=16):
    """LinDS task of global index g: `distinct` sampler tasks (ns = 32 takes ~0.2 s of rejection sampling each), tiled — the
    tables are per task index, so every index owns its copy of the matrices in HBM"""
    from .linds import LinearDSSampler
    key = (ns, g % distinct)
    if key not in _LINDS_CACHE:
        t = LinearDSSampler(ns, 8, 8, seed=g % distinct)
        t["max_steps"] = 500
        _LINDS_CACHE[key] = t
    return _LINDS_CACHE[key]


class MixedShare(object):
    """Rank `rank` of `world`: its envs of the three families on ONE HIP stream (torch's current stream), [T, n, ...] ring
    buffers, the fused launch (`xv_mixed_step_many`, one kernel per vector step of all three families) and the chunk pack."""

    def __init__(self, rank, world, n_anymdp, n_linds, n_cartpole, T=32, seed=0, device="cuda:0", linds_ns=32,
                 anymdp_task_seed=7, autoreset_mode="same_step", bucket_lines="auto"):
        from .anymdp import AnyMDPVecEnv, row_lines
        from .linds import LinDSVecEnv
        from .metacontrol import CartPoleVecEnv, sample_cartpole
        self.rank, self.world, self.T = int(rank), int(world), int(T)
        self.chunk = MixedChunk(T, n_anymdp, n_linds, n_cartpole, world)
        self.lo = {f: self.chunk.share[f][rank][0] for f in ENVS_PER_TASK}
        self.n = {f: self.chunk.n_local(f, rank) for f in ENVS_PER_TASK}
        for f, per in ENVS_PER_TASK.items():
            if self.lo[f] % per or self.n[f] % per or self.n[f] == 0:
                raise ValueError("family %s: a rank's share (%d envs from %d) must be whole tasks of %d envs"
                                 % (f, self.n[f], self.lo[f], per))
        self.mode = autoreset_mode
        na, nl, nc = self.n["anymdp"], self.n["linds"], self.n["cartpole"]
        self.ea = AnyMDPVecEnv(na, device=device, seed=seed, env_id_base=self.lo["anymdp"], autoreset_mode=autoreset_mode,
                               bucket_lines=bucket_lines)
        self.el = LinDSVecEnv(nl, device=device, seed=seed, env_id_base=self.lo["linds"], autoreset_mode=autoreset_mode)
        self.ec = CartPoleVecEnv(nc, device=device, seed=seed, env_id_base=self.lo["cartpole"], frameskip=1,
                                 autoreset_mode=autoreset_mode)
        d = self.device = self.ea.device
        S, A = 64, 8
        nt = na // ENVS_PER_TASK["anymdp"]
        tab = dict(S=S, A=A, s0_max=4, rows=torch.empty((nt, S, A, row_lines(S), 16), dtype=torch.float64, device=d),
                   state_map=torch.empty((nt, S), dtype=torch.int32, device=d),
                   term_mask=torch.empty((nt, 1), dtype=torch.int64, device=d),
                   s0_cdf=torch.empty((nt, 4), dtype=torch.float64, device=d),
                   s0_ids=torch.empty((nt, 4), dtype=torch.int32, device=d),
                   max_steps=torch.empty(nt, dtype=torch.int32, device=d))
        # synthetic tasks of SURVEY 8(d) config 2, generated on the device, keyed by the GLOBAL task index
        _lib.check(self.ea.lib.xv_anymdp_synth_tasks(self.ea.engine.handle, anymdp_task_seed, self.lo["anymdp"] // 64, nt, S, A, 4,
                                                     *[_lib.ptr(tab[k]) for k in ("rows", "state_map", "term_mask", "s0_cdf",
                                                                                  "s0_ids", "max_steps")]))
        self.ea.engine.sync()
        self.ea.set_task(tab)
        g0 = self.lo["linds"] // 64
        self.el.set_task([linds_task(g, linds_ns) for g in range(g0, g0 + nl // 64)])
        g0 = self.lo["cartpole"] // 8
        self.ec.set_task([sample_cartpole(seed=g) for g in range(g0, g0 + nc // 8)])
        LA, LO = self.el.NA, self.el.NO
        z = torch.zeros
        self.ring = dict(
            aa=z((T, na), device=d, dtype=torch.int32), ao=z((T, na), device=d, dtype=torch.int32), ar=z((T, na), device=d),
            ag=z((T, na), device=d), at=z((T, na), device=d, dtype=torch.uint8), au=z((T, na), device=d, dtype=torch.uint8),
            af=z((T, na), device=d, dtype=torch.int32),
            la=z((T, nl, LA), device=d), lo=z((T, nl, LO), device=d), lr=z((T, nl), device=d),
            lt=z((T, nl), device=d, dtype=torch.uint8), lu=z((T, nl), device=d, dtype=torch.uint8), lc=z((T, nl, LO), device=d),
            le=z((T, nl), device=d), lf=z((T, nl, LO), device=d),
            ca=z((T, nc), device=d, dtype=torch.int32), co=z((T, nc, 4), device=d), cr=z((T, nc), device=d),
            ct=z((T, nc), device=d, dtype=torch.uint8), cu=z((T, nc), device=d, dtype=torch.uint8), cf=z((T, nc, 4), device=d))
        self._io = _MixedIO(*[_lib.ptr(self.ring[k]) for k in ("aa", "ao", "ar", "ag", "at", "au", "af", "la", "lo", "lr", "lt",
                                                               "lu", "lc", "le", "lf", "ca", "co", "cr", "ct", "cu", "cf")])
        self.fused = bool(self.ea.lib.xv_mixed_supported(self.ea._h, self.el._h, self.ec._h))

    @property
    def num_envs(self):
        return sum(self.n.values())

    def set_actions(self, anymdp, linds, cartpole):
        """[T, n_local(, 8)] action rings (any array-like); LinDS actions are padded to the engine's action pad"""
        self.ring["aa"].copy_(torch.as_tensor(anymdp).to(self.device, torch.int32))
        la = torch.as_tensor(linds).to(self.device, torch.float32)
        self.ring["la"].zero_()
        self.ring["la"][..., :la.shape[-1]] = la
        self.ring["ca"].copy_(torch.as_tensor(cartpole).to(self.device, torch.int32))

    def random_actions(self, seed):
        g = torch.Generator(device=self.device)
        g.manual_seed(int(seed))
        self.ring["aa"].copy_(torch.randint(0, 8, self.ring["aa"].shape, generator=g, device=self.device, dtype=torch.int32))
        self.ring["la"].copy_(torch.rand(self.ring["la"].shape, generator=g, device=self.device) * 2 - 1)
        self.ring["ca"].copy_(torch.randint(0, 2, self.ring["ca"].shape, generator=g, device=self.device, dtype=torch.int32))

    def reset(self):
        return self.ea.reset(), self.el.reset(), self.ec.reset()

    def step_many(self, n_steps):
        """n_steps vector steps of the whole share issued from C; step k uses ring slot k % T"""
        mode = AUTORESET[self.mode]
        if self.fused:
            _lib.check(self.ea.lib.xv_mixed_step_many(self.ea._h, self.el._h, self.ec._h, C.byref(self._io), int(n_steps), self.T,
                                                      mode))
            return
        r = self.ring       # handles without a fused instantiation: three launches per vector step, same results
        lib = self.ea.lib
        for k in range(int(n_steps)):
            s = k % self.T
            _lib.check(lib.xv_anymdp_step(self.ea._h, _lib.ptr(r["aa"][s]), _lib.ptr(r["ao"][s]), _lib.ptr(r["ar"][s]),
                                          _lib.ptr(r["ag"][s]), _lib.ptr(r["at"][s]), _lib.ptr(r["au"][s]), _lib.ptr(r["af"][s]), mode))
            _lib.check(lib.xv_linds_step(self.el._h, _lib.ptr(r["la"][s]), _lib.ptr(r["lo"][s]), _lib.ptr(r["lr"][s]),
                                         _lib.ptr(r["lt"][s]), _lib.ptr(r["lu"][s]), _lib.ptr(r["lc"][s]), _lib.ptr(r["le"][s]),
                                         _lib.ptr(r["lf"][s]), mode))
            _lib.check(lib.xv_cartpole_step(self.ec._h, _lib.ptr(r["ca"][s]), _lib.ptr(r["co"][s]), _lib.ptr(r["cr"][s]),
                                            _lib.ptr(r["ct"][s]), _lib.ptr(r["cu"][s]), _lib.ptr(r["cf"][s]), mode))

    def set_overlap(self, on=True):
        """step_many issues consecutive fused steps alternately on two HIP streams; each wave of step k + 1 takes its envs
        over from the same wave of step k (xeno.h: overlapped xv_mixed_step_many).  Same results; calls of >= 64 steps over
        an even T only.  One overlapped share (or AnyMDP env) per device."""
        self.ea.set_step_many_overlap(on)

    @property
    def overlap_state(self):
        """1: the last step_many overlapped its ring cycles, 0: it did not, -1: the overlapped path failed on this device,
        -2: it overlapped, a hand-off expired and the call was replayed on one stream (results are right)"""
        return int(self.ea.lib.xv_mixed_step_many_overlap_state(self.ea._h))

    def rings_for_pack(self):
        r = self.ring
        return {"anymdp": dict(obs=r["ao"], action=r["aa"], reward=r["ar"], terminated=r["at"], truncated=r["au"]),
                "linds": dict(obs=r["lo"], reward=r["lr"], terminated=r["lt"], truncated=r["lu"]),
                "cartpole": dict(obs=r["co"], action=r["ca"], reward=r["cr"], terminated=r["ct"], truncated=r["cu"])}

    def pack(self, out):
        """the finished T-step chunk of the three families -> `out` (uint8 [chunk.bytes_per_rank]) on the current stream"""
        return self.chunk.pack(self.rank, self.rings_for_pack(), out)

    def check_errors(self):
        return self.ea.check_errors() | self.el.check_errors() | self.ec.check_errors()

    def algorithmic_bytes_per_vector_step(self):
        """562 B per anymdp env-step (fp64 CDF, S = 64), 432 B per linds, 74 B per cartpole"""
        return 562 * self.n["anymdp"] + 432 * self.n["linds"] + 74 * self.n["cartpole"]

    def close(self):
        for e in (self.ea, self.el, self.ec):
            e.close()
