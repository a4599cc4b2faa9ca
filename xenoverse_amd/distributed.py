"""Sharding of an env batch over the GPUs of a node and the one collective the path has.

Stepping needs no communication: env instances are independent and task tables are read-only, so rank r owns
the contiguous env range `shard_range(n_total, r, world)` and the tasks those envs reference.  Draws are keyed
by the GLOBAL env id (Engine(env_id_base=lo)), so a sharded run reproduces the unsharded trajectories bit for
bit whatever the GPU count.  The only exchange step is optional: an all-gather of finished rollout chunks so
that every rank (learner replica) sees the whole batch.  `torch.distributed` is the transport — backend "nccl"
is RCCL over xGMI on ROCm, "gloo" on CPU (tests) — one in-place `all_gather_into_tensor` of a uint8 payload.
"""
import numpy as np
import torch

REC_BYTES = 8    # one 64-bit word per env-step: obs u16 | action u8 | flags u8 | reward f32   (include/xeno.h)


def shard_range(n_total, rank, world):
    """contiguous, near-equal ranges: rank r owns [lo, hi)"""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_env_task(env_task, rank, world):
    """Slice a global env->task map to this rank and renumber the tasks it references.
    -> (lo, hi, local_env_task int32[hi-lo], task_ids int64[k]) with env_task[lo:hi] == task_ids[local_env_task]"""
    env_task = np.asarray(env_task)
    lo, hi = shard_range(len(env_task), rank, world)
    ids, local = np.unique(env_task[lo:hi], return_inverse=True)
    return lo, hi, local.astype(np.int32), ids.astype(np.int64)


def pack_records(obs, action, reward, terminated, truncated, out=None):
    """[T, N] int32 / int32 / float32 / uint8 / uint8 -> uint8 [T, N, 8]: one 64-bit record per env-step (obs < 65536,
    action < 256).  Device tensors are packed by one HIP kernel (xv_pack_rollout, a coalesced 8-byte store per
    record) on the current stream; CPU tensors (the gloo tests of the N > 1 path) with torch integer ops."""
    T, N = obs.shape
    if out is None:
        out = torch.empty((T, N, REC_BYTES), dtype=torch.uint8, device=obs.device)
    if obs.is_cuda:
        from . import _lib
        lib = _lib.load()
        args = [obs.contiguous(), action.contiguous(), reward.contiguous(), terminated.contiguous(), truncated.contiguous()]
        assert out.is_contiguous()
        st = torch.cuda.current_stream(obs.device).cuda_stream
        _lib.check(lib.xv_pack_rollout(st, T * N, *[_lib.ptr(a) for a in args], _lib.ptr(out)))
        return out
    lo = (obs.to(torch.int64) & 0xFFFF) | ((action.to(torch.int64) & 0xFF) << 16) | \
         ((terminated != 0).to(torch.int64) << 24) | ((truncated != 0).to(torch.int64) << 25)
    hi = reward.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    out.copy_((lo | (hi << 32)).contiguous().view(torch.uint8).view(T, N, REC_BYTES))
    return out


def unpack_records(rec):
    """inverse of pack_records (rec uint8 [..., 8]) -> obs int32, action int32, reward float32, terminated, truncated uint8"""
    r = rec.contiguous()
    lead = tuple(r.shape[:-1])
    if r.is_cuda:
        from . import _lib
        lib = _lib.load()
        d = r.device
        obs = torch.empty(lead, dtype=torch.int32, device=d); act = torch.empty(lead, dtype=torch.int32, device=d)
        rew = torch.empty(lead, dtype=torch.float32, device=d)
        te = torch.empty(lead, dtype=torch.uint8, device=d); tr = torch.empty(lead, dtype=torch.uint8, device=d)
        st = torch.cuda.current_stream(d).cuda_stream
        _lib.check(lib.xv_unpack_rollout(st, obs.numel(), _lib.ptr(r), _lib.ptr(obs), _lib.ptr(act), _lib.ptr(rew),
                                         _lib.ptr(te), _lib.ptr(tr)))
        return obs, act, rew, te, tr
    w = r.view(torch.int64).view(lead)
    obs = (w & 0xFFFF).to(torch.int32)
    act = ((w >> 16) & 0xFF).to(torch.int32)
    te = ((w >> 24) & 1).to(torch.uint8)
    tr = ((w >> 25) & 1).to(torch.uint8)
    rew = (w >> 32).to(torch.int32).view(torch.float32)      # arithmetic shift keeps the 32 reward bits
    return obs, act, rew, te, tr


def pack_records_f32(obs, reward, terminated, truncated, action=None, out=None):
    """Float-observation families (LinDS, CartPole): obs float32 [T, N, D], reward float32 [T, N], terminated / truncated uint8
    [T, N], action int32 [T, N] or None -> uint8 [T, N, 4 (D + 2)]: D fp32 observation words, the reward, a flag word (bit 0
    terminated, bit 1 truncated, bits 8-31 the discrete action).  Device tensors: one HIP kernel (xv_pack_rollout_f32, a
    thread per 32-bit word); CPU tensors (the gloo tests of the N > 1 path): torch integer ops."""
    T, N, D = obs.shape
    if out is None:
        out = torch.empty((T, N, 4 * (D + 2)), dtype=torch.uint8, device=obs.device)
    assert out.is_contiguous() and out.numel() == T * N * 4 * (D + 2)
    if obs.is_cuda:
        from . import _lib
        lib = _lib.load()
        args = [obs.contiguous(), reward.contiguous(), terminated.contiguous(), truncated.contiguous(),
                None if action is None else action.contiguous()]
        st = torch.cuda.current_stream(obs.device).cuda_stream
        _lib.check(lib.xv_pack_rollout_f32(st, T * N, D, *[_lib.ptr(a) for a in args], _lib.ptr(out)))
        return out
    w = torch.empty((T, N, D + 2), dtype=torch.int32)
    w[..., :D] = obs.contiguous().view(torch.int32)
    w[..., D] = reward.contiguous().view(torch.int32)
    flags = (terminated != 0).to(torch.int64) | ((truncated != 0).to(torch.int64) << 1)
    if action is not None:
        flags = flags | ((action.to(torch.int64) & 0xFFFFFF) << 8)
    w[..., D + 1] = torch.where(flags >= 2**31, flags - 2**32, flags).to(torch.int32)      # the 32 bits, as a signed word
    out.view(-1).copy_(w.view(torch.uint8).view(-1))
    return out


def unpack_records_f32(rec, obs_dim, with_action=False):
    """inverse of pack_records_f32 (rec uint8 [..., 4 (obs_dim + 2)]) -> obs float32 [..., obs_dim], reward float32,
    terminated, truncated uint8 (and action int32 when with_action)"""
    r = rec.contiguous()
    lead = tuple(r.shape[:-1])
    D = int(obs_dim)
    assert r.shape[-1] == 4 * (D + 2)
    if r.is_cuda:
        from . import _lib
        lib = _lib.load()
        d = r.device
        obs = torch.empty(lead + (D,), dtype=torch.float32, device=d)
        rew = torch.empty(lead, dtype=torch.float32, device=d)
        te = torch.empty(lead, dtype=torch.uint8, device=d)
        tr = torch.empty(lead, dtype=torch.uint8, device=d)
        act = torch.empty(lead, dtype=torch.int32, device=d) if with_action else None
        st = torch.cuda.current_stream(d).cuda_stream
        _lib.check(lib.xv_unpack_rollout_f32(st, rew.numel(), D, _lib.ptr(r), _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(te),
                                             _lib.ptr(tr), _lib.ptr(act)))
        return (obs, rew, te, tr, act) if with_action else (obs, rew, te, tr)
    w = r.view(torch.int32).view(lead + (D + 2,))
    obs = w[..., :D].contiguous().view(torch.float32)
    rew = w[..., D].contiguous().view(torch.float32)
    f = w[..., D + 1].to(torch.int64) & 0xFFFFFFFF
    te = (f & 1).to(torch.uint8)
    tr = ((f >> 1) & 1).to(torch.uint8)
    if with_action:
        return obs, rew, te, tr, (f >> 8).to(torch.int32)
    return obs, rew, te, tr


class MixedChunk(object):
    """Byte layout of one rank's T-step rollout chunk of a mixed batch (BASELINE config 5: anymdp + linds + cartpole) and of
    the all-gathered buffer.  A rank's chunk is three record blocks back to back,

        [T, n_a, 8]  AnyMDP 8-byte records | [T, n_l, 4 (16 + 2)]  LinDS records | [T, n_c, 4 (4 + 2)]  CartPole records

    each padded to the largest share of any rank (an all-gather moves equal byte counts; with the env counts of config 5 no
    share needs padding).  Shares follow `shard_range` per family, so rank r's block of family f holds the global envs
    shard_range(N_f, r, world) of that family, and concatenating the ranks' blocks in rank order is the unsharded batch."""
    LINDS_DIM, CART_DIM = 16, 4

    def __init__(self, T, n_anymdp, n_linds, n_cartpole, world):
        self.T, self.world = int(T), int(world)
        self.n_total = {"anymdp": int(n_anymdp), "linds": int(n_linds), "cartpole": int(n_cartpole)}
        self.rec = {"anymdp": REC_BYTES, "linds": 4 * (self.LINDS_DIM + 2), "cartpole": 4 * (self.CART_DIM + 2)}
        self.share = {f: [shard_range(n, r, self.world) for r in range(self.world)] for f, n in self.n_total.items()}
        self.n_max = {f: max(hi - lo for lo, hi in sh) for f, sh in self.share.items()}
        self.offset, off = {}, 0
        for f in ("anymdp", "linds", "cartpole"):
            self.offset[f] = off
            off += self.T * self.n_max[f] * self.rec[f]
        self.bytes_per_rank = (off + 15) // 16 * 16

    def n_local(self, family, rank):
        lo, hi = self.share[family][rank]
        return hi - lo

    def block(self, buf, family, rank=None):
        """view [T, n_local, rec] of family's block inside a rank's chunk `buf` (uint8 [bytes_per_rank]); rank: whose chunk it
        is (its share may be smaller than the padded block)"""
        n = self.n_max[family] if rank is None else self.n_local(family, rank)
        o = self.offset[family]
        return buf[o:o + self.T * self.n_max[family] * self.rec[family]].view(self.T, self.n_max[family], self.rec[family])[:, :n]

    def pack(self, rank, rings, out):
        """rings: {"anymdp": dict(obs, action, reward, terminated, truncated), "linds": dict(obs, reward, terminated,
        truncated), "cartpole": dict(obs, action, reward, terminated, truncated)}, each [T, n_local(...)] -> fills `out`
        (uint8 [bytes_per_rank]) on the current stream"""
        a, l, c = rings["anymdp"], rings["linds"], rings["cartpole"]
        full = all(self.n_local(f, rank) == self.n_max[f] for f in self.rec)
        tgt = {f: (self.block(out, f) if full else torch.empty((self.T, self.n_local(f, rank), self.rec[f]), dtype=torch.uint8,
                                                                device=out.device)) for f in self.rec}
        pack_records(a["obs"], a["action"], a["reward"], a["terminated"], a["truncated"], out=tgt["anymdp"])
        pack_records_f32(l["obs"], l["reward"], l["terminated"], l["truncated"], None, out=tgt["linds"])
        pack_records_f32(c["obs"], c["reward"], c["terminated"], c["truncated"], c["action"], out=tgt["cartpole"])
        if not full:
            for f in self.rec:
                self.block(out, f, rank).copy_(tgt[f])
        return out

    def unpack(self, gathered):
        """gathered: uint8 [world, bytes_per_rank] -> {"anymdp": (obs, action, reward, terminated, truncated), "linds": (obs,
        reward, terminated, truncated), "cartpole": (obs, reward, terminated, truncated, action)}, each [T, N_family] in
        global env order (rank blocks concatenated)"""
        out = {}
        for f in ("anymdp", "linds", "cartpole"):
            parts = []
            for r in range(self.world):
                b = self.block(gathered[r], f, r).contiguous()
                if f == "anymdp":
                    parts.append(unpack_records(b))
                elif f == "linds":
                    parts.append(unpack_records_f32(b, self.LINDS_DIM))
                else:
                    parts.append(unpack_records_f32(b, self.CART_DIM, with_action=True))
            out[f] = tuple(torch.cat([p[k] for p in parts], dim=1) for k in range(len(parts[0])))
        return out


_RCCL_COMM_SEQ = [0]      # communicators are created in the same order on every rank: the key of an id carries the count


def _rendezvous_store(rank, world, store, host, port, timeout_s):
    """The key-value store that carries the RCCL unique id from rank 0 to the others — an EXISTING one whenever there is one:
    1. `store` handed in by the caller (anything with set / get);
    2. the default process group's store (torch.distributed is initialised: bench.py, trainers);
    3. the launcher's own store (torchrun / torch.distributed.run hosts a TCPStore at MASTER_ADDR:MASTER_PORT for its
       workers, TORCHELASTIC_USE_AGENT_STORE=True): joined as a client;
    4. none of these: rank 0 hosts a TCPStore at `port` (default MASTER_PORT — nobody else is serving it in that case) and
       keeps it for the life of the communicator: a process that also wants `dist.init_process_group(env://)` on the same
       port must call it FIRST (case 2 then applies) — documented in INTEGRATION.md §4.
    No second port is taken (earlier rounds opened MASTER_PORT + 1, which another job on the node may own)."""
    import datetime
    import os
    if store is not None:
        return store
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        from torch.distributed.distributed_c10d import _get_default_store
        return dist.PrefixStore("xv_rccl", _get_default_store())
    from torch.distributed import TCPStore
    host = host or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port or os.environ.get("MASTER_PORT", "29500"))
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true"
    tmo = datetime.timedelta(seconds=timeout_s)
    if agent:
        return dist.PrefixStore("xv_rccl", TCPStore(host, port, None, False, timeout=tmo))
    return dist.PrefixStore("xv_rccl", TCPStore(host, port, world, rank == 0, timeout=tmo))


class RcclComm(object):
    """A RCCL communicator made through the C-ABI (xv_rccl_*), without a process group of its own: rank 0 creates the 128-byte
    unique id and an existing key-value store carries it to the others (`_rendezvous_store`: the caller's, the default
    process group's, the launcher's — a fresh TCPStore on MASTER_PORT only when there is none).  rank / world default to the
    launcher's environment (RANK, WORLD_SIZE)."""

    def __init__(self, engine, rank=None, world=None, host=None, port=None, timeout_s=120, store=None):
        import ctypes as C
        import os
        from . import _lib
        self.lib = engine.lib
        self.engine = engine
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        ident = C.create_string_buffer(128)
        if self.world > 1:
            st = _rendezvous_store(self.rank, self.world, store, host, port, timeout_s)
            # the restart attempt is part of the key: after an elastic restart the agent's store still holds the previous
            # attempt's ids, and a rank must never pick one of those up before rank 0 has written the new one
            key = "id_%s_%d" % (os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), _RCCL_COMM_SEQ[0])
            _RCCL_COMM_SEQ[0] += 1
            if self.rank == 0:
                _lib.check(self.lib.xv_rccl_unique_id(ident))
                st.set(key, ident.raw)
            else:
                ident.raw = bytes(st.get(key))[:128]
            self._store = st
        else:
            _lib.check(self.lib.xv_rccl_unique_id(ident))
        h = C.c_void_p()
        _lib.check(self.lib.xv_rccl_comm_create(engine.handle, self.world, self.rank, ident, C.byref(h)))
        self.handle = h

    def all_gather(self, local, out):
        """out[world, ...] <- every rank's `local` (device tensors), asynchronously on the engine's stream"""
        from . import _lib
        assert local.is_cuda and out.is_cuda and local.is_contiguous() and out.is_contiguous()
        nbytes = local.numel() * local.element_size()
        assert out.numel() * out.element_size() == nbytes * self.world
        _lib.check(self.lib.xv_rollout_allgather(self.engine.handle, self.handle, _lib.ptr(local), _lib.ptr(out), nbytes))

    def count(self):
        """ncclCommCount: the number of ranks the communicator really spans"""
        import ctypes as C
        from . import _lib
        n = C.c_int(0)
        _lib.check(self.lib.xv_rccl_comm_count(self.handle, C.byref(n)))
        return int(n.value)

    def close(self):
        if getattr(self, "handle", None) is not None:
            self.lib.xv_rccl_comm_destroy(self.handle)
            self.handle = None


class RolloutGather(object):
    """All-gather of equally sized per-rank rollout chunks.  On GPU it runs on its own HIP stream so that the
    next chunk's stepping overlaps the transfer; call wait() before reading `out`.
    transport: "torch" = torch.distributed (backend "nccl" is RCCL on ROCm, "gloo" on CPU); "rccl" = ncclAllGather through
    the C-ABI (xv_rollout_allgather) on a communicator made without a process group."""

    def __init__(self, chunk_shape, dtype=torch.uint8, device="cpu", group=None, transport="torch", rank=None, world=None):
        self.transport = transport
        self.is_cuda = torch.device(device).type == "cuda"
        self.stream = torch.cuda.Stream(device=device) if self.is_cuda else None
        if transport == "rccl":
            if not self.is_cuda:
                raise ValueError("the rccl transport moves device buffers")
            from .engine import Engine
            self._engine = Engine(device, stream=self.stream)       # the collective is launched on the side stream
            self.comm = RcclComm(self._engine, rank=rank, world=world)
            self.world, self.rank = self.comm.world, self.comm.rank
        else:
            import torch.distributed as dist
            self.dist = dist
            self.group = group
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
        self.local = torch.empty(tuple(chunk_shape), dtype=dtype, device=device)
        self.out = torch.empty((self.world,) + tuple(chunk_shape), dtype=dtype, device=device)
        self._use_into = True

    def close(self):
        if self.transport == "rccl":
            self.comm.close()
            self._engine.close()

    def _gather(self):
        if self.transport == "rccl":
            self.comm.all_gather(self.local, self.out)
            return
        if self._use_into:
            try:
                self.dist.all_gather_into_tensor(self.out, self.local, group=self.group)
                return
            except (RuntimeError, NotImplementedError, AttributeError):
                self._use_into = False     # a backend without the fused form: list form below
        self.dist.all_gather([self.out[r] for r in range(self.world)], self.local, group=self.group)

    def launch(self):
        """gather self.local from all ranks into self.out (asynchronously on GPU)"""
        if self.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.local.device))
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                self._gather()
        else:
            self._gather()

    def wait(self):
        if self.is_cuda:
            torch.cuda.current_stream(self.local.device).wait_stream(self.stream)
        return self.out
