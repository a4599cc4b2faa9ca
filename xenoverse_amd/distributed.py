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


class RcclComm(object):
    """A RCCL communicator made through the C-ABI (xv_rccl_*), without torch.distributed: rank 0 creates the 128-byte
    unique id, a `torch.distributed.TCPStore` (plain key-value rendezvous, no process group) carries it to the others.
    rank / world / address default to the launcher's environment (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT + 1)."""

    def __init__(self, engine, rank=None, world=None, host=None, port=None, timeout_s=120):
        import ctypes as C
        import datetime
        import os
        from . import _lib
        self.lib = engine.lib
        self.engine = engine
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        ident = C.create_string_buffer(128)
        if self.world > 1:
            from torch.distributed import TCPStore
            host = host or os.environ.get("MASTER_ADDR", "127.0.0.1")
            port = int(port or int(os.environ.get("MASTER_PORT", "29500")) + 1)
            store = TCPStore(host, port, self.world, self.rank == 0, timeout=datetime.timedelta(seconds=timeout_s))
            if self.rank == 0:
                _lib.check(self.lib.xv_rccl_unique_id(ident))
                store.set("xv_rccl_id", ident.raw)
            else:
                ident.raw = store.get("xv_rccl_id")
            self._store = store
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
