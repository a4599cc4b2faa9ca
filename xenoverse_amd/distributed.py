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

REC_BYTES = 14   # obs i32 | action i32 | reward f32 | terminated u8 | truncated u8   


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
    """[T, N] int32 / int32 / float32 / uint8 / uint8 -> uint8 [T, N, 14] records (one per env-step)"""
    T, N = obs.shape
    if out is None:
        out = torch.empty((T, N, REC_BYTES), dtype=torch.uint8, device=obs.device)
    out[..., 0:4] = obs.contiguous().view(torch.uint8).view(T, N, 4)
    out[..., 4:8] = action.contiguous().view(torch.uint8).view(T, N, 4)
    out[..., 8:12] = reward.contiguous().view(torch.uint8).view(T, N, 4)
    out[..., 12] = terminated
    out[..., 13] = truncated
    return out


def unpack_records(rec):
    """inverse of pack_records (rec uint8 [..., 14])"""
    r = rec.contiguous()
    lead = r.shape[:-1]
    obs = r[..., 0:4].contiguous().view(torch.int32).view(lead)
    act = r[..., 4:8].contiguous().view(torch.int32).view(lead)
    rew = r[..., 8:12].contiguous().view(torch.float32).view(lead)
    return obs, act, rew, r[..., 12], r[..., 13]


class RolloutGather(object):
    """All-gather of equally sized per-rank rollout chunks.  On GPU it runs on its own HIP stream so that the
    next chunk's stepping overlaps the transfer; call wait() before reading `out`."""

    def __init__(self, chunk_shape, dtype=torch.uint8, device="cpu", group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.local = torch.empty(tuple(chunk_shape), dtype=dtype, device=device)
        self.out = torch.empty((self.world,) + tuple(chunk_shape), dtype=dtype, device=device)
        self.is_cuda = torch.device(device).type == "cuda"
        self.stream = torch.cuda.Stream(device=device) if self.is_cuda else None
        self._use_into = True

    def _gather(self):
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
