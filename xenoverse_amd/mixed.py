"""MixedBatch — several environment families stepped concurrently on one GPU (BASELINE.json config 5: anymdp +
linds + metacontrol, 262,144 envs sharded over 8 GPUs = per GPU 16,384 anymdp + 8,192 linds + 8,192 cartpole).

Each family keeps its own engine.  `streams="shared"` (default) launches all families on the caller's stream;
`streams="separate"` gives every family a private HIP stream so that their step kernels can overlap, with
`sync()` making the caller's stream wait for all of them.  Measured on one MI355X at the config-5 share
(`scripts/bench_families.py --families mixed`): the three step kernels take 16.4 us back to back on one stream and
47.6 us on three streams — they are 4-8 us each, so the cross-stream event waits a dependent step loop needs
(actions in, observations out) cost more than the overlap saves; separate streams pay off only when a family's
step is long (MazeWorld frames) or the consumer does not need all families at every step.  A family's trajectory
is identical either way and identical to what it would be stepped alone: engines share nothing (no global RNG).
Across GPUs each rank owns a contiguous slice of every family (`distributed.shard_range`).
"""
import torch

from .engine import Engine


class MixedBatch(object):
    def __init__(self, device="cuda:0", seed=0, streams="shared"):
        assert streams in ("shared", "separate")
        self.device = torch.device(device)
        self.seed = int(seed)
        self.separate = streams == "separate"
        self.envs = {}
        self.streams = {}

    def add(self, name, env_cls, num_envs, env_id_base=0, **kwargs):
        """Create `env_cls(num_envs, engine=<engine on a private stream>, **kwargs)` under `name`."""
        st = torch.cuda.Stream(device=self.device) if self.separate else torch.cuda.current_stream(self.device)
        eng = Engine(self.device, seed=self.seed, env_id_base=env_id_base, stream=st)
        env = env_cls(num_envs, engine=eng, **kwargs)
        env._own_engine = True   # closed with the env
        self.envs[name] = env
        self.streams[name] = st
        return env

    def _on(self, name):
        st = self.streams[name]
        if self.separate:
            st.wait_stream(torch.cuda.current_stream(self.device))   # inputs produced on the caller's stream
        return torch.cuda.stream(st)

    def set_task(self, tasks):
        for name, t in tasks.items():
            with self._on(name):
                if isinstance(t, tuple):
                    self.envs[name].set_task(t[0], env_task_index=t[1])
                else:
                    self.envs[name].set_task(t)

    def reset(self):
        out = {}
        for name, env in self.envs.items():
            with self._on(name):
                out[name] = env.reset()
        self.sync()
        return out

    def step(self, actions):
        """actions: dict name -> batched action.  Launches every family's step on its own stream, then makes the
        caller's stream wait for all of them.  -> dict name -> (obs, reward, terminated, truncated, infos)"""
        out = {}
        for name, env in self.envs.items():
            with self._on(name):
                out[name] = env.step(actions[name])
        self.sync()
        return out

    def sync(self):
        if not self.separate:
            return
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams.values():
            cur.wait_stream(st)

    @property
    def num_envs(self):
        return sum(e.num_envs for e in self.envs.values())

    def close(self):
        for env in self.envs.values():
            env.close()
