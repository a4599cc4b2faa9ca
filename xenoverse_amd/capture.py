"""Closed-loop stepping at kernel rate: [policy -> step] captured once in a torch.cuda.graph and replayed.

The reference's usage pattern is a Python loop per env object — action = policy(obs); obs, r, done, ... = env.step(action)
(xenoverse/anymdp/test_utils.py:45-57).  Vectorised, one iteration is a handful of tiny kernels (a 65,536-env AnyMDP
step is 5 us), and issuing them from Python costs 12-24 us per iteration: the host, not the GPU, sets the rate.  A
`CapturedLoop` records the iteration once — the caller's policy ops and the env's step launches — and replays it with
one `hipGraphLaunch` per `unroll` iterations.

What makes the step capturable (include/xeno.h, xv_engine_set_device_tick): the Philox launch tick lives in device
memory and is advanced by a node of the graph, so a replay draws fresh numbers; a step call allocates nothing,
synchronises nothing and keeps no host-side counter in its kernel arguments.  The replayed trajectory is bit for bit the
one the same calls issued eagerly produce (tests/test_gpu_capture.py: anymdp, linds, cartpole, mixed, 256 steps).

    env = AnyMDPVecEnv(n, copy=False); env.set_task(...); obs, _ = env.reset()
    loop = env.capture(lambda obs: my_policy(obs), obs)         # one eager warm-up iteration, then the capture
    for _ in range(1000):
        loop.replay()                                           # obs, reward, ... of the last step: loop.out
"""
import torch


class CapturedLoop(object):
    def __init__(self, step_fn, engines, policy_fn, obs, unroll=1, warmup=1, device=None, hold=None):
        """step_fn(actions) -> (obs, reward, terminated, truncated, infos) writing into FIXED buffers (copy=False envs);
        engines: the Engine objects whose launches the step issues; policy_fn(obs) -> actions (torch ops on the current
        stream; it may keep its own state in tensors it owns); obs: what reset() / the last step() returned.
        `warmup` >= 1 eager iterations are REAL steps (they also tell which buffers the step writes); then `unroll`
        iterations are captured.  hold(on): env hook that pins the output set while the loop exists."""
        if warmup < 1:
            raise ValueError("capture needs at least one eager warm-up iteration (it is a real step)")
        self.engines = list(engines)
        self.device = torch.device(device if device is not None else self.engines[0].device)
        self.unroll = int(unroll)
        self._hold = hold
        self.graph = None
        self._tick_was = [e.device_tick for e in self.engines]      # restored by close()
        try:
            self._build(step_fn, policy_fn, obs, warmup)
        except Exception:
            self.close()      # a failed construction leaves the env as it found it: output sets alternate, infos complete, host tick
            raise

    def _build(self, step_fn, policy_fn, obs, warmup):
        if self._hold is not None:
            self._hold(True)
        for e in self.engines:
            e.set_device_tick(True)
        out = None
        for _ in range(int(warmup)):
            out = step_fn(policy_fn(obs))
            obs = out[0]
        self._obs_in = obs
        cur = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(cur)
        self.graph = torch.cuda.CUDAGraph()
        orig = [e.torch_stream for e in self.engines]
        try:
            with torch.cuda.graph(self.graph, stream=side):
                for e in self.engines:
                    e.set_stream(torch.cuda.current_stream(self.device))
                for e in self.engines:      # the unrolled steps read tick + 0 .. unroll - 1; one node advances the word
                    e.tick_batch(True)
                o = obs
                for _ in range(self.unroll):
                    out = step_fn(policy_fn(o))
                    o = out[0]
                for e in self.engines:
                    e.tick_batch(False)
        finally:
            for e, st in zip(self.engines, orig):
                try:
                    e.tick_batch(False)      # a capture that failed half way must not leave the batch open (no-op when closed)
                except Exception:
                    pass
                e.set_stream(st)
        cur.wait_stream(side)
        if not self._same_memory(out[0], obs):
            raise RuntimeError("the captured step does not write its observation where the policy reads it: build the env "
                               "with copy=False (fixed output buffers) and to_numpy=False")
        self.out = out
        self.steps_replayed = 0

    @staticmethod
    def _same_memory(a, b):
        if isinstance(a, dict):
            return all(CapturedLoop._same_memory(a[k], b[k]) for k in a)
        return a.data_ptr() == b.data_ptr() and a.shape == b.shape

    def replay(self, n=1):
        """n graph launches = n * unroll vector steps; returns the (static) outputs of the last step"""
        g = self.graph
        for _ in range(n):
            g.replay()
        self.steps_replayed += n * self.unroll
        return self.out

    def close(self):
        """drops the graph, releases the env's hold and puts every engine's tick back where the loop found it (host tick:
        eager launches then pay no tick kernel and step_many may replay its graphs again)"""
        self.graph = None
        if self._hold is not None:
            self._hold(False)
            self._hold = None
        for e, was in zip(self.engines, getattr(self, "_tick_was", [])):
            try:
                if not was and e.handle is not None and e.device_tick:
                    e.set_device_tick(False)
            except Exception:
                pass
        self._tick_was = []
