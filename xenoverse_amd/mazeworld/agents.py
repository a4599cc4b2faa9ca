"""SmartSLAMAgent / OracleAgent — the reference's rule-based maze teachers for a whole batch, on the device.

Mirrors xenoverse/mazeworld/agents (agent_base.py:10-107, smart_slam_agent.py:105-231, oracle_agent.py): constructed
with `maze_env=` (a MazeWorldVecEnv with a Discrete16 / Discrete32 action space, agent_base.py:33-34), `step(observation,
r)` returns the actions for `env.step`.  The agents read the env's state on the device (pose, cell, command) and what the
ray caster exposes of the maze; `observation` and `r` are accepted and unused, as in the reference's policy.  pygame
rendering (`render=True`) is out of scope."""
import ctypes as C

import torch

from .. import _lib


class AgentBase(object):
    _ORACLE = False

    def __init__(self, **kwargs):
        if "maze_env" not in kwargs:
            raise Exception("Must use maze_env as arguments")                                   # agent_base.py:19-20
        if kwargs.get("render", False):
            raise NotImplementedError("agent rendering (pygame) is out of scope of the GPU engine")
        env = self.maze_env = kwargs["maze_env"]
        if env.list_actions is None:
            raise Exception("For smart agents, maze environment must use Discrete16 or Discrete32")   # :33-34
        if env._h is None:
            raise Exception("Must call \"set_task\" before creating an agent")
        self.short_term_memory_size = int(kwargs.get("short_term_memory_size", 3))
        self.memory_keep_ratio = float(kwargs.get("memory_keep_ratio", 1.0))
        self.keep_cost_map = bool(kwargs.get("keep_cost_map", False))
        self.lib = env.lib
        h = C.c_void_p()
        _lib.check(self.lib.xv_maze_agent_create(env._h, self.short_term_memory_size, self.memory_keep_ratio,
                                                 int(self._ORACLE), len(env.list_actions), int(self.keep_cost_map),
                                                 C.byref(h)))
        self._h = h
        self._env_handle = env._h.value
        if not hasattr(env, "_agents"):
            env._agents = []
        env._agents.append(self)
        self._action = torch.zeros(env.num_envs, dtype=torch.int32, device=env.device)

    def step(self, observation=None, r=None, exposed=None):
        """actions int32[num_envs] (device tensor) for the envs' present states.  `exposed` (uint8[N, NG, NG]): hand in
        maze_core._cell_exposed instead of letting the ray caster's walk produce it (parity hook)."""
        env = self.maze_env
        if self._h is None or env._h is None or env._h.value != self._env_handle:
            raise Exception("the env's task changed; create a new agent")
        ex = None if exposed is None else env._dev(exposed, torch.uint8).contiguous()
        self._action = torch.empty_like(self._action)
        _lib.check(self.lib.xv_maze_agent_act(self._h, _lib.ptr(ex), _lib.ptr(self._action)))
        return self._action

    def inspect(self, cost=False):
        """_mask_info, path head and cell_exposed (and _cost_map if kept) of the last decision, as device tensors"""
        env = self.maze_env
        n, NG, d = env.num_envs, int(env._tab["walls"].shape[-1]), env.device
        out = dict(mask=torch.empty((n, NG, NG), dtype=torch.uint8, device=d),
                   path=torch.empty((n, 5), dtype=torch.int32, device=d),
                   exposed=torch.empty((n, NG, NG), dtype=torch.uint8, device=d))
        if cost:
            out["cost"] = torch.empty((n, NG, NG), dtype=torch.float64, device=d)
        _lib.check(self.lib.xv_maze_agent_get(self._h, _lib.ptr(out["mask"]), _lib.ptr(out.get("cost")),
                                              _lib.ptr(out["path"]), _lib.ptr(out["exposed"])))
        return out

    def close(self):
        if self._h is not None:
            env = self.maze_env
            if env._h is not None and env._h.value == self._env_handle:
                self.lib.xv_maze_agent_destroy(self._h)
            self._h = None
            if self in getattr(env, "_agents", ()):
                env._agents.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SmartSLAMAgent(AgentBase):
    """explores by cost-minus-novelty, walks to the commanded landmark once it has been seen"""


class OracleAgent(SmartSLAMAgent):
    """SmartSLAMAgent that knows the whole maze from the start (oracle_agent.py)"""
    _ORACLE = True


def teacher_rollout(env, agent, T, frames=False):
    """T steps of env driven by agent, nothing leaving the device: dict(action, reward, terminated, truncated
    [, frames]) stacked over T.  The reference's data-collection loop (agent.step -> env.step)."""
    acts, rews, tes, trs, frs = [], [], [], [], []
    obs = None
    for _ in range(T):
        a = agent.step(obs, None)
        obs, r, te, tr, info = env.step(a)
        acts.append(a); rews.append(r); tes.append(te); trs.append(tr)
        if frames:
            frs.append(obs)
    out = dict(action=torch.stack(acts), reward=torch.stack([torch.as_tensor(x) for x in rews]),
               terminated=torch.stack([torch.as_tensor(x) for x in tes]),
               truncated=torch.stack([torch.as_tensor(x) for x in trs]))
    if frames:
        out["frames"] = torch.stack([torch.as_tensor(x) for x in frs])
    return out
