"""MazeWorldVecEnv — N procedurally generated 3-D mazes stepped and ray-cast per launch on one MI355X.

Mirrors the reference's interface (xenoverse/mazeworld/envs/maze_env.py: MazeWorldContinuous3D.__init__
:109-149, set_task :30-32, reset :34-45, step :50-66, action_control :151-162) behind the gymnasium VectorEnv
surface.  Same task dicts, same info keys (`steps`, `command` = RGB of the target
landmark).  GUI pieces (pygame rendering, god-view maps, keyboard control) are out of scope.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..engine import AUTORESET
from ..spaces import Box, Discrete
from ..vector import VectorEnv
from .tables import DEFAULT_ACTION_SPACE_16, DEFAULT_ACTION_SPACE_32, build_tables
from .textures import check_texture_library, make_texture_library


class _Tables(C.Structure):   # xv_maze_tables (include/xeno.h)
    _fields_ = [(k, C.c_void_p) for k in ("walls", "texts", "landmarks", "ints", "dbl", "commands", "lm_coord",
                                          "tex_walls", "tex_grounds", "tex_ceilings")] + \
               [(k, C.c_int32) for k in ("n_tex_walls", "n_tex_grounds", "n_tex_ceilings")]


class MazeWorldVecEnv(VectorEnv):
    ACTION_MODES = {"Continuous": 0, "Discrete16": 1, "Discrete32": 2}

    def __init__(self, num_envs, enable_render=False, render_scale=480, max_steps=5000, resolution=(320, 320),
                 visibility_3D=12.0, command_in_observation=False, action_space_type="Discrete16",
                 collision_dist=0.20, textures=None, device="cuda:0", seed=0, env_id_base=0,
                 autoreset_mode="same_step", to_numpy=False, engine=None, copy=True, with_final_obs=False,
                 precision="exact", typing="numpy2"):
        """Constructor arguments as MazeWorldContinuous3D (maze_env.py:110-118); the registered id `mazeworld-v2`
        uses resolution (256, 256), max_steps 5000, visibility_3D 12.0, Discrete16 (mazeworld/__init__.py:19-33).
        `textures`: dict(walls, grounds, ceilings) of float32 [n,256,256,3] arrays — `load_texture_library(dir)` reads a
        folder of wall* / ground* / ceiling* images the way the reference does (task_sampler.py:60-77); default = the
        procedural library in the reference's sizes (37 / 29 / 21; its JPG assets are not redistributed), so reference
        tasks and `MazeTaskSampler` defaults index it."""
        super().__init__(num_envs, device=device, seed=seed, env_id_base=env_id_base,
                         autoreset_mode=autoreset_mode, to_numpy=to_numpy, engine=engine, copy=copy)
        if enable_render:
            raise NotImplementedError("pygame rendering is out of scope of the GPU engine (use the frames)")
        if action_space_type not in self.ACTION_MODES:
            raise ValueError("Invalid Action Space Type {}. Can only accept Discrete16, Discrete32, Continuous"
                             .format(action_space_type))
        self.max_steps = int(max_steps)
        self.resolution = (int(resolution[0]), int(resolution[1]))
        self.visibility_3D = float(visibility_3D)
        self.command_in_observation = bool(command_in_observation)
        self.action_space_type = action_space_type
        self.collision_dist = float(collision_dist)
        self.with_final_obs = bool(with_final_obs)
        if precision not in ("exact", "f32", "exact_direct"):
            raise ValueError("precision must be 'exact' (the reference's typing, default), 'exact_direct' (the same bytes, "
                             "every pixel filtered directly: what 'exact' is tested against) or 'f32'")
        if typing not in ("numpy2", "numba"):
            raise ValueError("typing must be 'numpy2' (the reference's source as plain Python under NumPy 2, default) or "
                             "'numba' (the types numba infers: DDA and wall-column geometry in float64)")
        self.typing = typing
        self.precision = precision      # "f32": texture filter in float32, +-1 level on <= 0.5 % of the frame values
        self.inner_action_list = {"Discrete16": DEFAULT_ACTION_SPACE_16, "Discrete32": DEFAULT_ACTION_SPACE_32}.get(
            action_space_type)
        act = Box(-1, 1, shape=(2,), dtype=np.float32) if action_space_type == "Continuous" else \
            Discrete(len(self.inner_action_list))
        self._set_spaces(Box(0, 255, shape=(self.resolution[0], self.resolution[1], 3), dtype=np.uint8), act)
        self._textures = textures
        self._h = None

    @property
    def list_actions(self):
        return self.inner_action_list

    def set_task(self, tasks, env_task_index=None):
        tab = tasks if (isinstance(tasks, dict) and "walls" in tasks) else build_tables(tasks)
        tex = self._textures if self._textures is not None else make_texture_library()
        check_texture_library(tex)      # the kernels address 256 x 256 x 3 texels: any other size is an out-of-bounds read
        d = self.device
        dev = {k: torch.from_numpy(np.ascontiguousarray(tab[k])).to(d) for k in
               ("walls", "texts", "landmarks", "ints", "dbl", "commands", "lm_coord")}
        for k, name in (("tex_walls", "walls"), ("tex_grounds", "grounds"), ("tex_ceilings", "ceilings")):
            v = tex[name]
            dev[k] = (v.to(d, torch.float32) if torch.is_tensor(v)
                      else torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(d)).contiguous()
        n_task = int(dev["ints"].shape[0])
        if int(dev["texts"].max()) >= dev["tex_walls"].shape[0] or \
                int(dev["ints"][:, 3].max()) >= dev["tex_grounds"].shape[0] or \
                int(dev["ints"][:, 4].max()) >= dev["tex_ceilings"].shape[0]:
            raise ValueError("task texture ids exceed the texture library size")
        if env_task_index is None:
            if self.num_envs % n_task != 0:
                raise ValueError("num_envs is not a multiple of the task count; pass env_task_index")
            env_task = torch.arange(self.num_envs, device=d, dtype=torch.int32) // (self.num_envs // n_task)
        else:
            env_task = self._dev(env_task_index, torch.int32)
            if env_task.shape != (self.num_envs,) or int(env_task.min()) < 0 or int(env_task.max()) >= n_task:
                raise ValueError("env_task_index must be (num_envs,) with entries in [0, n_task)")
        dev["env_task"] = env_task.contiguous()
        self._close_agents()
        if self._h is not None:
            self.lib.xv_maze_destroy(self._h)
            self._h = None
        ct = _Tables(*[_lib.ptr(dev[k]) for k in ("walls", "texts", "landmarks", "ints", "dbl", "commands",
                                                   "lm_coord", "tex_walls", "tex_grounds", "tex_ceilings")],
                     int(dev["tex_walls"].shape[0]), int(dev["tex_grounds"].shape[0]),
                     int(dev["tex_ceilings"].shape[0]))
        h = C.c_void_p()
        W, H = self.resolution
        _lib.check(self.lib.xv_maze_create(self.engine.handle, self.num_envs, n_task, int(tab["NG"]),
                                           int(tab["n_cmd"]), self.max_steps, W, H,
                                           int(self.command_in_observation), self.collision_dist,
                                           self.visibility_3D, C.byref(ct), _lib.ptr(dev["env_task"]), C.byref(h)))
        self._h = h
        self._tab = dev
        if self.precision != "exact":
            _lib.check(self.lib.xv_maze_set_precision(h, 1 if self.precision == "f32" else 2))
        if self.typing == "numba":
            _lib.check(self.lib.xv_maze_set_typing(h, 1))
        n = self.num_envs
        self._frames = torch.zeros((n, W, H, 3), dtype=torch.uint8, device=d)
        self._final = torch.zeros((n, W, H, 3), dtype=torch.uint8, device=d) if self.with_final_obs else None
        self._cmd_rgb = torch.zeros((n, 3), dtype=torch.float32, device=d)
        self._reward = torch.zeros(n, dtype=torch.float32, device=d)
        self._term = torch.zeros(n, dtype=torch.uint8, device=d)
        self._trunc = torch.zeros(n, dtype=torch.uint8, device=d)
        self._steps = torch.zeros(n, dtype=torch.int32, device=d)
        self.task_set = True
        self.need_reset = True

    def _steps_now(self):
        self._renew("_steps")      # the launch writes every entry: a fresh buffer, handed out as it is (no copy)
        _lib.check(self.lib.xv_maze_get_state(self._h, None, None, None, _lib.ptr(self._steps), None, None, None, None))
        return self._steps if (self.copy and not self.to_numpy) else self._steps.clone()

    def reset(self, *, seed=None, options=None):
        if not self.task_set:
            raise Exception("Must call \"set_task\" before reset")   # maze_env.py:35-36
        mask = None
        if options is not None and options.get("reset_mask") is not None:
            mask = self._dev(options["reset_mask"], torch.uint8)
        self._detach("_frames", "_cmd_rgb")
        _lib.check(self.lib.xv_maze_reset(self._h, _lib.ptr(mask), _lib.ptr(self._frames), _lib.ptr(self._cmd_rgb)))
        self.need_reset = False
        return self._o(self._frames), {"steps": self._out(self._steps_now()),
                                                  "command": self._o(self._cmd_rgb)}

    def step(self, actions):
        if self.need_reset:
            raise Exception("Must \"reset\" before doing any actions")   # maze_env.py:51-52
        mode = self.ACTION_MODES[self.action_space_type]
        if mode == 0:
            a = self._dev(actions, torch.float64)
            assert a.shape == (self.num_envs, 2)
        else:
            a = self._dev(actions, torch.int32)
            assert a.shape == (self.num_envs,)
        # reference quirk: info["steps"] is read BEFORE do_action (maze_env.py:57)
        steps_before = self._steps_now()
        self._renew("_frames", "_reward", "_term", "_trunc", "_cmd_rgb")   # all fully written by the step
        _lib.check(self.lib.xv_maze_step(self._h, _lib.ptr(a), mode, _lib.ptr(self._frames), _lib.ptr(self._reward),
                                         _lib.ptr(self._term), _lib.ptr(self._trunc), _lib.ptr(self._cmd_rgb),
                                         _lib.ptr(self._final), AUTORESET[self.autoreset_mode]))
        infos = {"steps": self._out(steps_before), "command": self._of(self._cmd_rgb)}
        if self.with_final_obs and self.autoreset_mode == "same_step":
            infos["final_obs"] = self._o(self._final)
            infos["_final_obs"] = self._out((self._term | self._trunc).view(torch.bool))   # flags are 0 / 1 bytes: one op, no conversion
        return (self._of(self._frames), self._of(self._reward), self._obf(self._term),
                self._obf(self._trunc), infos)

    def set_move_kernel(self, kernel):
        """"auto" (default: nine or three lanes per env by batch size; from 10,240 envs up the nine-lane kernel walks only the
        envs that can move or touch a wall), "nine_lanes", "nine_lanes_compact", "three_lanes" or "lane_per_env":
        arrangements of the same move / collision arithmetic, identical results"""
        _lib.check(self.lib.xv_maze_set_move_kernel(self._h, {"lane_per_env": 0, "nine_lanes": 1, "three_lanes": 2,
                                                              "auto": 3, "nine_lanes_compact": 4}[kernel]))

    def set_raycast_mapping(self, mapping):
        """"auto" (default), "columns" or "rows": which lanes of the ray caster paint which pixels (xv_maze_set_raycast_mapping);
        the frames are the same bytes either way"""
        _lib.check(self.lib.xv_maze_set_raycast_mapping(self._h, {"auto": 0, "columns": 1, "rows": 2}[mapping]))

    def render_frames(self):
        """frames of the current state, without stepping"""
        self._detach("_frames", "_cmd_rgb")
        _lib.check(self.lib.xv_maze_render(self._h, _lib.ptr(self._frames), _lib.ptr(self._cmd_rgb)))
        return self._o(self._frames)

    def get_state(self):
        n, d = self.num_envs, self.device
        out = dict(pos=torch.empty((2, n), dtype=torch.float64, device=d), ori=torch.empty(n, dtype=torch.float64, device=d),
                   grid=torch.empty((2, n), dtype=torch.int32, device=d), steps=torch.empty(n, dtype=torch.int32, device=d),
                   cmd_idx=torch.empty(n, dtype=torch.int32, device=d), cmd_age=torch.empty(n, dtype=torch.int32, device=d),
                   need_reset=torch.empty(n, dtype=torch.uint8, device=d),
                   collision=torch.empty(n, dtype=torch.float64, device=d))
        _lib.check(self.lib.xv_maze_get_state(self._h, *[_lib.ptr(out[k]) for k in
                                              ("pos", "ori", "grid", "steps", "cmd_idx", "cmd_age", "need_reset",
                                               "collision")]))
        return out

    def set_state(self, pos=None, ori=None, steps=None, cmd_idx=None, cmd_age=None, need_reset=None):
        args = [None if pos is None else self._dev(pos, torch.float64),
                None if ori is None else self._dev(ori, torch.float64),
                None if steps is None else self._dev(steps, torch.int32),
                None if cmd_idx is None else self._dev(cmd_idx, torch.int32),
                None if cmd_age is None else self._dev(cmd_age, torch.int32),
                None if need_reset is None else self._dev(need_reset, torch.uint8)]
        _lib.check(self.lib.xv_maze_set_state(self._h, *[_lib.ptr(a) for a in args]))
        self.engine.sync()
        self.need_reset = False

    def get_target_location(self):
        """(distance, angle) of the target landmark relative to the agent, in grid units (maze_env.py:86-102)"""
        st = self.get_state()
        t = self._tab["env_task"].long()
        idx = torch.clamp(st["cmd_idx"].long(), max=self._tab["commands"].shape[1] - 1)
        cmd = self._tab["commands"][t, idx].long()
        tg = self._tab["lm_coord"][t, cmd].to(torch.float32)
        d = (tg - st["grid"].t().to(torch.float32))
        ang = torch.atan2(d[:, 1], d[:, 0]).to(torch.float64) - st["ori"]
        ang = torch.where(ang < -np.pi, ang + 2 * np.pi, torch.where(ang > np.pi, ang - 2 * np.pi, ang))
        return self._out(torch.sqrt((d * d).sum(1))), self._out(ang)

    def _close_agents(self):
        """agents hold device memory sized by this handle's batch: they go before it does"""
        for a in list(getattr(self, "_agents", ())):
            a.close()
        self._agents = []

    def close_extras(self, **kwargs):
        self._close_agents()
        if self._h is not None:
            self.lib.xv_maze_destroy(self._h)
            self._h = None
        self._tab = None
