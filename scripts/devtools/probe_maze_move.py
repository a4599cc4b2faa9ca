"""What the move kernel costs on a compacted batch: 6,144 envs that all move (the ~37.5 % of config 4 whose Discrete16
action has a walk speed) against 16,384 envs with uniform actions.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xenoverse_amd import _lib
from xenoverse_amd.engine import AUTORESET
from xenoverse_amd.mazeworld import MazeTaskSampler, MazeWorldVecEnv, make_texture_library
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench_families import timed

tasks = [MazeTaskSampler(n_range=(15, 16), seed=k, n_wall_textures=8, n_ground_textures=4, n_ceiling_textures=4) for k in range(256)]
lib = make_texture_library(8, 4, 4, seed=0)
movers = torch.tensor([0, 11, 12, 13, 14, 15], dtype=torch.int32)
for per, which in ((64, "uniform"), (24, "movers"), (24, "uniform"), (64, "movers"), (40, "turners")):
    n = 256 * per
    env = MazeWorldVecEnv(n, resolution=(64, 64), textures=lib, autoreset_mode="same_step", action_space_type="Discrete16")
    env.set_task(tasks)
    env.reset()
    g = torch.Generator().manual_seed(1)
    if which == "uniform":
        a = torch.randint(0, 16, (n,), generator=g, dtype=torch.int32)
    elif which == "movers":
        a = movers[torch.randint(0, 6, (n,), generator=g)]
    else:
        a = torch.randint(1, 11, (n,), generator=g, dtype=torch.int32)
    a = a.to(env.device)

    def move():
        _lib.check(env.lib.xv_maze_step(env._h, _lib.ptr(a), 1, None, _lib.ptr(env._reward), _lib.ptr(env._term),
                                        _lib.ptr(env._trunc), None, None, AUTORESET["same_step"]))
    us = [timed(move, 20, 5) for _ in range(3)]
    print("%6d envs, %-8s actions: move+rules %s us" % (n, which, ["%.1f" % u for u in us]), flush=True)
    env.close()
