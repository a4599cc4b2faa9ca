"""Randomised soak of the MazeWorld kernels against the CPU oracle (not collected by pytest; run as a script on a GPU box,
`PYTHONPATH=.:tests python tests/soak_maze.py [seconds]`): random mazes from the seed-compatible sampler, random texture
libraries, resolutions, typings and action tables; every step's pose (1e-9: device sin / cos against libm), cell, counters, flags
and reward against the oracle, and — with the oracle's pose set to the device's — whole frames BYTE FOR BYTE (the default exact
filter, i.e. the speculated one) on random steps."""
import sys
import time

import numpy as np

import oracle
from xenoverse_amd.mazeworld import (DEFAULT_ACTION_SPACE_16, DEFAULT_ACTION_SPACE_32, MazeTaskSampler, MazeWorldVecEnv,
                                     build_tables, make_texture_library)

MODES = {"disabled": 0, "next_step": 1, "same_step": 2}


def _np(t):
    return t.detach().cpu().numpy()


class Mismatch(Exception):
    pass


def _check(ok, what):
    if not ok:
        raise Mismatch(what)


def soak(rng, seed):
    n_task = int(rng.randint(1, 5))
    tasks = [MazeTaskSampler(n_range=(int(rng.randint(7, 11)), int(rng.randint(11, 22))), seed=seed + k,
                             n_wall_textures=int(rng.randint(1, 6)), n_ground_textures=int(rng.randint(1, 4)),
                             n_ceiling_textures=int(rng.randint(1, 4))) for k in range(n_task)]
    tex = make_texture_library(5, 3, 3, seed=seed % 997)
    tab = build_tables(tasks)
    res = [(32, 32), (64, 64), (48, 40), (40, 72), (128, 128), (24, 96), (160, 120)][int(rng.randint(0, 7))]   # the last: rows mapping
    per = int(rng.randint(1, max(2, 60000 // (res[0] * res[1] * n_task) + 1)))
    env_task = np.repeat(np.arange(n_task, dtype=np.int32), per)
    rng.shuffle(env_task)
    n = len(env_task)
    typing = str(rng.choice(["numpy2", "numpy2", "numba"]))
    space = str(rng.choice(["Discrete16", "Discrete32"]))
    table = np.array(DEFAULT_ACTION_SPACE_16 if space == "Discrete16" else DEFAULT_ACTION_SPACE_32, np.float64)
    mode = str(rng.choice(["same_step", "next_step"]))
    cio = bool(rng.randint(0, 2))
    max_steps = int(rng.randint(6, 30))
    env = MazeWorldVecEnv(n, resolution=res, textures=tex, autoreset_mode=mode, max_steps=max_steps, action_space_type=space,
                          typing=typing, command_in_observation=cio, seed=seed)
    env.set_task(tasks, env_task_index=env_task)
    mapping = str(rng.choice(["auto", "auto", "rows", "columns"]))
    env.set_raycast_mapping(mapping)
    ora = oracle.MazeOracle(tab, tex, env_task, resolution=res, max_steps=max_steps, command_in_observation=cio)
    f0, _ = env.reset()
    ora.reset()
    fo, co = ora.render(n_threads=8, typing="numba" if typing == "numba" else "stub")
    _check(np.array_equal(_np(f0), fo), "reset frames")
    T = int(rng.randint(5, 40))
    compared = 1
    for t in range(T):
        a = rng.randint(0, len(table), n).astype(np.int32)
        frames, r, term, trunc, info = env.step(a)
        ro, teo, tro = ora.step(table[a], MODES[mode])
        st = env.get_state()
        _check(np.max(np.abs(_np(st["pos"]) - ora.pos)) < 1e-9 and np.max(np.abs(_np(st["ori"]) - ora.ori)) < 1e-9, "pose")
        _check(np.array_equal(_np(st["grid"]), ora.grid) and np.array_equal(_np(st["steps"]), ora.steps), "cell / steps")
        _check(np.array_equal(_np(term).astype(np.uint8), teo) and np.array_equal(_np(trunc).astype(np.uint8), tro), "flags")
        _check(np.array_equal(_np(r), ro), "reward")
        ora.pos[:] = _np(st["pos"]); ora.ori[:] = _np(st["ori"])      # no compounding of last-bit differences
        if rng.random_sample() < 0.3 or tro.any():
            fo, co = ora.render(n_threads=8, typing="numba" if typing == "numba" else "stub")
            bad = int((_np(frames) != fo).sum())
            _check(bad == 0, "frames at step %d: %d bytes differ" % (t, bad))
            _check(np.array_equal(_np(info["command"]), co), "command colour")
            compared += 1
    env.close()
    return "maze tasks=%d envs=%d res=%s typing=%s %s mode=%s cmd_in_obs=%d steps=%d frames_compared=%d (%d pixels)" % (
        n_task, n, res, typing, space, mode, cio, T, compared, compared * n * res[0] * res[1])


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    t_end = time.time() + budget
    master = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
    n, px = 0, 0
    while time.time() < t_end:
        seed = int(master.randint(1, 1 << 30))
        try:
            line = soak(np.random.RandomState(seed), seed)
        except Mismatch as ex:
            print("MISMATCH with seed %d: %s" % (seed, ex), flush=True)
            sys.exit(1)
        n += 1
        px += int(line.split("(")[-1].split()[0])
        print("ok seed=%d %s" % (seed, line), flush=True)
    print("TOTAL %d configurations, %d pixels compared byte for byte, 0 mismatches" % (n, px))
