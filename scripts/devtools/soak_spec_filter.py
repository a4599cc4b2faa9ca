"""Dev soak (not shipped): the speculated exact texture filter (precision="exact") against the direct one ("exact_direct") on
many random batches — other seeds, resolutions, texture libraries, typings and action spaces than the unit test uses.
Prints one line per configuration: pixels compared, bytes that differ (must be 0)."""
import sys, time
import numpy as np
import torch
from xenoverse_amd.mazeworld import MazeTaskSampler, MazeWorldVecEnv, make_texture_library

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t_end = time.time() + budget
rng = np.random.RandomState(2026)
total_px, total_bad, k = 0, 0, 0
while time.time() < t_end:
    k += 1
    res = [(64, 64), (128, 128), (256, 256), (96, 48), (32, 32), (80, 200), (160, 120)][k % 7]
    n_task = int(rng.randint(2, 9))
    per = max(1, int(3.0e6 / (res[0] * res[1] * n_task)))
    seed = int(rng.randint(1, 1 << 30))
    typing = "numba" if k % 3 == 0 else "numpy2"
    tasks = [MazeTaskSampler(n_range=(int(rng.randint(7, 12)), int(rng.randint(12, 22))), seed=seed + j,
                             n_wall_textures=5, n_ground_textures=3, n_ceiling_textures=3) for j in range(n_task)]
    env_task = np.repeat(np.arange(n_task, dtype=np.int32), per)
    n = len(env_task)
    tex = make_texture_library(5, 3, 3, seed=seed % 1000)
    T = int(rng.randint(3, 30))
    a = rng.randint(0, 16, (T, n)).astype(np.int32)
    frames = {}
    for prec in ("exact", "exact_direct"):
        env = MazeWorldVecEnv(n, resolution=res, textures=tex, autoreset_mode="same_step", precision=prec, typing=typing,
                              seed=seed, command_in_observation=bool(k % 2))
        env.set_task(tasks, env_task_index=env_task)
        env.reset()
        for t in range(T):
            out = env.step(a[t])
        frames[prec] = out[0].clone()
        env.close()
    bad = int((frames["exact"] != frames["exact_direct"]).sum().item())
    px = frames["exact"].numel() // 3
    total_px += px
    total_bad += bad
    print("res %s typing %s tasks %d envs %d steps %d: %d pixels, %d differing bytes" % (res, typing, n_task, n, T, px, bad), flush=True)
print("TOTAL %d pixels, %d differing bytes" % (total_px, total_bad))
sys.exit(1 if total_bad else 0)
