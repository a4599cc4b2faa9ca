"""Dev probe (not shipped): how many pixels the speculated exact texture filter re-runs.  Run with
XV_LIB_PATH=scripts/devtools/_build/libxeno_mark.so (built with -DXV_MAZE_SPEC_MARK=1: re-run pixels are painted with a marker),
frames of precision="exact" are compared with themselves from precision="exact_direct" (which the marker build leaves alone)."""
import numpy as np
import torch
from xenoverse_amd.mazeworld import MazeTaskSampler, MazeWorldVecEnv, make_texture_library

for res, n_task, per in (((64, 64), 64, 64), ((256, 256), 16, 16)):
    tasks = [MazeTaskSampler(n_range=(15, 16), seed=k, n_wall_textures=8, n_ground_textures=4, n_ceiling_textures=4) for k in range(n_task)]
    tex = make_texture_library(8, 4, 4, seed=0)
    n = n_task * per
    a = np.random.RandomState(1).randint(0, 16, (40, n)).astype(np.int32)
    frames = {}
    for prec in ("exact", "exact_direct"):
        env = MazeWorldVecEnv(n, resolution=res, textures=tex, autoreset_mode="same_step", precision=prec, seed=1)
        env.set_task(tasks)
        env.reset()
        got = []
        for t in range(40):
            out = env.step(a[t])
            if t >= 32:
                got.append(out[0].clone())
        frames[prec] = torch.stack(got)
        env.close()
    marked = (frames["exact"] != frames["exact_direct"]).any(-1)
    px = marked.numel()
    print("%s: %d of %d pixels re-run = 1 in %.0f (%.2e); per 64-row block of a wave: %.2f" % (
        res, int(marked.sum()), px, px / max(1, int(marked.sum())), float(marked.sum()) / px, float(marked.sum()) / px * 4096))
    d = frames["exact_direct"][marked]            # true bytes of the re-run pixels
    z = (d == 0).any(-1).float().mean().item()
    allz = (d == 0).all(-1).float().mean().item()
    sat = (d == 255).any(-1).float().mean().item()
    print("   of the re-run pixels: %.1f %% have a channel that is 0, %.1f %% are black, %.1f %% have a channel at 255" % (100 * z, 100 * allz, 100 * sat))
    allpx = frames["exact_direct"].reshape(-1, 3)
    print("   of all pixels: %.1f %% have a channel that is 0" % (100 * (allpx == 0).any(-1).float().mean().item()))
