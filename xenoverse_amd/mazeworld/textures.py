"""Procedural textures for MazeWorld.

The reference ships 87 JPG textures (xenoverse/mazeworld/envs/img, loaded at
mazeworld/envs/task_sampler.py:59-78 into float32 arrays [n, 256, 256, 3]); those are reference assets and
are neither copied nor available on the GPU box.  This module generates deterministic stand-ins with the same
array contract (float32, values in [0, 255], shape [n, 256, 256, 3], wall / ground / ceiling libraries).
Integer lattice value-noise from numpy's legacy RandomState, so the bytes are identical on every machine.
"""
import numpy as np

TEX_SIZE = 256


def _value_noise(rng, size, cells):
    lat = rng.randint(0, 256, size=(cells + 1, cells + 1)).astype(np.float64)
    lat[-1, :] = lat[0, :]
    lat[:, -1] = lat[:, 0]          # tileable
    t = np.arange(size) * (cells / size)
    i0 = np.floor(t).astype(np.int64)
    f = t - i0
    f = f * f * (3.0 - 2.0 * f)
    a = lat[i0][:, i0]
    b = lat[i0 + 1][:, i0]
    c = lat[i0][:, i0 + 1]
    d = lat[i0 + 1][:, i0 + 1]
    fx, fy = f[:, None], f[None, :]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_texture(seed, kind):
    rng = np.random.RandomState(int(seed))
    base = rng.randint(40, 216, size=3).astype(np.float64)
    n = np.zeros((TEX_SIZE, TEX_SIZE))
    amp, tot = 1.0, 0.0
    for cells in (4, 8, 16, 32, 64):
        n += amp * _value_noise(rng, TEX_SIZE, cells)
        tot += amp
        amp *= 0.55
    n = n / tot / 255.0                      # [0, 1]
    img = np.empty((TEX_SIZE, TEX_SIZE, 3))
    for ch in range(3):
        img[..., ch] = base[ch] * (0.55 + 0.9 * n)
    if kind == "wall":                        # brick-like courses
        yy = (np.arange(TEX_SIZE) % 32 < 3)[None, :]
        xx = ((np.arange(TEX_SIZE)[:, None] + 16 * ((np.arange(TEX_SIZE)[None, :] // 32) % 2)) % 64 < 3)
        img[np.broadcast_to(yy, xx.shape) | xx] *= 0.6
    elif kind == "ground":
        img[(np.arange(TEX_SIZE) % 64 < 2)[:, None] | (np.arange(TEX_SIZE) % 64 < 2)[None, :]] *= 0.75
    return np.round(np.clip(img, 0, 255)).astype(np.float32)   # integral values, like a decoded image


def make_texture_library(n_walls=8, n_grounds=4, n_ceilings=4, seed=0):
    """-> dict(walls=[n_walls,256,256,3], grounds=[...], ceilings=[...]) float32 in [0,255]"""
    walls = np.stack([make_texture(seed * 1000 + k, "wall") for k in range(n_walls)])
    grounds = np.stack([make_texture(seed * 1000 + 300 + k, "ground") for k in range(n_grounds)])
    ceilings = np.stack([make_texture(seed * 1000 + 600 + k, "ceiling") for k in range(n_ceilings)])
    return dict(walls=walls, grounds=grounds, ceilings=ceilings)
