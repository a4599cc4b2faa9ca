"""Procedural textures for MazeWorld.

The reference ships 87 JPG textures (xenoverse/mazeworld/envs/img, loaded at
mazeworld/envs/task_sampler.py:59-78 into float32 arrays [n, 256, 256, 3]); those are reference assets and
are neither copied nor available on the GPU box.  This module generates deterministic stand-ins with the same
array contract (float32, values in [0, 255], shape [n, 256, 256, 3], wall / ground / ceiling libraries).
Integer lattice value-noise from numpy's legacy RandomState, so the bytes are identical on every machine.
"""
import os

import numpy as np

TEX_SIZE = 256
# library sizes of the reference's own texture folder (xenoverse/mazeworld/envs/img: 37 wall*, 29 ground*, 21 ceiling*
# files).  MazeTaskSampler draws texture ids with these bounds, so they are part of "the task for a seed".
REFERENCE_TEXTURE_COUNTS = (37, 29, 21)


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


_LIB_CACHE = {}


def make_texture_library(n_walls=None, n_grounds=None, n_ceilings=None, seed=0):
    """-> dict(walls=[n_walls,256,256,3], grounds=[...], ceilings=[...]) float32 in [0,255].  Default sizes: the
    reference's (37 / 29 / 21), so that a reference task's texture ids index it."""
    n_walls = REFERENCE_TEXTURE_COUNTS[0] if n_walls is None else n_walls
    n_grounds = REFERENCE_TEXTURE_COUNTS[1] if n_grounds is None else n_grounds
    n_ceilings = REFERENCE_TEXTURE_COUNTS[2] if n_ceilings is None else n_ceilings
    key = (n_walls, n_grounds, n_ceilings, seed)
    if key in _LIB_CACHE:
        return _LIB_CACHE[key]
    walls = np.stack([make_texture(seed * 1000 + k, "wall") for k in range(n_walls)])
    grounds = np.stack([make_texture(seed * 1000 + 300 + k, "ground") for k in range(n_grounds)])
    ceilings = np.stack([make_texture(seed * 1000 + 600 + k, "ceiling") for k in range(n_ceilings)])
    _LIB_CACHE[key] = dict(walls=walls, grounds=grounds, ceilings=ceilings)
    return _LIB_CACHE[key]


def check_texture_library(lib):
    """The ray caster and the texture-packing kernel address a texture as 256 x 256 x 3 (csrc/maze.hip: text_id * 256 * 256 *
    3, taps masked with & 255): a library of any other size would be read past its allocation.  Raises ValueError unless
    every library is [n, 256, 256, 3] with n >= 1.  (The reference takes any size through texture.shape,
    ray_caster_utils.py:124-140; its own folder is 256 x 256 throughout.)"""
    for key in ("walls", "grounds", "ceilings"):
        if key not in lib:
            raise ValueError("texture library lacks %r" % key)
        shp = tuple(lib[key].shape)
        if len(shp) != 4 or shp[0] < 1 or shp[1:] != (TEX_SIZE, TEX_SIZE, 3):
            raise ValueError("texture library %r has shape %s; the engine needs [n, %d, %d, 3] (load_texture_library(dir, "
                             "resize=True) resamples other image sizes)" % (key, shp, TEX_SIZE, TEX_SIZE))


def load_texture_library(texture_dir, resize=False):
    """The reference's way of building its libraries (mazeworld/envs/task_sampler.py:60-77): every file of `texture_dir`
    in SORTED name order whose name starts with `wall` / `ground` / `ceiling` is decoded to RGB and appended to that
    library as pygame.surfarray.array3d returns it — axes (W, H, 3), i.e. the decoded (H, W, 3) image transposed — as
    float32.  Other files are ignored.  -> dict(walls, grounds, ceilings) of float32 [n, W, H, 3]; pass it to
    `MazeWorldVecEnv(textures=...)` and its sizes to the sampler (`texture_counts(lib)`).  Needs Pillow.
    The engine's textures are 256 x 256 (as every image of the reference's own folder): an image of another size raises
    ValueError, or — resize=True — is resampled to 256 x 256 (bilinear; a deviation from the reference, which would sample
    the image at its own size)."""
    from PIL import Image
    libs = {"wall": [], "ground": [], "ceiling": []}
    for name in sorted(os.listdir(texture_dir)):
        for kind, dst in libs.items():
            if name.find(kind) == 0:
                with Image.open(os.path.join(texture_dir, name)) as im:
                    im = im.convert("RGB")
                    if im.size != (TEX_SIZE, TEX_SIZE):
                        if not resize:
                            raise ValueError("%s is %d x %d; the engine's textures are %d x %d (pass resize=True to resample)"
                                             % (name, im.size[0], im.size[1], TEX_SIZE, TEX_SIZE))
                        im = im.resize((TEX_SIZE, TEX_SIZE), Image.BILINEAR)
                    rgb = np.asarray(im)            # (H, W, 3)
                dst.append(np.ascontiguousarray(rgb.transpose(1, 0, 2)))
    out = {}
    for kind, key in (("wall", "walls"), ("ground", "grounds"), ("ceiling", "ceilings")):
        if not libs[kind]:
            raise ValueError("no %s* image in %s" % (kind, texture_dir))
        shapes = {a.shape for a in libs[kind]}
        if len(shapes) != 1:
            raise ValueError("%s* images differ in size: %s" % (kind, sorted(shapes)))
        out[key] = np.asarray(libs[kind], dtype="float32")
    check_texture_library(out)
    return out


def texture_counts(lib):
    """(n_walls, n_grounds, n_ceilings) of a library: the bounds MazeTaskSampler draws texture ids with"""
    return int(len(lib["walls"])), int(len(lib["grounds"])), int(len(lib["ceilings"]))
