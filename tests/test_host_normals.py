"""The 16 + 16-bit Box-Muller pair of the LinDS process noise (csrc/philox.h: xv_box_muller16; the reference draws
numpy.random.normal, linds_env.py:147-150).  One Philox word gives a pair: radius from the high 16 bits (u1 = (hi + 1) /
65536 in (0, 1]), angle from the low 16 (u2 = lo / 65536 revolutions).  Over all 2^32 words the distribution factorises into
65,536 radii x 65,536 angles, so its moments are computed EXACTLY here — no sampling, nothing shared with the oracle's
restatement — and held against N(0, 1): what the truncation (|z| <= 4.71) and the upper-edge radius (variance a hair low)
amount to.  The device generator itself is sampled in tests/test_gpu_linds.py against these numbers."""
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def grid_moments():
    k = np.arange(1, 65537, dtype=np.float64)
    r2 = -2.0 * np.log(k / 65536.0)                      # radius squared, one value per high half-word
    th = 2.0 * np.pi * np.arange(65536, dtype=np.float64) / 65536.0
    c = np.cos(th)
    return dict(mean=float(np.sqrt(r2).mean() * c.mean()), var=float(r2.mean() * (c ** 2).mean()),
                m4=float((r2 ** 2).mean() * (c ** 4).mean()), zmax=float(np.sqrt(r2.max())), r2=r2, c=c)


def test_the_source_still_says_what_this_test_restates():
    src = open(os.path.join(HERE, "..", "xenoverse_amd", "csrc", "philox.h")).read()
    body = src[src.index("void xv_box_muller16("):]
    body = body[:body.index("\n}")]
    assert re.search(r"\(w >> 16\) \+ 1\.0f\) \* \(1\.0f / 65536\.0f\)", body)
    assert re.search(r"\(w & 0xFFFFu\) \* \(1\.0f / 65536\.0f\)", body)
    assert "-1.3862943611198906f" in body and "__builtin_amdgcn_logf" in body       # -2 ln 2 * log2(u1)
    assert "__builtin_amdgcn_cosf(u2)" in body and "__builtin_amdgcn_sinf(u2)" in body   # argument in revolutions


def test_exact_moments_of_the_grid_against_the_standard_normal():
    m = grid_moments()
    assert abs(m["mean"]) < 1e-12                         # the angles are symmetric
    assert 0.9997 < m["var"] < 1.0                        # u1 at the bucket's upper edge: variance 1 - 9.9e-5
    assert abs(m["var"] - (1.0 - 9.86e-5)) < 1e-6
    kurt = m["m4"] / m["var"] ** 2
    assert abs(kurt - 3.0) < 5e-3                         # the truncated tail carries 1e-4 of the fourth moment
    assert 4.70 < m["zmax"] < 4.72                        # sqrt(32 ln 2)


def test_tail_mass_of_the_grid():
    """P(|z| > t) over the whole grid, by a (radius x angle) count on a sub-grid of the angles (every 16th: the radii are
    exact), against the normal's: equal within 2 % at 2 and 3 sigma, present up to 4.5, nothing beyond 4.71"""
    from math import erfc, sqrt
    m = grid_moments()
    r = np.sqrt(m["r2"])
    c = np.abs(m["c"][::16])
    for t, tol in ((2.0, 0.02), (3.0, 0.02), (4.0, 0.10)):
        # |r c| > t  <=>  r > t / |c|
        thr = np.where(c > 0, t / np.maximum(c, 1e-300), np.inf)
        rs = np.sort(r)
        p = float(np.mean((len(rs) - np.searchsorted(rs, thr, "right")) / len(rs)))
        assert abs(p - erfc(t / sqrt(2.0))) <= tol * erfc(t / sqrt(2.0)), (t, p)
    assert r.max() < 4.72
