"""The bucket ("cut") lines of the AnyMDP step engine (xenoverse_amd/csrc/anymdp_cutline.h, the function the device
build kernel calls), compiled for the host: whatever a line answers equals numpy's searchsorted(cdf, u, 'right'), on the
reference sampler's own skewed rows, on dense synthetic bands, on rows with repeated / zero / sub-ulp entries and on rows
whose CDF ends below 1; and on the reference's rows almost nothing is left to the fence search."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("cut") / "libcut.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so,
                           os.path.join(HERE, "native", "cutline_host.cpp")])
    return C.CDLL(so)


def _check(lib, rows, nbk, K, n_rand=64):
    rows = np.ascontiguousarray(rows, np.float64)
    out = np.zeros(5, np.int64)
    mass = np.zeros(1, np.float64)
    lib.cutline_check(rows.ctypes.data_as(C.c_void_p), C.c_int(rows.shape[0]), C.c_int(rows.shape[1]), C.c_int(nbk), C.c_int(K),
                      C.c_int(n_rand), out.ctypes.data_as(C.c_void_p), mass.ctypes.data_as(C.c_void_p))
    return dict(wrong=int(out[0]), answered=int(out[1]), fenced=int(out[2]), dirty_lines=int(out[3]), faults=int(out[4]),
                p_fallback=float(mass[0]) / rows.shape[0])


def _cdf(T):
    c = np.cumsum(T, -1)
    return c / c[..., -1:]


def test_reference_rows_are_answered_exactly_and_almost_always(lib):
    d = np.load(os.path.join(HERE, "golden", "anymdp_64x8_seed1.npz"), allow_pickle=True)
    T = d["transition"]
    live = np.setdiff1d(np.arange(T.shape[0]), d["s_e"])
    rows = _cdf(T[live].reshape(-1, T.shape[-1]))
    for nbk in (16, 32, 64):
        for K in (6, 7):
            r = _check(lib, rows, nbk, K)
            # (the probes sit ON the row's CDF entries, so many of them land in the lumped runs by construction:
            # what matters is the probability mass of those runs, checked below)
            assert r["wrong"] == 0 and r["faults"] == 0 and r["answered"] > 0, (nbk, K, r)
    # consecutive entries (rounds 2-3) left 2.7e-2 of the draws of this task to the fence search
    assert _check(lib, rows, 16, 7)["p_fallback"] < 5e-7
    assert _check(lib, rows, 16, 6)["p_fallback"] < 5e-6
    assert _check(lib, rows, 32, 7)["p_fallback"] < 1e-7


def test_dense_bands_need_no_fence_search(lib):
    rng = np.random.RandomState(1)
    S = 64
    T = np.zeros((512, S))
    for r in range(512):
        s = rng.randint(0, S)
        lo = rng.randint(max(0, s - 33), s + 1)
        hi = rng.randint(s + 2, min(S, s + 17) + 1) if s + 2 <= min(S, s + 17) else S
        T[r, lo:hi] = np.clip(np.abs(rng.randn(hi - lo)), 0.1, 1.0)
    r = _check(lib, _cdf(T), 16, 7)
    assert r["wrong"] == 0 and r["faults"] == 0 and r["fenced"] == 0 and r["dirty_lines"] == 0, r


@pytest.mark.parametrize("S", [2, 7, 8, 15, 64, 113, 256, 300, 512])
def test_awkward_rows(lib, S):
    rng = np.random.RandomState(S)
    rows = []
    for _ in range(48):
        kind = rng.randint(6)
        if kind == 0:        # uniform: S / NBK states per bucket, far more than a line holds
            p = np.ones(S)
        elif kind == 1:      # softmax over a huge range: most entries below one ulp of their neighbours
            p = np.exp(-rng.uniform(0, 700, S))
        elif kind == 2:      # many exact zeros
            p = rng.rand(S) * (rng.rand(S) < 0.2)
            p[rng.randint(S)] += 1.0
        elif kind == 3:      # one state only
            p = np.zeros(S)
            p[rng.randint(S)] = 1.0
        elif kind == 4:      # powers of two: cuts on bucket edges
            p = 2.0 ** -rng.randint(1, 12, S).astype(np.float64)
        else:
            p = rng.rand(S) ** 8
        rows.append(_cdf(p))
    rows = np.stack(rows)
    rows[1] = np.minimum(rows[1], 0.75)          # a caller-supplied row whose CDF ends below 1: s' clamps to S - 1
    rows[2] = 1.0                                  # the all-ones row of a terminal state
    for nbk in (16, 64):
        for K in (6, 7):
            r = _check(lib, rows, nbk, K, n_rand=32)
            assert r["wrong"] == 0 and r["faults"] == 0, (S, nbk, K, r)


def test_host_sampled_tasks(lib):
    from xenoverse_amd.anymdp.task_sampler import AnyMDPTaskSampler
    tot = []
    for seed in (2, 3):
        t = AnyMDPTaskSampler(64, 8, seed=seed)
        T = np.asarray(t["transition"])
        live = np.setdiff1d(np.arange(64), np.asarray(t["s_e"]))
        r = _check(lib, _cdf(T[live].reshape(-1, 64)), 16, 7)
        assert r["wrong"] == 0 and r["faults"] == 0
        tot.append(r["p_fallback"])
    assert max(tot) < 1e-6, tot


@pytest.mark.parametrize("n_obs,density", [(64, 0.08), (200, 0.03), (256, 0.5), (15, 1.0)])
def test_observation_rows_with_fourteen_cuts(lib, n_obs, density):
    """observation bucket lines list 14 cuts (csrc/anymdp.hip: anymdp_build_obs_cutlines_kernel): the reference's observation
    rows are sparse (scipy.sparse.random rows + a 1 in empty rows, task_sampler.py:78-87) — the zero-probability symbols between
    the few live ones no longer use up a line: nothing is left to the per-lane search"""
    rng = np.random.RandomState(n_obs)
    rows = []
    for _ in range(256):
        p = rng.rand(n_obs) * (rng.rand(n_obs) < density)
        if p.sum() == 0:
            p[rng.randint(n_obs)] = 1.0
        rows.append(_cdf(p))
    r = _check(lib, np.stack(rows), 16, 14)
    assert r["wrong"] == 0 and r["faults"] == 0
    if density <= 0.1:
        assert r["dirty_lines"] == 0 and r["fenced"] == 0, r       # <= 14 live symbols meet any bucket
    assert r["p_fallback"] < (0.02 if density > 0.3 else 1e-12), r
