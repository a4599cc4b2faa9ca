"""The sampler arithmetic of the oracle (oracle/xeno_oracle_sampler.c) pinned to the reference: value matrices the
reference's own update_value_matrix produced (tests/golden/sampler_vi_ref.npz, oracle/gen_golden.py anymdp_vi) must come
out bit for bit, NumPy's pairwise summation must equal numpy itself, and the product's host value iteration
(libxeno_hip.so xv_anymdp_value_iteration_gs — host code, no GPU) must equal the oracle's on random MDPs."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from util import GOLD


def test_pairwise_sum_equals_numpy():
    rng = np.random.RandomState(0)
    for n in list(range(1, 40)) + [63, 64, 65, 127, 128, 129, 255, 256, 257, 512, 1000, 1024, 4096, 5000]:
        for _ in range(20):
            a = rng.standard_normal(n) * 10 ** rng.uniform(-3, 3)
            assert oracle.np_pairwise_sum(a) == np.add.reduce(a)
            assert (0.0 + oracle.np_pairwise_sum(a)) / n == a.mean()


def test_update_value_matrix_equals_the_reference_bit_for_bit():
    g = np.load(os.path.join(GOLD, "sampler_vi_ref.npz"))
    for seed in (0, 2):
        T, R = g["T%d" % seed], g["R%d" % seed]
        ns, na, _ = T.shape
        for gi, gamma in enumerate(g["gamma%d" % seed]):
            for greedy in (1, 0):
                vm, sweeps = oracle.update_value_matrix(T, R, float(gamma), np.zeros((ns, na)), bool(greedy))
                assert np.array_equal(vm, g["vm_s%d_g%d_%d" % (seed, gi, greedy)]), (seed, gi, greedy)
                assert sweeps > 10
        vm = np.zeros((ns, na))
        for k in range(3):      # warm starts with shifted terminal rewards (the repair loop's calling pattern)
            bonus = g["chain_bonus_s%d_%d" % (seed, k)]
            vm, _ = oracle.update_value_matrix(T, R + bonus[None, None, :], 0.99, vm, True)
            assert np.array_equal(vm, g["chain_s%d_%d" % (seed, k)]), (seed, k)


def _random_mdp(rng, ns, na, band):
    T = np.zeros((ns, na, ns))
    for s in range(ns):
        if rng.random_sample() < 0.15 and s > 2:
            continue                       # terminal state: all-zero rows
        for a in range(na):
            lo = rng.randint(0, max(1, ns - band))
            w = np.clip(rng.normal(size=band), 0.0, None) + (rng.random_sample(band) < 0.3) * 0.0
            w[rng.randint(band)] += 0.1
            T[s, a, lo:lo + band] = w / w.sum()
    R = rng.normal(size=(ns, na, ns)) * 2.0
    return T, R


@pytest.mark.parametrize("ns,na", [(8, 2), (16, 4), (33, 5), (64, 8), (40, 9), (24, 17)])
def test_product_host_value_iteration_equals_the_oracle(ns, na):
    from xenoverse_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(ns * 100 + na)
    T, R = _random_mdp(rng, ns, na, band=max(3, ns // 2))
    for greedy in (1, 0):
        for gamma in (0.99, 2.0 ** (-1.0 / ns)):
            start = rng.normal(size=(ns, na)) if greedy else np.zeros((ns, na))
            ref, sweeps = oracle.update_value_matrix(T, R, gamma, start, bool(greedy))
            vm = np.array(start, copy=True)
            it = C.c_int32(0)
            _lib.check(lib.xv_anymdp_value_iteration_gs(T.ctypes.data, R.ctypes.data, ns, na, gamma, greedy,
                                                        vm.ctypes.data, C.addressof(it)))
            assert np.array_equal(vm, ref) and it.value == sweeps, (ns, na, greedy, gamma)


def test_device_algorithm_population_matches_the_reference_stream_population():
    """The device sampler's algorithm (restated in the oracle: counter-based draws, synchronous value iteration) and the
    reference's own stream of candidates (the seeded host sampler, pinned bit for bit to the reference by the golden
    tasks) are two samplers of ONE distribution: over ~700 candidates each, the share of unrepairable and of accepted
    candidates, the pitfall count, goal share, band width, non-zeros per row, largest transition probability and the
    reward spread agree within sampling error (4 sigma of the difference of two independent means)."""
    from xenoverse_amd.anymdp import task_sampler as ts
    ns, na = 16, 4

    def shape_stats(T, s_e, lo, hi):
        live = [s for s in range(ns) if s not in set(s_e)]
        return dict(bw=np.mean([hi[s] - lo[s] for s in live]), nnz=np.mean([(T[s] > 0).sum(-1).mean() for s in live]),
                    tmax=np.mean([T[s].max(-1).mean() for s in live]))
    ref = []
    for seed in range(5000, 5220):
        rng = np.random.RandomState(seed)
        task, real = ts._task_head(rng, ns, na, None)
        for _ in range(60):
            g = ts._ReferenceStream(rng, real, na)
            res = g.candidate()
            rec = dict(none=float(res is None), pits=float(len(g.s_e)), goal=float(g.goal_terminates), n_s0=float(len(g.s_0)),
                       **shape_stats(g.T, g.s_e, g.lo, g.hi))
            rec["acc"] = 0.0
            if res is not None:
                task.update(res)
                rec["acc"] = float(ts.reference_acceptance(task))
                rec["rstd"] = float(res["reward"].std())
                rec["nzn"] = float((res["reward_noise"] > 0).mean())
            ref.append(rec)
            if rec["acc"]:
                break
    dev = []
    for c in range(len(ref)):
        d = oracle.anymdp_sample_candidate(4242, c, ns, na)
        rec = dict(none=float(d["status"] == 1), acc=float(d["status"] == 0), pits=float(len(d["s_e"])), goal=float(d["goal"]),
                   n_s0=float(len(d["s_0"])), **shape_stats(d["transition"], d["s_e"], d["band_lo"], d["band_hi"]))
        if d["status"] != 1:
            rec["rstd"] = float(d["reward"].std())
            rec["nzn"] = float((d["reward_noise"] > 0).mean())
        dev.append(rec)
    assert len(ref) > 500
    for k in ("none", "acc", "pits", "goal", "n_s0", "bw", "nnz", "tmax", "rstd", "nzn"):
        a = np.array([r[k] for r in ref if k in r])
        b = np.array([r[k] for r in dev if k in r])
        se = np.sqrt(a.var() / len(a) + b.var() / len(b))
        assert abs(a.mean() - b.mean()) <= 4 * se + 1e-12, (k, a.mean(), b.mean(), se)


def test_observation_model_sampler_follows_the_reference_rules():
    """xo_anymdp_sample_observation_model (the restatement of the device's observation-model sampler) against the
    reference's construction (task_sampler.py:78-87: scipy.sparse.random + empty-row fix + normalisation) run on NumPy's
    stream: exactly round(density * S * n_obs) non-zeros per matrix before the fix, rows that sum to 1, the same share of
    fixed rows and the same row-occupancy distribution within 4 sigma"""
    import scipy.sparse as sp
    S, n_obs, d_obs, n_task = 64, 64, 2, 40
    cdf = oracle.anymdp_sample_observation_model(7, 0, n_task, S, n_obs, d_obs)
    pmf = np.diff(np.concatenate([np.zeros(cdf.shape[:-1] + (1,)), cdf], -1), axis=-1)
    assert np.all(pmf >= 0) and np.array_equal(cdf[..., -1], np.ones(cdf.shape[:-1]))
    dens = min(0.20, 4 / n_obs)
    k = int(round(dens * S * n_obs))
    nnz = (pmf > 0).sum(-1)                                    # [n_task, d_obs, S]
    fixed = nnz.sum(-1) - k                                     # cells added by the empty-row fix, per matrix
    assert np.all(fixed >= 0)
    single_one = (nnz == 1) & (pmf.max(-1) == 1.0)
    assert np.all(fixed <= single_one.sum(-1))
    # the reference's construction on its own stream
    rng = np.random.RandomState(3)
    ref_nnz = []
    for _ in range(n_task * d_obs):
        m = sp.random(S, n_obs, density=dens, format="csr", random_state=rng).toarray()
        ref_nnz.append((m > 0).sum(-1))
    ref_nnz = np.asarray(ref_nnz)
    p_empty_ref = (ref_nnz == 0).mean()
    p_empty = fixed.sum() / (n_task * d_obs * S)
    n = n_task * d_obs * S
    assert abs(p_empty - p_empty_ref) < 4 * np.sqrt(2 * p_empty_ref * (1 - p_empty_ref) / n) + 1e-3
    for c in range(1, 8):                                        # occupancy histogram of the rows that were not fixed
        a = ((nnz == c) & ~single_one).mean() + (single_one.mean() - p_empty if c == 1 else 0.0)
        b = (ref_nnz == c).mean()
        assert abs(a - b) < 4 * np.sqrt(2 * max(b, 1e-3) / n) + 2e-3, (c, a, b)
    # values: uniform on [0, 1) before normalisation -> within a row of c >= 2 entries the largest share averages c / (c + 1)
    # only in expectation of order statistics; cheaper invariant: a different seed gives different matrices, the same seed
    # and task index the same ones whatever the batch split
    again = oracle.anymdp_sample_observation_model(7, 10, 5, S, n_obs, d_obs)
    assert np.array_equal(again, cdf[10:15])
    other = oracle.anymdp_sample_observation_model(8, 0, 2, S, n_obs, d_obs)
    assert not np.array_equal(other, cdf[:2])
    tiny = oracle.anymdp_sample_observation_model(1, 0, 3, 5, 7, 1, density=0.0)      # k = 0: every row is a fixed row
    assert np.all(np.isin(tiny, (0.0, 1.0))) and np.all(tiny[..., -1] == 1.0)
