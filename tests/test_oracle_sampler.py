"""The sampler arithmetic of the oracle (oracle/xeno_oracle_sampler.c) pinned to the reference: value matrices the
reference's own update_value_matrix produced (tests/golden/anymdp_vi_ref.npz, oracle/gen_golden.py anymdp_vi) must come
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
    g = np.load(os.path.join(GOLD, "anymdp_vi_ref.npz"))
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
