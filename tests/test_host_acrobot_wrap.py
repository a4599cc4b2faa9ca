"""The O(log x) evaluation of gymnasium's acrobot wrap() used by the HIP kernel (xenoverse_amd/csrc/acrobot_wrap.h)
is bit-identical to the plain subtract-until-in-range loop: random arguments over many binades, binade edges, values
just above the points where a subtraction crosses into a finer binade, both signs."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("wrap") / "libwrap.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", so,
                           os.path.join(HERE, "native", "acrobot_wrap_host.c")])
    return C.CDLL(so)


def _both(lib, x, run_loop=True):
    x = np.ascontiguousarray(x, np.float64)
    fast, loop = np.empty_like(x), np.empty_like(x)
    stuck = np.zeros(len(x), np.int32)
    lib.wrap_both(x.ctypes.data_as(C.c_void_p), C.c_int(len(x)), fast.ctypes.data_as(C.c_void_p),
                  loop.ctypes.data_as(C.c_void_p) if run_loop else None, stuck.ctypes.data_as(C.c_void_p))
    return fast, loop, stuck


def test_bit_identical_to_the_loop(lib):
    rng = np.random.RandomState(0)
    xs = [rng.uniform(-40, 40, 20000)]
    for e in range(2, 22):                                    # |x| up to 4e6: the loop needs up to 7e5 iterations
        lo, hi = 2.0 ** e, 2.0 ** (e + 1)
        n = 4000 if e < 16 else 200
        v = rng.uniform(lo, hi, n)
        edge = np.array([lo, np.nextafter(lo, 0), np.nextafter(lo, np.inf), np.nextafter(hi, 0),
                         lo + 6.283185307179586, np.nextafter(lo + 6.283185307179586, np.inf), lo + 3.0, lo + 9.5])
        xs += [v, -v, edge, -edge]
    x = np.concatenate(xs)
    fast, loop, stuck = _both(lib, x)
    assert not stuck.any()
    assert np.array_equal(fast.view(np.int64), loop.view(np.int64))
    assert np.all(np.abs(fast) <= np.pi)


def test_large_and_non_finite_arguments(lib):
    x = np.array([1e12, -3e13, 2.0 ** 52 * 1.7, 2.0 ** 56, 1e300, np.inf, -np.inf, np.nan, 0.0, np.pi, -np.pi])
    fast, _, stuck = _both(lib, x, run_loop=False)     # the plain loop would need 1e11..1e15 iterations here
    assert np.all(np.abs(fast[:3]) <= np.pi) and not stuck[:3].any()
    assert stuck[3:7].all()                                    # x - 2 pi == x: the reference loop would never end
    assert np.isnan(fast[7]) and not stuck[7]
    assert np.array_equal(fast[8:], x[8:])
