"""Test-tooling stand-in for `numba` (absent from this image): `njit`/`jit` are identity decorators.

Used ONLY by oracle/gen_golden.py, in the build container, to import the Python reference and
capture golden vectors.  Contains no reference code.  Never imported by the product.
"""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


jit = njit
