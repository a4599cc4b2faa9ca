"""ac_sincos of csrc/acrobot.hip, restated in NumPy float64 with the constants READ FROM THE SOURCE (same operations in
the same order; the library is built with -ffp-contract=off, so NumPy's unfused arithmetic is the kernel's), against
NumPy's sin / cos (glibc): max difference in ulp and share of identical results.  `python scripts/devtools/check_ac_sincos.py`;
tests/test_host_samplers.py calls check()."""
import os
import re

import numpy as np

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "xenoverse_amd", "csrc",
                   "acrobot.hip")


def constants():
    text = open(SRC).read()
    body = text[text.index("void ac_sincos("):text.index("double ac_cos(")]
    nums = [float(x) for x in re.findall(r"-?\d\.\d{10,}e[-+]\d\d", body)]
    assert len(nums) == 16, nums
    return nums


def ac_sincos(x):
    (invpio2, pio2_1, pio2_2, pio2_2t, S2, S3, S4, S5, S6, S1, C1, C2, C3, C4, C5, C6) = constants()
    fn = np.rint(x * invpio2)
    r = x - fn * pio2_1
    t = r
    w = fn * pio2_2
    r = t - w
    w = fn * pio2_2t - ((t - r) - w)
    y0 = r - w
    y1 = (r - y0) - w
    z = y0 * y0
    v = z * y0
    rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))
    s = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1)
    rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))))
    hz = 0.5 * z
    wv = 1.0 - hz
    c = wv + (((1.0 - wv) - hz) + (z * rc - y0 * y1))
    n = fn.astype(np.int64) & 3
    a = np.where(n & 1, c, s)
    b = np.where(n & 1, s, c)
    return np.where(n & 2, -a, a), np.where((n + 1) & 2, -b, b)


def check(n=400000, scales=(1.0, 4.0, 30.0, 1000.0)):
    rng = np.random.RandomState(0)
    worst, same = 0.0, 1.0
    for sc in scales:
        x = rng.uniform(-sc, sc, n)
        s, c = ac_sincos(x)
        for got, ref in ((s, np.sin(x)), (c, np.cos(x))):
            d = np.abs(got - ref) / np.spacing(np.abs(ref))
            worst = max(worst, float(d.max()))
            same = min(same, float((d == 0).mean()))
    return worst, same


if __name__ == "__main__":
    w, s = check(2000000)
    print("max difference %.2f ulp, identical in at least %.2f %% of the arguments" % (w, 100 * s))
