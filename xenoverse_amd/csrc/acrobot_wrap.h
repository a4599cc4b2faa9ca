// acrobot_wrap.h — gymnasium's acrobot `wrap(x, -pi, pi)`:   while x > M: x = x - diff;  while x < m: x = x + diff
// evaluated EXACTLY but in O(log x) instead of O(x) iterations.
//
// Inside rk4 the intermediate velocities are unbounded (the clip to 4 pi / 9 pi is applied after the step), and for
// ill-conditioned tasks of the reference sampler the integrated angle leaves [-pi, pi] by 1e3..1e5 rad: the
// reference's loop then runs 1e2..1e4 dependent subtractions — for one lane, with the other 63 waiting (measured:
// 500 us per vector step instead of 8).  Each iteration ROUNDS, so fmod() is not the same function; but the rounding
// is regular: for x = m * u in the binade [2^e, 2^(e+1)), u = 2^(e-52), and diff = d * 2^-50 (d = 0x1921fb54442d18
// for 2 pi), one subtraction that stays in the binade gives exactly m' = m - c with c = (d >> j) + (frac > 1/2),
// j = e - 2, independent of m (frac = the low j bits of d over 2^j; the only tie, j = 4, lies below the threshold
// used here).  So k subtractions inside a binade are one integer multiply, and only the crossings between binades
// (and everything below 128) are done with the real subtraction.  tests/test_host_acrobot_wrap.py checks bit-equality
// with the plain loop over binade edges and random arguments.  Plain C, shared by the HIP kernel and the CPU test.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define XV_ACW_FN __host__ __device__ static inline
#else
#define XV_ACW_FN static inline
#endif

// floor(a / c) for a < 2^53, 0 < c < 2^53: both are exact doubles and their quotient is correctly rounded, so the
// truncated quotient is the floor or one above it (where a / c lies within half an ulp below an integer) — one multiply
// decides.  (A 64-bit integer division is ~200 instructions on the GPU; the lanes that wrap from 1e3..1e5 rad were
// what a whole acrobot step waited for: 51 us with it, 8 us with an inexact wrap.)
XV_ACW_FN uint64_t xv_acw_div(uint64_t a, uint64_t c) {
  uint64_t k = (uint64_t)((double)a / (double)c);
  if (k * c > a) --k;
  else if ((k + 1) * c <= a) ++k;
  return k;
}

// while (x > M) x = x - D, for D = 2 pi and M = pi as doubles.  *stuck is set when x is so large that x - D == x
// (the reference loop would never end) or x is +inf.
// Shape (it matters on the GPU, where one lane's loop trips are paid by its whole wave and a divergent trip costs
// ~500 cycles of exec-mask bookkeeping whatever its body: the step kernel ran 51 us with a trip per subtraction below
// 128 and two per binade above, 8 us without any wrap): above 128 ONE trip per binade — the subtractions that stay in
// the binade as one multiply, then the one that leaves it; below 128 the at most 20 subtractions as 20 selects.
XV_ACW_FN double xv_acrobot_wrap_down(double x, int* stuck) {
  const double M = 3.141592653589793, D = 6.283185307179586;
  const uint64_t d = 0x1921fb54442d18ull;   // D * 2^50
  while (x >= 128.0) {
    uint64_t bits;
    __builtin_memcpy(&bits, &x, 8);
    const int e = (int)(bits >> 52) - 1023;             // x in [2^e, 2^(e+1)), e >= 7 (sign bit is 0)
    const int j = e - 2;
    if (j >= 54) { *stuck = 1; return x; }              // the subtraction no longer changes x (or x is +inf)
    const uint64_t q = d >> j;
    const uint64_t c = q + (((d & ((1ull << j) - 1)) > (1ull << (j - 1))) ? 1u : 0u);   // >= 1 for j <= 53
    uint64_t m = (bits & ((1ull << 52) - 1)) | (1ull << 52);
    const uint64_t lo = (1ull << 52) + q + 1;           // from m >= lo the difference stays in this binade
    if (m >= lo) {
      const uint64_t k = xv_acw_div(m - lo, c) + 1;
      m -= k * c;
      bits = ((uint64_t)(e + 1023) << 52) | (m & ((1ull << 52) - 1));
      __builtin_memcpy(&x, &bits, 8);
    }
    x = x - D;                                          // crossing into the finer binade: the real subtraction
  }
  // x < 128: at most 20 subtractions are left (128 - 20 D < M)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int i = 0; i < 20; ++i) x = (x > M) ? x - D : x;
  return x;
}

XV_ACW_FN double xv_acrobot_wrap(double x, int* stuck) {
  const double M = 3.141592653589793;
  if (x > M) return xv_acrobot_wrap_down(x, stuck);
  if (x < -M) return -xv_acrobot_wrap_down(-x, stuck);   // IEEE rounding is symmetric: the x + diff loop mirrored
  return x;
}
