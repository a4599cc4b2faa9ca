// anymdp_vi_host.hip — host-side value iteration in the reference's order of operations (no device code).
//
// The reference's AnyMDP task sampler repairs and accepts candidate tasks with `update_value_matrix`
// (xenoverse/anymdp/solver.py:57-82): damped Gauss-Seidel sweeps over (s, a), the value matrix updated in place,
// fp64.  The repaired rewards are functions of those values, so a sampler that wants the SAME task for the same
// seed has to reproduce the sweep arithmetic, not just its fixed point.  This is that arithmetic, arranged for
// speed where the arrangement cannot change a bit:
//   * each (s, a) row is reduced over its non-zero span only — a skipped term is t * (...) = +-0 and adding it to
//     the running sum leaves the sum unchanged (the sum starts at +0 and is never -0);
//   * max_a / mean_a of a value-matrix row are cached per state and refreshed for state s after every update of
//     vm[s, .] — what the reference recomputes inside its innermost loop;
//   * numpy.mean is NumPy's pairwise summation (plain loop below 8 elements, eight partial sums up to 128, halves
//     rounded to a multiple of 8 above), divided by the count.  That is what the reference executes when its @njit
//     functions run as plain Python — how the golden fixtures were produced (numba is absent from this image) — and the
//     default here.  A numba-compiled reference reduces np.mean as one sequential loop instead; that order is offered
//     as summation mode 1 (xv_anymdp_value_iteration_set_summation) and is NOT pinned by any fixture.  The two differ
//     in the last bits of means over >= 8 values (the uniform-policy values for na >= 8, the rms stopping test), which
//     reach the task only through threshold comparisons (gap >= 2, diff > 1e-4).
// Compiled with -ffp-contract=off like every other file: a*b+c stays two roundings, as in the interpreter.
#include <cmath>
#include <cstring>
#include <vector>

#include "xv_common.h"

#include <atomic>

namespace {

std::atomic<int> g_sequential_sum{0};

double np_sum(const double* a, long n) {
  if (n < 8 || g_sequential_sum.load(std::memory_order_relaxed)) {
    double res = 0.0;
    for (long i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    long i = 8;
    for (; i < n - (n % 8); i += 8) {
      r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
      r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  long half = n / 2;
  half -= half % 8;
  return np_sum(a, half) + np_sum(a + half, n - half);
}

inline double row_stat(const double* row, int na, bool greedy) {
  if (greedy) {
    double m = row[0];
    for (int k = 1; k < na; ++k) m = row[k] > m ? row[k] : m;
    return m;
  }
  return (0.0 + np_sum(row, na)) / (double)na;
}

}  // namespace

extern "C" int xv_anymdp_value_iteration_set_summation(int mode) {
  XV_CHECK_ARG(mode == 0 || mode == 1);
  g_sequential_sum.store(mode, std::memory_order_relaxed);
  return XV_OK;
}

extern "C" int xv_anymdp_value_iteration_gs(const double* t_mat, const double* r_mat, int ns, int na, double gamma,
                                            int is_greedy, double* vm, int32_t* sweeps_out) {
  XV_CHECK_ARG(t_mat && r_mat && vm);
  XV_CHECK_ARG(ns >= 1 && na >= 1 && ns <= 4096 && na <= 4096);
  const bool greedy = is_greedy != 0;
  const size_t n = (size_t)ns * na;
  std::vector<int> lo(n), hi(n);
  for (size_t row = 0; row < n; ++row) {
    const double* t = t_mat + row * ns;
    int a = 0, b = ns;
    while (a < ns && t[a] == 0.0) ++a;
    while (b > a && t[b - 1] == 0.0) --b;
    lo[row] = a; hi[row] = b;
  }
  std::vector<double> stat(ns), old(n), sq(n);
  for (int s = 0; s < ns; ++s) stat[s] = row_stat(vm + (size_t)s * na, na, greedy);
  double diff = 1.0, alpha = 1.0;
  int sweeps = 0;
  while (diff > 1.0e-4) {
    ++sweeps;
    std::memcpy(old.data(), vm, sizeof(double) * n);
    for (int s = 0; s < ns; ++s) {
      double* vrow = vm + (size_t)s * na;
      for (int a = 0; a < na; ++a) {
        const size_t row = (size_t)s * na + a;
        const double* t = t_mat + row * ns;
        const double* r = r_mat + row * ns;
        double exp_q = 0.0;
        for (int sn = lo[row]; sn < hi[row]; ++sn) {
          if (t[sn] == 0.0) continue;
          exp_q += t[sn] * (gamma * stat[sn] + r[sn]);
        }
        vrow[a] += alpha * (exp_q - vrow[a]);
        stat[s] = row_stat(vrow, na, greedy);
      }
    }
    for (size_t i = 0; i < n; ++i) {
      const double d = old[i] - vm[i];
      sq[i] = d * d;
    }
    diff = std::sqrt((0.0 + np_sum(sq.data(), (long)n)) / (double)n);
    if (!(diff == diff)) {   // NaN: the reference would spin forever on `nan > 1e-4` being False -> it stops too
      break;
    }
    alpha = 0.80 * alpha > 0.50 ? 0.80 * alpha : 0.50;
  }
  if (sweeps_out) *sweeps_out = sweeps;
  return XV_OK;
}
