/* xeno_oracle_sampler.c — CPU restatement of the AnyMDP task-sampler arithmetic.  TEST INFRASTRUCTURE, NOT PRODUCT
 * (see oracle/__init__.py).  Reference: xenoverse/anymdp/solver.py (update_value_matrix :57-82,
 * get_opt_trajectory_dist :84-103, check_valuefunction :105-148) and task_sampler_utils.py (:65-256).
 *
 * Part 1: the reference's value iteration in ITS order of operations — damped Gauss-Seidel sweeps over (s, a) with
 * the running value matrix updated in place, fp64, no FMA contraction (-ffp-contract=off) — so that a task sampled
 * with the reference's random stream comes out bit for bit (the repair loop of sample_mdp feeds the values back into
 * the reward tensor).  Third-party arithmetic on that path: NumPy's pairwise summation behind numpy.mean (NumPy
 * >= 1.9, numpy/_core/src/umath/loops_utils.h.src `@TYPE@_pairwise_sum`: < 8 elements a plain loop, <= 128 elements
 * eight running partial sums combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus the tail, larger blocks split in
 * halves rounded to a multiple of 8); restated here and pinned by tests/test_oracle_sampler.py against numpy itself
 * and against value matrices computed by the reference's own update_value_matrix (tests/golden/anymdp_vi_*.npz).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "xeno_oracle.h"

double xo_np_pairwise_sum(const double* a, int64_t n) {
  if (n < 8) {
    double res = 0.0;      /* numpy: res = 0.; for i: res += a[i]  (the leading "+0" is exact except for -0.0) */
    for (int64_t i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int64_t i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int64_t n2 = n / 2;
  n2 -= n2 % 8;
  return xo_np_pairwise_sum(a, n2) + xo_np_pairwise_sum(a + n2, n - n2);
}

/* numpy.mean of a contiguous float64 vector: add.reduce (identity 0 + pairwise sum) / n */
static double np_mean(const double* a, int64_t n) { return (0.0 + xo_np_pairwise_sum(a, n)) / (double)n; }

/* solver.py:57-82 with max_iteration = -1 (the only way the sampler calls it): returns the number of sweeps.
 * vm double[ns][na] is read as the starting point and overwritten with the result. */
int xo_update_value_matrix(const double* t_mat, const double* r_mat, int ns, int na, double gamma, double* vm,
                           int is_greedy) {
  const size_t n = (size_t)ns * na;
  double* old = (double*)malloc(sizeof(double) * n);
  double* sq = (double*)malloc(sizeof(double) * n);
  double diff = 1.0, alpha = 1.0;
  int iteration = 0;
  while (diff > 1.0e-4) {
    ++iteration;
    memcpy(old, vm, sizeof(double) * n);
    for (int s = 0; s < ns; ++s)
      for (int a = 0; a < na; ++a) {
        double exp_q = 0.0;
        for (int sn = 0; sn < ns; ++sn) {
          const double* row = vm + (size_t)sn * na;       /* the value matrix AS IT IS NOW (Gauss-Seidel) */
          double v;
          if (is_greedy) {
            v = row[0];
            for (int k = 1; k < na; ++k) v = row[k] > v ? row[k] : v;
          } else {
            v = np_mean(row, na);
          }
          const size_t idx = ((size_t)s * na + a) * ns + sn;
          exp_q += t_mat[idx] * (gamma * v + r_mat[idx]);
        }
        vm[(size_t)s * na + a] += alpha * (exp_q - vm[(size_t)s * na + a]);
      }
    for (size_t i = 0; i < n; ++i) {
      double d = old[i] - vm[i];
      sq[i] = d * d;
    }
    diff = sqrt(np_mean(sq, (int64_t)n));
    alpha = 0.80 * alpha > 0.50 ? 0.80 * alpha : 0.50;
  }
  free(old);
  free(sq);
  return iteration;
}
