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
 * and against value matrices computed by the reference's own update_value_matrix (tests/golden/sampler_vi_ref.npz).
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

/* =====================================================================================================================
 * Part 2: CPU restatement of the DEVICE task sampler (xenoverse_amd/csrc/anymdp_sampler.hip, xv_anymdp_sample_tasks).
 *
 * The device sampler runs the reference's generative model (task_sampler_utils.py:65-256) and acceptance test
 * (solver.py:84-148) for thousands of candidate tasks at once, one workgroup per candidate.  It cannot consume NumPy's
 * sequential MT19937 stream, so every random quantity is a counter-based draw instead:
 *     Philox4x32-10( counter = {candidate lo, candidate hi, index, purpose}, key = seed )
 * i.e. a pure function of (seed, candidate, purpose, index).  The distributions are the reference's (the same uniform /
 * normal / exponential / integer draws, the same clipping, retry and widening rules); the stream is not, so a seed names
 * a different task than in the reference.  What this file pins is the device implementation: same draws, same formulas
 * -> integers equal, floats to ~1e-12 (libm vs device libm in log / sincos / exp).  The equivalence with the reference's
 * distribution is tested separately against a reference-sampled population (tests/golden/sampler_refpop_16x4.npz).
 * Value iteration here is the synchronous (Jacobi) sweep, Q <- ER + gamma T V(Q), iterated to rms update <= 1e-4: the
 * same fixed point and the same stopping rule as the reference's damped Gauss-Seidel, in a form that parallelises.
 * ===================================================================================================================*/
enum { XS_HEAD = 0, XS_PERM, XS_S0, XS_PIT, XS_PITS, XS_BAND, XS_BANDW, XS_ACT, XS_ACTW, XS_POT, XS_POS, XS_POSN, XS_POSU,
       XS_SA, XS_SAM, XS_STEP, XS_REPAIR };
#define XS_MAX_SWEEPS 20000
#define XS_EPS 1e-10

static void xs_draw(uint64_t seed, uint64_t cand, uint32_t purpose, uint32_t idx, uint32_t w[4]) {
  uint32_t ctr[4] = {(uint32_t)cand, (uint32_t)(cand >> 32), idx, purpose};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  xo_philox4x32_10(ctr, key, w);
}
static double xs_u32(uint32_t w) { return (double)w * (1.0 / 4294967296.0); }
static void xs_normal2(uint32_t wa, uint32_t wb, double* z0, double* z1) {
  const double u1 = ((double)wa + 1.0) * (1.0 / 4294967296.0), u2 = xs_u32(wb);
  const double r = sqrt(-2.0 * log(u1));
  *z0 = r * cos(6.283185307179586476925286766559 * u2);
  *z1 = r * sin(6.283185307179586476925286766559 * u2);
}
static void xs_normal4(uint64_t seed, uint64_t cand, uint32_t purpose, uint32_t idx, double z[4]) {
  uint32_t w[4];
  xs_draw(seed, cand, purpose, idx, w);
  xs_normal2(w[0], w[1], &z[0], &z[1]);
  xs_normal2(w[2], w[3], &z[2], &z[3]);
}
static double xs_clip(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* synchronous value iteration: q[ns*na] in/out; returns sweeps */
static int xs_vi(const double* T, const double* er, int ns, int na, double gamma, int greedy, double* q) {
  double* v = (double*)malloc(sizeof(double) * ns);
  double* qn = (double*)malloc(sizeof(double) * ns * na);
  int it = 0;
  for (; it < XS_MAX_SWEEPS;) {
    for (int j = 0; j < ns; ++j) {
      const double* row = q + (size_t)j * na;
      double x;
      if (greedy) {
        x = row[0];
        for (int a = 1; a < na; ++a) x = row[a] > x ? row[a] : x;
      } else {
        x = 0.0;
        for (int a = 0; a < na; ++a) x += row[a];
        x /= (double)na;
      }
      v[j] = x;
    }
    double d2 = 0.0;
    for (int sa = 0; sa < ns * na; ++sa) {
      const double* t = T + (size_t)sa * ns;
      double acc = 0.0;
      for (int j = 0; j < ns; ++j) acc = fma(t[j], v[j], acc);
      qn[sa] = fma(gamma, acc, er[sa]);
      d2 += (qn[sa] - q[sa]) * (qn[sa] - q[sa]);
    }
    memcpy(q, qn, sizeof(double) * ns * na);
    ++it;
    if (sqrt(d2 / (double)(ns * na)) <= 1.0e-4) break;
  }
  free(v);
  free(qn);
  return it;
}

/* One candidate.  Outputs (all caller-allocated): T, R, noise double[ns][na][ns]; info (see xeno_oracle.h).
 * Returns the status: 0 accepted, 1 terminal rewards could not be repaired, 2 value gap below 2, 3 occupancy too
 * concentrated, 4 value iteration did not converge. */
int xo_anymdp_sample_candidate(uint64_t seed, uint64_t cand, int ns, int na, double* T, double* R, double* noise,
                               xo_cand_info* info) {
  uint32_t w[4];
  double z[4];
  memset(info, 0, sizeof(*info));
  /* ---- head: max_steps (task_sampler.py:30-33), state_mapping (:44) ---- */
  const double lower = 4.0 * ns > 100 ? 4.0 * ns : 100;
  double upper = 8.0 * ns < 500 ? 8.0 * ns : 500;
  if (upper < lower + 1) upper = lower + 1;
  xs_draw(seed, cand, XS_HEAD, 0, w);
  const double max_steps = lower + xo_u53(w[0], w[1]) * (upper - lower);
  info->max_steps = max_steps;
  for (int i = 0; i < ns; ++i) info->state_map[i] = i;
  for (int i = ns - 1; i >= 1; --i) {
    xs_draw(seed, cand, XS_PERM, (uint32_t)(i >> 2), w);
    const int j = (int)(w[i & 3] % (uint32_t)(i + 1));
    const int32_t tmp = info->state_map[i]; info->state_map[i] = info->state_map[j]; info->state_map[j] = tmp;
  }
  /* ---- start states (task_sampler_utils.py:70-79) ---- */
  double w0[3] = {1.0, 0.0, 0.0};
  for (uint32_t r = 0; r < 16; ++r) {
    xs_normal4(seed, cand, XS_S0, r, z);
    double sum = 0.0;
    for (int k = 0; k < 3; ++k) sum += z[k] > 0.0 ? z[k] : 0.0;
    if (sum >= XS_EPS) {
      for (int k = 0; k < 3; ++k) w0[k] = z[k] > 0.0 ? z[k] : 0.0;
      break;
    }
  }
  int n_s0 = 0;
  double s0sum = 0.0;
  for (int k = 0; k < 3; ++k)
    if (w0[k] > XS_EPS) { info->s0[n_s0] = k; info->s0_prob[n_s0] = w0[k]; s0sum += w0[k]; ++n_s0; }
  for (int k = 0; k < n_s0; ++k) info->s0_prob[k] /= s0sum;
  info->n_s0 = n_s0;
  /* ---- terminal states (:81-93) ---- */
  xs_draw(seed, cand, XS_PIT, 0, w);
  double p_pit = -0.20 + 0.60 * xo_u53(w[0], w[1]);
  if (p_pit < 0.0) p_pit = 0.0;
  const int goal = xs_u32(w[2]) < 0.3;
  uint8_t* pit = info->s_e;
  for (uint32_t r = 0; r < 64; ++r) {
    int cnt = 0;
    for (int j = 0; j < ns; ++j) {
      if ((j & 3) == 0) xs_draw(seed, cand, XS_PITS, r * 64u + (uint32_t)(j >> 2), w);
      pit[j] = xs_u32(w[j & 3]) < p_pit;
      cnt += pit[j];
    }
    if ((double)cnt < (double)ns * p_pit + 1.0) break;
  }
  for (int k = 0; k < n_s0; ++k) pit[info->s0[k]] = 0;
  pit[ns - 1] = (uint8_t)goal;
  info->goal = goal;
  int n_se = 0;
  for (int j = 0; j < ns; ++j) n_se += pit[j];
  /* ---- banded kernel and its split over the actions (:95-175) ---- */
  const int fwd_max = ns / 4 + 1 > 2 ? ns / 4 + 1 : 2, back_max = ns / 2 + 1 > 2 ? ns / 2 + 1 : 2;
  memset(T, 0, sizeof(double) * (size_t)ns * na * ns);
  double* ss = (double*)malloc(sizeof(double) * ns);
  double* cen = (double*)malloc(sizeof(double) * na);
  double* ecol = (double*)malloc(sizeof(double) * na);
  for (int s = 0; s < ns; ++s) {
    info->band_lo[s] = info->band_hi[s] = 0;
    if (pit[s]) continue;
    const int a_lo = s - back_max > 0 ? s - back_max : 0;
    int a_hi = s - 1 > 0 ? s - 1 : 0;
    if (a_hi < a_lo + 1) a_hi = a_lo + 1;
    const int b_hi = ns < s + fwd_max ? ns : s + fwd_max;
    int b_lo = ns - 1 < s + 1 ? ns - 1 : s + 1;
    if (b_lo > b_hi - 1) b_lo = b_hi - 1;
    xs_draw(seed, cand, XS_BAND, (uint32_t)s, w);
    const int first = a_lo + (int)(w[0] % (uint32_t)(a_hi - a_lo));
    int last = b_lo + (int)(w[1] % (uint32_t)(b_hi - b_lo));
    while (last < ns) {          /* widen until two live states lie ahead inside the band */
      int ahead = 0;
      for (int j = s + 1; j < last; ++j) ahead += !pit[j];
      if (ahead > 1) break;
      ++last;
    }
    info->band_lo[s] = first; info->band_hi[s] = last;
    /* band weights: clip(N(0,1), 0.1, 1).  The reference redraws the band while it carries no mass on the states
     * ahead or fewer than two non-zero entries (:126-133); with every weight >= 0.1 and a band of at least two
     * entries (first <= s - 2 for s >= 2, first = 0 for s = 1, last >= 3 for s = 0) the first fill always passes,
     * so there is exactly one */
    for (int j = 0; j < ns; ++j) ss[j] = 0.0;
    for (int j = first; j < last; ++j) {
      xs_normal4(seed, cand, XS_BANDW, ((uint32_t)s * 8u) * 64u + (uint32_t)(j >> 2), z);
      ss[j] = xs_clip(z[j & 3], 0.10, 1.0);
    }
    ss[s] /= 2.0;
    if (s == ns - 1) ss[s] = 0.0;
    double tot = 0.0;
    for (int j = first; j < last; ++j) tot += ss[j];
    for (int j = first; j < last; ++j) ss[j] /= tot;
    /* actions: Gaussian bumps around random centres, shared out column by column */
    for (int a = 0; a < na; ++a) {
      if ((a & 3) == 0) xs_draw(seed, cand, XS_ACT, (uint32_t)s * 16u + (uint32_t)(a >> 2), w);
      cen[a] = (double)(first - 1) + xs_u32(w[a & 3]) * (double)(last - (first - 1));
    }
    xs_draw(seed, cand, XS_ACTW, (uint32_t)s, w);
    const double width = xs_clip(-log(1.0 - xo_u53(w[0], w[1])), 0.20, 1.6);
    const double inv_w2 = 1.0 / (width * width);
    for (int j = first; j < last; ++j) {
      double col = 0.0;
      int amin = 0;
      double dmin = 0.0;
      for (int a = 0; a < na; ++a) {
        const double d = cen[a] - (double)j, d2 = d * d;
        ecol[a] = exp(-d2 * inv_w2);
        col += ecol[a];
        if (a == 0 || d2 < dmin) { dmin = d2; amin = a; }
      }
      if (col < XS_EPS) {      /* no action reaches this next state: it goes to the nearest one */
        ecol[amin] = 1.0;
        col = 0.0;
        for (int a = 0; a < na; ++a) col += ecol[a];
      }
      for (int a = 0; a < na; ++a) T[((size_t)s * na + a) * ns + j] = (ecol[a] / col) * ss[j];
    }
    for (int a = 0; a < na; ++a) {
      double* row = T + ((size_t)s * na + a) * ns;
      double rs = 0.0;
      for (int j = first; j < last; ++j) rs += row[j];
      for (int j = first; j < last; ++j) row[j] /= rs;
    }
  }
  free(ss); free(cen); free(ecol);
  /* ---- rewards (:11-63, :193-207) ---- */
  double* pot = (double*)calloc(ns, sizeof(double));
  double* rpos = (double*)calloc(ns, sizeof(double));
  double* npos = (double*)calloc(ns, sizeof(double));
  double* rsa = (double*)calloc((size_t)ns * na, sizeof(double));
  double* nsa = (double*)calloc((size_t)ns * na, sizeof(double));
  {
    xs_draw(seed, cand, XS_POT, 0, w);
    const double base = xo_u53(w[0], w[1]) < 0.5 ? 0.0 : xs_clip(-log(1.0 - xo_u53(w[2], w[3])), 0.20, 5.0);
    xs_draw(seed, cand, XS_POT, 1, w);
    double box = -base + 2.0 * base * xo_u53(w[0], w[1]);
    if (box < 0.0) box = 0.0;
    const int n_items = 1 + (int)(w[2] % 3u);
    const double scale = box / sqrt((double)n_items);
    for (int k = 0; k <= n_items; ++k) {
      double order = 0.0, za, zb, zo, zdummy;
      xs_draw(seed, cand, XS_POT, 2u + (uint32_t)k, w);
      xs_normal2(w[0], w[1], &za, &zb);
      const double ca = za * (-log(1.0 - xs_u32(w[2])) * scale), cb = zb * (-log(1.0 - xs_u32(w[3])) * scale);
      if (k > 0) {
        xs_draw(seed, cand, XS_POT, 8u + (uint32_t)k, w);
        xs_normal2(w[1], w[2], &zo, &zdummy);
        order = (double)(1 + (int)(w[0] % 5u)) + zo;
      }
      for (int j = 0; j < ns; ++j) {
        const double x = (double)j / (double)(2 * ns);
        pot[j] += ca * sin(order * x) + cb * cos(order * x);
      }
    }
    xs_draw(seed, cand, XS_POS, 0, w);
    const double pbase = 0.2 * (-log(1.0 - xo_u53(w[0], w[1]))), ub = xo_u53(w[2], w[3]);
    double c = 0.0;
    for (int j = 0; j < ns; ++j) {
      xs_normal4(seed, cand, XS_POSN, (uint32_t)(j >> 2), z);
      double pdf = z[j & 3] > 0.0 ? z[j & 3] : 0.0;
      if (j == ns - 1) pdf += 0.20;
      pdf *= pbase;
      c += pdf;
      rpos[j] = c;
    }
    const double baseline = 0.1 * c + ub * (0.9 * c - 0.1 * c);
    for (int j = 0; j < ns; ++j) {
      if ((j & 3) == 0) xs_draw(seed, cand, XS_POSU, (uint32_t)(j >> 2), w);
      double u = -0.30 + 0.60 * xs_u32(w[j & 3]);
      npos[j] = (u > 0.0 ? u : 0.0) * pbase;
      rpos[j] -= baseline;
      if (pit[j]) { rpos[j] = 0.0; npos[j] = 0.0; }
    }
    xs_draw(seed, cand, XS_SA, 0, w);
    const double sbase = xs_clip(0.05 * (-log(1.0 - xo_u53(w[0], w[1]))), 0.0, 0.10);
    for (int sa = 0; sa < ns * na; ++sa) {
      double zr, zn;
      xs_draw(seed, cand, XS_SAM, (uint32_t)sa, w);
      const double on = xs_u32(w[0]) > 0.7 ? 1.0 : 0.0;
      xs_normal2(w[1], w[2], &zr, &zn);
      rsa[sa] = sbase * zr * on;
      nsa[sa] = 0.30 * sbase * (zn > 0.0 ? zn : 0.0) * on;
    }
  }
  double r_step = 0.0;
  {
    double zs, zd;
    xs_draw(seed, cand, XS_STEP, 0, w);
    xs_normal2(w[0], w[1], &zs, &zd);
    if (goal) r_step = (zs < 0.0 ? zs : 0.0) * 0.01;
    else if (n_se > 0) r_step = (zs > 0.0 ? zs : 0.0) * 0.01;
  }
  /* ---- terminal rewards repaired against the value function (:209-256) ---- */
  double* bonus = (double*)calloc(ns, sizeof(double));
  double* er = (double*)malloc(sizeof(double) * ns * na);
  double* q = (double*)calloc((size_t)ns * na, sizeof(double));
  bonus[ns - 1] = 1.0;
  const int last_live = goal ? ns - 2 : ns - 1;
  int status = 1;
  for (int tries = 0; tries < 5; ++tries) {
    for (int s = 0; s < ns; ++s)
      for (int a = 0; a < na; ++a) {
        const double* t = T + ((size_t)s * na + a) * ns;
        double e = 0.0;
        for (int j = 0; j < ns; ++j) {
          double r = (((pot[s] - pot[j]) + rpos[j]) + rsa[s * na + a]) + r_step;
          r += bonus[j];
          e = fma(t[j], r, e);
        }
        er[s * na + a] = e;
      }
    const int sw = xs_vi(T, er, ns, na, 0.99, 1, q);
    info->sweeps[tries] = sw;
    info->repair_rounds = tries + 1;
    if (sw >= XS_MAX_SWEEPS) { status = 4; break; }
    double vmin_live = 0.0, vmax_s0 = 0.0, bmin = bonus[0];
    int first_live = 1;
    for (int j = 0; j < ns; ++j) {
      double v = q[j * na];
      for (int a = 1; a < na; ++a) v = q[j * na + a] > v ? q[j * na + a] : v;
      if (!pit[j] && (first_live || v < vmin_live)) { vmin_live = v; first_live = 0; }
      if (bonus[j] < bmin) bmin = bonus[j];
    }
    double v_last = q[last_live * na];
    for (int a = 1; a < na; ++a) v_last = q[last_live * na + a] > v_last ? q[last_live * na + a] : v_last;
    for (int k = 0; k < n_s0; ++k) {
      const int s = info->s0[k];
      double v = q[s * na];
      for (int a = 1; a < na; ++a) v = q[s * na + a] > v ? q[s * na + a] : v;
      if (k == 0 || v > vmax_s0) vmax_s0 = v;
    }
    xs_draw(seed, cand, XS_REPAIR, (uint32_t)tries, w);
    const double pit_gap = bmin - vmin_live + 1.0;
    const double goal_gap = vmax_s0 - v_last + (2.0 + 3.0 * xo_u53(w[0], w[1]));
    if (pit_gap <= 0.0 && goal_gap <= 0.0) { status = 0; break; }
    if (pit_gap > 0.0) {
      const double dec = pit_gap + (1.0 + 9.0 * xs_u32(w[2]));
      for (int j = 0; j < ns; ++j)
        if (pit[j] && !(goal && j == ns - 1)) bonus[j] -= dec;
    }
    if (goal_gap > 0.0) {
      const double extra = 1.0 + 9.0 * xs_u32(w[3]);
      const double lift = 2.0 * goal_gap > extra ? 2.0 * goal_gap : extra;
      bonus[ns - 1] += goal ? lift : (1.0 - 0.99) * lift;
    }
  }
  /* dense outputs */
  for (int s = 0; s < ns; ++s)
    for (int a = 0; a < na; ++a)
      for (int j = 0; j < ns; ++j) {
        const size_t idx = ((size_t)s * na + a) * ns + j;
        double r = (((pot[s] - pot[j]) + rpos[j]) + rsa[s * na + a]) + r_step;
        R[idx] = r + bonus[j];
        noise[idx] = npos[j] + nsa[s * na + a];
      }
  /* ---- acceptance (solver.py:105-148) ---- */
  if (status == 0) {
    const double g2 = exp2(-1.0 / (double)ns);
    for (int sa = 0; sa < ns * na; ++sa) {
      const double* t = T + (size_t)sa * ns;
      double e = 0.0;
      for (int j = 0; j < ns; ++j) e = fma(t[j], R[(size_t)sa * ns + j], e);
      er[sa] = e;
    }
    double* qr = (double*)calloc((size_t)ns * na, sizeof(double));
    memset(q, 0, sizeof(double) * ns * na);
    const int sw_o = xs_vi(T, er, ns, na, g2, 1, q);
    const int sw_r = xs_vi(T, er, ns, na, g2, 0, qr);
    info->sweeps[5] = sw_o; info->sweeps[6] = sw_r;
    if (sw_o >= XS_MAX_SWEEPS || sw_r >= XS_MAX_SWEEPS) status = 4;
    const double scale = (1.0 - g2) * max_steps;
    double gap_min = 0.0;
    for (int k = 0; k < n_s0 && status == 0; ++k) {
      const int s = info->s0[k];
      double vo = q[s * na], vr = qr[s * na];
      for (int a = 1; a < na; ++a) {
        vo = q[s * na + a] > vo ? q[s * na + a] : vo;
        vr = qr[s * na + a] > vr ? qr[s * na + a] : vr;
      }
      const double gap = vo * scale - vr * scale;
      if (k == 0 || gap < gap_min) gap_min = gap;
    }
    info->gap_min = gap_min;
    if (status == 0 && gap_min < 2.0) status = 2;
    if (status == 0) {
      const int K = (int)log2(max_steps) + 1;
      double* P = (double*)calloc((size_t)ns * ns, sizeof(double));
      double* P2 = (double*)malloc(sizeof(double) * ns * ns);
      for (int i = 0; i < ns; ++i) {
        if (pit[i]) {
          for (int k = 0; k < n_s0; ++k) P[(size_t)i * ns + info->s0[k]] = info->s0_prob[k];
        } else {
          int best = 0;
          for (int a = 1; a < na; ++a) if (q[i * na + a] > q[i * na + best]) best = a;
          memcpy(P + (size_t)i * ns, T + ((size_t)i * na + best) * ns, sizeof(double) * ns);
        }
      }
      for (int rep = 0; rep < K; ++rep) {
        for (int i = 0; i < ns; ++i)
          for (int j = 0; j < ns; ++j) {
            double acc = 0.0;
            for (int k = 0; k < ns; ++k) acc = fma(P[(size_t)i * ns + k], P[(size_t)k * ns + j], acc);
            P2[(size_t)i * ns + j] = acc;
          }
        memcpy(P, P2, sizeof(double) * ns * ns);
      }
      double gini = 0.0, ent = 0.0;
      for (int k = 0; k < n_s0; ++k) {
        const double* row = P + (size_t)info->s0[k] * ns;
        double s2 = 0.0, h = 0.0;
        for (int j = 0; j < ns; ++j) {
          const double p = row[j] + 1.0e-12;
          s2 += p * p;
          h += p * log(p);
        }
        const double gk = 1.0 - s2, ek = -h / log((double)ns);
        if (k == 0 || gk < gini) gini = gk;
        if (k == 0 || ek < ent) ent = ek;
      }
      info->gini = gini; info->ent = ent;
      if (!(gini > 0.70 && ent > 0.35)) status = 3;
      free(P); free(P2);
    }
    free(qr);
  }
  info->status = status;
  free(pot); free(rpos); free(npos); free(rsa); free(nsa); free(bonus); free(er); free(q);
  return status;
}

/* =====================================================================================================================
 * Part 3: the ground-truth teacher (AnyMDPSolverOpt, anymdp_solver_opt.py:30-51) as the device computes it for whole
 * task batches (xv_anymdp_solve): value iteration on the tasks' own tables — T recovered from the inclusive CDF rows
 * (p_j = cdf_j - cdf_{j-1}; rows of terminal states are zero, as in the reference), R the fp32 table rewards —
 * synchronous sweeps Q <- ER + gamma T max_a Q until rms(Q_new - Q) <= tol or max_iter sweeps; greedy = first argmax.
 * ===================================================================================================================*/
void xo_anymdp_solve(const xo_anymdp* h, double gamma, double tol, int max_iter, double* q_out, uint8_t* greedy_out,
                     int32_t* iters_out) {
  const int S = h->S, A = h->A, SA = S * A;
  double* T = (double*)malloc(sizeof(double) * (size_t)SA * S);
  double* er = (double*)malloc(sizeof(double) * SA);
  double* q = (double*)malloc(sizeof(double) * SA);
  double* qn = (double*)malloc(sizeof(double) * SA);
  double* v = (double*)malloc(sizeof(double) * S);
  for (int t = 0; t < h->n_task; ++t) {
    for (int s = 0; s < S; ++s) {
      const int term = (int)((h->term_mask[(size_t)t * ((S + 63) / 64) + (s >> 6)] >> (s & 63)) & 1u);
      for (int a = 0; a < A; ++a) {
        const size_t row = (((size_t)t * S + s) * A + a) * (size_t)S;
        double prev = 0.0, e = 0.0;
        for (int j = 0; j < S; ++j) {
          const double c = h->cdf[row + j];
          const double p = term ? 0.0 : c - prev;
          prev = c;
          T[((size_t)s * A + a) * S + j] = p;
          e = fma(p, (double)h->rs[(row + j) * 2], e);
        }
        er[s * A + a] = e;
      }
    }
    for (int k = 0; k < SA; ++k) q[k] = 0.0;
    int it = 0;
    for (; it < max_iter;) {
      for (int j = 0; j < S; ++j) {
        double x = q[j * A];
        for (int a = 1; a < A; ++a) x = q[j * A + a] > x ? q[j * A + a] : x;
        v[j] = x;
      }
      double d2 = 0.0;
      for (int k = 0; k < SA; ++k) {
        double acc = 0.0;
        for (int j = 0; j < S; ++j) acc = fma(T[(size_t)k * S + j], v[j], acc);
        qn[k] = fma(gamma, acc, er[k]);
        d2 += (qn[k] - q[k]) * (qn[k] - q[k]);
      }
      memcpy(q, qn, sizeof(double) * SA);
      ++it;
      if (sqrt(d2 / (double)SA) <= tol) break;
    }
    if (q_out) memcpy(q_out + (size_t)t * SA, q, sizeof(double) * SA);
    if (greedy_out)
      for (int s = 0; s < S; ++s) {
        int best = 0;
        for (int a = 1; a < A; ++a) if (q[s * A + a] > q[s * A + best]) best = a;
        greedy_out[(size_t)t * S + s] = (uint8_t)best;
      }
    if (iters_out) iters_out[t] = it;
  }
  free(T); free(er); free(q); free(qn); free(v);
}

/* =====================================================================================================================
 * Observation models (csrc/anymdp_sampler.hip: anymdp_obs_model_kernel; reference task_sampler.py:78-87, :103-117):
 * scipy.sparse.random(S, n_obs, density) = exactly k = round(density * S * n_obs) cells chosen uniformly without
 * replacement, U[0,1) values; an empty row gets a 1 in a random column; rows normalised; emitted as inclusive row CDFs.
 * Same draws as the device: cell c of matrix m -> Philox(counter = {c, m, 0x40}, key = seed), key = {high random bits |
 * cell index}, the k smallest keys are chosen.
 * ===================================================================================================================*/
static uint64_t xs_obs_key(uint64_t seed, uint64_t mat, uint32_t cell, uint64_t idx_mask, double* val) {
  uint32_t ctr[4] = {cell, (uint32_t)mat, (uint32_t)(mat >> 32), 0x40u};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, w[4];
  xo_philox4x32_10(ctr, key, w);
  if (val) *val = xo_u53(w[2], w[3]);
  return ((((uint64_t)w[0] << 32) | (uint64_t)w[1]) & ~idx_mask) | (uint64_t)cell;
}
static int xs_cmp_u64(const void* a, const void* b) {
  const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}
void xo_anymdp_sample_observation_model(uint64_t seed, int64_t task_base, int n_task, int S, int n_obs, int d_obs, double density,
                                        double maximum_distribution, double* obs_cdf) {
  const double d = density < maximum_distribution / (double)n_obs ? density : maximum_distribution / (double)n_obs;
  const long long k = (long long)rint(d * (double)S * (double)n_obs);
  const int M = S * n_obs;
  uint64_t idx_mask = 1;
  while (idx_mask < (uint64_t)M) idx_mask <<= 1;
  idx_mask -= 1;
  uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)M);
  for (size_t q = 0; q < (size_t)n_task * d_obs; ++q) {
    const uint64_t mat = (uint64_t)task_base * (uint64_t)d_obs + q;
    uint64_t thr = 0;
    if (k > 0) {   /* the k-th smallest key (a sort here, a bisection on the device: the same threshold) */
      for (int c = 0; c < M; ++c) keys[c] = xs_obs_key(seed, mat, (uint32_t)c, idx_mask, 0);
      qsort(keys, (size_t)M, sizeof(uint64_t), xs_cmp_u64);
      thr = keys[(k < M ? k : M) - 1];
    }
    double* out = obs_cdf + q * (size_t)M;
    for (int row = 0; row < S; ++row) {
      double* o = out + (size_t)row * n_obs;
      double acc = 0.0;
      for (int j = 0; j < n_obs; ++j) {
        double v;
        const uint64_t key = xs_obs_key(seed, mat, (uint32_t)(row * n_obs + j), idx_mask, &v);
        acc += (k > 0 && key <= thr) ? v : 0.0;
        o[j] = acc;
      }
      if (acc == 0.0) {
        uint32_t ctr[4] = {(uint32_t)row, (uint32_t)mat, (uint32_t)(mat >> 32), 0x41u};
        uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, w[4];
        xo_philox4x32_10(ctr, key, w);
        const int col = (int)(((uint64_t)w[0] * (uint64_t)(uint32_t)n_obs) >> 32);
        for (int j = 0; j < n_obs; ++j) o[j] = j >= col ? 1.0 : 0.0;
        acc = 1.0;
      }
      for (int j = 0; j < n_obs; ++j) o[j] = o[j] / acc;
    }
  }
  free(keys);
}
