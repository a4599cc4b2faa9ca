// anymdp_sampler.hip — AnyMDP task sampler on the device: generate, repair and accept thousands of candidate tasks per
// launch, one workgroup per candidate, and write the accepted ones straight into the step engine's row records.
//
// Reference: xenoverse/anymdp/task_sampler.py:15-65 (AnyMDPTaskSampler), task_sampler_utils.py:11-256 (generative
// model: start / terminal states, banded transition kernel split over the actions, potential + position + state-action
// rewards, terminal rewards repaired against the value function) and solver.py:57-148 (acceptance: value gap between
// the optimal and the uniform policy, long-run occupancy of the greedy policy).  There the work is a Python triple loop
// (518 s per 64x8 task without numba).  Here:
//   * thread (s, a) of the workgroup owns row T[s, a, :] in registers for the whole life of the candidate (S <= 64):
//     generation, 5 + 2 value iterations and the emission never move the transition tensor through memory;
//   * every random quantity is a counter-based draw, Philox4x32-10(counter = {candidate, index, purpose}, key = seed):
//     a pure function of its coordinates, so the threads of a state simply recompute what they share (band, weights)
//     instead of exchanging it, and a candidate is reproducible whatever the launch geometry.  Same distributions as the
//     reference, not NumPy's sequential stream (the seed-compatible sampler is host code: anymdp/task_sampler.py);
//   * value iteration is the synchronous sweep Q <- ER + gamma T V(Q) to rms update <= 1e-4 (the reference's stopping
//     rule; same fixed point as its damped Gauss-Seidel): V through LDS broadcast reads, two barriers per sweep;
//   * accepted candidates are emitted in the layout xv_anymdp_create reads (blocks of 7 {cdf, reward, noise}).
// oracle/xeno_oracle_sampler.c (part 2) restates this file draw for draw; tests/test_gpu_sampler.py diffs them.
#include "philox.h"
#include "xv_common.h"

namespace {

enum : uint32_t { XS_HEAD = 0, XS_PERM, XS_S0, XS_PIT, XS_PITS, XS_BAND, XS_BANDW, XS_ACT, XS_ACTW, XS_POT, XS_POS,
                  XS_POSN, XS_POSU, XS_SA, XS_SAM, XS_STEP, XS_REPAIR };
constexpr int XS_MAX_SWEEPS = 20000;
constexpr double XS_EPS = 1e-10;
constexpr double XS_TWO_PI = 6.283185307179586476925286766559;

struct SamplerArgs {
  uint64_t seed;
  int64_t cand_base;
  int n_cand, S, A, s0_max, row_lines;
  // table outputs, slot = candidate index within the launch (written for accepted candidates only)
  double* rows;          // [n_cand][S][A][row_lines][16]
  int32_t* state_map;    // [n_cand][S]
  uint64_t* term_mask;   // [n_cand][1]
  double* s0_cdf;        // [n_cand][s0_max]
  int32_t* s0_ids;       // [n_cand][s0_max]
  int32_t* max_steps;    // [n_cand]
  // optional dense outputs (written for every candidate that was generated, accepted or not)
  double *transition, *reward, *noise;   // [n_cand][S][A][S]
  xv_anymdp_cand_info* info;             // [n_cand]
  int32_t* status;                       // [n_cand]
};

__device__ __forceinline__ xv_u32x4 xs_draw(const SamplerArgs& P, uint64_t cand, uint32_t purpose, uint32_t idx) {
  return xv_philox4x32_10((uint32_t)cand, (uint32_t)(cand >> 32), idx, purpose, (uint32_t)P.seed, (uint32_t)(P.seed >> 32));
}
__device__ __forceinline__ double xs_u32(uint32_t w) { return (double)w * (1.0 / 4294967296.0); }
__device__ __forceinline__ void xs_normal2(uint32_t wa, uint32_t wb, double& z0, double& z1) {
  const double u1 = ((double)wa + 1.0) * (1.0 / 4294967296.0), u2 = xs_u32(wb);
  const double r = sqrt(-2.0 * log(u1));
  z0 = r * cos(XS_TWO_PI * u2);
  z1 = r * sin(XS_TWO_PI * u2);
}
__device__ __forceinline__ uint32_t xs_word(const xv_u32x4& w, int k) {
  return k == 0 ? w.x : (k == 1 ? w.y : (k == 2 ? w.z : w.w));
}
// element k (0..3) of the four normals of one Philox call
__device__ __forceinline__ double xs_normal_of(const xv_u32x4& w, int k) {
  double z0, z1;
  if (k < 2) xs_normal2(w.x, w.y, z0, z1); else xs_normal2(w.z, w.w, z0, z1);
  return (k & 1) ? z1 : z0;
}
__device__ __forceinline__ double xs_clip(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ double xs_expo(double u) { return -log(1.0 - u); }

// Synchronous value iteration on the candidate's rows.  q: this thread's Q[s, a] in/out.  Returns the sweep count
// (uniform).  Qs[SA], Vs[SP], part[8]: LDS.  Each sweep: Q -> LDS | barrier | V = max / mean per state, and the
// previous sweep's update norm is inspected (uniform decision) | barrier | 64 broadcast reads of V, one fma each.
template <int SP>
__device__ __forceinline__ int xs_value_iteration(const double (&Trow)[SP], double er, double gamma, bool greedy, bool own, int S, int A,
                                  double& q, double* Qs, double* Vs, double* part, int n_waves) {
  const int tid = threadIdx.x, SA = S * A;
  double d2 = 0.0;
  int it = 0;
  for (;;) {
    if (own) Qs[tid] = q;
    {
      double x = d2;
      for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
      if ((tid & 63) == 0) part[tid >> 6] = x;
    }
    __syncthreads();
    if (it > 0) {
      double tot = 0.0;
      for (int w = 0; w < n_waves; ++w) tot += part[w];
      if (sqrt(tot / (double)SA) <= 1.0e-4 || it >= XS_MAX_SWEEPS) break;
    }
    if (tid < S) {
      const double* row = Qs + tid * A;
      double v;
      if (greedy) {
        v = row[0];
        for (int a = 1; a < A; ++a) v = row[a] > v ? row[a] : v;
      } else {
        v = 0.0;
        for (int a = 0; a < A; ++a) v += row[a];
        v /= (double)A;
      }
      Vs[tid] = v;
    }
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < SP; ++j)
      if (j < S) acc = fma(Trow[j], Vs[j], acc);
    const double qn = fma(gamma, acc, er);
    d2 = own ? (qn - q) * (qn - q) : 0.0;
    q = qn;
    ++it;
  }
  __syncthreads();
  return it;
}

template <int SP>
__global__ __launch_bounds__(512) void anymdp_sampler_kernel(SamplerArgs P) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int S = P.S, A = P.A, SA = S * A, tid = threadIdx.x;
  const int n_waves = (int)(blockDim.x >> 6);
  const int ci = blockIdx.x;
  const uint64_t cand = (uint64_t)(P.cand_base + ci);
  // LDS carve-up
  double* Qs = lds;                    // [SA]    (also: action centres during generation)
  double* Vs = Qs + SA;                // [SP]
  double* part = Vs + SP;              // [8]
  double* pot = part + 8;              // [SP]
  double* rpos = pot + SP;             // [SP]    (first: position-reward pdf)
  double* npos = rpos + SP;            // [SP]
  double* bonus = npos + SP;           // [SP]
  double* vstate = bonus + SP;         // [SP]    V(s) = max_a Q after a value iteration
  int* pitm = reinterpret_cast<int*>(vstate + SP);   // [SP]
  double* Pm = reinterpret_cast<double*>(pitm + SP); // [2][SP*SP] occupancy matrices (acceptance only)

  const bool own = tid < SA;
  const int s = own ? tid / A : 0, a = own ? tid - s * A : 0;

  // ---- uniform scalars: every thread derives them from the same draws ----
  xv_u32x4 w = xs_draw(P, cand, XS_HEAD, 0);
  const double lower = 4.0 * S > 100 ? 4.0 * S : 100;
  double upper = 8.0 * S < 500 ? 8.0 * S : 500;
  if (upper < lower + 1) upper = lower + 1;
  const double max_steps = lower + xv_u53(w.x, w.y) * (upper - lower);
  // start states (task_sampler_utils.py:70-79)
  double w0[3] = {1.0, 0.0, 0.0};
  for (uint32_t r = 0; r < 16; ++r) {
    const xv_u32x4 q4 = xs_draw(P, cand, XS_S0, r);
    double z0, z1, z2, z3;
    xs_normal2(q4.x, q4.y, z0, z1);
    xs_normal2(q4.z, q4.w, z2, z3);
    const double c0 = z0 > 0.0 ? z0 : 0.0, c1 = z1 > 0.0 ? z1 : 0.0, c2 = z2 > 0.0 ? z2 : 0.0;
    if ((c0 + c1) + c2 >= XS_EPS) { w0[0] = c0; w0[1] = c1; w0[2] = c2; break; }
  }
  int s0_id[3], n_s0 = 0;
  double s0_p[3], s0sum = 0.0;
  for (int k = 0; k < 3; ++k)
    if (w0[k] > XS_EPS) { s0_id[n_s0] = k; s0_p[n_s0] = w0[k]; s0sum += w0[k]; ++n_s0; }
  for (int k = n_s0; k < 3; ++k) { s0_id[k] = 0; s0_p[k] = 0.0; }
  for (int k = 0; k < n_s0; ++k) s0_p[k] /= s0sum;
  // terminal states (:81-93)
  w = xs_draw(P, cand, XS_PIT, 0);
  double p_pit = -0.20 + 0.60 * xv_u53(w.x, w.y);
  if (p_pit < 0.0) p_pit = 0.0;
  const int goal = xs_u32(w.z) < 0.3 ? 1 : 0;
  {
    int mine = 0;
    for (uint32_t r = 0; r < 64; ++r) {
      mine = 0;
      if (tid < S) {
        const xv_u32x4 q4 = xs_draw(P, cand, XS_PITS, r * 64u + (uint32_t)(tid >> 2));
        mine = xs_u32(xs_word(q4, tid & 3)) < p_pit ? 1 : 0;
      }
      const int cnt = __syncthreads_count(mine);
      if ((double)cnt < (double)S * p_pit + 1.0) break;
    }
    if (tid < S) {
      for (int k = 0; k < n_s0; ++k) if (tid == s0_id[k]) mine = 0;
      if (tid == S - 1) mine = goal;
      pitm[tid] = mine;
    }
  }
  __syncthreads();
  int n_se = 0;
  for (int j = 0; j < S; ++j) n_se += pitm[j];
  const bool live_row = own && !pitm[s];

  // ---- band of every live state (:95-124), by the state's own thread ----
  int* bandlo = reinterpret_cast<int*>(Pm + 2 * SP * SP);   // [SP]
  int* bandhi = bandlo + SP;                                  // [SP]
  if (tid < S) {
    int first = 0, last = 0;
    if (!pitm[tid]) {
      const int st = tid;
      const int fwd_max = S / 4 + 1 > 2 ? S / 4 + 1 : 2, back_max = S / 2 + 1 > 2 ? S / 2 + 1 : 2;
      const int a_lo = st - back_max > 0 ? st - back_max : 0;
      int a_hi = st - 1 > 0 ? st - 1 : 0;
      if (a_hi < a_lo + 1) a_hi = a_lo + 1;
      const int b_hi = S < st + fwd_max ? S : st + fwd_max;
      int b_lo = S - 1 < st + 1 ? S - 1 : st + 1;
      if (b_lo > b_hi - 1) b_lo = b_hi - 1;
      w = xs_draw(P, cand, XS_BAND, (uint32_t)st);
      first = a_lo + (int)(w.x % (uint32_t)(a_hi - a_lo));
      last = b_lo + (int)(w.y % (uint32_t)(b_hi - b_lo));
      while (last < S) {          // widen until two live states lie ahead inside the band
        int ahead = 0;
        for (int j = st + 1; j < last; ++j) ahead += !pitm[j];
        if (ahead > 1) break;
        ++last;
      }
    }
    bandlo[tid] = first; bandhi[tid] = last;
  }
  __syncthreads();
  // band weights clip(N(0,1), 0.1, 1) (:126-133), element (state, next state) by element over the whole workgroup,
  // staged in the (still unused) occupancy buffer
  for (int el = tid; el < S * S; el += (int)blockDim.x) {
    const int st = el / S, j = el - st * S;
    double v = 0.0;
    if (j >= bandlo[st] && j < bandhi[st]) {
      const xv_u32x4 q4 = xs_draw(P, cand, XS_BANDW, ((uint32_t)st * 8u) * 64u + (uint32_t)(j >> 2));
      v = xs_clip(xs_normal_of(q4, j & 3), 0.10, 1.0);
    }
    Pm[el] = v;
  }
  __syncthreads();
  if (tid < S && !pitm[tid]) {     // damp the self loop (none at the last state), then the row total
    Pm[tid * S + tid] = (tid == S - 1) ? 0.0 : Pm[tid * S + tid] / 2.0;
    double tot = 0.0;
    for (int j = bandlo[tid]; j < bandhi[tid]; ++j) tot += Pm[tid * S + j];
    Vs[tid] = tot;
  }
  __syncthreads();

  // ---- this thread's row of the transition tensor (:154-175) ----
  double Trow[SP];
#pragma unroll
  for (int j = 0; j < SP; ++j) Trow[j] = 0.0;
  int first = 0, last = 0;
  if (live_row) {
    first = bandlo[s]; last = bandhi[s];
    const double tot = Vs[s];
#pragma unroll
    for (int j = 0; j < SP; ++j)
      if (j >= first && j < last) Trow[j] = Pm[s * S + j] / tot;
    // this action's centre; the state's A centres are shared through LDS
    const xv_u32x4 q4 = xs_draw(P, cand, XS_ACT, (uint32_t)s * 16u + (uint32_t)(a >> 2));
    Qs[tid] = (double)(first - 1) + xs_u32(xs_word(q4, a & 3)) * (double)(last - (first - 1));
  }
  __syncthreads();
  if (live_row) {
    w = xs_draw(P, cand, XS_ACTW, (uint32_t)s);
    const double width = xs_clip(xs_expo(xv_u53(w.x, w.y)), 0.20, 1.6);
    const double inv_w2 = 1.0 / (width * width);
    const double* cen = Qs + s * A;
    double rs = 0.0;
#pragma unroll
    for (int j = 0; j < SP; ++j) {
      if (j >= first && j < last) {
        double col = 0.0, dmin = 0.0, e_me = 0.0;
        int amin = 0;
        for (int b = 0; b < A; ++b) {
          const double d = cen[b] - (double)j, d2 = d * d;
          const double e = exp(-d2 * inv_w2);
          col += e;
          if (b == a) e_me = e;
          if (b == 0 || d2 < dmin) { dmin = d2; amin = b; }
        }
        if (col < XS_EPS) {      // no action reaches this next state: it goes to the nearest one
          col = 0.0;
          for (int b = 0; b < A; ++b) {
            const double d = cen[b] - (double)j;
            col += (b == amin) ? 1.0 : exp(-(d * d) * inv_w2);
          }
          if (a == amin) e_me = 1.0;
        }
        Trow[j] = (e_me / col) * Trow[j];
        rs += Trow[j];
      }
    }
#pragma unroll
    for (int j = 0; j < SP; ++j)
      if (j >= first && j < last) Trow[j] /= rs;
  }
  __syncthreads();

  // ---- rewards (:11-63, :193-207): per-state pieces in LDS, the state-action piece in registers ----
  double pbase, ub;
  {
    w = xs_draw(P, cand, XS_POS, 0);
    pbase = 0.2 * xs_expo(xv_u53(w.x, w.y));
    ub = xv_u53(w.z, w.w);
  }
  if (tid < S) {
    const int j = tid;
    // potential: a few Fourier terms over the state index (RandomFourier, utils/random_nn.py:346-368)
    xv_u32x4 q4 = xs_draw(P, cand, XS_POT, 0);
    const double base = xv_u53(q4.x, q4.y) < 0.5 ? 0.0 : xs_clip(xs_expo(xv_u53(q4.z, q4.w)), 0.20, 5.0);
    q4 = xs_draw(P, cand, XS_POT, 1);
    double box = -base + 2.0 * base * xv_u53(q4.x, q4.y);
    if (box < 0.0) box = 0.0;
    const int n_items = 1 + (int)(q4.z % 3u);
    const double scale = box / sqrt((double)n_items);
    const double x = (double)j / (double)(2 * S);
    double pj = 0.0;
    for (int k = 0; k <= n_items; ++k) {
      double za, zb, order = 0.0;
      q4 = xs_draw(P, cand, XS_POT, 2u + (uint32_t)k);
      xs_normal2(q4.x, q4.y, za, zb);
      const double ca = za * (xs_expo(xs_u32(q4.z)) * scale), cb = zb * (xs_expo(xs_u32(q4.w)) * scale);
      if (k > 0) {
        double zo, zd;
        q4 = xs_draw(P, cand, XS_POT, 8u + (uint32_t)k);
        xs_normal2(q4.y, q4.z, zo, zd);
        order = (double)(1 + (int)(q4.x % 5u)) + zo;
      }
      pj += ca * sin(order * x) + cb * cos(order * x);
    }
    pot[j] = pj;
    q4 = xs_draw(P, cand, XS_POSN, (uint32_t)(j >> 2));
    double pdf = xs_normal_of(q4, j & 3);
    pdf = pdf > 0.0 ? pdf : 0.0;
    if (j == S - 1) pdf += 0.20;
    rpos[j] = pdf * pbase;
    q4 = xs_draw(P, cand, XS_POSU, (uint32_t)(j >> 2));
    const double u = -0.30 + 0.60 * xs_u32(xs_word(q4, j & 3));
    npos[j] = pitm[j] ? 0.0 : (u > 0.0 ? u : 0.0) * pbase;
    bonus[j] = (j == S - 1) ? 1.0 : 0.0;
  }
  __syncthreads();
  double cdf_j = 0.0, c_all = 0.0;
  for (int i = 0; i < S; ++i) {         // cumulative sums in index order, as numpy.cumsum forms them
    c_all += rpos[i];
    if (i == tid) cdf_j = c_all;
  }
  __syncthreads();
  if (tid < S) {
    const double baseline = 0.1 * c_all + ub * (0.9 * c_all - 0.1 * c_all);
    rpos[tid] = pitm[tid] ? 0.0 : cdf_j - baseline;
  }
  double rsa = 0.0, nsa = 0.0, r_step = 0.0;
  {
    w = xs_draw(P, cand, XS_SA, 0);
    const double sbase = xs_clip(0.05 * xs_expo(xv_u53(w.x, w.y)), 0.0, 0.10);
    if (own) {
      double zr, zn;
      const xv_u32x4 q4 = xs_draw(P, cand, XS_SAM, (uint32_t)tid);
      const double on = xs_u32(q4.x) > 0.7 ? 1.0 : 0.0;
      xs_normal2(q4.y, q4.z, zr, zn);
      rsa = sbase * zr * on;
      nsa = 0.30 * sbase * (zn > 0.0 ? zn : 0.0) * on;
    }
    double zs, zd;
    w = xs_draw(P, cand, XS_STEP, 0);
    xs_normal2(w.x, w.y, zs, zd);
    if (goal) r_step = (zs < 0.0 ? zs : 0.0) * 0.01;
    else if (n_se > 0) r_step = (zs > 0.0 ? zs : 0.0) * 0.01;
  }
  __syncthreads();

  // ---- terminal rewards repaired against the value function (:209-256) ----
  const int last_live = goal ? S - 2 : S - 1;
  int status = 1, repair_rounds = 0;
  int sweeps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double q = 0.0, er = 0.0;
  const double pot_s = own ? pot[s] : 0.0;
  for (int tries = 0; tries < 5; ++tries) {
    er = 0.0;
#pragma unroll
    for (int j = 0; j < SP; ++j)
      if (j < S) {
        double r = (((pot_s - pot[j]) + rpos[j]) + rsa) + r_step;
        r += bonus[j];
        er = fma(Trow[j], r, er);
      }
    const int sw = xs_value_iteration<SP>(Trow, er, 0.99, true, own, S, A, q, Qs, Vs, part, n_waves);
    sweeps[tries] = sw;
    repair_rounds = tries + 1;
    if (sw >= XS_MAX_SWEEPS) { status = 4; break; }
    if (own) Qs[tid] = q;
    __syncthreads();
    if (tid < S) {
      double v = Qs[tid * A];
      for (int b = 1; b < A; ++b) v = Qs[tid * A + b] > v ? Qs[tid * A + b] : v;
      vstate[tid] = v;
    }
    __syncthreads();
    double vmin_live = 0.0, vmax_s0 = 0.0, bmin = bonus[0];
    bool first_live = true;
    for (int j = 0; j < S; ++j) {
      const double v = vstate[j];
      if (!pitm[j] && (first_live || v < vmin_live)) { vmin_live = v; first_live = false; }
      if (bonus[j] < bmin) bmin = bonus[j];
    }
    for (int k = 0; k < n_s0; ++k) {
      const double v = vstate[s0_id[k]];
      if (k == 0 || v > vmax_s0) vmax_s0 = v;
    }
    w = xs_draw(P, cand, XS_REPAIR, (uint32_t)tries);
    const double pit_gap = bmin - vmin_live + 1.0;
    const double goal_gap = vmax_s0 - vstate[last_live] + (2.0 + 3.0 * xv_u53(w.x, w.y));
    if (pit_gap <= 0.0 && goal_gap <= 0.0) { status = 0; break; }
    __syncthreads();       // everyone has read bonus / vstate
    if (tid < S) {
      double bj = bonus[tid];
      if (pit_gap > 0.0 && pitm[tid] && !(goal && tid == S - 1)) bj -= pit_gap + (1.0 + 9.0 * xs_u32(w.z));
      if (goal_gap > 0.0 && tid == S - 1) {
        const double extra = 1.0 + 9.0 * xs_u32(w.w);
        const double lift = 2.0 * goal_gap > extra ? 2.0 * goal_gap : extra;
        bj += goal ? lift : (1.0 - 0.99) * lift;
      }
      bonus[tid] = bj;
    }
    __syncthreads();
  }

  // ---- acceptance (solver.py:105-148) ----
  double gini = 0.0, ent = 0.0, gap_min = 0.0;
  if (status == 0) {
    const double g2 = exp2(-1.0 / (double)S);
    er = 0.0;
#pragma unroll
    for (int j = 0; j < SP; ++j)
      if (j < S) {
        double r = (((pot_s - pot[j]) + rpos[j]) + rsa) + r_step;
        r += bonus[j];
        er = fma(Trow[j], r, er);
      }
    double qo = 0.0, qr = 0.0;
    sweeps[5] = xs_value_iteration<SP>(Trow, er, g2, true, own, S, A, qo, Qs, Vs, part, n_waves);
    if (own) Qs[tid] = qo;
    __syncthreads();
    int greedy_a = 0;
    if (tid < S) {
      double v = Qs[tid * A];
      for (int b = 1; b < A; ++b) if (Qs[tid * A + b] > v) { v = Qs[tid * A + b]; greedy_a = b; }
      vstate[tid] = v;
    }
    __syncthreads();
    // occupancy matrix row of the greedy action: thread (s, a) with a == greedy(s) owns row s
    {
      int g_s = 0;
      if (own) {
        double v = Qs[s * A];
        for (int b = 1; b < A; ++b) if (Qs[s * A + b] > v) { v = Qs[s * A + b]; g_s = b; }
      }
      if (own && a == g_s) {
#pragma unroll
        for (int j = 0; j < SP; ++j)
          if (j < S) {
            double pj = Trow[j];
            if (pitm[s]) {
              pj = 0.0;
              for (int k = 0; k < n_s0; ++k) if (j == s0_id[k]) pj = s0_p[k];
            }
            Pm[s * S + j] = pj;
          }
      }
    }
    (void)greedy_a;
    double vo_s0[3];
    for (int k = 0; k < 3; ++k) vo_s0[k] = k < n_s0 ? vstate[s0_id[k]] : 0.0;
    __syncthreads();
    sweeps[6] = xs_value_iteration<SP>(Trow, er, g2, false, own, S, A, qr, Qs, Vs, part, n_waves);
    if (sweeps[5] >= XS_MAX_SWEEPS || sweeps[6] >= XS_MAX_SWEEPS) status = 4;
    if (own) Qs[tid] = qr;
    __syncthreads();
    const double scale = (1.0 - g2) * max_steps;
    for (int k = 0; k < n_s0; ++k) {
      const int s0 = s0_id[k];
      double vr = Qs[s0 * A];
      for (int b = 1; b < A; ++b) vr = Qs[s0 * A + b] > vr ? Qs[s0 * A + b] : vr;
      const double gap = vo_s0[k] * scale - vr * scale;
      if (k == 0 || gap < gap_min) gap_min = gap;
    }
    if (status == 0 && gap_min < 2.0) status = 2;
    if (status == 0) {      // uniform
      const int K = (int)log2(max_steps) + 1;
      double* Pa = Pm;
      double* Pb = Pm + SP * SP;
      for (int rep = 0; rep < K; ++rep) {
        __syncthreads();
        for (int e = tid; e < S * S; e += blockDim.x) {
          const int i = e / S, j = e - i * S;
          double acc = 0.0;
          for (int k = 0; k < S; ++k) acc = fma(Pa[i * S + k], Pa[k * S + j], acc);
          Pb[e] = acc;
        }
        double* t = Pa; Pa = Pb; Pb = t;
      }
      __syncthreads();
      for (int k = 0; k < n_s0; ++k) {
        const double* row = Pa + s0_id[k] * S;
        double s2 = 0.0, h = 0.0;
        for (int j = 0; j < S; ++j) {
          const double p = row[j] + 1.0e-12;
          s2 += p * p;
          h += p * log(p);
        }
        const double gk = 1.0 - s2, ek = -h / log((double)S);
        if (k == 0 || gk < gini) gini = gk;
        if (k == 0 || ek < ent) ent = ek;
      }
      if (!(gini > 0.70 && ent > 0.35)) status = 3;
    }
  }

  // ---- outputs ----
  if (tid == 0) {
    P.status[ci] = status;
    if (P.info) {
      xv_anymdp_cand_info& o = P.info[ci];
      o.status = status; o.goal = goal; o.n_s0 = n_s0; o.repair_rounds = repair_rounds;
      for (int k = 0; k < 4; ++k) { o.s0[k] = k < n_s0 ? s0_id[k] : 0; o.s0_prob[k] = k < n_s0 ? s0_p[k] : 0.0; }
      for (int k = 0; k < 8; ++k) o.sweeps[k] = sweeps[k];
      o.max_steps = max_steps; o.gini = gini; o.ent = ent; o.gap_min = gap_min;
    }
  }
  if (P.info && own && a == 0) {
    P.info[ci].band_lo[s] = first;
    P.info[ci].band_hi[s] = last;
    P.info[ci].s_e[s] = (uint8_t)pitm[s];
  }
  if (own && (P.transition || P.reward || P.noise)) {
    const size_t o = (((size_t)ci * S + s) * A + a) * S;
#pragma unroll
    for (int j = 0; j < SP; ++j)
      if (j < S) {
        if (P.transition) P.transition[o + j] = Trow[j];
        if (P.reward) {
          const double r = (((pot_s - pot[j]) + rpos[j]) + rsa) + r_step;
          P.reward[o + j] = r + bonus[j];
        }
        if (P.noise) P.noise[o + j] = npos[j] + nsa;
      }
  }
  // state_mapping: a Fisher-Yates permutation of the state ids (task_sampler.py:44), sequential by nature
  if (tid == 0 && (P.state_map || P.info)) {
    int* sm = reinterpret_cast<int*>(Qs);
    for (int i = 0; i < S; ++i) sm[i] = i;
    for (int i = S - 1; i >= 1; --i) {
      const xv_u32x4 q4 = xs_draw(P, cand, XS_PERM, (uint32_t)(i >> 2));
      const int j = (int)(xs_word(q4, i & 3) % (uint32_t)(i + 1));
      const int t = sm[i]; sm[i] = sm[j]; sm[j] = t;
    }
    for (int i = 0; i < S; ++i) {
      if (P.info) P.info[ci].state_map[i] = sm[i];
      if (P.state_map && status == 0) P.state_map[(size_t)ci * S + i] = sm[i];
    }
  }
  if (status != 0 || P.rows == nullptr) return;
  // accepted: the step engine's tables.  Row record of (s, a): line 0 (fence) left zero — xv_anymdp_create completes
  // it — then blocks of 7 entries {cdf fp64, reward fp32, noise fp32}; cdf = cumsum(row) / cumsum(row)[-1] as
  // numpy.random.choice forms it, 1.0 for the all-zero rows of terminal states, 2.0 past the last state.
  if (own) {
    double* rec = P.rows + (((size_t)ci * S + s) * A + a) * (size_t)P.row_lines * 16;
    for (int k = 0; k < P.row_lines * 16; ++k) rec[k] = 0.0;
    double tot = 0.0;
#pragma unroll
    for (int j = 0; j < SP; ++j) if (j < S) tot += Trow[j];
    const bool zero_row = tot == 0.0;
    double c = 0.0;
    const int n_ent = (P.row_lines - 1) * 7;
    int j = 0;
#pragma unroll
    for (int jj = 0; jj < SP; ++jj)
      if (jj < S) {
        c += Trow[jj];
        const double cdf = zero_row ? 1.0 : c / tot;
        const double r = ((((pot_s - pot[jj]) + rpos[jj]) + rsa) + r_step) + bonus[jj];
        const float rf = (float)r, nf = (float)(npos[jj] + nsa);
        double* ent = rec + 16 * (1 + jj / 7) + 2 * (jj % 7);
        ent[0] = cdf;
        ent[1] = __hiloint2double((int)__float_as_uint(nf), (int)__float_as_uint(rf));
        j = jj + 1;
      }
    for (; j < n_ent; ++j) rec[16 * (1 + j / 7) + 2 * (j % 7)] = 2.0;
  }
  if (tid == 0) {
    uint64_t m = 0;
    for (int j = 0; j < S; ++j) if (pitm[j]) m |= 1ull << j;
    P.term_mask[ci] = m;
    double c = 0.0, ctot = 0.0;
    for (int k = 0; k < n_s0; ++k) ctot += s0_p[k];
    for (int k = 0; k < P.s0_max; ++k) {
      if (k < n_s0) {      // cumsum(p) / cumsum(p)[-1], as numpy.random.choice forms it (anymdp_env.py:89)
        c += s0_p[k];
        P.s0_cdf[(size_t)ci * P.s0_max + k] = c / ctot;
        P.s0_ids[(size_t)ci * P.s0_max + k] = s0_id[k];
      } else {
        P.s0_cdf[(size_t)ci * P.s0_max + k] = 1.0;
        P.s0_ids[(size_t)ci * P.s0_max + k] = s0_id[n_s0 - 1];
      }
    }
    const double ms = ceil(max_steps);
    P.max_steps[ci] = (int32_t)ms;
  }
}


// ------------------------------------------------------------------------------------------------
// The same sampler for 64 < S <= 256 (round 4; the reference's GarnetTaskSampler defaults to 128 states and its
// MultiTokensAnyPOMDPTaskSampler to 256, task_sampler.py:90-126).  A candidate's transition tensor is S * A rows of S
// doubles (2.6 MB at 256 x 5): it no longer lives in registers but in a per-candidate global scratch, TRANSPOSED in blocks
// of 64 rows — Tt[(row >> 6) * S + j][row & 63] — so that the 64 lanes of a wave, which own 64 consecutive rows, read
// element j of their rows as one coalesced 512-byte access.  Thread tid owns rows tid, tid + blockDim, ... (R <= 4 of them);
// every chain keeps the order of the register kernel (fma over j ascending: a zero outside a row's band adds nothing), so
// both kernels and the oracle restatement (oracle/xeno_oracle_sampler.c) agree to rounding, candidates draw for draw.
// Same Philox coordinates as the register kernel: the index strides (64 words per 256 states) were laid out for 256 states.
// ------------------------------------------------------------------------------------------------
struct BigScratch {
  double* Tt;        // [n_cand][n_rb * S * 64]
  double* Pm;        // [n_cand][2 * S * S]   band weights during generation, occupancy matrices at acceptance
  int n_rb;
};

__device__ __forceinline__ double xsb_t(const double* Tt, int S, int r, int j) { return Tt[((size_t)(r >> 6) * S + j) * 64 + (r & 63)]; }

// reward of (row with potential pot_s and state-action piece rsa) -> next state j, without the terminal bonus
__device__ __forceinline__ void xsb_sa_pieces(const SamplerArgs& P, uint64_t cand, double sbase, int r, double& rsa, double& nsa) {
  double zr, zn;
  const xv_u32x4 q4 = xs_draw(P, cand, XS_SAM, (uint32_t)r);
  const double on = xs_u32(q4.x) > 0.7 ? 1.0 : 0.0;
  xs_normal2(q4.y, q4.z, zr, zn);
  rsa = sbase * zr * on;
  nsa = 0.30 * sbase * (zn > 0.0 ? zn : 0.0) * on;
}

template <int R>
__device__ __forceinline__ int xsb_value_iteration(const double* Tt, int S, int A, int SA, double gamma, bool greedy,
                                                   const double (&er)[R], double (&q)[R], const int (&wf)[R], const int (&wl)[R],
                                                   double* Qs, double* Vs, double* part, int n_waves) {
  const int tid = threadIdx.x, BD = blockDim.x;
  double d2 = 0.0;
  int it = 0;
  for (;;) {
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int r = k * BD + tid;
      if (r < SA) Qs[r] = q[k];
    }
    {
      double x = d2;
      for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
      if ((tid & 63) == 0) part[tid >> 6] = x;
    }
    __syncthreads();
    if (it > 0) {
      double tot = 0.0;
      for (int w = 0; w < n_waves; ++w) tot += part[w];
      if (sqrt(tot / (double)SA) <= 1.0e-4 || it >= XS_MAX_SWEEPS) break;
    }
    if (tid < S) {
      const double* row = Qs + tid * A;
      double v;
      if (greedy) {
        v = row[0];
        for (int a = 1; a < A; ++a) v = row[a] > v ? row[a] : v;
      } else {
        v = 0.0;
        for (int a = 0; a < A; ++a) v += row[a];
        v /= (double)A;
      }
      Vs[tid] = v;
    }
    __syncthreads();
    d2 = 0.0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int r = k * BD + tid;
      const double* col = Tt + ((size_t)(r >> 6) * S) * 64 + (r & 63);
      double acc = 0.0;
      for (int j = wf[k]; j < wl[k]; ++j) acc = fma(col[(size_t)j * 64], Vs[j], acc);
      const double qn = fma(gamma, acc, er[k]);
      if (r < SA) d2 += (qn - q[k]) * (qn - q[k]);
      q[k] = qn;
    }
    ++it;
  }
  __syncthreads();
  return it;
}

template <int R>
__global__ __launch_bounds__(1024) void anymdp_sampler_big_kernel(SamplerArgs P, BigScratch X) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int S = P.S, A = P.A, SA = S * A, tid = threadIdx.x, BD = blockDim.x;
  const int n_waves = BD >> 6;
  const int ci = blockIdx.x;
  const uint64_t cand = (uint64_t)(P.cand_base + ci);
  const int SAp = X.n_rb * 64;
  double* Tt = X.Tt + (size_t)ci * SAp * S;
  double* Pm = X.Pm + (size_t)ci * 2 * S * S;
  double* Qs = lds;                    // [SAp]
  double* Vs = Qs + SAp;               // [S]
  double* part = Vs + S;               // [16]
  double* pot = part + 16;             // [S]
  double* rpos = pot + S;
  double* npos = rpos + S;
  double* bonus = npos + S;
  double* vstate = bonus + S;
  int* pitm = reinterpret_cast<int*>(vstate + S);
  int* bandlo = pitm + S;
  int* bandhi = bandlo + S;

  // ---- uniform scalars (as the register kernel) ----
  xv_u32x4 w = xs_draw(P, cand, XS_HEAD, 0);
  const double lower = 4.0 * S > 100 ? 4.0 * S : 100;
  double upper = 8.0 * S < 500 ? 8.0 * S : 500;
  if (upper < lower + 1) upper = lower + 1;
  const double max_steps = lower + xv_u53(w.x, w.y) * (upper - lower);
  double w0[3] = {1.0, 0.0, 0.0};
  for (uint32_t r = 0; r < 16; ++r) {
    const xv_u32x4 q4 = xs_draw(P, cand, XS_S0, r);
    double z0, z1, z2, z3;
    xs_normal2(q4.x, q4.y, z0, z1);
    xs_normal2(q4.z, q4.w, z2, z3);
    const double c0 = z0 > 0.0 ? z0 : 0.0, c1 = z1 > 0.0 ? z1 : 0.0, c2 = z2 > 0.0 ? z2 : 0.0;
    if ((c0 + c1) + c2 >= XS_EPS) { w0[0] = c0; w0[1] = c1; w0[2] = c2; break; }
  }
  int s0_id[3], n_s0 = 0;
  double s0_p[3], s0sum = 0.0;
  for (int k = 0; k < 3; ++k)
    if (w0[k] > XS_EPS) { s0_id[n_s0] = k; s0_p[n_s0] = w0[k]; s0sum += w0[k]; ++n_s0; }
  for (int k = n_s0; k < 3; ++k) { s0_id[k] = 0; s0_p[k] = 0.0; }
  for (int k = 0; k < n_s0; ++k) s0_p[k] /= s0sum;
  w = xs_draw(P, cand, XS_PIT, 0);
  double p_pit = -0.20 + 0.60 * xv_u53(w.x, w.y);
  if (p_pit < 0.0) p_pit = 0.0;
  const int goal = xs_u32(w.z) < 0.3 ? 1 : 0;
  {
    int mine = 0;
    for (uint32_t r = 0; r < 64; ++r) {
      mine = 0;
      if (tid < S) {
        const xv_u32x4 q4 = xs_draw(P, cand, XS_PITS, r * 64u + (uint32_t)(tid >> 2));
        mine = xs_u32(xs_word(q4, tid & 3)) < p_pit ? 1 : 0;
      }
      const int cnt = __syncthreads_count(mine);
      if ((double)cnt < (double)S * p_pit + 1.0) break;
    }
    if (tid < S) {
      for (int k = 0; k < n_s0; ++k) if (tid == s0_id[k]) mine = 0;
      if (tid == S - 1) mine = goal;
      pitm[tid] = mine;
    }
  }
  __syncthreads();
  int n_se = 0;
  for (int j = 0; j < S; ++j) n_se += pitm[j];

  // ---- bands ----
  if (tid < S) {
    int first = 0, last = 0;
    if (!pitm[tid]) {
      const int st = tid;
      const int fwd_max = S / 4 + 1 > 2 ? S / 4 + 1 : 2, back_max = S / 2 + 1 > 2 ? S / 2 + 1 : 2;
      const int a_lo = st - back_max > 0 ? st - back_max : 0;
      int a_hi = st - 1 > 0 ? st - 1 : 0;
      if (a_hi < a_lo + 1) a_hi = a_lo + 1;
      const int b_hi = S < st + fwd_max ? S : st + fwd_max;
      int b_lo = S - 1 < st + 1 ? S - 1 : st + 1;
      if (b_lo > b_hi - 1) b_lo = b_hi - 1;
      w = xs_draw(P, cand, XS_BAND, (uint32_t)st);
      first = a_lo + (int)(w.x % (uint32_t)(a_hi - a_lo));
      last = b_lo + (int)(w.y % (uint32_t)(b_hi - b_lo));
      while (last < S) {
        int ahead = 0;
        for (int j = st + 1; j < last; ++j) ahead += !pitm[j];
        if (ahead > 1) break;
        ++last;
      }
    }
    bandlo[tid] = first; bandhi[tid] = last;
  }
  __syncthreads();
  for (int el = tid; el < S * S; el += BD) {
    const int st = el / S, j = el - st * S;
    double v = 0.0;
    if (j >= bandlo[st] && j < bandhi[st]) {
      const xv_u32x4 q4 = xs_draw(P, cand, XS_BANDW, ((uint32_t)st * 8u) * 64u + (uint32_t)(j >> 2));
      v = xs_clip(xs_normal_of(q4, j & 3), 0.10, 1.0);
    }
    Pm[el] = v;
  }
  __syncthreads();
  if (tid < S && !pitm[tid]) {
    Pm[tid * S + tid] = (tid == S - 1) ? 0.0 : Pm[tid * S + tid] / 2.0;
    double tot = 0.0;
    for (int j = bandlo[tid]; j < bandhi[tid]; ++j) tot += Pm[tid * S + j];
    Vs[tid] = tot;
  }
  __syncthreads();

  // ---- the rows of the transition tensor, into the transposed scratch; wave-uniform column ranges of each row group ----
  int wf[R], wl[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int r = k * BD + tid;
    const bool in = r < SA;
    const int s = in ? r / A : 0, a = in ? r - s * A : 0;
    const bool live = in && !pitm[s];
    const int first = live ? bandlo[s] : 0, last = live ? bandhi[s] : 0;
    int lo = live ? first : S, hi = live ? last : 0;
    for (int o = 32; o > 0; o >>= 1) {
      const int l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
      lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    wf[k] = lo < hi ? lo : 0; wl[k] = lo < hi ? hi : 0;
    if (r >= SAp) continue;
    double* col = Tt + ((size_t)(r >> 6) * S) * 64 + (r & 63);
    if (!live) {
      for (int j = 0; j < S; ++j) col[(size_t)j * 64] = 0.0;
      continue;
    }
    const double tot = Vs[s];
    double cen[64];
    for (int b = 0; b < A; ++b) {
      const xv_u32x4 q4 = xs_draw(P, cand, XS_ACT, (uint32_t)s * 16u + (uint32_t)(b >> 2));
      cen[b] = (double)(first - 1) + xs_u32(xs_word(q4, b & 3)) * (double)(last - (first - 1));
    }
    w = xs_draw(P, cand, XS_ACTW, (uint32_t)s);
    const double width = xs_clip(xs_expo(xv_u53(w.x, w.y)), 0.20, 1.6);
    const double inv_w2 = 1.0 / (width * width);
    double rs = 0.0;
    for (int j = 0; j < S; ++j) {
      double v = 0.0;
      if (j >= first && j < last) {
        double colsum = 0.0, dmin = 0.0, e_me = 0.0;
        int amin = 0;
        for (int b = 0; b < A; ++b) {
          const double d = cen[b] - (double)j, d2 = d * d;
          const double e = exp(-d2 * inv_w2);
          colsum += e;
          if (b == a) e_me = e;
          if (b == 0 || d2 < dmin) { dmin = d2; amin = b; }
        }
        if (colsum < XS_EPS) {
          colsum = 0.0;
          for (int b = 0; b < A; ++b) {
            const double d = cen[b] - (double)j;
            colsum += (b == amin) ? 1.0 : exp(-(d * d) * inv_w2);
          }
          if (a == amin) e_me = 1.0;
        }
        v = (e_me / colsum) * (Pm[s * S + j] / tot);
        rs += v;
      }
      col[(size_t)j * 64] = v;
    }
    for (int j = first; j < last; ++j) col[(size_t)j * 64] /= rs;
  }
  __syncthreads();

  // ---- rewards: per-state pieces in LDS (as the register kernel) ----
  double pbase, ub;
  {
    w = xs_draw(P, cand, XS_POS, 0);
    pbase = 0.2 * xs_expo(xv_u53(w.x, w.y));
    ub = xv_u53(w.z, w.w);
  }
  if (tid < S) {
    const int j = tid;
    xv_u32x4 q4 = xs_draw(P, cand, XS_POT, 0);
    const double base = xv_u53(q4.x, q4.y) < 0.5 ? 0.0 : xs_clip(xs_expo(xv_u53(q4.z, q4.w)), 0.20, 5.0);
    q4 = xs_draw(P, cand, XS_POT, 1);
    double box = -base + 2.0 * base * xv_u53(q4.x, q4.y);
    if (box < 0.0) box = 0.0;
    const int n_items = 1 + (int)(q4.z % 3u);
    const double scale = box / sqrt((double)n_items);
    const double x = (double)j / (double)(2 * S);
    double pj = 0.0;
    for (int k = 0; k <= n_items; ++k) {
      double za, zb, order = 0.0;
      q4 = xs_draw(P, cand, XS_POT, 2u + (uint32_t)k);
      xs_normal2(q4.x, q4.y, za, zb);
      const double ca = za * (xs_expo(xs_u32(q4.z)) * scale), cb = zb * (xs_expo(xs_u32(q4.w)) * scale);
      if (k > 0) {
        double zo, zd;
        q4 = xs_draw(P, cand, XS_POT, 8u + (uint32_t)k);
        xs_normal2(q4.y, q4.z, zo, zd);
        order = (double)(1 + (int)(q4.x % 5u)) + zo;
      }
      pj += ca * sin(order * x) + cb * cos(order * x);
    }
    pot[j] = pj;
    q4 = xs_draw(P, cand, XS_POSN, (uint32_t)(j >> 2));
    double pdf = xs_normal_of(q4, j & 3);
    pdf = pdf > 0.0 ? pdf : 0.0;
    if (j == S - 1) pdf += 0.20;
    rpos[j] = pdf * pbase;
    q4 = xs_draw(P, cand, XS_POSU, (uint32_t)(j >> 2));
    const double u = -0.30 + 0.60 * xs_u32(xs_word(q4, j & 3));
    npos[j] = pitm[j] ? 0.0 : (u > 0.0 ? u : 0.0) * pbase;
    bonus[j] = (j == S - 1) ? 1.0 : 0.0;
  }
  __syncthreads();
  double cdf_j = 0.0, c_all = 0.0;
  for (int i = 0; i < S; ++i) {
    c_all += rpos[i];
    if (i == tid) cdf_j = c_all;
  }
  __syncthreads();
  if (tid < S) {
    const double baseline = 0.1 * c_all + ub * (0.9 * c_all - 0.1 * c_all);
    rpos[tid] = pitm[tid] ? 0.0 : cdf_j - baseline;
  }
  double r_step = 0.0, sbase;
  {
    w = xs_draw(P, cand, XS_SA, 0);
    sbase = xs_clip(0.05 * xs_expo(xv_u53(w.x, w.y)), 0.0, 0.10);
    double zs, zd;
    w = xs_draw(P, cand, XS_STEP, 0);
    xs_normal2(w.x, w.y, zs, zd);
    if (goal) r_step = (zs < 0.0 ? zs : 0.0) * 0.01;
    else if (n_se > 0) r_step = (zs > 0.0 ? zs : 0.0) * 0.01;
  }
  double rsa[R], nsa[R], pot_s[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int r = k * BD + tid;
    rsa[k] = 0.0; nsa[k] = 0.0; pot_s[k] = 0.0;
    if (r < SA) {
      xsb_sa_pieces(P, cand, sbase, r, rsa[k], nsa[k]);
      pot_s[k] = pot[r / A];
    }
  }
  __syncthreads();

  // expected reward of this thread's rows under the current bonus: the chain of the register kernel over the row's band
  auto expected = [&](double (&er)[R]) {
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int r = k * BD + tid;
      double e = 0.0;
      if (r < SA) {
        const double* col = Tt + ((size_t)(r >> 6) * S) * 64 + (r & 63);
        for (int j = wf[k]; j < wl[k]; ++j) {
          double rr = (((pot_s[k] - pot[j]) + rpos[j]) + rsa[k]) + r_step;
          rr += bonus[j];
          e = fma(col[(size_t)j * 64], rr, e);
        }
      }
      er[k] = e;
    }
  };

  // ---- terminal rewards repaired against the value function ----
  const int last_live = goal ? S - 2 : S - 1;
  int status = 1, repair_rounds = 0;
  int sweeps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double q[R], er[R];
#pragma unroll
  for (int k = 0; k < R; ++k) q[k] = 0.0;
  for (int tries = 0; tries < 5; ++tries) {
    expected(er);
    const int sw = xsb_value_iteration<R>(Tt, S, A, SA, 0.99, true, er, q, wf, wl, Qs, Vs, part, n_waves);
    sweeps[tries] = sw;
    repair_rounds = tries + 1;
    if (sw >= XS_MAX_SWEEPS) { status = 4; break; }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int r = k * BD + tid;
      if (r < SA) Qs[r] = q[k];
    }
    __syncthreads();
    if (tid < S) {
      double v = Qs[tid * A];
      for (int b = 1; b < A; ++b) v = Qs[tid * A + b] > v ? Qs[tid * A + b] : v;
      vstate[tid] = v;
    }
    __syncthreads();
    double vmin_live = 0.0, vmax_s0 = 0.0, bmin = bonus[0];
    bool first_live = true;
    for (int j = 0; j < S; ++j) {
      const double v = vstate[j];
      if (!pitm[j] && (first_live || v < vmin_live)) { vmin_live = v; first_live = false; }
      if (bonus[j] < bmin) bmin = bonus[j];
    }
    for (int k = 0; k < n_s0; ++k) {
      const double v = vstate[s0_id[k]];
      if (k == 0 || v > vmax_s0) vmax_s0 = v;
    }
    w = xs_draw(P, cand, XS_REPAIR, (uint32_t)tries);
    const double pit_gap = bmin - vmin_live + 1.0;
    const double goal_gap = vmax_s0 - vstate[last_live] + (2.0 + 3.0 * xv_u53(w.x, w.y));
    if (pit_gap <= 0.0 && goal_gap <= 0.0) { status = 0; break; }
    __syncthreads();
    if (tid < S) {
      double bj = bonus[tid];
      if (pit_gap > 0.0 && pitm[tid] && !(goal && tid == S - 1)) bj -= pit_gap + (1.0 + 9.0 * xs_u32(w.z));
      if (goal_gap > 0.0 && tid == S - 1) {
        const double extra = 1.0 + 9.0 * xs_u32(w.w);
        const double lift = 2.0 * goal_gap > extra ? 2.0 * goal_gap : extra;
        bj += goal ? lift : (1.0 - 0.99) * lift;
      }
      bonus[tid] = bj;
    }
    __syncthreads();
  }

  // ---- acceptance ----
  double gini = 0.0, ent = 0.0, gap_min = 0.0;
  if (status == 0) {
    const double g2 = exp2(-1.0 / (double)S);
    expected(er);
    double qo[R], qr[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { qo[k] = 0.0; qr[k] = 0.0; }
    sweeps[5] = xsb_value_iteration<R>(Tt, S, A, SA, g2, true, er, qo, wf, wl, Qs, Vs, part, n_waves);
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int r = k * BD + tid;
      if (r < SA) Qs[r] = qo[k];
    }
    __syncthreads();
    if (tid < S) {
      double v = Qs[tid * A];
      for (int b = 1; b < A; ++b) if (Qs[tid * A + b] > v) v = Qs[tid * A + b];
      vstate[tid] = v;
    }
    __syncthreads();
    // occupancy matrix: row s = the greedy action's row of T (first maximum), or the start distribution for terminal states
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int r = k * BD + tid;
      if (r < SA) {
        const int s = r / A, a = r - s * A;
        int g_s = 0;
        double v = Qs[s * A];
        for (int b = 1; b < A; ++b) if (Qs[s * A + b] > v) { v = Qs[s * A + b]; g_s = b; }
        if (a == g_s) {
          const double* col = Tt + ((size_t)(r >> 6) * S) * 64 + (r & 63);
          for (int j = 0; j < S; ++j) {
            double pj = col[(size_t)j * 64];
            if (pitm[s]) {
              pj = 0.0;
              for (int kk = 0; kk < n_s0; ++kk) if (j == s0_id[kk]) pj = s0_p[kk];
            }
            Pm[s * S + j] = pj;
          }
        }
      }
    }
    double vo_s0[3];
    for (int k = 0; k < 3; ++k) vo_s0[k] = k < n_s0 ? vstate[s0_id[k]] : 0.0;
    __syncthreads();
    sweeps[6] = xsb_value_iteration<R>(Tt, S, A, SA, g2, false, er, qr, wf, wl, Qs, Vs, part, n_waves);
    if (sweeps[5] >= XS_MAX_SWEEPS || sweeps[6] >= XS_MAX_SWEEPS) status = 4;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int r = k * BD + tid;
      if (r < SA) Qs[r] = qr[k];
    }
    __syncthreads();
    const double scale = (1.0 - g2) * max_steps;
    for (int k = 0; k < n_s0; ++k) {
      const int s0 = s0_id[k];
      double vr = Qs[s0 * A];
      for (int b = 1; b < A; ++b) vr = Qs[s0 * A + b] > vr ? Qs[s0 * A + b] : vr;
      const double gap = vo_s0[k] * scale - vr * scale;
      if (k == 0 || gap < gap_min) gap_min = gap;
    }
    if (status == 0 && gap_min < 2.0) status = 2;
    if (status == 0) {
      const int K = (int)log2(max_steps) + 1;
      double* Pa = Pm;
      double* Pb = Pm + (size_t)S * S;
      for (int rep = 0; rep < K; ++rep) {
        __syncthreads();
        for (int e = tid; e < S * S; e += BD) {
          const int i = e / S, j = e - i * S;
          double acc = 0.0;
          for (int k = 0; k < S; ++k) acc = fma(Pa[i * S + k], Pa[k * S + j], acc);
          Pb[e] = acc;
        }
        double* t = Pa; Pa = Pb; Pb = t;
      }
      __syncthreads();
      for (int k = 0; k < n_s0; ++k) {
        const double* row = Pa + s0_id[k] * S;
        double s2 = 0.0, h = 0.0;
        for (int j = 0; j < S; ++j) {
          const double p = row[j] + 1.0e-12;
          s2 += p * p;
          h += p * log(p);
        }
        const double gk = 1.0 - s2, ek = -h / log((double)S);
        if (k == 0 || gk < gini) gini = gk;
        if (k == 0 || ek < ent) ent = ek;
      }
      if (!(gini > 0.70 && ent > 0.35)) status = 3;
    }
  }

  // ---- outputs ----
  if (tid == 0) {
    P.status[ci] = status;
    if (P.info) {
      xv_anymdp_cand_info& o = P.info[ci];
      o.status = status; o.goal = goal; o.n_s0 = n_s0; o.repair_rounds = repair_rounds;
      for (int k = 0; k < 4; ++k) { o.s0[k] = k < n_s0 ? s0_id[k] : 0; o.s0_prob[k] = k < n_s0 ? s0_p[k] : 0.0; }
      for (int k = 0; k < 8; ++k) o.sweeps[k] = sweeps[k];
      o.max_steps = max_steps; o.gini = gini; o.ent = ent; o.gap_min = gap_min;
    }
  }
  if (P.info && tid < S) {
    P.info[ci].band_lo[tid] = bandlo[tid];
    P.info[ci].band_hi[tid] = bandhi[tid];
    P.info[ci].s_e[tid] = (uint8_t)pitm[tid];
  }
  if (P.transition || P.reward || P.noise) {
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int r = k * BD + tid;
      if (r < SA) {
        const size_t o = ((size_t)ci * SA + r) * S;
        const double* col = Tt + ((size_t)(r >> 6) * S) * 64 + (r & 63);
        for (int j = 0; j < S; ++j) {
          if (P.transition) P.transition[o + j] = col[(size_t)j * 64];
          if (P.reward) {
            const double rr = (((pot_s[k] - pot[j]) + rpos[j]) + rsa[k]) + r_step;
            P.reward[o + j] = rr + bonus[j];
          }
          if (P.noise) P.noise[o + j] = npos[j] + nsa[k];
        }
      }
    }
  }
  if (tid == 0 && (P.state_map || P.info)) {
    int* sm = reinterpret_cast<int*>(Qs);
    for (int i = 0; i < S; ++i) sm[i] = i;
    for (int i = S - 1; i >= 1; --i) {
      const xv_u32x4 q4 = xs_draw(P, cand, XS_PERM, (uint32_t)(i >> 2));
      const int j = (int)(xs_word(q4, i & 3) % (uint32_t)(i + 1));
      const int t = sm[i]; sm[i] = sm[j]; sm[j] = t;
    }
    for (int i = 0; i < S; ++i) {
      if (P.info) P.info[ci].state_map[i] = sm[i];
      if (P.state_map && status == 0) P.state_map[(size_t)ci * S + i] = sm[i];
    }
  }
  if (status != 0 || P.rows == nullptr) return;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int r = k * BD + tid;
    if (r >= SA) continue;
    const double* col = Tt + ((size_t)(r >> 6) * S) * 64 + (r & 63);
    double* rec = P.rows + ((size_t)ci * SA + r) * (size_t)P.row_lines * 16;
    for (int e = 0; e < P.row_lines * 16; ++e) rec[e] = 0.0;
    double tot = 0.0;
    for (int j = 0; j < S; ++j) tot += col[(size_t)j * 64];
    const bool zero_row = tot == 0.0;
    double c = 0.0;
    const int n_ent = (P.row_lines - 1) * 7;
    for (int jj = 0; jj < S; ++jj) {
      c += col[(size_t)jj * 64];
      const double cdf = zero_row ? 1.0 : c / tot;
      const double rr = ((((pot_s[k] - pot[jj]) + rpos[jj]) + rsa[k]) + r_step) + bonus[jj];
      const float rf = (float)rr, nf = (float)(npos[jj] + nsa[k]);
      double* ent2 = rec + 16 * (1 + jj / 7) + 2 * (jj % 7);
      ent2[0] = cdf;
      ent2[1] = __hiloint2double((int)__float_as_uint(nf), (int)__float_as_uint(rf));
    }
    for (int j = S; j < n_ent; ++j) rec[16 * (1 + j / 7) + 2 * (j % 7)] = 2.0;
  }
  if (tid < (S + 63) / 64) {
    uint64_t m = 0;
    for (int j = 0; j < 64 && 64 * tid + j < S; ++j) if (pitm[64 * tid + j]) m |= 1ull << j;
    P.term_mask[(size_t)ci * ((S + 63) / 64) + tid] = m;
  }
  if (tid == 0) {
    double c = 0.0, ctot = 0.0;
    for (int k = 0; k < n_s0; ++k) ctot += s0_p[k];
    for (int k = 0; k < P.s0_max; ++k) {
      if (k < n_s0) {
        c += s0_p[k];
        P.s0_cdf[(size_t)ci * P.s0_max + k] = c / ctot;
        P.s0_ids[(size_t)ci * P.s0_max + k] = s0_id[k];
      } else {
        P.s0_cdf[(size_t)ci * P.s0_max + k] = 1.0;
        P.s0_ids[(size_t)ci * P.s0_max + k] = s0_id[n_s0 - 1];
      }
    }
    P.max_steps[ci] = (int32_t)ceil(max_steps);
  }
}

}  // namespace

extern "C" int xv_anymdp_sample_tasks(xv_engine* e, uint64_t seed, int64_t cand_base, int n_cand, int S, int A, int s0_max,
                                      void* rows, int32_t* state_map, uint64_t* term_mask, double* s0_cdf,
                                      int32_t* s0_ids, int32_t* max_steps, double* transition, double* reward,
                                      double* reward_noise, xv_anymdp_cand_info* info, int32_t* status) {
  XV_CHECK_ARG(e != nullptr && status != nullptr);
  XV_CHECK_ARG(n_cand > 0 && cand_base >= 0);
  const bool small = S >= 8 && S <= 64 && A >= 2 && A <= 64 && S * A <= 512;
  const bool big = !small && S > 16 && S <= 256 && A >= 2 && A <= 64 && S * A <= 4096;
  if (!small && !big) {
    xv_set_error("xv_anymdp_sample_tasks: the device sampler covers 8 <= S <= 64 with S * A <= 512 and S <= 256 with S * A <= 4096 "
                 "(got S = %d, A = %d); larger tasks: the seeded host sampler", S, A);
    return XV_ERR_UNSUPPORTED;
  }
  XV_CHECK_ARG(s0_max >= 3 && s0_max <= 256);
  if (rows != nullptr) XV_CHECK_ARG(state_map && term_mask && s0_cdf && s0_ids && max_steps);
  XV_HIP(hipSetDevice(e->device));
  SamplerArgs P;
  P.seed = seed; P.cand_base = cand_base; P.n_cand = n_cand; P.S = S; P.A = A; P.s0_max = s0_max;
  P.row_lines = XV_ANYMDP_ROW_LINES(S);
  P.rows = static_cast<double*>(rows); P.state_map = state_map; P.term_mask = term_mask; P.s0_cdf = s0_cdf;
  P.s0_ids = s0_ids; P.max_steps = max_steps; P.transition = transition; P.reward = reward; P.noise = reward_noise;
  P.info = info; P.status = status;
  if (big) {
    // rows in a transposed global scratch (stream-ordered allocation: no synchronisation), R rows per thread
    const int SA = S * A, n_rb = (SA + 63) / 64, R = (SA + 1023) / 1024;
    const int threads = 64 * ((n_rb + R - 1) / R);
    BigScratch X;
    X.n_rb = n_rb; X.Tt = nullptr; X.Pm = nullptr;
    const size_t tt_bytes = sizeof(double) * (size_t)n_cand * n_rb * 64 * S, pm_bytes = sizeof(double) * (size_t)n_cand * 2 * S * S;
    hipError_t r = hipMallocAsync((void**)&X.Tt, tt_bytes, e->stream);
    if (r == hipSuccess) r = hipMallocAsync((void**)&X.Pm, pm_bytes, e->stream);
    if (r != hipSuccess) {
      (void)hipGetLastError();
      if (X.Tt) (void)hipFreeAsync(X.Tt, e->stream);
      xv_set_error("xv_anymdp_sample_tasks: cannot allocate %.1f GiB of candidate scratch (lower n_cand)",
                   (double)(tt_bytes + pm_bytes) / (double)(1ull << 30));
      return XV_ERR_NOMEM;
    }
    const size_t lds = sizeof(double) * ((size_t)n_rb * 64 + 6 * (size_t)S + 16) + sizeof(int) * 3 * (size_t)S;
    hipError_t attr = hipSuccess;      // a failed attribute call must not leave with the scratch allocated
#define XSB_LAUNCH(R_)                                                                                              \
  do {                                                                                                              \
    if (lds > 48 * 1024)                                                                                            \
      attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&anymdp_sampler_big_kernel<R_>),                     \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
    if (attr == hipSuccess)                                                                                         \
      hipLaunchKernelGGL(anymdp_sampler_big_kernel<R_>, dim3(n_cand), dim3(threads), lds, e->stream, P, X);         \
  } while (0)
    if (R == 1) XSB_LAUNCH(1);
    else if (R == 2) XSB_LAUNCH(2);
    else if (R == 3) XSB_LAUNCH(3);
    else XSB_LAUNCH(4);
#undef XSB_LAUNCH
    const hipError_t le = attr != hipSuccess ? attr : hipGetLastError();
    (void)hipFreeAsync(X.Tt, e->stream);
    (void)hipFreeAsync(X.Pm, e->stream);
    if (le != hipSuccess) {
      xv_set_error("xv_anymdp_sample_tasks: kernel launch failed: %s", hipGetErrorString(le));
      return XV_ERR_HIP;
    }
    return XV_OK;
  }
  const int threads = (S * A + 63) / 64 * 64;
  const int SP = S <= 16 ? 16 : (S <= 32 ? 32 : 64);
  const size_t lds = sizeof(double) * ((size_t)S * A + 6 * SP + 8 + 2 * (size_t)SP * SP) + sizeof(int) * 3 * SP;
#define XS_LAUNCH(SP_)                                                                                            \
  do {                                                                                                            \
    if (lds > 48 * 1024)                                                                                          \
      XV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&anymdp_sampler_kernel<SP_>),                     \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
    hipLaunchKernelGGL(anymdp_sampler_kernel<SP_>, dim3(n_cand), dim3(threads), lds, e->stream, P);               \
  } while (0)
  if (SP == 16) XS_LAUNCH(16);
  else if (SP == 32) XS_LAUNCH(32);
  else XS_LAUNCH(64);
#undef XS_LAUNCH
  XV_LAUNCH_CHECK();
  return XV_OK;
}

// ------------------------------------------------------------------------------------------------
// Observation models of AnyPOMDPTaskSampler / MultiTokensAnyPOMDPTaskSampler (task_sampler.py:78-87, :103-117) for whole
// task batches: per (task, observation token) a matrix obs[S][n_obs] = scipy.sparse.random(S, n_obs, density) — exactly
// k = round(density * S * n_obs) cells, uniformly chosen without replacement over the WHOLE matrix, values U[0, 1) —
// where a row left empty gets a 1 in a random column, every row normalised; density = min(density,
// maximum_distribution / n_obs).  Emitted as the step engine's table: the inclusive row CDF cumsum(row) / cumsum(row)[-1]
// (what numpy.random.choice forms, anymdp_env.py:150-157), obs_cdf[task][token][s][:].
// Same distribution and rules as the reference, its own stream: every cell c of matrix m draws Philox(counter = {c, m,
// purpose}, key = seed); the k chosen cells are the k smallest 64-bit keys {random high bits | cell index}, found by a
// 64-step bisection of the key threshold (no sort, no storage: keys are recomputed).  One workgroup per matrix.
// Oracle: xo_anymdp_sample_observation_model (same draws, bit-identical tables).
// ------------------------------------------------------------------------------------------------
#define XS_OBS_KEY 0x40u
#define XS_OBS_FIX 0x41u
__device__ __forceinline__ uint64_t xs_obs_key(uint64_t seed, uint64_t mat, uint32_t cell, uint64_t idx_mask, double* val) {
  const xv_u32x4 w = xv_philox4x32_10(cell, (uint32_t)mat, (uint32_t)(mat >> 32), XS_OBS_KEY, (uint32_t)seed, (uint32_t)(seed >> 32));
  if (val) *val = xv_u53(w.z, w.w);
  return ((((uint64_t)w.x << 32) | (uint64_t)w.y) & ~idx_mask) | (uint64_t)cell;
}

__global__ __launch_bounds__(256) void anymdp_obs_model_kernel(uint64_t seed, uint64_t mat_base, int S, int n_obs, long long k_cells,
                                                               double* obs_cdf) {
  __shared__ unsigned long long s_cnt;
  __shared__ unsigned long long s_thr;
  const uint64_t mat = mat_base + blockIdx.x;
  const int M = S * n_obs, tid = threadIdx.x;
  uint64_t idx_mask = 1;
  while (idx_mask < (uint64_t)M) idx_mask <<= 1;
  idx_mask -= 1;
  // smallest threshold T with #{key <= T} >= k  (keys are distinct: their low bits are the cell index)
  uint64_t lo = 0, hi = ~0ull;
  if (k_cells > 0) {
    for (int it = 0; it < 64; ++it) {
      const uint64_t mid = lo + ((hi - lo) >> 1);
      if (tid == 0) s_cnt = 0ull;
      __syncthreads();
      unsigned long long c = 0;
      for (int cell = tid; cell < M; cell += blockDim.x) c += xs_obs_key(seed, mat, (uint32_t)cell, idx_mask, nullptr) <= mid ? 1ull : 0ull;
      atomicAdd(&s_cnt, c);
      __syncthreads();
      const bool enough = s_cnt >= (unsigned long long)k_cells;
      __syncthreads();
      if (enough) hi = mid; else lo = mid + 1;
      if (lo >= hi) break;
    }
  }
  if (tid == 0) s_thr = hi;
  __syncthreads();
  const uint64_t thr = s_thr;
  double* out = obs_cdf + (size_t)blockIdx.x * (size_t)M;
  for (int row = tid; row < S; row += blockDim.x) {   // one thread per row: values, empty-row fix, cumsum, normalisation
    double* o = out + (size_t)row * n_obs;
    double acc = 0.0;
    for (int j = 0; j < n_obs; ++j) {
      double v;
      const uint64_t key = xs_obs_key(seed, mat, (uint32_t)(row * n_obs + j), idx_mask, &v);
      acc += (k_cells > 0 && key <= thr) ? v : 0.0;
      o[j] = acc;
    }
    if (acc == 0.0) {   // obs_mat[i][random.randint(observation_space)] = 1
      const xv_u32x4 w = xv_philox4x32_10((uint32_t)row, (uint32_t)mat, (uint32_t)(mat >> 32), XS_OBS_FIX, (uint32_t)seed,
                                          (uint32_t)(seed >> 32));
      const int col = (int)(((uint64_t)w.x * (uint64_t)(uint32_t)n_obs) >> 32);
      for (int j = 0; j < n_obs; ++j) o[j] = j >= col ? 1.0 : 0.0;
      acc = 1.0;
    }
    for (int j = 0; j < n_obs; ++j) o[j] = o[j] / acc;
  }
}

extern "C" int xv_anymdp_sample_observation_model(xv_engine* e, uint64_t seed, int64_t task_base, int n_task, int S, int n_obs,
                                                  int d_obs, double density, double maximum_distribution, double* obs_cdf) {
  XV_CHECK_ARG(e != nullptr && obs_cdf != nullptr);
  XV_CHECK_ARG(n_task > 0 && task_base >= 0 && S >= 1 && S <= 512 && n_obs >= 1 && n_obs <= 65536 && d_obs >= 1 && d_obs <= 64);
  XV_CHECK_ARG(density >= 0.0 && density <= 1.0 && maximum_distribution > 0.0);
  XV_CHECK_ARG((long long)S * n_obs < (1ll << 31));
  XV_HIP(hipSetDevice(e->device));
  const double d = density < maximum_distribution / (double)n_obs ? density : maximum_distribution / (double)n_obs;
  const long long k = (long long)rint(d * (double)S * (double)n_obs);     // scipy.sparse.random: int(round(density * m * n))
  const size_t n_mat = (size_t)n_task * d_obs;
  const size_t chunk = 1u << 20;
  for (size_t m0 = 0; m0 < n_mat; m0 += chunk) {
    const size_t nm = n_mat - m0 < chunk ? n_mat - m0 : chunk;
    hipLaunchKernelGGL(anymdp_obs_model_kernel, dim3((unsigned)nm), dim3(256), 0, e->stream, seed,
                       (uint64_t)task_base * (uint64_t)d_obs + m0, S, n_obs, k, obs_cdf + m0 * (size_t)S * n_obs);
  }
  XV_LAUNCH_CHECK();
  return XV_OK;
}
