// anymdp.hip — AnyMDP batched step / reset / rollout kernels for gfx950 and their C-ABI.
//
// Reproduces xenoverse/anymdp/anymdp_env.py: reset :81-90, single_step :92-110, step :112-132,
// get_observation :145-159 (MDP branch), per env, for N envs per launch.  One wavefront lane owns one env.
//
// A step is a chain of DEPENDENT memory round trips (state -> which row -> which next state -> its reward
// and observation); at 65,536 envs per launch the kernel lives or dies by the length of that chain and by
// the bytes each link moves.  Layout and kernel are built to make it three links:
//
//   1. per-env words (state, steps, action, task id): struct-of-arrays, coalesced dword streams.
//   2. per row (t,s,a) a 32-B FENCE record = the last CDF entry of blocks 0..2 of the row, and per task a
//      128-B HEADER (terminal mask, max_steps, the s_0 distribution with its observation ids, and the 64
//      observation ids as bytes).  k = #{fences <= u} names the one 16-entry block that contains s'
//      (the CDF is non-decreasing, so blocks < k are entirely <= u and blocks > k entirely > u).
//   3. that ONE 256-B block: 16 fp64 CDF entries + the 16 {reward, noise} pairs of the same next states
//      (include/xeno.h, "rows").  It is read by 16 lanes, coalesced — a wave fetches the blocks of 4 envs per
//      load instruction and keeps 16 such loads in flight — and searched with one v_cmp_le_f64 against the
//      env's uniform, a 64-bit ballot and a 16-bit popcount:  s' = 16k + popcount(cdf[j] <= u), which is
//      numpy.searchsorted(cdf, u, 'right').  The reward pair is taken from the same registers with a
//      ds_bpermute; the observation id is a byte of the header.  No dependent gather follows.
//
// Per env-step that is 32 + 256 + 128 B of table reads instead of the 512-B row + 4 gathers of a flat layout.
// BINARY mode (any S <= 256, any s0 table) is the general per-lane fallback.
#include "philox.h"
#include "xv_common.h"

#include <cstddef>

// One 128-byte line per task (fast path: S <= 64, s0_max <= 4, observation ids < 256); built at create time.
struct __attribute__((aligned(128))) AnyMDPHdr {
  uint64_t term_mask;    // bit s set <=> s terminal
  int32_t max_steps;
  uint32_t s0_ids;       // 4 x u8 inner-state ids of s_0 (padded with the last)
  uint32_t s0_obs;       // 4 x u8 observation ids of those states
  uint32_t pad0[3];
  double s0_cdf[4];      // bytes 32..63: inclusive CDF of s_0_prob padded with 1.0
  uint32_t obs[16];      // 64 x u8: observation id of inner state s (state_mapping)
};
static_assert(sizeof(AnyMDPHdr) == 128, "header must be one 128-byte line");
static_assert(offsetof(AnyMDPHdr, s0_cdf) == 32 && offsetof(AnyMDPHdr, obs) == 64, "register view below");

struct AnyMDPArgs {
  const AnyMDPHdr* hdr;  // engine-owned, nullptr when the fast path does not apply
  const double* fence;   // engine-owned [n_rows][4] (S <= 64), nullptr otherwise
  // borrowed task tables
  const double* rows;    // blocked rows, addressed in 8-byte units: block = 32 units
  const int32_t* state_map;
  const uint64_t* term_mask;
  const double* s0_cdf;
  const int32_t* s0_ids;
  const int32_t* max_steps;
  const int32_t* env_task;
  // engine-owned env state
  int32_t* state;
  int32_t* steps;
  uint8_t* need_reset;
  uint32_t* err;
  int n_env, n_task, S, A, s0_max, words, NB;
  uint64_t seed, gid_base, tick;
};

struct AnyMDPStepIO {
  const int32_t* action;
  const double* u;        // injected draws (INJECT only)
  const float* z;
  const double* u_reset;
  int32_t* obs;
  float* reward;
  float* reward_gt;
  uint8_t* terminated;
  uint8_t* truncated;
  int32_t* final_obs;     // nullable
  // teacher rollout (nullable): action = greedy[task][state] w.p. 1-epsilon, else uniform; written to action_out
  const uint8_t* greedy;
  int32_t* action_out;
  float epsilon;
};

struct xv_anymdp {
  xv_engine* eng;
  AnyMDPArgs a;
  int search;  // XV_ANYMDP_SEARCH_*
  const double* obs_cdf;   // observation model (POMDP / MTPOMDP), nullptr for MDP
  int n_obs, d_obs, d_act;
};

__device__ __forceinline__ bool anymdp_is_term(const AnyMDPArgs& P, int t, uint64_t tm0, int s) {
  if (P.words == 1) return (tm0 >> s) & 1ull;
  return (P.term_mask[(size_t)t * P.words + (s >> 6)] >> (s & 63)) & 1ull;
}

// s = s0_ids[upper_bound(s0_cdf, u)]   (anymdp_env.py:89: numpy.random.choice(self.s_0, p=self.s_0_prob))
__device__ __forceinline__ int anymdp_draw_s0(const AnyMDPArgs& P, int t, double u) {
  const double* c = P.s0_cdf + (size_t)t * P.s0_max;
  int k = 0;
  while (k < P.s0_max - 1 && c[k] <= u) ++k;
  return P.s0_ids[(size_t)t * P.s0_max + k];
}

// blocked-row accessors: CDF entry j / reward pair j of row r
__device__ __forceinline__ const double* anymdp_cdf_ptr(const AnyMDPArgs& P, uint32_t r, int j) {
  return P.rows + ((size_t)r * P.NB + (j >> 4)) * 32 + (j & 15);
}
__device__ __forceinline__ float2 anymdp_rs(const AnyMDPArgs& P, uint32_t r, int j) {
  return reinterpret_cast<const float2*>(P.rows + ((size_t)r * P.NB + (j >> 4)) * 32 + 16)[j & 15];
}

__device__ __forceinline__ double xv_shfl_f64(double v, int src) {
  return __hiloint2double(__shfl(__double2hiint(v), src), __shfl(__double2loint(v), src));
}

// The header is held in named registers, never as an indexable aggregate: hipcc folds a select chain over
// loaded values back into ONE load at a selected address (a dynamically indexed stack array in scratch, or a
// dependent global load) — exactly the round trip this header exists to remove.  The empty asm statements
// make each observation word an opaque register value, which keeps the 15-select tree in VALU.
struct AnyMDPHdrRegs {
  uint4 q0, q1, q2, q3;   // term_mask | max_steps, s0_ids, s0_obs, pad | s0_cdf[0..3]
  uint32_t w0, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11, w12, w13, w14, w15;   // 64 obs ids, 1 byte each
};
#define XV_OPAQUE(x) asm volatile("" : "+v"(x))
__device__ __forceinline__ AnyMDPHdrRegs anymdp_load_hdr(const AnyMDPHdr* hdr, int t) {
  const uint4* p = reinterpret_cast<const uint4*>(hdr + t);
  AnyMDPHdrRegs r;
  r.q0 = p[0]; r.q1 = p[1]; r.q2 = p[2]; r.q3 = p[3];
  const uint4 a = p[4], b = p[5], c = p[6], d = p[7];
  r.w0 = a.x; r.w1 = a.y; r.w2 = a.z; r.w3 = a.w; r.w4 = b.x; r.w5 = b.y; r.w6 = b.z; r.w7 = b.w;
  r.w8 = c.x; r.w9 = c.y; r.w10 = c.z; r.w11 = c.w; r.w12 = d.x; r.w13 = d.y; r.w14 = d.z; r.w15 = d.w;
  XV_OPAQUE(r.w0); XV_OPAQUE(r.w1); XV_OPAQUE(r.w2); XV_OPAQUE(r.w3);
  XV_OPAQUE(r.w4); XV_OPAQUE(r.w5); XV_OPAQUE(r.w6); XV_OPAQUE(r.w7);
  XV_OPAQUE(r.w8); XV_OPAQUE(r.w9); XV_OPAQUE(r.w10); XV_OPAQUE(r.w11);
  XV_OPAQUE(r.w12); XV_OPAQUE(r.w13); XV_OPAQUE(r.w14); XV_OPAQUE(r.w15);
  return r;
}
// observation id of inner state idx (0..63): byte idx of w0..w15, by selects only
__device__ __forceinline__ uint32_t anymdp_hdr_obs(const AnyMDPHdrRegs& H, int idx) {
  const bool b0 = idx & 4, b1 = idx & 8, b2 = idx & 16, b3 = idx & 32;   // bits of the word index idx>>2
  const uint32_t a0 = b0 ? H.w1 : H.w0, a1 = b0 ? H.w3 : H.w2, a2 = b0 ? H.w5 : H.w4, a3 = b0 ? H.w7 : H.w6;
  const uint32_t a4 = b0 ? H.w9 : H.w8, a5 = b0 ? H.w11 : H.w10, a6 = b0 ? H.w13 : H.w12, a7 = b0 ? H.w15 : H.w14;
  const uint32_t c0 = b1 ? a1 : a0, c1 = b1 ? a3 : a2, c2 = b1 ? a5 : a4, c3 = b1 ? a7 : a6;
  const uint32_t d0 = b2 ? c1 : c0, d1 = b2 ? c3 : c2;
  const uint32_t v = b3 ? d1 : d0;
  return (v >> (8 * (idx & 3))) & 0xFFu;
}
__device__ __forceinline__ double xv_u2d(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }

enum { SEARCH_BINARY = 0, SEARCH_FENCE = 1 };

// T_steps == 1: one vector step.  T_steps > 1: fused rollout, io arrays are [T][n_env], mode SAME_STEP.
// HDR: per-task scalars and observation ids come from the packed 128-B header.
template <bool INJECT, int SEARCH, bool HDR, bool ROLLOUT>
__global__ __launch_bounds__(256) void anymdp_step_kernel(AnyMDPArgs P, AnyMDPStepIO io, int T_steps,
                                                          int mode) {
  static_assert(SEARCH != SEARCH_FENCE || HDR, "the fence path needs the header");
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < P.n_env;
  const int ic = valid ? i : P.n_env - 1;
  const int lane = threadIdx.x & 63;
  const int S = P.S, A = P.A;

  // ---- link 1: per-env words (coalesced) ----
  const int t = P.env_task[ic];
  int s = P.state[ic];
  int steps = P.steps[ic];
  int nr = P.need_reset[ic];
  int a_next = io.action ? io.action[ic] : 0;
  const uint64_t gid = P.gid_base + (uint64_t)ic;
  uint32_t err = 0;

  AnyMDPHdrRegs H;
  int max_steps = 0;
  uint64_t tm0 = 0;
  bool hdr_loaded = false;

  const int T = ROLLOUT ? T_steps : 1;   // single step: straight-line code, counted vmcnt waits
  for (int ts = 0; ts < T; ++ts) {
    const size_t o = (size_t)ts * P.n_env + ic;

    // random inputs first: pure ALU, overlaps the latency of link 1
    double u, u_reset;
    float z;
    if (INJECT) {
      u = io.u[o];
      z = io.z[o];
      u_reset = io.u_reset[o];
    } else {
      const xv_u32x4 w = xv_env_draw(P.seed, gid, P.tick + (uint64_t)ts, XV_DRAW_STEP);
      u = xv_u53(w.x, w.y);
      z = xv_normal1(w.z, w.w);
      const xv_u32x4 v = xv_env_draw(P.seed, gid, P.tick + (uint64_t)ts, XV_DRAW_RESET);
      u_reset = xv_u53(v.x, v.y);
    }

    int a = a_next;
    if (io.greedy) {   // teacher policy: argmax_a Q[inner_state] (anymdp_solver_opt.py:38-51), epsilon-greedy
      a = io.greedy[(size_t)t * S + s];
      if (io.epsilon > 0.0f) {
        const xv_u32x4 e = xv_env_draw(P.seed, gid, P.tick + (uint64_t)ts, 2u);
        if ((float)(e.x >> 8) * (1.0f / 16777216.0f) < io.epsilon) a = (int)(e.y % (uint32_t)A);
      }
      if (valid) io.action_out[o] = a;
    }
    if (a < 0 || a >= A) {  // reference: assert action < self.na (:97)
      if (!(mode == XV_AUTORESET_NEXT_STEP && nr)) err |= XV_DEVERR_ACTION_RANGE;
      a = a < 0 ? 0 : A - 1;
    }
    const uint32_t rowidx = ((uint32_t)t * S + s) * A + a;

    // ---- link 2: fence record of the row (+ the task header, once) ----
    double f0 = 0, f1 = 0, f2 = 0;
    if (SEARCH == SEARCH_FENCE) {
      const double* f = P.fence + (size_t)rowidx * 4;
      f0 = f[0]; f1 = f[1]; f2 = f[2];
    }
    if (!hdr_loaded) {
      if (HDR) {
        H = anymdp_load_hdr(P.hdr, t);
        max_steps = (int)H.q0.z;
        tm0 = (uint64_t)H.q0.x | ((uint64_t)H.q0.y << 32);
      } else {
        max_steps = P.max_steps[t];
        tm0 = P.term_mask[(size_t)t * P.words];
      }
      hdr_loaded = true;
    }

    // ---- link 3: s' = upper_bound(cdf[s,a,:], u)   (:99-100, numpy.random.choice) and its reward pair ----
    int s2;
    float2 rsv;
    if (SEARCH == SEARCH_FENCE) {
      const int k = (int)(f0 <= u) + (int)(f1 <= u) + (int)(f2 <= u);   // fences of absent blocks hold 2.0
      const uint32_t bidx = rowidx * (uint32_t)P.NB + (uint32_t)k;
      const int g = lane >> 4, j = lane & 15;
      // (a) all block addresses first (16 independent ds_bpermute), then all 32 loads back to back
      uint32_t bi[16];
#pragma unroll
      for (int it = 0; it < 16; ++it) bi[it] = (uint32_t)__shfl((int)bidx, it * 4 + g);
      double cv[16];
      float2 rv[16];
#pragma unroll
      for (int it = 0; it < 16; ++it) {   // lanes 16g..16g+15 read the block of env 4*it+g
        const double* blk = P.rows + (size_t)bi[it] * 32;
        cv[it] = blk[j];
        rv[it] = reinterpret_cast<const float2*>(blk + 16)[j];
      }
      if (ROLLOUT && io.action && ts + 1 < T) a_next = io.action[o + P.n_env];   // prefetch behind the blocks
      __builtin_amdgcn_sched_barrier(0);
      // (b) while the loads fly: broadcast each env's uniform to the 16 lanes that hold its block
      double ue[16];
#pragma unroll
      for (int it = 0; it < 16; ++it) ue[it] = xv_shfl_f64(u, it * 4 + g);
      __builtin_amdgcn_sched_barrier(0);
      // (c) compare + ballot + popcount; the owner of group q in iteration `it` is lane 4*it+q
      int cnt_own = 0;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const unsigned long long m = __ballot(cv[it] <= ue[it]);
        const int cnt = __popc((unsigned)(m >> (16 * (lane & 3))) & 0xFFFFu);
        if ((lane >> 2) == it) cnt_own = cnt;
      }
      // (d) the reward pair of s' sits in lane 16q + cnt of the same registers: 32 independent bpermutes
      const int pick = 16 * (lane & 3) + (cnt_own < 15 ? cnt_own : 15);
      float rx = 0.0f, ry = 0.0f;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const float px = __shfl(rv[it].x, pick);
        const float py = __shfl(rv[it].y, pick);
        if ((lane >> 2) == it) {
          rx = px;
          ry = py;
        }
      }
      s2 = 16 * k + cnt_own;
      s2 = s2 < S - 1 ? s2 : S - 1;
      rsv = make_float2(rx, ry);
    } else {
      int lo = 0, n = S;
      while (n > 0) {
        const int half = n >> 1;
        if (*anymdp_cdf_ptr(P, rowidx, lo + half) <= u) {
          lo += half + 1;
          n -= half + 1;
        } else {
          n = half;
        }
      }
      s2 = lo < S - 1 ? lo : S - 1;
      if (ROLLOUT && io.action && ts + 1 < T) a_next = io.action[o + P.n_env];
      rsv = anymdp_rs(P, rowidx, s2);                              // :103-104
    }
    const int obs2 = HDR ? (int)anymdp_hdr_obs(H, s2) : P.state_map[(size_t)t * S + s2];   // :146-148
    const bool term2 = anymdp_is_term(P, t, tm0, s2);              // :107-108

    int o_obs, o_fobs = -1;
    float o_r, o_rgt;
    bool o_term, o_trunc;
    bool do_reset = false;
    if (mode == XV_AUTORESET_NEXT_STEP && nr) {
      // the call after a done ignores the action and returns the reset observation
      do_reset = true;
      o_r = 0.0f; o_rgt = 0.0f; o_term = false; o_trunc = false; o_obs = 0;
    } else if (mode == XV_AUTORESET_DISABLED && anymdp_is_term(P, t, tm0, s)) {
      // reference raises "given an terminated state" (:95-96): env untouched, error bit set
      err |= XV_DEVERR_STEP_TERMINAL;
      o_obs = HDR ? (int)anymdp_hdr_obs(H, s) : P.state_map[(size_t)t * S + s];
      o_r = 0.0f; o_rgt = 0.0f; o_term = true; o_trunc = steps >= max_steps;
    } else {
      steps += 1;                                              // :113
      o_trunc = steps >= max_steps;                            // :114
      o_rgt = rsv.x;
      o_r = fmaf(rsv.y, z, rsv.x);                             // :105 normal(mu, sigma) = mu + sigma*z
      o_term = term2;
      s = s2;
      o_obs = obs2;
      if (o_term || o_trunc) {
        if (mode == XV_AUTORESET_SAME_STEP) {
          o_fobs = obs2;
          do_reset = true;
        } else if (mode == XV_AUTORESET_NEXT_STEP) {
          nr = 1;
        }
      }
    }
    if (do_reset) {                                            // reset(): :85-90
      if (HDR) {
        // upper_bound over the 4 padded CDF entries; ids and obs ids come packed in the header
        const int k0 = (int)(xv_u2d(H.q2.x, H.q2.y) <= u_reset) + (int)(xv_u2d(H.q2.z, H.q2.w) <= u_reset) +
                       (int)(xv_u2d(H.q3.x, H.q3.y) <= u_reset);
        s = (int)((H.q0.w >> (8 * k0)) & 0xFFu);
        o_obs = (int)((H.q1.x >> (8 * k0)) & 0xFFu);
      } else {
        s = anymdp_draw_s0(P, t, u_reset);
        o_obs = P.state_map[(size_t)t * S + s];
      }
      steps = 0;
      nr = 0;
    }
    if (valid) {
      io.obs[o] = o_obs;
      io.reward[o] = o_r;
      io.reward_gt[o] = o_rgt;
      io.terminated[o] = o_term ? 1 : 0;
      io.truncated[o] = o_trunc ? 1 : 0;
      if (io.final_obs) io.final_obs[o] = o_fobs;
    }
  }
  if (valid) {
    P.state[i] = s;
    P.steps[i] = steps;
    P.need_reset[i] = (uint8_t)nr;
  }
  if (err) atomicOr(P.err, err);
}

// fence[r][k] = last CDF entry of block k of row r for k < NB-1, else 2.0 (never <= u)
__global__ __launch_bounds__(256) void anymdp_build_fence_kernel(const double* rows, double* fence,
                                                                 size_t n_rows, int NB) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * 4) return;
  const size_t r = idx >> 2;
  const int k = (int)(idx & 3);
  fence[idx] = (k < NB - 1) ? rows[(r * NB + k) * 32 + 15] : 2.0;
}

// packs the per-task scalars and observation ids into 128-byte headers (once, at create time)
__global__ __launch_bounds__(256) void anymdp_pack_hdr_kernel(AnyMDPArgs P, AnyMDPHdr* hdr) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P.n_task) return;
  AnyMDPHdr h;
  h.term_mask = P.term_mask[t];
  h.max_steps = P.max_steps[t];
  uint32_t ids = 0, obs = 0;
  for (int k = 0; k < 4; ++k) {
    const int kk = k < P.s0_max ? k : P.s0_max - 1;
    const int sid = P.s0_ids[(size_t)t * P.s0_max + kk];
    ids |= (uint32_t)(sid & 0xFF) << (8 * k);
    obs |= (uint32_t)(P.state_map[(size_t)t * P.S + sid] & 0xFF) << (8 * k);
    h.s0_cdf[k] = k < P.s0_max ? P.s0_cdf[(size_t)t * P.s0_max + k] : 1.0;
  }
  h.s0_ids = ids;
  h.s0_obs = obs;
  h.pad0[0] = h.pad0[1] = h.pad0[2] = 0;
  for (int q = 0; q < 16; ++q) {
    uint32_t w = 0;
    for (int b = 0; b < 4; ++b) {
      const int sidx = 4 * q + b;
      if (sidx < P.S) w |= (uint32_t)(P.state_map[(size_t)t * P.S + sidx] & 0xFF) << (8 * b);
    }
    h.obs[q] = w;
  }
  hdr[t] = h;
}

// largest observation id (decides whether ids fit the header's bytes)
__global__ __launch_bounds__(256) void anymdp_max_obs_kernel(const int32_t* state_map, size_t n, int* out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) atomicMax(out, state_map[idx]);
}

template <bool INJECT>
__global__ __launch_bounds__(256) void anymdp_reset_kernel(AnyMDPArgs P, const uint8_t* mask,
                                                           const double* u_in, int32_t* obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  if (mask && !mask[i]) return;
  const int t = P.env_task[i];
  double u;
  if (INJECT) {
    u = u_in[i];
  } else {
    const xv_u32x4 v = xv_env_draw(P.seed, P.gid_base + (uint64_t)i, P.tick, XV_DRAW_RESET);
    u = xv_u53(v.x, v.y);
  }
  const int s = anymdp_draw_s0(P, t, u);
  P.state[i] = s;
  P.steps[i] = 0;
  P.need_reset[i] = 0;
  if (obs) obs[i] = P.state_map[(size_t)t * P.S + s];
}

// info["transition_gt"] = transition_obs[self.state, action]   (anymdp_env.py:130, :12-20)
__global__ __launch_bounds__(256) void anymdp_tgt_kernel(AnyMDPArgs P, const int32_t* action, double* out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)P.n_env * P.S;
  if (idx >= total) return;
  const int i = (int)(idx / P.S), j = (int)(idx % P.S);
  const int t = P.env_task[i], s = P.state[i];
  int a = action[i];
  a = a < 0 ? 0 : (a >= P.A ? P.A - 1 : a);
  const uint64_t tm0 = P.term_mask[(size_t)t * P.words];
  const uint32_t r = ((uint32_t)t * P.S + s) * P.A + a;
  double v = 0.0;
  if (!anymdp_is_term(P, t, tm0, s)) v = *anymdp_cdf_ptr(P, r, j) - (j ? *anymdp_cdf_ptr(P, r, j - 1) : 0.0);
  out[(size_t)i * P.S + P.state_map[(size_t)t * P.S + j]] = v;
}

// ------------------------------------------------------------------------------------------------
// POMDP / multi-token POMDP (anymdp_env.py:116-128 token loop, :148-157 observation draws).  One lane per env,
// per-lane binary searches: d_act transition draws, then d_obs observation draws from obs_cdf[t][k][s][:].
// Draw order and Philox purposes as oracle/xeno_oracle.c (tok_*).
// ------------------------------------------------------------------------------------------------
struct AnyMDPTokArgs {
  const double* obs_cdf;   // [n_task][d_obs][S][n_obs]
  int n_obs, d_obs, d_act;
};

struct AnyMDPTokIO {
  const int32_t* action;       // [n_env][d_act]
  const double* u;             // [d_act][n_env]   (INJECT)
  const float* z;              // [d_act][n_env]
  const double* u_obs;         // [d_obs][n_env]
  const double* u_reset;       // [n_env]
  const double* u_obs_reset;   // [d_obs][n_env]
  int32_t* obs;                // [n_env][d_obs]
  float* reward;
  float* reward_gt;
  uint8_t* terminated;
  uint8_t* truncated;
  int32_t* final_obs;          // [n_env][d_obs], nullable
};

__device__ __forceinline__ int xv_upper_bound_f64(const double* row, int n, double u) {
  int lo = 0, m = n;
  while (m > 0) {
    const int half = m >> 1;
    if (row[lo + half] <= u) { lo += half + 1; m -= half + 1; }
    else m = half;
  }
  return lo < n - 1 ? lo : n - 1;
}

// observation tokens of inner state s (after a step: RESET=false, after a reset: RESET=true)
template <bool INJECT, bool RESET>
__device__ __forceinline__ void anymdp_tok_observe(const AnyMDPArgs& P, const AnyMDPTokArgs& K, const AnyMDPTokIO& io,
                                                   int i, int t, int s, uint64_t gid, int32_t* out) {
  for (int k = 0; k < K.d_obs; ++k) {
    double u;
    if (INJECT) {
      u = (RESET ? io.u_obs_reset : io.u_obs)[(size_t)k * P.n_env + i];
    } else {
      const xv_u32x4 w = xv_env_draw(P.seed, gid, P.tick, 64u + (uint32_t)k);
      u = RESET ? xv_u53(w.z, w.w) : xv_u53(w.x, w.y);
    }
    const double* row = K.obs_cdf + ((((size_t)t * K.d_obs + k) * P.S) + s) * (size_t)K.n_obs;
    out[(size_t)i * K.d_obs + k] = xv_upper_bound_f64(row, K.n_obs, u);
  }
}

template <bool INJECT>
__global__ __launch_bounds__(256) void anymdp_tok_step_kernel(AnyMDPArgs P, AnyMDPTokArgs K, AnyMDPTokIO io, int mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  const int S = P.S, A = P.A, N = P.n_env;
  const int t = P.env_task[i];
  const uint64_t gid = P.gid_base + (uint64_t)i;
  const uint64_t tm0 = P.term_mask[(size_t)t * P.words];
  const int max_steps = P.max_steps[t];
  int s = P.state[i], steps = P.steps[i], nr = P.need_reset[i];
  uint32_t err = 0;
  if (io.final_obs) for (int k = 0; k < K.d_obs; ++k) io.final_obs[(size_t)i * K.d_obs + k] = -1;
  float rsum = 0.0f, rgsum = 0.0f;
  int term = 0, trunc = 0;
  bool do_reset = false;
  if (mode == XV_AUTORESET_NEXT_STEP && nr) {
    do_reset = true;
  } else if (mode == XV_AUTORESET_DISABLED && anymdp_is_term(P, t, tm0, s)) {
    err |= XV_DEVERR_STEP_TERMINAL;   // reference raises (:95-96)
    anymdp_tok_observe<INJECT, false>(P, K, io, i, t, s, gid, io.obs);
    term = 1; trunc = steps >= max_steps;
  } else {
    steps += 1;                       // :113, once per step
    trunc = steps >= max_steps;       // :114
    for (int k = 0; k < K.d_act; ++k) {   // :120-126
      int a = io.action[(size_t)i * K.d_act + k];
      if (a < 0 || a >= A) { err |= XV_DEVERR_ACTION_RANGE; a = a < 0 ? 0 : A - 1; }
      double u;
      float z;
      if (INJECT) {
        u = io.u[(size_t)k * N + i];
        z = io.z[(size_t)k * N + i];
      } else {
        const xv_u32x4 w = xv_env_draw(P.seed, gid, P.tick, 32u + (uint32_t)k);
        u = xv_u53(w.x, w.y);
        z = xv_normal1(w.z, w.w);
      }
      const uint32_t rowidx = ((uint32_t)t * S + s) * A + a;
      int lo = 0, m = S;
      while (m > 0) {
        const int half = m >> 1;
        if (*anymdp_cdf_ptr(P, rowidx, lo + half) <= u) { lo += half + 1; m -= half + 1; }
        else m = half;
      }
      const int s2 = lo < S - 1 ? lo : S - 1;
      const float2 rsv = anymdp_rs(P, rowidx, s2);
      rsum = rsum + fmaf(rsv.y, z, rsv.x);
      rgsum = rgsum + rsv.x;
      s = s2;
      if (anymdp_is_term(P, t, tm0, s2)) { term = 1; break; }
    }
    anymdp_tok_observe<INJECT, false>(P, K, io, i, t, s, gid, io.obs);
    if (term || trunc) {
      if (mode == XV_AUTORESET_SAME_STEP) {
        if (io.final_obs)
          for (int k = 0; k < K.d_obs; ++k) io.final_obs[(size_t)i * K.d_obs + k] = io.obs[(size_t)i * K.d_obs + k];
        do_reset = true;
      } else if (mode == XV_AUTORESET_NEXT_STEP) {
        nr = 1;
      }
    }
  }
  if (do_reset) {
    double ur;
    if (INJECT) ur = io.u_reset[i];
    else {
      const xv_u32x4 v = xv_env_draw(P.seed, gid, P.tick, XV_DRAW_RESET);
      ur = xv_u53(v.x, v.y);
    }
    s = anymdp_draw_s0(P, t, ur);
    steps = 0;
    nr = 0;
    anymdp_tok_observe<INJECT, true>(P, K, io, i, t, s, gid, io.obs);
  }
  P.state[i] = s; P.steps[i] = steps; P.need_reset[i] = (uint8_t)nr;
  io.reward[i] = rsum; io.reward_gt[i] = rgsum;
  io.terminated[i] = (uint8_t)term; io.truncated[i] = (uint8_t)trunc;
  if (err) atomicOr(P.err, err);
}

template <bool INJECT>
__global__ __launch_bounds__(256) void anymdp_tok_reset_kernel(AnyMDPArgs P, AnyMDPTokArgs K, AnyMDPTokIO io,
                                                               const uint8_t* mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  if (mask && !mask[i]) return;
  const int t = P.env_task[i];
  const uint64_t gid = P.gid_base + (uint64_t)i;
  double ur;
  if (INJECT) ur = io.u_reset[i];
  else {
    const xv_u32x4 v = xv_env_draw(P.seed, gid, P.tick, XV_DRAW_RESET);
    ur = xv_u53(v.x, v.y);
  }
  const int s = anymdp_draw_s0(P, t, ur);
  P.state[i] = s; P.steps[i] = 0; P.need_reset[i] = 0;
  if (io.obs) anymdp_tok_observe<INJECT, true>(P, K, io, i, t, s, gid, io.obs);
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" int xv_anymdp_create(xv_engine* e, int n_env, int n_task, int S, int A, int s0_max,
                                const void* rows, const int32_t* state_map, const uint64_t* term_mask,
                                const double* s0_cdf, const int32_t* s0_ids, const int32_t* max_steps,
                                const int32_t* env_task, xv_anymdp** out) {
  XV_CHECK_ARG(out != nullptr);
  *out = nullptr;
  XV_CHECK_ARG(e != nullptr);
  XV_CHECK_ARG(n_env > 0 && n_task > 0);
  XV_CHECK_ARG(S >= 2 && S <= 256 && A >= 2 && A <= 64 && s0_max >= 1 && s0_max <= 256);
  XV_CHECK_ARG(rows && state_map && term_mask && s0_cdf && s0_ids && max_steps && env_task);
  const int NB = (S + 15) / 16;
  XV_CHECK_ARG((uint64_t)n_task * S * A * NB < 0xFFFFFFFFull);  // block index is a 32-bit word on the device
  XV_HIP(hipSetDevice(e->device));
  xv_anymdp* h = new (std::nothrow) xv_anymdp();
  if (!h) {
    xv_set_error("xv_anymdp_create: out of host memory");
    return XV_ERR_NOMEM;
  }
  h->eng = e;
  h->search = XV_ANYMDP_SEARCH_AUTO;
  h->obs_cdf = nullptr; h->n_obs = 0; h->d_obs = 0; h->d_act = 0;
  AnyMDPArgs& a = h->a;
  a.rows = (const double*)rows; a.state_map = state_map; a.term_mask = term_mask;
  a.s0_cdf = s0_cdf; a.s0_ids = s0_ids; a.max_steps = max_steps; a.env_task = env_task;
  a.n_env = n_env; a.n_task = n_task; a.S = S; a.A = A; a.s0_max = s0_max; a.words = (S + 63) / 64;
  a.NB = NB;
  a.err = e->d_err;
  a.state = nullptr; a.steps = nullptr; a.need_reset = nullptr; a.hdr = nullptr; a.fence = nullptr;
  a.seed = e->seed; a.gid_base = e->env_id_base; a.tick = 0;

  // the fast path needs S <= 64, s0_max <= 4 and observation ids that fit a byte
  bool fast = (S <= 64 && s0_max <= 4);
  if (fast) {
    int* d_max = nullptr;
    int h_max = 0;
    XV_HIP(hipMalloc(&d_max, sizeof(int)));
    XV_HIP(hipMemsetAsync(d_max, 0, sizeof(int), e->stream));
    const size_t n = (size_t)n_task * S;
    hipLaunchKernelGGL(anymdp_max_obs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream,
                       state_map, n, d_max);
    XV_HIP(hipMemcpyAsync(&h_max, d_max, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    XV_HIP(hipStreamSynchronize(e->stream));
    XV_HIP(hipFree(d_max));
    fast = h_max < 256;
  }

  AnyMDPHdr* hdr = nullptr;
  double* fence = nullptr;
  const size_t n_rows = (size_t)n_task * S * A;
  hipError_t m = hipMalloc(&a.state, sizeof(int32_t) * (size_t)n_env);
  if (m == hipSuccess) m = hipMalloc(&a.steps, sizeof(int32_t) * (size_t)n_env);
  if (m == hipSuccess) m = hipMalloc(&a.need_reset, (size_t)n_env);
  if (m == hipSuccess && fast) m = hipMalloc(&hdr, sizeof(AnyMDPHdr) * (size_t)n_task);
  if (m == hipSuccess && fast) m = hipMalloc(&fence, n_rows * 4 * sizeof(double));
  if (m == hipSuccess) m = hipMemsetAsync(a.state, 0, sizeof(int32_t) * (size_t)n_env, e->stream);
  if (m == hipSuccess) m = hipMemsetAsync(a.steps, 0, sizeof(int32_t) * (size_t)n_env, e->stream);
  if (m == hipSuccess) m = hipMemsetAsync(a.need_reset, 1, (size_t)n_env, e->stream);
  if (m != hipSuccess) {
    xv_set_error("xv_anymdp_create: device allocation failed: %s", hipGetErrorString(m));
    if (a.state) (void)hipFree(a.state);
    if (a.steps) (void)hipFree(a.steps);
    if (a.need_reset) (void)hipFree(a.need_reset);
    if (hdr) (void)hipFree(hdr);
    if (fence) (void)hipFree(fence);
    delete h;
    return XV_ERR_HIP;
  }
  if (fast) {
    hipLaunchKernelGGL(anymdp_pack_hdr_kernel, dim3(xv_div_up(n_task, 256)), dim3(256), 0, e->stream, a, hdr);
    hipLaunchKernelGGL(anymdp_build_fence_kernel, dim3((unsigned)((n_rows * 4 + 255) / 256)), dim3(256), 0,
                       e->stream, a.rows, fence, n_rows, NB);
    a.hdr = hdr;
    a.fence = fence;
  }
  XV_LAUNCH_CHECK();
  *out = h;
  return XV_OK;
}

extern "C" int xv_anymdp_destroy(xv_anymdp* h) {
  if (!h) return XV_OK;
  (void)hipSetDevice(h->eng->device);
  (void)hipStreamSynchronize(h->eng->stream);
  (void)hipFree(h->a.state);
  (void)hipFree(h->a.steps);
  (void)hipFree(h->a.need_reset);
  if (h->a.hdr) (void)hipFree((void*)h->a.hdr);
  if (h->a.fence) (void)hipFree((void*)h->a.fence);
  delete h;
  return XV_OK;
}

static inline void anymdp_bind_rng(xv_anymdp* h, uint64_t ticks) {
  h->a.seed = h->eng->seed;
  h->a.gid_base = h->eng->env_id_base;
  h->a.tick = h->eng->tick;
  h->eng->tick += ticks;
}

extern "C" int xv_anymdp_reset(xv_anymdp* h, const uint8_t* mask, int32_t* obs) {
  XV_CHECK_ARG(h != nullptr);
  anymdp_bind_rng(h, 1);
  hipLaunchKernelGGL(anymdp_reset_kernel<false>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0,
                     h->eng->stream, h->a, mask, (const double*)nullptr, obs);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_anymdp_reset_injected(xv_anymdp* h, const uint8_t* mask, const double* u, int32_t* obs) {
  XV_CHECK_ARG(h != nullptr && u != nullptr);
  anymdp_bind_rng(h, 0);
  hipLaunchKernelGGL(anymdp_reset_kernel<true>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0,
                     h->eng->stream, h->a, mask, u, obs);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

template <bool INJECT>
static int anymdp_launch_step(xv_anymdp* h, const AnyMDPStepIO& io, int T, int mode) {
  const dim3 grid(xv_div_up(h->a.n_env, 256)), block(256);
#define XV_LAUNCH_STEP(SEARCH, HDR, ROLL)                                                          \
  hipLaunchKernelGGL((anymdp_step_kernel<INJECT, SEARCH, HDR, ROLL>), grid, block, 0, h->eng->stream, \
                     h->a, io, T, mode)
  const bool roll = T > 1;
  const bool fast = h->a.hdr != nullptr;
  if (fast && h->search != XV_ANYMDP_SEARCH_BINARY) {
    if (roll) XV_LAUNCH_STEP(SEARCH_FENCE, true, true); else XV_LAUNCH_STEP(SEARCH_FENCE, true, false);
  } else if (fast) {
    if (roll) XV_LAUNCH_STEP(SEARCH_BINARY, true, true); else XV_LAUNCH_STEP(SEARCH_BINARY, true, false);
  } else {
    if (roll) XV_LAUNCH_STEP(SEARCH_BINARY, false, true); else XV_LAUNCH_STEP(SEARCH_BINARY, false, false);
  }
#undef XV_LAUNCH_STEP
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_anymdp_step(xv_anymdp* h, const int32_t* action, int32_t* obs, float* reward,
                              float* reward_gt, uint8_t* terminated, uint8_t* truncated,
                              int32_t* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && action && obs && reward && reward_gt && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  anymdp_bind_rng(h, 1);
  AnyMDPStepIO io{action, nullptr, nullptr, nullptr, obs, reward, reward_gt, terminated, truncated, final_obs, nullptr, nullptr, 0.0f};
  return anymdp_launch_step<false>(h, io, 1, autoreset_mode);
}

extern "C" int xv_anymdp_step_injected(xv_anymdp* h, const int32_t* action, const double* u,
                                       const float* z, const double* u_reset, int32_t* obs,
                                       float* reward, float* reward_gt, uint8_t* terminated,
                                       uint8_t* truncated, int32_t* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && action && u && z && u_reset && obs && reward && reward_gt && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  anymdp_bind_rng(h, 0);
  AnyMDPStepIO io{action, u, z, u_reset, obs, reward, reward_gt, terminated, truncated, final_obs, nullptr, nullptr, 0.0f};
  return anymdp_launch_step<true>(h, io, 1, autoreset_mode);
}

extern "C" int xv_anymdp_step_many(xv_anymdp* h, int n_steps, int period, const int32_t* actions,
                                   int32_t* obs, float* reward, float* reward_gt, uint8_t* terminated,
                                   uint8_t* truncated, int32_t* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && n_steps > 0 && period > 0);
  XV_CHECK_ARG(actions && obs && reward && reward_gt && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  const size_t n = (size_t)h->a.n_env;
  for (int k = 0; k < n_steps; ++k) {
    const size_t off = (size_t)(k % period) * n;
    anymdp_bind_rng(h, 1);
    AnyMDPStepIO io{actions + off, nullptr, nullptr, nullptr, obs + off, reward + off, reward_gt + off,
                    terminated + off, truncated + off, final_obs ? final_obs + off : nullptr, nullptr, nullptr, 0.0f};
    const int rc = anymdp_launch_step<false>(h, io, 1, autoreset_mode);
    if (rc != XV_OK) return rc;
  }
  return XV_OK;
}

extern "C" int xv_anymdp_rollout(xv_anymdp* h, int T, const int32_t* actions, int32_t* obs, float* reward,
                                 float* reward_gt, uint8_t* terminated, uint8_t* truncated,
                                 int32_t* final_obs) {
  XV_CHECK_ARG(h && T > 0 && actions && obs && reward && reward_gt && terminated && truncated);
  anymdp_bind_rng(h, (uint64_t)T);
  AnyMDPStepIO io{actions, nullptr, nullptr, nullptr, obs, reward, reward_gt, terminated, truncated, final_obs, nullptr, nullptr, 0.0f};
  return anymdp_launch_step<false>(h, io, T, XV_AUTORESET_SAME_STEP);
}

extern "C" int xv_anymdp_rollout_teacher(xv_anymdp* h, int T, const uint8_t* greedy, float epsilon, int32_t* actions_out,
                                         int32_t* obs, float* reward, float* reward_gt, uint8_t* terminated,
                                         uint8_t* truncated, int32_t* final_obs) {
  XV_CHECK_ARG(h && T > 0 && greedy && actions_out && obs && reward && reward_gt && terminated && truncated);
  XV_CHECK_ARG(epsilon >= 0.0f && epsilon <= 1.0f);
  anymdp_bind_rng(h, (uint64_t)T);
  AnyMDPStepIO io{nullptr, nullptr, nullptr, nullptr, obs, reward, reward_gt, terminated, truncated, final_obs, greedy,
                  actions_out, epsilon};
  // T == 1 must still take the rollout instantiation (it is the one that honours the teacher fields per step)
  return anymdp_launch_step<false>(h, io, T, XV_AUTORESET_SAME_STEP);
}

extern "C" int xv_anymdp_set_search(xv_anymdp* h, int search) {
  XV_CHECK_ARG(h != nullptr);
  XV_CHECK_ARG(search == XV_ANYMDP_SEARCH_AUTO || search == XV_ANYMDP_SEARCH_BINARY ||
               search == XV_ANYMDP_SEARCH_FENCE);
  if (search == XV_ANYMDP_SEARCH_FENCE && !h->a.hdr) {
    xv_set_error("xv_anymdp_set_search: FENCE needs S <= 64, s0_max <= 4 and observation ids < 256");
    return XV_ERR_UNSUPPORTED;
  }
  h->search = search;
  return XV_OK;
}

extern "C" int xv_anymdp_get_state(xv_anymdp* h, int32_t* inner_state, int32_t* steps, uint8_t* need_reset) {
  XV_CHECK_ARG(h != nullptr);
  const size_t n = (size_t)h->a.n_env;
  if (inner_state) XV_HIP(hipMemcpyAsync(inner_state, h->a.state, n * 4, hipMemcpyDeviceToDevice, h->eng->stream));
  if (steps) XV_HIP(hipMemcpyAsync(steps, h->a.steps, n * 4, hipMemcpyDeviceToDevice, h->eng->stream));
  if (need_reset) XV_HIP(hipMemcpyAsync(need_reset, h->a.need_reset, n, hipMemcpyDeviceToDevice, h->eng->stream));
  return XV_OK;
}

extern "C" int xv_anymdp_set_state(xv_anymdp* h, const int32_t* inner_state, const int32_t* steps,
                                   const uint8_t* need_reset) {
  XV_CHECK_ARG(h != nullptr);
  const size_t n = (size_t)h->a.n_env;
  if (inner_state) XV_HIP(hipMemcpyAsync(h->a.state, inner_state, n * 4, hipMemcpyDeviceToDevice, h->eng->stream));
  if (steps) XV_HIP(hipMemcpyAsync(h->a.steps, steps, n * 4, hipMemcpyDeviceToDevice, h->eng->stream));
  if (need_reset) XV_HIP(hipMemcpyAsync(h->a.need_reset, need_reset, n, hipMemcpyDeviceToDevice, h->eng->stream));
  return XV_OK;
}

extern "C" int xv_anymdp_transition_gt(xv_anymdp* h, const int32_t* action, double* out) {
  XV_CHECK_ARG(h && action && out);
  const size_t total = (size_t)h->a.n_env * h->a.S;
  hipLaunchKernelGGL(anymdp_tgt_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->eng->stream,
                     h->a, action, out);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

// ---- POMDP / MTPOMDP entry points ----
extern "C" int xv_anymdp_set_observation_model(xv_anymdp* h, int n_obs, int d_obs, int d_act, const double* obs_cdf) {
  XV_CHECK_ARG(h && obs_cdf && n_obs >= 1 && d_obs >= 1 && d_obs <= 64 && d_act >= 1 && d_act <= 64);
  h->obs_cdf = obs_cdf; h->n_obs = n_obs; h->d_obs = d_obs; h->d_act = d_act;
  return XV_OK;
}

template <bool INJECT>
static int anymdp_tok_launch_step(xv_anymdp* h, const AnyMDPTokIO& io, int mode) {
  AnyMDPTokArgs K{h->obs_cdf, h->n_obs, h->d_obs, h->d_act};
  hipLaunchKernelGGL(anymdp_tok_step_kernel<INJECT>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                     h->a, K, io, mode);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_anymdp_step_tokens(xv_anymdp* h, const int32_t* action, int32_t* obs, float* reward,
                                     float* reward_gt, uint8_t* terminated, uint8_t* truncated, int32_t* final_obs,
                                     int autoreset_mode) {
  XV_CHECK_ARG(h && h->obs_cdf && action && obs && reward && reward_gt && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  anymdp_bind_rng(h, 1);
  AnyMDPTokIO io{action, nullptr, nullptr, nullptr, nullptr, nullptr, obs, reward, reward_gt, terminated, truncated, final_obs};
  return anymdp_tok_launch_step<false>(h, io, autoreset_mode);
}

extern "C" int xv_anymdp_step_tokens_injected(xv_anymdp* h, const int32_t* action, const double* u, const float* z,
                                              const double* u_obs, const double* u_reset, const double* u_obs_reset,
                                              int32_t* obs, float* reward, float* reward_gt, uint8_t* terminated,
                                              uint8_t* truncated, int32_t* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && h->obs_cdf && action && u && z && u_obs && u_reset && u_obs_reset && obs && reward && reward_gt &&
               terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  anymdp_bind_rng(h, 0);
  AnyMDPTokIO io{action, u, z, u_obs, u_reset, u_obs_reset, obs, reward, reward_gt, terminated, truncated, final_obs};
  return anymdp_tok_launch_step<true>(h, io, autoreset_mode);
}

extern "C" int xv_anymdp_reset_tokens(xv_anymdp* h, const uint8_t* mask, int32_t* obs) {
  XV_CHECK_ARG(h && h->obs_cdf);
  anymdp_bind_rng(h, 1);
  AnyMDPTokArgs K{h->obs_cdf, h->n_obs, h->d_obs, h->d_act};
  AnyMDPTokIO io{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, obs, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(anymdp_tok_reset_kernel<false>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                     h->a, K, io, mask);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_anymdp_reset_tokens_injected(xv_anymdp* h, const uint8_t* mask, const double* u_reset,
                                               const double* u_obs_reset, int32_t* obs) {
  XV_CHECK_ARG(h && h->obs_cdf && u_reset && u_obs_reset);
  anymdp_bind_rng(h, 0);
  AnyMDPTokArgs K{h->obs_cdf, h->n_obs, h->d_obs, h->d_act};
  AnyMDPTokIO io{nullptr, nullptr, nullptr, nullptr, u_reset, u_obs_reset, obs, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(anymdp_tok_reset_kernel<true>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                     h->a, K, io, mask);
  XV_LAUNCH_CHECK();
  return XV_OK;
}
