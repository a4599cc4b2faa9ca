// anymdp.hip — AnyMDP batched step / reset / rollout kernels for gfx950 and their C-ABI.
//
// Reproduces xenoverse/anymdp/anymdp_env.py: reset :81-90, single_step :92-110, step :112-132,
// get_observation :145-159 (MDP branch), per env, for N envs per launch.  One wavefront lane owns one env.
//
// Data movement per env-step (fp64 CDF, S = 64): one 512-B CDF row, one 8-B {reward, noise} pair, a few
// coalesced per-env words.  The row read is the whole cost, so the kernel is built around it:
//
//   * search mode W64 (S == 64): a wave owns 64 envs.  For env e of the wave the row cdf[t,s,a,:] is read
//     by ALL 64 lanes, one fp64 each — a single coalesced 512-B global_load_dwordx2 with a scalar base.
//     All 64 row loads of the wave are issued back to back (32 KiB in flight per wave, 128 KiB per CU), then
//     each row is searched with one v_cmp_le_f64 against the broadcast uniform, a 64-bit ballot and a
//     popcount:  s' = popcount(cdf[j] <= u) = numpy.searchsorted(cdf, u, 'right').  The rows live in
//     registers; no LDS round trip and no per-lane dependent probe chain.
//   * search mode GENERIC (any S <= 256): per-lane binary search straight from global memory
//     (ceil(log2 S)+1 dependent probes).
//
// The per-env arrays are struct-of-arrays (lane i reads word i): every access other than the row and the
// reward pair is a coalesced dword/byte stream.
#include "philox.h"
#include "xv_common.h"

// One 64-byte line per task holding everything a step needs besides the CDF row and the reward pair, so that
// it is fetched by ONE load issued together with the rows (fast path: S <= 64, s0_max <= 4).  Built once at
// create time from the ABI arrays.  Cuts the dependent-load chain of a step from six round trips to three:
// [per-env words] -> [header + 64 rows] -> [reward pair + obs id].
struct __attribute__((aligned(64))) AnyMDPHdr {
  uint64_t term_mask;    // bit s set <=> s terminal
  int32_t max_steps;
  uint32_t s0_ids;       // 4 x u8 inner-state ids of s_0 (padded with the last)
  uint64_t s0_obs;       // 4 x u16 observation ids of those states: reset needs no state_map gather
  double s0_cdf[4];      // inclusive CDF of s_0_prob padded with 1.0
  uint64_t pad;
};
static_assert(sizeof(AnyMDPHdr) == 64, "header must be one 64-byte line");

struct AnyMDPArgs {
  const AnyMDPHdr* hdr;  // engine-owned, nullptr when the fast path does not apply
  // borrowed task tables
  const double* cdf;
  const float2* rs;
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
  int n_env, n_task, S, A, s0_max, words;
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
};

struct xv_anymdp {
  xv_engine* eng;
  AnyMDPArgs a;
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

__device__ __forceinline__ double xv_readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

enum { SEARCH_GENERIC = 0, SEARCH_W64 = 1 };

// T_steps == 1: one vector step.  T_steps > 1: fused rollout, io arrays are [T][n_env], mode SAME_STEP.
// HDR: per-task scalars come from the packed 64-B header (S <= 64, s0_max <= 4) instead of five arrays.
template <bool INJECT, int SEARCH, bool HDR, bool ROLLOUT>
__global__ __launch_bounds__(256) void anymdp_step_kernel(AnyMDPArgs P, AnyMDPStepIO io, int T_steps,
                                                          int mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < P.n_env;
  const int ic = valid ? i : P.n_env - 1;
  const int lane = threadIdx.x & 63;
  const int S = P.S, A = P.A;

  // round trip 1: per-env words (coalesced)
  const int t = P.env_task[ic];
  int s = P.state[ic];
  int steps = P.steps[ic];
  int nr = P.need_reset[ic];
  int a_next = io.action[ic];

  // round trip 2 (issued with the rows below): per-task scalars
  AnyMDPHdr H;
  int max_steps;
  uint64_t tm0;
  if (HDR) {
    H = P.hdr[t];
    max_steps = H.max_steps;
    tm0 = H.term_mask;
  } else {
    max_steps = P.max_steps[t];
    tm0 = P.term_mask[(size_t)t * P.words];
  }
  const uint64_t gid = P.gid_base + (uint64_t)ic;
  uint32_t err = 0;

  const int T = ROLLOUT ? T_steps : 1;   // single step: straight-line code, counted vmcnt waits
  for (int ts = 0; ts < T; ++ts) {
    const size_t o = (size_t)ts * P.n_env + ic;
    int a = a_next;
    if (a < 0 || a >= A) {  // reference: assert action < self.na (:97)
      if (!(mode == XV_AUTORESET_NEXT_STEP && nr)) err |= XV_DEVERR_ACTION_RANGE;
      a = a < 0 ? 0 : A - 1;
    }
    const uint32_t rowidx = ((uint32_t)t * S + s) * A + a;

    // ---- issue the row reads first, so that the RNG arithmetic below overlaps their latency ----
    double rowv[64];
    if (SEARCH == SEARCH_W64) {
      const double* lane_base = P.cdf + lane;
#pragma unroll
      for (int e = 0; e < 64; ++e) {
        const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)rowidx, e);
        rowv[e] = lane_base[(size_t)r * 64];
      }
      // keep all 64 loads ahead of everything below: hipcc otherwise drains the first 8 to vmcnt(0)
      // before issuing the rest (three HBM round trips instead of one)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ROLLOUT && ts + 1 < T) a_next = io.action[o + P.n_env];   // prefetch behind the rows

    // ---- random inputs ----
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

    // ---- s' = upper_bound(cdf[s,a,:], u)   (:99-100, numpy.random.choice) ----
    int s2 = 0;
    if (SEARCH == SEARCH_W64) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 64; ++e) {
        const double ue = xv_readlane_f64(u, e);
        const unsigned long long m = __ballot(rowv[e] <= ue);
        const int cnt = __popcll(m);
        if (lane == e) s2 = cnt;
      }
      s2 = s2 < 63 ? s2 : 63;
    } else {
      const double* row = P.cdf + (size_t)rowidx * S;
      int lo = 0, n = S;
      while (n > 0) {
        const int half = n >> 1;
        if (row[lo + half] <= u) {
          lo += half + 1;
          n -= half + 1;
        } else {
          n = half;
        }
      }
      s2 = lo < S - 1 ? lo : S - 1;
    }

    // ---- round trip 3: dependent gathers on s' ----
    const float2 rsv = P.rs[(size_t)rowidx * S + s2];          // :103-104
    const int obs2 = P.state_map[(size_t)t * S + s2];          // :146-148
    const bool term2 = anymdp_is_term(P, t, tm0, s2);          // :107-108

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
      o_obs = P.state_map[(size_t)t * S + s];
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
        const int k = (int)(H.s0_cdf[0] <= u_reset) + (int)(H.s0_cdf[1] <= u_reset) +
                      (int)(H.s0_cdf[2] <= u_reset);
        s = (int)((H.s0_ids >> (8 * k)) & 0xFFu);
        o_obs = (int)((H.s0_obs >> (16 * k)) & 0xFFFFull);
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

// packs the per-task scalars into 64-byte headers (once, at create time)
__global__ __launch_bounds__(256) void anymdp_pack_hdr_kernel(AnyMDPArgs P, AnyMDPHdr* hdr) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P.n_task) return;
  AnyMDPHdr h;
  h.term_mask = P.term_mask[t];
  h.max_steps = P.max_steps[t];
  uint32_t ids = 0;
  uint64_t obs = 0;
  for (int k = 0; k < 4; ++k) {
    const int kk = k < P.s0_max ? k : P.s0_max - 1;
    const int sid = P.s0_ids[(size_t)t * P.s0_max + kk];
    ids |= (uint32_t)(sid & 0xFF) << (8 * k);
    obs |= (uint64_t)(P.state_map[(size_t)t * P.S + sid] & 0xFFFF) << (16 * k);
    h.s0_cdf[k] = k < P.s0_max ? P.s0_cdf[(size_t)t * P.s0_max + k] : 1.0;
  }
  h.s0_ids = ids;
  h.s0_obs = obs;
  h.pad = 0;
  hdr[t] = h;
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
  const double* c = P.cdf + (((size_t)t * P.S + s) * P.A + a) * (size_t)P.S;
  double v = 0.0;
  if (!anymdp_is_term(P, t, tm0, s)) v = c[j] - (j ? c[j - 1] : 0.0);
  out[(size_t)i * P.S + P.state_map[(size_t)t * P.S + j]] = v;
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" int xv_anymdp_create(xv_engine* e, int n_env, int n_task, int S, int A, int s0_max,
                                const double* cdf, const float* rs, const int32_t* state_map,
                                const uint64_t* term_mask, const double* s0_cdf, const int32_t* s0_ids,
                                const int32_t* max_steps, const int32_t* env_task, xv_anymdp** out) {
  XV_CHECK_ARG(out != nullptr);
  *out = nullptr;
  XV_CHECK_ARG(e != nullptr);
  XV_CHECK_ARG(n_env > 0 && n_task > 0);
  XV_CHECK_ARG(S >= 2 && S <= 256 && A >= 2 && A <= 64 && s0_max >= 1 && s0_max <= 256);
  XV_CHECK_ARG(cdf && rs && state_map && term_mask && s0_cdf && s0_ids && max_steps && env_task);
  XV_CHECK_ARG((uint64_t)n_task * S * A < 0xFFFFFFFFull);  // row index is a 32-bit word on the device
  XV_HIP(hipSetDevice(e->device));
  xv_anymdp* h = new (std::nothrow) xv_anymdp();
  if (!h) {
    xv_set_error("xv_anymdp_create: out of host memory");
    return XV_ERR_NOMEM;
  }
  h->eng = e;
  AnyMDPArgs& a = h->a;
  a.cdf = cdf; a.rs = (const float2*)rs; a.state_map = state_map; a.term_mask = term_mask;
  a.s0_cdf = s0_cdf; a.s0_ids = s0_ids; a.max_steps = max_steps; a.env_task = env_task;
  a.n_env = n_env; a.n_task = n_task; a.S = S; a.A = A; a.s0_max = s0_max; a.words = (S + 63) / 64;
  a.err = e->d_err;
  a.state = nullptr; a.steps = nullptr; a.need_reset = nullptr; a.hdr = nullptr;
  a.seed = e->seed; a.gid_base = e->env_id_base; a.tick = 0;
  AnyMDPHdr* hdr = nullptr;
  const bool use_hdr = (S <= 64 && s0_max <= 4);
  hipError_t m = hipMalloc(&a.state, sizeof(int32_t) * (size_t)n_env);
  if (m == hipSuccess && use_hdr) m = hipMalloc(&hdr, sizeof(AnyMDPHdr) * (size_t)n_task);
  if (m == hipSuccess) m = hipMalloc(&a.steps, sizeof(int32_t) * (size_t)n_env);
  if (m == hipSuccess) m = hipMalloc(&a.need_reset, (size_t)n_env);
  if (m == hipSuccess) m = hipMemsetAsync(a.state, 0, sizeof(int32_t) * (size_t)n_env, e->stream);
  if (m == hipSuccess) m = hipMemsetAsync(a.steps, 0, sizeof(int32_t) * (size_t)n_env, e->stream);
  if (m == hipSuccess) m = hipMemsetAsync(a.need_reset, 1, (size_t)n_env, e->stream);
  if (m != hipSuccess) {
    xv_set_error("xv_anymdp_create: env state allocation failed: %s", hipGetErrorString(m));
    if (a.state) hipFree(a.state);
    if (a.steps) hipFree(a.steps);
    if (a.need_reset) hipFree(a.need_reset);
    if (hdr) hipFree(hdr);
    delete h;
    return XV_ERR_HIP;
  }
  if (use_hdr) {
    hipLaunchKernelGGL(anymdp_pack_hdr_kernel, dim3(xv_div_up(n_task, 256)), dim3(256), 0, e->stream, a, hdr);
    a.hdr = hdr;
  }
  *out = h;
  return XV_OK;
}

extern "C" int xv_anymdp_destroy(xv_anymdp* h) {
  if (!h) return XV_OK;
  hipSetDevice(h->eng->device);
  hipStreamSynchronize(h->eng->stream);
  hipFree(h->a.state);
  hipFree(h->a.steps);
  hipFree(h->a.need_reset);
  if (h->a.hdr) hipFree((void*)h->a.hdr);
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
  if (h->a.S == 64 && h->a.hdr) {
    if (roll) XV_LAUNCH_STEP(SEARCH_W64, true, true); else XV_LAUNCH_STEP(SEARCH_W64, true, false);
  } else if (h->a.hdr) {
    if (roll) XV_LAUNCH_STEP(SEARCH_GENERIC, true, true); else XV_LAUNCH_STEP(SEARCH_GENERIC, true, false);
  } else {
    if (roll) XV_LAUNCH_STEP(SEARCH_GENERIC, false, true); else XV_LAUNCH_STEP(SEARCH_GENERIC, false, false);
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
  AnyMDPStepIO io{action, nullptr, nullptr, nullptr, obs, reward, reward_gt, terminated, truncated, final_obs};
  return anymdp_launch_step<false>(h, io, 1, autoreset_mode);
}

extern "C" int xv_anymdp_step_injected(xv_anymdp* h, const int32_t* action, const double* u,
                                       const float* z, const double* u_reset, int32_t* obs,
                                       float* reward, float* reward_gt, uint8_t* terminated,
                                       uint8_t* truncated, int32_t* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && action && u && z && u_reset && obs && reward && reward_gt && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  anymdp_bind_rng(h, 0);
  AnyMDPStepIO io{action, u, z, u_reset, obs, reward, reward_gt, terminated, truncated, final_obs};
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
                    terminated + off, truncated + off, final_obs ? final_obs + off : nullptr};
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
  AnyMDPStepIO io{actions, nullptr, nullptr, nullptr, obs, reward, reward_gt, terminated, truncated, final_obs};
  return anymdp_launch_step<false>(h, io, T, XV_AUTORESET_SAME_STEP);
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
