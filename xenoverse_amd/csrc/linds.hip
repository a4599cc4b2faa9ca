// linds.hip — LinDS (randomised LTI control) batched step / reset kernels for gfx950 and their C-ABI.
//
// Reproduces xenoverse/linds/linds_env.py: dynamics :78-80, get_observation :83-91, get_inner_cmd :93-98,
// reset :108-131, step :133-169, and RandomFourier.__call__ (utils/random_nn.py:362-368), for N envs per
// launch in fp32.  Operation order is fixed and restated by oracle/xeno_oracle.c (xo_linds_*):
//   x'_j = fmaf chain of Phi[j][k] x[k] over k in the MFMA accumulator order (linds_yorder), then k=0..NA-1 of
//          Gamma[j][k] act[k], + Xt[j], then fmaf(noise_scale, z_j, .)
//   y_j  = fmaf chain of C[j][k] x'[k] over k in the same order, + Y[j]
// (the accumulator order for BOTH products: the new state comes out of the matrix unit in exactly the registers the
//  next state product reads its operands from, so a fused multi-step kernel never re-arranges a state)
//
// One lane owns one env: its state vector lives in registers (component-major global layout, so the NS loads
// and stores of a wave are NS coalesced 256-B streams).  The task matrices are NOT read per lane: inside a
// waterfall over the distinct tasks present in a wave the task index is wave-uniform, so Phi/Gamma/C rows
// arrive through the scalar cache as SGPR operands of v_fma_f32 (s_load_dwordx16) — 6 KB per (wave, task)
// instead of 6 KB per env.  With envs grouped by task (64 per task = one wave per task) every wave makes one
// pass.
#include <vector>

#include "philox.h"
#include "xv_common.h"

struct LinDSArgs {
  xv_linds_tables T;
  const int32_t* env_task;
  float* x;            // [NS][n_env]
  int32_t* steps;
  uint8_t* need_reset;
  uint32_t* err;
  int n_env, n_task, NS, NA, NO, NI;
  uint64_t seed, gid_base, tick;
  // engine-built command table (nullptr if it would not fit the budget): cmd_tab[task][tt - ct_tmin][NO] holds
  // get_inner_cmd at integer time tt, already multiplied by target_valid, for tt in [ct_tmin, ct_tmin + ct_len)
  const float* cmd_tab;
  int ct_len, ct_tmin;
  // engine-built reset table: rst_tab[task][init index][NO + 4] = the observation of initial_states[idx] (NO floats)
  // and its tracking error against cmd(0) (slot NO): a restarting env reads 80 B instead of redoing y = C x + Y
  const float* rst_tab;
  // Slot layout (nullptr / n_slot == n_env: identity).  When the caller's env -> task map does not put 16 envs of one
  // task side by side, the engine orders its own state by task instead: envs are sorted by task (stably) and packed
  // into tiles of 16 slots, a task's last tile padded with empty slots (slot_env = -1).  State arrays (x, steps,
  // need_reset) are indexed by SLOT with stride n_slot; everything the caller sees (actions, outputs, global env id of
  // the random draws) stays indexed by ENV, so results do not depend on the layout.
  const int32_t* slot_env;   // [n_slot] env of a slot or -1
  const int32_t* env_slot;   // [n_env]  slot of an env
  const int32_t* tile_task;  // [n_slot / 16]
  int n_slot;
};

struct LinDSStepIO {
  const float* action;      // [n_env][NA]
  const float* z;           // [NS][n_env]  (INJECT)
  const int32_t* init_index;// [n_env]      (INJECT)
  float* obs;               // [n_env][NO]
  float* reward;
  uint8_t* terminated;
  uint8_t* truncated;
  float* cmd;               // [n_env][NO]
  float* error;
  float* final_obs;         // nullable
};

struct xv_linds {
  xv_engine* eng;
  LinDSArgs a;
  bool tiles_uniform;   // every aligned 16-env group of the CALLER's order shares a task (else: slot layout)
  int32_t *d_slot_env, *d_env_slot, *d_tile_task;   // owned; null in the identity layout
  int path;             // XV_LINDS_PATH_*
  float* cmd_tab;       // owned; a.cmd_tab points here while the table is enabled
  float* rst_tab;       // owned; likewise
};

// Task tables are read-only for the lifetime of a launch: reading them through the constant address space lets
// hipcc turn every wave-uniform access into an s_load (scalar cache, SGPR operand) instead of 64 identical
// vector loads.  (A plain `const float*` stays a vector global_load even when its address is uniform.)
#define XV_CONST_AS __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const XV_CONST_AS T* xv_cptr(const T* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (const XV_CONST_AS T*)p;
#pragma clang diagnostic pop
}

// k order of the observation product: the order in which a chain of 16x16x4 MFMAs visits k when x' is consumed from
// the accumulator layout of the previous product (M-tile m, register r, lane group g hold row 16 m + 4 g + r; slab
// s = 4 m + r adds its four k's in g order): position p = 4 s + g
__host__ __device__ constexpr int linds_yorder_at(int p) {
  return 16 * (p >> 4) + 4 * (p & 3) + ((p >> 2) & 3);
}

// command at integer time tt, times target_valid (uniform task tu; per-lane time)
template <int NO>
__device__ __forceinline__ void linds_cmd(const LinDSArgs& P, int tu, int nf, int tt, float (&out)[NO]) {
  const XV_CONST_AS float* valid = xv_cptr(P.T.valid) + (size_t)tu * NO;
  if (nf == 0) {   // static target: command * target_valid (:95-96)
    const XV_CONST_AS float* c0 = xv_cptr(P.T.cmd0) + (size_t)tu * NO;
#pragma unroll
    for (int j = 0; j < NO; ++j) out[j] = c0[j] * valid[j];
    return;
  }
#pragma unroll
  for (int j = 0; j < NO; ++j) out[j] = 0.0f;
  const double inv_period_t = (double)tt / xv_cptr(P.T.four_period)[tu];
  for (int k = 0; k < nf; ++k) {   // random_nn.py:362-368
    double ang = xv_cptr(P.T.four_omega)[(size_t)tu * XV_LINDS_KMAX + k] * inv_period_t;
    ang -= 6.283185307179586476925286766559 * rint(ang * 0.15915494309189533576888376337251);
    float sn, cs;
    sincosf((float)ang, &sn, &cs);
    const XV_CONST_AS float* c = xv_cptr(P.T.four_coef) + (((size_t)tu * XV_LINDS_KMAX + k) * NO) * 2;
#pragma unroll
    for (int j = 0; j < NO; ++j) {
      out[j] = fmaf(c[2 * j], sn, out[j]);
      out[j] = fmaf(c[2 * j + 1], cs, out[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < NO; ++j) out[j] *= valid[j];   // :98
}

// The command of an env depends on (task, integer time) only, and a step needs it at two times (tracked and
// reported), 2 x NO x nf sin/cos pairs per env-step when evaluated directly.  The engine tabulates it once per
// task at create time with the function above (same code, same bits); a step then reads two 64-B rows.  Times
// outside the table (an env stepped on past truncation with auto-reset disabled) are evaluated directly.
template <int NO>
__device__ __forceinline__ void linds_cmd_at(const LinDSArgs& P, int tu, int nf, int tt, float (&out)[NO]) {
  const int idx = tt - P.ct_tmin;
  if (P.cmd_tab != nullptr && idx >= 0 && idx < P.ct_len) {
    const float4* p = reinterpret_cast<const float4*>(P.cmd_tab + ((size_t)tu * P.ct_len + idx) * NO);
#pragma unroll
    for (int q = 0; q < NO / 4; ++q) {
      const float4 v = p[q];
      out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
    }
  } else {
    linds_cmd<NO>(P, tu, nf, tt, out);
  }
}

// y = C x + Y with the fixed k order (uniform task tu)
template <int NS, int NO>
__device__ __forceinline__ void linds_observe(const LinDSArgs& P, int tu, const float (&xs)[NS], float (&y)[NO]) {
  const XV_CONST_AS float* cT = xv_cptr(P.T.cT) + (size_t)tu * NS * NO;
  const XV_CONST_AS float* y0 = xv_cptr(P.T.y0) + (size_t)tu * NO;
#pragma unroll
  for (int j = 0; j < NO; ++j) y[j] = 0.0f;
#pragma unroll
  for (int p = 0; p < 32; ++p) {
    const int k = linds_yorder_at(p);
    if (k < NS) {
#pragma unroll
      for (int j = 0; j < NO; ++j) y[j] = fmaf(cT[k * NO + j], xs[k], y[j]);
      __builtin_amdgcn_sched_barrier(0);   // one row of scalar loads at a time (see linds_step_kernel)
    }
  }
#pragma unroll
  for (int j = 0; j < NO; ++j) y[j] = y[j] + y0[j];   // :85
}

// Sums over the observation rows (tracking error :127,:153, observation scale :154) run as FOUR fmaf chains, chain g over
// the rows j = 16 mo + 4 g + r (mo, r ascending) — the rows lane group g of the matrix kernel holds after y = C x' — and
// are combined as (p0 + p1) + (p2 + p3), which is what two xor-shuffles (16, 32) produce in every lane.  Oracle and
// scalar kernel use the same order (xeno_oracle.c: linds_err, linds_sumsq).
template <int NO>
__device__ __forceinline__ float linds_err(const LinDSArgs& P, int tu, const float (&y)[NO], const float (&c)[NO]) {
  const XV_CONST_AS float* valid = xv_cptr(P.T.valid) + (size_t)tu * NO;
  float p[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float acc = 0.0f;
#pragma unroll
    for (int mo = 0; mo < NO / 16; ++mo)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * mo + 4 * g + r;
        const float d = (y[j] - c[j]) * valid[j];
        acc = fmaf(d, d, acc);
      }
    p[g] = acc;
  }
  return sqrtf((p[0] + p[1]) + (p[2] + p[3]));
}
template <int NO>
__device__ __forceinline__ float linds_sumsq(const float (&y)[NO]) {
  float p[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float acc = 0.0f;
#pragma unroll
    for (int mo = 0; mo < NO / 16; ++mo)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = fmaf(y[16 * mo + 4 * g + r], y[16 * mo + 4 * g + r], acc);
    p[g] = acc;
  }
  return (p[0] + p[1]) + (p[2] + p[3]);
}
// the same two sums from a lane's own rows (register r of tile mo = row 16 mo + 4 g + r), combined across the four lane
// groups of an env: identical bits in all four
template <int MO>
__device__ __forceinline__ float linds_quad_sum(float part) {
  part = part + __shfl_xor(part, 16);
  return part + __shfl_xor(part, 32);
}

template <int N>
__device__ __forceinline__ void linds_store_row(float* dst, const float (&v)[N]) {
  float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
  for (int q = 0; q < N / 4; ++q) d4[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

template <int NO>
__global__ __launch_bounds__(256) void linds_build_cmd_tab_kernel(LinDSArgs P, float* tab) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)P.n_task * P.ct_len) return;
  const int t = (int)(idx / P.ct_len), k = (int)(idx % P.ct_len);
  float out[NO];
  linds_cmd<NO>(P, t, P.T.ints[(size_t)t * 4 + 3], k + P.ct_tmin, out);
  linds_store_row<NO>(tab + idx * NO, out);
}

// max over tasks of max_steps and of the command delay (sizes the command table)
__global__ __launch_bounds__(256) void linds_max_ints_kernel(const int32_t* ints, int n_task, int* out2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_task) return;
  atomicMax(out2, ints[(size_t)t * 4]);
  atomicMax(out2 + 1, ints[(size_t)t * 4 + 1]);
}


// reset of one env inside the uniform-task region: x = initial_states[idx]; obs; cmd(0); error
template <int NS, int NO>
__device__ __forceinline__ void linds_reset_env(const LinDSArgs& P, int tu, int nf, int n_init, int idx,
                                                float (&xs)[NS], float (&y)[NO], float (&c)[NO], float& err) {
  idx = idx < 0 ? 0 : (idx >= n_init ? n_init - 1 : idx);
  const float* x0 = P.T.init + ((size_t)tu * P.NI + idx) * NS;
#pragma unroll
  for (int k = 0; k < NS; ++k) xs[k] = x0[k];   // :117
  linds_cmd_at<NO>(P, tu, nf, 0, c);                // :120-126: the last pre-filled command is cmd(0)
  if (P.rst_tab != nullptr) {
    const float* row = P.rst_tab + ((size_t)tu * P.NI + idx) * (NO + 4);
#pragma unroll
    for (int j = 0; j < NO; ++j) y[j] = row[j];
    err = row[NO];
    return;
  }
  linds_observe<NS, NO>(P, tu, xs, y);
  err = linds_err<NO>(P, tu, y, c);
}

// builds rst_tab with the functions above (same code, same bits as evaluating at reset time)
template <int NS, int NO>
__global__ __launch_bounds__(256) void linds_build_reset_tab_kernel(LinDSArgs P, float* tab) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)P.n_task * P.NI) return;
  const int t = (int)(idx / P.NI), k = (int)(idx % P.NI);
  const int n_init = P.T.ints[(size_t)t * 4 + 2], nf = P.T.ints[(size_t)t * 4 + 3];
  float xs[NS], y[NO], c[NO], e = 0.0f;
  LinDSArgs Q = P;
  Q.rst_tab = nullptr;
  linds_reset_env<NS, NO>(Q, t, nf, n_init > 0 ? n_init : 1, k < n_init ? k : 0, xs, y, c, e);
  float* row = tab + idx * (NO + 4);
#pragma unroll
  for (int j = 0; j < NO; ++j) row[j] = y[j];
  row[NO] = e; row[NO + 1] = 0.0f; row[NO + 2] = 0.0f; row[NO + 3] = 0.0f;
}

__device__ __forceinline__ int linds_draw_init(const LinDSArgs& P, uint64_t gid, int n_init) {
  const xv_u32x4 v = xv_env_draw(P.seed, gid, P.tick, XV_DRAW_RESET);
  const int idx = (int)(xv_u53(v.x, v.y) * (double)n_init);
  return idx < n_init ? idx : n_init - 1;
}

template <int NS, int NA, int NO, bool INJECT>
__global__ __launch_bounds__(256) void linds_step_kernel(LinDSArgs P, LinDSStepIO io, int mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  const int N = P.n_env;
  const int NSL = P.n_slot;
  const int si = P.env_slot ? P.env_slot[i] : i;   // where this env's state lives
  const int t = P.env_task[i];
  const uint64_t gid = P.gid_base + (uint64_t)i;

  float xs[NS], a_raw[NA];
#pragma unroll
  for (int k = 0; k < NS; ++k) xs[k] = P.x[(size_t)k * NSL + si];
  {
    const float4* a4 = reinterpret_cast<const float4*>(io.action + (size_t)i * NA);
#pragma unroll
    for (int q = 0; q < NA / 4; ++q) {
      const float4 v = a4[q];
      a_raw[4 * q] = v.x; a_raw[4 * q + 1] = v.y; a_raw[4 * q + 2] = v.z; a_raw[4 * q + 3] = v.w;
    }
  }
  int steps = P.steps[si];
  int nr = P.need_reset[si];
  int init_idx = 0;
  if (INJECT) init_idx = io.init_index[i];

  float y[NO], crep[NO], fobs[NO];
  float o_r = 0.0f, o_err = 0.0f;
  // flags are per-lane integers (VGPRs): as bools they are SGPR lane masks that must survive later waterfall passes
  int o_term = 0, o_trunc = 0, wrote_fobs = 0;
  uint32_t err = 0;

  // waterfall over the distinct tasks of this wave: inside, `tu` is wave-uniform (SGPR) and so is every
  // table address derived from it -> scalar loads, broadcast operands
  for (;;) {
    const int tu_cmp = __builtin_amdgcn_readfirstlane(t);
    // hipcc's equality propagation rewrites the compared value as the per-lane `t` inside the branch, which would
    // turn every table address divergent (vector loads + spills): the copy used inside is laundered through an
    // SGPR-constrained empty asm BEFORE the branch, so the compiler cannot relate it to `t`
    int tu = tu_cmp;
    asm volatile("" : "+s"(tu));
    if (t == tu_cmp) {
      const XV_CONST_AS float* sc = xv_cptr(P.T.scal) + (size_t)tu * 8;
      const XV_CONST_AS int32_t* in = xv_cptr(P.T.ints) + (size_t)tu * 4;
      const int max_steps = in[0], delay = in[1], n_init = in[2], nf = in[3];
      if (!INJECT) init_idx = linds_draw_init(P, gid, n_init);
      if (mode == XV_AUTORESET_NEXT_STEP && nr) {
        // the call after a done ignores the action and returns the reset observation
        linds_reset_env<NS, NO>(P, tu, nf, n_init, init_idx, xs, y, crep, o_err);
        steps = 0;
        nr = 0;
      } else {
        float sa = 0.0f, act[NA];
#pragma unroll
        for (int k = 0; k < NA; ++k) {   // :138 clip; :164 cost on the RAW padded action
          sa = fmaf(a_raw[k], a_raw[k], sa);
          act[k] = a_raw[k] < -1.0f ? -1.0f : (a_raw[k] > 1.0f ? 1.0f : a_raw[k]);
        }
        float xn[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) xn[j] = 0.0f;
        const XV_CONST_AS float* phiT = xv_cptr(P.T.phiT) + (size_t)tu * NS * NS;
        const XV_CONST_AS float* gamT = xv_cptr(P.T.gamT) + (size_t)tu * NA * NS;
        const XV_CONST_AS float* xtv = xv_cptr(P.T.xt) + (size_t)tu * NS;
#pragma unroll
        for (int p = 0; p < 32; ++p) {   // :78-80, Phi x, k in linds_yorder
          const int k = linds_yorder_at(p);
          if (k < NS) {
#pragma unroll
            for (int j = 0; j < NS; ++j) xn[j] = fmaf(phiT[k * NS + j], xs[k], xn[j]);
            // without this hipcc hoists all ~160 s_load_dwordx16 of the three products to the top and spills
            // ~2,500 SGPRs into VGPR lanes (v_writelane/v_readlane dominate the kernel)
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {   // + Gamma act
#pragma unroll
          for (int j = 0; j < NS; ++j) xn[j] = fmaf(gamT[k * NS + j], act[k], xn[j]);
          __builtin_amdgcn_sched_barrier(0);
        }
        const float noise_scale = sc[4];
        bool bad = false;
#pragma unroll
        for (int q = 0; q < NS / 4; ++q) {   // + Xt + noise, four components at a time (the normals are made here, not
          float z[4];                        // held in 32 registers across the products: that spilled to scratch)
          if (INJECT) {
#pragma unroll
            for (int k = 0; k < 4; ++k) z[k] = io.z[(size_t)(4 * q + k) * N + i];
          } else {   // purpose 16+q: words (0,1) -> z[4q], z[4q+1]; (2,3) -> z[4q+2], z[4q+3]
            const xv_u32x4 w = xv_env_draw(P.seed, gid, P.tick, XV_DRAW_NOISE + (uint32_t)q);
            xv_box_muller_fast(w.x, w.y, &z[0], &z[1]);
            xv_box_muller_fast(w.z, w.w, &z[2], &z[3]);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int j = 4 * q + k;
            xn[j] = xn[j] + xtv[j];
            xn[j] = fmaf(noise_scale, z[k], xn[j]);
            bad = bad || !(fabsf(xn[j]) <= 3.0e38f);
          }
        }
        if (bad) err |= XV_DEVERR_NONFINITE;
        linds_observe<NS, NO>(P, tu, xn, y);      // :145
        steps += 1;                               // :147
        float ctrack[NO];
        linds_cmd_at<NO>(P, tu, nf, steps - 1 - delay, ctrack);   // :150-151 tracked command
        linds_cmd_at<NO>(P, tu, nf, steps, crep);                 // :168 reported command
        o_err = linds_err<NO>(P, tu, y, ctrack);               // :153
        const float obs_scale = sqrtf(linds_sumsq<NO>(y));     // :154
        o_term = ((o_err > 10.0f) || (obs_scale > 20.0f)) ? 1 : 0;   // :156
        o_r = o_term ? -sc[2] : 0.0f;                          // :158-161
        float tmp = fmaf(-sc[3], o_err, sc[1]);
        tmp = fmaf(-sc[0], sa, tmp);
        o_r = fmaf(tmp, sc[5], o_r);                           // :163-164
        o_trunc = (steps >= max_steps - 1) ? 1 : 0;            // :165
#pragma unroll
        for (int k = 0; k < NS; ++k) xs[k] = xn[k];
        if (o_term || o_trunc) {
          if (mode == XV_AUTORESET_SAME_STEP) {
#pragma unroll
            for (int j = 0; j < NO; ++j) fobs[j] = y[j];
            wrote_fobs = 1;
            linds_reset_env<NS, NO>(P, tu, nf, n_init, init_idx, xs, y, crep, o_err);
            steps = 0;
          } else if (mode == XV_AUTORESET_NEXT_STEP) {
            nr = 1;
          }
        }
      }
      break;
    }
  }

#pragma unroll
  for (int k = 0; k < NS; ++k) P.x[(size_t)k * NSL + si] = xs[k];
  P.steps[si] = steps;
  P.need_reset[si] = (uint8_t)nr;
  linds_store_row<NO>(io.obs + (size_t)i * NO, y);
  linds_store_row<NO>(io.cmd + (size_t)i * NO, crep);
  io.reward[i] = o_r;
  io.error[i] = o_err;
  io.terminated[i] = (uint8_t)o_term;
  io.truncated[i] = (uint8_t)o_trunc;
  if (io.final_obs) {
    if (!wrote_fobs) {
#pragma unroll
      for (int j = 0; j < NO; ++j) fobs[j] = 0.0f;
    }
    linds_store_row<NO>(io.final_obs + (size_t)i * NO, fobs);
  }
  if (err) atomicOr(P.err, err);
}

template <int NS, int NO, bool INJECT>
__global__ __launch_bounds__(256) void linds_reset_kernel(LinDSArgs P, const uint8_t* mask,
                                                          const int32_t* init_index, float* obs, float* cmd,
                                                          float* error) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  if (mask && !mask[i]) return;
  const int NSL = P.n_slot;
  const int si = P.env_slot ? P.env_slot[i] : i;
  const int t = P.env_task[i];
  float xs[NS], y[NO], c[NO];
  float e = 0.0f;
  for (;;) {
    const int tu_cmp = __builtin_amdgcn_readfirstlane(t);
    int tu = tu_cmp;
    asm volatile("" : "+s"(tu));   // keep the task index scalar (see linds_step_kernel)
    if (t == tu_cmp) {
      const XV_CONST_AS int32_t* in = xv_cptr(P.T.ints) + (size_t)tu * 4;
      const int n_init = in[2], nf = in[3];
      const int idx = INJECT ? init_index[i] : linds_draw_init(P, P.gid_base + (uint64_t)i, n_init);
      linds_reset_env<NS, NO>(P, tu, nf, n_init, idx, xs, y, c, e);
      break;
    }
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) P.x[(size_t)k * NSL + si] = xs[k];
  P.steps[si] = 0;
  P.need_reset[si] = 0;
  if (obs) linds_store_row<NO>(obs + (size_t)i * NO, y);
  if (cmd) linds_store_row<NO>(cmd + (size_t)i * NO, c);
  if (error) error[i] = e;
}

// ------------------------------------------------------------------------------------------------
// MFMA path: one wave = one tile of 16 envs that share a task (the caller's order, or the engine's slot layout).
//
//   v_mfma_f32_16x16x4_f32, lane l = 16 g + n:   A[i = n][k = g]   B[k = g][col = n]   D reg r = D[row = 4 g + r][col = n]
//   (probed on gfx950, scripts/devtools/mfma16_probe.hip: the four k-products are added to the accumulator as an
//   ascending fmaf chain — bit for bit the chains of the scalar kernel and of the oracle).
//
//   x'^T = Phi X^T + Gamma A^T, rows in M-tiles of 16:  D_m[j = 16 m + row][env]: lane (n, g) ends up with
//     x'_env(n)[16 m + 4 g + r] in register r of tile m — exactly the four components Philox call q = 4 m + g provides
//     the process noise for.  Those registers are the B operands of BOTH products that consume the state: slab
//     s = 4 m + r multiplies the k-group {16 m + 4 g + r : g = 0..3}
//       y^T = C x'^T            (A operand C[16 mo + n][that k]  = cT[k][16 mo + n])
//       x''^T = Phi x'^T + ...  (A operand Phi[16 m' + n][that k] = phiT[k][16 m' + n], 64-B segments of phiT rows)
//     No LDS, no shuffles between the products nor between steps (the fused roll-out kernel keeps the state in these
//     registers for T steps); the price is the k order of the two chains, linds_yorder_at, which scalar kernel and
//     oracle follow.  From memory the B operand of slab s is X[16 (s>>2) + 4 g + (s&3)][tile + n], coalesced.
//   Half the tile width of a 32x32x2 formulation: twice the waves for the same batch (4 per SIMD at 65,536 envs) and
//   a quarter of the MFMA latency per k (8 passes per 4 k instead of 16 per 2) — this kernel is latency-bound.
//   Per-env scalar work (error, reward, flags) is done redundantly by the four lanes of an env; each lane stores its
//   own 16-byte quarter of the observation / command rows.
// ------------------------------------------------------------------------------------------------
typedef float xv_f32x4 __attribute__((ext_vector_type(4)));

// v[g] for a lane-varying g in 0..3, opaque to hipcc (which otherwise folds a select over array elements into one
// dynamically indexed stack access, i.e. scratch memory)
// hipcc (ROCm 7.2) counts the wait states between an MFMA and the first VALU read of its result along the fall-through
// path only: when the read sits behind a conditional branch taken on the common path (the "outside the command table"
// test below), the taken path gets `s_nop 0` where 10 wait states are due, and the LAST result register (row 4 g + 3) is
// read before the matrix unit has written it — observed as wrong observation rows 3, 7, 11, 15 while the state stayed
// right.  The chains whose results are consumed after a branch are therefore followed by the wait states themselves.
// (the asm names the accumulator tuple as an in/out AGPR operand, which orders it after the MFMA and before the reads)
__device__ __forceinline__ void xv_mfma_settle(xv_f32x4& acc) { asm volatile("s_nop 7\n\ts_nop 3" : "+a"(acc)); }

__device__ __forceinline__ float xv_sel4(int g, float a, float b, float c, float d) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  const float lo = (g & 1) ? b : a, hi = (g & 1) ? d : c;
  return (g & 2) ? hi : lo;
}

template <int NS, int NA, int NO, bool INJECT>
__device__ __forceinline__ void linds_step_mfma_body(const LinDSArgs& P, const LinDSStepIO& io, int mode) {
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int tile0 = wave * 16;
  if (tile0 >= P.n_slot) return;   // wave-uniform
  const int N = P.n_env;           // stride of the caller's env-ordered arrays
  const int NSL = P.n_slot;        // stride of the engine's slot-ordered state
  const int n = lane & 15, g = lane >> 4;
  constexpr int MT = NS / 16, MO = NO / 16, KS = NS / 4, KA = NA / 4;
  // es: this lane's state slot; e: the env it serves (its I/O rows and the global id of its draws)
  const int es = tile0 + n < NSL ? tile0 + n : NSL - 1;
  int e_raw = tile0 + n;
  if (P.slot_env != nullptr) e_raw = P.slot_env[es];
  const bool valid = tile0 + n < NSL && e_raw >= 0 && e_raw < N;
  const int e = valid ? e_raw : 0;
  const int t = __builtin_amdgcn_readfirstlane(P.tile_task ? P.tile_task[wave] : P.env_task[tile0]);
  const uint64_t gid = P.gid_base + (uint64_t)e;

  const XV_CONST_AS float* sc = xv_cptr(P.T.scal) + (size_t)t * 8;
  const XV_CONST_AS int32_t* in = xv_cptr(P.T.ints) + (size_t)t * 4;
  const int max_steps = in[0], delay = in[1], n_init = in[2], nf = in[3];

  int steps = P.steps[es];
  int nr = P.need_reset[es];
  float a_raw[NA];
  {
    const float4* a4 = reinterpret_cast<const float4*>(io.action + (size_t)e * NA);
#pragma unroll
    for (int q = 0; q < NA / 4; ++q) {
      const float4 v = a4[q];
      a_raw[4 * q] = v.x; a_raw[4 * q + 1] = v.y; a_raw[4 * q + 2] = v.z; a_raw[4 * q + 3] = v.w;
    }
  }
  int init_idx = 0;
  if (INJECT) init_idx = io.init_index[e];   // free-running: drawn below, only in waves that restart an env

  // ---- all operand fragments first: every load of the step is in flight before the first MFMA ----
  const float* phiT = P.T.phiT + (size_t)t * NS * NS;
  const float* gamT = P.T.gamT + (size_t)t * NA * NS;
  const float* cT = P.T.cT + (size_t)t * NS * NO;
  const float* xtv = P.T.xt + (size_t)t * NS;
  float pa[MT][KS], pb[KS], ga[MT][KA], ca[MO][KS], xtr[MT][4];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {   // slab kk multiplies the k-group {16 (kk>>2) + 4 g + (kk&3)}: linds_yorder
    const int k = 16 * (kk >> 2) + 4 * g + (kk & 3);
    pb[kk] = P.x[(size_t)k * NSL + es];
#pragma unroll
    for (int m = 0; m < MT; ++m) pa[m][kk] = phiT[k * NS + 16 * m + n];
  }
#pragma unroll
  for (int kk = 0; kk < KA; ++kk)
#pragma unroll
    for (int m = 0; m < MT; ++m) ga[m][kk] = gamT[(4 * kk + g) * NS + 16 * m + n];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = 16 * (s >> 2) + 4 * g + (s & 3);
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) ca[mo][s] = cT[k * NO + 16 * mo + n];
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) xtr[m][r] = xtv[16 * m + 4 * g + r];
  // command rows, requested with everything else so that their latency (a second dependent level: steps -> row
  // address) runs under the products: the tracked command cmd(steps - delay) in full (:150-151; the error needs all
  // of it) and this lane's quarter of the reported one cmd(steps + 1) (:168).  A time outside the table (or no table)
  // is evaluated directly further down
  const int steps_new = steps + 1;                            // :147
  const int trk_time = steps_new - 1 - delay, rep_time = steps_new;
  const int trk_idx = trk_time - P.ct_tmin, rep_idx = rep_time - P.ct_tmin;
  const bool trk_in = P.cmd_tab != nullptr && trk_idx >= 0 && trk_idx < P.ct_len;
  const bool rep_in = P.cmd_tab != nullptr && rep_idx >= 0 && rep_idx < P.ct_len;
  float ctr[MO][4], crep[MO][4];   // this lane's rows 16 mo + 4 g + r of the two commands
  if (P.cmd_tab != nullptr) {
    const float4* p = reinterpret_cast<const float4*>(P.cmd_tab + ((size_t)t * P.ct_len + (trk_in ? trk_idx : 0)) * NO);
    const float4* pr = reinterpret_cast<const float4*>(P.cmd_tab + ((size_t)t * P.ct_len + (rep_in ? rep_idx : 0)) * NO);
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) {
      const float4 u = p[4 * mo + g], v = pr[4 * mo + g];
      ctr[mo][0] = u.x; ctr[mo][1] = u.y; ctr[mo][2] = u.z; ctr[mo][3] = u.w;
      crep[mo][0] = v.x; crep[mo][1] = v.y; crep[mo][2] = v.z; crep[mo][3] = v.w;
    }
  }
  float vld[MO][4];
  {
    const float* vp = P.T.valid + (size_t)t * NO;
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) {
      const float4 v = *reinterpret_cast<const float4*>(vp + 16 * mo + 4 * g);
      vld[mo][0] = v.x; vld[mo][1] = v.y; vld[mo][2] = v.z; vld[mo][3] = v.w;
    }
  }
  // process noise: independent of every load above, so it is computed while they are in flight.  Philox call
  // q = 4 m + g yields the normals of components 4 q .. 4 q + 3 = 16 m + 4 g + r
  float zr[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (INJECT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) zr[m][r] = io.z[(size_t)(16 * m + 4 * g + r) * N + e];
    } else {
      const xv_u32x4 w = xv_env_draw(P.seed, gid, P.tick, XV_DRAW_NOISE + (uint32_t)(4 * m + g));
      xv_box_muller_fast(w.x, w.y, &zr[m][0], &zr[m][1]);
      xv_box_muller_fast(w.z, w.w, &zr[m][2], &zr[m][3]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- x' = Phi x + Gamma act  (:78-80): MT independent accumulator chains, interleaved ----
  xv_f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = xv_f32x4{0, 0, 0, 0};
#pragma unroll
  for (int kk = 0; kk < KS; ++kk)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[m][kk], pb[kk], acc[m], 0, 0, 0);
  float sa = 0.0f;
#pragma unroll
  for (int k = 0; k < NA; ++k) sa = fmaf(a_raw[k], a_raw[k], sa);   // :164 cost on the RAW padded action
#pragma unroll
  for (int kk = 0; kk < KA; ++kk) {
    const float ar = xv_sel4(g, a_raw[4 * kk], a_raw[4 * kk + 1], a_raw[4 * kk + 2], a_raw[4 * kk + 3]);
    const float b = ar < -1.0f ? -1.0f : (ar > 1.0f ? 1.0f : ar);   // :138 clip
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[m][kk], b, acc[m], 0, 0, 0);
  }
  // + Xt + noise on this lane's components
  const float noise_scale = sc[4];
  xv_f32x4 xn[MT];
  int bad = 0;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc[m][r] + xtr[m][r];
      v = fmaf(noise_scale, zr[m][r], v);
      bad |= !(fabsf(v) <= 3.0e38f);
      xn[m][r] = v;
    }

  // ---- y = C x' + Y (:145): slab s = 4 m + r takes register r of tile m ----
  const XV_CONST_AS float* y0 = xv_cptr(P.T.y0) + (size_t)t * NO;
  xv_f32x4 ym[MO];   // this lane's rows 16 mo + 4 g + r
#pragma unroll
  for (int mo = 0; mo < MO; ++mo) ym[mo] = xv_f32x4{0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int mo = 0; mo < MO; ++mo)
      ym[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[mo][s], xn[s >> 2][s & 3], ym[mo], 0, 0, 0);
#pragma unroll
  for (int mo = 0; mo < MO; ++mo) xv_mfma_settle(ym[mo]);
#pragma unroll
  for (int mo = 0; mo < MO; ++mo)
#pragma unroll
    for (int r = 0; r < 4; ++r) ym[mo][r] = ym[mo][r] + xv_sel4(g, y0[16 * mo + r], y0[16 * mo + 4 + r], y0[16 * mo + 8 + r], y0[16 * mo + 12 + r]);   // :85
  if (__ballot(!(trk_in && rep_in)) != 0ull) {   // rare: no table, or an env stepped on outside it (auto-reset disabled)
    float full[NO];
    if (!trk_in) {
      linds_cmd<NO>(P, t, nf, trk_time, full);
#pragma unroll
      for (int mo = 0; mo < MO; ++mo)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ctr[mo][r] = xv_sel4(g, full[16 * mo + r], full[16 * mo + 4 + r], full[16 * mo + 8 + r], full[16 * mo + 12 + r]);
    }
    if (!rep_in) {
      linds_cmd<NO>(P, t, nf, rep_time, full);
#pragma unroll
      for (int mo = 0; mo < MO; ++mo)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          crep[mo][r] = xv_sel4(g, full[16 * mo + r], full[16 * mo + 4 + r], full[16 * mo + 8 + r], full[16 * mo + 12 + r]);
    }
  }
  // tracking error (:153) and observation scale (:154): chain g over this lane's own rows, the four lane groups of an
  // env combined by two xor-shuffles — the order linds_err / linds_sumsq define
  float pe = 0.0f, ps = 0.0f;
#pragma unroll
  for (int mo = 0; mo < MO; ++mo)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = (ym[mo][r] - ctr[mo][r]) * vld[mo][r];
      pe = fmaf(d, d, pe);
      ps = fmaf(ym[mo][r], ym[mo][r], ps);
    }
  float o_err = sqrtf(linds_quad_sum<MO>(pe));
  const float obs_scale = sqrtf(linds_quad_sum<MO>(ps));
  int o_term = ((o_err > 10.0f) || (obs_scale > 20.0f)) ? 1 : 0;   // :156
  float o_r = o_term ? -sc[2] : 0.0f;                         // :158-161
  float tmp = fmaf(-sc[3], o_err, sc[1]);
  tmp = fmaf(-sc[0], sa, tmp);
  o_r = fmaf(tmp, sc[5], o_r);                                // :163-164
  int o_trunc = (steps_new >= max_steps - 1) ? 1 : 0;         // :165

  // ---- which envs (re)start this call ----
  const bool skip = (mode == XV_AUTORESET_NEXT_STEP) && nr;   // the call after a done: reset only
  const bool done = !skip && (o_term || o_trunc);
  const bool do_reset = skip || (done && mode == XV_AUTORESET_SAME_STEP);
  int wrote_fobs = 0;
  xv_f32x4 fobs[MO];
#pragma unroll
  for (int mo = 0; mo < MO; ++mo) fobs[mo] = xv_f32x4{0, 0, 0, 0};
  if (skip) {
    o_r = 0.0f; o_term = 0; o_trunc = 0; bad = 0;
  } else {
    steps = steps_new;
    if (done && mode == XV_AUTORESET_NEXT_STEP) nr = 1;
  }
  if (__ballot(do_reset) != 0ull) {   // wave-uniform: the restarted envs take their initial state
    if (!INJECT) init_idx = linds_draw_init(P, gid, n_init);
    const int idx = init_idx < 0 ? 0 : (init_idx >= n_init ? n_init - 1 : init_idx);
    const float* x0 = P.T.init + ((size_t)t * P.NI + idx) * NS;
    xv_f32x4 xr[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      xr[m] = xn[m];
      if (do_reset) {
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[m][r] = x0[16 * m + 4 * g + r];   // :117
      }
    }
    float c0[NO], e0;
    linds_cmd_at<NO>(P, t, nf, 0, c0);                // :120-126
    xv_f32x4 yr[MO];
    if (P.rst_tab != nullptr) {                        // observation and error of initial_states[idx], tabulated
      const float4* row = reinterpret_cast<const float4*>(P.rst_tab + ((size_t)t * P.NI + idx) * (NO + 4));
#pragma unroll
      for (int mo = 0; mo < MO; ++mo) {
        const float4 v = row[4 * mo + g];
        yr[mo][0] = v.x; yr[mo][1] = v.y; yr[mo][2] = v.z; yr[mo][3] = v.w;
      }
      e0 = row[NO / 4].x;
    } else {
#pragma unroll
      for (int mo = 0; mo < MO; ++mo) yr[mo] = xv_f32x4{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int mo = 0; mo < MO; ++mo)
          yr[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[mo][s], xr[s >> 2][s & 3], yr[mo], 0, 0, 0);
#pragma unroll
      for (int mo = 0; mo < MO; ++mo) xv_mfma_settle(yr[mo]);
      float yfull[NO];
#pragma unroll
      for (int mo = 0; mo < MO; ++mo) {
#pragma unroll
        for (int r = 0; r < 4; ++r) yr[mo][r] = yr[mo][r] + xv_sel4(g, y0[16 * mo + r], y0[16 * mo + 4 + r], y0[16 * mo + 8 + r], y0[16 * mo + 12 + r]);
#pragma unroll
        for (int gs = 0; gs < 4; ++gs)
#pragma unroll
          for (int r = 0; r < 4; ++r) yfull[16 * mo + 4 * gs + r] = __shfl(yr[mo][r], n + 16 * gs);
      }
      e0 = linds_err<NO>(P, t, yfull, c0);
    }
    if (do_reset) {
      if (!skip) {
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) fobs[mo] = ym[mo];
        wrote_fobs = 1;
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) xn[m] = xr[m];
#pragma unroll
      for (int mo = 0; mo < MO; ++mo) {
        ym[mo] = yr[mo];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          crep[mo][r] = xv_sel4(g, c0[16 * mo + r], c0[16 * mo + 4 + r], c0[16 * mo + 8 + r], c0[16 * mo + 12 + r]);
      }
      o_err = e0;
      steps = 0;
      nr = 0;
    }
  }
  if (skip) bad = 0;

  // ---- stores ----
  if (valid) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) P.x[(size_t)(16 * m + 4 * g + r) * NSL + es] = xn[m][r];
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) {   // each lane stores its own 16-byte quarter of the rows
      const size_t ro = (size_t)e * NO + 16 * mo + 4 * g;
      *reinterpret_cast<float4*>(io.obs + ro) = make_float4(ym[mo][0], ym[mo][1], ym[mo][2], ym[mo][3]);
      *reinterpret_cast<float4*>(io.cmd + ro) = make_float4(crep[mo][0], crep[mo][1], crep[mo][2], crep[mo][3]);
      if (io.final_obs)
        *reinterpret_cast<float4*>(io.final_obs + ro) =
            wrote_fobs ? make_float4(fobs[mo][0], fobs[mo][1], fobs[mo][2], fobs[mo][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (g == 0) {
      P.steps[es] = steps;
      P.need_reset[es] = (uint8_t)nr;
      io.reward[e] = o_r;
      io.error[e] = o_err;
      io.terminated[e] = (uint8_t)o_term;
      io.truncated[e] = (uint8_t)o_trunc;
    }
  }
  if (bad && valid) atomicOr(P.err, (uint32_t)XV_DEVERR_NONFINITE);
}

// ------------------------------------------------------------------------------------------------
// Fused roll-out: T steps of the tile in one launch, SAME_STEP auto-reset, free-running noise.  The task's operand
// fragments are loaded once and the state never leaves the registers the matrix unit wrote it to (see the k order
// above); per step only the action row and the two command rows come in and the outputs go out.  Step t draws with
// tick0 + t, so the result equals T calls of xv_linds_step bit for bit (tested).
// ------------------------------------------------------------------------------------------------
struct LinDSRolloutIO {
  const float* action;      // [T][n_env][NA]
  float* obs;               // [T][n_env][NO]
  float* reward;            // [T][n_env]
  uint8_t* terminated;
  uint8_t* truncated;
  float* cmd;               // [T][n_env][NO]  nullable
  float* error;             // [T][n_env]      nullable
  float* final_obs;         // [T][n_env][NO]  nullable
};

template <int NS, int NA, int NO>
__global__ __launch_bounds__(256) void linds_rollout_mfma_kernel(LinDSArgs P, LinDSRolloutIO io, int T) {
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int tile0 = wave * 16;
  if (tile0 >= P.n_slot) return;   // wave-uniform
  const int N = P.n_env, NSL = P.n_slot;
  const int n = lane & 15, g = lane >> 4;
  constexpr int MT = NS / 16, MO = NO / 16, KS = NS / 4, KA = NA / 4;
  const int es = tile0 + n < NSL ? tile0 + n : NSL - 1;
  int e_raw = tile0 + n;
  if (P.slot_env != nullptr) e_raw = P.slot_env[es];
  const bool valid = tile0 + n < NSL && e_raw >= 0 && e_raw < N;
  const int e = valid ? e_raw : 0;
  const int t = __builtin_amdgcn_readfirstlane(P.tile_task ? P.tile_task[wave] : P.env_task[tile0]);
  const uint64_t gid = P.gid_base + (uint64_t)e;
  const XV_CONST_AS float* sc = xv_cptr(P.T.scal) + (size_t)t * 8;
  const XV_CONST_AS int32_t* in = xv_cptr(P.T.ints) + (size_t)t * 4;
  const int max_steps = in[0], delay = in[1], n_init = in[2], nf = in[3];
  const float noise_scale = sc[4];

  // ---- once per launch: the task's operand fragments and the state ----
  const float* phiT = P.T.phiT + (size_t)t * NS * NS;
  const float* gamT = P.T.gamT + (size_t)t * NA * NS;
  const float* cT = P.T.cT + (size_t)t * NS * NO;
  const float* xtv = P.T.xt + (size_t)t * NS;
  float pa[MT][KS], ga[MT][KA], ca[MO][KS], xtr[MT][4], y0r[MO][4];
  xv_f32x4 xs[MT];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const int k = 16 * (kk >> 2) + 4 * g + (kk & 3);
    xs[kk >> 2][kk & 3] = P.x[(size_t)k * NSL + es];
#pragma unroll
    for (int m = 0; m < MT; ++m) pa[m][kk] = phiT[k * NS + 16 * m + n];
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) ca[mo][kk] = cT[k * NO + 16 * mo + n];
  }
#pragma unroll
  for (int kk = 0; kk < KA; ++kk)
#pragma unroll
    for (int m = 0; m < MT; ++m) ga[m][kk] = gamT[(4 * kk + g) * NS + 16 * m + n];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) xtr[m][r] = xtv[16 * m + 4 * g + r];
  float vld[MO][4];
  {
    const float* vp = P.T.valid + (size_t)t * NO;
    const float* yp = P.T.y0 + (size_t)t * NO;
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) {
      const float4 v = *reinterpret_cast<const float4*>(vp + 16 * mo + 4 * g);
      const float4 w = *reinterpret_cast<const float4*>(yp + 16 * mo + 4 * g);
      vld[mo][0] = v.x; vld[mo][1] = v.y; vld[mo][2] = v.z; vld[mo][3] = v.w;
      y0r[mo][0] = w.x; y0r[mo][1] = w.y; y0r[mo][2] = w.z; y0r[mo][3] = w.w;
    }
  }
  int steps = P.steps[es];
  int bad_any = 0;

  for (int ts = 0; ts < T; ++ts) {
    const uint64_t tick = P.tick + (uint64_t)ts;
    const size_t ob = (size_t)ts * N + e;          // this step's slot of the [T][n_env] outputs
    float a_raw[NA];
    {
      const float4* a4 = reinterpret_cast<const float4*>(io.action + ob * NA);
#pragma unroll
      for (int q = 0; q < NA / 4; ++q) {
        const float4 v = a4[q];
        a_raw[4 * q] = v.x; a_raw[4 * q + 1] = v.y; a_raw[4 * q + 2] = v.z; a_raw[4 * q + 3] = v.w;
      }
    }
    const int steps_new = steps + 1;                            // :147
    const int trk_time = steps_new - 1 - delay, rep_time = steps_new;
    const int trk_idx = trk_time - P.ct_tmin, rep_idx = rep_time - P.ct_tmin;
    const bool trk_in = P.cmd_tab != nullptr && trk_idx >= 0 && trk_idx < P.ct_len;
    const bool rep_in = P.cmd_tab != nullptr && rep_idx >= 0 && rep_idx < P.ct_len;
    float ctr[MO][4], crep[MO][4];
    if (P.cmd_tab != nullptr) {
      const float4* p = reinterpret_cast<const float4*>(P.cmd_tab + ((size_t)t * P.ct_len + (trk_in ? trk_idx : 0)) * NO);
      const float4* pr = reinterpret_cast<const float4*>(P.cmd_tab + ((size_t)t * P.ct_len + (rep_in ? rep_idx : 0)) * NO);
#pragma unroll
      for (int mo = 0; mo < MO; ++mo) {
        const float4 u = p[4 * mo + g], v = pr[4 * mo + g];
        ctr[mo][0] = u.x; ctr[mo][1] = u.y; ctr[mo][2] = u.z; ctr[mo][3] = u.w;
        crep[mo][0] = v.x; crep[mo][1] = v.y; crep[mo][2] = v.z; crep[mo][3] = v.w;
      }
    }
    float zr[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const xv_u32x4 w = xv_env_draw(P.seed, gid, tick, XV_DRAW_NOISE + (uint32_t)(4 * m + g));
      xv_box_muller_fast(w.x, w.y, &zr[m][0], &zr[m][1]);
      xv_box_muller_fast(w.z, w.w, &zr[m][2], &zr[m][3]);
    }
    // ---- x' = Phi x + Gamma act ----
    xv_f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = xv_f32x4{0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[m][kk], xs[kk >> 2][kk & 3], acc[m], 0, 0, 0);
    float sa = 0.0f;
#pragma unroll
    for (int k = 0; k < NA; ++k) sa = fmaf(a_raw[k], a_raw[k], sa);
#pragma unroll
    for (int kk = 0; kk < KA; ++kk) {
      const float ar = xv_sel4(g, a_raw[4 * kk], a_raw[4 * kk + 1], a_raw[4 * kk + 2], a_raw[4 * kk + 3]);
      const float b = ar < -1.0f ? -1.0f : (ar > 1.0f ? 1.0f : ar);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[m][kk], b, acc[m], 0, 0, 0);
    }
    xv_f32x4 xn[MT];
    int bad = 0;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[m][r] + xtr[m][r];
        v = fmaf(noise_scale, zr[m][r], v);
        bad |= !(fabsf(v) <= 3.0e38f);
        xn[m][r] = v;
      }
    // ---- y = C x' + Y ----
    xv_f32x4 ym[MO];
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) ym[mo] = xv_f32x4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int mo = 0; mo < MO; ++mo)
        ym[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[mo][s], xn[s >> 2][s & 3], ym[mo], 0, 0, 0);
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) xv_mfma_settle(ym[mo]);
#pragma unroll
    for (int mo = 0; mo < MO; ++mo)
#pragma unroll
      for (int r = 0; r < 4; ++r) ym[mo][r] = ym[mo][r] + y0r[mo][r];
    if (__ballot(!(trk_in && rep_in)) != 0ull) {
      float full[NO];
      if (!trk_in) {
        linds_cmd<NO>(P, t, nf, trk_time, full);
#pragma unroll
        for (int mo = 0; mo < MO; ++mo)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            ctr[mo][r] = xv_sel4(g, full[16 * mo + r], full[16 * mo + 4 + r], full[16 * mo + 8 + r], full[16 * mo + 12 + r]);
      }
      if (!rep_in) {
        linds_cmd<NO>(P, t, nf, rep_time, full);
#pragma unroll
        for (int mo = 0; mo < MO; ++mo)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            crep[mo][r] = xv_sel4(g, full[16 * mo + r], full[16 * mo + 4 + r], full[16 * mo + 8 + r], full[16 * mo + 12 + r]);
      }
    }
    float pe = 0.0f, ps = 0.0f;
#pragma unroll
    for (int mo = 0; mo < MO; ++mo)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = (ym[mo][r] - ctr[mo][r]) * vld[mo][r];
        pe = fmaf(d, d, pe);
        ps = fmaf(ym[mo][r], ym[mo][r], ps);
      }
    float o_err = sqrtf(linds_quad_sum<MO>(pe));
    const float obs_scale = sqrtf(linds_quad_sum<MO>(ps));
    const int o_term = ((o_err > 10.0f) || (obs_scale > 20.0f)) ? 1 : 0;
    float o_r = o_term ? -sc[2] : 0.0f;
    float tmp = fmaf(-sc[3], o_err, sc[1]);
    tmp = fmaf(-sc[0], sa, tmp);
    o_r = fmaf(tmp, sc[5], o_r);
    const int o_trunc = (steps_new >= max_steps - 1) ? 1 : 0;
    const bool do_reset = o_term || o_trunc;
    steps = steps_new;
    int wrote_fobs = 0;
    xv_f32x4 fobs[MO];
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) fobs[mo] = xv_f32x4{0, 0, 0, 0};
    if (__ballot(do_reset) != 0ull) {
      const xv_u32x4 v = xv_env_draw(P.seed, gid, tick, XV_DRAW_RESET);
      int idx = (int)(xv_u53(v.x, v.y) * (double)n_init);
      idx = idx < n_init ? idx : n_init - 1;
      idx = idx < 0 ? 0 : idx;
      const float* x0 = P.T.init + ((size_t)t * P.NI + idx) * NS;
      xv_f32x4 xr[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        xr[m] = xn[m];
        if (do_reset) {
#pragma unroll
          for (int r = 0; r < 4; ++r) xr[m][r] = x0[16 * m + 4 * g + r];
        }
      }
      float c0[NO], e0;
      linds_cmd_at<NO>(P, t, nf, 0, c0);
      xv_f32x4 yr[MO];
      if (P.rst_tab != nullptr) {
        const float4* row = reinterpret_cast<const float4*>(P.rst_tab + ((size_t)t * P.NI + idx) * (NO + 4));
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) {
          const float4 q4 = row[4 * mo + g];
          yr[mo][0] = q4.x; yr[mo][1] = q4.y; yr[mo][2] = q4.z; yr[mo][3] = q4.w;
        }
        e0 = row[NO / 4].x;
      } else {
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) yr[mo] = xv_f32x4{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int mo = 0; mo < MO; ++mo)
            yr[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[mo][s], xr[s >> 2][s & 3], yr[mo], 0, 0, 0);
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) xv_mfma_settle(yr[mo]);
        float yfull[NO];
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) {
#pragma unroll
          for (int r = 0; r < 4; ++r) yr[mo][r] = yr[mo][r] + y0r[mo][r];
#pragma unroll
          for (int gs = 0; gs < 4; ++gs)
#pragma unroll
            for (int r = 0; r < 4; ++r) yfull[16 * mo + 4 * gs + r] = __shfl(yr[mo][r], n + 16 * gs);
        }
        e0 = linds_err<NO>(P, t, yfull, c0);
      }
      if (do_reset) {
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) fobs[mo] = ym[mo];
        wrote_fobs = 1;
#pragma unroll
        for (int m = 0; m < MT; ++m) xn[m] = xr[m];
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) {
          ym[mo] = yr[mo];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            crep[mo][r] = xv_sel4(g, c0[16 * mo + r], c0[16 * mo + 4 + r], c0[16 * mo + 8 + r], c0[16 * mo + 12 + r]);
        }
        o_err = e0;
        steps = 0;
      }
    }
    bad_any |= bad;
    if (valid) {
#pragma unroll
      for (int mo = 0; mo < MO; ++mo) {
        const size_t ro = ob * NO + 16 * mo + 4 * g;
        *reinterpret_cast<float4*>(io.obs + ro) = make_float4(ym[mo][0], ym[mo][1], ym[mo][2], ym[mo][3]);
        if (io.cmd) *reinterpret_cast<float4*>(io.cmd + ro) = make_float4(crep[mo][0], crep[mo][1], crep[mo][2], crep[mo][3]);
        if (io.final_obs)
          *reinterpret_cast<float4*>(io.final_obs + ro) =
              wrote_fobs ? make_float4(fobs[mo][0], fobs[mo][1], fobs[mo][2], fobs[mo][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (g == 0) {
        io.reward[ob] = o_r;
        if (io.error) io.error[ob] = o_err;
        io.terminated[ob] = (uint8_t)o_term;
        io.truncated[ob] = (uint8_t)o_trunc;
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) xs[m] = xn[m];
  }
  if (valid) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) P.x[(size_t)(16 * m + 4 * g + r) * NSL + es] = xs[m][r];
    if (g == 0) {
      P.steps[es] = steps;
      P.need_reset[es] = 0;
    }
  }
  if (bad_any && valid) atomicOr(P.err, (uint32_t)XV_DEVERR_NONFINITE);
}

// two entry points over the same body: with 16 observation rows the step fits 128 registers and is capped there
// (4 waves per SIMD: the kernel is latency-bound); with 32 rows the cap would spill, so it runs at 2-3 waves per SIMD
template <int NS, int NA, int NO, bool INJECT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
void linds_step_mfma_kernel(LinDSArgs P, LinDSStepIO io, int mode) {
  linds_step_mfma_body<NS, NA, NO, INJECT>(P, io, mode);
}
template <int NS, int NA, int NO, bool INJECT>
__global__ __launch_bounds__(256) void linds_step_mfma_wide_kernel(LinDSArgs P, LinDSStepIO io, int mode) {
  linds_step_mfma_body<NS, NA, NO, INJECT>(P, io, mode);
}

// every aligned group of 16 envs shares one task?  (else the engine builds its slot layout)
__global__ __launch_bounds__(256) void linds_check_tiles_kernel(const int32_t* env_task, int n_env, int* not_uniform) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_env) return;
  if (env_task[i] != env_task[i & ~15]) atomicOr(not_uniform, 1);
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" int xv_linds_create(xv_engine* e, int n_env, int n_task, int NS, int NA, int NO, int NI,
                               const xv_linds_tables* tables, const int32_t* env_task, xv_linds** out) {
  XV_CHECK_ARG(out != nullptr);
  *out = nullptr;
  XV_CHECK_ARG(e != nullptr && tables != nullptr && env_task != nullptr);
  XV_CHECK_ARG(n_env > 0 && n_task > 0 && NI > 0);
  XV_CHECK_ARG((NS == 16 || NS == 32) && (NA == 8 || NA == 16) && (NO == 16 || NO == 32));
  XV_CHECK_ARG(tables->phiT && tables->gamT && tables->cT && tables->xt && tables->y0 && tables->valid &&
               tables->cmd0 && tables->four_coef && tables->four_omega && tables->four_period &&
               tables->scal && tables->ints && tables->init);
  XV_HIP(hipSetDevice(e->device));
  xv_linds* h = new (std::nothrow) xv_linds();
  if (!h) {
    xv_set_error("xv_linds_create: out of host memory");
    return XV_ERR_NOMEM;
  }
  h->eng = e;
  LinDSArgs& a = h->a;
  a.T = *tables;
  a.env_task = env_task;
  a.n_env = n_env; a.n_task = n_task; a.NS = NS; a.NA = NA; a.NO = NO; a.NI = NI;
  a.err = e->d_err;
  a.seed = e->seed; a.gid_base = e->env_id_base; a.tick = 0;
  a.x = nullptr; a.steps = nullptr; a.need_reset = nullptr;
  a.slot_env = nullptr; a.env_slot = nullptr; a.tile_task = nullptr; a.n_slot = n_env;
  h->d_slot_env = nullptr; h->d_env_slot = nullptr; h->d_tile_task = nullptr;
  hipError_t m = hipSuccess;
  {
    int* d_flag = nullptr;
    int h_flag = 1;
    m = hipMalloc(&d_flag, sizeof(int));
    if (m == hipSuccess) m = hipMemsetAsync(d_flag, 0, sizeof(int), e->stream);
    if (m == hipSuccess) {
      hipLaunchKernelGGL(linds_check_tiles_kernel, dim3(xv_div_up(n_env, 256)), dim3(256), 0, e->stream, env_task,
                         n_env, d_flag);
      m = hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, e->stream);
    }
    if (m == hipSuccess) m = hipStreamSynchronize(e->stream);
    if (d_flag) (void)hipFree(d_flag);
    h->tiles_uniform = (h_flag == 0);
    h->path = XV_LINDS_PATH_AUTO;
  }
  if (m == hipSuccess && !h->tiles_uniform) {
    // slot layout: stable counting sort of the envs by task, every task's envs packed into whole 16-slot tiles
    std::vector<int32_t> et((size_t)n_env);
    m = hipMemcpy(et.data(), env_task, sizeof(int32_t) * (size_t)n_env, hipMemcpyDeviceToHost);
    if (m == hipSuccess) {
      std::vector<int64_t> first((size_t)n_task + 1, 0);
      bool ok = true;
      for (int i = 0; i < n_env; ++i) {
        if (et[i] < 0 || et[i] >= n_task) { ok = false; break; }
        first[(size_t)et[i] + 1] += 1;
      }
      if (!ok) {
        xv_set_error("xv_linds_create: env_task entry outside [0, n_task)");
        delete h;
        return XV_ERR_INVALID;
      }
      int64_t n_slot = 0;
      std::vector<int64_t> base((size_t)n_task);
      for (int t = 0; t < n_task; ++t) {
        base[t] = n_slot;
        n_slot += (first[(size_t)t + 1] + 15) / 16 * 16;
      }
      if (n_slot > (int64_t)1 << 30) {
        xv_set_error("xv_linds_create: slot layout too large");
        delete h;
        return XV_ERR_UNSUPPORTED;
      }
      std::vector<int32_t> slot_env((size_t)n_slot, -1), env_slot((size_t)n_env), tile_task((size_t)(n_slot / 16));
      std::vector<int64_t> fill(base);
      for (int i = 0; i < n_env; ++i) {
        const int64_t sl = fill[et[i]]++;
        slot_env[(size_t)sl] = i;
        env_slot[i] = (int32_t)sl;
      }
      for (int t = 0; t < n_task; ++t)
        for (int64_t q = base[t] / 16; q < (t + 1 < n_task ? base[t + 1] : n_slot) / 16; ++q) tile_task[(size_t)q] = t;
      a.n_slot = (int)n_slot;
      m = hipMalloc(&h->d_slot_env, sizeof(int32_t) * (size_t)n_slot);
      if (m == hipSuccess) m = hipMalloc(&h->d_env_slot, sizeof(int32_t) * (size_t)n_env);
      if (m == hipSuccess) m = hipMalloc(&h->d_tile_task, sizeof(int32_t) * (size_t)(n_slot / 16));
      if (m == hipSuccess) m = hipMemcpy(h->d_slot_env, slot_env.data(), sizeof(int32_t) * (size_t)n_slot, hipMemcpyHostToDevice);
      if (m == hipSuccess) m = hipMemcpy(h->d_env_slot, env_slot.data(), sizeof(int32_t) * (size_t)n_env, hipMemcpyHostToDevice);
      if (m == hipSuccess) m = hipMemcpy(h->d_tile_task, tile_task.data(), sizeof(int32_t) * (size_t)(n_slot / 16), hipMemcpyHostToDevice);
      a.slot_env = h->d_slot_env; a.env_slot = h->d_env_slot; a.tile_task = h->d_tile_task;
    }
  }
  const size_t nsl = (size_t)a.n_slot;
  if (m == hipSuccess) m = hipMalloc(&a.x, sizeof(float) * (size_t)NS * nsl);
  if (m == hipSuccess) m = hipMalloc(&a.steps, sizeof(int32_t) * nsl);
  if (m == hipSuccess) m = hipMalloc(&a.need_reset, nsl);
  if (m == hipSuccess) m = hipMemsetAsync(a.x, 0, sizeof(float) * (size_t)NS * nsl, e->stream);
  if (m == hipSuccess) m = hipMemsetAsync(a.steps, 0, sizeof(int32_t) * nsl, e->stream);
  if (m == hipSuccess) m = hipMemsetAsync(a.need_reset, 1, nsl, e->stream);
  if (m != hipSuccess) {
    xv_set_error("xv_linds_create: device allocation failed: %s", hipGetErrorString(m));
    if (a.x) (void)hipFree(a.x);
    if (a.steps) (void)hipFree(a.steps);
    if (a.need_reset) (void)hipFree(a.need_reset);
    if (h->d_slot_env) (void)hipFree(h->d_slot_env);
    if (h->d_env_slot) (void)hipFree(h->d_env_slot);
    if (h->d_tile_task) (void)hipFree(h->d_tile_task);
    delete h;
    return XV_ERR_HIP;
  }
  // command table: [n_task][max_steps_max + 2 + delay_max][NO] floats, within a 2-GiB budget
  a.cmd_tab = nullptr; a.ct_len = 0; a.ct_tmin = 0; a.rst_tab = nullptr;
  h->cmd_tab = nullptr; h->rst_tab = nullptr;
  {
    int* d2 = nullptr;
    int h2[2] = {0, 0};
    XV_HIP(hipMalloc(&d2, 2 * sizeof(int)));
    XV_HIP(hipMemsetAsync(d2, 0, 2 * sizeof(int), e->stream));
    hipLaunchKernelGGL(linds_max_ints_kernel, dim3(xv_div_up(n_task, 256)), dim3(256), 0, e->stream, tables->ints, n_task, d2);
    XV_HIP(hipMemcpyAsync(h2, d2, 2 * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    XV_HIP(hipStreamSynchronize(e->stream));
    XV_HIP(hipFree(d2));
    const long long len = (long long)h2[0] + 2 + h2[1];
    const unsigned long long bytes = (unsigned long long)n_task * (unsigned long long)len * NO * sizeof(float);
    float* tab = nullptr;
    if (h2[0] > 0 && h2[1] >= 0 && len < (1 << 20) && bytes <= (2ull << 30) && hipMalloc(&tab, bytes) == hipSuccess) {
      a.ct_len = (int)len;
      a.ct_tmin = -(1 + h2[1]);
      const size_t n = (size_t)n_task * a.ct_len;
      if (NO == 16)
        hipLaunchKernelGGL((linds_build_cmd_tab_kernel<16>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, a, tab);
      else
        hipLaunchKernelGGL((linds_build_cmd_tab_kernel<32>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, a, tab);
      a.cmd_tab = tab;
      h->cmd_tab = tab;
    } else {
      (void)hipGetLastError();
    }
  }
  {   // reset table: [n_task][NI][NO + 4] floats
    float* rt = nullptr;
    const size_t nrow = (size_t)n_task * NI;
    if (hipMalloc(&rt, nrow * (NO + 4) * sizeof(float)) == hipSuccess) {
      const dim3 grid((unsigned)((nrow + 255) / 256)), block(256);
#define LINDS_RST(NS_, NO_) hipLaunchKernelGGL((linds_build_reset_tab_kernel<NS_, NO_>), grid, block, 0, e->stream, a, rt)
      if (NS == 16 && NO == 16) LINDS_RST(16, 16);
      else if (NS == 16) LINDS_RST(16, 32);
      else if (NO == 16) LINDS_RST(32, 16);
      else LINDS_RST(32, 32);
#undef LINDS_RST
      a.rst_tab = rt;
      h->rst_tab = rt;
    } else {
      (void)hipGetLastError();
    }
  }
  XV_LAUNCH_CHECK();
  *out = h;
  return XV_OK;
}

extern "C" int xv_linds_set_path(xv_linds* h, int path) {
  XV_CHECK_ARG(h != nullptr && path >= 0 && path <= 2);
  h->path = path;   // both kernels serve any env -> task map (the MFMA kernel through the engine's slot layout)
  return XV_OK;
}

extern "C" int xv_linds_set_command_table(xv_linds* h, int enable) {
  XV_CHECK_ARG(h != nullptr);
  if (enable && !h->cmd_tab) {
    xv_set_error("xv_linds_set_command_table: no table was built (over the 2-GiB budget or allocation failed)");
    return XV_ERR_UNSUPPORTED;
  }
  h->a.cmd_tab = enable ? h->cmd_tab : nullptr;
  h->a.rst_tab = enable ? h->rst_tab : nullptr;
  return XV_OK;
}

extern "C" int xv_linds_destroy(xv_linds* h) {
  if (!h) return XV_OK;
  (void)hipSetDevice(h->eng->device);
  (void)hipStreamSynchronize(h->eng->stream);
  (void)hipFree(h->a.x);
  (void)hipFree(h->a.steps);
  (void)hipFree(h->a.need_reset);
  if (h->d_slot_env) (void)hipFree(h->d_slot_env);
  if (h->d_env_slot) (void)hipFree(h->d_env_slot);
  if (h->d_tile_task) (void)hipFree(h->d_tile_task);
  if (h->cmd_tab) (void)hipFree(h->cmd_tab);
  if (h->rst_tab) (void)hipFree(h->rst_tab);
  delete h;
  return XV_OK;
}

static inline void linds_bind_rng(xv_linds* h, uint64_t ticks) {
  h->a.seed = h->eng->seed;
  h->a.gid_base = h->eng->env_id_base;
  h->a.tick = h->eng->tick;
  h->eng->tick += ticks;
}

#define LINDS_DISPATCH(FN, ...)                                                                      \
  do {                                                                                               \
    const int key = (h->a.NS == 32 ? 4 : 0) | (h->a.NA == 16 ? 2 : 0) | (h->a.NO == 32 ? 1 : 0);    \
    switch (key) {                                                                                   \
      case 0: FN(16, 8, 16, __VA_ARGS__); break;                                                     \
      case 1: FN(16, 8, 32, __VA_ARGS__); break;                                                     \
      case 2: FN(16, 16, 16, __VA_ARGS__); break;                                                    \
      case 3: FN(16, 16, 32, __VA_ARGS__); break;                                                    \
      case 4: FN(32, 8, 16, __VA_ARGS__); break;                                                     \
      case 5: FN(32, 8, 32, __VA_ARGS__); break;                                                     \
      case 6: FN(32, 16, 16, __VA_ARGS__); break;                                                    \
      default: FN(32, 16, 32, __VA_ARGS__); break;                                                   \
    }                                                                                                \
  } while (0)

template <bool INJECT>
static int linds_launch_step(xv_linds* h, const LinDSStepIO& io, int mode) {
  const dim3 block(256);
  const bool mfma = h->path != XV_LINDS_PATH_SCALAR;
  if (mfma) {
    const dim3 grid(xv_div_up(xv_div_up(h->a.n_slot, 16), 4));   // one wave per 16-slot tile, 4 tiles per block
#define LINDS_STEP_M(NS_, NA_, NO_, dummy)                                                                            \
  do {                                                                                                                \
    if (NO_ == 16)                                                                                                    \
      hipLaunchKernelGGL((linds_step_mfma_kernel<NS_, NA_, 16, INJECT>), grid, block, 0, h->eng->stream, h->a, io, mode); \
    else                                                                                                              \
      hipLaunchKernelGGL((linds_step_mfma_wide_kernel<NS_, NA_, 32, INJECT>), grid, block, 0, h->eng->stream, h->a, io, mode); \
  } while (0)
    LINDS_DISPATCH(LINDS_STEP_M, 0);
#undef LINDS_STEP_M
  } else {
    const dim3 grid(xv_div_up(h->a.n_env, 256));
#define LINDS_STEP(NS_, NA_, NO_, dummy) \
  hipLaunchKernelGGL((linds_step_kernel<NS_, NA_, NO_, INJECT>), grid, block, 0, h->eng->stream, h->a, io, mode)
    LINDS_DISPATCH(LINDS_STEP, 0);
#undef LINDS_STEP
  }
  XV_LAUNCH_CHECK();
  return XV_OK;
}

template <bool INJECT>
static int linds_launch_reset(xv_linds* h, const uint8_t* mask, const int32_t* init_index, float* obs,
                              float* cmd, float* error) {
  const dim3 grid(xv_div_up(h->a.n_env, 256)), block(256);
#define LINDS_RESET(NS_, NA_, NO_, dummy)                                                              \
  hipLaunchKernelGGL((linds_reset_kernel<NS_, NO_, INJECT>), grid, block, 0, h->eng->stream, h->a, mask, \
                     init_index, obs, cmd, error)
  LINDS_DISPATCH(LINDS_RESET, 0);
#undef LINDS_RESET
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_linds_reset(xv_linds* h, const uint8_t* mask, float* obs, float* cmd, float* error) {
  XV_CHECK_ARG(h != nullptr);
  linds_bind_rng(h, 1);
  return linds_launch_reset<false>(h, mask, nullptr, obs, cmd, error);
}

extern "C" int xv_linds_reset_injected(xv_linds* h, const uint8_t* mask, const int32_t* init_index,
                                       float* obs, float* cmd, float* error) {
  XV_CHECK_ARG(h != nullptr && init_index != nullptr);
  linds_bind_rng(h, 0);
  return linds_launch_reset<true>(h, mask, init_index, obs, cmd, error);
}

extern "C" int xv_linds_step(xv_linds* h, const float* action, float* obs, float* reward, uint8_t* terminated,
                             uint8_t* truncated, float* cmd, float* error, float* final_obs,
                             int autoreset_mode) {
  XV_CHECK_ARG(h && action && obs && reward && terminated && truncated && cmd && error);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  linds_bind_rng(h, 1);
  LinDSStepIO io{action, nullptr, nullptr, obs, reward, terminated, truncated, cmd, error, final_obs};
  return linds_launch_step<false>(h, io, autoreset_mode);
}

extern "C" int xv_linds_step_injected(xv_linds* h, const float* action, const float* z,
                                      const int32_t* init_index, float* obs, float* reward,
                                      uint8_t* terminated, uint8_t* truncated, float* cmd, float* error,
                                      float* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && action && z && init_index && obs && reward && terminated && truncated && cmd && error);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  linds_bind_rng(h, 0);
  LinDSStepIO io{action, z, init_index, obs, reward, terminated, truncated, cmd, error, final_obs};
  return linds_launch_step<true>(h, io, autoreset_mode);
}

// state <-> caller order when the engine keeps it in slot order
template <bool TO_ENV>
__global__ __launch_bounds__(256) void linds_permute_state_kernel(LinDSArgs P, float* x, int32_t* steps, uint8_t* need_reset) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  const int N = P.n_env, NSL = P.n_slot, si = P.env_slot[i];
  if (x) {
    for (int k = 0; k < P.NS; ++k) {
      if (TO_ENV) x[(size_t)k * N + i] = P.x[(size_t)k * NSL + si];
      else P.x[(size_t)k * NSL + si] = x[(size_t)k * N + i];
    }
  }
  if (steps) { if (TO_ENV) steps[i] = P.steps[si]; else P.steps[si] = steps[i]; }
  if (need_reset) { if (TO_ENV) need_reset[i] = P.need_reset[si]; else P.need_reset[si] = need_reset[i]; }
}

extern "C" int xv_linds_rollout(xv_linds* h, int T, const float* action, float* obs, float* reward, uint8_t* terminated,
                                uint8_t* truncated, float* cmd, float* error, float* final_obs) {
  XV_CHECK_ARG(h && action && obs && reward && terminated && truncated && T > 0);
  linds_bind_rng(h, (uint64_t)T);
  LinDSRolloutIO io{action, obs, reward, terminated, truncated, cmd, error, final_obs};
  const dim3 block(256), grid(xv_div_up(xv_div_up(h->a.n_slot, 16), 4));
#define LINDS_ROLL(NS_, NA_, NO_, dummy) \
  hipLaunchKernelGGL((linds_rollout_mfma_kernel<NS_, NA_, NO_>), grid, block, 0, h->eng->stream, h->a, io, T)
  LINDS_DISPATCH(LINDS_ROLL, 0);
#undef LINDS_ROLL
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_linds_get_state(xv_linds* h, float* x, int32_t* steps, uint8_t* need_reset) {
  XV_CHECK_ARG(h != nullptr);
  if (h->a.env_slot != nullptr) {
    hipLaunchKernelGGL((linds_permute_state_kernel<true>), dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                       h->a, x, steps, need_reset);
    XV_LAUNCH_CHECK();
    return XV_OK;
  }
  const size_t n = (size_t)h->a.n_env;
  if (x) XV_HIP(hipMemcpyAsync(x, h->a.x, n * h->a.NS * 4, hipMemcpyDeviceToDevice, h->eng->stream));
  if (steps) XV_HIP(hipMemcpyAsync(steps, h->a.steps, n * 4, hipMemcpyDeviceToDevice, h->eng->stream));
  if (need_reset) XV_HIP(hipMemcpyAsync(need_reset, h->a.need_reset, n, hipMemcpyDeviceToDevice, h->eng->stream));
  return XV_OK;
}

extern "C" int xv_linds_set_state(xv_linds* h, const float* x, const int32_t* steps, const uint8_t* need_reset) {
  XV_CHECK_ARG(h != nullptr);
  if (h->a.env_slot != nullptr) {
    hipLaunchKernelGGL((linds_permute_state_kernel<false>), dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                       h->a, const_cast<float*>(x), const_cast<int32_t*>(steps), const_cast<uint8_t*>(need_reset));
    XV_LAUNCH_CHECK();
    return XV_OK;
  }
  const size_t n = (size_t)h->a.n_env;
  if (x) XV_HIP(hipMemcpyAsync(h->a.x, x, n * h->a.NS * 4, hipMemcpyDeviceToDevice, h->eng->stream));
  if (steps) XV_HIP(hipMemcpyAsync(h->a.steps, steps, n * 4, hipMemcpyDeviceToDevice, h->eng->stream));
  if (need_reset) XV_HIP(hipMemcpyAsync(h->a.need_reset, need_reset, n, hipMemcpyDeviceToDevice, h->eng->stream));
  return XV_OK;
}
