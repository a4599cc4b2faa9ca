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
// The product path is the MFMA kernel (one wave per tile of 16 envs that share a task, below).  The scalar kernel — one
// lane per env, task matrices as scalar-cache broadcast operands inside a waterfall over the wave's tasks — is kept as
// the independent second implementation the parity tests compare it with (XV_LINDS_PATH_SCALAR; never chosen by AUTO).
//
// Engine-owned layouts (round 3; everything the caller sees stays in the caller's env order):
//   state  x:   "fragment tiles" — float4 unit ((tile * MT + m) * 64 + lane) holds components 16 m + 4 g + r (r = 0..3)
//               of slot 16 tile + n, lane = 16 g + n: exactly the four accumulator registers lane (n, g) of the matrix
//               kernel receives for M-tile m, which are also its B operands of the next product.  A step reads and
//               writes the state with MT coalesced 16-byte accesses per lane (linds_xidx for everybody else).
//   sn:         one word per slot: steps | need_reset << 31.
//   frag:       the task matrices re-arranged at create time into the A-operand fragments of the three products, a
//               float4 list per lane: [task][q][lane] — 7 coalesced 16-byte loads per lane at (32, 8, 16) instead of
//               28 dword loads with their address arithmetic.
//   tvec:       [task][NS + 2 NO] = Xt | Y | target_valid, read as float4 quarters.
#include <vector>

#include "philox.h"
#include "xv_common.h"
#include "xv_hand.h"

struct LinDSArgs {
  xv_linds_tables T;
  const int32_t* env_task;
  float* x;            // fragment tiles, see above
  int32_t* sn;         // [n_tile * 16]  steps | need_reset << 31
  uint32_t* err;
  int n_env, n_task, NS, NA, NO, NI;
  uint64_t seed, gid_base, tick;
  const uint64_t* tick_dev;   // device tick mode of the engine: the launch tick is *tick_dev + tick (xv_launch_tick)
  uint32_t* hand;             // [waves] HAND kernels only (mixed.hip): tile hand-off words, xv_hand.h
  // engine-built command table (nullptr if it would not fit the budget): cmd_tab[task][tt - ct_tmin][NO] holds
  // get_inner_cmd at integer time tt, already multiplied by target_valid, for tt in [ct_tmin, ct_tmin + ct_len)
  // Rows hold the first ct_w columns only (a multiple of 4): every column past the last one whose target_valid is
  // non-zero in some task is identically zero, so with observation_dim 8 padded to 16 a row is 32 bytes, not 64.
  const float* cmd_tab;
  int ct_len, ct_tmin, ct_w;
  // engine-built reset table: rst_tab[task][init index][NO + 4] = the observation of initial_states[idx] (NO floats)
  // and its tracking error against cmd(0) (slot NO): a restarting env reads 80 B instead of redoing y = C x + Y
  const float* rst_tab;
  const float4* frag;  // [n_task][NQ][64]
  const float* tvec;   // [n_task][NS + 2 NO]
  // Slot layout (nullptr / n_slot == n_env: identity).  When the caller's env -> task map does not put 16 envs of one
  // task side by side, the engine orders its own state by task instead: envs are sorted by task (stably) and packed
  // into tiles of 16 slots, a task's last tile padded with empty slots (slot_env = -1).  State (x, sn) is indexed by
  // SLOT; everything the caller sees (actions, outputs, global env id of the random draws) stays indexed by ENV, so
  // results do not depend on the layout.
  const int32_t* slot_env;   // [n_slot] env of a slot or -1
  const int32_t* env_slot;   // [n_env]  slot of an env
  const int32_t* tile_task;  // [n_slot / 16]
  int n_slot;
  int task_shift;            // >= 4: env_task[i] == i >> task_shift for every env (the usual blocks of a power-of-two
                             // number of envs per task): the tile's task is arithmetic, not a load ahead of every load
};

#define XV_LINDS_NR_BIT 0x80000000u

// measurement switches (scripts/devtools/build_variant.py): kept only while an A/B is being taken
#ifndef XV_LINDS_NT_STORES
#define XV_LINDS_NT_STORES 1   // 1: obs / cmd / final_obs leave with non-temporal stores: written once, read by somebody else —
#endif                         //    they must not displace the state, sn, the fragments and the command rows in the L2
                               //    (measured at config 3: 10.25 -> 9.41 us per step, fused roll-out 7.3 -> 5.7 us per step)
#ifndef XV_LINDS_NT_MORE
#define XV_LINDS_NT_MORE 12    // bit 0: action reads non-temporal (measured: no gain, slower beyond 131k envs); bit 1: scalar
#endif                         // outputs (no gain); bit 2: the state stores too (9.57 -> 9.29 us: kept); bit 3: the state
                               // loads too (8.73 -> 8.34 us issued from C: kept) — the state is read once and written once per
                               // step, the L2 is better spent on the fragments and the command rows
#ifndef XV_LINDS_WARM
#define XV_LINDS_WARM 0        // 1: touch the rows a restarting env will read (initial state, its tabulated observation, cmd(0))
#endif                         //    while the first loads are in flight, so that the restart reads hit the cache
#ifndef XV_LINDS_OCC
#define XV_LINDS_OCC 4         // waves per SIMD the (NO = 16) step kernel is capped for
#endif

// float index of component k of slot s in the fragment-tile state (MT = NS / 16)
__host__ __device__ __forceinline__ size_t linds_xidx(int s, int k, int MT) {
  return ((((size_t)(s >> 4) * MT + (k >> 4)) * 64 + 16 * ((k >> 2) & 3) + (s & 15)) << 2) + (k & 3);
}

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
  // info["steps"] and the terminated | truncated mask of the same step (xv_linds_step_info, matrix kernel; nullable)
  int32_t* steps_out;
  uint8_t* done_out;
};

struct xv_linds {
  xv_engine* eng;
  LinDSArgs a;
  bool tiles_uniform;   // every aligned 16-env group of the CALLER's order shares a task (else: slot layout)
  int32_t *d_slot_env, *d_env_slot, *d_tile_task;   // owned; null in the identity layout
  int path;             // XV_LINDS_PATH_*
  float* cmd_tab;       // owned; a.cmd_tab points here while the table is enabled
  float* rst_tab;       // owned; likewise
  float4* frag;         // owned
  float* tvec;          // owned
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

// sin / cos of an angle already reduced to [-pi, pi] in fp64: quadrant in fp64, Taylor polynomials on [-pi/4, pi/4] in
// fp32 (truncation < 3e-8; ~25 instructions and no slow path, unlike sincosf, which this replaces: the Fourier command
// is evaluated in a cold corner of the step kernel and must not set its register allocation).  The command table is
// built with this same function, so table and direct evaluation agree bit for bit.
__device__ __forceinline__ void xv_sincos_reduced(double a, float* sn, float* cs) {
  const double q = rint(a * 0.63661977236758134308);
  const float r = (float)fma(-q, 1.57079632679489661923, a);
  const float r2 = r * r;
  float s = fmaf(r2, 2.7557319e-6f, -1.9841270e-4f);
  s = fmaf(s, r2, 8.3333333e-3f);
  s = fmaf(s, r2, -1.6666667e-1f);
  s = fmaf(s * r2, r, r);
  float c = fmaf(r2, 2.4801587e-5f, -1.3888889e-3f);
  c = fmaf(c, r2, 4.1666667e-2f);
  c = fmaf(c, r2, -0.5f);
  c = fmaf(c, r2, 1.0f);
  const int qi = (int)q & 3;
  const float s1 = (qi & 1) ? c : s, c1 = (qi & 1) ? s : c;
  *sn = (qi & 2) ? -s1 : s1;
  *cs = ((qi + 1) & 2) ? -c1 : c1;
}
__device__ __forceinline__ void linds_four_sincos(const LinDSArgs& P, int tu, int k, double inv_period_t, float* sn, float* cs) {
  double ang = xv_cptr(P.T.four_omega)[(size_t)tu * XV_LINDS_KMAX + k] * inv_period_t;   // random_nn.py:362-368
  ang -= 6.283185307179586476925286766559 * rint(ang * 0.15915494309189533576888376337251);
  xv_sincos_reduced(ang, sn, cs);
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
  for (int k = 0; k < nf; ++k) {
    float sn, cs;
    linds_four_sincos(P, tu, k, inv_period_t, &sn, &cs);
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

// the same for the rows 16 mo + 4 g + r of one lane group only (the matrix kernel's cold path: a time outside the command
// table, or no table): per-lane coefficient reads, four accumulators — same operations per element as above, same bits
template <int NO>
__device__ __forceinline__ void linds_cmd_quarter(const LinDSArgs& P, int tu, int nf, int tt, int g, float (&out)[NO / 16][4]) {
  constexpr int MO = NO / 16;
  const float* valid = P.T.valid + (size_t)tu * NO + 4 * g;
  if (nf == 0) {
    const float* c0 = P.T.cmd0 + (size_t)tu * NO + 4 * g;
#pragma unroll
    for (int mo = 0; mo < MO; ++mo)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[mo][r] = c0[16 * mo + r] * valid[16 * mo + r];
    return;
  }
#pragma unroll
  for (int mo = 0; mo < MO; ++mo)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[mo][r] = 0.0f;
  const double inv_period_t = (double)tt / xv_cptr(P.T.four_period)[tu];
#pragma clang loop unroll(disable)
  for (int k = 0; k < nf; ++k) {
    float sn, cs;
    linds_four_sincos(P, tu, k, inv_period_t, &sn, &cs);
    const float* c = P.T.four_coef + (((size_t)tu * XV_LINDS_KMAX + k) * NO + 4 * g) * 2;
#pragma unroll
    for (int mo = 0; mo < MO; ++mo)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        out[mo][r] = fmaf(c[2 * (16 * mo + r)], sn, out[mo][r]);
        out[mo][r] = fmaf(c[2 * (16 * mo + r) + 1], cs, out[mo][r]);
      }
  }
#pragma unroll
  for (int mo = 0; mo < MO; ++mo)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[mo][r] *= valid[16 * mo + r];
}

// The command of an env depends on (task, integer time) only, and a step needs it at two times (tracked and
// reported), 2 x NO x nf sin/cos pairs per env-step when evaluated directly.  The engine tabulates it once per
// task at create time with the function above (same code, same bits); a step then reads two 64-B rows.  Times
// outside the table (an env stepped on past truncation with auto-reset disabled) are evaluated directly.
template <int NO>
__device__ __forceinline__ void linds_cmd_at(const LinDSArgs& P, int tu, int nf, int tt, float (&out)[NO]) {
  const int idx = tt - P.ct_tmin;
  if (P.cmd_tab != nullptr && idx >= 0 && idx < P.ct_len) {
    const float4* p = reinterpret_cast<const float4*>(P.cmd_tab + ((size_t)tu * P.ct_len + idx) * P.ct_w);
#pragma unroll
    for (int q = 0; q < NO / 4; ++q) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (4 * q < P.ct_w) v = p[q];
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
__device__ __forceinline__ float linds_errsq(const LinDSArgs& P, int tu, const float (&y)[NO], const float (&c)[NO]) {
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
  return (p[0] + p[1]) + (p[2] + p[3]);
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
  float4* row = reinterpret_cast<float4*>(tab + idx * P.ct_w);
#pragma unroll
  for (int q = 0; q < NO / 4; ++q)
    if (4 * q < P.ct_w) row[q] = make_float4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
}

// max over tasks of max_steps and of the command delay (sizes the command table), and the number of live command
// columns: 1 + the last column whose target_valid is non-zero in some task (commands are multiplied by target_valid)
static __global__ __launch_bounds__(256) void linds_max_ints_kernel(const int32_t* ints, const float* valid, int NO, int n_task, int* out3) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_task) return;
  atomicMax(out3, ints[(size_t)t * 4]);
  atomicMax(out3 + 1, ints[(size_t)t * 4 + 1]);
  int live = 0;
  for (int j = 0; j < NO; ++j)
    if (valid[(size_t)t * NO + j] != 0.0f) live = j + 1;
  atomicMax(out3 + 2, live);
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
  err = __builtin_amdgcn_sqrtf(linds_errsq<NO>(P, tu, y, c));   // v_sqrt_f32 everywhere on the device (1 ulp of sqrtf)
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

// initial-state index of a restarting env: floor(w0 * n_init / 2^32), w0 the first word of the env's RESET draw (one
// v_mul_hi_u32; oracle: linds_draw_init)
__device__ __forceinline__ int linds_init_from_word(uint32_t w0, int n_init) {
  const int idx = (int)__umulhi(w0, (uint32_t)n_init);
  return idx < n_init ? idx : n_init - 1;
}
__device__ __forceinline__ int linds_draw_init(const LinDSArgs& P, uint64_t gid, uint64_t tick, int n_init) {
  return linds_init_from_word(xv_env_draw(P.seed, gid, tick, XV_DRAW_RESET).x, n_init);
}

// Process noise of state component j: normal i = 4 (j >> 4) + (j & 3) of Philox call XV_DRAW_NOISE + ((j >> 2) & 3)
// (philox.h: xv_box_muller16) — lane (n, g) of the matrix kernel owns components 16 m + 4 g + r, i.e. ONE call.
// Returns the xor of the call's four words: for g = 0 this is the env's RESTART word of the step — an env that finishes
// in this call draws its initial state with it (linds_init_from_word), so the auto-reset costs no Philox call of its own
// (the noise it is derived from belongs to the episode that has just ended).
// (XV_LINDS_AB_NOISE: measurement builds only, scripts/runs_r06/gpu_e.sh — 1: no Philox call and no Box-Muller, zeros; 2: the
//  Philox call but no Box-Muller, the normals replaced by scaled integer bits.  Never defined in the product build.)
template <int MT>
__device__ __forceinline__ uint32_t linds_noise_group(const LinDSArgs& P, uint64_t gid, uint64_t tick, int g, float (&z)[MT][4]) {
#if defined(XV_LINDS_AB_NOISE) && XV_LINDS_AB_NOISE == 1
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) z[m][r] = 0.0f;
  return (uint32_t)gid * 2654435761u ^ (uint32_t)tick * 40503u ^ (uint32_t)g;
#endif
  const xv_u32x4 w = xv_env_draw(P.seed, gid, tick, XV_DRAW_NOISE + (uint32_t)g);
#if defined(XV_LINDS_AB_NOISE) && XV_LINDS_AB_NOISE == 2
  z[0][0] = (float)(int)(w.x & 0xFFFFu) * 1.0e-5f; z[0][1] = (float)(int)(w.x >> 16) * 1.0e-5f;
  z[0][2] = (float)(int)(w.y & 0xFFFFu) * 1.0e-5f; z[0][3] = (float)(int)(w.y >> 16) * 1.0e-5f;
  if (MT > 1) {
    z[MT - 1][0] = (float)(int)(w.z & 0xFFFFu) * 1.0e-5f; z[MT - 1][1] = (float)(int)(w.z >> 16) * 1.0e-5f;
    z[MT - 1][2] = (float)(int)(w.w & 0xFFFFu) * 1.0e-5f; z[MT - 1][3] = (float)(int)(w.w >> 16) * 1.0e-5f;
  }
  return (w.x ^ w.y) ^ (w.z ^ w.w);
#endif
  xv_box_muller16(w.x, &z[0][0], &z[0][1]);
  xv_box_muller16(w.y, &z[0][2], &z[0][3]);
  if (MT > 1) {
    xv_box_muller16(w.z, &z[MT - 1][0], &z[MT - 1][1]);
    xv_box_muller16(w.w, &z[MT - 1][2], &z[MT - 1][3]);
  }
  return (w.x ^ w.y) ^ (w.z ^ w.w);
}
// the restart word for a kernel that makes no noise for the env (xo: linds_step_restart_word)
__device__ __forceinline__ uint32_t linds_restart_word(const LinDSArgs& P, uint64_t gid, uint64_t tick) {
  const xv_u32x4 w = xv_env_draw(P.seed, gid, tick, XV_DRAW_NOISE);
  return (w.x ^ w.y) ^ (w.z ^ w.w);
}

// sum of squares of the RAW padded action (:164), four fmaf chains like the row sums above: chain g over k = g, 4 + g, ...
template <int NA>
__device__ __forceinline__ float linds_action_sq(const float (&a)[NA]) {
  float p[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float acc = 0.0f;
#pragma unroll
    for (int kk = 0; kk < NA / 4; ++kk) acc = fmaf(a[4 * kk + g], a[4 * kk + g], acc);
    p[g] = acc;
  }
  return (p[0] + p[1]) + (p[2] + p[3]);
}

template <int NS, int NA, int NO, bool INJECT>
__global__ __launch_bounds__(256) void linds_step_kernel(LinDSArgs P, LinDSStepIO io, int mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  const int N = P.n_env;
  constexpr int MT = NS / 16;
  const int si = P.env_slot ? P.env_slot[i] : i;   // where this env's state lives
  const int t = P.env_task[i];
  const uint64_t gid = P.gid_base + (uint64_t)i;

  float xs[NS], a_raw[NA];
#pragma unroll
  for (int k = 0; k < NS; ++k) xs[k] = P.x[linds_xidx(si, k, MT)];
  {
    const float4* a4 = reinterpret_cast<const float4*>(io.action + (size_t)i * NA);
#pragma unroll
    for (int q = 0; q < NA / 4; ++q) {
      const float4 v = a4[q];
      a_raw[4 * q] = v.x; a_raw[4 * q + 1] = v.y; a_raw[4 * q + 2] = v.z; a_raw[4 * q + 3] = v.w;
    }
  }
  const uint32_t sn0 = (uint32_t)P.sn[si];
  int steps = (int)(sn0 & ~XV_LINDS_NR_BIT);
  int nr = (int)(sn0 >> 31);
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
      if (!INJECT) init_idx = linds_init_from_word(linds_restart_word(P, gid, xv_launch_tick(P.tick, P.tick_dev)), n_init);
      if (mode == XV_AUTORESET_NEXT_STEP && nr) {
        // the call after a done ignores the action and returns the reset observation
        linds_reset_env<NS, NO>(P, tu, nf, n_init, init_idx, xs, y, crep, o_err);
        steps = 0;
        nr = 0;
      } else {
        float act[NA];
        const float sa = linds_action_sq<NA>(a_raw);   // :164 cost on the RAW padded action
#pragma unroll
        for (int k = 0; k < NA; ++k) act[k] = a_raw[k] < -1.0f ? -1.0f : (a_raw[k] > 1.0f ? 1.0f : a_raw[k]);   // :138 clip
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
        for (int gq = 0; gq < 4; ++gq) {   // + Xt + noise: the components 16 m + 4 gq + r of one noise call at a time
          float z[MT][4];
          if (INJECT) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int r = 0; r < 4; ++r) z[m][r] = io.z[(size_t)(16 * m + 4 * gq + r) * N + i];
          } else {
            (void)linds_noise_group<MT>(P, gid, xv_launch_tick(P.tick, P.tick_dev), gq, z);
          }
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = 16 * m + 4 * gq + r;
              xn[j] = xn[j] + xtv[j];
              xn[j] = fmaf(noise_scale, z[m][r], xn[j]);
              bad = bad || !(fabsf(xn[j]) <= 3.0e38f);
            }
        }
        if (bad) err |= XV_DEVERR_NONFINITE;
        linds_observe<NS, NO>(P, tu, xn, y);      // :145
        steps += 1;                               // :147
        float ctrack[NO];
        linds_cmd_at<NO>(P, tu, nf, steps - 1 - delay, ctrack);   // :150-151 tracked command
        linds_cmd_at<NO>(P, tu, nf, steps, crep);                 // :168 reported command
        const float s_err = linds_errsq<NO>(P, tu, y, ctrack);   // :153
        o_err = __builtin_amdgcn_sqrtf(s_err);
        // :154, :156 on the squares (see the matrix kernel): sqrtf(s) > c  <=>  s > nextafterf(c * c)
        o_term = ((s_err > 100.00000762939453125f) || (linds_sumsq<NO>(y) > 400.000030517578125f)) ? 1 : 0;
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
  for (int k = 0; k < NS; ++k) P.x[linds_xidx(si, k, MT)] = xs[k];
  P.sn[si] = (int32_t)((uint32_t)steps | (nr ? XV_LINDS_NR_BIT : 0u));
  linds_store_row<NO>(io.obs + (size_t)i * NO, y);
  linds_store_row<NO>(io.cmd + (size_t)i * NO, crep);
  io.reward[i] = o_r;
  io.error[i] = o_err;
  io.terminated[i] = (uint8_t)o_term;
  io.truncated[i] = (uint8_t)o_trunc;
  if (io.final_obs && wrote_fobs) linds_store_row<NO>(io.final_obs + (size_t)i * NO, fobs);   // rows of finished envs only
  if (err) atomicOr(P.err, err);
}

template <int NS, int NO, bool INJECT>
__global__ __launch_bounds__(256) void linds_reset_kernel(LinDSArgs P, const uint8_t* mask,
                                                          const int32_t* init_index, float* obs, float* cmd,
                                                          float* error) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  if (mask && !mask[i]) return;
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
      const int idx = INJECT ? init_index[i] : linds_draw_init(P, P.gid_base + (uint64_t)i, xv_launch_tick(P.tick, P.tick_dev), n_init);
      linds_reset_env<NS, NO>(P, tu, nf, n_init, idx, xs, y, c, e);
      break;
    }
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) P.x[linds_xidx(si, k, NS / 16)] = xs[k];
  P.sn[si] = 0;
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
#ifndef XV_LINDS_ACC_AGPR
#define XV_LINDS_ACC_AGPR 1
#endif
#if XV_LINDS_ACC_AGPR
__device__ __forceinline__ void xv_mfma_settle(xv_f32x4& acc) { asm volatile("s_nop 7\n\ts_nop 3" : "+a"(acc)); }
#else
__device__ __forceinline__ void xv_mfma_settle(xv_f32x4& acc) { asm volatile("s_nop 7\n\ts_nop 3" : "+v"(acc)); }
#endif

__device__ __forceinline__ float xv_sel4(int g, float a, float b, float c, float d) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  const float lo = (g & 1) ? b : a, hi = (g & 1) ? d : c;
  return (g & 2) ? hi : lo;
}

// (p0 + p1) + (p2 + p3) of a per-lane-group partial in every lane of the env's quad (lanes n, n + 16, n + 32, n + 48): the
// order linds_err / linds_sumsq / linds_action_sq define.  gfx950's row / half swaps instead of two LDS-crossbar
// permutes: v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of its second,
// v_permlane32_swap the upper half of the first with the lower half of the second; applied to two copies of a value
// they leave {own, neighbour} in the two registers, and the addition is commutative.
__device__ __forceinline__ float linds_quad_sum(float part) {
  const uint32_t v = __float_as_uint(part);
  const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const uint32_t w = __float_as_uint(s);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

__device__ __forceinline__ void linds_store_out4(float* p, float a, float b, float c, float d) {
#if XV_LINDS_NT_STORES
  typedef float f4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(f4{a, b, c, d}, reinterpret_cast<f4*>(p));
#else
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
#endif
}

// the value lanes 0..15 hold, in all four 16-lane rows (lane n + 16 g <- lane n)
__device__ __forceinline__ uint32_t linds_bcast_row0(uint32_t v) {
  const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);   // a[0] = rows {0, 0, 2, 2}
  const auto b = __builtin_amdgcn_permlane32_swap(a[0], a[0], false, false);   // b[0] = rows {0, 0, 0, 0}
  return b[0];
}

// what a wave loads once per task: the A-operand fragments of the three products and the per-row vectors of its lane group
template <int NS, int NA, int NO>
struct LinDSFrag {
  static constexpr int MT = NS / 16, MO = NO / 16, KS = NS / 4, KA = NA / 4;
  static constexpr int NF = MT * KS + MT * KA + MO * KS, NQ = (NF + 3) / 4;
  float f[NQ * 4];
  float xtr[MT][4], y0r[MO][4], vld[MO][4];
  __device__ __forceinline__ float pa(int m, int kk) const { return f[m * KS + kk]; }
  __device__ __forceinline__ float ga(int m, int kk) const { return f[MT * KS + m * KA + kk]; }
  __device__ __forceinline__ float ca(int mo, int s) const { return f[MT * KS + MT * KA + mo * KS + s]; }
  __device__ __forceinline__ void load(const LinDSArgs& P, int t, int lane) {
    const float4* fq = P.frag + (size_t)t * NQ * 64 + lane;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float4 v = fq[q * 64];
      f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
    }
    const float* tv = P.tvec + (size_t)t * (NS + 2 * NO) + 4 * (lane >> 4);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float4 v = *reinterpret_cast<const float4*>(tv + 16 * m);
      xtr[m][0] = v.x; xtr[m][1] = v.y; xtr[m][2] = v.z; xtr[m][3] = v.w;
    }
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) {
      const float4 v = *reinterpret_cast<const float4*>(tv + NS + 16 * mo);
      const float4 w = *reinterpret_cast<const float4*>(tv + NS + NO + 16 * mo);
      y0r[mo][0] = v.x; y0r[mo][1] = v.y; y0r[mo][2] = v.z; y0r[mo][3] = v.w;
      vld[mo][0] = w.x; vld[mo][1] = w.y; vld[mo][2] = w.z; vld[mo][3] = w.w;
    }
  }
};

// builds frag (one thread per (task, q, lane)) and tvec from the caller's tables
static __global__ __launch_bounds__(256) void linds_build_frag_kernel(LinDSArgs P, float4* frag, float* tvec) {
  const int NS = P.NS, NA = P.NA, NO = P.NO;
  const int MT = NS / 16, MO = NO / 16, KS = NS / 4, KA = NA / 4;
  const int NF = MT * KS + MT * KA + MO * KS, NQ = (NF + 3) / 4;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (size_t)P.n_task * (NS + 2 * NO)) {
    const int t = (int)(idx / (NS + 2 * NO)), j = (int)(idx % (NS + 2 * NO));
    tvec[idx] = j < NS ? P.T.xt[(size_t)t * NS + j]
              : j < NS + NO ? P.T.y0[(size_t)t * NO + j - NS] : P.T.valid[(size_t)t * NO + j - NS - NO];
  }
  if (idx >= (size_t)P.n_task * NQ * 64) return;
  const int lane = (int)(idx & 63), q = (int)((idx >> 6) % NQ), t = (int)((idx >> 6) / NQ);
  const int n = lane & 15, g = lane >> 4;
  float v[4];
  for (int c = 0; c < 4; ++c) {
    int fi = 4 * q + c;
    float val = 0.0f;
    if (fi < MT * KS) {          // pa[m][kk] = Phi[16 m + n][k],  k = 16 (kk >> 2) + 4 g + (kk & 3)   (linds_yorder)
      const int m = fi / KS, kk = fi % KS, k = 16 * (kk >> 2) + 4 * g + (kk & 3);
      val = P.T.phiT[((size_t)t * NS + k) * NS + 16 * m + n];
    } else if ((fi -= MT * KS) < MT * KA) {   // ga[m][kk] = Gamma[16 m + n][4 kk + g]
      const int m = fi / KA, kk = fi % KA;
      val = P.T.gamT[((size_t)t * NA + 4 * kk + g) * NS + 16 * m + n];
    } else if ((fi -= MT * KA) < MO * KS) {   // ca[mo][s] = C[16 mo + n][k],  k as above for slab s
      const int mo = fi / KS, sl = fi % KS, k = 16 * (sl >> 2) + 4 * g + (sl & 3);
      val = P.T.cT[((size_t)t * NS + k) * NO + 16 * mo + n];
    }
    v[c] = val;
  }
  frag[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

// One step of one tile, state in / out of registers.  Shared by the single-step kernel and the fused roll-out.
//   xs    in: the tile's state fragments (this lane: components 16 m + 4 g + r); out: the state after the step / restart
//   steps in / out; nr: need_reset, in / out
template <int NS, int NA, int NO, bool INJECT>
struct LinDSTileStep {
  using F = LinDSFrag<NS, NA, NO>;
  static constexpr int MT = F::MT, MO = F::MO, KS = F::KS, KA = F::KA;

  __device__ __forceinline__ static void run(const LinDSArgs& P, const F& fr, int t, int lane, int e, bool valid, uint64_t gid,
                                             uint64_t tick, int mode, const float* action_row, const float* z_inj, int N,
                                             int init_inj, size_t orow /* row of this env in the [.][n_env] outputs */,
                                             float* o_obs, float* o_cmd, float* o_fobs, float* o_reward, float* o_error,
                                             uint8_t* o_term_p, uint8_t* o_trunc_p, xv_f32x4 (&xs)[MT], int& steps, int& nr,
                                             int& bad_out, int32_t* o_steps_p = nullptr, uint8_t* o_done_p = nullptr) {
    const int g = lane >> 4;
    const XV_CONST_AS float* sc = xv_cptr(P.T.scal) + (size_t)t * 8;
    const XV_CONST_AS int32_t* in = xv_cptr(P.T.ints) + (size_t)t * 4;
    const int max_steps = in[0], delay = in[1], n_init = in[2], nf = in[3];

    // ---- this step's inputs: two action words, two command quarters ----
    float ak[KA];
#pragma unroll
    for (int kk = 0; kk < KA; ++kk)
      ak[kk] = (XV_LINDS_NT_MORE & 1) ? __builtin_nontemporal_load(action_row + 4 * kk + g) : action_row[4 * kk + g];
    const int steps_new = steps + 1;                            // :147
    const int trk_time = steps_new - 1 - delay, rep_time = steps_new;   // tracked :150-151, reported :168
    const int trk_idx = trk_time - P.ct_tmin, rep_idx = rep_time - P.ct_tmin;
    const bool trk_in = P.cmd_tab != nullptr && trk_idx >= 0 && trk_idx < P.ct_len;
    const bool rep_in = P.cmd_tab != nullptr && rep_idx >= 0 && rep_idx < P.ct_len;
    float ctr[MO][4], crep[MO][4];   // this lane's rows 16 mo + 4 g + r of the two commands
    float4 cu[MO], cv[MO];   // raw reads; quarters past the table's ct_w live columns read a live one and are zeroed at use
    if (P.cmd_tab != nullptr) {
      const float* p = P.cmd_tab + ((size_t)t * P.ct_len + (trk_in ? trk_idx : 0)) * P.ct_w;
      const float* pr = P.cmd_tab + ((size_t)t * P.ct_len + (rep_in ? rep_idx : 0)) * P.ct_w;
#pragma unroll
      for (int mo = 0; mo < MO; ++mo) {
        const int col = 16 * mo + 4 * g;
        const int cc = col < P.ct_w ? col : 0;
        cu[mo] = *reinterpret_cast<const float4*>(p + cc);
        cv[mo] = *reinterpret_cast<const float4*>(pr + cc);
      }
    }
    // process noise: independent of every load, computed while they are in flight
    float zr[MT][4];
    uint32_t rword = 0;   // lane group 0: the env's restart word
    if (INJECT) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) zr[m][r] = z_inj[(size_t)(16 * m + 4 * g + r) * N + e];
    } else {
      rword = linds_noise_group<MT>(P, gid, tick, g, zr);
    }
    // the normals are made HERE, under the latency of the loads above: hipcc's IR-level sinking otherwise moves the whole
    // Philox / Box-Muller chain behind the first product, onto the critical path (sched_barrier binds the machine
    // scheduler only)
#pragma unroll
    for (int m = 0; m < MT; ++m) asm volatile("" : "+v"(zr[m][0]), "+v"(zr[m][1]), "+v"(zr[m][2]), "+v"(zr[m][3]));
    asm volatile("" : "+v"(rword));
#if XV_LINDS_WARM
    float warm = 0.0f;
    if (!INJECT) {   // one dword per lane group of the env: g = 0 the initial state, 1 its tabulated observation, 2 cmd(0)
      const int widx = linds_init_from_word(linds_bcast_row0(rword), n_init);
      const float* wp = g == 0 ? P.T.init + ((size_t)t * P.NI + widx) * NS
                      : g == 1 && P.rst_tab != nullptr ? P.rst_tab + ((size_t)t * P.NI + widx) * (NO + 4)
                      : P.cmd_tab != nullptr ? P.cmd_tab + ((size_t)t * P.ct_len + (0 - P.ct_tmin)) * P.ct_w : P.T.init;
      warm = *reinterpret_cast<const volatile float*>(wp);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);

    // ---- x' = Phi x + Gamma act  (:78-80): MT independent accumulator chains, interleaved ----
    xv_f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = xv_f32x4{0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr.pa(m, kk), xs[kk >> 2][kk & 3], acc[m], 0, 0, 0);
    float psa = 0.0f;   // :164 cost on the RAW padded action: this lane group's chain of linds_action_sq
#pragma unroll
    for (int kk = 0; kk < KA; ++kk) {
      psa = fmaf(ak[kk], ak[kk], psa);
      float b = ak[kk] > 1.0f ? 1.0f : ak[kk];   // :138 clip (two selects: a NaN action stays NaN, as in the oracle)
      b = ak[kk] < -1.0f ? -1.0f : b;
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr.ga(m, kk), b, acc[m], 0, 0, 0);
    }
    // + Xt + noise on this lane's components
    const float noise_scale = sc[4];
    xv_f32x4 xn[MT];
    int bad = 0;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[m][r] + fr.xtr[m][r];
        v = fmaf(noise_scale, zr[m][r], v);
        bad |= !(fabsf(v) <= 3.0e38f);
        xn[m][r] = v;
      }

    // ---- y = C x' + Y (:145): slab s = 4 m + r takes register r of tile m ----
    xv_f32x4 ym[MO];   // this lane's rows 16 mo + 4 g + r
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) ym[mo] = xv_f32x4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int mo = 0; mo < MO; ++mo)
        ym[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr.ca(mo, s), xn[s >> 2][s & 3], ym[mo], 0, 0, 0);
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) xv_mfma_settle(ym[mo]);
#pragma unroll
    for (int mo = 0; mo < MO; ++mo)
#pragma unroll
      for (int r = 0; r < 4; ++r) ym[mo][r] = ym[mo][r] + fr.y0r[mo][r];   // :85
    if (P.cmd_tab != nullptr) {
#pragma unroll
      for (int mo = 0; mo < MO; ++mo) {
        // the tracked command enters only through (y - c) * target_valid, which is zero in every column past ct_w: the
        // finite value read there instead of a zero cannot show; the reported command is stored, so it is zeroed
        const bool live = 16 * mo + 4 * g < P.ct_w;
        ctr[mo][0] = cu[mo].x; ctr[mo][1] = cu[mo].y; ctr[mo][2] = cu[mo].z; ctr[mo][3] = cu[mo].w;
        crep[mo][0] = live ? cv[mo].x : 0.f; crep[mo][1] = live ? cv[mo].y : 0.f; crep[mo][2] = live ? cv[mo].z : 0.f; crep[mo][3] = live ? cv[mo].w : 0.f;
      }
    }
    if (__ballot(!(trk_in && rep_in)) != 0ull) {   // rare: no table, or an env stepped on outside it (auto-reset disabled)
      if (!trk_in) linds_cmd_quarter<NO>(P, t, nf, trk_time, g, ctr);
      if (!rep_in) linds_cmd_quarter<NO>(P, t, nf, rep_time, g, crep);
    }
    // tracking error (:153), observation scale (:154), action cost: chain g over this lane's own rows, the four lane
    // groups of an env combined by linds_quad_sum — the order linds_err / linds_sumsq / linds_action_sq define
    float pe = 0.0f, ps = 0.0f;
#pragma unroll
    for (int mo = 0; mo < MO; ++mo)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = (ym[mo][r] - ctr[mo][r]) * fr.vld[mo][r];
        pe = fmaf(d, d, pe);
        ps = fmaf(ym[mo][r], ym[mo][r], ps);
      }
    // error > 10 or obs_scale > 20 (:156) decided on the SQUARES: for a correctly rounded square root sqrtf(s) > c holds
    // exactly when s > nextafterf(c * c) (checked over +-5e6 floats around 100 and 400), so the flag does not depend on
    // which square root is used: the observation scale needs none, and the error output takes the one-instruction
    // v_sqrt_f32 (1 ulp; the oracle uses sqrtf — the flags agree exactly, the error to 1 ulp)
    const float s_err = linds_quad_sum(pe), s_obs = linds_quad_sum(ps);
    float o_err = __builtin_amdgcn_sqrtf(s_err);
    const float sa = linds_quad_sum(psa);
    int o_term = ((s_err > 100.00000762939453125f) || (s_obs > 400.000030517578125f)) ? 1 : 0;   // :156
    float o_r = o_term ? -sc[2] : 0.0f;                         // :158-161
    float tmp = fmaf(-sc[3], o_err, sc[1]);
    tmp = fmaf(-sc[0], sa, tmp);
    o_r = fmaf(tmp, sc[5], o_r);                                // :163-164
    int o_trunc = (steps_new >= max_steps - 1) ? 1 : 0;         // :165

    // ---- which envs (re)start this call ----
    const bool skip = (mode == XV_AUTORESET_NEXT_STEP) && nr;   // the call after a done: reset only
    const bool done = !skip && (o_term || o_trunc);
    const bool do_reset = skip || (done && mode == XV_AUTORESET_SAME_STEP);
    if (skip) {
      o_r = 0.0f; o_term = 0; o_trunc = 0; bad = 0;
    } else {
      steps = steps_new;
      if (done && mode == XV_AUTORESET_NEXT_STEP) nr = 1;
    }
    if (valid && done && mode == XV_AUTORESET_SAME_STEP && o_fobs != nullptr) {   // final observation: finished envs only
#pragma unroll
      for (int mo = 0; mo < MO; ++mo)
        linds_store_out4(o_fobs + orow * NO + 16 * mo + 4 * g, ym[mo][0], ym[mo][1], ym[mo][2], ym[mo][3]);
    }
    if (__ballot(do_reset) != 0ull) {   // wave-uniform: the restarted envs take their initial state
      int idx = INJECT ? init_inj : linds_init_from_word(linds_bcast_row0(rword), n_init);
      idx = idx < 0 ? 0 : (idx >= n_init ? n_init - 1 : idx);
      const float* x0 = P.T.init + ((size_t)t * P.NI + idx) * NS + 4 * g;
      xv_f32x4 xr[MT];   // every lane reads the initial state its restart word names; only restarting envs keep it
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float4 v = *reinterpret_cast<const float4*>(x0 + 16 * m);   // :117
        xr[m] = xv_f32x4{v.x, v.y, v.z, v.w};
      }
      float c0[MO][4];   // :120-126: the last pre-filled command is cmd(0)
      const int z_idx = 0 - P.ct_tmin;
      if (P.cmd_tab != nullptr && z_idx >= 0 && z_idx < P.ct_len) {
        const float4* p = reinterpret_cast<const float4*>(P.cmd_tab + ((size_t)t * P.ct_len + z_idx) * P.ct_w);
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (16 * mo + 4 * g < P.ct_w) v = p[4 * mo + g];
          c0[mo][0] = v.x; c0[mo][1] = v.y; c0[mo][2] = v.z; c0[mo][3] = v.w;
        }
      } else {
        linds_cmd_quarter<NO>(P, t, nf, 0, g, c0);
      }
      xv_f32x4 yr[MO];
      float e0;
      if (P.rst_tab != nullptr) {                        // observation and error of initial_states[idx], tabulated
        const float4* row = reinterpret_cast<const float4*>(P.rst_tab + ((size_t)t * P.NI + idx) * (NO + 4));
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) {
          const float4 v = row[4 * mo + g];
          yr[mo] = xv_f32x4{v.x, v.y, v.z, v.w};
        }
        e0 = row[NO / 4].x;
      } else {
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) yr[mo] = xv_f32x4{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int mo = 0; mo < MO; ++mo)
            yr[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr.ca(mo, s), xr[s >> 2][s & 3], yr[mo], 0, 0, 0);
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) xv_mfma_settle(yr[mo]);
        float pr0 = 0.0f;
#pragma unroll
        for (int mo = 0; mo < MO; ++mo)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            yr[mo][r] = yr[mo][r] + fr.y0r[mo][r];
            const float d = (yr[mo][r] - c0[mo][r]) * fr.vld[mo][r];
            pr0 = fmaf(d, d, pr0);
          }
        e0 = __builtin_amdgcn_sqrtf(linds_quad_sum(pr0));
      }
      if (do_reset) {
#pragma unroll
        for (int m = 0; m < MT; ++m) xn[m] = xr[m];
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) {
          ym[mo] = yr[mo];
#pragma unroll
          for (int r = 0; r < 4; ++r) crep[mo][r] = c0[mo][r];
        }
        o_err = e0;
        steps = 0;
        nr = 0;
      }
    }

    // ---- this step's outputs (the state stays with the caller of run) ----
    if (valid) {
#pragma unroll
      for (int mo = 0; mo < MO; ++mo) {   // each lane stores its own 16-byte quarter of the rows
        const size_t ro = orow * NO + 16 * mo + 4 * g;
        linds_store_out4(o_obs + ro, ym[mo][0], ym[mo][1], ym[mo][2], ym[mo][3]);
        if (o_cmd) linds_store_out4(o_cmd + ro, crep[mo][0], crep[mo][1], crep[mo][2], crep[mo][3]);
      }
      if (g == 0) {
#if XV_LINDS_NT_MORE & 2
        __builtin_nontemporal_store(o_r, o_reward + orow);
        if (o_error) __builtin_nontemporal_store(o_err, o_error + orow);
        __builtin_nontemporal_store((uint8_t)o_term, o_term_p + orow);
        __builtin_nontemporal_store((uint8_t)o_trunc, o_trunc_p + orow);
#else
        o_reward[orow] = o_r;
        if (o_error) o_error[orow] = o_err;
        o_term_p[orow] = (uint8_t)o_term;
        o_trunc_p[orow] = (uint8_t)o_trunc;
#endif
        if (o_steps_p) o_steps_p[orow] = steps;                               // the counter after the step (0 after a restart)
        if (o_done_p) o_done_p[orow] = (uint8_t)((o_term | o_trunc) ? 1 : 0);
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) xs[m] = xn[m];
    bad_out |= bad;
#if XV_LINDS_WARM
    asm volatile("" :: "v"(warm));   // the touch's destination stays allocated until the load has landed
#endif
  }
};

// identity of the tile a wave serves
struct LinDSTileId {
  int lane, wave, n, g, es, e, t;
  bool valid;
  uint64_t gid;
};
__device__ __forceinline__ bool linds_tile_id(const LinDSArgs& P, LinDSTileId& id, int bid) {
  id.lane = threadIdx.x & 63;
  id.wave = (int)((bid * blockDim.x + threadIdx.x) >> 6);
  const int tile0 = id.wave * 16;
  if (tile0 >= P.n_slot) return false;   // wave-uniform
  id.n = id.lane & 15; id.g = id.lane >> 4;
  id.es = tile0 + id.n;                  // this lane's state slot (x and sn are allocated in whole tiles)
  int e_raw = id.es;
  if (P.slot_env != nullptr) e_raw = id.es < P.n_slot ? P.slot_env[id.es] : -1;
  id.valid = id.es < P.n_slot && e_raw >= 0 && e_raw < P.n_env;
  id.e = id.valid ? e_raw : 0;           // the env it serves: its I/O rows and the global id of its draws
  id.t = P.task_shift >= 4 ? (tile0 >> P.task_shift)
                           : __builtin_amdgcn_readfirstlane(P.tile_task ? P.tile_task[id.wave] : P.env_task[tile0]);
  id.gid = P.gid_base + (uint64_t)id.e;
  return true;
}

// HAND (overlapped step_many of the mixed batch, mixed.hip): the launch may start while the step before it is still running.
// The wave asks for its task's fragments, then waits for ITS tile's word P.hand[wave] to carry this step's tick (written by
// the same wave of the step before, after its x / sn stores completed), loads x and sn with agent-scope loads, and hands
// the tile on the same way (xv_hand.h).
template <int NS, int NA, int NO, bool INJECT, bool HAND = false>
__device__ __forceinline__ void linds_step_mfma_body(const LinDSArgs& P, const LinDSStepIO& io, int mode, int bid) {
  using F = LinDSFrag<NS, NA, NO>;
  LinDSTileId id;
  if (!linds_tile_id(P, id, bid)) return;
  float4* xq = reinterpret_cast<float4*>(P.x) + (size_t)id.wave * F::MT * 64 + id.lane;
  xv_f32x4 xs[F::MT];
  uint32_t sn0;
  F fr;
  bool late = false;
  if (HAND) {
    fr.load(P, id.t, id.lane);
    late = !xv_hand_wait(P.hand + id.wave, (uint32_t)xv_launch_tick(P.tick, P.tick_dev), P.err);
    asm volatile("" ::: "memory");
    sn0 = xv_agent_load32(P.sn + id.es);
#pragma unroll
    for (int m = 0; m < F::MT; ++m) {
      const uint64_t lo = xv_agent_load64(reinterpret_cast<const uint64_t*>(xq + m * 64));
      const uint64_t hi = xv_agent_load64(reinterpret_cast<const uint64_t*>(xq + m * 64) + 1);
      xs[m] = xv_f32x4{__uint_as_float((uint32_t)lo), __uint_as_float((uint32_t)(lo >> 32)), __uint_as_float((uint32_t)hi),
                       __uint_as_float((uint32_t)(hi >> 32))};
    }
  } else {
  // ---- every load of the step is in flight before the first MFMA; the step counter first (the command rows wait for it) ----
  sn0 = (uint32_t)P.sn[id.es];
#pragma unroll
  for (int m = 0; m < F::MT; ++m) {
#if XV_LINDS_NT_MORE & 8
    typedef float f4 __attribute__((ext_vector_type(4)));
    xs[m] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(xq + m * 64));
#else
    const float4 v = xq[m * 64];
    xs[m] = xv_f32x4{v.x, v.y, v.z, v.w};
#endif
  }
  fr.load(P, id.t, id.lane);
  }
  int steps = (int)(sn0 & ~XV_LINDS_NR_BIT), nr = (int)(sn0 >> 31), bad = 0;
  const int init_inj = INJECT ? io.init_index[id.e] : 0;
  LinDSTileStep<NS, NA, NO, INJECT>::run(P, fr, id.t, id.lane, id.e, id.valid, id.gid, xv_launch_tick(P.tick, P.tick_dev), mode,
                                         io.action + (size_t)id.e * NA, io.z, P.n_env, init_inj, (size_t)id.e, io.obs, io.cmd,
                                         io.final_obs, io.reward, io.error, io.terminated, io.truncated, xs, steps, nr, bad,
                                         io.steps_out, io.done_out);
  if (HAND) {
    if (id.valid) {
#pragma unroll
      for (int m = 0; m < F::MT; ++m) {
        uint64_t* q = reinterpret_cast<uint64_t*>(xq + m * 64);
        xv_agent_store64(q, (uint64_t)__float_as_uint(xs[m][0]) | ((uint64_t)__float_as_uint(xs[m][1]) << 32));
        xv_agent_store64(q + 1, (uint64_t)__float_as_uint(xs[m][2]) | ((uint64_t)__float_as_uint(xs[m][3]) << 32));
      }
      if (id.g == 0) xv_agent_store32(P.sn + id.es, (uint32_t)steps | (nr ? XV_LINDS_NR_BIT : 0u));
    }
    xv_hand_publish(P.hand + id.wave, (uint32_t)xv_launch_tick(P.tick, P.tick_dev) + 1u);
    if (id.valid && (bad || late))
      atomicOr(P.err, (uint32_t)((bad ? XV_DEVERR_NONFINITE : 0u) | (late ? XV_DEVERR_HANDOFF : 0u)));
    return;
  }
  if (id.valid) {
#pragma unroll
    for (int m = 0; m < F::MT; ++m) {
#if XV_LINDS_NT_MORE & 4
      typedef float f4 __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(f4{xs[m][0], xs[m][1], xs[m][2], xs[m][3]}, reinterpret_cast<f4*>(xq + m * 64));
#else
      xq[m * 64] = make_float4(xs[m][0], xs[m][1], xs[m][2], xs[m][3]);
#endif
    }
    if (id.g == 0) P.sn[id.es] = (int32_t)((uint32_t)steps | (nr ? XV_LINDS_NR_BIT : 0u));
    if (bad) atomicOr(P.err, (uint32_t)XV_DEVERR_NONFINITE);
  }
}

// Replay of an overlapped call of the mixed batch whose hand-off expired (mixed.hip: mixed_replay_kernel): the tile starts from
// the state the call's snapshot kept (x_snap, sn_snap), runs the call's n_steps steps with the state in registers as the fused
// roll-out below does — step k reads ring slot k % period of the actions, writes that slot of every output and draws with tick
// tick0 + k, in the call's auto-reset mode — and leaves the state where the one-stream loop would have left it.
template <int NS, int NA, int NO>
__device__ __forceinline__ void linds_replay_body(const LinDSArgs& P, const LinDSStepIO& io /* ring slot 0 */, int period, int n_steps,
                                                  int mode, const float* x_snap, const int32_t* sn_snap, int bid) {
  using F = LinDSFrag<NS, NA, NO>;
  LinDSTileId id;
  if (!linds_tile_id(P, id, bid)) return;
  F fr;
  fr.load(P, id.t, id.lane);
  const float4* xs_q = reinterpret_cast<const float4*>(x_snap) + (size_t)id.wave * F::MT * 64 + id.lane;
  float4* xq = reinterpret_cast<float4*>(P.x) + (size_t)id.wave * F::MT * 64 + id.lane;
  xv_f32x4 xs[F::MT];
#pragma unroll
  for (int m = 0; m < F::MT; ++m) {
    const float4 v = xs_q[m * 64];
    xs[m] = xv_f32x4{v.x, v.y, v.z, v.w};
  }
  const uint32_t sn0 = (uint32_t)sn_snap[id.es];
  int steps = (int)(sn0 & ~XV_LINDS_NR_BIT), nr = (int)(sn0 >> 31), bad = 0;
  const size_t N = (size_t)P.n_env;
  const uint64_t tick0 = xv_launch_tick(P.tick, P.tick_dev);
  for (int k = 0; k < n_steps; ++k) {
    const size_t ob = (size_t)(k % period) * N + id.e;
    LinDSTileStep<NS, NA, NO, false>::run(P, fr, id.t, id.lane, id.e, id.valid, id.gid, tick0 + (uint64_t)k, mode, io.action + ob * NA,
                                          nullptr, P.n_env, 0, ob, io.obs, io.cmd, io.final_obs, io.reward, io.error, io.terminated,
                                          io.truncated, xs, steps, nr, bad);
  }
  if (id.valid) {
#pragma unroll
    for (int m = 0; m < F::MT; ++m) xq[m * 64] = make_float4(xs[m][0], xs[m][1], xs[m][2], xs[m][3]);
    if (id.g == 0) P.sn[id.es] = (int32_t)((uint32_t)steps | (nr ? XV_LINDS_NR_BIT : 0u));
    if (bad) atomicOr(P.err, (uint32_t)XV_DEVERR_NONFINITE);
  }
}

// ------------------------------------------------------------------------------------------------
// Fused roll-out: T steps of the tile in one launch, SAME_STEP auto-reset, free-running noise.  The task's operand
// fragments are loaded once and the state never leaves the registers the matrix unit wrote it to (see the k order
// above); per step only two action words and two command quarters come in and the outputs go out.  Step t draws with
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
  float* final_obs;         // [T][n_env][NO]  nullable; rows of finished envs only
};

template <int NS, int NA, int NO>
__global__ __launch_bounds__(256) void linds_rollout_mfma_kernel(LinDSArgs P, LinDSRolloutIO io, int T) {
  using F = LinDSFrag<NS, NA, NO>;
  LinDSTileId id;
  if (!linds_tile_id(P, id, (int)blockIdx.x)) return;
  F fr;
  fr.load(P, id.t, id.lane);
  float4* xq = reinterpret_cast<float4*>(P.x) + (size_t)id.wave * F::MT * 64 + id.lane;
  xv_f32x4 xs[F::MT];
#pragma unroll
  for (int m = 0; m < F::MT; ++m) {
    const float4 v = xq[m * 64];
    xs[m] = xv_f32x4{v.x, v.y, v.z, v.w};
  }
  int steps = (int)((uint32_t)P.sn[id.es] & ~XV_LINDS_NR_BIT), nr = 0, bad = 0;
  const size_t N = (size_t)P.n_env;
  for (int ts = 0; ts < T; ++ts) {
    const size_t ob = (size_t)ts * N + id.e;          // this step's row of the [T][n_env] outputs
    LinDSTileStep<NS, NA, NO, false>::run(P, fr, id.t, id.lane, id.e, id.valid, id.gid, xv_launch_tick(P.tick, P.tick_dev) + (uint64_t)ts,
                                          XV_AUTORESET_SAME_STEP, io.action + ob * NA, nullptr, P.n_env, 0, ob, io.obs, io.cmd,
                                          io.final_obs, io.reward, io.error, io.terminated, io.truncated, xs, steps, nr, bad);
  }
  if (id.valid) {
#pragma unroll
    for (int m = 0; m < F::MT; ++m) xq[m * 64] = make_float4(xs[m][0], xs[m][1], xs[m][2], xs[m][3]);
    if (id.g == 0) P.sn[id.es] = steps;
    if (bad) atomicOr(P.err, (uint32_t)XV_DEVERR_NONFINITE);
  }
}

// two entry points over the same body: with 16 observation rows the step fits 128 registers and is capped there
// (4 waves per SIMD: the kernel is latency-bound); with 32 rows the cap would spill, so it runs at 2-3 waves per SIMD
template <int NS, int NA, int NO, bool INJECT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(XV_LINDS_OCC, XV_LINDS_OCC == 4 ? 8 : XV_LINDS_OCC)))
void linds_step_mfma_kernel(LinDSArgs P, LinDSStepIO io, int mode) {
#if XV_LINDS_OCC < 4   // measurement only: an LDS allocation that admits XV_LINDS_OCC workgroups (= waves per SIMD) per CU
  __shared__ float occ_pad[(XV_LINDS_OCC == 2 ? 60 : XV_LINDS_OCC == 3 ? 45 : 150) * 256];
  if (P.n_env < 0) occ_pad[threadIdx.x] = 0.0f;
#endif
  linds_step_mfma_body<NS, NA, NO, INJECT>(P, io, mode, (int)blockIdx.x);
}
template <int NS, int NA, int NO, bool INJECT>
__global__ __launch_bounds__(256) void linds_step_mfma_wide_kernel(LinDSArgs P, LinDSStepIO io, int mode) {
  linds_step_mfma_body<NS, NA, NO, INJECT>(P, io, mode, (int)blockIdx.x);
}

// every aligned group of 16 envs shares one task?  (else the engine builds its slot layout)   bit 1: env_task[i] != i >> shift
static __global__ __launch_bounds__(256) void linds_check_tiles_kernel(const int32_t* env_task, int n_env, int shift, int* not_uniform) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_env) return;
  if (env_task[i] != env_task[i & ~15]) atomicOr(not_uniform, 1);
  if (shift < 4 || env_task[i] != (i >> shift)) atomicOr(not_uniform, 2);
}

static inline void linds_bind_rng(xv_linds* h, uint64_t ticks, bool advance = true) {
  h->a.seed = h->eng->seed;
  h->a.gid_base = h->eng->env_id_base;
  const XvTickBind b = xv_engine_bind_tick(h->eng, ticks, advance);
  h->a.tick = b.tick;
  h->a.tick_dev = b.tick_dev;
}

#ifndef XV_KERNELS_ONLY   // mixed.hip includes this file for its kernels and handle types only
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
  a.seed = e->seed; a.gid_base = e->env_id_base; a.tick = 0; a.tick_dev = nullptr;
  a.x = nullptr; a.sn = nullptr; a.frag = nullptr; a.tvec = nullptr;
  h->frag = nullptr; h->tvec = nullptr;
  a.slot_env = nullptr; a.env_slot = nullptr; a.tile_task = nullptr; a.n_slot = n_env; a.task_shift = -1;
  h->d_slot_env = nullptr; h->d_env_slot = nullptr; h->d_tile_task = nullptr;
  hipError_t m = hipSuccess;
  {
    int* d_flag = nullptr;
    int h_flag = 1;
    m = hipMalloc(&d_flag, sizeof(int));
    if (m == hipSuccess) m = hipMemsetAsync(d_flag, 0, sizeof(int), e->stream);
    int shift = -1;   // candidate: n_env = n_task << shift
    if (n_env % n_task == 0) {
      const int per = n_env / n_task;
      if ((per & (per - 1)) == 0) for (shift = 0; (1 << shift) < per; ++shift) {}
    }
    if (m == hipSuccess) {
      hipLaunchKernelGGL(linds_check_tiles_kernel, dim3(xv_div_up(n_env, 256)), dim3(256), 0, e->stream, env_task,
                         n_env, shift, d_flag);
      m = hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, e->stream);
    }
    if (m == hipSuccess) m = hipStreamSynchronize(e->stream);
    if (d_flag) (void)hipFree(d_flag);
    h->tiles_uniform = ((h_flag & 1) == 0);
    a.task_shift = (m == hipSuccess && (h_flag & 2) == 0) ? shift : -1;
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
  // state and counters in whole 16-slot tiles; operand fragments and row vectors per task
  const size_t n_tile = ((size_t)a.n_slot + 15) / 16;
  const int MTc = NS / 16, MOc = NO / 16;
  const int NQ = (MTc * (NS / 4) + MTc * (NA / 4) + MOc * (NS / 4) + 3) / 4;
  const size_t x_bytes = sizeof(float) * n_tile * MTc * 256, sn_bytes = sizeof(int32_t) * n_tile * 16;
  const size_t frag_n = (size_t)n_task * NQ * 64, tvec_n = (size_t)n_task * (NS + 2 * NO);
  if (m == hipSuccess) m = hipMalloc(&a.x, x_bytes);
  if (m == hipSuccess) m = hipMalloc(&a.sn, sn_bytes);
  if (m == hipSuccess) m = hipMalloc(&h->frag, sizeof(float4) * frag_n);
  if (m == hipSuccess) m = hipMalloc(&h->tvec, sizeof(float) * tvec_n);
  if (m == hipSuccess) m = hipMemsetAsync(a.x, 0, x_bytes, e->stream);
  if (m == hipSuccess) m = hipMemsetD32Async((hipDeviceptr_t)a.sn, (int)XV_LINDS_NR_BIT, n_tile * 16, e->stream);   // steps 0, need_reset
  if (m != hipSuccess) {
    xv_set_error("xv_linds_create: device allocation failed: %s", hipGetErrorString(m));
    if (a.x) (void)hipFree(a.x);
    if (a.sn) (void)hipFree(a.sn);
    if (h->frag) (void)hipFree(h->frag);
    if (h->tvec) (void)hipFree(h->tvec);
    if (h->d_slot_env) (void)hipFree(h->d_slot_env);
    if (h->d_env_slot) (void)hipFree(h->d_env_slot);
    if (h->d_tile_task) (void)hipFree(h->d_tile_task);
    delete h;
    return XV_ERR_HIP;
  }
  {
    const size_t nthr = frag_n > tvec_n ? frag_n : tvec_n;
    hipLaunchKernelGGL(linds_build_frag_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, e->stream, a, h->frag, h->tvec);
    a.frag = h->frag; a.tvec = h->tvec;
  }
  // command table: [n_task][max_steps_max + 2 + delay_max][NO] floats, within a 2-GiB budget
  a.cmd_tab = nullptr; a.ct_len = 0; a.ct_tmin = 0; a.ct_w = NO; a.rst_tab = nullptr;
  h->cmd_tab = nullptr; h->rst_tab = nullptr;
  {
    int* d3 = nullptr;
    int h3[3] = {0, 0, 0};
    XV_HIP(hipMalloc(&d3, 3 * sizeof(int)));
    XV_HIP(hipMemsetAsync(d3, 0, 3 * sizeof(int), e->stream));
    hipLaunchKernelGGL(linds_max_ints_kernel, dim3(xv_div_up(n_task, 256)), dim3(256), 0, e->stream, tables->ints, tables->valid,
                       NO, n_task, d3);
    XV_HIP(hipMemcpyAsync(h3, d3, 3 * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    XV_HIP(hipStreamSynchronize(e->stream));
    XV_HIP(hipFree(d3));
    const int ct_w = h3[2] <= 0 ? 4 : (h3[2] + 3) / 4 * 4;
    const long long len = (long long)h3[0] + 2 + h3[1];
    const unsigned long long bytes = (unsigned long long)n_task * (unsigned long long)len * ct_w * sizeof(float);
    float* tab = nullptr;
    if (h3[0] > 0 && h3[1] >= 0 && len < (1 << 20) && bytes <= (2ull << 30) && hipMalloc(&tab, bytes) == hipSuccess) {
      a.ct_len = (int)len;
      a.ct_tmin = -(1 + h3[1]);
      a.ct_w = ct_w;
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
  (void)hipFree(h->a.sn);
  if (h->frag) (void)hipFree(h->frag);
  if (h->tvec) (void)hipFree(h->tvec);
  if (h->d_slot_env) (void)hipFree(h->d_slot_env);
  if (h->d_env_slot) (void)hipFree(h->d_env_slot);
  if (h->d_tile_task) (void)hipFree(h->d_tile_task);
  if (h->cmd_tab) (void)hipFree(h->cmd_tab);
  if (h->rst_tab) (void)hipFree(h->rst_tab);
  delete h;
  return XV_OK;
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

extern "C" int xv_linds_step_info(xv_linds* h, const float* action, float* obs, float* reward, uint8_t* terminated,
                                  uint8_t* truncated, float* cmd, float* error, float* final_obs, int32_t* steps, uint8_t* done,
                                  int autoreset_mode) {
  XV_CHECK_ARG(h && action && obs && reward && terminated && truncated && cmd && error);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  if (h->path == XV_LINDS_PATH_SCALAR) {
    xv_set_error("xv_linds_step_info: served by the matrix kernel (the scalar test kernel: xv_linds_step + xv_linds_get_state)");
    return XV_ERR_UNSUPPORTED;
  }
  linds_bind_rng(h, 1);
  LinDSStepIO io{action, nullptr, nullptr, obs, reward, terminated, truncated, cmd, error, final_obs, steps, done};
  return linds_launch_step<false>(h, io, autoreset_mode);
}

// K vector steps issued from C: step k reads actions slot k % period and writes output slot k % period of [period][...]
// ring buffers (a host loop over xv_linds_step costs ~5 us of Python / ctypes per launch, the kernel ~9)
extern "C" int xv_linds_step_many(xv_linds* h, int n_steps, int period, const float* action, float* obs, float* reward,
                                  uint8_t* terminated, uint8_t* truncated, float* cmd, float* error, float* final_obs,
                                  int autoreset_mode) {
  XV_CHECK_ARG(h && n_steps > 0 && period > 0);
  XV_CHECK_ARG(action && obs && reward && terminated && truncated && cmd && error);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  const size_t n = (size_t)h->a.n_env;
  for (int k = 0; k < n_steps; ++k) {
    const size_t o = (size_t)(k % period) * n;
    linds_bind_rng(h, 1);
    LinDSStepIO io{action + o * h->a.NA, nullptr, nullptr, obs + o * h->a.NO, reward + o, terminated + o, truncated + o,
                   cmd + o * h->a.NO, error + o, final_obs ? final_obs + o * h->a.NO : nullptr};
    const int rc = linds_launch_step<false>(h, io, autoreset_mode);
    if (rc != XV_OK) return rc;
  }
  return XV_OK;
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

// engine state (fragment tiles, slot order) <-> the caller's component-major float[NS][n_env] / steps / need_reset
template <bool TO_ENV>
__global__ __launch_bounds__(256) void linds_permute_state_kernel(LinDSArgs P, float* x, int32_t* steps, uint8_t* need_reset) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  const int N = P.n_env, MT = P.NS / 16, si = P.env_slot ? P.env_slot[i] : i;
  if (x) {
    for (int k = 0; k < P.NS; ++k) {
      if (TO_ENV) x[(size_t)k * N + i] = P.x[linds_xidx(si, k, MT)];
      else P.x[linds_xidx(si, k, MT)] = x[(size_t)k * N + i];
    }
  }
  const uint32_t w = (uint32_t)P.sn[si];
  if (TO_ENV) {
    if (steps) steps[i] = (int32_t)(w & ~XV_LINDS_NR_BIT);
    if (need_reset) need_reset[i] = (uint8_t)(w >> 31);
  } else if (steps || need_reset) {
    const uint32_t st = steps ? ((uint32_t)steps[i] & ~XV_LINDS_NR_BIT) : (w & ~XV_LINDS_NR_BIT);
    const uint32_t nr = need_reset ? (need_reset[i] ? XV_LINDS_NR_BIT : 0u) : (w & XV_LINDS_NR_BIT);
    P.sn[si] = (int32_t)(st | nr);
  }
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
  hipLaunchKernelGGL((linds_permute_state_kernel<true>), dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                     h->a, x, steps, need_reset);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_linds_set_state(xv_linds* h, const float* x, const int32_t* steps, const uint8_t* need_reset) {
  XV_CHECK_ARG(h != nullptr);
  hipLaunchKernelGGL((linds_permute_state_kernel<false>), dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                     h->a, const_cast<float*>(x), const_cast<int32_t*>(steps), const_cast<uint8_t*>(need_reset));
  XV_LAUNCH_CHECK();
  return XV_OK;
}
#endif   // XV_KERNELS_ONLY
