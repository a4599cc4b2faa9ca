// anymdp_synth.hip — synthetic AnyMDP task tables generated directly in HBM.
//
// The headline config has one task per env: 65,536 tasks x (S=64, A=8) = 44 GiB of row records (fp64 CDF
// entries + {reward, noise} pairs in 128-byte blocks, a fence line per row) — more than the host can stage —
// so the tables are produced on the device, directly in the row layout of include/xeno.h.  Integer-only construction, bit-identical to oracle/xeno_oracle.c: xo_anymdp_synth (weights are
// integers, partial sums exact in uint32, one IEEE fp64 division per CDF entry), so the generator itself
// is parity-tested.  Row shape follows the reference sampler's banded transitions
// (anymdp/task_sampler_utils.py:65-175): support of row (s,a) is a band [lo,hi) around s.
#include "philox.h"
#include "xv_common.h"

__device__ __forceinline__ xv_u32x4 synth_draw(uint64_t seed, uint64_t task, uint32_t c2, uint32_t c3) {
  return xv_philox4x32_10((uint32_t)task, (uint32_t)(task >> 32), c2, c3, (uint32_t)seed,
                          (uint32_t)(seed >> 32));
}
__device__ __forceinline__ uint32_t pick(const xv_u32x4& w, int k) {
  return k == 0 ? w.x : (k == 1 ? w.y : (k == 2 ? w.z : w.w));
}

// one thread per task: max_steps, s_0 distribution, state_map permutation, terminal set
__global__ __launch_bounds__(64) void anymdp_synth_header_kernel(uint64_t seed, int64_t task_base,
                                                                 int n_task, int S, int s0_max,
                                                                 int32_t* state_map, uint64_t* term_mask,
                                                                 double* s0_cdf, int32_t* s0_ids,
                                                                 int32_t* max_steps) {
  const int tl = blockIdx.x * blockDim.x + threadIdx.x;
  if (tl >= n_task) return;
  const uint64_t task = (uint64_t)(task_base + tl);
  const int words = (S + 63) / 64;
  xv_u32x4 w = synth_draw(seed, task, 0xFFFFFFFFu, 0);
  max_steps[tl] = 256 + (int32_t)(w.x % 245u);
  int s0_len = 3;
  if (s0_len > s0_max) s0_len = s0_max;
  if (s0_len > S) s0_len = S;
  uint32_t cum = 0, tot = 0;
  for (int k = 0; k < s0_len; ++k) tot += 1u + (pick(w, 1 + k) >> 8);
  for (int k = 0; k < s0_max; ++k) {
    if (k < s0_len) {
      cum += 1u + (pick(w, 1 + k) >> 8);
      s0_cdf[(size_t)tl * s0_max + k] = (double)cum / (double)tot;
      s0_ids[(size_t)tl * s0_max + k] = k;
    } else {
      s0_cdf[(size_t)tl * s0_max + k] = 1.0;
      s0_ids[(size_t)tl * s0_max + k] = s0_len - 1;
    }
  }
  int32_t* sm = state_map + (size_t)tl * S;
  for (int i = 0; i < S; ++i) sm[i] = i;
  for (int i = S - 1; i >= 1; --i) {
    w = synth_draw(seed, task, 0xFFFFFFFFu, 0x100u + (uint32_t)(i >> 2));
    const int j = (int)(pick(w, i & 3) % (uint32_t)(i + 1));
    const int32_t tmp = sm[i]; sm[i] = sm[j]; sm[j] = tmp;
  }
  uint64_t tm[4] = {0, 0, 0, 0};
  int n_c = S - 3, n_term = (15 * S) / 100;
  if (n_c < 0) n_c = 0;
  if (n_term > n_c) n_term = n_c;
  uint8_t cand[256];
  for (int k = 0; k < n_c; ++k) cand[k] = (uint8_t)(3 + k);
  for (int k = 0; k < n_term; ++k) {
    w = synth_draw(seed, task, 0xFFFFFFFFu, 0x200u + (uint32_t)(k >> 2));
    const int j = k + (int)(pick(w, k & 3) % (uint32_t)(n_c - k));
    const uint8_t tmp = cand[k]; cand[k] = cand[j]; cand[j] = tmp;
    const int c = cand[k];
    if ((c >> 6) == 0) tm[0] |= 1ull << (c & 63);
    else if ((c >> 6) == 1) tm[1] |= 1ull << (c & 63);
    else if ((c >> 6) == 2) tm[2] |= 1ull << (c & 63);
    else tm[3] |= 1ull << (c & 63);
  }
  for (int k = 0; k < words; ++k) term_mask[(size_t)tl * words + k] = tm[k];
}

// one wave per (task, s, a) row; lane j owns entries j, j+64, ...
__global__ __launch_bounds__(256) void anymdp_synth_rows_kernel(uint64_t seed, int64_t task_base,
                                                                int n_task, int S, int A,
                                                                const uint64_t* term_mask, void* rows) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const size_t n_rows = (size_t)n_task * S * A;
  if (wave >= n_rows) return;
  const int tl = (int)(wave / ((size_t)S * A));
  const uint32_t rowid = (uint32_t)(wave % ((size_t)S * A));
  const int s = (int)(rowid / A);
  const uint64_t task = (uint64_t)(task_base + tl);
  const int words = (S + 63) / 64;
  const bool term = (term_mask[(size_t)tl * words + (s >> 6)] >> (s & 63)) & 1ull;

  const xv_u32x4 hw = synth_draw(seed, task, rowid, 0x1000u);
  const int lo_min = s - 33 > 0 ? s - 33 : 0;
  const int lo = lo_min + (int)(hw.x % (uint32_t)(s - lo_min + 1));
  const int hi_min = s + 2 < S ? s + 2 : S;
  const int hi_max = s + 17 < S ? s + 17 : S;
  const int hi = hi_min + (int)(hw.y % (uint32_t)(hi_max - hi_min + 1));

  const int chunks = (S + 63) / 64;
  uint32_t wt[4];
  uint32_t total = 0;
  for (int c = 0; c < chunks; ++c) {
    const int j = c * 64 + lane;
    uint32_t v = 0;
    if (j < S && j >= lo && j < hi) {
      const xv_u32x4 w = synth_draw(seed, task, rowid, (uint32_t)(j >> 2));
      v = 104858u + (pick(w, j & 3) >> 12) % 943718u;
    }
    wt[c] = v;
    uint32_t r = v;  // wave sum
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) r += __shfl_xor(r, off);
    total += r;
  }
  uint32_t carry = 0;
  // row record: include/xeno.h "rows" — line 0 (fence) and the metadata unit of each block are completed by
  // xv_anymdp_create; here only the 16-byte entries {cdf, reward, noise}, 7 per 128-byte block
  const int NB0 = (S + 6) / 7, G = (NB0 + 15) / 16;
  const int NB = (NB0 + G - 1) / G * G;   // XV_ANYMDP_ROW_LINES(S) - 1
  uint4* row = reinterpret_cast<uint4*>(rows) + wave * (size_t)(1 + NB) * 8;
  for (int c = 0; c < chunks; ++c) {
    const int j = c * 64 + lane;
    uint32_t incl = wt[c];  // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t n = __shfl_up(incl, off);
      if (lane >= off) incl += n;
    }
    const uint32_t chunk_sum = __shfl(incl, 63);
    if (j < S) {
      const double cj = term ? 1.0 : (double)(carry + incl) / (double)total;
      const xv_u32x4 w = synth_draw(seed, task, rowid, 0x100u + (uint32_t)(j >> 1));
      const uint32_t wa = (j & 1) ? w.z : w.x, wb = (j & 1) ? w.w : w.y;
      const float r = (float)((int32_t)(wa >> 8) - 8388608) * (1.0f / 4194304.0f);
      const float sg = (wb & 1u) ? (float)(wb >> 8) * (1.0f / 67108864.0f) : 0.0f;
      const int b = j / 7;
      row[(size_t)(1 + b) * 8 + (j - 7 * b)] =
          make_uint4((uint32_t)__double2loint(cj), (uint32_t)__double2hiint(cj), __float_as_uint(r), __float_as_uint(sg));
    }
    carry += chunk_sum;
  }
  // padding entries of the last block: cdf 2.0 (never <= u), zero reward pair
  const int jp = S + lane;
  if (jp < NB * 7) {
    const int b = jp / 7;
    row[(size_t)(1 + b) * 8 + (jp - 7 * b)] = make_uint4((uint32_t)__double2loint(2.0), (uint32_t)__double2hiint(2.0), 0u, 0u);
  }
}

extern "C" int xv_anymdp_synth_tasks(xv_engine* e, uint64_t seed, int64_t task_index_base, int n_task,
                                     int S, int A, int s0_max, void* rows,
                                     int32_t* state_map, uint64_t* term_mask, double* s0_cdf,
                                     int32_t* s0_ids, int32_t* max_steps) {
  XV_CHECK_ARG(e != nullptr && n_task > 0);
  XV_CHECK_ARG(S >= 4 && S <= 256 && A >= 2 && A <= 64 && s0_max >= 1 && s0_max <= 256);
  XV_CHECK_ARG(rows && state_map && term_mask && s0_cdf && s0_ids && max_steps);
  XV_HIP(hipSetDevice(e->device));
  hipLaunchKernelGGL(anymdp_synth_header_kernel, dim3(xv_div_up(n_task, 64)), dim3(64), 0, e->stream, seed,
                     task_index_base, n_task, S, s0_max, state_map, term_mask, s0_cdf, s0_ids, max_steps);
  XV_LAUNCH_CHECK();
  // one wave per row: a launch holds at most 2^32 threads, so large batches go in chunks of whole tasks
  const int words = (S + 63) / 64;
  const size_t row_bytes = (size_t)XV_ANYMDP_ROW_LINES(S) * 128;
  const int chunk = (int)((size_t)(1u << 22) * 4 / ((size_t)S * A));     // tasks per launch: <= 2^22 blocks
  for (int t0 = 0; t0 < n_task; t0 += chunk) {
    const int nt = n_task - t0 < chunk ? n_task - t0 : chunk;
    const size_t blocks = ((size_t)nt * S * A + 3) / 4;
    hipLaunchKernelGGL(anymdp_synth_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, e->stream, seed,
                       task_index_base + t0, nt, S, A, (const uint64_t*)term_mask + (size_t)t0 * words,
                       static_cast<char*>(rows) + (size_t)t0 * S * A * row_bytes);
    XV_LAUNCH_CHECK();
  }
  return XV_OK;
}
