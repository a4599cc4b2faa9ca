// anymdp_tables.hip — the device half of set_task for raw task tensors: transition / reward / reward_noise (fp64, as the
// reference's task dicts hold them, anymdp/task_sampler.py:46-50) -> the blocked row records the step kernels read.
//
// What the reference does per step is numpy.random.choice(n, p=row) (anymdp_env.py:99-100), which forms
// `cdf = p.cumsum(); cdf /= cdf[-1]` — a sequential fp64 accumulate and one IEEE division per entry.  The host builder
// (xenoverse_amd/anymdp/tables.py: build_tables) forms exactly that with NumPy, 7.5 ms and 1.2 MB of host arrays per 64 x 8
// task.  Here one thread per row (task, s, a) runs the same accumulate in the same order and the same divisions (hipcc's
// fp64 `/` is correctly rounded), so the rows equal build_tables' bit for bit (tests/test_gpu_tables.py), and writes the
// record: blocks of 7 entries {cdf, reward f32, noise f32}, padding entries {2.0, 0, 0}; the fence line and the blocks'
// metadata stay zero — xv_anymdp_create completes them, as for host-built rows.  All-zero rows (terminal states: the
// reference never samples from them) become 1.0, as in row_cdf.  The reference's row check, (sum(row) - 1)^2 < 1e-6 unless
// the state is in s_e (anymdp_env.py:66-71), is made on the accumulate's last value; the smallest failing row index lands
// in *bad_row.
#include "xv_common.h"

#include <cstddef>

#define XV_TAB_BLK 7

static __global__ __launch_bounds__(256) void anymdp_build_rows_kernel(size_t n_rows, int S, int A, int NB, const double* tr,
                                                                       const double* rw, const double* rn,
                                                                       const uint64_t* term_mask, int words, uint4* rows,
                                                                       unsigned long long* bad_row, unsigned long long row_base) {
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const double* p = tr + r * (size_t)S;
  double last = 0.0;
  for (int j = 0; j < S; ++j) last += p[j];      // numpy.cumsum: c[j] = c[j - 1] + p[j]
  const size_t sa = r / (size_t)A, t = sa / (size_t)S;
  const int s = (int)(sa - t * (size_t)S);
  const bool terminal = (term_mask[t * (size_t)words + (s >> 6)] >> (s & 63)) & 1ull;
  const double dev = last - 1.0;
  if (!terminal && dev * dev >= 1.0e-6) atomicMin(bad_row, row_base + (unsigned long long)r);      // `(err >= 1e-6).any()`, :70
  const bool zero = last == 0.0;
  const double* q = rw + r * (size_t)S;
  const double* z = rn + r * (size_t)S;
  uint4* line = rows + r * (size_t)(1 + NB) * 8;      // 8 units of 16 bytes per 128-byte line
#pragma unroll
  for (int u = 0; u < 8; ++u) line[u] = make_uint4(0u, 0u, 0u, 0u);      // the fence line: completed by xv_anymdp_create
  double c = 0.0;
  for (int b = 0; b < NB; ++b) {
    uint4* blk = line + (size_t)(1 + b) * 8;
    for (int k = 0; k < XV_TAB_BLK; ++k) {
      const int j = b * XV_TAB_BLK + k;
      double cdf = 2.0;
      float rv = 0.0f, nv = 0.0f;
      if (j < S) {
        c += p[j];
        cdf = zero ? 1.0 : c / last;
        rv = (float)q[j];
        nv = (float)z[j];
      }
      const unsigned long long cb = (unsigned long long)__double_as_longlong(cdf);
      blk[k] = make_uint4((uint32_t)cb, (uint32_t)(cb >> 32), __float_as_uint(rv), __float_as_uint(nv));
    }
    blk[7] = make_uint4(0u, 0u, 0u, 0u);             // metadata: completed by xv_anymdp_create
  }
}

extern "C" int xv_anymdp_build_rows(xv_engine* e, int n_task, int S, int A, const double* transition, const double* reward,
                                    const double* reward_noise, const uint64_t* term_mask, void* rows_out,
                                    unsigned long long* bad_row, unsigned long long row_index_base) {
  XV_CHECK_ARG(e && transition && reward && reward_noise && term_mask && rows_out && bad_row);
  XV_CHECK_ARG(n_task > 0 && S >= 2 && S <= 512 && A >= 2 && A <= 64);
  XV_HIP(hipSetDevice(e->device));
  const int NB0 = (S + XV_TAB_BLK - 1) / XV_TAB_BLK, G = (NB0 + 15) / 16, NB = (NB0 + G - 1) / G * G;
  const size_t n_rows = (size_t)n_task * S * A;
  XV_CHECK_ARG((n_rows + 255) / 256 < 0x7FFFFFFFull);
  hipLaunchKernelGGL(anymdp_build_rows_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, e->stream, n_rows, S, A, NB,
                     transition, reward, reward_noise, term_mask, (S + 63) / 64, (uint4*)rows_out, bad_row, row_index_base);
  XV_LAUNCH_CHECK();
  return XV_OK;
}
