// devtool: what a wave-to-wave hand-off costs at agent scope (sc1: through the fabric, what the overlapped step kernels use) and at
// workgroup scope (sc0: through the XCD's own L2) when the two waves sit on the same XCD.  Two kernels coexist on two streams;
// block j of B plays ping-pong with block (j + shift) mod n of A: A stores i to x, B polls x, stores i to y, A polls y.  For every
// shift 0..7: how many pairs share an XCD (HW_REG_XCC_ID), the one-way latency at each scope, and the pairs that timed out (a
// workgroup-scope poll on another XCD keeps reading its own L2's stale line).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SCOPE>   // 0: agent, 1: workgroup
__device__ __forceinline__ uint32_t ld(const uint32_t* p) {
  return SCOPE ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
               : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int SCOPE>
__device__ __forceinline__ void st(uint32_t* p, uint32_t v) {
  if (SCOPE) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int SCOPE>
__global__ void pp(int role, uint32_t* x, uint32_t* y, int n, int shift, int iters, uint32_t* xcc, uint64_t* ticks, uint32_t* timeouts) {
  if (threadIdx.x != 0) return;
  uint32_t id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  const int b = blockIdx.x;
  const int pair = role == 0 ? b : (b + shift) % n;          // index of the pair's words = A's block index
  xcc[role * n + b] = id & 0xF;
  uint32_t* px = x + 32 * pair;                                // one 128-byte line per word
  uint32_t* py = y + 32 * pair;
  const uint64_t t0 = wall_clock64();
  uint32_t to = 0;
  for (int i = 1; i <= iters && !to; ++i) {
    if (role == 0) {
      st<SCOPE>(px, (uint32_t)i);
      uint32_t polls = 0;
      while (ld<SCOPE>(py) != (uint32_t)i) if (++polls > 2000000u) { to = 1; break; }
    } else {
      uint32_t polls = 0;
      while (ld<SCOPE>(px) != (uint32_t)i) if (++polls > 2000000u) { to = 1; break; }
      st<SCOPE>(py, (uint32_t)i);
    }
  }
  ticks[role * n + b] = wall_clock64() - t0;
  if (to) atomicAdd(timeouts, 1u);
}

int main() {
  const int n = 64, iters = 2000;
  uint32_t *x, *y, *xcc, *to;
  uint64_t* ticks;
  hipMalloc(&x, n * 128); hipMalloc(&y, n * 128); hipMalloc(&xcc, 2 * n * 4); hipMalloc(&ticks, 2 * n * 8); hipMalloc(&to, 4);
  hipStream_t s0, s1;
  hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
  int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi);
  for (int scope = 0; scope < 2; ++scope)
    for (int shift = 0; shift < 8; ++shift) {
      hipMemset(x, 0, n * 128); hipMemset(y, 0, n * 128); hipMemset(to, 0, 4);
      hipDeviceSynchronize();
      if (scope == 0) {
        hipLaunchKernelGGL(pp<0>, dim3(n), dim3(64), 0, s0, 0, x, y, n, shift, iters, xcc, ticks, to);
        hipLaunchKernelGGL(pp<0>, dim3(n), dim3(64), 0, s1, 1, x, y, n, shift, iters, xcc, ticks, to);
      } else {
        hipLaunchKernelGGL(pp<1>, dim3(n), dim3(64), 0, s0, 0, x, y, n, shift, iters, xcc, ticks, to);
        hipLaunchKernelGGL(pp<1>, dim3(n), dim3(64), 0, s1, 1, x, y, n, shift, iters, xcc, ticks, to);
      }
      hipDeviceSynchronize();
      std::vector<uint32_t> hx(2 * n); std::vector<uint64_t> ht(2 * n); uint32_t hto = 0;
      hipMemcpy(hx.data(), xcc, 2 * n * 4, hipMemcpyDeviceToHost);
      hipMemcpy(ht.data(), ticks, 2 * n * 8, hipMemcpyDeviceToHost);
      hipMemcpy(&hto, to, 4, hipMemcpyDeviceToHost);
      int same = 0; double us = 0;
      for (int b = 0; b < n; ++b) { same += hx[(b + shift) % n] == hx[n + b]; us += (double)ht[b] / 100.0 / iters / 2.0; }
      printf("%s scope, shift %d: %2d / %d pairs on one XCD, one-way hand-off %.3f us (mean over the A blocks), blocks timed out %u; XCD of A0 %u, of B0 %u\n",
             scope ? "workgroup (sc0)" : "agent     (sc1)", shift, same, n, us / n, hto, hx[0], hx[n]);
    }
  return 0;
}
