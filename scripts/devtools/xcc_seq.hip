// devtool: is the workgroup -> XCD map of a grid the same in consecutive launches on ONE stream?  (a) the same grid back to back,
// (b) with a one-workgroup kernel between the launches, for grids that are / are not multiples of 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void rec(uint32_t* out) {
  uint32_t x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = x & 0xF;
}
__global__ void tiny(uint32_t* p) { if (threadIdx.x == 0) p[0] += 1; }

int main() {
  uint32_t *buf, *t;
  hipMalloc(&buf, 4 * 4096 * 4); hipMalloc(&t, 4); hipMemset(t, 0, 4);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int grid : {1024, 1026, 1032, 256, 300}) {
    for (int between = 0; between < 2; ++between) {
      for (int k = 0; k < 4; ++k) {
        hipLaunchKernelGGL(rec, dim3(grid), dim3(256), 0, s, buf + k * 4096);
        if (between) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, t);
      }
      hipStreamSynchronize(s);
      std::vector<uint32_t> h(4 * 4096);
      hipMemcpy(h.data(), buf, 4 * 4096 * 4, hipMemcpyDeviceToHost);
      int same01 = 0, same12 = 0, same23 = 0, mod8 = 0;
      for (int i = 0; i < grid; ++i) {
        same01 += h[i] == h[4096 + i]; same12 += h[4096 + i] == h[8192 + i]; same23 += h[8192 + i] == h[12288 + i];
        mod8 += h[i] == (uint32_t)(i % 8);
      }
      printf("grid %4d, %s: same XCD launch 0/1 %d, 1/2 %d, 2/3 %d of %d; launch 0 == b %% 8: %d; first 10 of launches 0..3:", grid,
             between ? "1-WG kernel between" : "back to back      ", same01, same12, same23, grid, mod8);
      for (int k = 0; k < 4; ++k) { printf(" |"); for (int i = 0; i < 10; ++i) printf(" %u", h[k * 4096 + i]); }
      printf("\n");
    }
  }
  return 0;
}
