// devtool: which XCD runs workgroup b?  Two launches of the same grid on two streams (the second spins until the first has
// started, so they overlap), each workgroup records HW_REG_XCC_ID.  Is the map a function of b alone?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void rec(uint32_t* out, uint32_t* flag, int wait) {
  uint32_t x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = x & 0xF;
  if (wait == 0 && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // keep the workgroups resident for a while so that the two launches really coexist
  const uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000ull) __builtin_amdgcn_s_sleep(4);
}

int main() {
  for (int grid : {64, 224, 256, 300, 512, 1024}) {
    uint32_t *a, *b, *flag;
    hipMalloc(&a, grid * 4); hipMalloc(&b, grid * 4); hipMalloc(&flag, 4);
    hipMemset(flag, 0, 4);
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi);
    int same = 0, rr = 0, rr2 = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(rec, dim3(grid), dim3(256), 0, s0, a, flag, 0);
      hipLaunchKernelGGL(rec, dim3(grid), dim3(256), 0, s1, b, flag, 1);
      hipDeviceSynchronize();
      std::vector<uint32_t> ha(grid), hb(grid);
      hipMemcpy(ha.data(), a, grid * 4, hipMemcpyDeviceToHost);
      hipMemcpy(hb.data(), b, grid * 4, hipMemcpyDeviceToHost);
      same = rr = rr2 = 0;
      for (int i = 0; i < grid; ++i) { same += ha[i] == hb[i]; rr += ha[i] == (uint32_t)(i % 8); rr2 += hb[i] == (uint32_t)(i % 8); }
      if (rep == 2) {
        printf("grid %4d: same XCD in both launches %d / %d; launch A == b %% 8: %d, launch B == b %% 8: %d; first 16 of A:", grid, same, grid, rr, rr2);
        for (int i = 0; i < 16 && i < grid; ++i) printf(" %u", ha[i]);
        printf(" | B:");
        for (int i = 0; i < 16 && i < grid; ++i) printf(" %u", hb[i]);
        printf("\n");
      }
    }
    hipFree(a); hipFree(b); hipFree(flag); hipStreamDestroy(s0); hipStreamDestroy(s1);
  }
  return 0;
}
