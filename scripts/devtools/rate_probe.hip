// Issue rate of the fp64 instructions the exact texture filter is made of (gfx950): 8 independent chains per lane,
// 4 waves per SIMD, long loop; cycles per wave-instruction = time * clock / (instructions per wave * waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o rate_probe scripts/devtools/rate_probe.hip && ./rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  double a[8];
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + i * 0.125 + threadIdx.x * 1e-3; f[i] = (float)a[i]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) a[i] = __builtin_fma(a[i], 1.0000001, 0.5);                       // v_fma_f64
      if (OP == 1) a[i] = a[i] + 0.5;                                                // v_add_f64
      if (OP == 2) a[i] = a[i] * 1.0000001;                                          // v_mul_f64
      if (OP == 3) { f[i] = (float)a[i]; asm volatile("" : "+v"(f[i])); a[i] = (double)f[i]; asm volatile("" : "+v"(a[i])); }   // cvt pair
      if (OP == 4) { f[i] = __builtin_fmaf(f[i], 1.0000001f, 0.5f); }               // v_fma_f32
      if (OP == 5) { double t = a[i] * 536870913.0; asm volatile("" : "+v"(t)); double u = t - a[i]; asm volatile("" : "+v"(u)); a[i] = t - u; asm volatile("" : "+v"(a[i])); }   // Veltkamp
      if (OP == 6) a[i] = __builtin_fmax(a[i], 0.01);                                // v_max_f64
      if (OP == 7) { f[i] = (float)a[i]; asm volatile("" : "+v"(f[i])); }           // v_cvt_f32_f64 alone (a not updated)
      if (OP == 8) { a[i] = (double)f[i]; asm volatile("" : "+v"(a[i])); }          // v_cvt_f64_f32 alone
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static void run(const char* name, int per_iter) {
  const int blocks = 256 * 4, iters = 20000;   // 4 workgroups of 4 waves per CU = 4 waves per SIMD
  double* out;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const double insts_per_simd = 4.0 * iters * 8 * per_iter;   // 4 waves per SIMD
  printf("%-28s %8.3f ms  -> %.2f cycles per wave-instruction at %.2f GHz (%d instr per chain step)\n", name, ms,
         ms * 1e-3 * clk_khz * 1e3 / insts_per_simd, clk_khz * 1e-6, per_iter);
  hipFree(out);
}

int main() {
  run<0>("v_fma_f64", 1);
  run<1>("v_add_f64", 1);
  run<2>("v_mul_f64", 1);
  run<6>("v_max_f64", 1);
  run<4>("v_fma_f32", 1);
  run<3>("cvt f64->f32->f64 pair", 2);
  run<7>("v_cvt_f32_f64", 1);
  run<8>("v_cvt_f64_f32", 1);
  run<5>("Veltkamp (mul, sub, sub)", 3);
  return 0;
}
