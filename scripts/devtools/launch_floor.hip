// Dev microbenchmark (not shipped): back-to-back launch cost of an (almost) empty kernel vs grid shape.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out, unsigned tick) { if (tick == 0xFFFFFFFFu) out[blockIdx.x * blockDim.x + threadIdx.x] = tick; }
static void run(int blocks, int threads, unsigned* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 3000;
  for (int i = 0; i < 100; ++i) k<<<blocks, threads>>>(out, i);
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) k<<<blocks, threads>>>(out, i);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("grid %5d x %4d threads (%6d lanes): %.2f us per launch\n", blocks, threads, blocks * threads, ms * 1e3 / iters);
}
int main() {
  unsigned* out; (void)hipMalloc(&out, 1 << 22);
  run(1, 64, out); run(64, 64, out); run(256, 64, out); run(1024, 64, out); run(512, 128, out); run(256, 256, out);
  run(128, 512, out); run(64, 1024, out); run(4096, 64, out); run(1024, 256, out);
  return 0;
}
