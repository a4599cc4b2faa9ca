// Dev microbenchmark (not shipped): does a hipGraph of K dependent (almost) empty kernels run faster per kernel than K
// stream launches?  (It decides whether xv_anymdp_step_many should replay a captured graph.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out, unsigned tick) { if (tick == 0xFFFFFFFFu) out[blockIdx.x * blockDim.x + threadIdx.x] = tick; }
int main() {
  unsigned* out; (void)hipMalloc(&out, 1 << 22);
  hipStream_t st; (void)hipStreamCreate(&st);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int K = 200, reps = 20;
  for (int i = 0; i < 100; ++i) k<<<256, 256, 0, st>>>(out, i);
  (void)hipEventRecord(e0, st);
  for (int r = 0; r < reps; ++r) for (int i = 0; i < K; ++i) k<<<256, 256, 0, st>>>(out, i);
  (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("stream launches : %.2f us per kernel\n", ms * 1e3 / (K * reps));
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < K; ++i) k<<<256, 256, 0, st>>>(out, i);
  (void)hipStreamEndCapture(st, &g);
  if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
  for (int r = 0; r < 3; ++r) (void)hipGraphLaunch(ge, st);
  (void)hipEventRecord(e0, st);
  for (int r = 0; r < reps; ++r) (void)hipGraphLaunch(ge, st);
  (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("graph replay    : %.2f us per kernel (%d-node graph)\n", ms * 1e3 / (K * reps), K);
  return 0;
}
