// Dev microbenchmark (not shipped): what keeps the one-line AnyMDP step (bucket search) above the bare "one random 128-B
// line per env" floor?  Variants of the cooperative one-line kernel of latency_floor.hip with, one at a time, the real
// kernel's block shape, its number of coalesced input / output streams and a Philox-sized dependent ALU chain in front of
// the line address.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ inline uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

template <int NIN, int NOUT, int ALU, int POST>
__global__ void kv(const uint4* __restrict__ table, uint64_t n_lines128, uint32_t* const* ins, uint32_t* const* outs,
                   uint32_t tick, int n) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, g = lane >> 3, j = lane & 7;
  uint32_t acc = 0;
#pragma unroll
  for (int q = 0; q < NIN; ++q) acc ^= ins[q][e] + q;            // coalesced input streams
  uint64_t h = ((uint64_t)e << 32) ^ tick;
#pragma unroll
  for (int q = 0; q < ALU; ++q) h = mix(h);                      // dependent ALU that does NOT need the loads (Philox)
  h = mix(h ^ acc);                                              // the address needs both
  const uint32_t lo = (uint32_t)(h % n_lines128);
  uint4 v[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const uint32_t li = (uint32_t)__shfl((int)lo, it * 8 + g);
    v[it] = table[(uint64_t)li * 8 + j];
  }
  uint32_t own = 0;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const unsigned long long m = __ballot((v[it].x & 1u) != 0);
    uint32_t c = __popc((unsigned)(m >> (8 * (lane & 7))) & 0xFFu) + v[it].y;
#pragma unroll
    for (int p = 0; p < POST; ++p) c += __shfl((int)(v[it].z + p), 8 * j + (c & 7));   // the real kernel's 3 shuffles per iteration
    if ((lane >> 3) == it) own = c;
  }
  acc += own;
#pragma unroll
  for (int q = 0; q < NOUT; ++q) outs[q][e] = acc + q;
}

template <int NIN, int NOUT, int ALU, int POST>
static void run(const char* name, const uint4* table, uint64_t n_lines, uint32_t** d_ins, uint32_t** d_outs, int block, int n = 65536) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  for (int i = 0; i < 100; ++i) kv<NIN, NOUT, ALU, POST><<<n / block, block>>>(table, n_lines, d_ins, d_outs, i, n);
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) kv<NIN, NOUT, ALU, POST><<<n / block, block>>>(table, n_lines, d_ins, d_outs, 1000 + i, n);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-64s block %-3d %.2f us per launch\n", name, block, ms * 1e3 / iters);
}

int main() {
  const uint64_t bytes = 32ull << 30, n_lines = bytes / 128;
  uint4* table;
  if (hipMalloc(&table, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(table, 0x5A, bytes);
  uint32_t* h_ins[16]; uint32_t* h_outs[16];
  for (int q = 0; q < 16; ++q) { (void)hipMalloc(&h_ins[q], 65536 * 4); (void)hipMemset(h_ins[q], q + 1, 65536 * 4); (void)hipMalloc(&h_outs[q], 65536 * 4); }
  uint32_t **d_ins, **d_outs;
  (void)hipMalloc(&d_ins, sizeof(h_ins)); (void)hipMalloc(&d_outs, sizeof(h_outs));
  (void)hipMemcpy(d_ins, h_ins, sizeof(h_ins), hipMemcpyHostToDevice); (void)hipMemcpy(d_outs, h_outs, sizeof(h_outs), hipMemcpyHostToDevice);
  (void)hipDeviceSynchronize();
  run<3, 2, 0, 0>("3 in, 2 out, no ALU (the floor)", table, n_lines, d_ins, d_outs, 64);
  run<3, 2, 0, 0>("3 in, 2 out, no ALU", table, n_lines, d_ins, d_outs, 256);
  run<11, 2, 0, 0>("11 in, 2 out", table, n_lines, d_ins, d_outs, 256);
  run<11, 10, 0, 0>("11 in, 10 out", table, n_lines, d_ins, d_outs, 256);
  run<3, 2, 6, 0>("3 in, 2 out, 6 mix rounds of ALU before the address", table, n_lines, d_ins, d_outs, 256);
  run<3, 2, 20, 0>("3 in, 2 out, 20 mix rounds of ALU before the address", table, n_lines, d_ins, d_outs, 256);
  run<3, 2, 0, 3>("3 in, 2 out, 3 shuffles per iteration after the line", table, n_lines, d_ins, d_outs, 256);
  run<11, 10, 20, 3>("11 in, 10 out, 20 rounds, 3 shuffles (everything)", table, n_lines, d_ins, d_outs, 256);
  run<11, 10, 20, 3>("11 in, 10 out, 20 rounds, 3 shuffles (everything)", table, n_lines, d_ins, d_outs, 64);
  return 0;
}
