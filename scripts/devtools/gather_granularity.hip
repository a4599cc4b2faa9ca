// Dev microbenchmark (not shipped): cost of random HBM gathers as a function of how many bytes of which
// 64-B / 128-B / 256-B neighbourhood each lane touches.  262,144 lanes, 16 GiB table (no cache reuse).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ inline uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// MODE: 0 none, 1: 16 B, 2: 2x16 B same 64-B sector, 3: 2x16 B in the two 64-B halves of one 128-B line,
//       4: 2x16 B in the two 128-B halves of a 256-B block, 5: 64 B (4x16 contiguous), 6: 128 B, 7: 256 B,
//       8: 2x16 B in two unrelated lines
template <int MODE>
__global__ __launch_bounds__(64) void k(const uint4* __restrict__ table, uint64_t n_blocks, uint32_t* out, uint32_t tick) {
  const uint32_t e = blockIdx.x * 64 + threadIdx.x;
  const uint64_t h = mix(((uint64_t)e << 32) ^ tick);
  const uint4* blk = table + (h % n_blocks) * 16;   // 256-B block = 16 uint4
  uint32_t acc = 0;
  if (MODE == 1) acc = blk[0].x;
  if (MODE == 2) acc = blk[0].x ^ blk[3].y;
  if (MODE == 3) acc = blk[0].x ^ blk[4].y;
  if (MODE == 4) acc = blk[0].x ^ blk[8].y;
  if (MODE == 5) { for (int i = 0; i < 4; ++i) acc ^= blk[i].x; }
  if (MODE == 6) { for (int i = 0; i < 8; ++i) acc ^= blk[i].x; }
  if (MODE == 7) { for (int i = 0; i < 16; ++i) acc ^= blk[i].x; }
  if (MODE == 8) acc = blk[0].x ^ table[(mix(h) % n_blocks) * 16].y;
  out[e] = acc;
}

template <int MODE>
static void run(const char* name, const uint4* table, uint64_t n_blocks, uint32_t* out, int n) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 500;
  for (int i = 0; i < 50; ++i) k<MODE><<<n / 64, 64>>>(table, n_blocks, out, i);
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) k<MODE><<<n / 64, 64>>>(table, n_blocks, out, 1000 + i);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-52s n=%-7d %.2f us per launch\n", name, n, ms * 1e3 / iters);
}

int main() {
  const uint64_t bytes = 16ull << 30, n_blocks = bytes / 256;
  uint4* table; uint32_t* out;
  if (hipMalloc(&table, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(table, 0x5A, bytes);
  (void)hipMalloc(&out, 1048576 * 4);
  (void)hipDeviceSynchronize();
  for (int n : {65536, 262144, 1048576}) {
    run<0>("no gather", table, n_blocks, out, n);
    run<1>("16 B", table, n_blocks, out, n);
    run<2>("2 x 16 B, same 64-B sector", table, n_blocks, out, n);
    run<3>("2 x 16 B, two 64-B halves of a 128-B line", table, n_blocks, out, n);
    run<4>("2 x 16 B, two 128-B halves of a 256-B block", table, n_blocks, out, n);
    run<8>("2 x 16 B, two unrelated lines", table, n_blocks, out, n);
    run<5>("64 B contiguous", table, n_blocks, out, n);
    run<6>("128 B contiguous", table, n_blocks, out, n);
    run<7>("256 B contiguous", table, n_blocks, out, n);
  }
  return 0;
}
