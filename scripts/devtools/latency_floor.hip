// Dev microbenchmark (not shipped): what one vector step of 65,536 lanes costs as a function of the number of
// DEPENDENT random HBM gathers in it -- the structure of the AnyMDP step (env state -> header+fence -> row block).
//   k0: empty kernel (1,024 waves)            k1: coalesced state read + write
//   k2: k1 + one random 128-B gather          k3: k2 + a second gather whose address depends on the first
//   k4: k3 + a third dependent gather
// The table is 16 GiB, so every gather misses every cache.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ inline uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

template <int DEPTH>
__global__ __launch_bounds__(64) void k(const uint4* __restrict__ table, uint64_t n_lines /* 256-B lines */,
                                        uint32_t* state, const uint32_t* action, uint32_t* out, uint32_t tick) {
  const uint32_t e = blockIdx.x * 64 + threadIdx.x;
  if (DEPTH == 0) return;
  uint32_t s = state[e], a = action[e];
  uint64_t h = mix(((uint64_t)e << 32) ^ s ^ ((uint64_t)a << 8) ^ tick);
  uint32_t acc = s;
#pragma unroll
  for (int d = 1; d < DEPTH; ++d) {
    const uint4* line = table + (h % n_lines) * 16;        // 256-B line = 16 uint4
    // the 16 lanes of a group read the whole line in the real kernel; here each lane reads 2 x 16 B of its own line
    const uint4 v0 = line[(e + d) & 15], v1 = line[(e + d + 7) & 15];
    acc += v0.x ^ v1.y;
    h = mix(h ^ v0.z ^ v1.w);                              // next address depends on the data
  }
  state[e] = acc;
  out[e] = acc ^ a;
}

// the planned AnyMDP layout: per env two DEPENDENT random 128-B lines (fence line, then one block line), each read
// cooperatively by 8 lanes x 16 B (8 envs per load instruction, 8 instructions per wave and level), + dword streams
template <int LEVELS>
__global__ __launch_bounds__(64) void kc(const uint4* __restrict__ table, uint64_t n_lines128, uint32_t* state,
                                         const uint32_t* action, uint32_t* out, const uint4* __restrict__ rec, uint32_t tick) {
  const uint32_t e = blockIdx.x * 64 + threadIdx.x;
  const int lane = threadIdx.x & 63, g = lane >> 3, j = lane & 7;
  uint32_t s = state[e], a = action[e];
  const uint4 rr = rec[e];   // 16 B of per-env reset record, streamed
  uint64_t h = mix(((uint64_t)e << 32) ^ s ^ ((uint64_t)a << 8) ^ tick);
  uint32_t acc = s ^ rr.x;
#pragma unroll
  for (int lv = 0; lv < LEVELS; ++lv) {
    const uint32_t lo = (uint32_t)(h % n_lines128), hi = 0;
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
      const uint32_t c = __popc((unsigned)(m >> (8 * (lane & 7))) & 0xFFu) + v[it].y;
      if ((lane >> 3) == it) own = c;
    }
    acc += own;
    h = mix(h ^ own);
    (void)hi;
  }
  state[e] = acc;
  out[e] = acc ^ a;
}

template <int LEVELS>
static void runc(const char* name, const uint4* table, uint64_t n_lines, uint32_t* state, uint32_t* action, uint32_t* out,
                 const uint4* rec, int n = 65536) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  for (int i = 0; i < 100; ++i) kc<LEVELS><<<n / 64, 64>>>(table, n_lines, state, action, out, rec, i);
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) kc<LEVELS><<<n / 64, 64>>>(table, n_lines, state, action, out, rec, 1000 + i);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-46s n=%-7d %.2f us per launch\n", name, n, ms * 1e3 / iters);
}

template <int DEPTH>
static void run(const char* name, const uint4* table, uint64_t n_lines, uint32_t* state, uint32_t* action, uint32_t* out,
                int n = 65536) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  for (int i = 0; i < 100; ++i) k<DEPTH><<<n / 64, 64>>>(table, n_lines, state, action, out, i);
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) k<DEPTH><<<n / 64, 64>>>(table, n_lines, state, action, out, 1000 + i);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-46s n=%-7d %.2f us per launch\n", name, n, ms * 1e3 / iters);
}

int main() {
  const uint64_t bytes = 16ull << 30, n_lines = bytes / 256;
  uint4* table; uint32_t *state, *action, *out;
  if (hipMalloc(&table, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(table, 0x5A, bytes);
  (void)hipMalloc(&state, 262144 * 4); (void)hipMalloc(&action, 262144 * 4); (void)hipMalloc(&out, 262144 * 4);
  (void)hipMemset(state, 1, 262144 * 4); (void)hipMemset(action, 2, 262144 * 4);
  (void)hipDeviceSynchronize();
  run<0>("k0 empty kernel, 1,024 waves", table, n_lines, state, action, out);
  run<1>("k1 coalesced state read + write", table, n_lines, state, action, out);
  run<2>("k2 + 1 random HBM gather", table, n_lines, state, action, out);
  run<3>("k3 + 2 dependent random HBM gathers", table, n_lines, state, action, out);
  run<4>("k4 + 3 dependent random HBM gathers", table, n_lines, state, action, out);
  uint4* rec;
  (void)hipMalloc(&rec, 262144 * 16);
  (void)hipMemset(rec, 3, 262144 * 16);
  runc<1>("cooperative: 1 random 128-B line / env", table, n_lines * 2, state, action, out, rec);
  runc<2>("cooperative: 2 dependent random 128-B lines / env", table, n_lines * 2, state, action, out, rec);
  runc<3>("cooperative: 3 dependent random 128-B lines / env", table, n_lines * 2, state, action, out, rec);
  runc<2>("cooperative: 2 dependent lines / env", table, n_lines * 2, state, action, out, rec, 262144);
  for (int n : {4096, 16384, 262144}) {
    run<1>("k1 coalesced state read + write", table, n_lines, state, action, out, n);
    run<2>("k2 + 1 random HBM gather", table, n_lines, state, action, out, n);
    run<3>("k3 + 2 dependent random HBM gathers", table, n_lines, state, action, out, n);
  }
  return 0;
}
