// Dev check (not shipped): the hoisted-reciprocal quotient used by the MazeWorld ray-caster is bit-identical to
// hipcc's own fp64 division for operands in the ranges the ray-caster produces.  Prints the mismatch count.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ inline uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ inline double u01(uint64_t x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }

__global__ void check(unsigned long long* bad, int mode, uint64_t seed) {
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long nb = 0;
  for (int it = 0; it < 256; ++it) {
    const uint64_t h1 = mix(seed + gid * 512 + it * 2), h2 = mix(seed + gid * 512 + it * 2 + 1);
    double a, b;
    if (mode == 0) {        // tap weight: 10*dist / d2
      b = exp(log(1e-8) + u01(h2) * (log(1e3) - log(1e-8)));
      a = (h1 & 15) == 0 ? 0.0 : exp(log(1e-40) + u01(h1) * (log(1e4) - log(1e-40)));
    } else if (mode == 1) { // colour sum / weight sum
      b = 0.16 + u01(h2) * 15.84;
      a = (double)(float)(u01(h1) * 4080.0);
    } else {                // geometry: x / {cos, visibility, cell_size, l_focal}
      b = 0.05 + u01(h2) * 20.0;
      a = (u01(h1) - 0.3) * 200.0;
    }
    double y = __builtin_amdgcn_rcp(b);
    y = __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
    y = __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
    const double q0 = a * y;
    const double q = __builtin_fma(__builtin_fma(-b, q0, a), y, q0);
    const double ref = a / b;
    if (__double_as_longlong(q) != __double_as_longlong(ref)) ++nb;
  }
  if (nb) atomicAdd(bad, nb);
}

int main() {
  unsigned long long* d; unsigned long long h;
  hipMalloc(&d, 8);
  for (int mode = 0; mode < 3; ++mode) {
    hipMemset(d, 0, 8);
    for (int rep = 0; rep < 16; ++rep) check<<<4096, 256>>>(d, mode, 0x1234567ull * (rep + 1) + mode);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("mode %d: %llu mismatches in %llu quotients\n", mode, h, 16ull * 4096 * 256 * 256);
  }
  return 0;
}
