// Dev probe (not shipped): operand / result layout of v_mfma_f32_16x16x4_f32 on gfx950 and the order in which it adds
// the four k-products to the accumulator.  LinDS wants its products as fp32 fmaf chains in a FIXED order (the scalar
// kernel and the CPU oracle restate that order), so a kernel built on this instruction is only usable if the hardware
// order is a plain ascending chain  d = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c)))).
// Prints, for several hypotheses, how many of the 256 results match bit for bit.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* A /*[16][8]*/, const float* B /*[8][16]*/, const float* C /*[16][16]*/, float* D, int two) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  f32x4 acc;
  for (int r = 0; r < 4; ++r) acc[r] = C[(4 * g + r) * 16 + i];
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 8 + g], B[g * 16 + i], acc, 0, 0, 0);
  if (two) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 8 + 4 + g], B[(4 + g) * 16 + i], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = acc[r];
}

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main() {
  std::vector<float> A(16 * 8), B(8 * 16), C(256), D(256);
  uint64_t s = 12345;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; };
  for (auto& v : A) v = (float)((rnd() - 0.5) * std::pow(10.0, rnd() * 6 - 3));
  for (auto& v : B) v = (float)((rnd() - 0.5) * std::pow(10.0, rnd() * 6 - 3));
  for (auto& v : C) v = (float)((rnd() - 0.5) * std::pow(10.0, rnd() * 6 - 3));
  float *dA, *dB, *dC, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
  for (int two = 0; two < 2; ++two) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, two);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    const int K = two ? 8 : 4;
    int asc = 0, desc = 0, pair = 0, close = 0, transposed = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        float c = C[i * 16 + j], x = c;
        for (int k = 0; k < K; ++k) x = fmaf(A[i * 8 + k], B[k * 16 + j], x);
        asc += bits(x) == bits(D[i * 16 + j]);
        transposed += bits(x) == bits(D[j * 16 + i]);
        float y = c;
        for (int blk = 0; blk < K; blk += 4)
          for (int k = blk + 3; k >= blk; --k) y = fmaf(A[i * 8 + k], B[k * 16 + j], y);
        desc += bits(y) == bits(D[i * 16 + j]);
        float z = c;
        for (int blk = 0; blk < K; blk += 4) {
          const float p01 = fmaf(A[i * 8 + blk + 1], B[(blk + 1) * 16 + j], A[i * 8 + blk] * B[blk * 16 + j]);
          const float p23 = fmaf(A[i * 8 + blk + 3], B[(blk + 3) * 16 + j], A[i * 8 + blk + 2] * B[(blk + 2) * 16 + j]);
          z = z + (p01 + p23);
        }
        pair += bits(z) == bits(D[i * 16 + j]);
        close += std::fabs(x - D[i * 16 + j]) <= 1e-5f * std::fabs(x) + 1e-30f;
      }
    printf("mfma_f32_16x16x4_f32 x%d: ascending fma chain %d/256, descending %d/256, pairwise %d/256, transposed layout %d/256, "
           "within 1e-5 %d/256\n", two + 1, asc, desc, pair, transposed, close);
  }
  return 0;
}
