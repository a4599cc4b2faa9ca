// philox.h — counter-based RNG on the device: Philox4x32-10 (Salmon et al., SC'11) and the draw
// conventions shared bit-for-bit with oracle/xeno_oracle.c (xo_philox4x32_10, xo_env_draw, xo_u53).
//
// Replaces numpy's process-global MT19937 stream used by the reference at anymdp_env.py:89,100,105,
// linds_env.py:79 and random_cartpole.py:70.  counter = {env_gid_lo, env_gid_hi, tick_lo,
// purpose | tick_hi<<8}, key = engine seed: a draw is a pure function of (seed, global env id, launch
// tick, purpose), so results do not depend on launch geometry or on how envs are sharded over GPUs.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

struct xv_u32x4 {
  uint32_t x, y, z, w;
};

__host__ __device__ __forceinline__ xv_u32x4 xv_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                                uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return xv_u32x4{c0, c1, c2, c3};
}

// purposes (low byte of counter word 3)
#define XV_DRAW_STEP 0u   // words 0,1 -> transition uniform; words 2,3 -> Box-Muller normal
#define XV_DRAW_RESET 1u  // words 0,1 -> initial-state uniform
#define XV_DRAW_NOISE 16u // + g, g = 0..3: eight normals, the linds process noise of components 16 m + 4 g + r (below)

__host__ __device__ __forceinline__ xv_u32x4 xv_env_draw(uint64_t seed, uint64_t gid, uint64_t tick,
                                                           uint32_t purpose) {
  return xv_philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)tick,
                          (purpose & 0xFFu) | ((uint32_t)(tick >> 32) << 8), (uint32_t)seed,
                          (uint32_t)(seed >> 32));
}

// a draw family with a wide sub-index (ray x cell of the maze exposure map, cell of the memory-keep map): the
// sub-index goes into the key, so that every (purpose, sub) is its own Philox stream
#define XV_DRAW_EXPOSE 5u  // sub = 64 * column + (k >> 2): word k & 3 decides the k-th cell the column's ray lists
#define XV_DRAW_KEEP 6u    // sub = cell >> 2: word cell & 3 -> long-term-memory keep draw of the maze teacher
__host__ __device__ __forceinline__ xv_u32x4 xv_env_draw_sub(uint64_t seed, uint64_t gid, uint64_t tick,
                                                               uint32_t purpose, uint32_t sub) {
  return xv_philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)tick,
                          (purpose & 0xFFu) | ((uint32_t)(tick >> 32) << 8), (uint32_t)seed,
                          (uint32_t)(seed >> 32) ^ (0x80000000u | sub));
}

// numpy legacy random_sample: two 32-bit words -> 53-bit double in [0,1)
__host__ __device__ __forceinline__ double xv_u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// Box-Muller from two words; u1 in (0,1], u2 in [0,1)
__device__ __forceinline__ void xv_box_muller(uint32_t a, uint32_t b, float* z0, float* z1) {
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
  const float r = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincospif(2.0f * u2, &sn, &cs);
  *z0 = r * cs;
  *z1 = r * sn;
}
// The same pair from the transcendental unit: v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in
// revolutions, so 2 pi u2 needs no range reduction): ~10 instructions instead of ~70 for the libm-grade pair above.
// Absolute error of z a few 1e-7 (the hardware sin/cos are accurate to ~1e-6 absolute, log2/sqrt to 1 ulp); used where
// z enters a result scaled far below that result's tolerance (LinDS process noise: noise_drift * dt * z with
// noise_drift * dt <= 2e-3 against 1e-5 relative on a state of order 0.1-1).
__device__ __forceinline__ void xv_box_muller_fast(uint32_t a, uint32_t b, float* z0, float* z1) {
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
  const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // -2 ln 2 * log2(u1)
  *z0 = r * __builtin_amdgcn_cosf(u2);
  *z1 = r * __builtin_amdgcn_sinf(u2);
}
// Eight normals from ONE Philox call (LinDS process noise, round 3): word p -> Box-Muller pair p, radius from the high 16
// bits (u1 = (hi + 1) / 65536 in (0, 1]), angle from the low 16 (u2 = lo / 65536 revolutions).  65,536 radii x 65,536
// angles per pair, |z| <= 4.71; the noise enters scaled by noise_drift * dt <= 2e-3.  Philox's 32-bit multiplies run at a
// quarter of the VALU rate and were ~40 % of the LinDS step's instruction issue with two calls per lane.
__device__ __forceinline__ void xv_box_muller16(uint32_t w, float* z0, float* z1) {
  const float u1 = ((float)(w >> 16) + 1.0f) * (1.0f / 65536.0f);
  const float u2 = (float)(w & 0xFFFFu) * (1.0f / 65536.0f);
  const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // -2 ln 2 * log2(u1)
  *z0 = r * __builtin_amdgcn_cosf(u2);
  *z1 = r * __builtin_amdgcn_sinf(u2);
}
__device__ __forceinline__ float xv_normal1(uint32_t a, uint32_t b) {
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
}
