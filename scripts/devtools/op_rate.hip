// Dev microbenchmark (not shipped): VALU issue cost of the fp64/convert instructions the ray-caster uses.
// Every SIMD runs `waves` waves; each wave issues ITER x 8 independent copies of one instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 20000
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DD(i) asm volatile(ASMSTR : "+v"(a##i) : "v"(b0), "v"(c0));     /* f64 <- f64, f64 */
#define DF(i) asm volatile(ASMSTR : "+v"(a##i) : "v"(g0), "v"(g1));     /* f64 <- b32 */
#define FD(i) asm volatile(ASMSTR : "+v"(f##i) : "v"(b0), "v"(c0));     /* b32 <- f64 */
#define FF(i) asm volatile(ASMSTR : "+v"(f##i) : "v"(g0), "v"(g1));     /* b32 <- b32, b32 */
#define LOOP(KIND) for (int it = 0; it < ITER; ++it) { REP8(KIND) }

template <int OP>
__global__ void k(double* out, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
  double b0 = seed * 0.999, c0 = seed * 1e-3;
  float g0 = (float)seed * 0.5f, g1 = (float)seed * 0.25f;
#define ASMSTR "v_add_f64 %0, %1, %2"
  if (OP == 0) LOOP(DD)
#undef ASMSTR
#define ASMSTR "v_mul_f64 %0, %1, %2"
  if (OP == 1) LOOP(DD)
#undef ASMSTR
#define ASMSTR "v_fma_f64 %0, %1, %2, %0"
  if (OP == 2) LOOP(DD)
#undef ASMSTR
#define ASMSTR "v_max_f64 %0, %1, %2"
  if (OP == 3) LOOP(DD)
#undef ASMSTR
#define ASMSTR "v_rcp_f64 %0, %1"
  if (OP == 4) LOOP(DD)
#undef ASMSTR
#define ASMSTR "v_cvt_f64_u32 %0, %1"
  if (OP == 5) LOOP(DF)
#undef ASMSTR
#define ASMSTR "v_cvt_f64_f32 %0, %1"
  if (OP == 6) LOOP(DF)
#undef ASMSTR
#define ASMSTR "v_cvt_f32_f64 %0, %1"
  if (OP == 7) LOOP(FD)
#undef ASMSTR
#define ASMSTR "v_cndmask_b32 %0, %1, %2, vcc"
  if (OP == 8) LOOP(FF)
#undef ASMSTR
#define ASMSTR "v_bfe_u32 %0, %1, 8, 8"
  if (OP == 9) LOOP(FF)
#undef ASMSTR
#define ASMSTR "v_cvt_f32_ubyte1 %0, %1"
  if (OP == 10) LOOP(FF)
#undef ASMSTR
#define ASMSTR "v_fma_f32 %0, %1, %2, %0"
  if (OP == 11) LOOP(FF)
#undef ASMSTR
#define ASMSTR "v_pk_fma_f32 %0, %1, %2, %0"
  if (OP == 12) LOOP(DD)
#undef ASMSTR
#define ASMSTR "v_cvt_i32_f64 %0, %1"
  if (OP == 13) LOOP(FD)
#undef ASMSTR
#define ASMSTR "v_floor_f64 %0, %1"
  if (OP == 14) LOOP(DD)
#undef ASMSTR
#define ASMSTR "v_div_scale_f64 %0, vcc, %1, %1, %2"
  if (OP == 15) LOOP(DD)
#undef ASMSTR
#define ASMSTR "v_div_fmas_f64 %0, %1, %2, %0"
  if (OP == 16) LOOP(DD)
#undef ASMSTR
#define ASMSTR "v_div_fixup_f64 %0, %1, %2, %0"
  if (OP == 17) LOOP(DD)
#undef ASMSTR
#define ASMSTR "v_cndmask_b32_e64 %0, %1, %2, s[20:21]"
  if (OP == 18) { asm volatile("s_mov_b64 s[20:21], 0x55" ::: "s20", "s21"); LOOP(FF) }
#undef ASMSTR
#define ASMSTR "v_cndmask_b32_e32 %0, %1, %2, vcc"
  if (OP == 19) { asm volatile("s_mov_b64 vcc, 0x55" ::: "vcc"); LOOP(FF) }
#undef ASMSTR
#define ASMSTR "v_cmp_lt_f64 vcc, %1, %2"
  if (OP == 20) LOOP(DD)
#undef ASMSTR
#define ASMSTR "v_and_b32 %0, %1, %2"
  if (OP == 21) LOOP(FF)
#undef ASMSTR
#define ASMSTR "v_add_f32 %0, %1, %2"
  if (OP == 22) LOOP(FF)
#undef ASMSTR
#define ASMSTR "v_cndmask_b32_e64 %0, %1, %2, vcc"
  if (OP == 24) { asm volatile("s_mov_b64 vcc, 0x55" ::: "vcc"); LOOP(FF) }
#undef ASMSTR
#define ASMSTR "v_cndmask_b32_e32 %0, %1, %2, vcc\n v_add_f32 %0, %0, %0\n"
  if (OP == 25) { asm volatile("s_mov_b64 vcc, 0x55" ::: "vcc"); LOOP(FF) }
#undef ASMSTR
#define ASMSTR "v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32_e32 %0, %1, %2, vcc\n"
  if (OP == 26) { LOOP(FF) }
#undef ASMSTR
#define ASMSTR "v_cmp_lt_f32 s[20:21], %1, %2\n v_cndmask_b32_e64 %0, %1, %2, s[20:21]\n"
  if (OP == 27) { LOOP(FF) }
#undef ASMSTR
#define ASMSTR "v_mov_b32 %0, %1"
  if (OP == 23) LOOP(FF)
#undef ASMSTR
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
}

template <int OP>
void run(const char* name, double* d, int waves) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 4 * waves / 4;   // 256-thread blocks = 4 waves
  k<OP><<<blocks, 256>>>(d, 1.5);
  (void)hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(d, 1.5);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double inst_per_simd = (double)waves * ITER * 8;
  printf("%-22s waves/SIMD %d: %.2f ns per wave-instruction per SIMD (x2.4 = %.1f cycles at 2.4 GHz)\n", name, waves,
         ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
}

int main() {
  double* d;
  (void)hipMalloc(&d, 256 * 4 * 8 * 64 * 8);
  for (int waves : {1, 4}) {
    run<0>("v_add_f64", d, waves); run<1>("v_mul_f64", d, waves); run<2>("v_fma_f64", d, waves);
    run<3>("v_max_f64", d, waves); run<4>("v_rcp_f64", d, waves); run<5>("v_cvt_f64_u32", d, waves);
    run<6>("v_cvt_f64_f32", d, waves); run<7>("v_cvt_f32_f64", d, waves); run<8>("v_cndmask_b32", d, waves);
    run<9>("v_bfe_u32", d, waves); run<10>("v_cvt_f32_ubyte1", d, waves); run<11>("v_fma_f32", d, waves);
    run<12>("v_pk_fma_f32", d, waves); run<13>("v_cvt_i32_f64", d, waves); run<14>("v_floor_f64", d, waves);
    run<18>("v_cndmask_e64 sgpr", d, waves); run<19>("v_cndmask_e32 vcc set", d, waves); run<20>("v_cmp_lt_f64", d, waves);
    run<21>("v_and_b32", d, waves); run<22>("v_add_f32", d, waves); run<23>("v_mov_b32", d, waves);
    run<24>("v_cndmask_e64 vcc", d, waves); run<25>("cndmask_e32 vcc + add_f32 (2 instr)", d, waves);
    run<26>("cmp->vcc + cndmask_e32 (2 instr)", d, waves); run<27>("cmp->sgpr + cndmask_e64 (2 instr)", d, waves);
    run<15>("v_div_scale_f64", d, waves); run<16>("v_div_fmas_f64", d, waves); run<17>("v_div_fixup_f64", d, waves);
  }
  return 0;
}
