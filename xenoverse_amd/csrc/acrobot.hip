// acrobot.hip — domain-randomised Acrobot batched step / reset for gfx950 and its C-ABI.
//
// Reproduces xenoverse/metacontrol/random_acrobot.py: _dsdt :58-96 (book dynamics), _terminal :98-101,
// set_task :103-106, step :108-117 (= `frameskip` repeats of gymnasium's AcrobotEnv.step), reset :119-130.
// The integrator around _dsdt (rk4 over [0, 0.2], wrap to [-pi, pi], velocity bounds 4pi / 9pi, torques
// {-1, 0, +1}) is gymnasium's (third-party, not vendored, not installed here: restated from the public 1.x
// source — that part of the parity is unpinned); _dsdt and _terminal are pinned to the reference's own code
// through oracle/xeno_oracle.c (tests/golden/acrobot_dsdt.npz).
// One lane per env, fp64 state (the reference integrates in float64) in component-major arrays, 56 B of task
// parameters.  Same operation order as oracle/xeno_oracle.c: acrobot_substep.
#include "acrobot_wrap.h"
#include "philox.h"
#include "xv_common.h"

#define AC_PI 3.141592653589793

struct AcrobotArgs {
  const double* params;      // [n_task][7]: l1, l2, m1, m2, lc1, lc2, g
  double reset_scale[4];
  int scale_is_vector;
  const int32_t* env_task;
  double* state;             // [4][n_env]
  uint8_t* fresh;            // state still holds the float32 reset values
  int32_t* steps;
  uint8_t* need_reset;
  uint32_t* err;
  int n_env, n_task, frameskip, max_steps;
  uint64_t seed, gid_base, tick;
  const uint64_t* tick_dev;   // device tick mode of the engine: the launch tick is *tick_dev + tick (xv_launch_tick)
};

struct AcrobotIO {
  const int32_t* action;
  const double* u_reset;     // [4][n_env] (INJECT)
  float* obs;                // [n_env][6]
  float* reward;
  uint8_t* terminated;
  uint8_t* truncated;
  float* final_obs;          // nullable
  uint8_t* done_out;         // nullable: terminated | truncated of the same step (xv_acrobot_step_info)
};

struct xv_acrobot {
  xv_engine* eng;
  AcrobotArgs a;
};

struct AcrobotTask {
  double l1, m1, m2, lc1, lc2, g, I1, I2;
};

// sin and cos of an angle of this system in ~45 fp64 instructions (ocml's sincos is ~200 with its large-argument path,
// and a step evaluates 16 of them: they were four fifths of the kernel).  The classic two-step Cody-Waite reduction by
// pi/2 (x - n pio2_1 is exact: 33-bit constant, |n| < 2^20) followed by the degree-13 / degree-14 minimax kernels with the
// reduction's tail — the construction of fdlibm / msun (k_sin.c, k_cos.c, e_rem_pio2.c: public constants), restated.
// Within 1 ulp of glibc's sin / cos (96.8 % of 8e6 arguments in +-1000 identical, the rest 1 ulp off; the oracle calls
// glibc).  Domain: |x| < ~1e6 — the state is wrapped to [-pi, pi] and the velocities are bounded by 9 pi.
__device__ __forceinline__ void ac_sincos(double x, double* sn, double* cs) {
  const double fn = rint(x * 6.36619772367581382433e-01);
  double r = x - fn * 1.57079632673412561417e+00;
  const double t = r;
  double w = fn * 6.07710050630396597660e-11;
  r = t - w;
  w = fn * 2.02226624879595063154e-21 - ((t - r) - w);
  const double y0 = r - w, y1 = (r - y0) - w;
  const double z = y0 * y0, v = z * y0;
  const double rs = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                    z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  const double s = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * -1.66666666666666324348e-01);
  const double rc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double hz = 0.5 * z, wv = 1.0 - hz;
  const double c = wv + (((1.0 - wv) - hz) + (z * rc - y0 * y1));
  const int n = (int)fn & 3;
  const double a = (n & 1) ? c : s, b = (n & 1) ? s : c;
  *sn = (n & 2) ? -a : a;
  *cs = ((n + 1) & 2) ? -b : b;
}
__device__ __forceinline__ double ac_cos(double x) { double sn, cs; ac_sincos(x, &sn, &cs); return cs; }

// random_acrobot.py:58-96; y = (theta1, theta2, dtheta1, dtheta2), a = torque
__device__ __forceinline__ void acrobot_dsdt(const AcrobotTask& K, const double (&y)[4], double a, double (&out)[4]) {
  const double theta1 = y[0], theta2 = y[1], dtheta1 = y[2], dtheta2 = y[3];
  double s2, c2;
  ac_sincos(theta2, &s2, &c2);
  const double d1 = K.m1 * (K.lc1 * K.lc1) + K.m2 * (K.l1 * K.l1 + K.lc2 * K.lc2 + 2 * K.l1 * K.lc2 * c2) + K.I1 + K.I2;
  const double d2 = K.m2 * (K.lc2 * K.lc2 + K.l1 * K.lc2 * c2) + K.I2;
  const double phi2 = K.m2 * K.lc2 * K.g * ac_cos(theta1 + theta2 - AC_PI / 2.0);
  const double phi1 = -K.m2 * K.l1 * K.lc2 * (dtheta2 * dtheta2) * s2 - 2 * K.m2 * K.l1 * K.lc2 * dtheta2 * dtheta1 * s2 +
                      (K.m1 * K.lc1 + K.m2 * K.l1) * K.g * ac_cos(theta1 - AC_PI / 2) + phi2;
  const double ddtheta2 = (a + d2 / d1 * phi1 - K.m2 * K.l1 * K.lc2 * (dtheta1 * dtheta1) * s2 - phi2) /
                          (K.m2 * (K.lc2 * K.lc2) + K.I2 - d2 * d2 / d1);
  const double ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;
  out[0] = dtheta1; out[1] = dtheta2; out[2] = ddtheta1; out[3] = ddtheta2;
}

__device__ __forceinline__ double acrobot_bound(double x, double m, double M) {
  const double t = (m > x) ? m : x;
  return (M < t) ? M : t;
}

// observation of a state: float32 math while the state is the float32 reset array, float64 afterwards
__device__ __forceinline__ void acrobot_obs(const double (&s)[4], bool f32, float (&o)[6]) {
  if (f32) {
    float sn, cs;
    sincosf((float)s[0], &sn, &cs); o[0] = cs; o[1] = sn;
    sincosf((float)s[1], &sn, &cs); o[2] = cs; o[3] = sn;
  } else {
    double sn, cs;
    ac_sincos(s[0], &sn, &cs); o[0] = (float)cs; o[1] = (float)sn;
    ac_sincos(s[1], &sn, &cs); o[2] = (float)cs; o[3] = (float)sn;
  }
  o[4] = (float)s[2]; o[5] = (float)s[3];
}

template <bool INJECT>
__device__ __forceinline__ void acrobot_reset_state(const AcrobotArgs& P, const double* u_in, int i, uint64_t tick, double (&s)[4]) {
  double u[4];
  if (INJECT) {
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = u_in[(size_t)k * P.n_env + i];
  } else {
    const xv_u32x4 w = xv_env_draw(P.seed, P.gid_base + (uint64_t)i, tick, XV_DRAW_RESET);
    const xv_u32x4 v = xv_env_draw(P.seed, P.gid_base + (uint64_t)i, tick, 3u);
    u[0] = xv_u53(w.x, w.y); u[1] = xv_u53(w.z, w.w); u[2] = xv_u53(v.x, v.y); u[3] = xv_u53(v.z, v.w);
  }
  // state = uniform(-1, 1, 4).astype(float32) * reset_bounds_scale   (:123-125)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float f = (float)(-1.0 + 2.0 * u[k]);
    s[k] = P.scale_is_vector ? (double)f * P.reset_scale[k] : (double)(f * (float)P.reset_scale[k]);
  }
}

__device__ __forceinline__ void acrobot_store_obs(float* dst, const float (&o)[6]) {
  float2* d2 = reinterpret_cast<float2*>(dst);
  d2[0] = make_float2(o[0], o[1]); d2[1] = make_float2(o[2], o[3]); d2[2] = make_float2(o[4], o[5]);
}

template <bool INJECT>
__global__ __launch_bounds__(64) void acrobot_step_kernel(AcrobotArgs P, AcrobotIO io, int mode, int T) {
  // T steps per launch (xv_acrobot_rollout; T = 1 for xv_acrobot_step), state in registers, step ts draws with tick + ts
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  const size_t N = (size_t)P.n_env;
  double s[4] = {P.state[i], P.state[N + i], P.state[2 * N + i], P.state[3 * N + i]};
  int steps = P.steps[i];
  int nr = P.need_reset[i];
  int fresh = P.fresh[i];
  uint32_t err = 0;
  const double* prm = P.params + (size_t)P.env_task[i] * 7;
  AcrobotTask K;
  K.l1 = prm[0]; K.m1 = prm[2]; K.m2 = prm[3]; K.lc1 = prm[4]; K.lc2 = prm[5]; K.g = prm[6];
  const double l2 = prm[1];
  K.I1 = K.m1 * (K.lc1 * K.lc1 + (K.l1 - K.lc1) * (K.l1 - K.lc1)) / 6.0;
  K.I2 = K.m2 * (K.lc2 * K.lc2 + (l2 - K.lc2) * (l2 - K.lc2)) / 6.0;
  for (int ts = 0; ts < T; ++ts) {
  const size_t ob = (size_t)ts * N + i;
  int action = io.action[ob];
  float reward = 0.0f;
  int term = 0, trunc = 0;
  float fobs[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  bool do_reset = false;
  if (mode == XV_AUTORESET_NEXT_STEP && nr) {
    do_reset = true;
  } else {
    if (action < 0 || action > 2) {
      err |= XV_DEVERR_ACTION_RANGE;
      action = action < 0 ? 0 : 2;
    }
    const double torque = (double)(action - 1);   // AVAIL_TORQUE = [-1.0, 0.0, +1]
    const double dt = 0.2, dt2 = dt / 2.0;
    double total = 0.0;
    for (int f = 0; f < P.frameskip; ++f) {       // :112-116
      double k1[4], k2[4], k3[4], k4[4], y[4];
      acrobot_dsdt(K, s, torque, k1);
#pragma unroll
      for (int q = 0; q < 4; ++q) y[q] = s[q] + dt2 * k1[q];
      acrobot_dsdt(K, y, torque, k2);
#pragma unroll
      for (int q = 0; q < 4; ++q) y[q] = s[q] + dt2 * k2[q];
      acrobot_dsdt(K, y, torque, k3);
#pragma unroll
      for (int q = 0; q < 4; ++q) y[q] = s[q] + dt * k3[q];
      acrobot_dsdt(K, y, torque, k4);
#pragma unroll
      for (int q = 0; q < 4; ++q) y[q] = s[q] + dt / 6.0 * (k1[q] + 2 * k2[q] + 2 * k3[q] + k4[q]);
      int stuck = 0;                              // wrap(x, -pi, pi): the reference's loop, evaluated exactly
      s[0] = xv_acrobot_wrap(y[0], &stuck);       // in O(log x) steps (acrobot_wrap.h)
      s[1] = xv_acrobot_wrap(y[1], &stuck);
      if (stuck) err |= XV_DEVERR_NONFINITE;      // the reference would loop forever here
      s[2] = acrobot_bound(y[2], -4 * AC_PI, 4 * AC_PI);
      s[3] = acrobot_bound(y[3], -9 * AC_PI, 9 * AC_PI);
      term = (-ac_cos(s[0]) - ac_cos(s[1] + s[0]) > K.l1) ? 1 : 0;   // _terminal :98-101
      total += term ? 0.0 : -1.0;
      if (term) break;
    }
    reward = (float)total;
    fresh = 0;
    steps += 1;
    trunc = (P.max_steps > 0 && steps >= P.max_steps) ? 1 : 0;
    if (!(fabs(s[2]) <= 1.0e300) || !(fabs(s[0]) <= 1.0e300)) err |= XV_DEVERR_NONFINITE;
    if (term || trunc) {
      if (mode == XV_AUTORESET_SAME_STEP) {
        acrobot_obs(s, false, fobs);
        do_reset = true;
      } else if (mode == XV_AUTORESET_NEXT_STEP) {
        nr = 1;
      }
    }
  }
  if (do_reset) {
    acrobot_reset_state<INJECT>(P, io.u_reset, i, xv_launch_tick(P.tick, P.tick_dev) + (uint64_t)ts, s);
    fresh = 1;
    steps = 0;
    nr = 0;
  }
  float o[6];
  acrobot_obs(s, fresh && !P.scale_is_vector, o);
  acrobot_store_obs(io.obs + ob * 6, o);
  io.reward[ob] = reward;
  io.terminated[ob] = (uint8_t)term;
  io.truncated[ob] = (uint8_t)trunc;
  if (io.done_out) io.done_out[ob] = (uint8_t)((term || trunc) ? 1 : 0);
  if (io.final_obs) acrobot_store_obs(io.final_obs + ob * 6, fobs);
  }
  P.state[i] = s[0]; P.state[N + i] = s[1]; P.state[2 * N + i] = s[2]; P.state[3 * N + i] = s[3];
  P.steps[i] = steps;
  P.need_reset[i] = (uint8_t)nr;
  P.fresh[i] = (uint8_t)fresh;
  if (err) atomicOr(P.err, err);
}

template <bool INJECT>
__global__ __launch_bounds__(64) void acrobot_reset_kernel(AcrobotArgs P, const uint8_t* mask, const double* u, float* obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  if (mask && !mask[i]) return;
  const size_t N = (size_t)P.n_env;
  double s[4];
  acrobot_reset_state<INJECT>(P, u, i, xv_launch_tick(P.tick, P.tick_dev), s);
  P.state[i] = s[0]; P.state[N + i] = s[1]; P.state[2 * N + i] = s[2]; P.state[3 * N + i] = s[3];
  P.steps[i] = 0;
  P.need_reset[i] = 0;
  P.fresh[i] = 1;
  if (obs) {
    float o[6];
    acrobot_obs(s, !P.scale_is_vector, o);
    acrobot_store_obs(obs + (size_t)i * 6, o);
  }
}

extern "C" int xv_acrobot_create(xv_engine* e, int n_env, int n_task, int frameskip, int max_steps, const double* params,
                                 const double* reset_scale, int scale_is_vector, const int32_t* env_task,
                                 xv_acrobot** out) {
  XV_CHECK_ARG(out != nullptr);
  *out = nullptr;
  XV_CHECK_ARG(e && params && reset_scale && env_task && n_env > 0 && n_task > 0 && frameskip >= 1);
  XV_HIP(hipSetDevice(e->device));
  xv_acrobot* h = new (std::nothrow) xv_acrobot();
  if (!h) {
    xv_set_error("xv_acrobot_create: out of host memory");
    return XV_ERR_NOMEM;
  }
  h->eng = e;
  AcrobotArgs& a = h->a;
  a.params = params; a.env_task = env_task;
  a.n_env = n_env; a.n_task = n_task; a.frameskip = frameskip; a.max_steps = max_steps;
  a.scale_is_vector = scale_is_vector ? 1 : 0;
  a.err = e->d_err;
  a.seed = e->seed; a.gid_base = e->env_id_base; a.tick = 0; a.tick_dev = nullptr;
  XV_HIP(hipMemcpyAsync(a.reset_scale, reset_scale, sizeof(a.reset_scale), hipMemcpyDeviceToHost, e->stream));
  XV_HIP(hipStreamSynchronize(e->stream));
  a.state = nullptr; a.steps = nullptr; a.need_reset = nullptr; a.fresh = nullptr;
  const size_t n = (size_t)n_env;
  hipError_t m = hipMalloc(&a.state, sizeof(double) * 4 * n);
  if (m == hipSuccess) m = hipMalloc(&a.steps, sizeof(int32_t) * n);
  if (m == hipSuccess) m = hipMalloc(&a.need_reset, n);
  if (m == hipSuccess) m = hipMalloc(&a.fresh, n);
  if (m == hipSuccess) m = hipMemsetAsync(a.state, 0, sizeof(double) * 4 * n, e->stream);
  if (m == hipSuccess) m = hipMemsetAsync(a.steps, 0, sizeof(int32_t) * n, e->stream);
  if (m == hipSuccess) m = hipMemsetAsync(a.need_reset, 1, n, e->stream);
  if (m == hipSuccess) m = hipMemsetAsync(a.fresh, 0, n, e->stream);
  if (m != hipSuccess) {
    xv_set_error("xv_acrobot_create: device allocation failed: %s", hipGetErrorString(m));
    void* ps[] = {a.state, a.steps, a.need_reset, a.fresh};
    for (void* q : ps) if (q) (void)hipFree(q);
    delete h;
    return XV_ERR_HIP;
  }
  *out = h;
  return XV_OK;
}

extern "C" int xv_acrobot_destroy(xv_acrobot* h) {
  if (!h) return XV_OK;
  (void)hipSetDevice(h->eng->device);
  (void)hipStreamSynchronize(h->eng->stream);
  void* ps[] = {h->a.state, h->a.steps, h->a.need_reset, h->a.fresh};
  for (void* q : ps) if (q) (void)hipFree(q);
  delete h;
  return XV_OK;
}

static inline void acrobot_bind_rng(xv_acrobot* h, uint64_t ticks) {
  h->a.seed = h->eng->seed;
  h->a.gid_base = h->eng->env_id_base;
  const XvTickBind b = xv_engine_bind_tick(h->eng, ticks);
  h->a.tick = b.tick;
  h->a.tick_dev = b.tick_dev;
}

extern "C" int xv_acrobot_reset(xv_acrobot* h, const uint8_t* mask, float* obs) {
  XV_CHECK_ARG(h != nullptr);
  acrobot_bind_rng(h, 1);
  hipLaunchKernelGGL(acrobot_reset_kernel<false>, dim3(xv_div_up(h->a.n_env, 64)), dim3(64), 0, h->eng->stream, h->a, mask,
                     (const double*)nullptr, obs);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_acrobot_reset_injected(xv_acrobot* h, const uint8_t* mask, const double* u, float* obs) {
  XV_CHECK_ARG(h != nullptr && u != nullptr);
  acrobot_bind_rng(h, 0);
  hipLaunchKernelGGL(acrobot_reset_kernel<true>, dim3(xv_div_up(h->a.n_env, 64)), dim3(64), 0, h->eng->stream, h->a, mask,
                     u, obs);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_acrobot_step(xv_acrobot* h, const int32_t* action, float* obs, float* reward, uint8_t* terminated,
                               uint8_t* truncated, float* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && action && obs && reward && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  acrobot_bind_rng(h, 1);
  AcrobotIO io{action, nullptr, obs, reward, terminated, truncated, final_obs};
  hipLaunchKernelGGL(acrobot_step_kernel<false>, dim3(xv_div_up(h->a.n_env, 64)), dim3(64), 0, h->eng->stream, h->a, io,
                     autoreset_mode, 1);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

// xv_acrobot_step that also writes the terminated | truncated mask from the same launch (done uint8[n_env], nullable)
extern "C" int xv_acrobot_step_info(xv_acrobot* h, const int32_t* action, float* obs, float* reward, uint8_t* terminated,
                                    uint8_t* truncated, float* final_obs, uint8_t* done, int autoreset_mode) {
  XV_CHECK_ARG(h && action && obs && reward && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  acrobot_bind_rng(h, 1);
  AcrobotIO io{action, nullptr, obs, reward, terminated, truncated, final_obs, done};
  hipLaunchKernelGGL(acrobot_step_kernel<false>, dim3(xv_div_up(h->a.n_env, 64)), dim3(64), 0, h->eng->stream, h->a, io,
                     autoreset_mode, 1);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_acrobot_rollout(xv_acrobot* h, int T, const int32_t* action, float* obs, float* reward,
                                  uint8_t* terminated, uint8_t* truncated, float* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && action && obs && reward && terminated && truncated && T > 0);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  acrobot_bind_rng(h, (uint64_t)T);
  AcrobotIO io{action, nullptr, obs, reward, terminated, truncated, final_obs};
  hipLaunchKernelGGL(acrobot_step_kernel<false>, dim3(xv_div_up(h->a.n_env, 64)), dim3(64), 0, h->eng->stream, h->a, io,
                     autoreset_mode, T);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_acrobot_step_injected(xv_acrobot* h, const int32_t* action, const double* u_reset, float* obs,
                                        float* reward, uint8_t* terminated, uint8_t* truncated, float* final_obs,
                                        int autoreset_mode) {
  XV_CHECK_ARG(h && action && u_reset && obs && reward && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  acrobot_bind_rng(h, 0);
  AcrobotIO io{action, u_reset, obs, reward, terminated, truncated, final_obs};
  hipLaunchKernelGGL(acrobot_step_kernel<true>, dim3(xv_div_up(h->a.n_env, 64)), dim3(64), 0, h->eng->stream, h->a, io,
                     autoreset_mode, 1);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_acrobot_get_state(xv_acrobot* h, double* state, int32_t* steps, uint8_t* need_reset) {
  XV_CHECK_ARG(h != nullptr);
  const size_t n = (size_t)h->a.n_env;
  if (state) XV_HIP(hipMemcpyAsync(state, h->a.state, n * 32, hipMemcpyDeviceToDevice, h->eng->stream));
  if (steps) XV_HIP(hipMemcpyAsync(steps, h->a.steps, n * 4, hipMemcpyDeviceToDevice, h->eng->stream));
  if (need_reset) XV_HIP(hipMemcpyAsync(need_reset, h->a.need_reset, n, hipMemcpyDeviceToDevice, h->eng->stream));
  return XV_OK;
}

extern "C" int xv_acrobot_set_state(xv_acrobot* h, const double* state, const int32_t* steps, const uint8_t* need_reset) {
  XV_CHECK_ARG(h != nullptr);
  const size_t n = (size_t)h->a.n_env;
  if (state) {
    XV_HIP(hipMemcpyAsync(h->a.state, state, n * 32, hipMemcpyDeviceToDevice, h->eng->stream));
    XV_HIP(hipMemsetAsync(h->a.fresh, 0, n, h->eng->stream));   // a caller-supplied state is a float64 array
  }
  if (steps) XV_HIP(hipMemcpyAsync(h->a.steps, steps, n * 4, hipMemcpyDeviceToDevice, h->eng->stream));
  if (need_reset) XV_HIP(hipMemcpyAsync(h->a.need_reset, need_reset, n, hipMemcpyDeviceToDevice, h->eng->stream));
  return XV_OK;
}
