// cartpole.hip — domain-randomised CartPole batched step / reset for gfx950 and its C-ABI.
//
// Reproduces xenoverse/metacontrol/random_cartpole.py (set_task :46-50, step :52-61 = `frameskip` repeats of
// gymnasium's CartPoleEnv.step, reset :63-75).  The physics equations are gymnasium's (third-party, not
// vendored, not installed here: restated from the public 1.x source — parity unpinned).
// One lane per env, 4 fp32 state words in component-major arrays (coalesced), 16 B of task parameters.
// Same operation order as oracle/xeno_oracle.c: cartpole_step_one.
#include "philox.h"
#include "xv_common.h"

struct CartPoleArgs {
  const float4* params;      // [n_task]: gravity, masscart, masspole, length
  float4 reset_scale;
  const int32_t* env_task;
  float* state;              // [4][n_env]
  int32_t* steps;
  uint8_t* need_reset;
  uint32_t* err;
  int n_env, n_task, frameskip, max_steps;
  uint64_t seed, gid_base, tick;
};

struct CartPoleIO {
  const int32_t* action;
  const float* u_reset;      // [4][n_env] (INJECT)
  float* obs;                // [n_env][4]
  float* reward;
  uint8_t* terminated;
  uint8_t* truncated;
  float* final_obs;          // nullable
};

struct xv_cartpole {
  xv_engine* eng;
  CartPoleArgs a;
};

template <bool INJECT>
__device__ __forceinline__ float4 cartpole_reset_state(const CartPoleArgs& P, const float* u_in, int i) {
  float u0, u1, u2, u3;
  if (INJECT) {
    u0 = u_in[i]; u1 = u_in[(size_t)P.n_env + i]; u2 = u_in[(size_t)2 * P.n_env + i]; u3 = u_in[(size_t)3 * P.n_env + i];
  } else {
    const xv_u32x4 w = xv_env_draw(P.seed, P.gid_base + (uint64_t)i, P.tick, XV_DRAW_RESET);
    u0 = (float)(w.x >> 8) * (1.0f / 16777216.0f); u1 = (float)(w.y >> 8) * (1.0f / 16777216.0f);
    u2 = (float)(w.z >> 8) * (1.0f / 16777216.0f); u3 = (float)(w.w >> 8) * (1.0f / 16777216.0f);
  }
  // state = uniform(-1, 1, 4) * reset_bounds_scale   (random_cartpole.py:70)
  return make_float4(fmaf(2.0f, u0, -1.0f) * P.reset_scale.x, fmaf(2.0f, u1, -1.0f) * P.reset_scale.y,
                     fmaf(2.0f, u2, -1.0f) * P.reset_scale.z, fmaf(2.0f, u3, -1.0f) * P.reset_scale.w);
}

template <bool INJECT>
__global__ __launch_bounds__(256) void cartpole_step_kernel(CartPoleArgs P, CartPoleIO io, int mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  const size_t N = (size_t)P.n_env;
  float x = P.state[i], xd = P.state[N + i], th = P.state[2 * N + i], thd = P.state[3 * N + i];
  int steps = P.steps[i];
  int nr = P.need_reset[i];
  int action = io.action[i];
  const float4 prm = P.params[P.env_task[i]];
  float reward = 0.0f;
  int term = 0, trunc = 0;
  float4 fobs = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  bool do_reset = false;
  uint32_t err = 0;
  if (mode == XV_AUTORESET_NEXT_STEP && nr) {
    do_reset = true;
  } else {
    if (action != 0 && action != 1) {
      err |= XV_DEVERR_ACTION_RANGE;
      action = action > 0 ? 1 : 0;
    }
    const float gravity = prm.x, masscart = prm.y, masspole = prm.z, length = prm.w;
    const float polemass_length = masspole * length;   // :49
    const float total_mass = masspole + masscart;      // :50
    const float force = action == 1 ? 10.0f : -10.0f;
    for (int f = 0; f < P.frameskip; ++f) {            // :56-60
      float sn, cs;
      sincosf(th, &sn, &cs);
      const float temp = (force + polemass_length * thd * thd * sn) / total_mass;
      const float thacc = (gravity * sn - cs * temp) / (length * (4.0f / 3.0f - masspole * cs * cs / total_mass));
      const float xacc = temp - polemass_length * thacc * cs / total_mass;
      x = x + 0.02f * xd;
      xd = xd + 0.02f * xacc;
      th = th + 0.02f * thd;
      thd = thd + 0.02f * thacc;
      term = ((x < -2.4f) || (x > 2.4f) || (th < -0.20943951f) || (th > 0.20943951f)) ? 1 : 0;
      reward += 1.0f;
      if (term) break;
    }
    steps += 1;
    trunc = (P.max_steps > 0 && steps >= P.max_steps) ? 1 : 0;
    if (!(fabsf(x) <= 3.0e38f) || !(fabsf(thd) <= 3.0e38f)) err |= XV_DEVERR_NONFINITE;
    if (term || trunc) {
      if (mode == XV_AUTORESET_SAME_STEP) {
        fobs = make_float4(x, xd, th, thd);
        do_reset = true;
      } else if (mode == XV_AUTORESET_NEXT_STEP) {
        nr = 1;
      }
    }
  }
  if (do_reset) {
    const float4 s0 = cartpole_reset_state<INJECT>(P, io.u_reset, i);
    x = s0.x; xd = s0.y; th = s0.z; thd = s0.w;
    steps = 0;
    nr = 0;
  }
  P.state[i] = x; P.state[N + i] = xd; P.state[2 * N + i] = th; P.state[3 * N + i] = thd;
  P.steps[i] = steps;
  P.need_reset[i] = (uint8_t)nr;
  reinterpret_cast<float4*>(io.obs)[i] = make_float4(x, xd, th, thd);
  io.reward[i] = reward;
  io.terminated[i] = (uint8_t)term;
  io.truncated[i] = (uint8_t)trunc;
  if (io.final_obs) reinterpret_cast<float4*>(io.final_obs)[i] = fobs;
  if (err) atomicOr(P.err, err);
}

template <bool INJECT>
__global__ __launch_bounds__(256) void cartpole_reset_kernel(CartPoleArgs P, const uint8_t* mask, const float* u,
                                                             float* obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  if (mask && !mask[i]) return;
  const size_t N = (size_t)P.n_env;
  const float4 s0 = cartpole_reset_state<INJECT>(P, u, i);
  P.state[i] = s0.x; P.state[N + i] = s0.y; P.state[2 * N + i] = s0.z; P.state[3 * N + i] = s0.w;
  P.steps[i] = 0;
  P.need_reset[i] = 0;
  if (obs) reinterpret_cast<float4*>(obs)[i] = s0;
}

extern "C" int xv_cartpole_create(xv_engine* e, int n_env, int n_task, int frameskip, int max_steps,
                                  const float* params, const float* reset_scale, const int32_t* env_task,
                                  xv_cartpole** out) {
  XV_CHECK_ARG(out != nullptr);
  *out = nullptr;
  XV_CHECK_ARG(e && params && reset_scale && env_task && n_env > 0 && n_task > 0 && frameskip >= 1);
  XV_HIP(hipSetDevice(e->device));
  xv_cartpole* h = new (std::nothrow) xv_cartpole();
  if (!h) {
    xv_set_error("xv_cartpole_create: out of host memory");
    return XV_ERR_NOMEM;
  }
  h->eng = e;
  CartPoleArgs& a = h->a;
  a.params = (const float4*)params; a.env_task = env_task;
  a.n_env = n_env; a.n_task = n_task; a.frameskip = frameskip; a.max_steps = max_steps;
  a.err = e->d_err;
  a.seed = e->seed; a.gid_base = e->env_id_base; a.tick = 0;
  float sc[4];
  XV_HIP(hipMemcpyAsync(sc, reset_scale, sizeof(sc), hipMemcpyDeviceToHost, e->stream));
  XV_HIP(hipStreamSynchronize(e->stream));
  a.reset_scale = make_float4(sc[0], sc[1], sc[2], sc[3]);
  a.state = nullptr; a.steps = nullptr; a.need_reset = nullptr;
  hipError_t m = hipMalloc(&a.state, sizeof(float) * 4 * (size_t)n_env);
  if (m == hipSuccess) m = hipMalloc(&a.steps, sizeof(int32_t) * (size_t)n_env);
  if (m == hipSuccess) m = hipMalloc(&a.need_reset, (size_t)n_env);
  if (m == hipSuccess) m = hipMemsetAsync(a.state, 0, sizeof(float) * 4 * (size_t)n_env, e->stream);
  if (m == hipSuccess) m = hipMemsetAsync(a.steps, 0, sizeof(int32_t) * (size_t)n_env, e->stream);
  if (m == hipSuccess) m = hipMemsetAsync(a.need_reset, 1, (size_t)n_env, e->stream);
  if (m != hipSuccess) {
    xv_set_error("xv_cartpole_create: device allocation failed: %s", hipGetErrorString(m));
    if (a.state) (void)hipFree(a.state);
    if (a.steps) (void)hipFree(a.steps);
    if (a.need_reset) (void)hipFree(a.need_reset);
    delete h;
    return XV_ERR_HIP;
  }
  *out = h;
  return XV_OK;
}

extern "C" int xv_cartpole_destroy(xv_cartpole* h) {
  if (!h) return XV_OK;
  (void)hipSetDevice(h->eng->device);
  (void)hipStreamSynchronize(h->eng->stream);
  (void)hipFree(h->a.state);
  (void)hipFree(h->a.steps);
  (void)hipFree(h->a.need_reset);
  delete h;
  return XV_OK;
}

static inline void cartpole_bind_rng(xv_cartpole* h, uint64_t ticks) {
  h->a.seed = h->eng->seed;
  h->a.gid_base = h->eng->env_id_base;
  h->a.tick = h->eng->tick;
  h->eng->tick += ticks;
}

extern "C" int xv_cartpole_reset(xv_cartpole* h, const uint8_t* mask, float* obs) {
  XV_CHECK_ARG(h != nullptr);
  cartpole_bind_rng(h, 1);
  hipLaunchKernelGGL(cartpole_reset_kernel<false>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                     h->a, mask, (const float*)nullptr, obs);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_cartpole_reset_injected(xv_cartpole* h, const uint8_t* mask, const float* u, float* obs) {
  XV_CHECK_ARG(h != nullptr && u != nullptr);
  cartpole_bind_rng(h, 0);
  hipLaunchKernelGGL(cartpole_reset_kernel<true>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                     h->a, mask, u, obs);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_cartpole_step(xv_cartpole* h, const int32_t* action, float* obs, float* reward,
                                uint8_t* terminated, uint8_t* truncated, float* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && action && obs && reward && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  cartpole_bind_rng(h, 1);
  CartPoleIO io{action, nullptr, obs, reward, terminated, truncated, final_obs};
  hipLaunchKernelGGL(cartpole_step_kernel<false>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                     h->a, io, autoreset_mode);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_cartpole_step_injected(xv_cartpole* h, const int32_t* action, const float* u_reset, float* obs,
                                         float* reward, uint8_t* terminated, uint8_t* truncated,
                                         float* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && action && u_reset && obs && reward && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  cartpole_bind_rng(h, 0);
  CartPoleIO io{action, u_reset, obs, reward, terminated, truncated, final_obs};
  hipLaunchKernelGGL(cartpole_step_kernel<true>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                     h->a, io, autoreset_mode);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_cartpole_get_state(xv_cartpole* h, float* state, int32_t* steps, uint8_t* need_reset) {
  XV_CHECK_ARG(h != nullptr);
  const size_t n = (size_t)h->a.n_env;
  if (state) XV_HIP(hipMemcpyAsync(state, h->a.state, n * 16, hipMemcpyDeviceToDevice, h->eng->stream));
  if (steps) XV_HIP(hipMemcpyAsync(steps, h->a.steps, n * 4, hipMemcpyDeviceToDevice, h->eng->stream));
  if (need_reset) XV_HIP(hipMemcpyAsync(need_reset, h->a.need_reset, n, hipMemcpyDeviceToDevice, h->eng->stream));
  return XV_OK;
}

extern "C" int xv_cartpole_set_state(xv_cartpole* h, const float* state, const int32_t* steps,
                                     const uint8_t* need_reset) {
  XV_CHECK_ARG(h != nullptr);
  const size_t n = (size_t)h->a.n_env;
  if (state) XV_HIP(hipMemcpyAsync(h->a.state, state, n * 16, hipMemcpyDeviceToDevice, h->eng->stream));
  if (steps) XV_HIP(hipMemcpyAsync(h->a.steps, steps, n * 4, hipMemcpyDeviceToDevice, h->eng->stream));
  if (need_reset) XV_HIP(hipMemcpyAsync(h->a.need_reset, need_reset, n, hipMemcpyDeviceToDevice, h->eng->stream));
  return XV_OK;
}
