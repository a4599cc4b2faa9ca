// cartpole.hip — domain-randomised CartPole batched step / reset for gfx950 and its C-ABI.
//
// Reproduces xenoverse/metacontrol/random_cartpole.py (set_task :46-50, step :52-61 = `frameskip` repeats of
// gymnasium's CartPoleEnv.step, reset :63-75).  The physics equations are gymnasium's (third-party, not
// vendored, not installed here: restated from the public 1.x source — parity unpinned).
// One lane per env, 4 fp64 state words in component-major arrays (coalesced), 32 B of task parameters: gymnasium keeps
// `self.state` and the whole Euler update in float64 and casts only the returned observation to float32 — so does
// this kernel (an fp32 state would drift from the reference within tens of steps and move the step at which the
// |x| > 2.4 / |theta| > 12 deg termination fires).
// Same operation order as oracle/xeno_oracle.c: cartpole_step_one.
#include "philox.h"
#include "xv_common.h"
#include "xv_hand.h"

struct CartPoleArgs {
  const double* params;      // [n_task][4]: gravity, masscart, masspole, length
  double reset_scale[4];
  const int32_t* env_task;
  double* state;             // [4][n_env]
  int32_t* steps;
  uint8_t* need_reset;
  uint32_t* err;
  int n_env, n_task, frameskip, max_steps;
  uint64_t seed, gid_base, tick;
  const uint64_t* tick_dev;   // device tick mode of the engine: the launch tick is *tick_dev + tick (xv_launch_tick)
  uint32_t* hand;             // [waves] HAND kernels only (mixed.hip): hand-off words of the waves' envs, xv_hand.h
};

struct CartPoleIO {
  const int32_t* action;
  const double* u_reset;     // [4][n_env] (INJECT)
  float* obs;                // [n_env][4]
  float* reward;
  uint8_t* terminated;
  uint8_t* truncated;
  float* final_obs;          // nullable
  uint8_t* done_out;         // nullable: terminated | truncated of the same step (xv_cartpole_step_info)
};

struct xv_cartpole {
  xv_engine* eng;
  CartPoleArgs a;
};

struct CpState { double x, xd, th, thd; };

template <bool INJECT>
__device__ __forceinline__ CpState cartpole_reset_state(const CartPoleArgs& P, const double* u_in, int i, uint64_t tick) {
  double u0, u1, u2, u3;
  if (INJECT) {
    u0 = u_in[i]; u1 = u_in[(size_t)P.n_env + i]; u2 = u_in[(size_t)2 * P.n_env + i]; u3 = u_in[(size_t)3 * P.n_env + i];
  } else {
    const xv_u32x4 w = xv_env_draw(P.seed, P.gid_base + (uint64_t)i, tick, XV_DRAW_RESET);
    u0 = (double)w.x * (1.0 / 4294967296.0); u1 = (double)w.y * (1.0 / 4294967296.0);
    u2 = (double)w.z * (1.0 / 4294967296.0); u3 = (double)w.w * (1.0 / 4294967296.0);
  }
  // state = uniform(-1, 1, 4) * reset_bounds_scale   (random_cartpole.py:70)
  return CpState{fma(2.0, u0, -1.0) * P.reset_scale[0], fma(2.0, u1, -1.0) * P.reset_scale[1],
                 fma(2.0, u2, -1.0) * P.reset_scale[2], fma(2.0, u3, -1.0) * P.reset_scale[3]};
}

template <bool INJECT, bool HAND = false>
__device__ __forceinline__ void cartpole_step_body(const CartPoleArgs& P, const CartPoleIO& io, int mode, int T, int bid) {
  // T steps per launch (xv_cartpole_rollout; T = 1 for xv_cartpole_step): the state stays in registers, step ts reads
  // action[ts][i], writes row ts of the outputs and draws with tick + ts — the same values as T launches of one step
  // (`bid`: the workgroup's index within this family's part of the launch, see mixed.hip)
  const int i = bid * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  const size_t N = (size_t)P.n_env;
  double x, xd, th, thd;
  int steps, nr;
  uint32_t err = 0;
  const double* prm = P.params + (size_t)P.env_task[i] * 4;
  // HAND (overlapped step_many of the mixed batch, mixed.hip; T = 1): the wave waits for its word P.hand[wave] to carry this
  // step's tick — written by the same wave of the step before, after its state stores completed — and hands on likewise
  // (xv_hand.h).  The env's word in P.steps carries need_reset in bit 31 between HAND launches (mixed.hip packs / unpacks).
  if (HAND) {
    if (!xv_hand_wait(P.hand + (i >> 6), (uint32_t)xv_launch_tick(P.tick, P.tick_dev), P.err)) err |= XV_DEVERR_HANDOFF;
    asm volatile("" ::: "memory");
    x = xv_agent_load_f64(P.state + i); xd = xv_agent_load_f64(P.state + N + i);
    th = xv_agent_load_f64(P.state + 2 * N + i); thd = xv_agent_load_f64(P.state + 3 * N + i);
    const uint32_t w = xv_agent_load32(P.steps + i);
    steps = (int)(w & 0x7FFFFFFFu); nr = (int)(w >> 31);
  } else {
    x = P.state[i]; xd = P.state[N + i]; th = P.state[2 * N + i]; thd = P.state[3 * N + i];
    steps = P.steps[i];
    nr = P.need_reset[i];
  }
  for (int ts = 0; ts < T; ++ts) {
  const size_t o = (size_t)ts * N + i;
  int action = io.action[o];
  float reward = 0.0f;
  int term = 0, trunc = 0;
  float4 fobs = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  bool do_reset = false;
  if (mode == XV_AUTORESET_NEXT_STEP && nr) {
    do_reset = true;
  } else {
    if (action != 0 && action != 1) {
      err |= XV_DEVERR_ACTION_RANGE;
      action = action > 0 ? 1 : 0;
    }
    const double gravity = prm[0], masscart = prm[1], masspole = prm[2], length = prm[3];
    const double polemass_length = masspole * length;   // :49
    const double total_mass = masspole + masscart;      // :50
    const double force = action == 1 ? 10.0 : -10.0;
    const double theta_threshold = 12 * 2 * 3.141592653589793 / 360, x_threshold = 2.4, tau = 0.02;
    for (int f = 0; f < P.frameskip; ++f) {            // :56-60
      double sn, cs;
      sincos(th, &sn, &cs);
      const double temp = (force + polemass_length * (thd * thd) * sn) / total_mass;
      const double thacc = (gravity * sn - cs * temp) / (length * (4.0 / 3.0 - masspole * (cs * cs) / total_mass));
      const double xacc = temp - polemass_length * thacc * cs / total_mass;
      x = x + tau * xd;
      xd = xd + tau * xacc;
      th = th + tau * thd;
      thd = thd + tau * thacc;
      term = ((x < -x_threshold) || (x > x_threshold) || (th < -theta_threshold) || (th > theta_threshold)) ? 1 : 0;
      reward += 1.0f;
      if (term) break;
    }
    steps += 1;
    trunc = (P.max_steps > 0 && steps >= P.max_steps) ? 1 : 0;
    if (!(fabs(x) <= 1.0e300) || !(fabs(thd) <= 1.0e300)) err |= XV_DEVERR_NONFINITE;
    if (term || trunc) {
      if (mode == XV_AUTORESET_SAME_STEP) {
        fobs = make_float4((float)x, (float)xd, (float)th, (float)thd);
        do_reset = true;
      } else if (mode == XV_AUTORESET_NEXT_STEP) {
        nr = 1;
      }
    }
  }
  if (do_reset) {
    const CpState s0 = cartpole_reset_state<INJECT>(P, io.u_reset, i, xv_launch_tick(P.tick, P.tick_dev) + (uint64_t)ts);
    x = s0.x; xd = s0.xd; th = s0.th; thd = s0.thd;
    steps = 0;
    nr = 0;
  }
  reinterpret_cast<float4*>(io.obs)[o] = make_float4((float)x, (float)xd, (float)th, (float)thd);   // the float32 cast of :75
  io.reward[o] = reward;
  io.terminated[o] = (uint8_t)term;
  io.truncated[o] = (uint8_t)trunc;
  if (io.done_out) io.done_out[o] = (uint8_t)((term || trunc) ? 1 : 0);
  if (io.final_obs) reinterpret_cast<float4*>(io.final_obs)[o] = fobs;
  }
  if (HAND) {
    xv_agent_store_f64(P.state + i, x); xv_agent_store_f64(P.state + N + i, xd);
    xv_agent_store_f64(P.state + 2 * N + i, th); xv_agent_store_f64(P.state + 3 * N + i, thd);
    xv_agent_store32(P.steps + i, (uint32_t)steps | ((uint32_t)nr << 31));
    xv_hand_publish(P.hand + (i >> 6), (uint32_t)xv_launch_tick(P.tick, P.tick_dev) + 1u);
  } else {
    P.state[i] = x; P.state[N + i] = xd; P.state[2 * N + i] = th; P.state[3 * N + i] = thd;
    P.steps[i] = steps;
    P.need_reset[i] = (uint8_t)nr;
  }
  if (err) atomicOr(P.err, err);
}

template <bool INJECT>
__global__ __launch_bounds__(256) void cartpole_step_kernel(CartPoleArgs P, CartPoleIO io, int mode, int T) {
  cartpole_step_body<INJECT>(P, io, mode, T, (int)blockIdx.x);
}

template <bool INJECT>
__global__ __launch_bounds__(256) void cartpole_reset_kernel(CartPoleArgs P, const uint8_t* mask, const double* u,
                                                             float* obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  if (mask && !mask[i]) return;
  const size_t N = (size_t)P.n_env;
  const CpState s0 = cartpole_reset_state<INJECT>(P, u, i, xv_launch_tick(P.tick, P.tick_dev));
  P.state[i] = s0.x; P.state[N + i] = s0.xd; P.state[2 * N + i] = s0.th; P.state[3 * N + i] = s0.thd;
  P.steps[i] = 0;
  P.need_reset[i] = 0;
  if (obs) reinterpret_cast<float4*>(obs)[i] = make_float4((float)s0.x, (float)s0.xd, (float)s0.th, (float)s0.thd);
}

static inline void cartpole_bind_rng(xv_cartpole* h, uint64_t ticks, bool advance = true) {
  h->a.seed = h->eng->seed;
  h->a.gid_base = h->eng->env_id_base;
  const XvTickBind b = xv_engine_bind_tick(h->eng, ticks, advance);
  h->a.tick = b.tick;
  h->a.tick_dev = b.tick_dev;
}

#ifndef XV_KERNELS_ONLY   // mixed.hip includes this file for its kernels and handle types only
extern "C" int xv_cartpole_create(xv_engine* e, int n_env, int n_task, int frameskip, int max_steps,
                                  const double* params, const double* reset_scale, const int32_t* env_task,
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
  a.params = params; a.env_task = env_task;
  a.n_env = n_env; a.n_task = n_task; a.frameskip = frameskip; a.max_steps = max_steps;
  a.err = e->d_err;
  a.seed = e->seed; a.gid_base = e->env_id_base; a.tick = 0; a.tick_dev = nullptr;
  {
    hipError_t r = hipMemcpyAsync(a.reset_scale, reset_scale, sizeof(a.reset_scale), hipMemcpyDeviceToHost, e->stream);
    if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
    if (r != hipSuccess) {
      xv_set_error("xv_cartpole_create: reading reset_scale failed: %s", hipGetErrorString(r));
      delete h;
      return XV_ERR_HIP;
    }
  }
  a.state = nullptr; a.steps = nullptr; a.need_reset = nullptr;
  hipError_t m = hipMalloc(&a.state, sizeof(double) * 4 * (size_t)n_env);
  if (m == hipSuccess) m = hipMalloc(&a.steps, sizeof(int32_t) * (size_t)n_env);
  if (m == hipSuccess) m = hipMalloc(&a.need_reset, (size_t)n_env);
  if (m == hipSuccess) m = hipMemsetAsync(a.state, 0, sizeof(double) * 4 * (size_t)n_env, e->stream);
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


extern "C" int xv_cartpole_reset(xv_cartpole* h, const uint8_t* mask, float* obs) {
  XV_CHECK_ARG(h != nullptr);
  cartpole_bind_rng(h, 1);
  hipLaunchKernelGGL(cartpole_reset_kernel<false>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                     h->a, mask, (const double*)nullptr, obs);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_cartpole_reset_injected(xv_cartpole* h, const uint8_t* mask, const double* u, float* obs) {
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
                     h->a, io, autoreset_mode, 1);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

// xv_cartpole_step that also writes the terminated | truncated mask from the same launch (done uint8[n_env], nullable)
extern "C" int xv_cartpole_step_info(xv_cartpole* h, const int32_t* action, float* obs, float* reward, uint8_t* terminated,
                                     uint8_t* truncated, float* final_obs, uint8_t* done, int autoreset_mode) {
  XV_CHECK_ARG(h && action && obs && reward && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  cartpole_bind_rng(h, 1);
  CartPoleIO io{action, nullptr, obs, reward, terminated, truncated, final_obs, done};
  hipLaunchKernelGGL(cartpole_step_kernel<false>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                     h->a, io, autoreset_mode, 1);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_cartpole_rollout(xv_cartpole* h, int T, const int32_t* action, float* obs, float* reward,
                                   uint8_t* terminated, uint8_t* truncated, float* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && action && obs && reward && terminated && truncated && T > 0);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  cartpole_bind_rng(h, (uint64_t)T);
  CartPoleIO io{action, nullptr, obs, reward, terminated, truncated, final_obs};
  hipLaunchKernelGGL(cartpole_step_kernel<false>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                     h->a, io, autoreset_mode, T);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_cartpole_step_injected(xv_cartpole* h, const int32_t* action, const double* u_reset, float* obs,
                                         float* reward, uint8_t* terminated, uint8_t* truncated,
                                         float* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && action && u_reset && obs && reward && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  cartpole_bind_rng(h, 0);
  CartPoleIO io{action, u_reset, obs, reward, terminated, truncated, final_obs};
  hipLaunchKernelGGL(cartpole_step_kernel<true>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                     h->a, io, autoreset_mode, 1);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_cartpole_get_state(xv_cartpole* h, double* state, int32_t* steps, uint8_t* need_reset) {
  XV_CHECK_ARG(h != nullptr);
  const size_t n = (size_t)h->a.n_env;
  if (state) XV_HIP(hipMemcpyAsync(state, h->a.state, n * 32, hipMemcpyDeviceToDevice, h->eng->stream));
  if (steps) XV_HIP(hipMemcpyAsync(steps, h->a.steps, n * 4, hipMemcpyDeviceToDevice, h->eng->stream));
  if (need_reset) XV_HIP(hipMemcpyAsync(need_reset, h->a.need_reset, n, hipMemcpyDeviceToDevice, h->eng->stream));
  return XV_OK;
}

extern "C" int xv_cartpole_set_state(xv_cartpole* h, const double* state, const int32_t* steps,
                                     const uint8_t* need_reset) {
  XV_CHECK_ARG(h != nullptr);
  const size_t n = (size_t)h->a.n_env;
  if (state) XV_HIP(hipMemcpyAsync(h->a.state, state, n * 32, hipMemcpyDeviceToDevice, h->eng->stream));
  if (steps) XV_HIP(hipMemcpyAsync(h->a.steps, steps, n * 4, hipMemcpyDeviceToDevice, h->eng->stream));
  if (need_reset) XV_HIP(hipMemcpyAsync(h->a.need_reset, need_reset, n, hipMemcpyDeviceToDevice, h->eng->stream));
  return XV_OK;
}
#endif   // XV_KERNELS_ONLY
