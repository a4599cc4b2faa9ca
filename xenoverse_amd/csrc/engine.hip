// engine.hip — engine handle (device + stream + Philox key + sticky device error word).
#include "philox.h"
#include <mutex>

#include "xv_common.h"
#include "xv_pipe.h"

static thread_local char g_err[512] = "";

void xv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int xv_abi_version(void) { return XV_ABI_VERSION; }
extern "C" const char* xv_last_error(void) { return g_err; }

extern "C" int xv_engine_create(int device, uint64_t seed, uint64_t env_id_base, void* hip_stream,
                                xv_engine** out) {
  XV_CHECK_ARG(out != nullptr);
  *out = nullptr;
  int n_dev = 0;
  XV_HIP(hipGetDeviceCount(&n_dev));
  XV_CHECK_ARG(device >= 0 && device < n_dev);
  XV_HIP(hipSetDevice(device));
  xv_engine* e = new (std::nothrow) xv_engine();
  if (!e) {
    xv_set_error("xv_engine_create: out of host memory");
    return XV_ERR_NOMEM;
  }
  e->device = device;
  e->seed = seed;
  e->env_id_base = env_id_base;
  e->tick = 0;
  e->own_stream = (hip_stream == XV_STREAM_OWN);
  e->stream = e->own_stream ? nullptr : (hipStream_t)hip_stream;
  e->d_err = nullptr;
  e->d_tick = nullptr;
  e->dev_tick = false;
  e->tick_batch = false;
  e->tick_pending = 0;
  e->ev[0] = e->ev[1] = nullptr;
  e->ev_made = false;
  if (e->own_stream) {
    hipError_t s = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (s != hipSuccess) {
      xv_set_error("xv_engine_create: hipStreamCreate failed: %s", hipGetErrorString(s));
      delete e;
      return XV_ERR_HIP;
    }
  }
  hipError_t m = hipMalloc(&e->d_err, sizeof(uint32_t));
  if (m == hipSuccess) m = hipMemsetAsync(e->d_err, 0, sizeof(uint32_t), e->stream);
  if (m != hipSuccess) {
    xv_set_error("xv_engine_create: error word allocation failed: %s", hipGetErrorString(m));
    if (e->own_stream) hipStreamDestroy(e->stream);
    delete e;
    return XV_ERR_HIP;
  }
  *out = e;
  return XV_OK;
}

extern "C" int xv_engine_destroy(xv_engine* e) {
  if (!e) return XV_OK;
  hipSetDevice(e->device);
  hipStreamSynchronize(e->stream);
  if (e->d_err) hipFree(e->d_err);
  if (e->d_tick) hipFree(e->d_tick);
  if (e->ev_made) { hipEventDestroy(e->ev[0]); hipEventDestroy(e->ev[1]); }
  if (e->own_stream) hipStreamDestroy(e->stream);
  delete e;
  return XV_OK;
}

extern "C" int xv_engine_sync(xv_engine* e) {
  XV_CHECK_ARG(e != nullptr);
  XV_HIP(hipStreamSynchronize(e->stream));
  return XV_OK;
}

extern "C" void* xv_engine_stream(xv_engine* e) { return e ? (void*)e->stream : nullptr; }

extern "C" int xv_engine_error_flags(xv_engine* e, int clear, uint32_t* out_flags) {
  XV_CHECK_ARG(e != nullptr && out_flags != nullptr);
  uint32_t v = 0;
  XV_HIP(hipMemcpyAsync(&v, e->d_err, sizeof(v), hipMemcpyDeviceToHost, e->stream));
  XV_HIP(hipStreamSynchronize(e->stream));
  if (clear) XV_HIP(hipMemsetAsync(e->d_err, 0, sizeof(uint32_t), e->stream));
  *out_flags = v;
  return XV_OK;
}

static __global__ void xv_tick_set_kernel(uint64_t* t, uint64_t v) { *t = v; }
static __global__ void xv_tick_add_kernel(uint64_t* t, uint64_t dv) { *t += dv; }

// (no launch check of its own: the family's launch follows at once and its XV_LAUNCH_CHECK reads the same sticky error)
void xv_engine_advance_device_tick(xv_engine* e, uint64_t ticks) {
  hipLaunchKernelGGL(xv_tick_add_kernel, dim3(1), dim3(1), 0, e->stream, e->d_tick, ticks);
}

// d0, d1, d2: what each word advances by (a word shared by several handles is advanced once, by the ticks all of them consume)
static __global__ void xv_tick_add3_kernel(uint64_t* t0, uint64_t* t1, uint64_t* t2, uint64_t d0, uint64_t d1, uint64_t d2) {
  *t0 += d0;
  if (t1 != t0) *t1 += d1;
  if (t2 != t0 && t2 != t1) *t2 += d2;
}
// ONE handle per device may issue overlapped step_many calls at a time (two overlapped calls in flight can block each other on the
// hardware queues their streams share): the slot, whatever the family of the handle that holds it
static std::mutex g_overlap_slot_mu;
static const void* g_overlap_slot[64] = {nullptr};
bool xv_device_overlap_acquire(int device, const void* owner) {
  if (device < 0 || device >= 64) return false;
  std::lock_guard<std::mutex> lock(g_overlap_slot_mu);
  if (g_overlap_slot[device] != nullptr && g_overlap_slot[device] != owner) return false;
  g_overlap_slot[device] = owner;
  return true;
}
void xv_device_overlap_release(int device, const void* owner) {
  if (device < 0 || device >= 64) return;
  std::lock_guard<std::mutex> lock(g_overlap_slot_mu);
  if (g_overlap_slot[device] == owner) g_overlap_slot[device] = nullptr;
}

// live RCCL communicators of this process per device (rccl_gather.hip counts them; xv_pipe.h reads the figure)
static int g_device_collectives[64] = {0};
void xv_device_note_collective(int device, int delta) {
  if (device >= 0 && device < 64) __atomic_fetch_add(&g_device_collectives[device], delta, __ATOMIC_RELAXED);
}
int xv_device_collectives(int device) {
  return (device >= 0 && device < 64) ? __atomic_load_n(&g_device_collectives[device], __ATOMIC_RELAXED) : 0;
}

void xv_engine_advance_device_tick3(xv_engine* e0, xv_engine* e1, xv_engine* e2, uint64_t d0, uint64_t d1, uint64_t d2) {
  hipLaunchKernelGGL(xv_tick_add3_kernel, dim3(1), dim3(1), 0, e0->stream, e0->d_tick, e1->d_tick, e2->d_tick, d0, d1, d2);
}

// a tick batch is open between (e, 1) and (e, 0): launches inside it read word + pending, so reading or overwriting the
// word there would be off by the pending ticks
static int xv_engine_no_open_batch(const xv_engine* e, const char* fn) {
  if (e->dev_tick && e->tick_batch) {
    xv_set_error("%s: not while a tick batch is open (xv_engine_tick_batch(e, 0) first)", fn);
    return XV_ERR_UNSUPPORTED;
  }
  return XV_OK;
}

extern "C" int xv_engine_get_tick(xv_engine* e, uint64_t* out_tick) {
  XV_CHECK_ARG(e != nullptr && out_tick != nullptr);
  if (const int rc = xv_engine_no_open_batch(e, __func__)) return rc;
  if (e->dev_tick) {   // the device word is the truth (graph replays advance it without the host); synchronises
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(e->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
      xv_set_error("xv_engine_get_tick: the engine's stream is capturing; the device tick cannot be read back now");
      return XV_ERR_UNSUPPORTED;
    }
    XV_HIP(hipMemcpyAsync(&e->tick, e->d_tick, sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
    XV_HIP(hipStreamSynchronize(e->stream));
  }
  *out_tick = e->tick;
  return XV_OK;
}
extern "C" int xv_engine_set_tick(xv_engine* e, uint64_t tick) {
  XV_CHECK_ARG(e != nullptr);
  if (const int rc = xv_engine_no_open_batch(e, __func__)) return rc;
  e->tick = tick;
  if (e->dev_tick) {
    hipLaunchKernelGGL(xv_tick_set_kernel, dim3(1), dim3(1), 0, e->stream, e->d_tick, tick);
    XV_LAUNCH_CHECK();
  }
  return XV_OK;
}

extern "C" int xv_engine_set_device_tick(xv_engine* e, int on) {
  XV_CHECK_ARG(e != nullptr && (on == 0 || on == 1));
  if ((on != 0) == e->dev_tick) return XV_OK;
  XV_HIP(hipSetDevice(e->device));
  if (on) {
    if (!e->d_tick) XV_HIP(hipMalloc(&e->d_tick, sizeof(uint64_t)));
    hipLaunchKernelGGL(xv_tick_set_kernel, dim3(1), dim3(1), 0, e->stream, e->d_tick, e->tick);
    XV_LAUNCH_CHECK();
    e->dev_tick = true;
  } else {
    if (e->tick_batch && e->tick_pending) xv_engine_advance_device_tick(e, e->tick_pending);
    e->tick_batch = false;
    e->tick_pending = 0;
    XV_HIP(hipMemcpyAsync(&e->tick, e->d_tick, sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
    XV_HIP(hipStreamSynchronize(e->stream));
    e->dev_tick = false;
  }
  return XV_OK;
}
extern "C" int xv_engine_device_tick(xv_engine* e) { return e ? (e->dev_tick ? 1 : 0) : XV_ERR_INVALID; }

extern "C" int xv_engine_tick_batch(xv_engine* e, int on) {
  XV_CHECK_ARG(e != nullptr && (on == 0 || on == 1));
  if (!e->dev_tick) {
    xv_set_error("xv_engine_tick_batch: needs the device tick (xv_engine_set_device_tick)");
    return XV_ERR_UNSUPPORTED;
  }
  if (on) {
    XV_CHECK_ARG(!e->tick_batch);
    e->tick_batch = true;
    e->tick_pending = 0;
  } else if (e->tick_batch) {
    e->tick_batch = false;
    if (e->tick_pending) {
      xv_engine_advance_device_tick(e, e->tick_pending);
      XV_LAUNCH_CHECK();
    }
    e->tick_pending = 0;
  }
  return XV_OK;
}

extern "C" int xv_engine_set_stream(xv_engine* e, void* hip_stream) {
  XV_CHECK_ARG(e != nullptr);
  if (e->own_stream) {
    xv_set_error("xv_engine_set_stream: this engine owns its stream");
    return XV_ERR_UNSUPPORTED;
  }
  if (hip_stream != nullptr) {      // (the null stream belongs to whatever device is current: nothing to check)
    hipDevice_t dev = -1;
    if (hipStreamGetDevice((hipStream_t)hip_stream, &dev) == hipSuccess && (int)dev != e->device) {
      xv_set_error("xv_engine_set_stream: the stream belongs to device %d, the engine to device %d", (int)dev, e->device);
      return XV_ERR_INVALID;
    }
    (void)hipGetLastError();
  }
  e->stream = (hipStream_t)hip_stream;
  return XV_OK;
}

extern "C" int xv_engine_event_record(xv_engine* e, int slot) {
  XV_CHECK_ARG(e != nullptr && (slot == 0 || slot == 1));
  if (!e->ev_made) {
    XV_HIP(hipSetDevice(e->device));
    XV_HIP(hipEventCreate(&e->ev[0]));
    XV_HIP(hipEventCreate(&e->ev[1]));
    e->ev_made = true;
  }
  XV_HIP(hipEventRecord(e->ev[slot], e->stream));
  return XV_OK;
}

extern "C" int xv_engine_event_done(xv_engine* e, int slot, int* done) {
  XV_CHECK_ARG(e != nullptr && (slot == 0 || slot == 1) && done != nullptr && e->ev_made);
  const hipError_t q = hipEventQuery(e->ev[slot]);
  *done = q == hipSuccess;
  if (q != hipSuccess && q != hipErrorNotReady) XV_HIP(q);
  return XV_OK;
}

extern "C" int xv_engine_event_elapsed_ms(xv_engine* e, float* ms) {
  XV_CHECK_ARG(e != nullptr && ms != nullptr && e->ev_made);
  XV_HIP(hipEventElapsedTime(ms, e->ev[0], e->ev[1]));
  return XV_OK;
}

// diagnostic: which side streams would the overlapped step_many paths accept beside this engine's stream right now
// (xv_pipe.h)?  One row per candidate tried, in order; nothing is kept.
extern "C" int xv_engine_probe_side_streams(xv_engine* e, int max_rows, int* priority, float* two_stream_us, float* one_stream_us,
                                            int* accepted, int* n_rows) {
  XV_CHECK_ARG(e != nullptr && max_rows > 0 && priority && two_stream_us && one_stream_us && accepted && n_rows);
  XV_HIP(hipSetDevice(e->device));
  XvPipeCandidate rep[XV_PIPE_MAX_CANDIDATES];
  int n = 0;
  hipStream_t s = nullptr;
  (void)xv_pipe_pick_side_stream(e->stream, &s, rep, &n);
  if (s) (void)hipStreamDestroy(s);
  *n_rows = n < max_rows ? n : max_rows;
  for (int i = 0; i < *n_rows; ++i) {
    priority[i] = rep[i].priority; two_stream_us[i] = rep[i].two_us; one_stream_us[i] = rep[i].one_us; accepted[i] = rep[i].accepted;
  }
  return XV_OK;
}

__global__ void xv_philox_kat_kernel(const uint32_t* ctr, const uint32_t* key, uint32_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const xv_u32x4 r = xv_philox4x32_10(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], key[0], key[1]);
  out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

extern "C" int xv_philox4x32_10(xv_engine* e, const uint32_t* ctr, const uint32_t* key, uint32_t* out, int n) {
  XV_CHECK_ARG(e && ctr && key && out && n > 0);
  hipLaunchKernelGGL(xv_philox_kat_kernel, dim3(xv_div_up(n, 256)), dim3(256), 0, e->stream, ctr, key, out, n);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

// ------------------------------------------------------------------------------------------------
// Rollout records for the one exchange step of the path: an AnyMDP env-step as ONE 8-byte word
//   bits 0-15 observation id | 16-23 action | 24 terminated | 25 truncated | 32-63 reward (fp32 bits)
// so that the all-gather of a T-step chunk moves 8 B per env-step (the field-by-field form is 14 B) and the
// pack is one coalesced 8-byte store per record.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xv_pack_rollout_kernel(size_t n, const int32_t* obs, const int32_t* action,
                                                              const float* reward, const uint8_t* terminated,
                                                              const uint8_t* truncated, uint64_t* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t lo = ((uint32_t)obs[i] & 0xFFFFu) | (((uint32_t)action[i] & 0xFFu) << 16) |
                      ((terminated[i] ? 1u : 0u) << 24) | ((truncated[i] ? 1u : 0u) << 25);
  out[i] = (uint64_t)lo | ((uint64_t)__float_as_uint(reward[i]) << 32);
}

__global__ __launch_bounds__(256) void xv_unpack_rollout_kernel(size_t n, const uint64_t* rec, int32_t* obs, int32_t* action,
                                                                float* reward, uint8_t* terminated, uint8_t* truncated) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t r = rec[i];
  const uint32_t lo = (uint32_t)r;
  obs[i] = (int32_t)(lo & 0xFFFFu);
  action[i] = (int32_t)((lo >> 16) & 0xFFu);
  terminated[i] = (uint8_t)((lo >> 24) & 1u);
  truncated[i] = (uint8_t)((lo >> 25) & 1u);
  reward[i] = __uint_as_float((uint32_t)(r >> 32));
}

extern "C" int xv_pack_rollout(void* hip_stream, size_t n, const int32_t* obs, const int32_t* action, const float* reward,
                               const uint8_t* terminated, const uint8_t* truncated, uint64_t* out) {
  XV_CHECK_ARG(obs && action && reward && terminated && truncated && out);
  if (n == 0) return XV_OK;
  XV_CHECK_ARG((n + 255) / 256 < 0x7FFFFFFFull);
  hipLaunchKernelGGL(xv_pack_rollout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, n, obs,
                     action, reward, terminated, truncated, out);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_unpack_rollout(void* hip_stream, size_t n, const uint64_t* rec, int32_t* obs, int32_t* action, float* reward,
                                 uint8_t* terminated, uint8_t* truncated) {
  XV_CHECK_ARG(rec && obs && action && reward && terminated && truncated);
  if (n == 0) return XV_OK;
  XV_CHECK_ARG((n + 255) / 256 < 0x7FFFFFFFull);
  hipLaunchKernelGGL(xv_unpack_rollout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, n,
                     rec, obs, action, reward, terminated, truncated);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

// ------------------------------------------------------------------------------------------------
// Rollout records of the float-observation families (config 5: LinDS, CartPole): one env-step = D + 2 32-bit words
//   words 0..D-1 observation (fp32; LinDS D = 16 padded, linds_env.py:83-91; CartPole D = 4, random_cartpole.py:52-61)
//   word D       reward (fp32 bits)
//   word D + 1   bit 0 terminated | bit 1 truncated | bits 8..31 the discrete action (CartPole; 0 when `action` is null)
// One thread per word: the record stream is written / read fully coalesced, the per-field arrays nearly so.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xv_pack_rollout_f32_kernel(size_t n_words, int D, const float* obs, const float* reward,
                                                                  const uint8_t* terminated, const uint8_t* truncated,
                                                                  const int32_t* action, uint32_t* out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_words) return;
  const size_t r = idx / (size_t)(D + 2);
  const int w = (int)(idx - r * (size_t)(D + 2));
  uint32_t v;
  if (w < D) v = __float_as_uint(obs[r * (size_t)D + w]);
  else if (w == D) v = __float_as_uint(reward[r]);
  else v = (terminated[r] ? 1u : 0u) | (truncated[r] ? 2u : 0u) | (action ? ((uint32_t)action[r] << 8) : 0u);
  out[idx] = v;
}

__global__ __launch_bounds__(256) void xv_unpack_rollout_f32_kernel(size_t n_words, int D, const uint32_t* rec, float* obs,
                                                                    float* reward, uint8_t* terminated, uint8_t* truncated,
                                                                    int32_t* action) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_words) return;
  const size_t r = idx / (size_t)(D + 2);
  const int w = (int)(idx - r * (size_t)(D + 2));
  const uint32_t v = rec[idx];
  if (w < D) obs[r * (size_t)D + w] = __uint_as_float(v);
  else if (w == D) reward[r] = __uint_as_float(v);
  else {
    terminated[r] = (uint8_t)(v & 1u);
    truncated[r] = (uint8_t)((v >> 1) & 1u);
    if (action) action[r] = (int32_t)(v >> 8);
  }
}

extern "C" int xv_pack_rollout_f32(void* hip_stream, size_t n, int obs_dim, const float* obs, const float* reward,
                                   const uint8_t* terminated, const uint8_t* truncated, const int32_t* action, uint32_t* out) {
  XV_CHECK_ARG(obs && reward && terminated && truncated && out && obs_dim >= 1 && obs_dim <= 4096);
  if (n == 0) return XV_OK;
  const size_t nw = n * (size_t)(obs_dim + 2);
  XV_CHECK_ARG((nw + 255) / 256 < 0x7FFFFFFFull);
  hipLaunchKernelGGL(xv_pack_rollout_f32_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, nw,
                     obs_dim, obs, reward, terminated, truncated, action, out);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_unpack_rollout_f32(void* hip_stream, size_t n, int obs_dim, const uint32_t* rec, float* obs, float* reward,
                                     uint8_t* terminated, uint8_t* truncated, int32_t* action) {
  XV_CHECK_ARG(rec && obs && reward && terminated && truncated && obs_dim >= 1 && obs_dim <= 4096);
  if (n == 0) return XV_OK;
  const size_t nw = n * (size_t)(obs_dim + 2);
  XV_CHECK_ARG((nw + 255) / 256 < 0x7FFFFFFFull);
  hipLaunchKernelGGL(xv_unpack_rollout_f32_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, nw,
                     obs_dim, rec, obs, reward, terminated, truncated, action);
  XV_LAUNCH_CHECK();
  return XV_OK;
}
