// xv_common.h — internals shared by the translation units of libxeno_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/xeno.h"

struct xv_engine {
  int device;
  hipStream_t stream;
  bool own_stream;
  uint64_t seed;
  uint64_t env_id_base;
  uint64_t tick;        // launch counter: Philox counter word, advanced by every stochastic launch
  uint64_t* d_tick;     // the same counter in device memory (xv_engine_set_device_tick), nullptr until first used
  bool dev_tick;        // kernels read the launch tick from *d_tick: a step leaves nothing host-side in its arguments
  bool tick_batch;      // device tick mode, xv_engine_tick_batch: launches read *d_tick + tick_pending and advance nothing;
  uint64_t tick_pending;   // closing the batch advances the word once by the ticks it consumed
  uint32_t* d_err;      // sticky device error word
  hipEvent_t ev[2];     // xv_engine_event_*: created on first use
  bool ev_made;
};

// thread-local error text (xv_last_error)
void xv_set_error(const char* fmt, ...);

#define XV_CHECK_ARG(cond)                                         \
  do {                                                             \
    if (!(cond)) {                                                 \
      xv_set_error("%s: invalid argument: %s", __func__, #cond);   \
      return XV_ERR_INVALID;                                       \
    }                                                              \
  } while (0)

#define XV_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t _e = (call);                                                            \
    if (_e != hipSuccess) {                                                            \
      xv_set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(_e));       \
      return XV_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

#define XV_LAUNCH_CHECK()                                                              \
  do {                                                                                 \
    hipError_t _e = hipGetLastError();                                                 \
    if (_e != hipSuccess) {                                                            \
      xv_set_error("%s: kernel launch failed: %s", __func__, hipGetErrorString(_e));   \
      return XV_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

static inline int xv_div_up(int a, int b) { return (a + b - 1) / b; }

// ---- launch tick of a stochastic launch (engine.hip) ----
// Host tick (default): the launch gets tick = e->tick as a kernel argument.  Device tick (xv_engine_set_device_tick: what
// makes a step capturable in a hipGraph / torch.cuda.graph): a one-thread kernel advances *d_tick by `ticks` FIRST, and
// the launch reads *d_tick + (0 - ticks) — so that every entry point stays "bind, then launch" and a replayed graph
// draws fresh numbers at every replay.  Kernels evaluate xv_launch_tick(P.tick, P.tick_dev).
bool xv_device_overlap_acquire(int device, const void* owner);   // engine.hip: the device's one overlapped-step_many slot
void xv_device_overlap_release(int device, const void* owner);
void xv_device_note_collective(int device, int delta);   // engine.hip: live RCCL communicators of this process per device
int xv_device_collectives(int device);
void xv_engine_advance_device_tick(xv_engine* e, uint64_t ticks);
void xv_engine_advance_device_tick3(xv_engine* e0, xv_engine* e1, xv_engine* e2, uint64_t d0, uint64_t d1,
                                    uint64_t d2);   // one launch (mixed batch): per-word increments
struct XvTickBind {
  uint64_t tick;
  const uint64_t* tick_dev;
};
static inline XvTickBind xv_engine_bind_tick(xv_engine* e, uint64_t ticks, bool advance = true) {
  XvTickBind b;
  if (e->dev_tick) {
    b.tick_dev = e->d_tick;
    if (e->tick_batch) {      // an unrolled capture: step j of the batch reads *d_tick + j, ONE advance closes the batch
      b.tick = e->tick_pending;
      e->tick_pending += ticks;
    } else {
      if (ticks && advance) xv_engine_advance_device_tick(e, ticks);
      b.tick = (uint64_t)0 - ticks;
    }
  } else {
    b.tick = e->tick;
    b.tick_dev = nullptr;
  }
  e->tick += ticks;      // host mirror (exact while every launch is issued through the C-ABI; a replayed graph is not)
  return b;
}
#ifdef __HIPCC__
__device__ __forceinline__ uint64_t xv_launch_tick(uint64_t tick, const uint64_t* tick_dev) {
  return tick_dev ? tick + *tick_dev : tick;
}
#endif
