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
