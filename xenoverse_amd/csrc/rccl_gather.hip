// rccl_gather.hip — the one collective of the path: all-gather of finished rollout chunks across the
// GPUs of a node, over RCCL directly (ncclAllGather on the engine's stream; xGMI underneath), for callers that do not
// want a torch.distributed process group.  librccl.so is opened on first use, so libxeno_hip.so loads on hosts without
// it; the communicator is created from a 128-byte unique id that rank 0 makes and the caller distributes (any channel:
// the Python side uses a TCP store).  Stepping itself never communicates.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "xv_common.h"

// The handful of RCCL declarations used here, stated locally (they are ABI-stable across NCCL 2.x / RCCL) so that the
// library builds on hosts without the RCCL headers, as it loads on hosts without librccl.so.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;     // ncclSuccess = 0
typedef int ncclDataType_t;   // ncclUint8 = 1
}
static constexpr ncclResult_t ncclSuccess = 0;
static constexpr ncclDataType_t ncclUint8 = 1;

namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  bool ok = false;
  char why[256] = "symbols missing";   // dlerror() text of the failed dlopen, captured once (reading it clears it)
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
      const char* msg = dlerror();
      if (msg) snprintf(api.why, sizeof(api.why), "%s", msg);
    }
    if (!api.lib) return;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.lib, "ncclAllGather"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.lib, "ncclCommCount"));
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GetErrorString;
  });
  return api;
}

int rccl_missing(const char* fn) {
  xv_set_error("%s: librccl.so could not be used (%s)", fn, rccl().why);
  return XV_ERR_UNSUPPORTED;
}

// communicators made here, with their device: the overlapped step_many paths keep at most two launches in flight on a device
// that also runs RCCL's persistent kernels (xv_pipe.h: xv_device_collectives)
std::mutex g_comm_mu;
struct { void* comm; int device; } g_comms[64];

}  // namespace

#define XV_RCCL(call)                                                                        \
  do {                                                                                       \
    ncclResult_t _r = (call);                                                                \
    if (_r != ncclSuccess) {                                                                 \
      xv_set_error("%s: %s failed: %s", __func__, #call, rccl().GetErrorString(_r));         \
      return XV_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

extern "C" int xv_rccl_unique_id(void* out128) {
  XV_CHECK_ARG(out128 != nullptr);
  if (!rccl().ok) return rccl_missing(__func__);
  static_assert(sizeof(ncclUniqueId) == XV_RCCL_ID_BYTES, "unique id size");
  XV_RCCL(rccl().GetUniqueId(static_cast<ncclUniqueId*>(out128)));
  return XV_OK;
}

extern "C" int xv_rccl_comm_create(xv_engine* e, int world, int rank, const void* id128, void** comm_out) {
  XV_CHECK_ARG(e && id128 && comm_out && world >= 1 && rank >= 0 && rank < world);
  *comm_out = nullptr;
  if (!rccl().ok) return rccl_missing(__func__);
  XV_HIP(hipSetDevice(e->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  XV_RCCL(rccl().CommInitRank(&c, world, id, rank));
  *comm_out = c;
  {
    std::lock_guard<std::mutex> lock(g_comm_mu);
    for (auto& slot : g_comms)
      if (slot.comm == nullptr) { slot.comm = c; slot.device = e->device; xv_device_note_collective(e->device, +1); break; }
  }
  return XV_OK;
}

extern "C" int xv_rccl_comm_destroy(void* comm) {
  if (!comm) return XV_OK;
  if (!rccl().ok) return rccl_missing(__func__);
  {
    std::lock_guard<std::mutex> lock(g_comm_mu);
    for (auto& slot : g_comms)
      if (slot.comm == comm) { slot.comm = nullptr; xv_device_note_collective(slot.device, -1); break; }
  }
  XV_RCCL(rccl().CommDestroy(static_cast<ncclComm_t>(comm)));
  return XV_OK;
}

extern "C" int xv_rccl_comm_count(void* comm, int* count_out) {
  XV_CHECK_ARG(comm != nullptr && count_out != nullptr);
  *count_out = 0;
  if (!rccl().ok || !rccl().CommCount) return rccl_missing(__func__);
  XV_RCCL(rccl().CommCount(static_cast<ncclComm_t>(comm), count_out));
  return XV_OK;
}

extern "C" int xv_rollout_allgather(xv_engine* e, void* rccl_comm, const void* local, void* global,
                                    size_t bytes_per_rank) {
  XV_CHECK_ARG(e && rccl_comm && local && global && bytes_per_rank > 0);
  if (!rccl().ok) return rccl_missing(__func__);
  XV_HIP(hipSetDevice(e->device));
  XV_RCCL(rccl().AllGather(local, global, bytes_per_rank, ncclUint8, static_cast<ncclComm_t>(rccl_comm), e->stream));
  return XV_OK;
}
