// xv_hand.h — hand-off of env state between two launches that run at the same time (overlapped step_many of the mixed batch,
// mixed.hip).  The AnyMDP step keeps its tag inside the env record (anymdp.hip: one 8-byte store hands an env on); the
// LinDS and CartPole states are wider than one store, so their waves hand on through a WORD PER WAVE:
//
//   writer (step k):      state stores (agent scope) ... s_waitcnt vmcnt(0) ... word = tag(k + 1)
//   reader (step k + 1):  table loads in flight ... poll word == tag(k + 1) ... state loads (agent scope)
//
// Agent-scope relaxed atomics compile to sc1 loads / stores: coherent across the XCDs' L2s without cache maintenance (an
// agent-scope FENCE is a `buffer_wbl2` per wave, 39 us per step: profiles/r05_b_*).  Ordering comes from the wave itself:
// the stores have completed (vmcnt 0) before the word is written, and the state loads are issued after the poll returned.
// The wait is bounded by the 100-MHz wall clock; on expiry the wave goes on and the caller sets XV_DEVERR_HANDOFF —
// wrong results, flagged, never a hang.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define XV_HAND_TIMEOUT 5000000ull   // 50 ms of the 100-MHz wall clock

__device__ __forceinline__ uint64_t xv_agent_load64(const void* p) {
  return __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xv_agent_store64(void* p, uint64_t v) {
  __hip_atomic_store(reinterpret_cast<uint64_t*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t xv_agent_load32(const void* p) {
  return __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xv_agent_store32(void* p, uint32_t v) {
  __hip_atomic_store(reinterpret_cast<uint32_t*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double xv_agent_load_f64(const double* p) { return __longlong_as_double((long long)xv_agent_load64(p)); }
__device__ __forceinline__ void xv_agent_store_f64(double* p, double v) { xv_agent_store64(p, (uint64_t)__double_as_longlong(v)); }

// all lanes of the wave wait until *word == want; false: the bound expired
__device__ __forceinline__ bool xv_hand_wait(const uint32_t* word, uint32_t want) {
  const uint64_t t_begin = wall_clock64();
  for (;;) {
    const uint32_t v = __builtin_amdgcn_readfirstlane(xv_agent_load32(word));
    if (v == want) return true;
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t_begin > XV_HAND_TIMEOUT) return false;
  }
}

// the wave's earlier stores are complete, then the word is written (one lane)
__device__ __forceinline__ void xv_hand_publish(uint32_t* word, uint32_t tag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) xv_agent_store32(word, tag);
}
