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
// The wait is bounded: it gives up — the caller sets XV_DEVERR_HANDOFF, and the call is then replayed on one stream (anymdp.hip:
// anymdp_replay_kernel, mixed.hip: mixed_replay_kernel): time, not results, never a hang — once it has
// polled XV_HAND_MIN_POLLS times AND XV_HAND_TIMEOUT of the 100-MHz wall clock have passed.  Both, because the clock alone is not the
// wave's own time: the device's scheduler can take a process's queues off the hardware for tens of milliseconds when
// another process touches the GPU; a waiter that comes back finds "its" 50 ms gone although the step before it was simply
// suspended too (measured with a 50-ms clock bound: 11-15 spurious expiries per 2,700 overlapped soak calls, every family
// of a step at the same microsecond, 10-25 s apart — profiles/r05_v_*).  A suspended wave does not poll, so polls count
// the wave's own waiting.  Round 6: since an expiry is repaired by the replay, the bound is what a stuck call may cost, not
// what protects the results: 2^19 polls AND 0.5 s (round 5: 2^20 and 2 s), and once ANY wave of the call has given up —
// XV_DEVERR_HANDOFF in the engine's error word, which the call's opening kernel cleared — every other wait of that call ends
// at its next look at the word (every 64 polls: xv_hand_aborted): the failed attempt drains in microseconds instead of one
// bound per step (640 steps x 2 s when two of the call's streams end up on one hardware queue in the wrong order), then the
// replay runs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define XV_HAND_TIMEOUT 50000000ull    // 0.5 s of the 100-MHz wall clock ...
#define XV_HAND_MIN_POLLS (1u << 19)   // ... and this many polls (~0.35 s of polling)
#define XV_HAND_ABORT_EVERY 64u        // polls between two looks at the call's error word
#ifndef XV_HAND_POLL_SLEEP
#define XV_HAND_POLL_SLEEP 1           // s_sleep units (64 clocks) between two polls of a waiting wave
#endif

// has a wait that began at t_begin and has polled `polls` times run out?  (the clock is read only beyond the poll count)
__device__ __forceinline__ bool xv_hand_expired(uint32_t polls, uint64_t t_begin) {
#ifdef XV_HAND_CLOCK_EVERY_POLL      // devtools A/B: the instruction sequence of the clock-only bound (s_memrealtime per poll)
  const bool late = wall_clock64() - t_begin > XV_HAND_TIMEOUT;
  return late && polls > XV_HAND_MIN_POLLS;
#else
  return polls > XV_HAND_MIN_POLLS && wall_clock64() - t_begin > XV_HAND_TIMEOUT;
#endif
}

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

// has another wave of this call given up already?  (err: the engine's error word, cleared of the bit at the call's entry;
// nullptr: never — paths whose expired hand-offs are flagged only)
__device__ __forceinline__ bool xv_hand_aborted(uint32_t polls, const uint32_t* err) {
  return err != nullptr && (polls & (XV_HAND_ABORT_EVERY - 1u)) == XV_HAND_ABORT_EVERY - 1u && (xv_agent_load32(err) & 8u /* XV_DEVERR_HANDOFF */) != 0u;
}

// all lanes of the wave wait until *word == want; false: the bound expired, or another wave of the call had given up
__device__ __forceinline__ bool xv_hand_wait(const uint32_t* word, uint32_t want, const uint32_t* err = nullptr) {
  const uint64_t t_begin = wall_clock64();
  for (uint32_t polls = 0;; ++polls) {
    const uint32_t v = __builtin_amdgcn_readfirstlane(xv_agent_load32(word));
    if (v == want) return true;
    __builtin_amdgcn_s_sleep(XV_HAND_POLL_SLEEP);
    if (xv_hand_expired(polls, t_begin) || xv_hand_aborted(polls, err)) return false;
  }
}

// the wave's earlier stores are complete, then the word is written (one lane)
__device__ __forceinline__ void xv_hand_publish(uint32_t* word, uint32_t tag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) xv_agent_store32(word, tag);
}
