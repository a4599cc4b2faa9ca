// xv_pipe.h — the host side of the overlapped step_many paths (anymdp.hip, mixed.hip): picking the SIDE STREAMS, the cycle
// gate, how many ring cycles a cycle graph holds, how many steps are in flight, what must fit on the device.
//
// The overlapped paths issue consecutive vector steps alternately on the engine's stream and on a side stream; a wave of
// step k + 1 waits for the same wave of step k.  That pays only if the device really processes the two streams' launches
// side by side.  Whether it does depends on where the runtime puts the side stream: streams are mapped onto a few hardware
// queues, and two queues served by the same command-processor pipe take turns instead — measured on MI355X with an RCCL
// communicator created BEFORE the side stream: the same two-graph schedule ran at 23 us per step instead of 4.5
// (profiles/r05_t_*), while a plain "did the other stream start within 20 ms" probe still passed.
//
// So the side stream is chosen by MEASUREMENT: a ping-pong of n one-thread launches, launch k waiting (bounded) for the
// word launch k - 1 leaves, alternating between the two streams, timed with events against the same n launches on the
// engine's stream alone.  Candidates of several priorities are tried; the first whose ping-pong costs no more than
// XV_PIPE_ACCEPT_RATIO x the one-stream chain (+ XV_PIPE_ACCEPT_SLACK_US) is kept.  If none qualifies the overlapped path
// is not used on that handle (the ordinary one-stream path runs instead: same results).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "xv_common.h"
#include "xv_hand.h"

#define XV_PIPE_PING_STEPS 128
#define XV_PIPE_ACCEPT_RATIO 1.5f
#define XV_PIPE_ACCEPT_SLACK_US 1.0f
#define XV_PIPE_MAX_CANDIDATES 6

static __global__ void xv_pipe_ping_kernel(uint32_t* word, uint32_t want, uint32_t* late) {
  const uint64_t t_begin = wall_clock64();
  while (xv_agent_load32(word) != want) {
    if (wall_clock64() - t_begin > 200000ull) {      // 2 ms: the launch before this one has not run
      atomicOr(late, 1u);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  xv_agent_store32(word, want + 1u);
}

// n chained one-thread launches, alternating main / side (side == nullptr: all on main) -> us per launch, < 0 on failure
// or when a wait expired.  d_words: two uint32 of device memory.
static float xv_pipe_pingpong(hipStream_t main, hipStream_t side, uint32_t* d_words, hipEvent_t* ev /*[4]*/, int n) {
  if (hipMemsetAsync(d_words, 0, 2 * sizeof(uint32_t), main) != hipSuccess) return -1.0f;
  bool ok = hipEventRecord(ev[0], main) == hipSuccess;
  if (side) ok = ok && hipEventRecord(ev[2], main) == hipSuccess && hipStreamWaitEvent(side, ev[2], 0) == hipSuccess;
  // two streams: the odd launch of a pair is issued FIRST, on the side stream — if the streams share a hardware queue it
  // sits in front of the launch it waits for, its wait expires and the candidate is out (never a hang)
  for (int k = 0; ok && k < n; ++k) {
    const int kk = side ? (k ^ 1) : k;
    hipLaunchKernelGGL(xv_pipe_ping_kernel, dim3(1), dim3(1), 0, (side && (kk & 1)) ? side : main, d_words, (uint32_t)kk, d_words + 1);
    ok = hipGetLastError() == hipSuccess;
  }
  if (side) ok = hipEventRecord(ev[3], side) == hipSuccess && hipStreamWaitEvent(main, ev[3], 0) == hipSuccess && ok;
  ok = hipEventRecord(ev[1], main) == hipSuccess && ok;
  ok = hipStreamSynchronize(main) == hipSuccess && ok;
  if (side) ok = hipStreamSynchronize(side) == hipSuccess && ok;
  uint32_t w[2] = {0u, 1u};
  float ms = 0.0f;
  ok = ok && hipMemcpy(w, d_words, sizeof(w), hipMemcpyDeviceToHost) == hipSuccess && hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess;
  if (!ok) { (void)hipGetLastError(); return -1.0f; }
  if (w[0] != (uint32_t)n || w[1] != 0u) return -1.0f;
  return ms * 1000.0f / (float)n;
}

struct XvPipeCandidate {
  int priority;        // what the stream was created with
  float two_us;        // ping-pong over main + this stream, us per launch (< 0: a wait expired / failure)
  float one_us;        // the same chain on main alone
  int accepted;
};

// Tries up to XV_PIPE_MAX_CANDIDATES side streams; *out = the first accepted one (the others are destroyed), nullptr when
// none qualifies.  report (nullable): one row per candidate tried; *n_report rows filled.
static bool xv_pipe_pick_side_stream(hipStream_t main, hipStream_t* out, XvPipeCandidate* report, int* n_report,
                                     hipStream_t other = nullptr /* a side stream already chosen: the new one must run beside it too */) {
  *out = nullptr;
  if (n_report) *n_report = 0;
  uint32_t* d = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  if (hipMalloc(&d, 2 * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return false; }
  bool ok = true;
  for (int i = 0; i < 4; ++i) ok = ok && hipEventCreate(&ev[i]) == hipSuccess;
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  // a stream of another priority class gets a hardware queue of its own: the highest first, then ordinary streams (each
  // new one moves on to the runtime's next queue), the lowest last
  // (which class comes first makes no difference where both qualify: 3.75-3.78 against 3.73-3.76 us, profiles/r05_x_*)
  const int prio[XV_PIPE_MAX_CANDIDATES] = {greatest, 0, 0, least, 0, 0};
  float one = -1.0f;
  if (ok) {
    (void)xv_pipe_pingpong(main, nullptr, d, ev, 16);      // warm-up: code object load, first launch
    one = xv_pipe_pingpong(main, nullptr, d, ev, XV_PIPE_PING_STEPS);
    // the better of two rounds, here and below: the process can be off the hardware for tens of milliseconds in the middle
    // of a round (xv_hand.h); one inflated reference would wave every candidate through
    const float one_again = xv_pipe_pingpong(main, nullptr, d, ev, XV_PIPE_PING_STEPS);
    if (one_again > 0.0f && (one <= 0.0f || one_again < one)) one = one_again;
    ok = one > 0.0f;
  }
  hipStream_t rejected[XV_PIPE_MAX_CANDIDATES];
  int n_rej = 0;
  for (int c = 0; ok && c < XV_PIPE_MAX_CANDIDATES && !*out; ++c) {
    hipStream_t s = nullptr;
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio[c]) != hipSuccess) { (void)hipGetLastError(); continue; }
    (void)xv_pipe_pingpong(main, s, d, ev, 16);
    float two = xv_pipe_pingpong(main, s, d, ev, XV_PIPE_PING_STEPS);
    {      // the better of two rounds (a first use of a new queue can be slow once; a suspended process expires a wait)
      const float again = xv_pipe_pingpong(main, s, d, ev, XV_PIPE_PING_STEPS);
      if (again > 0.0f && (two <= 0.0f || again < two)) two = again;
    }
    bool take = two > 0.0f && two <= XV_PIPE_ACCEPT_RATIO * one + XV_PIPE_ACCEPT_SLACK_US;
    if (take && other) {      // three streams: the two side streams must not share a hardware queue with each other either
      float pair = xv_pipe_pingpong(other, s, d, ev, XV_PIPE_PING_STEPS);
      const float again = xv_pipe_pingpong(other, s, d, ev, XV_PIPE_PING_STEPS);
      if (again > 0.0f && (pair <= 0.0f || again < pair)) pair = again;
      take = pair > 0.0f && pair <= XV_PIPE_ACCEPT_RATIO * one + XV_PIPE_ACCEPT_SLACK_US;
      if (pair > two || pair <= 0.0f) two = pair;      // report the worse pairing
    }
    if (getenv("XV_PIPE_DEBUG"))
      fprintf(stderr, "xv_pipe: candidate %d priority %d: two streams %.2f us, one stream %.2f us per launch -> %s\n", c, prio[c],
              (double)two, (double)one, take ? "taken" : "rejected");
    if (report && n_report) {
      report[*n_report] = XvPipeCandidate{prio[c], two, one, take ? 1 : 0};
      *n_report += 1;
    }
    if (take) *out = s;
    else rejected[n_rej++] = s;      // kept alive until the choice is made: destroying one would hand its queue to the next
  }
  for (int i = 0; i < n_rej; ++i) (void)hipStreamDestroy(rejected[i]);
  for (int i = 0; i < 4; ++i) if (ev[i]) (void)hipEventDestroy(ev[i]);
  (void)hipFree(d);
  (void)hipGetLastError();
  return *out != nullptr;
}

// ------------------------------------------------------------------------------------------------------------------
// The cycle gate.  The host issues a ring cycle as two graph launches, the even half on the engine's stream and then the odd
// half on the side stream.  The even half's second step waits (bounded, xv_hand.h) for the odd half's first step: were the
// host held up between the two launches for longer than that — a page fault, a descheduled thread, the runtime blocking on
// a full queue — the wait would expire and the results be wrong (flagged, XV_DEVERR_HANDOFF).  So the even half starts
// with a one-thread GATE node: gate number g (counted in device memory) passes once the host has published
// `issued >= g`, which it does, in pinned host memory, after BOTH launches of that cycle have been enqueued.  A held-up
// host then just delays the device.  The gate's own wait is bounded too (polls and 10 s, as xv_hand.h; then flagged).
// EVERY graph of a set starts with such a gate (its own counter d_seen[q], the same `issued`): with three streams a host
// held up between the second and the third launch would otherwise leave the second graph's waves resident and spinning
// (2-s bound) on a first graph that is still gated (10-s bound).  No wave of a set becomes resident before the whole set
// is enqueued.
#define XV_PIPE_GATE_TIMEOUT 1000000000ull   // 10 s of the 100-MHz wall clock ...
#define XV_PIPE_GATE_MIN_POLLS (1u << 20)     // ... and this many reads of the host's word (~2 us each): the wave's own waiting

struct XvPipeGate {
  uint32_t* h_issued;   // pinned, mapped host words: [0] graph sets enqueued, [1] overlapped calls replayed after an expired hand-off
  uint32_t* d_issued;   // the device's view of them
  uint32_t* d_seen;     // [XV_PIPE_DEPTH_MAX] gates passed so far by the graphs of each stream (device memory)
  uint32_t issued;      // host mirror
  int unroll[2];        // ring cycles per cycle graph of the graph pairs built on this gate (0: none built); see below
  // streams 2 and 3 of the overlapped MDP step_many (depth 3 / 4: steps k .. k + 3 in flight), see xv_pipe_depth()
  hipStream_t side_n[2];
  hipStream_t side_n_for;   // the engine's stream they were chosen against
  hipEvent_t ev_n[2];
  hipGraph_t graph_n[2];
  hipGraphExec_t exec_n[2];
  int depth;            // streams the graphs of unroll[0] were built for (2 .. 4)
};

__device__ __forceinline__ void xv_pipe_gate_pass(uint32_t* seen, const uint32_t* issued, uint32_t* err) {
  const uint32_t g = *seen + 1u;
  *seen = g;
  const uint64_t t_begin = wall_clock64();
  uint32_t polls = 0;
  while ((int32_t)(__hip_atomic_load(issued, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - g) < 0) {
    if (++polls > XV_PIPE_GATE_MIN_POLLS && wall_clock64() - t_begin > XV_PIPE_GATE_TIMEOUT) {
      atomicOr(err, 8u /* XV_DEVERR_HANDOFF */);
      break;
    }
    if ((polls & 63u) == 63u && (xv_agent_load32(err) & 8u)) break;      // the call has failed elsewhere: it will be replayed
    __builtin_amdgcn_s_sleep(16);
  }
}

static bool xv_pipe_gate_create(XvPipeGate* g) {
  g->h_issued = nullptr; g->d_issued = nullptr; g->d_seen = nullptr; g->issued = 0; g->unroll[0] = g->unroll[1] = 0;
  for (int i = 0; i < 2; ++i) { g->side_n[i] = nullptr; g->ev_n[i] = nullptr; g->graph_n[i] = nullptr; g->exec_n[i] = nullptr; }
  g->side_n_for = nullptr; g->depth = 0;
  if (hipHostMalloc(reinterpret_cast<void**>(&g->h_issued), 64, hipHostMallocMapped) != hipSuccess) { g->h_issued = nullptr; return false; }
  for (int i = 0; i < 16; ++i) g->h_issued[i] = 0u;
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(&g->d_issued), g->h_issued, 0) != hipSuccess) return false;
  if (hipMalloc(&g->d_seen, 4 /* XV_PIPE_DEPTH_MAX */ * sizeof(uint32_t)) != hipSuccess) { g->d_seen = nullptr; return false; }
  return hipMemset(g->d_seen, 0, 4 * sizeof(uint32_t)) == hipSuccess;
}
// when a graph set is (re)built — every stream of the handle drained: all gate counters start from what the host has issued
// (a stream that sat out some sets, e.g. the third one while two steps were in flight, would otherwise pass its next gates early)
static bool xv_pipe_gate_sync(XvPipeGate* g) {
  const uint32_t v[4] = {g->issued, g->issued, g->issued, g->issued};
  return hipMemcpy(g->d_seen, v, sizeof(v), hipMemcpyHostToDevice) == hipSuccess;
}
static void xv_pipe_gate_destroy(XvPipeGate* g) {
  if (g->d_seen) (void)hipFree(g->d_seen);
  if (g->h_issued) (void)hipHostFree(g->h_issued);
  g->h_issued = nullptr; g->d_issued = nullptr; g->d_seen = nullptr; g->issued = 0;
}
// after both launches of a cycle are enqueued (or to let a half-issued cycle's gate go on the error path)
static inline void xv_pipe_gate_release(XvPipeGate* g) {
  g->issued += 1u;
  __atomic_store_n(g->h_issued, g->issued, __ATOMIC_RELEASE);
}

// test hook (tests/test_gpu_chains.py): XV_PIPE_TEST_STALL_MS=<ms> holds the host up between two launches of the first
// graph set of every overlapped call — longer than the hand-off bound, the cycle gates must make that harmless.
// XV_PIPE_TEST_STALL_AT = 0 (default): between the first and the second graph; 1: between the second and the third.
static inline void xv_pipe_test_stall(int cycle, int at = 0) {
  if (cycle != 0) return;
  const char* v = getenv("XV_PIPE_TEST_STALL_MS");
  const char* w = getenv("XV_PIPE_TEST_STALL_AT");
  if ((w ? atoi(w) : 0) != at) return;
  if (v && atoi(v) > 0) std::this_thread::sleep_for(std::chrono::milliseconds(atoi(v)));
}

// Two launches of `fn` must be able to be RESIDENT at once.  A workgroup of step k + 1 that holds a slot spins until the
// same workgroup of step k has run; if step k still has workgroups waiting for a slot while step k + 1's spinners hold them
// all (the side stream may be served first), nobody moves until the bounded waits expire.  With 2 x grid <= what the device
// holds of this kernel that cannot happen: every workgroup of step k gets its slot without waiting for one of step k + 1.
// The occupancy figure is what an EMPTY device holds.  Another resident kernel of the process (a policy network on another
// stream, RCCL's all-gather kernels) takes slots, so a quarter of the device is left out of the sum: n x grid <= 3/4 of the
// slots (XV_PIPE_RESIDENCY_NUM / _DEN).  A neighbour that is finite only delays the hand-offs — its workgroups end and the
// launches in flight (at most depth x grid of them are ever dispatchable: launch k + depth sits behind launch k on its stream)
// get their slots; a wait that expires all the same is repaired by the replay (anymdp.hip, mixed.hip), never left in the
// results.  The margin applies to three or more launches in flight; two may fill the device.  While the process holds an RCCL communicator on the device (xv_device_note_collective: persistent kernels that spin
// on peers) at most two launches are in flight.
#define XV_PIPE_RESIDENCY_NUM 3
#define XV_PIPE_RESIDENCY_DEN 4
static bool xv_pipe_two_launches_fit(const void* fn, int block_threads, size_t grid_blocks, int device, int n_launches = 2) {
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block_threads, 0) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  if (n_launches > 2 && xv_device_collectives(device) > 0) return false;
  // three (or more) in flight: within 3/4 of the slots; two: the whole device, as in round 5 (131,072 AnyMDP envs: 5.6 instead
  // of 7.0 us per step) — a neighbour can then starve a step for a while, which costs a replay and a back-off, not results
  if (n_launches <= 2) return (size_t)n_launches * grid_blocks <= (size_t)per_cu * (size_t)cus;
  return (size_t)n_launches * grid_blocks * XV_PIPE_RESIDENCY_DEN <= (size_t)per_cu * (size_t)cus * XV_PIPE_RESIDENCY_NUM;
}

// Back-off after a replayed call.  An expired hand-off costs its bound plus the replay — correct, but a hundred times a call's
// normal time — and what made it expire (a neighbour's kernels, two of the call's streams sharing a hardware queue for a
// while) tends to last.  The host sees the replay counter (pinned memory, no synchronisation) at the entry of a later call:
// the next `len` calls take the one-stream path, and `len` doubles with every further replay (32 .. 4,096 calls; switching
// the overlap on again starts over).
struct XvPipeBackoff {
  uint32_t fell_known;   // replays the host has accounted for
  int left;              // calls still to issue on one stream
  int len;               // length of the next back-off
};
static inline void xv_pipe_backoff_reset(XvPipeBackoff* b, const XvPipeGate* g) {
  b->fell_known = g->h_issued ? __atomic_load_n(g->h_issued + 1, __ATOMIC_ACQUIRE) : 0u;
  b->left = 0; b->len = 32;
}
// -> true: this call takes the one-stream path
static inline bool xv_pipe_backoff_step(XvPipeBackoff* b, const XvPipeGate* g) {
  if (getenv("XV_PIPE_NO_BACKOFF")) return false;      // tests: every call of a failing series is overlapped and replayed
  if (g->h_issued) {
    const uint32_t fell = __atomic_load_n(g->h_issued + 1, __ATOMIC_ACQUIRE);
    if (fell != b->fell_known) {
      b->fell_known = fell;
      if (b->len < 32) b->len = 32;
      b->left = b->len;
      b->len = b->len >= 2048 ? 4096 : 2 * b->len;
    }
  }
  if (b->left > 0) { --b->left; return true; }
  return false;
}

// Ring cycles per cycle graph.  Every cycle graph starts with a head node (tick word; on the even stream the cycle gate: a
// dispatch and a read of host memory on that stream's chain), so the more steps a graph holds the less the head costs per
// step — measured at 65,536 envs, period 32: 16 steps per graph 3.94 us, 32: 3.79, 64: 3.65-3.77, 128: 3.70-3.72 (2b: 3.51 /
// 3.30 / 3.23 / 3.20; a token step on a ring of 8: 6.15 us with 4 steps per graph, 4.95 with 16) — profiles/r05_x_*.
// A call is served by whole graphs; the ring cycles that do not fill one go out on one stream.  So the unroll U is chosen
// per call by a small cost model — steps in graphs cost 1 + 1.15 / (steps per graph and stream), left-over steps 1.3 (the
// one-stream step against the overlapped one) — over U = 1 .. XV_PIPE_GRAPH_STEPS_BIG / period, and it is sticky: the graphs
// a handle holds are kept while they are within 3 % of the best choice for the call at hand (a rebuild costs milliseconds;
// callers repeat one call length).  -> ring cycles per graph, 0: do not overlap this call.  have: what the handle's graphs
// for these rings hold now (0: none).
#define XV_PIPE_GRAPH_STEPS_BIG 128
static inline double xv_pipe_unroll_cost(int period, int cycles, int U, int depth) {
  const int covered = U * (cycles / U) * period, left = cycles * period - covered;
  if (covered <= 0 || (U * period) % depth != 0) return 1.0e30;      // a graph set takes whole steps-per-stream
  return (double)covered * (1.0 + 1.15 / ((double)U * (double)period / (double)depth)) + 1.3 * (double)left;
}
static inline int xv_pipe_pick_unroll(int period, int cycles, int have, int depth = 2) {
  static const int big_steps = getenv("XV_PIPE_GRAPH_STEPS") ? atoi(getenv("XV_PIPE_GRAPH_STEPS")) : XV_PIPE_GRAPH_STEPS_BIG;   // devtools A/B
  const int big = (period >= big_steps ? 1 : big_steps / period) * (depth == 2 ? 1 : depth);
  int best = 0;
  double best_cost = 1.0e30;
  for (int U = 1; U <= big && U <= cycles; ++U) {
    const double c = xv_pipe_unroll_cost(period, cycles, U, depth);
    if (c < 1.0e29 && c <= best_cost) { best = U; best_cost = c; }      // ties: the larger graph
  }
  if (best == 0) return 0;
  if (have > 0 && have <= cycles && xv_pipe_unroll_cost(period, cycles, have, depth) <= 1.03 * best_cost) return have;
  return best;
}
// Steps in flight (= streams).  With two streams, launch k + 2 waits behind launch k on its stream: every wave of k must have
// drained before k + 2 is dispatched, and the 2.7 us a dispatch costs are covered by one other launch only.  Three streams —
// steps k, k + 1, k + 2 resident together, every wave still depending on the same wave of the step before only — measured
// at 65,536 envs: 2a 3.64-3.85 -> **3.22-3.36 us**, 2b 3.21-3.25 -> **2.95-3.07**; four: 3.10-3.25 / 2.95-3.08, no better, and
// four launches of 256 workgroups are all the device holds of this kernel (profiles/r05_y_*).  So: three where three
// launches fit on the device together, else two, else none.  XV_PIPE_DEPTH = 2 .. 4 caps it (devtools A/B).
#define XV_PIPE_DEPTH_MAX 4
#define XV_PIPE_DEPTH_DEFAULT 3
static inline int xv_pipe_depth() {
  static const int d = getenv("XV_PIPE_DEPTH") ? atoi(getenv("XV_PIPE_DEPTH")) : XV_PIPE_DEPTH_DEFAULT;
  return d < 2 ? 2 : (d > XV_PIPE_DEPTH_MAX ? XV_PIPE_DEPTH_MAX : d);
}
// -> streams to use for launches of `fn` over grid_blocks workgroups: the wanted depth if that many launches can be resident
// together (see xv_pipe_two_launches_fit), else fewer, 0: not even two
static int xv_pipe_choose_depth(const void* fn, int block_threads, size_t grid_blocks, int device) {
  for (int d = xv_pipe_depth(); d >= 2; --d)
    if (xv_pipe_two_launches_fit(fn, block_threads, grid_blocks, device, d)) return d;
  return 0;
}
