// mixed.hip — one launch per vector step for a MIXED task batch (BASELINE.json config 5: per GPU 16,384 anymdp + 8,192
// linds + 8,192 cartpole envs).
//
// The families' step kernels are 4-6 us each at those sizes — about the cost of launching anything — so stepping them
// one after the other costs three launch latencies per vector step (14-17 us measured), and separate streams cost more
// in cross-stream waits than the overlap returns (HISTORY.md 5.1).  Here the three families share ONE grid: a range of
// workgroups runs the AnyMDP step body, a range the LinDS matrix body, the rest the CartPole body (the order of the ranges:
// see the kernel) — the very functions the families' own kernels wrap (anymdp_step_body / linds_step_mfma_body / cartpole_step_body, called with the
// workgroup's index inside its family), so every env gets bit for bit what xv_anymdp_step / xv_linds_step /
// xv_cartpole_step would give it.  Each family keeps its handle, its engine tick and its Philox stream; the handles must
// share one HIP stream.  The reference has no counterpart: it steps one env object per Python call.
#define XV_KERNELS_ONLY
#include "anymdp.hip"
#include "cartpole.hip"
#include "linds.hip"
#include "xv_pipe.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>

// HAND: the overlapped xv_mixed_step_many — the launch may start while the step before it is still running; every wave
// waits for its own envs / tile to be handed on (AnyMDP: the tag in the env record, anymdp.hip; LinDS and CartPole: a word
// per wave, xv_hand.h).
template <int AG, int ABK, int LNS, int LNO, bool HAND = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
void mixed_step_kernel(AnyMDPArgs A, AnyMDPStepIO aio, int nbA, LinDSArgs L, LinDSStepIO lio, int nbL, CartPoleArgs C, CartPoleIO cio,
                       int mode) {
  // Which family gets the first workgroups?  On one stream the step lasts as long as its longest dependent chain, LinDS's
  // (state -> command rows -> MFMAs -> stores): started first it ends 0.3 us sooner (5.42-5.54 -> 5.04-5.26 us,
  // profiles/r05_x_*).  Overlapped (HAND), AnyMDP first is the better order (4.57-4.66 against 4.73-4.77).
  const int b = (int)blockIdx.x;
  if (HAND) {
    if (b < nbA) {
      anymdp_step_body<false, AG, false, false, ABK, HAND>(A, aio, 1, mode, b);
    } else if (b < nbA + nbL) {
      linds_step_mfma_body<LNS, 8, LNO, false, HAND>(L, lio, mode, b - nbA);
    } else {
      cartpole_step_body<false, HAND>(C, cio, mode, 1, b - nbA - nbL);
    }
  } else {
    if (b < nbL) {
      linds_step_mfma_body<LNS, 8, LNO, false, HAND>(L, lio, mode, b);
    } else if (b < nbL + nbA) {
      anymdp_step_body<false, AG, false, false, ABK, HAND>(A, aio, 1, mode, b - nbL);
    } else {
      cartpole_step_body<false, HAND>(C, cio, mode, 1, b - nbA - nbL);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// An expired hand-off of an overlapped xv_mixed_step_many is REPAIRED, as for the AnyMDP step_many (anymdp.hip:
// anymdp_replay_kernel).  mixed_pipe_snap_kernel keeps the three families' states and error words as they are at the call's
// entry; behind the join (and the close kernel) mixed_replay_kernel runs on the engines' stream: no new HANDOFF bit in any of
// the three words — every workgroup returns at once (one nearly empty launch per call); else every wave restores its envs /
// its tile from the snapshot and replays the call's steps in one launch (AnyMDP and CartPole: the fused roll-out loops of
// their step bodies, ring cycle by ring cycle; LinDS: linds_replay_body), same ticks, every ring slot rewritten, and the last
// workgroup publishes error words = entry | what the replay raised and counts the replay in pinned host memory.
struct MixedSnap {
  const uint2* a_sr;
  const float* l_x;
  const int32_t* l_sn;
  const double* c_state;
  const int32_t* c_steps;
  const uint8_t* c_nr;
  uint32_t* w;            // [0..2] error words at entry (anymdp, linds, cartpole), [3..5] the replay's own bits, [6] workgroups finished
  uint32_t* err[3];       // the engines' words
  uint32_t* h_fell;       // pinned host memory: calls replayed
};
struct MixedSnapCopy {
  void* dst;
  const void* src;
  size_t bytes;
};
struct MixedSnapList {
  MixedSnapCopy e[6];
  uint32_t* w;
  uint32_t* err[3];
};
static __global__ __launch_bounds__(256) void mixed_pipe_snap_kernel(MixedSnapList L) {
  const MixedSnapCopy c = L.e[blockIdx.y];
  const size_t n16 = c.bytes / 16, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = i0; i < n16; i += stride) reinterpret_cast<uint4*>(c.dst)[i] = reinterpret_cast<const uint4*>(c.src)[i];
  for (size_t b = n16 * 16 + i0; b < c.bytes; b += stride) reinterpret_cast<uint8_t*>(c.dst)[b] = reinterpret_cast<const uint8_t*>(c.src)[b];
  // the error words as they are at entry; the HANDOFF bit leaves them for the time of the call (xv_hand_aborted) and returns
  // with the replay kernel
  if (blockIdx.y == 0 && i0 < 7) {
    uint32_t e = 0u;
    if (i0 < 3) { e = *L.err[i0]; *L.err[i0] = e & ~(uint32_t)XV_DEVERR_HANDOFF; }
    L.w[i0] = e;
  }
}
template <int AG, int ABK, int LNS, int LNO>
__global__ __launch_bounds__(256) void mixed_replay_kernel(AnyMDPArgs A, AnyMDPStepIO aio, int nbA, LinDSArgs L, LinDSStepIO lio, int nbL,
                                                           CartPoleArgs C, CartPoleIO cio, int period, int cycles, int mode, MixedSnap S) {
  uint32_t fresh = 0;      // (the snapshot kernel took the bit out of the words: set = raised by this call)
#pragma unroll
  for (int f = 0; f < 3; ++f) fresh |= __hip_atomic_load(S.err[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((fresh & XV_DEVERR_HANDOFF) == 0u) {
    if (blockIdx.x == 0 && threadIdx.x < 3 && (S.w[threadIdx.x] & XV_DEVERR_HANDOFF)) atomicOr(S.err[threadIdx.x], (uint32_t)XV_DEVERR_HANDOFF);
    return;
  }
  const int b = (int)blockIdx.x;
  if (b < nbL) {
    LinDSArgs Q = L;
    Q.err = S.w + 4; Q.tick_dev = nullptr;
    linds_replay_body<LNS, 8, LNO>(Q, lio, period, cycles * period, mode, S.l_x, S.l_sn, b);
  } else if (b < nbL + nbA) {
    const int bid = b - nbL, i = bid * (int)blockDim.x + (int)threadIdx.x;
    if (i < A.n_env) A.sr[i] = S.a_sr[i];
    __syncthreads();      // (an invalid lane looks at the last env's record: restored by a lane of this workgroup)
    AnyMDPArgs Q = A;
    Q.err = S.w + 3; Q.tick_dev = nullptr;
    for (int c = 0; c < cycles; ++c) {
      Q.tick = A.tick + (uint64_t)c * (uint64_t)period;
      anymdp_step_body<false, AG, true, false, ABK, false>(Q, aio, period, mode, bid);
    }
  } else {
    const int bid = b - nbA - nbL, i = bid * (int)blockDim.x + (int)threadIdx.x;
    if (i < C.n_env) {
      const size_t N = (size_t)C.n_env;
#pragma unroll
      for (int q = 0; q < 4; ++q) C.state[q * N + i] = S.c_state[q * N + i];
      C.steps[i] = S.c_steps[i];
      C.need_reset[i] = S.c_nr[i];
    }
    CartPoleArgs Q = C;
    Q.err = S.w + 5; Q.tick_dev = nullptr;
    for (int c = 0; c < cycles; ++c) {
      Q.tick = C.tick + (uint64_t)c * (uint64_t)period;
      cartpole_step_body<false, false>(Q, cio, mode, period, bid);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(S.w + 6, 1u) == gridDim.x - 1u) {
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        const uint32_t re = __hip_atomic_load(S.w + 3 + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(S.err[f], S.w[f] | re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        S.w[3 + f] = 0u;
      }
      S.w[6] = 0u;
      __hip_atomic_fetch_add(S.h_fell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// test hook (XV_PIPE_TEST_FAIL=1): what an expired hand-off leaves behind — the flag, wrong states, wrong ring contents
static __global__ __launch_bounds__(256) void mixed_test_fail_kernel(uint2* sr, int n_a, float* l_x, size_t n_x, double* c_state, int n_c,
                                                                     uint32_t* err_l, float* l_obs, size_t n_lobs, int32_t* a_obs, size_t n_aobs) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  if (i < (size_t)n_a) sr[i] = make_uint2((uint32_t)(i % 3u), 5u);
  for (size_t k = i; k < n_x; k += stride) l_x[k] = 0.25f;
  if (i < (size_t)n_c) c_state[i] = 0.125;
  for (size_t k = i; k < n_lobs; k += stride) l_obs[k] = -7.0f;
  for (size_t k = i; k < n_aobs; k += stride) a_obs[k] = -7;
  if (i == 0) atomicOr(err_l, (uint32_t)(XV_DEVERR_HANDOFF | XV_DEVERR_NONFINITE));
}

// which instantiation serves these handles (-1: none — the caller falls back to three launches)
static int mixed_variant(const xv_anymdp* a, const xv_linds* l) {
  const int eff = anymdp_effective_search(a);
  if (a->a.G != 1 || eff == XV_ANYMDP_SEARCH_BINARY) return -1;
  if (l->path == XV_LINDS_PATH_SCALAR || l->a.NA != 8 || l->a.NO != 16) return -1;
  // bucket lines in the 7-cut packing only (S <= 112 here, observation ids <= 255); the 6-cut packing takes the fence form
  const int bk = (eff == XV_ANYMDP_SEARCH_BUCKET && a->a.bfmt == 1) ? 1 : 0;
  return bk * 2 + (l->a.NS == 32 ? 1 : 0);
}

extern "C" int xv_mixed_supported(xv_anymdp* a, xv_linds* l, xv_cartpole* c) {
  if (!a || !l || !c) return 0;
  if (a->eng->stream != l->eng->stream || a->eng->stream != c->eng->stream || a->eng->device != l->eng->device ||
      a->eng->device != c->eng->device)
    return 0;
  return mixed_variant(a, l) >= 0 ? 1 : 0;
}

extern "C" int xv_mixed_step(xv_anymdp* a, xv_linds* l, xv_cartpole* c, const xv_mixed_io* io, int autoreset_mode) {
  XV_CHECK_ARG(a && l && c && io);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  XV_CHECK_ARG(io->a_action && io->a_obs && io->a_reward && io->a_reward_gt && io->a_terminated && io->a_truncated);
  XV_CHECK_ARG(io->l_action && io->l_obs && io->l_reward && io->l_terminated && io->l_truncated && io->l_cmd && io->l_error);
  XV_CHECK_ARG(io->c_action && io->c_obs && io->c_reward && io->c_terminated && io->c_truncated);
  if (a->eng->stream != l->eng->stream || a->eng->stream != c->eng->stream || a->eng->device != l->eng->device ||
      a->eng->device != c->eng->device) {
    xv_set_error("xv_mixed_step: the three handles must live on one device and one HIP stream");
    return XV_ERR_INVALID;
  }
  const int v = mixed_variant(a, l);
  if (v < 0) {
    xv_set_error("xv_mixed_step: no fused instantiation for these handles (needs the AnyMDP fence / bucket layout with "
                 "S <= 112, LinDS pads (16|32, 8, 16) on the matrix path): step the families separately");
    return XV_ERR_UNSUPPORTED;
  }
  XV_HIP(hipSetDevice(a->eng->device));
  // the same tick bookkeeping as three separate step calls, in the order anymdp, linds, cartpole
  // (device tick mode: the three engines' tick words are advanced by ONE one-thread launch in front of the step)
  const bool dev3 = a->eng->dev_tick && l->eng->dev_tick && c->eng->dev_tick;
  if (a->eng->dev_tick != l->eng->dev_tick || a->eng->dev_tick != c->eng->dev_tick) {
    xv_set_error("xv_mixed_step: the three engines must agree on the device tick mode (xv_engine_set_device_tick)");
    return XV_ERR_INVALID;
  }
  if (a->eng->tick_batch != l->eng->tick_batch || a->eng->tick_batch != c->eng->tick_batch) {
    xv_set_error("xv_mixed_step: the three engines must open and close their tick batches together (xv_engine_tick_batch)");
    return XV_ERR_INVALID;
  }
  // handles may SHARE an engine: the shared tick word then advances once per handle that uses it, and the handles read
  // consecutive ticks off it (T, T + 1, T + 2 in the order anymdp, linds, cartpole) — what three separate step calls, the
  // host tick and a tick batch all give.  (One advance for a shared word handed the three families the same tick.)
  xv_engine* const E[3] = {a->eng, l->eng, c->eng};
  uint64_t n_tot[3], n_before[3];
  for (int i = 0; i < 3; ++i) {
    n_tot[i] = n_before[i] = 0;
    for (int j = 0; j < 3; ++j) {
      if (E[j] == E[i]) { n_tot[i] += 1; if (j < i) n_before[i] += 1; }
    }
  }
  if (dev3 && !a->eng->tick_batch) xv_engine_advance_device_tick3(a->eng, l->eng, c->eng, n_tot[0], n_tot[1], n_tot[2]);
  anymdp_bind_rng(a, 1, !dev3);
  linds_bind_rng(l, 1, !dev3);
  cartpole_bind_rng(c, 1, !dev3);
  if (dev3 && !a->eng->tick_batch) {      // relative to the advanced word: -n, -n + 1, ... for the handles that share it
    a->a.tick = (uint64_t)0 - n_tot[0] + n_before[0];
    l->a.tick = (uint64_t)0 - n_tot[1] + n_before[1];
    c->a.tick = (uint64_t)0 - n_tot[2] + n_before[2];
  }
  AnyMDPStepIO aio{io->a_action, nullptr, nullptr, nullptr, io->a_obs, io->a_reward, io->a_reward_gt, io->a_terminated,
                   io->a_truncated, io->a_final_obs, nullptr, nullptr, 0.0f, io->a_steps, io->a_done};
  LinDSStepIO lio{io->l_action, nullptr, nullptr, io->l_obs, io->l_reward, io->l_terminated, io->l_truncated, io->l_cmd,
                  io->l_error, io->l_final_obs, io->l_steps, io->l_done};
  CartPoleIO cio{io->c_action, nullptr, io->c_obs, io->c_reward, io->c_terminated, io->c_truncated, io->c_final_obs, io->c_done};
  const int nbA = xv_div_up(a->a.n_env, 256), nbL = xv_div_up(xv_div_up(l->a.n_slot, 16), 4), nbC = xv_div_up(c->a.n_env, 256);
  const dim3 grid(nbA + nbL + nbC), block(256);
  hipStream_t st = a->eng->stream;
#define XV_MIXED_LAUNCH(AG, ABK, LNS) \
  hipLaunchKernelGGL((mixed_step_kernel<AG, ABK, LNS, 16>), grid, block, 0, st, a->a, aio, nbA, l->a, lio, nbL, c->a, cio, autoreset_mode)
  switch (v) {
    case 0: XV_MIXED_LAUNCH(1, 0, 16); break;
    case 1: XV_MIXED_LAUNCH(1, 0, 32); break;
    case 2: XV_MIXED_LAUNCH(1, 1, 16); break;
    default: XV_MIXED_LAUNCH(1, 1, 32); break;
  }
#undef XV_MIXED_LAUNCH
  XV_LAUNCH_CHECK();
  return XV_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// The overlapped xv_mixed_step_many.  A fused step of config 5 is one round of 224 workgroups: 5.5 us of launch, dependent
// loads and drain during which most of the device idles.  As for the AnyMDP step_many (anymdp.hip, "overlap"), the ring's
// even slots go to the engines' stream and the odd slots to a side stream, as two graphs of HAND kernels, so step k + 1
// starts while step k runs and every wave takes its envs over from the same wave of the step before.  Switched on by the
// AnyMDP handle's xv_anymdp_set_step_many_overlap (one overlapped handle per device: two overlapped calls in flight can
// deadlock on the hardware queues).  Needs the three handles on three engines of their own (one tick each per step), host
// ticks, an even period and a call of >= XV_MIXED_PIPE_MIN steps; anything else takes the ordinary loop below.
#define XV_MIXED_PIPE_MIN 64
struct MixedPipeKey {
  AnyMDPArgs A;
  LinDSArgs L;
  CartPoleArgs C;
  xv_mixed_io ring;
  int period, mode, variant, depth;
};
struct MixedPipe {
  hipStream_t side;
  hipStream_t side2;     // third stream (depth 3: steps k, k + 1, k + 2 in flight, xv_pipe.h)
  hipEvent_t ev[2];
  hipEvent_t ev2;
  uint64_t* d_tick;      // [family][stream], 4 words per family: the graphs' tick words
  uint32_t* d_hand;      // LinDS waves, then CartPole waves
  size_t hand_cap;
  hipGraph_t graph[3];
  hipGraphExec_t exec[3];
  MixedPipeKey key;
  bool key_valid, failed, used_last;
  hipStream_t side_for;  // the engines' stream the side stream was chosen against
  XvPipeGate gate;       // the even half of a cycle starts once the host has enqueued both halves (xv_pipe.h)
  const xv_anymdp* used_by;
  // the states at the entry of the last overlapped call (restored by mixed_replay_kernel should a hand-off expire)
  uint8_t* d_snap;
  size_t snap_cap;
  uint32_t* d_snap_w;    // 8 words, see MixedSnap::w
  uint32_t fell_seen;    // gate.h_issued[1] when the last overlapped call was issued
};
// one per device, for the life of the process (side stream, events, tick and hand-off words, the cached graph set: a few
// kilobytes; not released at exit — the HIP runtime may be gone by the time static destructors run)
static MixedPipe g_mixed_pipe[64];
static std::mutex g_mixed_mu;

static __global__ __launch_bounds__(256) void mixed_pipe_open_kernel(uint2* sr, int n_a, uint32_t tag_a, uint32_t* hand_l, int n_lw,
                                                                     uint32_t tag_l, uint32_t* hand_c, int n_cw, uint32_t tag_c,
                                                                     int32_t* c_steps, const uint8_t* c_nr, int n_c,
                                                                     uint64_t* d_tick, uint64_t ta, uint64_t tl, uint64_t tc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_a) sr[i].x = (sr[i].x & ((1u << XV_ANYMDP_SR_TAG_SHIFT) - 1u)) | (tag_a << XV_ANYMDP_SR_TAG_SHIFT);
  if (i < n_lw) hand_l[i] = tag_l;
  if (i < n_cw) hand_c[i] = tag_c;
  if (i < n_c) c_steps[i] = (int32_t)(((uint32_t)c_steps[i] & 0x7FFFFFFFu) | ((uint32_t)(c_nr[i] ? 1u : 0u) << 31));
  if (i < 12) d_tick[i] = i < 4 ? ta : (i < 8 ? tl : tc);
}
// after the join: CartPole's need_reset leaves the step words again
static __global__ __launch_bounds__(256) void mixed_pipe_close_kernel(int32_t* c_steps, uint8_t* c_nr, int n_c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_c) {
    const uint32_t w = (uint32_t)c_steps[i];
    c_steps[i] = (int32_t)(w & 0x7FFFFFFFu);
    c_nr[i] = (uint8_t)(w >> 31);
  }
}
// first node of a cycle graph: this cycle's tick bases (one word per family); the even half also passes the cycle gate
static __global__ void mixed_pipe_head_kernel(uint64_t* w, uint64_t dv, uint32_t* seen, const uint32_t* issued, uint32_t* err) {
  w[0] += dv; w[4] += dv; w[8] += dv;
  if (seen) xv_pipe_gate_pass(seen, issued, err);
}

static void mixed_pipe_drop_graphs(MixedPipe& M) {
  M.gate.unroll[0] = 0;
  for (int q = 0; q < 3; ++q) {
    if (M.exec[q]) (void)hipGraphExecDestroy(M.exec[q]);
    if (M.graph[q]) (void)hipGraphDestroy(M.graph[q]);
    M.exec[q] = nullptr; M.graph[q] = nullptr;
  }
  M.key_valid = false;
}

static bool mixed_pipe_setup(MixedPipe& M, hipStream_t st, size_t n_hand, int depth) {
  if (M.side && M.side_for != st) {      // measured against another engines' stream: choose again
    (void)hipStreamSynchronize(M.side);
    if (M.side2) (void)hipStreamSynchronize(M.side2);
    mixed_pipe_drop_graphs(M);
    (void)hipStreamDestroy(M.side);
    if (M.side2) (void)hipStreamDestroy(M.side2);
    M.side = nullptr; M.side2 = nullptr;
  }
  if (!M.side) {      // chosen by measurement (xv_pipe.h)
    M.side_for = st;
    if (!xv_pipe_pick_side_stream(st, &M.side, nullptr, nullptr)) { M.side = nullptr; return false; }
    if (!M.ev[0] && (hipEventCreateWithFlags(&M.ev[0], hipEventDisableTiming) != hipSuccess ||
                     hipEventCreateWithFlags(&M.ev[1], hipEventDisableTiming) != hipSuccess))
      return false;
  }
  if (depth > 2 && !M.side2) {
    if (!xv_pipe_pick_side_stream(st, &M.side2, nullptr, nullptr, M.side)) { M.side2 = nullptr; return false; }
    if (!M.ev2 && hipEventCreateWithFlags(&M.ev2, hipEventDisableTiming) != hipSuccess) return false;
  }
  if (!M.d_tick && hipMalloc(&M.d_tick, 12 * sizeof(uint64_t)) != hipSuccess) return false;
  if (!M.gate.d_seen && !xv_pipe_gate_create(&M.gate)) return false;
  if (!M.d_snap_w) {
    if (hipMalloc(&M.d_snap_w, 8 * sizeof(uint32_t)) != hipSuccess) { M.d_snap_w = nullptr; return false; }
    if (hipMemsetAsync(M.d_snap_w, 0, 8 * sizeof(uint32_t), st) != hipSuccess) return false;
  }
  if (M.hand_cap < n_hand) {
    (void)hipStreamSynchronize(M.side);
    if (M.side2) (void)hipStreamSynchronize(M.side2);
    (void)hipStreamSynchronize(st);
    if (M.d_hand) (void)hipFree(M.d_hand);
    M.d_hand = nullptr; M.hand_cap = 0; M.key_valid = false;
    if (hipMalloc(&M.d_hand, n_hand * sizeof(uint32_t)) != hipSuccess) return false;
    M.hand_cap = n_hand;
  }
  return true;
}

static void mixed_io_slot(const xv_mixed_io* ring, size_t s, size_t na, size_t nl, size_t nc, size_t LA, size_t LO, xv_mixed_io* out) {
  xv_mixed_io io = *ring;
  io.a_action += s * na; io.a_obs += s * na; io.a_reward += s * na; io.a_reward_gt += s * na;
  io.a_terminated += s * na; io.a_truncated += s * na;
  if (io.a_final_obs) io.a_final_obs += s * na;
  io.l_action += s * nl * LA; io.l_obs += s * nl * LO; io.l_reward += s * nl; io.l_terminated += s * nl;
  io.l_truncated += s * nl; io.l_cmd += s * nl * LO; io.l_error += s * nl;
  if (io.l_final_obs) io.l_final_obs += s * nl * LO;
  io.c_action += s * nc; io.c_obs += s * nc * 4; io.c_reward += s * nc; io.c_terminated += s * nc; io.c_truncated += s * nc;
  if (io.c_final_obs) io.c_final_obs += s * nc * 4;
  io.a_steps = nullptr; io.a_done = nullptr; io.l_steps = nullptr; io.l_done = nullptr; io.c_done = nullptr;   // xv_mixed_step only
  *out = io;
}

template <int AG, int ABK, int LNS>
static void mixed_launch_replay_v(dim3 grid, hipStream_t st, const AnyMDPArgs& A, const AnyMDPStepIO& aio, int nbA, const LinDSArgs& L,
                                  const LinDSStepIO& lio, int nbL, const CartPoleArgs& C, const CartPoleIO& cio, int period, int cycles,
                                  int mode, const MixedSnap& S) {
  hipLaunchKernelGGL((mixed_replay_kernel<AG, ABK, LNS, 16>), grid, dim3(256), 0, st, A, aio, nbA, L, lio, nbL, C, cio, period, cycles, mode, S);
}

static void* mixed_hand_fn(int v) {
  switch (v) {
    case 0: return reinterpret_cast<void*>(&mixed_step_kernel<1, 0, 16, 16, true>);
    case 1: return reinterpret_cast<void*>(&mixed_step_kernel<1, 0, 32, 16, true>);
    case 2: return reinterpret_cast<void*>(&mixed_step_kernel<1, 1, 16, 16, true>);
    default: return reinterpret_cast<void*>(&mixed_step_kernel<1, 1, 32, 16, true>);
  }
}

// -> ring cycles per graph (built / reused), 0: this call is not overlapped, -1: failure
static int mixed_pipe_graphs(MixedPipe& M, xv_anymdp* a, xv_linds* l, xv_cartpole* c, const xv_mixed_io* ring, int period, int ring_cycles,
                             int mode, int v, int n_lw, int D) {
  MixedPipeKey K;
  memset(&K, 0, sizeof(K));
  memcpy(&K.A, &a->a, sizeof(K.A)); memcpy(&K.L, &l->a, sizeof(K.L)); memcpy(&K.C, &c->a, sizeof(K.C));
  K.A.tick = 0; K.A.tick_dev = nullptr; K.L.tick = 0; K.L.tick_dev = nullptr; K.C.tick = 0; K.C.tick_dev = nullptr;
  K.L.hand = M.d_hand; K.C.hand = M.d_hand + n_lw;
  memcpy(&K.ring, ring, sizeof(K.ring));
  K.period = period; K.mode = mode; K.variant = v; K.depth = D;
  const bool same = M.key_valid && M.exec[0] && M.exec[1] && memcmp(&K, &M.key, sizeof(K)) == 0;
  const int U = xv_pipe_pick_unroll(period, ring_cycles, same ? M.gate.unroll[0] : 0, D);
  if (U == 0) return 0;
  if (same && U == M.gate.unroll[0]) return U;
  (void)hipStreamSynchronize(M.side);
  if (M.side2) (void)hipStreamSynchronize(M.side2);
  (void)hipStreamSynchronize(a->eng->stream);
  mixed_pipe_drop_graphs(M);
  if (!xv_pipe_gate_sync(&M.gate)) return -1;
  const size_t na = (size_t)a->a.n_env, nl = (size_t)l->a.n_env, nc = (size_t)c->a.n_env;
  int nbA = xv_div_up(a->a.n_env, 256), nbL = xv_div_up(xv_div_up(l->a.n_slot, 16), 4), nbC = xv_div_up(c->a.n_env, 256);
  void* fn = mixed_hand_fn(v);
  for (int q = 0; q < D; ++q) {
    if (hipGraphCreate(&M.graph[q], 0) != hipSuccess) return -1;
    hipGraphNode_t prev = nullptr;
    {
      uint64_t* w = M.d_tick + q;
      uint64_t dv = (uint64_t)period * (uint64_t)U;
      uint32_t* seen = M.gate.d_seen + q;      // every stream's graph starts with the cycle gate (xv_pipe.h)
      const uint32_t* issued = M.gate.d_issued;
      uint32_t* err = a->a.err;
      void* hparams[5] = {&w, &dv, &seen, &issued, &err};
      hipKernelNodeParams np;
      memset(&np, 0, sizeof(np));
      np.func = reinterpret_cast<void*>(&mixed_pipe_head_kernel); np.gridDim = dim3(1); np.blockDim = dim3(1); np.kernelParams = hparams;
      if (hipGraphAddKernelNode(&prev, M.graph[q], nullptr, 0, &np) != hipSuccess) return -1;
    }
    for (int g = q; g < U * period; g += D) {      // step g of the graph set: stream g % D, ring slot g % period, tick base + g
      const int s = g % period;
      AnyMDPArgs A = K.A; LinDSArgs L = K.L; CartPoleArgs C = K.C;
      const uint64_t tk = (uint64_t)g;
      A.tick = tk; A.tick_dev = M.d_tick + q;
      L.tick = tk; L.tick_dev = M.d_tick + 4 + q;
      C.tick = tk; C.tick_dev = M.d_tick + 8 + q;
      xv_mixed_io io;
      mixed_io_slot(ring, (size_t)s, na, nl, nc, (size_t)l->a.NA, (size_t)l->a.NO, &io);
      AnyMDPStepIO aio{io.a_action, nullptr, nullptr, nullptr, io.a_obs, io.a_reward, io.a_reward_gt, io.a_terminated,
                       io.a_truncated, io.a_final_obs, nullptr, nullptr, 0.0f};
      LinDSStepIO lio{io.l_action, nullptr, nullptr, io.l_obs, io.l_reward, io.l_terminated, io.l_truncated, io.l_cmd,
                      io.l_error, io.l_final_obs};
      CartPoleIO cio{io.c_action, nullptr, io.c_obs, io.c_reward, io.c_terminated, io.c_truncated, io.c_final_obs};
      void* params[9] = {&A, &aio, &nbA, &L, &lio, &nbL, &C, &cio, &mode};
      hipKernelNodeParams np;
      memset(&np, 0, sizeof(np));
      np.func = fn; np.gridDim = dim3(nbA + nbL + nbC); np.blockDim = dim3(256); np.kernelParams = params;
      hipGraphNode_t node = nullptr;
      if (hipGraphAddKernelNode(&node, M.graph[q], prev ? &prev : nullptr, prev ? 1 : 0, &np) != hipSuccess) return -1;
      prev = node;
    }
    if (hipGraphInstantiate(&M.exec[q], M.graph[q], nullptr, nullptr, 0) != hipSuccess) { M.exec[q] = nullptr; return -1; }
  }
  M.key = K; M.key_valid = true;
  M.gate.unroll[0] = U;
  return U;
}

// whole ring cycles of a call, overlapped; *issued = steps issued (0: the caller's loop takes all of them)
static int mixed_pipe_run(xv_anymdp* a, xv_linds* l, xv_cartpole* c, const xv_mixed_io* ring, int n_steps, int period, int mode,
                          int* issued) {
  *issued = 0;
  const int dev = a->eng->device;
  if (dev < 0 || dev >= 64) return XV_OK;
  std::lock_guard<std::mutex> lock(g_mixed_mu);
  MixedPipe& M = g_mixed_pipe[dev];
  M.used_last = false; M.used_by = a;
  const int ring_cycles = n_steps / period;
  static const int min_steps = getenv("XV_MIXED_PIPE_MIN_STEPS") ? atoi(getenv("XV_MIXED_PIPE_MIN_STEPS")) : XV_MIXED_PIPE_MIN;
  if (ring_cycles <= 0 || period % 2 != 0 || n_steps < min_steps) return XV_OK;
  if (M.failed && M.side_for == a->eng->stream) return XV_OK;      // tried beside this stream already
  // a recent call with this handle was replayed: the ordinary loop for a while (the count lives in the AnyMDP handle — switching
  // its overlap on again starts over — against this device's replay counter)
  if (a->backoff_mixed.len == 0) xv_pipe_backoff_reset(&a->backoff_mixed, &M.gate);
  if (xv_pipe_backoff_step(&a->backoff_mixed, &M.gate)) return XV_OK;
  M.failed = false;
  if (a->eng == l->eng || a->eng == c->eng || l->eng == c->eng) return XV_OK;      // one tick per family and step
  if (a->eng->dev_tick || l->eng->dev_tick || c->eng->dev_tick) return XV_OK;
  if (a->eng->stream != l->eng->stream || a->eng->stream != c->eng->stream || l->eng->device != dev || c->eng->device != dev) return XV_OK;
  const int v = mixed_variant(a, l);
  if (v < 0) return XV_OK;
  XV_HIP(hipSetDevice(dev));
  // two or three launches resident at once, or the ordinary loop (xv_pipe.h)
  int D = xv_pipe_choose_depth(mixed_hand_fn(v), 256,
                               (size_t)(xv_div_up(a->a.n_env, 256) + xv_div_up(xv_div_up(l->a.n_slot, 16), 4) + xv_div_up(c->a.n_env, 256)), dev);
  if (D < 2) return XV_OK;
  // three launches in flight pay for the AnyMDP step (anymdp.hip) but not here: 4.26-4.36 us with two, 4.41 with three
  // (profiles/r05_y_*) — the LinDS chain, not the dispatch gap, bounds the fused step.  (The machinery takes D = 3: a ring
  // must then hold at least three slots, steps k and k + 2 would otherwise write one output slot at the same time.)
  if (D > 2) D = 2;
  hipStream_t st = a->eng->stream;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return XV_OK; }
  const int n_lw = xv_div_up(l->a.n_slot, 16), n_cw = xv_div_up(c->a.n_env, 64);
  anymdp_bind_rng(a, 0, false);
  linds_bind_rng(l, 0, false);
  cartpole_bind_rng(c, 0, false);
  const int U = mixed_pipe_setup(M, st, (size_t)(n_lw + n_cw), D) ? mixed_pipe_graphs(M, a, l, c, ring, period, ring_cycles, mode, v, n_lw, D) : -1;
  if (U < 0) {
    (void)hipGetLastError();
    M.failed = true;
    return XV_OK;
  }
  if (U == 0) return XV_OK;      // too short for the graphs held: the ordinary loop
  const int per_launch = U * period, cycles = ring_cycles / U;      // launches of the two cycle graphs (U ring cycles each)
  const uint64_t ta = a->eng->tick, tl = l->eng->tick, tc = c->eng->tick;
  // the entry state of the call, kept for the replay (mixed_replay_kernel): env records, LinDS tiles, CartPole states, error words
  const int test_fail = getenv("XV_PIPE_TEST_FAIL") ? atoi(getenv("XV_PIPE_TEST_FAIL")) : 0;
  const size_t n_tile = ((size_t)l->a.n_slot + 15) / 16;
  const size_t snap_bytes[6] = {(size_t)a->a.n_env * sizeof(uint2), sizeof(float) * n_tile * (size_t)(l->a.NS / 16) * 256,
                                sizeof(int32_t) * n_tile * 16, sizeof(double) * 4 * (size_t)c->a.n_env,
                                sizeof(int32_t) * (size_t)c->a.n_env, (size_t)c->a.n_env};
  size_t snap_off[6], snap_total = 0;
  for (int i = 0; i < 6; ++i) { snap_off[i] = snap_total; snap_total += (snap_bytes[i] + 255) / 256 * 256; }
  bool snap_ok = true;
  if (M.snap_cap < snap_total) {
    (void)hipStreamSynchronize(st);
    if (M.d_snap) (void)hipFree(M.d_snap);
    M.d_snap = nullptr; M.snap_cap = 0;
    if (hipMalloc(&M.d_snap, snap_total) == hipSuccess) M.snap_cap = snap_total;
    else { (void)hipGetLastError(); M.d_snap = nullptr; snap_ok = false; }
  }
  if (snap_ok) {
    MixedSnapList SL;
    const void* srcs[6] = {a->a.sr, l->a.x, l->a.sn, c->a.state, c->a.steps, c->a.need_reset};
    for (int i = 0; i < 6; ++i) SL.e[i] = MixedSnapCopy{M.d_snap + snap_off[i], srcs[i], snap_bytes[i]};
    SL.w = M.d_snap_w; SL.err[0] = a->a.err; SL.err[1] = l->a.err; SL.err[2] = c->a.err;
    hipLaunchKernelGGL(mixed_pipe_snap_kernel, dim3(64, 6), dim3(256), 0, st, SL);
    snap_ok = hipGetLastError() == hipSuccess;
  }
  if (!snap_ok) { (void)hipGetLastError(); M.failed = true; return XV_OK; }      // no repair possible: the ordinary loop
  M.fell_seen = __atomic_load_n(M.gate.h_issued + 1, __ATOMIC_ACQUIRE);
  const int n_open = std::max(std::max(a->a.n_env, c->a.n_env), std::max(n_lw, 12));
  hipLaunchKernelGGL(mixed_pipe_open_kernel, dim3(xv_div_up(n_open, 256)), dim3(256), 0, st, a->a.sr, a->a.n_env, XV_ANYMDP_SR_TAG(ta),
                     M.d_hand, n_lw, (uint32_t)tl, M.d_hand + n_lw, n_cw, (uint32_t)tc, c->a.steps, c->a.need_reset, c->a.n_env,
                     M.d_tick, ta - (uint64_t)per_launch, tl - (uint64_t)per_launch, tc - (uint64_t)per_launch);   // the head nodes add it
  bool ok = hipGetLastError() == hipSuccess && hipEventRecord(M.ev[0], st) == hipSuccess &&
            hipStreamWaitEvent(M.side, M.ev[0], 0) == hipSuccess && (D < 3 || hipStreamWaitEvent(M.side2, M.ev[0], 0) == hipSuccess);
  int k = 0;
  bool broken = false;
  if (ok) {
    for (int cy = 0; cy < cycles; ++cy) {
      // both halves or neither: an even half without its odd half leaves the next even launch waiting (bounded, flagged)
      // (the even half starts with the cycle gate: it runs once both halves are enqueued, however long the host takes)
      if (hipGraphLaunch(M.exec[0], st) != hipSuccess) break;
      xv_pipe_test_stall(cy);
      if (hipGraphLaunch(M.exec[1], M.side) != hipSuccess) { broken = true; xv_pipe_gate_release(&M.gate); break; }
      if (D == 3) xv_pipe_test_stall(cy, 1);
      if (D == 3 && hipGraphLaunch(M.exec[2], M.side2) != hipSuccess) { broken = true; xv_pipe_gate_release(&M.gate); break; }
      xv_pipe_gate_release(&M.gate);
      k += per_launch;
    }
  }
  a->eng->tick = ta + (uint64_t)k; l->eng->tick = tl + (uint64_t)k; c->eng->tick = tc + (uint64_t)k;
  const bool joined = hipEventRecord(M.ev[1], M.side) == hipSuccess && hipStreamWaitEvent(st, M.ev[1], 0) == hipSuccess &&
                      (D < 3 || (hipEventRecord(M.ev2, M.side2) == hipSuccess && hipStreamWaitEvent(st, M.ev2, 0) == hipSuccess));
  hipLaunchKernelGGL(mixed_pipe_close_kernel, dim3(xv_div_up(c->a.n_env, 256)), dim3(256), 0, st, c->a.steps, c->a.need_reset, c->a.n_env);
  bool closed = hipGetLastError() == hipSuccess;
  if (closed && joined && !broken && k > 0) {
    // should a hand-off of this call have expired: the call is replayed from its entry state on this stream (a nearly
    // empty launch otherwise) — an expiry costs time, never results
    const size_t LA = (size_t)l->a.NA, LO = (size_t)l->a.NO;
    if (test_fail)
      hipLaunchKernelGGL(mixed_test_fail_kernel, dim3(256), dim3(256), 0, st, a->a.sr, a->a.n_env, l->a.x, snap_bytes[1] / sizeof(float),
                         c->a.state, c->a.n_env, l->a.err, ring->l_obs, (size_t)period * (size_t)l->a.n_env * LO, ring->a_obs,
                         (size_t)period * (size_t)a->a.n_env);
    AnyMDPArgs A = a->a; LinDSArgs L = l->a; CartPoleArgs C = c->a;
    A.seed = a->eng->seed; A.gid_base = a->eng->env_id_base; A.tick = ta; A.tick_dev = nullptr;
    L.seed = l->eng->seed; L.gid_base = l->eng->env_id_base; L.tick = tl; L.tick_dev = nullptr;
    C.seed = c->eng->seed; C.gid_base = c->eng->env_id_base; C.tick = tc; C.tick_dev = nullptr;
    (void)LA;
    AnyMDPStepIO aio{ring->a_action, nullptr, nullptr, nullptr, ring->a_obs, ring->a_reward, ring->a_reward_gt, ring->a_terminated,
                     ring->a_truncated, ring->a_final_obs, nullptr, nullptr, 0.0f};
    LinDSStepIO lio{ring->l_action, nullptr, nullptr, ring->l_obs, ring->l_reward, ring->l_terminated, ring->l_truncated, ring->l_cmd,
                    ring->l_error, ring->l_final_obs};
    CartPoleIO cio{ring->c_action, nullptr, ring->c_obs, ring->c_reward, ring->c_terminated, ring->c_truncated, ring->c_final_obs};
    MixedSnap S;
    S.a_sr = reinterpret_cast<const uint2*>(M.d_snap + snap_off[0]); S.l_x = reinterpret_cast<const float*>(M.d_snap + snap_off[1]);
    S.l_sn = reinterpret_cast<const int32_t*>(M.d_snap + snap_off[2]); S.c_state = reinterpret_cast<const double*>(M.d_snap + snap_off[3]);
    S.c_steps = reinterpret_cast<const int32_t*>(M.d_snap + snap_off[4]); S.c_nr = M.d_snap + snap_off[5];
    S.w = M.d_snap_w; S.err[0] = a->a.err; S.err[1] = l->a.err; S.err[2] = c->a.err; S.h_fell = M.gate.d_issued + 1;
    const int nbA = xv_div_up(a->a.n_env, 256), nbL = xv_div_up(xv_div_up(l->a.n_slot, 16), 4), nbC = xv_div_up(c->a.n_env, 256);
    const dim3 grid(nbA + nbL + nbC);
    switch (v) {
      case 0: mixed_launch_replay_v<1, 0, 16>(grid, st, A, aio, nbA, L, lio, nbL, C, cio, period, k / period, mode, S); break;
      case 1: mixed_launch_replay_v<1, 0, 32>(grid, st, A, aio, nbA, L, lio, nbL, C, cio, period, k / period, mode, S); break;
      case 2: mixed_launch_replay_v<1, 1, 16>(grid, st, A, aio, nbA, L, lio, nbL, C, cio, period, k / period, mode, S); break;
      default: mixed_launch_replay_v<1, 1, 32>(grid, st, A, aio, nbA, L, lio, nbL, C, cio, period, k / period, mode, S); break;
    }
    closed = hipGetLastError() == hipSuccess;
  }
  *issued = k;
  if (!ok || k < cycles * per_launch) { (void)hipGetLastError(); M.failed = true; }
  if (broken || !joined || !closed) {
    (void)hipGetLastError();
    M.failed = true;
    xv_set_error("xv_mixed_step_many: an overlapped ring cycle could be issued only in part; the envs' states are undefined");
    return XV_ERR_HIP;
  }
  M.used_last = k > 0;
  return XV_OK;
}

// 1: the last xv_mixed_step_many with this AnyMDP handle overlapped its ring cycles, 0: it did not, -1: the overlapped path
// failed on this device (no concurrent streams, graph build) and is no longer tried, -2: the last call overlapped, a hand-off
// expired and the call was replayed on one stream (results are right; meaningful once the stream has drained)
extern "C" int xv_mixed_step_many_overlap_state(xv_anymdp* a) {
  if (!a) return 0;
  const int dev = a->eng->device;
  if (dev < 0 || dev >= 64) return 0;
  std::lock_guard<std::mutex> lock(g_mixed_mu);
  const MixedPipe& M = g_mixed_pipe[dev];
  if (M.failed) return -1;
  if (M.used_by == a && M.used_last && M.gate.h_issued && __atomic_load_n(M.gate.h_issued + 1, __ATOMIC_ACQUIRE) != M.fell_seen) return -2;
  return (M.used_by == a && M.used_last) ? 1 : 0;
}

// n_steps fused vector steps issued from C over ring buffers: step k uses slot k % period of every [period][...] array
extern "C" int xv_mixed_step_many(xv_anymdp* a, xv_linds* l, xv_cartpole* c, const xv_mixed_io* ring, int n_steps, int period,
                                  int autoreset_mode) {
  XV_CHECK_ARG(a && l && c && ring && n_steps > 0 && period > 0);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  const size_t na = (size_t)a->a.n_env, nl = (size_t)l->a.n_env, nc = (size_t)c->a.n_env;
  int k = 0;
  if (a->overlap) {
    XV_CHECK_ARG(ring->a_action && ring->a_obs && ring->a_reward && ring->a_reward_gt && ring->a_terminated && ring->a_truncated);
    XV_CHECK_ARG(ring->l_action && ring->l_obs && ring->l_reward && ring->l_terminated && ring->l_truncated && ring->l_cmd && ring->l_error);
    XV_CHECK_ARG(ring->c_action && ring->c_obs && ring->c_reward && ring->c_terminated && ring->c_truncated);
    const int rc = mixed_pipe_run(a, l, c, ring, n_steps, period, autoreset_mode, &k);
    if (rc != XV_OK) return rc;
  }
  for (; k < n_steps; ++k) {
    xv_mixed_io io;
    mixed_io_slot(ring, (size_t)(k % period), na, nl, nc, (size_t)l->a.NA, (size_t)l->a.NO, &io);
    const int rc = xv_mixed_step(a, l, c, &io, autoreset_mode);
    if (rc != XV_OK) return rc;
  }
  return XV_OK;
}
