// anymdp.hip — AnyMDP batched step / reset / rollout kernels for gfx950 and their C-ABI.
//
// Reproduces xenoverse/anymdp/anymdp_env.py: reset :81-90, single_step :92-110, step :112-132,
// get_observation :145-159 (MDP branch), per env, for N envs per launch.  One wavefront lane owns one env.
//
// A step is a chain of DEPENDENT memory accesses (state -> which row -> which next state -> its reward and
// observation).  With one task per env every table access is a random 128-byte HBM line (measured,
// scripts/devtools/gather_granularity.hip: the fetch granule is 128 B, ~5e10 random lines/s, and a lane-divergent
// load instruction costs the texture-addresser once per distinct line), so the kernel is priced in LINES and in
// dependent LEVELS.  Layout and kernel make a step two lines in two levels:
//
//   1. per-env words — state, steps, action, task id and a 36-byte reset record (s_0 CDF, ids, observation ids,
//      max_steps; built per env at create time) — struct-of-arrays, coalesced streams, no dependency.
//   2. the FENCE line of row (t,s,a): 16 doubles, fence[k] = the CDF entry of the last next-state of block k.
//      k = #{fence <= u} names the one block that contains s' (the CDF is non-decreasing).
//   3. that ONE 128-byte BLOCK: 7 entries {fp64 cdf, reward, noise} and 16 bytes holding the observation id and
//      terminal flag of the same 7 next states.  s' = 7k + #{cdf <= u} is numpy.searchsorted(cdf, u, 'right');
//      reward, observation and termination come out of the same line.  No dependent gather follows.
//
// Both lines are read cooperatively: 8 lanes x 16 B per env, 8 envs per load instruction, 8 instructions per
// level kept in flight together; the search is v_cmp_le_f64 + 64-bit ballot + 8-bit popcount, and results
// return to the owning lane by ds_bpermute.  Per env-step: 2 x 128 B of table lines + ~90 B of streams
// (the first layout read a 32-B fence record, a 256-B block and a 128-B task header: 4 lines).
// BINARY mode (any S <= 256, any s0 table) is the general per-lane fallback.
//
// BUCKET mode (xv_anymdp_build_buckets: it spends memory) removes level 2.  The row's probability axis is cut into NBK
// equal buckets; bucket line (row, k) answers every u in [k / NBK, (k + 1) / NBK) by itself.  The line's address follows
// from the row index and the env's own uniform (k = floor(u * NBK), no memory), so a step is ONE table line in ONE
// dependent level.  Round 4: a line lists K CUTS of the row's CDF chosen for this bucket (anymdp_cutline.h) instead of
// the 7 consecutive entries that start at #{cdf <= k / NBK}: unit c = {cut_c, reward pair of group c's next state}, the
// metadata names each group's next state, its observation id and terminal flag, and flags the groups that lump a run of
// (practically never drawn) states together.  c = #{cut <= u}; a draw that lands in a flagged group or beyond the last
// cut sends its wave through the fence path — same result.  On the reference sampler's rows that is 5e-8 of the draws
// (2.7e-2 with consecutive entries, which made the search slower than the fence search there).
#include "anymdp_cutline.h"
#include "philox.h"
#include "xv_common.h"
#include "xv_pipe.h"

#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>

#define XV_ANYMDP_BLK 7   // next states per block
#define XV_ANYMDP_MAX_CHAINS 16
// (the owner lane keeps its result by SELECTS throughout: written as `if (g == it) {...}` hipcc emits an exec-masked branch per
//  env group — 45 -> 14 branches in the round-3 step kernel, 5.31-5.34 -> 5.05-5.13 us per step)
#ifndef XV_ANYMDP_NT_OUT
#define XV_ANYMDP_NT_OUT 1   // the step's outputs leave with non-temporal stores (written once, read by somebody else): 5.41-5.73
#endif                       // -> 5.27-5.32 us per 65,536-env step on the same box, two A/B rounds (scripts/runs_r03/gpu_n.sh)

struct AnyMDPArgs {
  // borrowed task tables
  const uint4* lines;    // rows in 16-byte units; a line = 8 units; row r starts at line r * RL
  const int32_t* state_map;
  const uint64_t* term_mask;
  const double* s0_cdf;
  const int32_t* s0_ids;
  const int32_t* max_steps;
  const int32_t* env_task;
  // engine-owned per-env reset record (fast path): three 16-byte units per env, one coalesced stream each
  const double2* rs_a;       // s0_cdf[0], s0_cdf[1]   (padded with 1.0)
  const uint4* rs_b;         // .xy = s0_cdf[2] (fp64 bits); .zw = 4 x u16 inner state ids of s_0, padded with the last
  const uint4* rs_c;         // .xy = 4 x u16 observation ids of those states; .z = max_steps (bits 0..26) | terminal flags
                             // of the four states (bits 27..30); .w = the env's task index
  // engine-owned env state: ONE 8-byte record per env — .x = inner state (bits 0..15) | current state is terminal (bit 16:
  // the reference raises when stepping from it, :95-96) | need_reset (bit 17); .y = steps
  uint2* sr;
  uint32_t* err;
  int n_env, n_task, S, A, s0_max, words, NB, RL, G;   // RL = 1 + NB lines per row; G blocks per fence entry
  const uint4* bucket;   // [row][NBK] bucket lines (engine-owned, xv_anymdp_build_buckets); nullptr if not built
  int NBK;
  int bfmt;              // metadata packing of the bucket lines: 1 = 7 cuts (S <= 256, observation ids <= 255), 2 = 6 cuts
  uint64_t seed, gid_base, tick;
  const uint64_t* tick_dev;   // graph replay: the launch tick is *tick_dev + tick (tick = node index); else nullptr
};

struct AnyMDPStepIO {
  const int32_t* action;
  const double* u;        // injected draws (INJECT only)
  const float* z;
  const double* u_reset;
  int32_t* obs;
  float* reward;
  float* reward_gt;
  uint8_t* terminated;
  uint8_t* truncated;
  int32_t* final_obs;     // nullable
  // teacher rollout (nullable): action = greedy[task][state] w.p. 1-epsilon, else uniform; written to action_out
  const uint8_t* greedy;
  int32_t* action_out;
  float epsilon;
  // info["steps"] and the terminated | truncated mask of the same step (xv_anymdp_step_info; nullable): a host that wants them
  // needs no second launch and no elementwise op of its own
  int32_t* steps_out;
  uint8_t* done_out;
};

struct xv_anymdp {
  xv_engine* eng;
  AnyMDPArgs a;
  int search;  // XV_ANYMDP_SEARCH_*
  bool fast;   // fence lines, block metadata and reset records are built
  uint4* bucket_rw;   // owned: the bucket lines
  int max_obs;        // largest observation id of the tables (-1: not probed)
  xv_anymdp_bucket_census census;   // of the lines that are built (xv_anymdp_build_buckets)
  const double* obs_cdf;   // observation model (POMDP / MTPOMDP), nullptr for MDP
  int n_obs, d_obs, d_act;
  uint4* obs_bucket;       // owned: observation bucket lines (built with the transition bucket lines)
  // xv_anymdp_step_many: one ring cycle (period launches + a tick update) as an instantiated hipGraph
  int graph_mode;            // 0 off, 1 on, 2 auto: on for n_env <= XV_ANYMDP_GRAPH_AUTO_MAX
  bool graph_failed;
  bool graph_used_last;      // the last xv_anymdp_step_many replayed the graph
  hipGraph_t graph;
  hipGraphExec_t graph_exec;
  uint64_t* d_tick;          // device copy of the launch tick the graph's kernels read
  uint64_t d_tick_value;     // what *d_tick holds once the stream has drained
  bool d_tick_valid;
  struct {
    int period, mode, search, fast, nbk;
    uint64_t seed, gid_base;
    size_t stride;
    const void* ptrs[7];
    const void* bucket;   // the bucket lines and their count are baked into the kernel nodes' arguments
  } graph_key;
  // overlapped step_many (xv_anymdp_set_step_many_overlap): even ring slots on the engine's stream, odd ones on `side`, two
  // cycle graphs of HAND kernels, each with its own tick word; hand[] as in AnyMDPArgs
  int overlap;               // 0 off, 1 on
  bool pipe_failed;
  bool pipe_used_last;       // the last xv_anymdp_step_many issued overlapped steps
  hipStream_t side;
  hipStream_t side_for;      // the engine's stream `side` was chosen against (xv_pipe.h)
  XvPipeGate gate;           // the even half of a cycle starts once the host has enqueued both halves (xv_pipe.h)
  hipEvent_t side_ev[2];     // fork, join
  hipGraph_t pgraph[2];
  hipGraphExec_t pgraph_exec[2];
  uint64_t* d_ptick;         // two tick words
  uint64_t ptick_value;      // what both hold once the streams have drained (valid with pgraph_exec)
  bool ptick_valid;
  // an expired hand-off is repaired, not just flagged: the call's entry state is kept and the call replayed on one stream
  // (anymdp_replay_kernel below)
  uint2* d_snap;             // the env records at the entry of the last overlapped call
  uint32_t* d_snap_w;        // [0] the error word at entry, [1] the replay's own error bits, [2] workgroups finished
  uint32_t fell_seen;        // gate.h_issued[1] (calls replayed so far) when the last overlapped call was issued
  XvPipeBackoff backoff;     // one-stream calls after a replayed one (xv_pipe.h)
  XvPipeBackoff backoff_mixed;   // the same for xv_mixed_step_many with this handle (mixed.hip; len == 0: not started)
  struct {
    int period, mode, search;
    size_t stride;
    const void* ptrs[7];
    const void* bucket;
    uint64_t seed, gid_base;
  } pipe_key;
  hipGraph_t tgraph[2];      // the same for xv_anymdp_step_tokens_many (cooperative token kernel, HAND)
  hipGraphExec_t tgraph_exec[2];
  struct {
    int period, mode, fmt, d_obs, d_act;
    const void* ptrs[7];
    const void* bucket;
    const void* obs_bucket;
    uint64_t seed, gid_base;
  } tpipe_key;
  // views (xv_anymdp_view): a handle over envs [view_lo, view_lo + a.n_env) of `parent` with an engine (stream, tick,
  // error word) of its own; tables, env records, bucket and observation lines are the parent's (borrowed, never freed here)
  xv_anymdp* parent;
  int view_lo;
  int n_views;               // parent: live views (the parent's tables may not change or go while > 0)
  hipEvent_t chain_ev;       // xv_anymdp_step_many_chains: fork (parent) / join (view) event, made on first use
  // xv_anymdp_step_many_chains, how = 1: ONE graph whose K branches are the views' chains (kept on the parent)
  hipGraph_t cgraph;
  hipGraphExec_t cgraph_exec;
  struct {
    int period, mode, search, n_views;
    const void* ptrs[7];
    const void* bucket;
    const void* views[XV_ANYMDP_MAX_CHAINS];
  } cgraph_key;
};

#define XV_ANYMDP_SR_TERM 0x10000u
#define XV_ANYMDP_SR_NR 0x20000u
// bits 18..31 of .x: hand-off tag of the overlapped step_many (HAND kernels) — the low 14 bits of the launch tick of the step
// that may take the env next; every other kernel ignores the bits and writes them as 0
#define XV_ANYMDP_SR_TAG_SHIFT 18
#define XV_ANYMDP_SR_TAG(tick) ((uint32_t)(tick) & 0x3FFFu)
__device__ __forceinline__ uint2 anymdp_sr_pack(int s, int steps, int nr, int cterm) {
  return make_uint2((uint32_t)s | (cterm ? XV_ANYMDP_SR_TERM : 0u) | (nr ? XV_ANYMDP_SR_NR : 0u), (uint32_t)steps);
}

static __global__ void anymdp_init_sr_kernel(uint2* sr, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sr[i] = make_uint2(XV_ANYMDP_SR_NR, 0u);
}
static __global__ void anymdp_set_tick_kernel(uint64_t* t, uint64_t v) { *t = v; }
static __global__ void anymdp_advance_tick_kernel(uint64_t* t, uint64_t dv) { *t += dv; }
// first node of an overlapped cycle graph: this cycle's tick base; the even half also passes the cycle gate (xv_pipe.h)
static __global__ void anymdp_pipe_head_kernel(uint64_t* t, uint64_t dv, uint32_t* seen, const uint32_t* issued, uint32_t* err) {
  *t += dv;
  if (seen) xv_pipe_gate_pass(seen, issued, err);
}

__device__ __forceinline__ bool anymdp_is_term(const AnyMDPArgs& P, int t, uint64_t tm0, int s) {
  if (P.words == 1) return (tm0 >> s) & 1ull;
  return (P.term_mask[(size_t)t * P.words + (s >> 6)] >> (s & 63)) & 1ull;
}

// s = s0_ids[upper_bound(s0_cdf, u)]   (anymdp_env.py:89: numpy.random.choice(self.s_0, p=self.s_0_prob))
__device__ __forceinline__ int anymdp_draw_s0(const AnyMDPArgs& P, int t, double u) {
  const double* c = P.s0_cdf + (size_t)t * P.s0_max;
  int k = 0;
  while (k < P.s0_max - 1 && c[k] <= u) ++k;
  return P.s0_ids[(size_t)t * P.s0_max + k];
}

// row accessors: the 16-byte entry of next state j of row r = {double cdf; float reward; float noise}
__device__ __forceinline__ const uint4* anymdp_entry_ptr(const AnyMDPArgs& P, uint32_t r, int j) {
  const int b = j / XV_ANYMDP_BLK;
  return P.lines + ((size_t)r * P.RL + 1 + b) * 8 + (j - b * XV_ANYMDP_BLK);
}
__device__ __forceinline__ double anymdp_cdf(const AnyMDPArgs& P, uint32_t r, int j) {
  return *reinterpret_cast<const double*>(anymdp_entry_ptr(P, r, j));
}
__device__ __forceinline__ float2 anymdp_rs(const AnyMDPArgs& P, uint32_t r, int j) {
  return reinterpret_cast<const float2*>(anymdp_entry_ptr(P, r, j))[1];
}

__device__ __forceinline__ double xv_shfl_f64(double v, int src) {
  return __hiloint2double(__shfl(__double2hiint(v), src), __shfl(__double2loint(v), src));
}
__device__ __forceinline__ double xv_u2d(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }

// Bucket ("cut") line metadata, unit 7 of the line (and unit 6 in the wide packing), per group `sg` of the line: observation id,
// next state, terminal flag, and whether the group lumps several states or is unused:
//   FMT 1: unit 7 = bytes 0..6 observation ids, byte 7 terminal bits, bytes 8..14 next states, byte 15 lumped bits (7 groups)
//   FMT 2: unit 6 = 6 x u16 next states; unit 7 = 6 x u16 observation ids, .w = terminal bits | lumped bits << 8 (6 groups)
// resolve_entry() hands the owner lane one word: observation id (bits 0..15) | next state (16..24) | terminal (25) | lumped (26).
//
// One 128-byte line per env, read by LPE lanes x UPL = 8 / LPE units of 16 bytes: round `it` (of LPE) serves the 64 / LPE envs
// it * 64 / LPE + g, lane LPE g + jl reads UPL units of the line of env g's round.  CONTIG: lane jl reads units UPL jl ..
// UPL jl + UPL - 1; otherwise units jl, jl + LPE, ... (one load instruction then covers LPE * 16 contiguous bytes of every
// line).  issue() only requests; the resolve functions wait and hand each owner lane (lane l owns env l) its result, so that
// independent work can be placed under the latency in between.  The uniforms travel to the reader lanes right behind the
// line requests (send(), under their latency), not inside the resolve loops (A/B: profiles/r04_k_*).
// One wave per SIMD runs the token step at 65,536 envs, so the resolve code IS on the critical path: per round it costs a
// ballot per unit slot, the metadata extraction and the trips back to the owner lanes; fewer lanes per env = fewer rounds
// (table below).
// Lanes per env, measured (65,536 envs, scripts/runs_r04/gpu_q.sh and gpu_r.sh):
//   token step (2 + 2 tokens, ~1.3 GiB of lines)   LPE 8: 13.2 us   4: 11.0   2: 10.6   1: 11.5      (interleaved units: +0.1)
//   MDP step (config 2a, 8 GiB of lines)           LPE 8: 5.23 us   4: 5.12   2: 5.37   1: 5.81      (interleaved units: same)
// fewer lanes per env = fewer resolve rounds but more distinct lines per load instruction; the token step has four resolves
// per step on its critical path, the MDP step one.
#ifndef XV_ANYMDP_STEP_LPE
#define XV_ANYMDP_STEP_LPE 4
#endif
#ifndef XV_ANYMDP_TOK_LPE
#define XV_ANYMDP_TOK_LPE 2
#endif
#ifndef XV_ANYMDP_COOP_CONTIG
#define XV_ANYMDP_COOP_CONTIG 1
#endif
template <int LPE, bool CONTIG>
struct AnyMDPCoopLineN {
  static constexpr int UPL = 8 / LPE, EPR = 64 / LPE;          // units per lane, envs per round
  static constexpr unsigned GM = (1u << LPE) - 1u;
  uint4 v[LPE][UPL];
  double ue[LPE];
  static __device__ __forceinline__ constexpr int unit_of(int jl, int k) { return CONTIG ? jl * UPL + k : jl + k * LPE; }
  static __device__ __forceinline__ int lane_of(int unit) { return CONTIG ? unit / UPL : unit % LPE; }   // within the group
  static __device__ __forceinline__ int slot_of(int unit) { return CONTIG ? unit % UPL : unit / LPE; }
  __device__ __forceinline__ void issue(const uint4* base, uint32_t line, int lane) {
    const int g = lane / LPE, jl = lane % LPE;
#pragma unroll
    for (int it = 0; it < LPE; ++it) {
      const uint32_t li = LPE == 1 ? line : (uint32_t)__shfl((int)line, it * EPR + g);
      const uint4* lp = base + (size_t)li * 8;
#pragma unroll
      for (int k = 0; k < UPL; ++k) v[it][k] = lp[unit_of(jl, k)];
    }
  }
  // the same for the envs whose owner lane sets `want` only: the others read line 0 of the table (one cached line for
  // all of them; their results are ignored).  A step is priced in random 128-byte lines (~5e10 per second): none is
  // requested without need.  (An address select, not a branch: hipcc drains the load queue at the end of every
  // conditional block that holds a load, which serialised the requests.)
  __device__ __forceinline__ void issue_if(const uint4* base, uint32_t line, bool want, int lane) {
    issue(base, want ? line : 0u, lane);
  }
  // the env's uniform to the lanes that hold its line: call right after issue*()
  __device__ __forceinline__ void send(double u, int lane) {
    const int g = lane / LPE;
#pragma unroll
    for (int it = 0; it < LPE; ++it) ue[it] = LPE == 1 ? u : xv_shfl_f64(u, it * EPR + g);
  }
  // slot `ks` (a run-time value) of round `it`
  __device__ __forceinline__ uint4 slot(int it, int ks) const {
    uint4 r = v[it][0];
#pragma unroll
    for (int k = 1; k < UPL; ++k) {
      const bool p = ks == k;
      r.x = p ? v[it][k].x : r.x; r.y = p ? v[it][k].y : r.y; r.z = p ? v[it][k].z : r.z; r.w = p ? v[it][k].w : r.w;
    }
    return r;
  }
  // transition bucket line (anymdp_cutline.h): K cuts {cut, reward, noise} + metadata  ->  the reward pair of the group
  // c = #{cut <= u} and meta = its observation id | next state << 16 | terminal flag << 25 | lumped << 26 (the packings
  // are described above the struct); beyond = the line cannot answer this draw (c == K, or the group lumps several states): search the row
  template <int FMT>
  __device__ __forceinline__ void resolve_entry(double u, int lane, bool& beyond, float& rx, float& ry, uint32_t& meta) const {
    constexpr int KC = FMT == 2 ? 6 : 7;
    const int g = lane / LPE, jl = lane % LPE, qo = lane % EPR, ro = lane / EPR;
    int cnt = 0;
    rx = 0.0f; ry = 0.0f; meta = 0u;
#pragma unroll
    for (int it = 0; it < LPE; ++it) {
      int cg = 0, co = 0;   // reader side: the env this lane group serves; owner side: the env this lane owns
#pragma unroll
      for (int k = 0; k < UPL; ++k) {
        const unsigned long long m = __ballot(unit_of(jl, k) < KC && xv_u2d(v[it][k].x, v[it][k].y) <= ue[it]);
        cg += __popc((unsigned)(m >> (LPE * g)) & GM);
        co += __popc((unsigned)(m >> (LPE * qo)) & GM);
      }
      const int sg = cg < KC - 1 ? cg : KC - 1, so = co < KC - 1 ? co : KC - 1;
      const uint4 b7 = v[it][UPL - 1];                // unit 7 on the last lane of the group
      uint32_t packed, st16 = 0u;
      if (FMT == 1) {
        const uint32_t ow = sg < 4 ? b7.x : b7.y, sw = sg < 4 ? b7.z : b7.w;
        const int sh = 8 * (sg & 3);
        packed = ((ow >> sh) & 0xFFu) | (((sw >> sh) & 0xFFu) << 16) | (((b7.y >> (24 + sg)) & 1u) << 25) |
                 (((b7.w >> (24 + sg)) & 1u) << 26);
      } else {
        const uint32_t ow = sg < 2 ? b7.x : (sg < 4 ? b7.y : b7.z);
        packed = ((ow >> (16 * (sg & 1))) & 0xFFFFu) | (((b7.w >> sg) & 1u) << 25) | (((b7.w >> (8 + sg)) & 1u) << 26);
        const uint4 b6 = v[it][slot_of(6)];           // unit 6 (the next states) on lane lane_of(6) of the group
        const uint32_t sw = sg < 2 ? b6.x : (sg < 4 ? b6.y : b6.z);
        st16 = (sw >> (16 * (sg & 1))) & 0x1FFu;
      }
      const uint4 e = slot(it, slot_of(sg));          // group sg's entry, meaningful on lane lane_of(sg) of the group
      const float px = __shfl(__uint_as_float(e.z), LPE * qo + lane_of(so));
      const float py = __shfl(__uint_as_float(e.w), LPE * qo + lane_of(so));
      uint32_t pm = (uint32_t)__shfl((int)packed, LPE * qo + LPE - 1);
      if (FMT == 2) pm |= (uint32_t)__shfl((int)st16, LPE * qo + lane_of(6)) << 16;
      { const bool own = ro == it; cnt = own ? co : cnt; rx = own ? px : rx; ry = own ? py : ry; meta = own ? pm : meta; }
    }
    beyond = cnt >= KC || ((meta >> 26) & 1u);
  }
  // observation bucket line (anymdp_cutline.h, 14 cuts): units 0..6 = the cuts as doubles (2.0 when unused), unit 7 = the
  // groups' symbol ids (bytes 0..13) and their lumped / unused bits (bits 16..29 of .w)  ->  id of the group c = #{cut <= u};
  // beyond = the line cannot answer this draw (c == 14, or the group lumps several symbols): search the row
  __device__ __forceinline__ void resolve_obs(double u, int lane, bool& beyond, int& id) const {
    if constexpr (LPE == 1) {   // the owner lane holds its whole line: no ballots, no trips
      int c = 0;
#pragma unroll
      for (int k = 0; k < 7; ++k) c += (int)(xv_u2d(v[0][k].x, v[0][k].y) <= u) + (int)(xv_u2d(v[0][k].z, v[0][k].w) <= u);
      const int sg = c < 13 ? c : 13;
      const uint4 b4 = v[0][7];
      const uint32_t word = sg < 4 ? b4.x : (sg < 8 ? b4.y : (sg < 12 ? b4.z : b4.w));
      beyond = c >= 14 || ((b4.w >> (16 + sg)) & 1u);
      id = (int)((word >> (8 * (sg & 3))) & 0xFFu);
      return;
    }
    const int g = lane / LPE, jl = lane % LPE, qo = lane % EPR, ro = lane / EPR;
    int cnt = 0;
    uint32_t meta = 0u;
#pragma unroll
    for (int it = 0; it < LPE; ++it) {
      int cg = 0, co = 0;
#pragma unroll
      for (int k = 0; k < UPL; ++k) {
        const bool cutu = unit_of(jl, k) < 7;
        const unsigned long long m0 = __ballot(cutu && xv_u2d(v[it][k].x, v[it][k].y) <= ue[it]);
        const unsigned long long m1 = __ballot(cutu && xv_u2d(v[it][k].z, v[it][k].w) <= ue[it]);
        cg += __popc((unsigned)(m0 >> (LPE * g)) & GM) + __popc((unsigned)(m1 >> (LPE * g)) & GM);
        co += __popc((unsigned)(m0 >> (LPE * qo)) & GM) + __popc((unsigned)(m1 >> (LPE * qo)) & GM);
      }
      const int sg = cg < 13 ? cg : 13;
      const uint4 b4 = v[it][UPL - 1];
      const uint32_t word = sg < 4 ? b4.x : (sg < 8 ? b4.y : (sg < 12 ? b4.z : b4.w));
      const uint32_t packed = ((word >> (8 * (sg & 3))) & 0xFFu) | (((b4.w >> (16 + sg)) & 1u) << 8);
      const uint32_t pm = (uint32_t)__shfl((int)packed, LPE * qo + LPE - 1);
      { const bool own = ro == it; cnt = own ? co : cnt; meta = own ? pm : meta; }
    }
    beyond = cnt >= 14 || ((meta >> 8) & 1u);
    id = (int)(meta & 0xFFu);
  }
};
typedef AnyMDPCoopLineN<XV_ANYMDP_STEP_LPE, XV_ANYMDP_COOP_CONTIG != 0> AnyMDPStepLine;
typedef AnyMDPCoopLineN<XV_ANYMDP_TOK_LPE, XV_ANYMDP_COOP_CONTIG != 0> AnyMDPTokLine;

// T_steps == 1: one vector step.  T_steps > 1: fused rollout, io arrays are [T][n_env], mode SAME_STEP.
// FAST: fence line + block line, per-env reset record; otherwise per-lane binary search and per-task tables.
// G: 0 = per-lane binary search and per-task tables; 1..5 = fence path, a fence entry names G consecutive blocks
//    (G = 1 for S <= 112, 2 for S <= 224, 3 for S <= 336, 4 for S <= 448, 5 up to 512: the fence always fits one line,
//    the last level reads G lines).
// TICKDEV: the launch tick is *P.tick_dev + P.tick (graph replay); otherwise P.tick (a kernel argument).
// BK: bucket mode (0 off, 1 / 2 = the lines' metadata packing, AnyMDPArgs::bfmt): the step's table line is named by
//     (row, floor(u * NBK)); G only shapes the fence fall-back.
// `bid`: the workgroup's index within this family's part of the launch (blockIdx.x for the family's own kernels; the fused
// mixed-batch kernel of mixed.hip hands every family a contiguous range of its workgroups)
// HAND: the launch may start while the launch of the step before it is still running (xv_anymdp_step_many, overlap mode:
//     consecutive steps alternate between two HIP streams with no dependency between the streams).  Everything that does not
//     depend on the env records is done first (action and reset-record loads, the transition uniform); then the wave waits
//     until its envs' records carry this launch's tag — bits 18..31 of the record's first word = the low 14 bits of the tick
//     of the step that may take the env next: the step before stores record and tag with ONE 8-byte agent-scope store, so
//     the record IS the hand-off (a separate word costs a second round trip: 4.45 vs 3.95 us per step).  Every lane polls
//     its own record and the wave goes on when all 64 tags are there (the store of a wave is four cache lines that may
//     land apart).  A wave depends on the same wave of the previous step only (one lane per env, anymdp_env.py:92-132 is
//     per env).  Only valid lanes count: a wave without any (n_env not a multiple of 256) would poll the last env's record,
//     which its owner re-tags for the NEXT step — a late look would never see this step's tag.  The wait is bounded
//     (xv_hand.h: polls AND wall clock): on expiry the wave goes on and sets XV_DEVERR_HANDOFF — wrong results, flagged,
//     never a hang.
template <bool INJECT, int G, bool ROLLOUT, bool TICKDEV = false, int BK = 0, bool HAND = false>
__device__ __forceinline__ void anymdp_step_body(const AnyMDPArgs& P, const AnyMDPStepIO& io, int T_steps, int mode, int bid) {
  constexpr bool FAST = G > 0;
  constexpr int GG = G > 0 ? G : 1;
  const int i = bid * blockDim.x + threadIdx.x;
  const bool valid = i < P.n_env;
  const int ic = valid ? i : P.n_env - 1;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 3, j = lane & 7;   // FAST: lanes 8g..8g+7 read unit j of a line together
  const int S = P.S, A = P.A;

  // ---- link 1: per-env words (coalesced): the 8-byte env record, the action and (fast path) three 16-byte reset units ----
  uint2 sr0 = make_uint2(0u, 0u);
  if (!HAND) sr0 = P.sr[ic];
  int a_next = io.action ? io.action[ic] : 0;
  const uint64_t gid = P.gid_base + (uint64_t)ic;
  uint32_t err = 0;

  double2 rc01 = make_double2(1.0, 1.0);
  double rc2 = 1.0;
  uint2 rids = make_uint2(0u, 0u);
  uint2 robs = make_uint2(0u, 0u);
  int max_steps, t;
  uint32_t s0_term = 0;
  uint64_t tm0 = 0;
  if (FAST) {
    rc01 = P.rs_a[ic];
    const uint4 rb = P.rs_b[ic], rcu = P.rs_c[ic];
    rc2 = xv_u2d(rb.x, rb.y);
    rids = make_uint2(rb.z, rb.w);
    robs = make_uint2(rcu.x, rcu.y);
    s0_term = (rcu.z >> 27) & 0xFu;
    max_steps = (int)(rcu.z & 0x7FFFFFFu);
    t = (int)rcu.w;
  } else {
    t = P.env_task[ic];
    max_steps = P.max_steps[t];
    tm0 = P.term_mask[(size_t)t * P.words];
  }

  // the launch tick: a kernel argument, or — device tick mode of the engine (xv_engine_set_device_tick) and graph replays
  // of step_many — relative to the engine's / the graph's tick word in device memory
  const uint64_t tick0 = TICKDEV ? *P.tick_dev + P.tick : xv_launch_tick(P.tick, P.tick_dev);
  xv_u32x4 w_pre{0u, 0u, 0u, 0u};
  if (HAND) {
    w_pre = xv_env_draw(P.seed, gid, tick0, XV_DRAW_STEP);      // the transition uniform does not need the env record
    asm volatile("" : "+v"(w_pre.x), "+v"(w_pre.y), "+v"(w_pre.z), "+v"(w_pre.w));      // made here, in front of the wait
    const uint32_t want = XV_ANYMDP_SR_TAG(tick0);
    const uint64_t* rp = reinterpret_cast<const uint64_t*>(P.sr) + ic;
    const uint64_t t_begin = wall_clock64();
    uint64_t r64 = 0;
    // records cross between the launches as agent-scope atomics (sc1: coherent over the device without cache maintenance;
    // an agent-scope release / acquire FENCE is a `buffer_wbl2` per wave: 39 us per step, profiles/r05_b_*).
    // Every lane polls its own record: one round trip per try (lane 0 first and then the wave costs a second one: 3.95 vs
    // 3.76 us per step; a second poll in flight half a period behind the first: 3.87 vs 3.83, profiles/r05_o_*); 1,024 waves x
    // 512 B per ~0.7 us is a tenth of the L2's bandwidth.
    for (uint32_t polls = 0;; ++polls) {
      r64 = __hip_atomic_load(rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__ballot(valid && ((uint32_t)r64 >> XV_ANYMDP_SR_TAG_SHIFT) != want) == 0ull) break;
      __builtin_amdgcn_s_sleep(1);
      if (xv_hand_expired(polls, t_begin) || xv_hand_aborted(polls, P.err)) {      // (the call is then replayed: anymdp_replay_kernel)
        err |= XV_DEVERR_HANDOFF;
        break;
      }
    }
    sr0 = make_uint2((uint32_t)r64, (uint32_t)(r64 >> 32));
  }
  int s = (int)(sr0.x & 0xFFFFu);
  int steps = (int)sr0.y;
  int nr = (sr0.x & XV_ANYMDP_SR_NR) ? 1 : 0;
  int cterm = (sr0.x & XV_ANYMDP_SR_TERM) ? 1 : 0;
  const int T = ROLLOUT ? T_steps : 1;   // single step: straight-line code, counted vmcnt waits
  for (int ts = 0; ts < T; ++ts) {
    const size_t o = (size_t)ts * P.n_env + ic;

    // the transition uniform first: pure ALU under the latency of link 1, and the only draw the table address waits for.
    // The reward normal and the restart uniform (a second Philox call) are made by late_draws() AFTER the step's table
    // line has been requested, under its latency — one wave per SIMD runs here, so every instruction in front of that
    // request is exposed (round 3: the compiler had placed both calls in front of it)
    double u, u_reset = 0.0;
    float z = 0.0f;
    xv_u32x4 w{0u, 0u, 0u, 0u};
    if (INJECT) {
      u = io.u[o];
    } else {
      w = HAND ? w_pre : xv_env_draw(P.seed, gid, tick0 + (uint64_t)ts, XV_DRAW_STEP);
      u = xv_u53(w.x, w.y);
    }
    auto late_draws = [&]() {
      if (INJECT) {
        z = io.z[o];
        u_reset = io.u_reset[o];
      } else {
        z = xv_normal1(w.z, w.w);
        const xv_u32x4 v = xv_env_draw(P.seed, gid, tick0 + (uint64_t)ts, XV_DRAW_RESET);
        u_reset = xv_u53(v.x, v.y);
        // pinned where late_draws() is called: without this hipcc's IR-level sinking moves both chains down to their
        // first use, behind the whole search, where nothing hides them
        int ulo = __double2loint(u_reset), uhi = __double2hiint(u_reset);
        asm volatile("" : "+v"(z), "+v"(ulo), "+v"(uhi));
        u_reset = __hiloint2double(uhi, ulo);
      }
    };

    int a = a_next;
    if (ROLLOUT && io.greedy) {   // teacher policy: argmax_a Q[inner_state] (anymdp_solver_opt.py:38-51), epsilon-greedy
      a = io.greedy[(size_t)t * S + s];
      if (io.epsilon > 0.0f) {
        const xv_u32x4 e = xv_env_draw(P.seed, gid, tick0 + (uint64_t)ts, 2u);
        if ((float)(e.x >> 8) * (1.0f / 16777216.0f) < io.epsilon) a = (int)(e.y % (uint32_t)A);
      }
      if (valid) io.action_out[o] = a;
    }
    if (a < 0 || a >= A) {  // reference: assert action < self.na (:97)
      if (!(mode == XV_AUTORESET_NEXT_STEP && nr)) err |= XV_DEVERR_ACTION_RANGE;
      a = a < 0 ? 0 : A - 1;
    }
    const uint32_t rowidx = ((uint32_t)t * S + s) * A + a;

    // ---- s' = upper_bound(cdf[s,a,:], u)   (:99-100, numpy.random.choice), its reward pair (:103-104),
    //      observation id (:146-148) and terminal flag (:107-108) ----
    int s2, obs2;
    bool term2;
    float2 rsv;
    bool need_fence = true;
    if (BK) {
      // the ONE line: bucket floor(u * NBK) of the row, read by the lanes of the env's group (AnyMDPCoopLineN)
      const uint32_t bk = rowidx * (uint32_t)P.NBK + (uint32_t)(int)(u * (double)P.NBK);
      AnyMDPStepLine L;
      L.issue(P.bucket, bk, lane);
      if (ROLLOUT && io.action && ts + 1 < T) a_next = io.action[o + P.n_env];   // prefetch behind the line
      __builtin_amdgcn_sched_barrier(0);
      late_draws();
      __builtin_amdgcn_sched_barrier(0);
      L.send(u, lane);
      __builtin_amdgcn_sched_barrier(0);
      bool beyond_own;
      float rx, ry;
      uint32_t meta_own;
      L.template resolve_entry<BK>(u, lane, beyond_own, rx, ry, meta_own);
      s2 = (int)((meta_own >> 16) & 0x1FFu);
      rsv = make_float2(rx, ry);
      obs2 = (int)(meta_own & 0xFFFFu);
      term2 = (meta_own >> 25) & 1u;
      // wave-uniform: some env's draw lies beyond the last cut of its line or in a group that lumps several states
      need_fence = __ballot(beyond_own) != 0ull;
    }
    if (FAST && need_fence) {
      const uint32_t fl = rowidx * (uint32_t)P.RL;   // fence line of the row
      // link 2: fence lines.  Iteration `it` serves envs 8*it .. 8*it+7: lanes 8q..8q+7 read the line of env 8*it+q.
      uint32_t li[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) li[it] = (uint32_t)__shfl((int)fl, it * 8 + g);
      uint4 fv[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) fv[it] = P.lines[(size_t)li[it] * 8 + j];
      __builtin_amdgcn_sched_barrier(0);
      if (!BK) {
        late_draws();
        __builtin_amdgcn_sched_barrier(0);
      }
      // while the loads fly: each env's uniform goes to the 8 lanes that hold its lines
      double ue[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) ue[it] = xv_shfl_f64(u, it * 8 + g);
      __builtin_amdgcn_sched_barrier(0);
      // k = #{fence <= u}: two compares per lane, two ballots, the owner (lane 8*it+q) counts byte q
      int k_own = 0;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const unsigned long long m0 = __ballot(xv_u2d(fv[it].x, fv[it].y) <= ue[it]);
        const unsigned long long m1 = __ballot(xv_u2d(fv[it].z, fv[it].w) <= ue[it]);
        const int cnt = __popc((unsigned)(m0 >> (8 * j)) & 0xFFu) + __popc((unsigned)(m1 >> (8 * j)) & 0xFFu);
        k_own = g == it ? cnt : k_own;
      }
      const int NF = P.NB / GG;
      k_own = k_own < NF - 1 ? k_own : NF - 1;       // fences of absent groups hold 2.0: cannot exceed
      // link 3: the GG block lines the fence entry names.  G <= 3 (S <= 336): all 8 x GG lines of the wave in flight together;
      // G = 4, 5 (S <= 512): two batches of 4 env groups, so that the line registers stay at 80
      const uint32_t bl = fl + 1u + (uint32_t)k_own * GG;
      const bool pf = !BK;   // in bucket mode the next action was requested behind the bucket line already
      constexpr int IB = GG > 3 ? 4 : 8;
      int cnt_own = 0;
      float rx = 0.0f, ry = 0.0f;
      uint32_t meta_own = 0;
#pragma unroll
      for (int b0 = 0; b0 < 8; b0 += IB) {
        uint32_t lb[IB];
#pragma unroll
        for (int it = 0; it < IB; ++it) lb[it] = (uint32_t)__shfl((int)bl, (b0 + it) * 8 + g);
        uint4 bv[IB][GG];
#pragma unroll
        for (int it = 0; it < IB; ++it)
#pragma unroll
          for (int q = 0; q < GG; ++q) bv[it][q] = P.lines[((size_t)lb[it] + q) * 8 + j];
        if (b0 == 0 && pf && ROLLOUT && io.action && ts + 1 < T) a_next = io.action[o + P.n_env];   // prefetch behind the blocks
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < IB; ++it) {
          // unit 7 of a block is its metadata, not an entry.  cg: the count of the env this lane group serves
          // (reader side); co: the count of the env this lane owns, whose lines sit in lanes 8q..8q+7 (owner side)
          int cg = 0, co = 0;
#pragma unroll
          for (int q = 0; q < GG; ++q) {
            const unsigned long long m = __ballot(j < 7 && xv_u2d(bv[it][q].x, bv[it][q].y) <= ue[b0 + it]);
            cg += __popc((unsigned)(m >> (8 * g)) & 0x7Fu);
            co += __popc((unsigned)(m >> (8 * j)) & 0x7Fu);
          }
          // the metadata lane extracts the chosen entry's observation id and terminal flag: the owner needs ONE word
          int lg = cg / XV_ANYMDP_BLK;
          lg = lg < GG - 1 ? lg : GG - 1;
          int sg = cg - XV_ANYMDP_BLK * lg;
          sg = sg < 6 ? sg : 6;
          uint32_t packed = 0;
#pragma unroll
          for (int q = 0; q < GG; ++q) {
            const uint4 b4 = bv[it][q];
            const uint32_t mw = sg < 2 ? b4.x : (sg < 4 ? b4.y : (sg < 6 ? b4.z : b4.w));
            const uint32_t pk = ((mw >> (16 * (sg & 1))) & 0xFFFFu) | (((b4.w >> (16 + sg)) & 1u) << 16);
            if (q == lg) packed = pk;
          }
          int lo_ = co / XV_ANYMDP_BLK;
          lo_ = lo_ < GG - 1 ? lo_ : GG - 1;
          int so = co - XV_ANYMDP_BLK * lo_;
          so = so < 6 ? so : 6;
          float px = 0.0f, py = 0.0f;
#pragma unroll
          for (int q = 0; q < GG; ++q) {
            const float x = __shfl(__uint_as_float(bv[it][q].z), 8 * j + so);
            const float y = __shfl(__uint_as_float(bv[it][q].w), 8 * j + so);
            if (q == lo_) { px = x; py = y; }
          }
          const uint32_t pm = (uint32_t)__shfl((int)packed, 8 * j + 7);
          { const bool own = g == b0 + it; cnt_own = own ? co : cnt_own; rx = own ? px : rx; ry = own ? py : ry; meta_own = own ? pm : meta_own; }
        }
      }
      k_own *= GG;
      s2 = XV_ANYMDP_BLK * k_own + cnt_own;
      s2 = s2 < S - 1 ? s2 : S - 1;
      rsv = make_float2(rx, ry);
      obs2 = (int)(meta_own & 0xFFFFu);
      term2 = (meta_own >> 16) & 1u;
    } else if (!FAST) {
      late_draws();
      int lo = 0, n = S;
      while (n > 0) {
        const int half = n >> 1;
        if (anymdp_cdf(P, rowidx, lo + half) <= u) {
          lo += half + 1;
          n -= half + 1;
        } else {
          n = half;
        }
      }
      s2 = lo < S - 1 ? lo : S - 1;
      if (ROLLOUT && io.action && ts + 1 < T) a_next = io.action[o + P.n_env];
      rsv = anymdp_rs(P, rowidx, s2);
      obs2 = P.state_map[(size_t)t * S + s2];
      term2 = anymdp_is_term(P, t, tm0, s2);
    }

    int o_obs, o_fobs = -1;
    float o_r, o_rgt;
    bool o_term, o_trunc;
    bool do_reset = false;
    if (mode == XV_AUTORESET_NEXT_STEP && nr) {
      // the call after a done ignores the action and returns the reset observation
      do_reset = true;
      o_r = 0.0f; o_rgt = 0.0f; o_term = false; o_trunc = false; o_obs = 0;
    } else if (mode == XV_AUTORESET_DISABLED && cterm) {
      // reference raises "given an terminated state" (:95-96): env untouched, error bit set
      err |= XV_DEVERR_STEP_TERMINAL;
      o_obs = P.state_map[(size_t)t * S + s];
      o_r = 0.0f; o_rgt = 0.0f; o_term = true; o_trunc = steps >= max_steps;
    } else {
      steps += 1;                                              // :113
      o_trunc = steps >= max_steps;                            // :114
      o_rgt = rsv.x;
      o_r = fmaf(rsv.y, z, rsv.x);                             // :105 normal(mu, sigma) = mu + sigma*z
      o_term = term2;
      s = s2;
      cterm = term2 ? 1 : 0;
      o_obs = obs2;
      if (o_term || o_trunc) {
        if (mode == XV_AUTORESET_SAME_STEP) {
          o_fobs = obs2;
          do_reset = true;
        } else if (mode == XV_AUTORESET_NEXT_STEP) {
          nr = 1;
        }
      }
    }
    if (do_reset) {                                            // reset(): :85-90
      if (FAST) {
        // upper_bound over the 4 padded CDF entries; ids, terminal flags and observation ids come packed
        const int k0 = (int)(rc01.x <= u_reset) + (int)(rc01.y <= u_reset) + (int)(rc2 <= u_reset);
        s = (int)(((k0 < 2 ? rids.x : rids.y) >> (16 * (k0 & 1))) & 0xFFFFu);
        cterm = (int)((s0_term >> k0) & 1u);
        o_obs = (int)(((k0 < 2 ? robs.x : robs.y) >> (16 * (k0 & 1))) & 0xFFFFu);
      } else {
        s = anymdp_draw_s0(P, t, u_reset);
        cterm = anymdp_is_term(P, t, tm0, s) ? 1 : 0;
        o_obs = P.state_map[(size_t)t * S + s];
      }
      steps = 0;
      nr = 0;
    }
    if (HAND) {
      // hand the envs on BEFORE the outputs leave: record and tag of the next step in one 8-byte agent-scope store
      const uint2 q = anymdp_sr_pack(s, steps, nr, cterm);
      const uint32_t x = q.x | (XV_ANYMDP_SR_TAG(tick0 + 1u) << XV_ANYMDP_SR_TAG_SHIFT);
      if (valid) __hip_atomic_store(reinterpret_cast<uint64_t*>(P.sr) + i, (uint64_t)x | ((uint64_t)q.y << 32), __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
    }
    if (valid) {
#if XV_ANYMDP_NT_OUT
      __builtin_nontemporal_store(o_obs, io.obs + o);
      __builtin_nontemporal_store(o_r, io.reward + o);
      __builtin_nontemporal_store(o_rgt, io.reward_gt + o);
      __builtin_nontemporal_store((uint8_t)(o_term ? 1 : 0), io.terminated + o);
      __builtin_nontemporal_store((uint8_t)(o_trunc ? 1 : 0), io.truncated + o);
      if (io.final_obs) __builtin_nontemporal_store(o_fobs, io.final_obs + o);
      if (io.steps_out) __builtin_nontemporal_store((int32_t)steps, io.steps_out + o);
      if (io.done_out) __builtin_nontemporal_store((uint8_t)((o_term || o_trunc) ? 1 : 0), io.done_out + o);
#else
      io.obs[o] = o_obs;
      io.reward[o] = o_r;
      io.reward_gt[o] = o_rgt;
      io.terminated[o] = o_term ? 1 : 0;
      io.truncated[o] = o_trunc ? 1 : 0;
      if (io.final_obs) io.final_obs[o] = o_fobs;
      if (io.steps_out) io.steps_out[o] = (int32_t)steps;
      if (io.done_out) io.done_out[o] = (uint8_t)((o_term || o_trunc) ? 1 : 0);
#endif
    }
  }
  if (!HAND && valid) P.sr[i] = anymdp_sr_pack(s, steps, nr, cterm);
  if (err) atomicOr(P.err, err);
}

template <bool INJECT, int G, bool ROLLOUT, bool TICKDEV = false, int BK = 0, bool HAND = false>
__global__ __launch_bounds__(256) void anymdp_step_kernel(AnyMDPArgs P, AnyMDPStepIO io, int T_steps, int mode) {
  anymdp_step_body<INJECT, G, ROLLOUT, TICKDEV, BK, HAND>(P, io, T_steps, mode, (int)blockIdx.x);
}
// An expired hand-off (XV_DEVERR_HANDOFF set during an overlapped call: a wave went on with a record nobody had handed it)
// is REPAIRED: the call's opening kernel kept every env record and the error word as they were at entry (snap, w[0]); this
// kernel runs on the engine's stream behind the join of every overlapped call.  All of its workgroups read the same two
// words; in the usual case (no new HANDOFF bit) they return at once — one nearly empty launch per CALL.  Otherwise every
// lane restores its env's record and replays the whole call — `cycles` ring cycles of `period` steps — as the fused roll-out
// does: same ticks, same draws, every ring slot rewritten, bit-equal to the one-stream path (tests/test_gpu_chains.py).
// Error bits the failed attempt raised from wrong states are dropped: the word becomes entry | what the replay raised.
// The last workgroup to finish publishes that and counts the replay in pinned host memory (h_fell:
// xv_anymdp_step_many_overlap_state -> -2).
template <int G, int BK>
__global__ __launch_bounds__(256) void anymdp_replay_kernel(AnyMDPArgs P, AnyMDPStepIO io, int period, int cycles, int mode,
                                                            const uint2* snap, uint32_t* w, uint32_t* real_err, uint32_t* h_fell) {
  const uint32_t e_now = __hip_atomic_load(real_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t e_in = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((e_now & XV_DEVERR_HANDOFF) == 0u) {      // (the opening kernel took the bit out of the word: set = raised by this call)
    if (blockIdx.x == 0 && threadIdx.x == 0 && (e_in & XV_DEVERR_HANDOFF)) atomicOr(real_err, (uint32_t)XV_DEVERR_HANDOFF);
    return;
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < P.n_env) P.sr[i] = snap[i];
  __syncthreads();      // (an invalid lane looks at the last env's record: restored by a lane of this workgroup)
  AnyMDPArgs Q = P;
  Q.err = w + 1;
  Q.tick_dev = nullptr;
  for (int c = 0; c < cycles; ++c) {
    Q.tick = P.tick + (uint64_t)c * (uint64_t)period;
    anymdp_step_body<false, G, true, false, BK, false>(Q, io, period, mode, (int)blockIdx.x);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(w + 2, 1u) == gridDim.x - 1u) {
      const uint32_t re = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(real_err, e_in | re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      w[1] = 0u; w[2] = 0u;
      __hip_atomic_fetch_add(h_fell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// test hook (XV_PIPE_TEST_FAIL=1, tests/test_gpu_chains.py): what an expired hand-off leaves behind — the flag, records and
// ring contents that are wrong — placed between the join and the replay kernel
static __global__ __launch_bounds__(256) void anymdp_test_fail_kernel(uint2* sr, int n, uint32_t* err, int32_t* obs, size_t n_obs) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)n) sr[i] = make_uint2((uint32_t)(i % 3u), 5u);
  for (size_t k = i; k < n_obs; k += (size_t)gridDim.x * blockDim.x) obs[k] = -7;
  if (i == 0) atomicOr(err, (uint32_t)(XV_DEVERR_HANDOFF | XV_DEVERR_STEP_TERMINAL));
}

// Completes the rows in place (create time): fence line and per-block metadata.  One thread per (row, k < 64).
//   fence[k] (k < 16) = CDF entry of the last next-state of block group k (G blocks) for k < NB/G - 1, else 2.0
//   meta of block k (k < NB) = {u16 obs[7]; u8 term_bits; u8 0} of next states 7k..7k+6 (clamped to S-1, as s' is)
static __global__ __launch_bounds__(256) void anymdp_finish_rows_kernel(AnyMDPArgs P, uint4* lines_rw, size_t row_base,
                                                                 size_t n_rows) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * 64) return;
  const size_t r = row_base + (idx >> 6);
  const int k = (int)(idx & 63);
  const int t = (int)(r / ((size_t)P.S * P.A));
  uint4* row = lines_rw + r * (size_t)P.RL * 8;
  if (k < 16) {
    double f = 2.0;
    if (k < P.NB / P.G - 1)
      f = reinterpret_cast<const double*>(row + (size_t)(1 + P.G * (k + 1) - 1) * 8 + (XV_ANYMDP_BLK - 1))[0];
    reinterpret_cast<double*>(row)[k] = f;
  }
  if (k < P.NB) {
    uint32_t w[4] = {0, 0, 0, 0};
    uint32_t tb = 0;
    for (int e = 0; e < XV_ANYMDP_BLK; ++e) {
      int sn = XV_ANYMDP_BLK * k + e;
      sn = sn < P.S - 1 ? sn : P.S - 1;
      const uint32_t ob = (uint32_t)P.state_map[(size_t)t * P.S + sn] & 0xFFFFu;
      w[e >> 1] |= ob << (16 * (e & 1));
      if ((P.term_mask[(size_t)t * P.words + (sn >> 6)] >> (sn & 63)) & 1ull) tb |= 1u << e;
    }
    w[3] |= tb << 16;
    row[(size_t)(1 + k) * 8 + 7] = make_uint4(w[0], w[1], w[2], w[3]);
    // padding entries (next states >= S): cdf 2.0 and the reward pair of next state S-1 — the pair a clamped s' must
    // get when u >= cdf[S-1] (a caller-supplied row whose last CDF entry stays below 1), as the per-lane search reads it
    if (XV_ANYMDP_BLK * (k + 1) > P.S) {
      const int jl = P.S - 1;
      const uint4 last = row[(size_t)(1 + jl / XV_ANYMDP_BLK) * 8 + (jl % XV_ANYMDP_BLK)];
      for (int e = 0; e < XV_ANYMDP_BLK; ++e)
        if (XV_ANYMDP_BLK * k + e >= P.S)
          row[(size_t)(1 + k) * 8 + e] = make_uint4(0u, 0x40000000u, last.z, last.w);
    }
  }
}

// Bucket lines (xv_anymdp_build_buckets): one thread per (row, bucket) chooses the line's cuts (anymdp_cutline.h) and writes
// its 8 units; `bucket` == nullptr: census only (xv_anymdp_probe_buckets).  census[0] += lines with draws they cannot
// answer, census[1] += their probability mass (fixed point, 2^-36, rounded up per wave), census[2] += live rows (rows of
// terminal states are never drawn from and count nowhere).
template <int FMT>
static __global__ __launch_bounds__(256) void anymdp_build_cutlines_kernel(AnyMDPArgs P, uint4* bucket, size_t row_base,
                                                                          size_t n_rows, int NBK, unsigned long long* census) {
  constexpr int KC = FMT == 2 ? 6 : 7;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double mass = 0.0;
  bool live_row = false;
  if (idx < n_rows * (size_t)NBK) {
    const size_t r = row_base + idx / (size_t)NBK;
    const int k = (int)(idx % (size_t)NBK);
    const int t = (int)(r / ((size_t)P.S * P.A)), s_cur = (int)((r / (size_t)P.A) % (size_t)P.S);
    const uint4* row = P.lines + r * (size_t)P.RL * 8;
    auto unit = [row](int jn) { return row + (size_t)(1 + jn / XV_ANYMDP_BLK) * 8 + (jn % XV_ANYMDP_BLK); };
    auto cdf = [&unit](int jn) { return reinterpret_cast<const double*>(unit(jn))[0]; };
    XvCutLine L;
    xv_cutline_build(cdf, P.S, (double)k / (double)NBK, (double)(k + 1) / (double)NBK, KC, L);
    const bool term_row = (P.term_mask[(size_t)t * P.words + (s_cur >> 6)] >> (s_cur & 63)) & 1ull;
    if (!term_row) mass = L.dirty_mass;
    live_row = !term_row && k == 0;
    if (bucket) {
      uint4* out = bucket + (r * (size_t)NBK + k) * 8;
      uint32_t ob[7], st[7], tb = 0;
#pragma unroll
      for (int c = 0; c < 7; ++c) {
        ob[c] = 0; st[c] = 0;
        if (c < KC) {
          const int sn = L.state[c];
          const uint4 e = *unit(L.entry[c]);
          out[c] = make_uint4((uint32_t)__double2loint(L.cut[c]), (uint32_t)__double2hiint(L.cut[c]), e.z, e.w);
          ob[c] = (uint32_t)P.state_map[(size_t)t * P.S + sn];
          st[c] = (uint32_t)sn;
          if ((P.term_mask[(size_t)t * P.words + (sn >> 6)] >> (sn & 63)) & 1ull) tb |= 1u << c;
        }
      }
      if (FMT == 1) {
        out[7] = make_uint4((ob[0] & 0xFFu) | ((ob[1] & 0xFFu) << 8) | ((ob[2] & 0xFFu) << 16) | ((ob[3] & 0xFFu) << 24),
                            (ob[4] & 0xFFu) | ((ob[5] & 0xFFu) << 8) | ((ob[6] & 0xFFu) << 16) | (tb << 24),
                            (st[0] & 0xFFu) | ((st[1] & 0xFFu) << 8) | ((st[2] & 0xFFu) << 16) | ((st[3] & 0xFFu) << 24),
                            (st[4] & 0xFFu) | ((st[5] & 0xFFu) << 8) | ((st[6] & 0xFFu) << 16) | ((L.dirty & 0x7Fu) << 24));
      } else {
        out[6] = make_uint4((st[0] & 0xFFFFu) | (st[1] << 16), (st[2] & 0xFFFFu) | (st[3] << 16), (st[4] & 0xFFFFu) | (st[5] << 16), 0u);
        out[7] = make_uint4((ob[0] & 0xFFFFu) | (ob[1] << 16), (ob[2] & 0xFFFFu) | (ob[3] << 16), (ob[4] & 0xFFFFu) | (ob[5] << 16),
                            (tb & 0x3Fu) | ((L.dirty & 0x3Fu) << 8));
      }
    }
  }
  const unsigned long long dm = __ballot(mass > 0.0), lm = __ballot(live_row);
  double sum = mass;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
  if ((threadIdx.x & 63) == 0) {
    if (dm) {
      atomicAdd(census + 0, (unsigned long long)__popcll(dm));
      atomicAdd(census + 1, (unsigned long long)ceil(sum * 68719476736.0));
    }
    if (lm) atomicAdd(census + 2, (unsigned long long)__popcll(lm));
  }
}

// per-env reset records (create time): the s_0 distribution of the env's task, ready for coalesced 16-byte reads
static __global__ __launch_bounds__(256) void anymdp_env_records_kernel(AnyMDPArgs P, double2* ra, uint4* rb, uint4* rc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  const int t = P.env_task[i];
  double c[4];
  uint32_t tbits = 0, id[4], ob[4];
  for (int k = 0; k < 4; ++k) {
    const int kk = k < P.s0_max ? k : P.s0_max - 1;
    const int sid = P.s0_ids[(size_t)t * P.s0_max + kk];
    const uint32_t tb = (P.term_mask[(size_t)t * P.words + (sid >> 6)] >> (sid & 63)) & 1ull ? 1u : 0u;
    id[k] = (uint32_t)sid & 0xFFFFu;
    tbits |= tb << k;
    ob[k] = (uint32_t)P.state_map[(size_t)t * P.S + sid] & 0xFFFFu;
    c[k] = k < P.s0_max ? P.s0_cdf[(size_t)t * P.s0_max + k] : 1.0;
  }
  ra[i] = make_double2(c[0], c[1]);
  rb[i] = make_uint4((uint32_t)__double2loint(c[2]), (uint32_t)__double2hiint(c[2]), id[0] | (id[1] << 16), id[2] | (id[3] << 16));
  rc[i] = make_uint4(ob[0] | (ob[1] << 16), ob[2] | (ob[3] << 16), ((uint32_t)P.max_steps[t] & 0x7FFFFFFu) | (tbits << 27),
                     (uint32_t)t);
}

// xv_anymdp_get_state / _set_state: the 8-byte env records <-> the caller's arrays (nullable each); the terminal flag of
// a state set from outside is recomputed from term_mask
static __global__ __launch_bounds__(256) void anymdp_get_state_kernel(AnyMDPArgs P, int32_t* state, int32_t* steps, uint8_t* need_reset) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  const uint2 r = P.sr[i];
  if (state) state[i] = (int32_t)(r.x & 0xFFFFu);
  if (steps) steps[i] = (int32_t)r.y;
  if (need_reset) need_reset[i] = (r.x & XV_ANYMDP_SR_NR) ? 1 : 0;
}
static __global__ __launch_bounds__(256) void anymdp_set_state_kernel(AnyMDPArgs P, const int32_t* state, const int32_t* steps,
                                                               const uint8_t* need_reset) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  const uint2 r = P.sr[i];
  int s = state ? state[i] : (int)(r.x & 0xFFFFu);
  s = s < 0 ? 0 : (s >= P.S ? P.S - 1 : s);
  const int t = P.env_task[i];
  const int cterm = state ? (int)((P.term_mask[(size_t)t * P.words + (s >> 6)] >> (s & 63)) & 1ull) : ((r.x & XV_ANYMDP_SR_TERM) ? 1 : 0);
  const int nr = need_reset ? (need_reset[i] ? 1 : 0) : ((r.x & XV_ANYMDP_SR_NR) ? 1 : 0);
  P.sr[i] = anymdp_sr_pack(s, steps ? steps[i] : (int)r.y, nr, cterm);
}

// largest observation id (decides whether ids fit the 16-bit block metadata)
static __global__ __launch_bounds__(256) void anymdp_max_obs_kernel(const int32_t* state_map, size_t n, int* out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) atomicMax(out, state_map[idx]);
}

template <bool INJECT>
__global__ __launch_bounds__(256) void anymdp_reset_kernel(AnyMDPArgs P, const uint8_t* mask,
                                                           const double* u_in, int32_t* obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  if (mask && !mask[i]) return;
  const int t = P.env_task[i];
  double u;
  if (INJECT) {
    u = u_in[i];
  } else {
    const xv_u32x4 v = xv_env_draw(P.seed, P.gid_base + (uint64_t)i, xv_launch_tick(P.tick, P.tick_dev), XV_DRAW_RESET);
    u = xv_u53(v.x, v.y);
  }
  const int s = anymdp_draw_s0(P, t, u);
  P.sr[i] = anymdp_sr_pack(s, 0, 0, (int)((P.term_mask[(size_t)t * P.words + (s >> 6)] >> (s & 63)) & 1ull));
  if (obs) obs[i] = P.state_map[(size_t)t * P.S + s];
}

// info["transition_gt"] = transition_obs[self.state, action]   (anymdp_env.py:130, :12-20)
static __global__ __launch_bounds__(256) void anymdp_tgt_kernel(AnyMDPArgs P, const int32_t* action, double* out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)P.n_env * P.S;
  if (idx >= total) return;
  const int i = (int)(idx / P.S), j = (int)(idx % P.S);
  const int t = P.env_task[i], s = (int)(P.sr[i].x & 0xFFFFu);
  int a = action[i];
  a = a < 0 ? 0 : (a >= P.A ? P.A - 1 : a);
  const uint64_t tm0 = P.term_mask[(size_t)t * P.words];
  const uint32_t r = ((uint32_t)t * P.S + s) * P.A + a;
  double v = 0.0;
  if (!anymdp_is_term(P, t, tm0, s)) v = anymdp_cdf(P, r, j) - (j ? anymdp_cdf(P, r, j - 1) : 0.0);
  out[(size_t)i * P.S + P.state_map[(size_t)t * P.S + j]] = v;
}

// ------------------------------------------------------------------------------------------------
// POMDP / multi-token POMDP (anymdp_env.py:116-128 token loop, :148-157 observation draws).  One lane per env,
// per-lane binary searches: d_act transition draws, then d_obs observation draws from obs_cdf[t][k][s][:].
// Draw order and Philox purposes as oracle/xeno_oracle.c (tok_*).
// ------------------------------------------------------------------------------------------------
struct AnyMDPTokArgs {
  const double* obs_cdf;   // [n_task][d_obs][S][n_obs]
  int n_obs, d_obs, d_act;
  const uint4* obs_bucket; // [n_task][d_obs][S][NBK] observation bucket lines (engine-owned) or nullptr
};

struct AnyMDPTokIO {
  const int32_t* action;       // [n_env][d_act]
  const double* u;             // [d_act][n_env]   (INJECT)
  const float* z;              // [d_act][n_env]
  const double* u_obs;         // [d_obs][n_env]
  const double* u_reset;       // [n_env]
  const double* u_obs_reset;   // [d_obs][n_env]
  int32_t* obs;                // [n_env][d_obs]
  float* reward;
  float* reward_gt;
  uint8_t* terminated;
  uint8_t* truncated;
  int32_t* final_obs;          // [n_env][d_obs], nullable
  int32_t* steps_out;          // [n_env] info["steps"] after the step and the terminated | truncated mask of the same step
  uint8_t* done_out;           //         (xv_anymdp_step_tokens_info; each nullable)
};

__device__ __forceinline__ int xv_upper_bound_f64(const double* row, int n, double u) {
  int lo = 0, m = n;
  while (m > 0) {
    const int half = m >> 1;
    if (row[lo + half] <= u) { lo += half + 1; m -= half + 1; }
    else m = half;
  }
  return lo < n - 1 ? lo : n - 1;
}

// observation tokens of inner state s (after a step: RESET=false, after a reset: RESET=true)
template <bool INJECT, bool RESET>
__device__ __forceinline__ void anymdp_tok_observe(const AnyMDPArgs& P, const AnyMDPTokArgs& K, const AnyMDPTokIO& io,
                                                   int i, int t, int s, uint64_t gid, int32_t* out) {
  const uint64_t tick_now = xv_launch_tick(P.tick, P.tick_dev);
  for (int k = 0; k < K.d_obs; ++k) {
    double u;
    if (INJECT) {
      u = (RESET ? io.u_obs_reset : io.u_obs)[(size_t)k * P.n_env + i];
    } else {
      const xv_u32x4 w = xv_env_draw(P.seed, gid, tick_now, 64u + (uint32_t)k);
      u = RESET ? xv_u53(w.z, w.w) : xv_u53(w.x, w.y);
    }
    const double* row = K.obs_cdf + ((((size_t)t * K.d_obs + k) * P.S) + s) * (size_t)K.n_obs;
    out[(size_t)i * K.d_obs + k] = xv_upper_bound_f64(row, K.n_obs, u);
  }
}

template <bool INJECT>
__global__ __launch_bounds__(256) void anymdp_tok_step_kernel(AnyMDPArgs P, AnyMDPTokArgs K, AnyMDPTokIO io, int mode) {
  const uint64_t tick_now = xv_launch_tick(P.tick, P.tick_dev);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  const int S = P.S, A = P.A, N = P.n_env;
  const int t = P.env_task[i];
  const uint64_t gid = P.gid_base + (uint64_t)i;
  const uint64_t tm0 = P.term_mask[(size_t)t * P.words];
  const int max_steps = P.max_steps[t];
  const uint2 sr0 = P.sr[i];
  int s = (int)(sr0.x & 0xFFFFu), steps = (int)sr0.y, nr = (sr0.x & XV_ANYMDP_SR_NR) ? 1 : 0;
  uint32_t err = 0;
  if (io.final_obs) for (int k = 0; k < K.d_obs; ++k) io.final_obs[(size_t)i * K.d_obs + k] = -1;
  float rsum = 0.0f, rgsum = 0.0f;
  int term = 0, trunc = 0;
  bool do_reset = false;
  if (mode == XV_AUTORESET_NEXT_STEP && nr) {
    do_reset = true;
  } else if (mode == XV_AUTORESET_DISABLED && anymdp_is_term(P, t, tm0, s)) {
    err |= XV_DEVERR_STEP_TERMINAL;   // reference raises (:95-96)
    anymdp_tok_observe<INJECT, false>(P, K, io, i, t, s, gid, io.obs);
    term = 1; trunc = steps >= max_steps;
  } else {
    steps += 1;                       // :113, once per step
    trunc = steps >= max_steps;       // :114
    for (int k = 0; k < K.d_act; ++k) {   // :120-126
      int a = io.action[(size_t)i * K.d_act + k];
      if (a < 0 || a >= A) { err |= XV_DEVERR_ACTION_RANGE; a = a < 0 ? 0 : A - 1; }
      double u;
      float z;
      if (INJECT) {
        u = io.u[(size_t)k * N + i];
        z = io.z[(size_t)k * N + i];
      } else {
        const xv_u32x4 w = xv_env_draw(P.seed, gid, tick_now, 32u + (uint32_t)k);
        u = xv_u53(w.x, w.y);
        z = xv_normal1(w.z, w.w);
      }
      const uint32_t rowidx = ((uint32_t)t * S + s) * A + a;
      int lo = -1;
      float2 rsv = make_float2(0.0f, 0.0f);
      {   // per-lane search of the row
        lo = 0;
        int m = S;
        while (m > 0) {
          const int half = m >> 1;
          if (anymdp_cdf(P, rowidx, lo + half) <= u) { lo += half + 1; m -= half + 1; }
          else m = half;
        }
        lo = lo < S - 1 ? lo : S - 1;
        rsv = anymdp_rs(P, rowidx, lo);
      }
      const int s2 = lo < S - 1 ? lo : S - 1;
      rsum = rsum + fmaf(rsv.y, z, rsv.x);
      rgsum = rgsum + rsv.x;
      s = s2;
      if (anymdp_is_term(P, t, tm0, s2)) { term = 1; break; }
    }
    anymdp_tok_observe<INJECT, false>(P, K, io, i, t, s, gid, io.obs);
    if (term || trunc) {
      if (mode == XV_AUTORESET_SAME_STEP) {
        if (io.final_obs)
          for (int k = 0; k < K.d_obs; ++k) io.final_obs[(size_t)i * K.d_obs + k] = io.obs[(size_t)i * K.d_obs + k];
        do_reset = true;
      } else if (mode == XV_AUTORESET_NEXT_STEP) {
        nr = 1;
      }
    }
  }
  if (do_reset) {
    double ur;
    if (INJECT) ur = io.u_reset[i];
    else {
      const xv_u32x4 v = xv_env_draw(P.seed, gid, tick_now, XV_DRAW_RESET);
      ur = xv_u53(v.x, v.y);
    }
    s = anymdp_draw_s0(P, t, ur);
    steps = 0;
    nr = 0;
    anymdp_tok_observe<INJECT, true>(P, K, io, i, t, s, gid, io.obs);
  }
  P.sr[i] = anymdp_sr_pack(s, steps, nr, anymdp_is_term(P, t, tm0, s) ? 1 : 0);
  io.reward[i] = rsum; io.reward_gt[i] = rgsum;
  io.terminated[i] = (uint8_t)term; io.truncated[i] = (uint8_t)trunc;
  if (io.steps_out) io.steps_out[i] = (int32_t)steps;
  if (io.done_out) io.done_out[i] = (uint8_t)((term || trunc) ? 1 : 0);
  if (err) atomicOr(P.err, err);
}

// ------------------------------------------------------------------------------------------------
// Cooperative multi-token step (round 3): the same dependent-level economy as the MDP step kernel.  A transition token is
// ONE bucket line read by the lanes of the env's group (AnyMDPCoopLineN); an observation token is one
// OBSERVATION bucket line — 15 consecutive entries of obs_cdf[t][k][s][:] that start at I = #{cdf <= b / NBK} plus I
// itself (xv_anymdp_build_buckets builds them beside the transition lines) — and the lines of two observation tokens
// are in flight together.  A draw whose line does not contain its answer (all entries <= u) takes the per-lane binary
// search (rare; per-lane branch).  Same draws, same results as anymdp_tok_step_kernel.
// ------------------------------------------------------------------------------------------------
// (the line reader itself, AnyMDPCoopLineN, sits above anymdp_step_body: the MDP step reads its bucket line the same way)

// observation bucket lines: one thread per (row obs_cdf[t][k][s][:], bucket) chooses 14 cuts (anymdp_cutline.h) and writes
// the line; n_obs <= 256 (symbol ids are bytes).  census as in anymdp_build_cutlines_kernel (every row counts as live).
static __global__ __launch_bounds__(256) void anymdp_build_obs_cutlines_kernel(const double* obs_cdf, size_t n_rows, int n_obs, int NBK,
                                                                        uint4* out, unsigned long long* census) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double mass = 0.0;
  bool first = false;
  if (idx < n_rows * (size_t)NBK) {
    const size_t w = idx / (size_t)NBK;
    const int kb = (int)(idx % (size_t)NBK);
    const double* row = obs_cdf + w * (size_t)n_obs;
    auto cdf = [row](int jn) { return row[jn]; };
    XvCutLine L;
    xv_cutline_build(cdf, n_obs, (double)kb / (double)NBK, (double)(kb + 1) / (double)NBK, 14, L);
    mass = L.dirty_mass;
    first = kb == 0;
    uint4* o = out + idx * 8;
    uint32_t idw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int c = 0; c < 14; ++c) idw[c >> 2] |= ((uint32_t)L.state[c] & 0xFFu) << (8 * (c & 3));
    idw[3] |= (L.dirty & 0x3FFFu) << 16;
#pragma unroll
    for (int q = 0; q < 7; ++q)
      o[q] = make_uint4((uint32_t)__double2loint(L.cut[2 * q]), (uint32_t)__double2hiint(L.cut[2 * q]),
                        (uint32_t)__double2loint(L.cut[2 * q + 1]), (uint32_t)__double2hiint(L.cut[2 * q + 1]));
    o[7] = make_uint4(idw[0], idw[1], idw[2], idw[3]);
  }
  const unsigned long long dm = __ballot(mass > 0.0), lm = __ballot(first);
  double sum = mass;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
  if ((threadIdx.x & 63) == 0) {
    if (dm) {
      atomicAdd(census + 0, (unsigned long long)__popcll(dm));
      atomicAdd(census + 1, (unsigned long long)ceil(sum * 68719476736.0));
    }
    if (lm) atomicAdd(census + 2, (unsigned long long)__popcll(lm));
  }
}

// PAIR: d_obs > 1 (two observation lines in flight per round of the observation loop; a single-token POMDP has no second line)
// HAND: overlapped xv_anymdp_step_tokens_many — as the MDP step's HAND (see anymdp_step_body): the wave waits for its envs'
// records to carry this launch's tag, and stores record + next tag in one agent-scope store as soon as the transition tokens
// are through — BEFORE the observation stage, so the next step's transition lines fly under this step's observation lines.
template <bool INJECT, int FMT, bool PAIR, bool HAND = false>
__device__ __forceinline__ void anymdp_tok_step_coop_body(const AnyMDPArgs& P, const AnyMDPTokArgs& K, const AnyMDPTokIO& io, int mode, int bid) {
  const uint64_t tick_now = xv_launch_tick(P.tick, P.tick_dev);
  const int i = bid * blockDim.x + threadIdx.x;
  const bool valid = i < P.n_env;
  const int ic = valid ? i : P.n_env - 1;
  const int lane = threadIdx.x & 63;
  const int S = P.S, A = P.A, N = P.n_env, DO = K.d_obs, NBK = P.NBK;
  uint2 sr0 = make_uint2(0u, 0u);
  if (!HAND) sr0 = P.sr[ic];
  const uint4 rcu = P.rs_c[ic];
  const double2 rc01 = P.rs_a[ic];
  const uint4 rb = P.rs_b[ic];
  int a_cur = io.action[(size_t)ic * K.d_act];
  const int t = (int)rcu.w, max_steps = (int)(rcu.z & 0x7FFFFFFu);
  const uint64_t gid = P.gid_base + (uint64_t)ic;
  uint32_t err = 0;
  xv_u32x4 w_first{0u, 0u, 0u, 0u};
  if (HAND) {
    w_first = xv_env_draw(P.seed, gid, tick_now, 32u);      // the first transition uniform needs no record
    asm volatile("" : "+v"(w_first.x), "+v"(w_first.y), "+v"(w_first.z), "+v"(w_first.w));
    const uint32_t want = XV_ANYMDP_SR_TAG(tick_now);
    const uint64_t* rp = reinterpret_cast<const uint64_t*>(P.sr) + ic;
    const uint64_t t_begin = wall_clock64();
    uint64_t r64 = 0;
    for (uint32_t polls = 0;; ++polls) {
      r64 = __hip_atomic_load(rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__ballot(valid && ((uint32_t)r64 >> XV_ANYMDP_SR_TAG_SHIFT) != want) == 0ull) break;
      __builtin_amdgcn_s_sleep(1);
      if (xv_hand_expired(polls, t_begin) || xv_hand_aborted(polls, P.err)) { err |= XV_DEVERR_HANDOFF; break; }      // (replayed: anymdp_tok_replay_kernel)
    }
    sr0 = make_uint2((uint32_t)r64, (uint32_t)(r64 >> 32));
  }
  int s = (int)(sr0.x & 0xFFFFu), steps = (int)sr0.y, nr = (sr0.x & XV_ANYMDP_SR_NR) ? 1 : 0;
  int cterm = (sr0.x & XV_ANYMDP_SR_TERM) ? 1 : 0;
  if (valid && io.final_obs) for (int k = 0; k < DO; ++k) io.final_obs[(size_t)i * DO + k] = -1;

  // uniforms: transition token k -> purpose 32 + k words (0,1), its reward normal words (2,3); observation token k ->
  // purpose 64 + k, words (0,1) after a step and (2,3) after a restart (oracle: tok_draws)
  auto act_draw = [&](int k, double& u, xv_u32x4& w) {
    if (INJECT) { u = io.u[(size_t)k * N + ic]; }
    else { w = (HAND && k == 0) ? w_first : xv_env_draw(P.seed, gid, tick_now, 32u + (uint32_t)k); u = xv_u53(w.x, w.y); }
  };
  // both uniforms of observation token k from ONE call: after a step (us), after a restart (ur_)
  auto obs_draw = [&](int k, double& us, double& ur_) {
    if (INJECT) {
      us = io.u_obs[(size_t)k * N + ic];
      ur_ = io.u_obs_reset[(size_t)k * N + ic];
    } else {
      const xv_u32x4 w = xv_env_draw(P.seed, gid, tick_now, 64u + (uint32_t)k);
      us = xv_u53(w.x, w.y);
      ur_ = xv_u53(w.z, w.w);
    }
  };
  auto pin2 = [](double& x, double& y) {   // keeps a pair of uniforms where it is computed (see the step kernel's late_draws)
    int a0 = __double2loint(x), a1 = __double2hiint(x), b0 = __double2loint(y), b1 = __double2hiint(y);
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
    x = __hiloint2double(a1, a0);
    y = __hiloint2double(b1, b0);
  };
  auto obs_line = [&](int k, int state, double u) -> uint32_t {
    return ((((uint32_t)t * DO + k) * S + state) * (uint32_t)NBK) + (uint32_t)(int)(u * (double)NBK);
  };
  auto obs_pick = [&](int k, int state, double u, int id, bool beyond, bool want) -> int {
    int ob = id;
    if (want && beyond) {   // the line cannot answer this draw: search the row
      const double* row = K.obs_cdf + ((((size_t)t * DO + k) * S) + state) * (size_t)K.n_obs;
      ob = xv_upper_bound_f64(row, K.n_obs, u);
    }
    return ob;
  };

  const bool skip = mode == XV_AUTORESET_NEXT_STEP && nr;                 // the call after a done: reset only
  const bool stuck = !skip && mode == XV_AUTORESET_DISABLED && cterm;     // reference raises (:95-96): observe only
  const bool active = !skip && !stuck;

  // The restart state depends on the env's records and one uniform only, and no observation uniform depends on the step:
  // the restart draw and the uniforms of the first two observation tokens are made under the latency of the FIRST transition
  // line (early_work, called inside the token loop).  The restart state's observation lines are requested with the step's
  // observation lines, for the envs that do restart only (requesting them for every env up front cost 2 of 6 lines per
  // env-step; the kernel runs at the random-line rate of the HBM system).
  const bool restarts = mode != XV_AUTORESET_DISABLED;   // wave-uniform
  int k0r = 0, s_new = 0;
  double uS0 = 0.0, uS1 = 0.0, uR0 = 0.0, uR1 = 0.0;
  const int kR1 = PAIR && DO > 1 ? 1 : 0;
  auto early_work = [&]() {
    obs_draw(0, uS0, uR0);
    if (PAIR) obs_draw(kR1, uS1, uR1);
    pin2(uS0, uS1);
    if (restarts) {
      double ur;
      if (INJECT) ur = io.u_reset[ic];
      else {
        const xv_u32x4 v = xv_env_draw(P.seed, gid, tick_now, XV_DRAW_RESET);
        ur = xv_u53(v.x, v.y);
      }
      k0r = (int)(rc01.x <= ur) + (int)(rc01.y <= ur) + (int)(xv_u2d(rb.x, rb.y) <= ur);
      s_new = (int)(((k0r < 2 ? rb.z : rb.w) >> (16 * (k0r & 1))) & 0xFFFFu);
      pin2(uR0, uR1);
      asm volatile("" : "+v"(k0r), "+v"(s_new));
    }
  };

  // ---- transition tokens (:120-126): one bucket line each; the reward normal and the next token's uniform are made
  //      under the line's latency ----
  float rsum = 0.0f, rgsum = 0.0f;
  int term = 0, trunc = 0;
  if (stuck) { err |= XV_DEVERR_STEP_TERMINAL; term = 1; trunc = steps >= max_steps; }
  if (active) { steps += 1; trunc = steps >= max_steps; }                // :113-114, once per step
  bool alive = active;
  double u_cur;
  xv_u32x4 w_cur{0u, 0u, 0u, 0u};
  act_draw(0, u_cur, w_cur);
  for (int k = 0; k < K.d_act; ++k) {                                      // wave-uniform trip count
    int a = a_cur;
    if (a < 0 || a >= A) { if (alive && valid) err |= XV_DEVERR_ACTION_RANGE; a = a < 0 ? 0 : A - 1; }
    const uint32_t rowidx = ((uint32_t)t * S + s) * A + a;
    AnyMDPTokLine L;
    L.issue_if(P.bucket, rowidx * (uint32_t)NBK + (uint32_t)(int)(u_cur * (double)NBK), alive, lane);
    L.send(u_cur, lane);
    __builtin_amdgcn_sched_barrier(0);
    float z;
    double u_next = 0.0;
    xv_u32x4 w_next{0u, 0u, 0u, 0u};
    if (INJECT) z = io.z[(size_t)k * N + ic];
    else z = xv_normal1(w_cur.z, w_cur.w);
    if (k + 1 < K.d_act) {
      act_draw(k + 1, u_next, w_next);
      a_cur = io.action[(size_t)ic * K.d_act + k + 1];
    }
    if (k == 0) early_work();
    {   // pinned here (hipcc's IR-level sinking would move them behind the search)
      int ulo = __double2loint(u_next), uhi = __double2hiint(u_next);
      asm volatile("" : "+v"(z), "+v"(ulo), "+v"(uhi), "+v"(w_next.z), "+v"(w_next.w));
      u_next = __hiloint2double(uhi, ulo);
    }
    __builtin_amdgcn_sched_barrier(0);
    bool beyond;
    float rx, ry;
    uint32_t meta;
    L.template resolve_entry<FMT>(u_cur, lane, beyond, rx, ry, meta);
    int s2 = (int)((meta >> 16) & 0x1FFu);
    bool term2 = (meta >> 25) & 1u;
    float2 rsv = make_float2(rx, ry);
    if (alive && beyond) {   // the line cannot answer this draw: search the row
      int lo = 0, m = S;
      while (m > 0) {
        const int half = m >> 1;
        if (anymdp_cdf(P, rowidx, lo + half) <= u_cur) { lo += half + 1; m -= half + 1; }
        else m = half;
      }
      s2 = lo < S - 1 ? lo : S - 1;
      rsv = anymdp_rs(P, rowidx, s2);
      term2 = (P.term_mask[(size_t)t * P.words + (s2 >> 6)] >> (s2 & 63)) & 1ull;
    }
    if (alive) {
      rsum = rsum + fmaf(rsv.y, z, rsv.x);
      rgsum = rgsum + rsv.x;
      s = s2;
      cterm = term2 ? 1 : 0;
      if (term2) { term = 1; alive = false; }
    }
    u_cur = u_next;
    w_cur = w_next;
  }
  const bool done = active && (term || trunc);
  bool do_reset = skip;
  if (done) {
    if (mode == XV_AUTORESET_SAME_STEP) do_reset = true;
    else if (mode == XV_AUTORESET_NEXT_STEP) nr = 1;
  }
  const bool keep_final = done && mode == XV_AUTORESET_SAME_STEP && io.final_obs != nullptr;
  if (HAND) {      // the env's record is final here (a restart state was drawn under the first line): hand the env on
    const uint2 q = do_reset ? anymdp_sr_pack(s_new, 0, 0, (int)((rcu.z >> (27 + k0r)) & 1u)) : anymdp_sr_pack(s, steps, nr, cterm);
    const uint32_t x = q.x | (XV_ANYMDP_SR_TAG(tick_now + 1u) << XV_ANYMDP_SR_TAG_SHIFT);
    if (valid) __hip_atomic_store(reinterpret_cast<uint64_t*>(P.sr) + i, (uint64_t)x | ((uint64_t)q.y << 32), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- observation tokens (:148-157) of the state the step ended in, two lines in flight; an env that restarts reports
  //      the observation of its restart state instead (and the step's as final_obs) ----
  for (int kp = 0; kp < DO; kp += 2) {
    const int k1 = PAIR && kp + 1 < DO ? kp + 1 : kp;
    double u0 = uS0, u1 = uS1, v0 = uR0, v1 = uR1;
    if (kp > 0) {
      obs_draw(kp, u0, v0);
      obs_draw(k1, u1, v1);
    }
    AnyMDPTokLine S0, S1, Q0, Q1;   // the step's two observation lines and, for restarting envs, the restart state's
    // (Q0, Q1 read by the owner lane alone, LPE 1 — most lanes then read line 0: no difference, scripts/runs_r04/gpu_s.sh)
    const bool wq = restarts && do_reset;
    S0.issue_if(K.obs_bucket, obs_line(kp, s, u0), !skip, lane);
    if (PAIR) S1.issue_if(K.obs_bucket, obs_line(k1, s, u1), !skip && k1 != kp, lane);
    if (restarts) {
      Q0.issue_if(K.obs_bucket, obs_line(kp, s_new, v0), wq, lane);
      if (PAIR) Q1.issue_if(K.obs_bucket, obs_line(k1, s_new, v1), wq && k1 != kp, lane);
    }
    S0.send(u0, lane);
    if (PAIR) S1.send(u1, lane);
    if (restarts) {
      Q0.send(v0, lane);
      if (PAIR) Q1.send(v1, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    // (every line requested is resolved by every wave: skipping the resolves nobody needs — the restart state's when no env
    //  of the wave restarts — behind wave-uniform branches cost 0.7 us at 2 + 2 tokens, scripts/runs_r04/gpu_s.sh: the
    //  branches split the schedule the resolves otherwise share)
    int c0, c1 = 0;
    bool f0, f1 = false;
    S0.resolve_obs(u0, lane, f0, c0);
    if (PAIR) S1.resolve_obs(u1, lane, f1, c1);
    const int ob0 = obs_pick(kp, s, u0, c0, f0, !skip), ob1 = PAIR ? obs_pick(k1, s, u1, c1, f1, !skip && k1 != kp) : 0;
    int rb0 = 0, rb1 = 0;
    if (restarts) {
      int cr0, cr1 = 0;
      bool fr0, fr1 = false;
      Q0.resolve_obs(v0, lane, fr0, cr0);
      if (PAIR) Q1.resolve_obs(v1, lane, fr1, cr1);
      rb0 = obs_pick(kp, s_new, v0, cr0, fr0, wq);
      if (PAIR) rb1 = obs_pick(k1, s_new, v1, cr1, fr1, wq && k1 != kp);
    }
    if (valid) {
      if (!skip || do_reset) {
        io.obs[(size_t)i * DO + kp] = do_reset ? rb0 : ob0;
        if (k1 != kp) io.obs[(size_t)i * DO + k1] = do_reset ? rb1 : ob1;
      }
      if (keep_final) {
        io.final_obs[(size_t)i * DO + kp] = ob0;
        if (k1 != kp) io.final_obs[(size_t)i * DO + k1] = ob1;
      }
    }
  }
  if (do_reset) {
    s = s_new;
    cterm = (int)((rcu.z >> (27 + k0r)) & 1u);
    steps = 0;
    nr = 0;
  }
  if (valid) {
    if (!HAND) P.sr[i] = anymdp_sr_pack(s, steps, nr, cterm);
    io.reward[i] = rsum; io.reward_gt[i] = rgsum;
    io.terminated[i] = (uint8_t)term; io.truncated[i] = (uint8_t)trunc;
    if (io.steps_out) io.steps_out[i] = (int32_t)steps;
    if (io.done_out) io.done_out[i] = (uint8_t)((term || trunc) ? 1 : 0);
  }
  if (err && valid) atomicOr(P.err, err);
}
template <bool INJECT, int FMT, bool PAIR, bool HAND = false>
__global__ __launch_bounds__(256) void anymdp_tok_step_coop_kernel(AnyMDPArgs P, AnyMDPTokArgs K, AnyMDPTokIO io, int mode) {
  anymdp_tok_step_coop_body<INJECT, FMT, PAIR, HAND>(P, K, io, mode, (int)blockIdx.x);
}
// The token steps' counterpart of anymdp_replay_kernel: behind the join of an overlapped xv_anymdp_step_tokens_many, a nearly
// empty launch unless a hand-off of the call expired; then every lane restores its env's record and re-runs the call's n_steps
// token steps (ring slot k % period, tick tick0 + k) with the one-stream body, and the last workgroup publishes the error word.
template <int FMT, bool PAIR>
__global__ __launch_bounds__(256) void anymdp_tok_replay_kernel(AnyMDPArgs P, AnyMDPTokArgs K, AnyMDPTokIO io /* ring slot 0 */, int period,
                                                                int n_steps, int mode, const uint2* snap, uint32_t* w, uint32_t* real_err,
                                                                uint32_t* h_fell) {
  const uint32_t e_now = __hip_atomic_load(real_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t e_in = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((e_now & XV_DEVERR_HANDOFF) == 0u) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && (e_in & XV_DEVERR_HANDOFF)) atomicOr(real_err, (uint32_t)XV_DEVERR_HANDOFF);
    return;
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < P.n_env) P.sr[i] = snap[i];
  __syncthreads();
  AnyMDPArgs Q = P;
  Q.err = w + 1;
  Q.tick_dev = nullptr;
  const size_t n = (size_t)P.n_env, da = (size_t)K.d_act, dob = (size_t)K.d_obs;
  for (int k = 0; k < n_steps; ++k) {
    const size_t o = (size_t)(k % period) * n;
    Q.tick = P.tick + (uint64_t)k;
    const AnyMDPTokIO q{io.action + o * da, nullptr, nullptr, nullptr, nullptr, nullptr, io.obs + o * dob, io.reward + o, io.reward_gt + o,
                        io.terminated + o, io.truncated + o, io.final_obs ? io.final_obs + o * dob : nullptr};
    anymdp_tok_step_coop_body<false, FMT, PAIR, false>(Q, K, q, mode, (int)blockIdx.x);
    __syncthreads();      // (an invalid lane reads the last env's record, written by a lane of this workgroup)
  }
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(w + 2, 1u) == gridDim.x - 1u) {
      const uint32_t re = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(real_err, e_in | re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      w[1] = 0u; w[2] = 0u;
      __hip_atomic_fetch_add(h_fell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

template <bool INJECT>
__global__ __launch_bounds__(256) void anymdp_tok_reset_kernel(AnyMDPArgs P, AnyMDPTokArgs K, AnyMDPTokIO io,
                                                               const uint8_t* mask) {
  const uint64_t tick_now = xv_launch_tick(P.tick, P.tick_dev);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_env) return;
  if (mask && !mask[i]) return;
  const int t = P.env_task[i];
  const uint64_t gid = P.gid_base + (uint64_t)i;
  double ur;
  if (INJECT) ur = io.u_reset[i];
  else {
    const xv_u32x4 v = xv_env_draw(P.seed, gid, tick_now, XV_DRAW_RESET);
    ur = xv_u53(v.x, v.y);
  }
  const int s = anymdp_draw_s0(P, t, ur);
  P.sr[i] = anymdp_sr_pack(s, 0, 0, (int)((P.term_mask[(size_t)t * P.words + (s >> 6)] >> (s & 63)) & 1ull));
  if (io.obs) anymdp_tok_observe<INJECT, true>(P, K, io, i, t, s, gid, io.obs);
}

// ------------------------------------------------------------------------------------------------
// Value iteration on the device, one workgroup per task (the optimal-policy teacher of the reference,
// anymdp_solver_opt.py:30-51, for whole task batches; the same synchronous sweep as
// xenoverse_amd.anymdp.task_sampler.value_iteration):
//     Q[s,a] <- ER[s,a] + gamma * sum_s' T[s,a,s'] * max_a' Q[s',a']        until rms(Q_new - Q) <= tol
// T is recovered from the row records (p_j = cdf_j - cdf_{j-1}; rows of terminal states are zero, as in the
// reference), ER[s,a] = sum_j p_j * R[s,a,j] with the fp32 rewards of the tables.
// REG (S <= 64, S*A <= 512): thread (s,a) keeps its T row in 64 fp64 registers for the whole solve; per sweep each
// wave loads V once (lane j <- V[j]) and feeds it to 64 FMAs through v_readlane (SGPR operand), so a sweep touches
// LDS only for the 4-KB Q exchange.  Otherwise: thread-strided loops over (s,a) with T re-read from the rows.
// ------------------------------------------------------------------------------------------------
template <bool REG>
__global__ __launch_bounds__(512) void anymdp_solve_kernel(AnyMDPArgs P, double gamma, double tol, int max_iter,
                                                           double* q_out, uint8_t* greedy_out, int32_t* iters_out) {
  extern __shared__ __attribute__((aligned(16))) double solve_lds[];   // Q[S*A] | V[S] | red[blockDim]
  const int t = blockIdx.x, tid = threadIdx.x, S = P.S, A = P.A, SA = S * A;
  double* Qs = solve_lds;
  double* Vs = Qs + SA;
  double* red = Vs + ((S + 63) & ~63);
  const uint64_t* tm = P.term_mask + (size_t)t * P.words;
  double Trow[REG ? 64 : 1];
  double er = 0.0;
  const int s_own = REG ? tid / A : 0, a_own = REG ? tid - s_own * A : 0;
  const bool own = REG && tid < SA;
  if (REG) {
#pragma unroll
    for (int j = 0; j < 64; ++j) Trow[j] = 0.0;
    if (own && !((tm[s_own >> 6] >> (s_own & 63)) & 1ull)) {
      const uint32_t r = ((uint32_t)t * S + s_own) * A + a_own;
      double prev = 0.0;
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        if (j < S) {
          const uint4 e = *anymdp_entry_ptr(P, r, j);
          const double c = xv_u2d(e.x, e.y);
          Trow[j] = c - prev;
          prev = c;
          er = fma(Trow[j], (double)__uint_as_float(e.z), er);
        }
      }
    }
  }
  for (int k = tid; k < SA; k += blockDim.x) Qs[k] = 0.0;
  for (int k = tid; k < ((S + 63) & ~63); k += blockDim.x) Vs[k] = 0.0;   // V of absent states multiplies T = 0
  __syncthreads();
  int it = 0;
  for (; it < max_iter; ++it) {
    for (int j = tid; j < S; j += blockDim.x) {   // V = max_a Q
      double v = Qs[j * A];
      for (int a = 1; a < A; ++a) v = fmax(v, Qs[j * A + a]);
      Vs[j] = v;
    }
    __syncthreads();
    double d2 = 0.0;
    if (REG) {
      const double vlane = Vs[threadIdx.x & 63];   // S <= 64: lane j of every wave holds V[j] (V of absent states: unused)
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        const double vj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(vlane), j),
                                           __builtin_amdgcn_readlane(__double2loint(vlane), j));
        acc = fma(Trow[j], vj, acc);
      }
      const double qn = fma(gamma, acc, er);
      if (own) {
        const double q = Qs[tid];
        d2 = (qn - q) * (qn - q);
      }
      __syncthreads();      // everyone has read the old Q
      if (own) Qs[tid] = qn;
    } else {
      // strided over (s,a); the new Q is staged behind red[] and copied after the barrier
      for (int k = tid; k < SA; k += blockDim.x) {
        const int s = k / A, a = k - s * A;
        double qn = 0.0;
        if (!((tm[s >> 6] >> (s & 63)) & 1ull)) {
          const uint32_t r = ((uint32_t)t * S + s) * A + a;
          double prev = 0.0, acc = 0.0, e_r = 0.0;
          for (int j = 0; j < S; ++j) {
            const uint4 e = *anymdp_entry_ptr(P, r, j);
            const double c = xv_u2d(e.x, e.y), p = c - prev;
            prev = c;
            e_r = fma(p, (double)__uint_as_float(e.z), e_r);
            acc = fma(p, Vs[j], acc);
          }
          qn = fma(gamma, acc, e_r);
        }
        const double q = Qs[k];
        d2 += (qn - q) * (qn - q);
        Qs[SA + ((S + 63) & ~63) + blockDim.x + k] = qn;   // staging area behind red[]
      }
      __syncthreads();
      for (int k = tid; k < SA; k += blockDim.x) Qs[k] = Qs[SA + ((S + 63) & ~63) + blockDim.x + k];
    }
    // deterministic tree reduction of the squared update
    red[tid] = d2;
    __syncthreads();
    for (int w = blockDim.x >> 1; w > 0; w >>= 1) {
      if (tid < w) red[tid] += red[tid + w];
      __syncthreads();
    }
    const double diff = sqrt(red[0] / (double)SA);
    __syncthreads();
    if (diff <= tol) { ++it; break; }
  }
  if (q_out) for (int k = tid; k < SA; k += blockDim.x) q_out[(size_t)t * SA + k] = Qs[k];
  if (greedy_out) {
    for (int j = tid; j < S; j += blockDim.x) {   // numpy.argmax: the first maximum
      int best = 0;
      double v = Qs[j * A];
      for (int a = 1; a < A; ++a) if (Qs[j * A + a] > v) { v = Qs[j * A + a]; best = a; }
      greedy_out[(size_t)t * S + j] = (uint8_t)best;
    }
  }
  if (iters_out && tid == 0) iters_out[t] = it;
}

static inline void anymdp_bind_rng(xv_anymdp* h, uint64_t ticks, bool advance = true) {
  h->a.seed = h->eng->seed;
  h->a.gid_base = h->eng->env_id_base;
  const XvTickBind b = xv_engine_bind_tick(h->eng, ticks, advance);
  h->a.tick = b.tick;
  h->a.tick_dev = b.tick_dev;
}

// What the handle's search setting means right now.  AUTO takes the bucket search only when its lines are built AND their
// census says a launch of this batch size rarely meets a draw they cannot answer: one such draw sends its wave, and with
// it the launch, through three dependent lines instead of one.  With lam = expected such draws per launch a launch costs
// t_bucket + (1 - exp(-lam)) * penalty against t_fence; measured on config 2 (65,536 envs, S = 64): rows that miss every
// cache (one task per env: 4 GiB of fence lines) 5.0-5.4 us vs 7.0-7.2 us with a penalty of ~4 us -> the bucket search wins up
// to lam ~ 0.7; rows whose fence lines stay cache resident (1,024 shared tasks: 64 MiB) 4.6 vs 4.8-5.1 us with a penalty of
// ~2 us (the synthetic 2b tasks, lam = 0.135, run 4.63 vs 5.12 us fallbacks included) -> up to lam ~ 0.2-0.3.  Thresholds with
// a margin (XV_ANYMDP_AUTO_FALLBACKS_*); the fence lines count as cache resident up to half of the 256 MB Infinity Cache.
#define XV_ANYMDP_AUTO_FALLBACKS_CACHED 0.2
#define XV_ANYMDP_AUTO_FALLBACKS_HBM 0.5
static inline double anymdp_auto_fallback_limit(const xv_anymdp* h) {
  const double fence_bytes = (double)h->a.n_task * h->a.S * h->a.A * 128.0;
  return fence_bytes <= 128.0 * 1048576.0 ? XV_ANYMDP_AUTO_FALLBACKS_CACHED : XV_ANYMDP_AUTO_FALLBACKS_HBM;
}
static inline int anymdp_effective_search(const xv_anymdp* h) {
  if (h->search == XV_ANYMDP_SEARCH_BINARY || !h->fast) return XV_ANYMDP_SEARCH_BINARY;
  if (h->a.bucket != nullptr &&
      (h->search == XV_ANYMDP_SEARCH_BUCKET || (h->search == XV_ANYMDP_SEARCH_AUTO && h->census.auto_uses_bucket)))
    return XV_ANYMDP_SEARCH_BUCKET;
  return XV_ANYMDP_SEARCH_FENCE;
}

#ifndef XV_KERNELS_ONLY   // mixed.hip includes this file for its kernels and handle types only
static void anymdp_pipe_clear(xv_anymdp* h) {      // fields of the overlapped step_many: nothing built
  h->overlap = 0; h->pipe_failed = false; h->pipe_used_last = false; h->side = nullptr; h->side_for = nullptr;
  memset(&h->gate, 0, sizeof(h->gate));      // gate, deep streams, graph bookkeeping (xv_pipe.h): nothing built
  h->side_ev[0] = h->side_ev[1] = nullptr;
  h->pgraph[0] = h->pgraph[1] = nullptr; h->pgraph_exec[0] = h->pgraph_exec[1] = nullptr;
  h->tgraph[0] = h->tgraph[1] = nullptr; h->tgraph_exec[0] = h->tgraph_exec[1] = nullptr;
  memset(&h->tpipe_key, 0, sizeof(h->tpipe_key));
  h->d_ptick = nullptr; h->ptick_value = 0; h->ptick_valid = false;
  h->d_snap = nullptr; h->d_snap_w = nullptr; h->fell_seen = 0;
  h->backoff.fell_known = 0; h->backoff.left = 0; h->backoff.len = 32;
  h->backoff_mixed.fell_known = 0; h->backoff_mixed.left = 0; h->backoff_mixed.len = 0;
  memset(&h->pipe_key, 0, sizeof(h->pipe_key));
}
static void anymdp_pipe_drop_graphs(xv_anymdp* h) {
  for (int q = 0; q < 2; ++q) {
    if (h->pgraph_exec[q]) { (void)hipGraphExecDestroy(h->pgraph_exec[q]); h->pgraph_exec[q] = nullptr; }
    if (h->pgraph[q]) { (void)hipGraphDestroy(h->pgraph[q]); h->pgraph[q] = nullptr; }
    if (h->tgraph_exec[q]) { (void)hipGraphExecDestroy(h->tgraph_exec[q]); h->tgraph_exec[q] = nullptr; }
    if (h->tgraph[q]) { (void)hipGraphDestroy(h->tgraph[q]); h->tgraph[q] = nullptr; }
  }
  memset(&h->pipe_key, 0, sizeof(h->pipe_key));
  memset(&h->tpipe_key, 0, sizeof(h->tpipe_key));
  h->gate.unroll[0] = h->gate.unroll[1] = 0;
  for (int i = 0; i < 2; ++i) {
    if (h->gate.exec_n[i]) { (void)hipGraphExecDestroy(h->gate.exec_n[i]); h->gate.exec_n[i] = nullptr; }
    if (h->gate.graph_n[i]) { (void)hipGraphDestroy(h->gate.graph_n[i]); h->gate.graph_n[i] = nullptr; }
  }
  h->gate.depth = 0;
  h->ptick_valid = false;
}
static void anymdp_pipe_release(xv_anymdp* h) {
  anymdp_pipe_drop_graphs(h);
  if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
  for (int q = 0; q < 2; ++q) if (h->side_ev[q]) (void)hipEventDestroy(h->side_ev[q]);
  if (h->d_ptick) (void)hipFree(h->d_ptick);
  if (h->d_snap) (void)hipFree(h->d_snap);
  if (h->d_snap_w) (void)hipFree(h->d_snap_w);
  for (int i = 0; i < 2; ++i) {
    if (h->gate.side_n[i]) { (void)hipStreamSynchronize(h->gate.side_n[i]); (void)hipStreamDestroy(h->gate.side_n[i]); h->gate.side_n[i] = nullptr; }
    if (h->gate.ev_n[i]) { (void)hipEventDestroy(h->gate.ev_n[i]); h->gate.ev_n[i] = nullptr; }
  }
  xv_pipe_gate_destroy(&h->gate);
  anymdp_pipe_clear(h);
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" int xv_anymdp_create(xv_engine* e, int n_env, int n_task, int S, int A, int s0_max,
                                void* rows, const int32_t* state_map, const uint64_t* term_mask,
                                const double* s0_cdf, const int32_t* s0_ids, const int32_t* max_steps,
                                const int32_t* env_task, xv_anymdp** out) {
  XV_CHECK_ARG(out != nullptr);
  *out = nullptr;
  XV_CHECK_ARG(e != nullptr);
  XV_CHECK_ARG(n_env > 0 && n_task > 0);
  XV_CHECK_ARG(S >= 2 && S <= 512 && A >= 2 && A <= 64 && s0_max >= 1 && s0_max <= 256);
  XV_CHECK_ARG(rows && state_map && term_mask && s0_cdf && s0_ids && max_steps && env_task);
  // blocks per row: ceil(S/7), rounded up to a multiple of G = ceil(blocks/16) so that a fence entry always names
  // G whole blocks (XV_ANYMDP_ROW_LINES)
  const int NB0 = (S + XV_ANYMDP_BLK - 1) / XV_ANYMDP_BLK, G = (NB0 + 15) / 16;
  const int NB = (NB0 + G - 1) / G * G, RL = 1 + NB;
  XV_CHECK_ARG((uint64_t)n_task * S * A * RL < 0xFFFFFFFFull);  // line index is a 32-bit word on the device
  XV_HIP(hipSetDevice(e->device));
  xv_anymdp* h = new (std::nothrow) xv_anymdp();
  if (!h) {
    xv_set_error("xv_anymdp_create: out of host memory");
    return XV_ERR_NOMEM;
  }
  h->eng = e;
  h->search = XV_ANYMDP_SEARCH_AUTO;
  h->fast = false;
  h->obs_cdf = nullptr; h->n_obs = 0; h->d_obs = 0; h->d_act = 0; h->obs_bucket = nullptr;
  h->max_obs = -1;
  memset(&h->census, 0, sizeof(h->census));
  h->graph_mode = 2; h->graph_failed = false; h->graph = nullptr; h->graph_exec = nullptr;
  h->d_tick = nullptr; h->d_tick_value = 0; h->d_tick_valid = false;
  memset(&h->graph_key, 0, sizeof(h->graph_key));
  h->parent = nullptr; h->view_lo = 0; h->n_views = 0; h->chain_ev = nullptr;
  h->cgraph = nullptr; h->cgraph_exec = nullptr;
  memset(&h->cgraph_key, 0, sizeof(h->cgraph_key));
  anymdp_pipe_clear(h);
  AnyMDPArgs& a = h->a;
  memset(&a, 0, sizeof(a));
  a.lines = (const uint4*)rows; a.state_map = state_map; a.term_mask = term_mask;
  a.s0_cdf = s0_cdf; a.s0_ids = s0_ids; a.max_steps = max_steps; a.env_task = env_task;
  a.n_env = n_env; a.n_task = n_task; a.S = S; a.A = A; a.s0_max = s0_max; a.words = (S + 63) / 64;
  a.NB = NB; a.RL = RL; a.G = G;
  a.err = e->d_err;
  a.seed = e->seed; a.gid_base = e->env_id_base; a.tick = 0;

  // the fast path needs s0_max <= 4, observation ids that fit 16 bits and max_steps < 2^27 (checked below)
  bool fast = (G <= 5 && s0_max <= 4);
  // largest entry of a device int array; every exit path releases the scratch word (the handle is released by the caller
  // of the lambda: nothing else has been allocated yet)
  auto device_max = [&](const int32_t* p, size_t n, int* out_max) -> hipError_t {
    int* d_max = nullptr;
    hipError_t r = hipMalloc(&d_max, sizeof(int));
    if (r == hipSuccess) r = hipMemsetAsync(d_max, 0, sizeof(int), e->stream);
    if (r == hipSuccess) {
      hipLaunchKernelGGL(anymdp_max_obs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, p, n, d_max);
      r = hipMemcpyAsync(out_max, d_max, sizeof(int), hipMemcpyDeviceToHost, e->stream);
    }
    if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
    if (d_max) (void)hipFree(d_max);
    return r;
  };
  {
    int h_max = 0;
    hipError_t r = hipSuccess;
    if (fast) {
      r = device_max(state_map, (size_t)n_task * S, &h_max);
      fast = r == hipSuccess && h_max < 65536;
      if (r == hipSuccess) h->max_obs = h_max;
    }
    if (fast) {   // max_steps shares its per-env word with four flags: it must fit 27 bits
      r = device_max(max_steps, (size_t)n_task, &h_max);
      fast = r == hipSuccess && h_max < (1 << 27);
    }
    if (r != hipSuccess) {
      xv_set_error("xv_anymdp_create: table probe failed: %s", hipGetErrorString(r));
      delete h;
      return XV_ERR_HIP;
    }
  }

  double2* ra = nullptr; uint4* rb = nullptr; uint4* rc = nullptr;
  const size_t n_rows = (size_t)n_task * S * A, ne = (size_t)n_env;
  hipError_t m = hipMalloc(&a.sr, sizeof(uint2) * ne);
  if (m == hipSuccess && fast) m = hipMalloc(&ra, sizeof(double2) * ne);
  if (m == hipSuccess && fast) m = hipMalloc(&rb, sizeof(uint4) * ne);
  if (m == hipSuccess && fast) m = hipMalloc(&rc, sizeof(uint4) * ne);
  if (m == hipSuccess)   // state 0, steps 0, need_reset
    hipLaunchKernelGGL(anymdp_init_sr_kernel, dim3(xv_div_up(n_env, 256)), dim3(256), 0, e->stream, a.sr, n_env);
  if (m != hipSuccess) {
    xv_set_error("xv_anymdp_create: device allocation failed: %s", hipGetErrorString(m));
    void* ps[] = {a.sr, ra, rb, rc};
    for (void* q : ps) if (q) (void)hipFree(q);
    delete h;
    return XV_ERR_HIP;
  }
  if (fast) {
    const size_t chunk = (size_t)1 << 24;      // rows per launch: a launch holds fewer than 2^32 threads
    for (size_t r0 = 0; r0 < n_rows; r0 += chunk) {
      const size_t nr = n_rows - r0 < chunk ? n_rows - r0 : chunk;
      hipLaunchKernelGGL(anymdp_finish_rows_kernel, dim3((unsigned)((nr * 64 + 255) / 256)), dim3(256), 0, e->stream, a,
                         (uint4*)rows, r0, nr);
    }
    hipLaunchKernelGGL(anymdp_env_records_kernel, dim3(xv_div_up(n_env, 256)), dim3(256), 0, e->stream, a, ra, rb, rc);
    a.rs_a = ra; a.rs_b = rb; a.rs_c = rc;
    h->fast = true;
  }
  XV_LAUNCH_CHECK();
  *out = h;
  return XV_OK;
}

extern "C" int xv_anymdp_set_step_many_overlap(xv_anymdp* h, int on);

static void anymdp_drop_chain_graph(xv_anymdp* h) {
  if (h->cgraph_exec) { (void)hipGraphExecDestroy(h->cgraph_exec); h->cgraph_exec = nullptr; }
  if (h->cgraph) { (void)hipGraphDestroy(h->cgraph); h->cgraph = nullptr; }
  memset(&h->cgraph_key, 0, sizeof(h->cgraph_key));
}

extern "C" int xv_anymdp_destroy(xv_anymdp* h) {
  if (!h) return XV_OK;
  if (h->n_views > 0) {   // its views read this handle's tables and env records
    xv_set_error("xv_anymdp_destroy: %d view(s) of this handle are alive: destroy them first", h->n_views);
    return XV_ERR_INVALID;
  }
  (void)hipSetDevice(h->eng->device);
  (void)hipStreamSynchronize(h->eng->stream);
  if (h->overlap) (void)xv_anymdp_set_step_many_overlap(h, 0);      // gives the device's overlap slot back
  AnyMDPArgs& a = h->a;
  if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
  if (h->graph) (void)hipGraphDestroy(h->graph);
  anymdp_drop_chain_graph(h);
  anymdp_pipe_release(h);
  if (h->d_tick) (void)hipFree(h->d_tick);
  if (h->chain_ev) (void)hipEventDestroy(h->chain_ev);
  if (h->parent) {        // a view owns its graph, tick word and event only
    anymdp_drop_chain_graph(h->parent);   // the parent's K-branch graph names this view's arguments
    h->parent->n_views -= 1;
    delete h;
    return XV_OK;
  }
  void* ps[] = {a.sr, (void*)a.rs_a, (void*)a.rs_b, (void*)a.rs_c};
  for (void* q : ps) if (q) (void)hipFree(q);
  if (h->bucket_rw) (void)hipFree(h->bucket_rw);
  if (h->obs_bucket) (void)hipFree(h->obs_bucket);
  delete h;
  return XV_OK;
}

// A view: envs [env_lo, env_lo + n_env) of `parent` behind a handle of their own.  It borrows the parent's tables, env
// records, bucket and observation lines and its search setting (as they are NOW: while views exist the parent refuses
// xv_anymdp_build_buckets / xv_anymdp_set_observation_model / xv_anymdp_destroy), and launches on the stream of `e`, whose
// seed must be the parent's and whose env_id_base must be the parent's + env_lo: the Philox counters of an env are then
// the same through either handle, so a view stepped with launch tick t writes what the parent stepped with tick t writes
// for those envs.  Every entry point that takes an xv_anymdp takes a view.  Sub-batches of one vector step are independent
// (anymdp_env.py:92-132 is per env; the batched loop of anymdp/test_utils.py:42-60 iterates envs), which is what
// xv_anymdp_step_many_chains uses.
extern "C" int xv_anymdp_view(xv_anymdp* parent, xv_engine* e, int env_lo, int n_env, xv_anymdp** out) {
  XV_CHECK_ARG(out != nullptr);
  *out = nullptr;
  XV_CHECK_ARG(parent != nullptr && e != nullptr && parent->parent == nullptr);
  XV_CHECK_ARG(env_lo >= 0 && n_env > 0 && (int64_t)env_lo + n_env <= (int64_t)parent->a.n_env);
  XV_CHECK_ARG(e->device == parent->eng->device && e->seed == parent->eng->seed);
  XV_CHECK_ARG(e->env_id_base == parent->eng->env_id_base + (uint64_t)env_lo);
  xv_anymdp* h = new (std::nothrow) xv_anymdp(*parent);
  if (!h) {
    xv_set_error("xv_anymdp_view: out of host memory");
    return XV_ERR_NOMEM;
  }
  h->eng = e;
  h->parent = parent; h->view_lo = env_lo; h->n_views = 0;
  h->bucket_rw = nullptr;                       // borrowed through a.bucket
  h->graph = nullptr; h->graph_exec = nullptr; h->graph_failed = false; h->graph_used_last = false;
  h->d_tick = nullptr; h->d_tick_value = 0; h->d_tick_valid = false;
  memset(&h->graph_key, 0, sizeof(h->graph_key));
  h->chain_ev = nullptr; h->cgraph = nullptr; h->cgraph_exec = nullptr;
  memset(&h->cgraph_key, 0, sizeof(h->cgraph_key));
  anymdp_pipe_clear(h);      // (overlap off: a view never overlaps)
  AnyMDPArgs& a = h->a;
  a.sr += env_lo;
  if (a.rs_a) { a.rs_a += env_lo; a.rs_b += env_lo; a.rs_c += env_lo; }
  a.env_task += env_lo;
  a.n_env = n_env;
  a.err = e->d_err;
  a.seed = e->seed; a.gid_base = e->env_id_base; a.tick = 0; a.tick_dev = nullptr;
  parent->n_views += 1;
  *out = h;
  return XV_OK;
}

extern "C" int xv_anymdp_reset(xv_anymdp* h, const uint8_t* mask, int32_t* obs) {
  XV_CHECK_ARG(h != nullptr);
  anymdp_bind_rng(h, 1);
  hipLaunchKernelGGL(anymdp_reset_kernel<false>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0,
                     h->eng->stream, h->a, mask, (const double*)nullptr, obs);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_anymdp_reset_injected(xv_anymdp* h, const uint8_t* mask, const double* u, int32_t* obs) {
  XV_CHECK_ARG(h != nullptr && u != nullptr);
  anymdp_bind_rng(h, 0);
  hipLaunchKernelGGL(anymdp_reset_kernel<true>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0,
                     h->eng->stream, h->a, mask, u, obs);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

template <bool INJECT>
static int anymdp_launch_step(xv_anymdp* h, const AnyMDPStepIO& io, int T, int mode) {
  const dim3 grid(xv_div_up(h->a.n_env, 256)), block(256);
#define XV_LAUNCH_STEP(GV, ROLL)                                                                  \
  hipLaunchKernelGGL((anymdp_step_kernel<INJECT, GV, ROLL>), grid, block, 0, h->eng->stream, h->a, io, T, mode)
#define XV_LAUNCH_STEP_G(GV) do { if (roll) XV_LAUNCH_STEP(GV, true); else XV_LAUNCH_STEP(GV, false); } while (0)
  const bool roll = T > 1 || io.greedy != nullptr;
  const int eff = anymdp_effective_search(h);
  if (eff == XV_ANYMDP_SEARCH_BUCKET) {
#define XV_LAUNCH_BK(GV, FV)                                                                                              \
  do {                                                                                                                     \
    if (roll) hipLaunchKernelGGL((anymdp_step_kernel<INJECT, GV, true, false, FV>), grid, block, 0, h->eng->stream, h->a, io, T, mode);  \
    else hipLaunchKernelGGL((anymdp_step_kernel<INJECT, GV, false, false, FV>), grid, block, 0, h->eng->stream, h->a, io, T, mode);      \
  } while (0)
    if (h->a.bfmt == 1) {        // 7 cuts per line: S <= 256, i.e. G <= 3
      if (h->a.G == 1) XV_LAUNCH_BK(1, 1);
      else if (h->a.G == 2) XV_LAUNCH_BK(2, 1);
      else XV_LAUNCH_BK(3, 1);
    } else {
      if (h->a.G == 1) XV_LAUNCH_BK(1, 2);
      else if (h->a.G == 2) XV_LAUNCH_BK(2, 2);
      else if (h->a.G == 3) XV_LAUNCH_BK(3, 2);
      else if (h->a.G == 4) XV_LAUNCH_BK(4, 2);
      else XV_LAUNCH_BK(5, 2);
    }
#undef XV_LAUNCH_BK
  } else if (eff == XV_ANYMDP_SEARCH_FENCE) {
    if (h->a.G == 1) XV_LAUNCH_STEP_G(1);
    else if (h->a.G == 2) XV_LAUNCH_STEP_G(2);
    else if (h->a.G == 3) XV_LAUNCH_STEP_G(3);
    else if (h->a.G == 4) XV_LAUNCH_STEP_G(4);
    else XV_LAUNCH_STEP_G(5);
  } else {
    XV_LAUNCH_STEP_G(0);
  }
#undef XV_LAUNCH_STEP_G
#undef XV_LAUNCH_STEP
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_anymdp_step(xv_anymdp* h, const int32_t* action, int32_t* obs, float* reward,
                              float* reward_gt, uint8_t* terminated, uint8_t* truncated,
                              int32_t* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && action && obs && reward && reward_gt && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  anymdp_bind_rng(h, 1);
  AnyMDPStepIO io{action, nullptr, nullptr, nullptr, obs, reward, reward_gt, terminated, truncated, final_obs, nullptr, nullptr, 0.0f};
  return anymdp_launch_step<false>(h, io, 1, autoreset_mode);
}

extern "C" int xv_anymdp_step_info(xv_anymdp* h, const int32_t* action, int32_t* obs, float* reward, float* reward_gt,
                                   uint8_t* terminated, uint8_t* truncated, int32_t* final_obs, int32_t* steps, uint8_t* done,
                                   int autoreset_mode) {
  XV_CHECK_ARG(h && action && obs && reward && reward_gt && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  anymdp_bind_rng(h, 1);
  AnyMDPStepIO io{action, nullptr, nullptr, nullptr, obs, reward, reward_gt, terminated, truncated, final_obs, nullptr, nullptr, 0.0f,
                  steps, done};
  return anymdp_launch_step<false>(h, io, 1, autoreset_mode);
}

extern "C" int xv_anymdp_step_injected(xv_anymdp* h, const int32_t* action, const double* u,
                                       const float* z, const double* u_reset, int32_t* obs,
                                       float* reward, float* reward_gt, uint8_t* terminated,
                                       uint8_t* truncated, int32_t* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && action && u && z && u_reset && obs && reward && reward_gt && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  anymdp_bind_rng(h, 0);
  AnyMDPStepIO io{action, u, z, u_reset, obs, reward, reward_gt, terminated, truncated, final_obs, nullptr, nullptr, 0.0f};
  return anymdp_launch_step<true>(h, io, 1, autoreset_mode);
}

// One ring cycle of xv_anymdp_step_many as a graph: `period` step-kernel nodes in a chain (node j reads actions slot j,
// writes output slot j, draws with tick *d_tick + j) and a node that advances *d_tick by `period`.  Back-to-back
// dependent launches cost ~3.3 us each on a stream and ~1.6 us as graph nodes (scripts/devtools/graph_floor.hip).
// `stride`: elements between two ring slots (n_env for the handle's own rings; the parent's n_env when a view steps its
// columns of the parent's rings, xv_anymdp_step_many_chains).
static void* anymdp_graph_step_fn(const xv_anymdp* h, int eff, bool hand = false) {
  const bool fast = eff != XV_ANYMDP_SEARCH_BINARY;
  const int bk = eff == XV_ANYMDP_SEARCH_BUCKET ? h->a.bfmt : 0;
  const int Gv = h->a.G;
  if (hand) {      // the hand-off kernels: fence and bucket searches
#define XV_STEP_FN(GV, BKV) reinterpret_cast<void*>(&anymdp_step_kernel<false, GV, false, true, BKV, true>)
    if (!fast) return nullptr;
    return bk == 1 ? (Gv == 1 ? XV_STEP_FN(1, 1) : Gv == 2 ? XV_STEP_FN(2, 1) : XV_STEP_FN(3, 1))
           : bk == 2 ? (Gv == 1 ? XV_STEP_FN(1, 2) : Gv == 2 ? XV_STEP_FN(2, 2) : Gv == 3 ? XV_STEP_FN(3, 2)
                        : Gv == 4 ? XV_STEP_FN(4, 2) : XV_STEP_FN(5, 2))
           : (Gv == 1 ? XV_STEP_FN(1, 0) : Gv == 2 ? XV_STEP_FN(2, 0) : Gv == 3 ? XV_STEP_FN(3, 0)
              : Gv == 4 ? XV_STEP_FN(4, 0) : XV_STEP_FN(5, 0));
#undef XV_STEP_FN
  }
#define XV_STEP_FN(GV, BKV) reinterpret_cast<void*>(&anymdp_step_kernel<false, GV, false, true, BKV>)
  void* fn = bk == 1 ? (Gv == 1 ? XV_STEP_FN(1, 1) : Gv == 2 ? XV_STEP_FN(2, 1) : XV_STEP_FN(3, 1))
             : bk == 2 ? (Gv == 1 ? XV_STEP_FN(1, 2) : Gv == 2 ? XV_STEP_FN(2, 2) : Gv == 3 ? XV_STEP_FN(3, 2)
                          : Gv == 4 ? XV_STEP_FN(4, 2) : XV_STEP_FN(5, 2))
             : !fast ? XV_STEP_FN(0, 0)
             : (Gv == 1 ? XV_STEP_FN(1, 0) : Gv == 2 ? XV_STEP_FN(2, 0) : Gv == 3 ? XV_STEP_FN(3, 0)
                : Gv == 4 ? XV_STEP_FN(4, 0) : XV_STEP_FN(5, 0));
#undef XV_STEP_FN
  return fn;
}

// the chain of `period` step nodes of handle `h` appended to `graph` behind `prev` (nullptr: a root); -> its last node
static bool anymdp_add_chain(xv_anymdp* h, hipGraph_t graph, hipGraphNode_t* prev, uint64_t* d_tick, int eff, int period,
                             size_t stride, const int32_t* actions, int32_t* obs, float* reward, float* reward_gt,
                             uint8_t* terminated, uint8_t* truncated, int32_t* final_obs, int mode, int j0 = 0, int dj = 1,
                             bool hand = false, int reps = 1) {
  const dim3 grid(xv_div_up(h->a.n_env, 256)), block(256);
  void* fn = anymdp_graph_step_fn(h, eff, hand);
  if (!fn) return false;
  // `reps` ring cycles in one graph (overlapped paths): step g of the graph set goes to stream g % dj, uses ring slot
  // g % period and tick base + g
  for (int g = j0; g < reps * period; g += dj) {
    const int j = g % period;
    hipKernelNodeParams np;
    memset(&np, 0, sizeof(np));
    AnyMDPArgs a = h->a;
    a.seed = h->eng->seed; a.gid_base = h->eng->env_id_base;
    a.tick = (uint64_t)g; a.tick_dev = d_tick;
    const size_t off = (size_t)j * stride;
    AnyMDPStepIO io{actions + off, nullptr, nullptr, nullptr, obs + off, reward + off, reward_gt + off, terminated + off,
                    truncated + off, final_obs ? final_obs + off : nullptr, nullptr, nullptr, 0.0f};
    int T = 1, md = mode;
    void* step_params[] = {&a, &io, &T, &md};
    np.func = fn; np.gridDim = grid; np.blockDim = block; np.kernelParams = step_params;
    hipGraphNode_t node;
    if (hipGraphAddKernelNode(&node, graph, *prev ? prev : nullptr, *prev ? 1 : 0, &np) != hipSuccess) return false;
    *prev = node;
  }
  return true;
}

static bool anymdp_add_tick_node(hipGraph_t graph, const hipGraphNode_t* deps, int n_deps, uint64_t* d_tick, int period) {
  hipKernelNodeParams np;
  memset(&np, 0, sizeof(np));
  uint64_t dv = (uint64_t)period;
  void* tick_params[] = {&d_tick, &dv};
  np.func = reinterpret_cast<void*>(&anymdp_advance_tick_kernel); np.gridDim = dim3(1); np.blockDim = dim3(1);
  np.kernelParams = tick_params;
  hipGraphNode_t node;
  return hipGraphAddKernelNode(&node, graph, deps, (size_t)n_deps, &np) == hipSuccess;
}

// head of cycle graph q of the overlapped paths: tick word q += period, then the cycle gate (every stream's graph: xv_pipe.h)
static bool anymdp_add_head_node(xv_anymdp* h, hipGraph_t graph, hipGraphNode_t* prev, int q, int period, int unroll) {
  hipKernelNodeParams np;
  memset(&np, 0, sizeof(np));
  uint64_t* t = h->d_ptick + q;
  uint64_t dv = (uint64_t)period * (uint64_t)unroll;
  uint32_t* seen = h->gate.d_seen + q;
  const uint32_t* issued = h->gate.d_issued;
  uint32_t* err = h->a.err;
  void* params[] = {&t, &dv, &seen, &issued, &err};
  np.func = reinterpret_cast<void*>(&anymdp_pipe_head_kernel); np.gridDim = dim3(1); np.blockDim = dim3(1);
  np.kernelParams = params;
  hipGraphNode_t node;
  if (hipGraphAddKernelNode(&node, graph, nullptr, 0, &np) != hipSuccess) return false;
  *prev = node;
  return true;
}

static bool anymdp_ensure_graph(xv_anymdp* h, int period, size_t stride, const int32_t* actions, int32_t* obs, float* reward,
                                float* reward_gt, uint8_t* terminated, uint8_t* truncated, int32_t* final_obs, int mode) {
  const int eff = anymdp_effective_search(h);
  const bool fast = eff != XV_ANYMDP_SEARCH_BINARY;
  const void* ptrs[7] = {actions, obs, reward, reward_gt, terminated, truncated, final_obs};
  auto& K = h->graph_key;
  if (h->graph_exec && K.period == period && K.mode == mode && K.search == eff && K.fast == (int)fast && K.stride == stride &&
      K.bucket == (const void*)h->a.bucket && K.nbk == h->a.NBK &&
      K.seed == h->eng->seed && K.gid_base == h->eng->env_id_base && memcmp(K.ptrs, ptrs, sizeof(ptrs)) == 0)
    return true;
  if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
  if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
  if (!h->d_tick && hipMalloc(&h->d_tick, sizeof(uint64_t)) != hipSuccess) return false;
  if (hipGraphCreate(&h->graph, 0) != hipSuccess) return false;
  hipGraphNode_t prev = nullptr;
  if (!anymdp_add_chain(h, h->graph, &prev, h->d_tick, eff, period, stride, actions, obs, reward, reward_gt, terminated,
                        truncated, final_obs, mode))
    return false;
  if (!anymdp_add_tick_node(h->graph, &prev, 1, h->d_tick, period)) return false;
  if (hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0) != hipSuccess) { h->graph_exec = nullptr; return false; }
  K.period = period; K.mode = mode; K.search = eff; K.fast = (int)fast; K.stride = stride;
  K.bucket = (const void*)h->a.bucket; K.nbk = h->a.NBK;
  K.seed = h->eng->seed; K.gid_base = h->eng->env_id_base;
  memcpy(K.ptrs, ptrs, sizeof(ptrs));
  return true;
}

// measured (scripts/devtools/graph_step_many.py, config 2b): graph replay wins 7-9 % at 1,024-4,096 envs, nothing at
// 16,384 and loses 2-3 % at 65,536 in config 2a when thousands of steps are issued (its kernels read the tick from memory;
// the stream is not the limiter) — but a SHORT burst is one submission instead of n_steps: 20 steps of 65,536 envs take
// 154 instead of 173 us.  AUTO: small batches, or short bursts.
#define XV_ANYMDP_GRAPH_AUTO_MAX 8192
#define XV_ANYMDP_GRAPH_AUTO_STEPS 128
static inline bool anymdp_graph_wanted(const xv_anymdp* h, int n_steps) {
  if (h->eng->dev_tick) return false;   // the graph keeps its own tick word, fed from the host tick: plain launches here
  return h->graph_mode == 1 ||
         (h->graph_mode == 2 && (h->a.n_env <= XV_ANYMDP_GRAPH_AUTO_MAX || n_steps <= XV_ANYMDP_GRAPH_AUTO_STEPS));
}

extern "C" int xv_anymdp_set_step_many_graph(xv_anymdp* h, int mode) {
  XV_CHECK_ARG(h != nullptr && mode >= 0 && mode <= 2);
  h->graph_mode = mode;
  return XV_OK;
}

extern "C" int xv_anymdp_step_many_graph_state(xv_anymdp* h) {   // 0 plain launches, 1 graph built and in use, -1 failed
  if (!h) return 0;
  if (h->graph_failed) return -1;
  return (h->graph_mode != 0 && (h->graph_exec || h->cgraph_exec) && h->graph_used_last) ? 1 : 0;
}

// the stepping of xv_anymdp_step_many in three pieces, so that xv_anymdp_step_many_chains can interleave the cycles of
// several handles: prepare (graph + its tick word), one ring cycle, one plain step
static bool anymdp_many_prepare(xv_anymdp* h, int n_steps, int period, size_t stride, const int32_t* actions, int32_t* obs,
                                float* reward, float* reward_gt, uint8_t* terminated, uint8_t* truncated, int32_t* final_obs,
                                int mode, bool force_graph) {
  h->graph_used_last = false;
  if (n_steps / period <= 0 || period <= 1 || h->graph_failed) return false;
  if (!(force_graph ? (!h->eng->dev_tick && h->graph_mode != 0) : anymdp_graph_wanted(h, n_steps))) return false;
  bool ok = anymdp_ensure_graph(h, period, stride, actions, obs, reward, reward_gt, terminated, truncated, final_obs, mode);
  if (ok && !(h->d_tick_valid && h->d_tick_value == h->eng->tick)) {
    hipLaunchKernelGGL(anymdp_set_tick_kernel, dim3(1), dim3(1), 0, h->eng->stream, h->d_tick, h->eng->tick);
    ok = hipGetLastError() == hipSuccess;
    if (ok) { h->d_tick_value = h->eng->tick; h->d_tick_valid = true; }
  }
  if (!ok) {   // same kernels, plain launches; never retried on this handle
    (void)hipGetLastError();
    h->graph_failed = true;
    h->d_tick_valid = false;
  }
  return ok;
}

static bool anymdp_many_cycle(xv_anymdp* h, int period) {
  if (hipGraphLaunch(h->graph_exec, h->eng->stream) != hipSuccess) {
    (void)hipGetLastError();
    h->graph_failed = true;
    h->d_tick_valid = false;
    return false;
  }
  h->eng->tick += (uint64_t)period;
  h->d_tick_value = h->eng->tick;
  h->d_tick_valid = true;
  h->graph_used_last = true;
  return true;
}

static int anymdp_many_plain(xv_anymdp* h, int k, int period, size_t stride, const int32_t* actions, int32_t* obs,
                             float* reward, float* reward_gt, uint8_t* terminated, uint8_t* truncated, int32_t* final_obs,
                             int mode) {
  const size_t off = (size_t)(k % period) * stride;
  anymdp_bind_rng(h, 1);
  AnyMDPStepIO io{actions + off, nullptr, nullptr, nullptr, obs + off, reward + off, reward_gt + off,
                  terminated + off, truncated + off, final_obs ? final_obs + off : nullptr, nullptr, nullptr, 0.0f};
  return anymdp_launch_step<false>(h, io, 1, mode);
}

// Overlapped step_many: consecutive vector steps alternate between the engine's stream (even steps) and a side stream (odd
// steps) with NO dependency between the streams — step k + 1 is dispatched while step k runs, and each of its waves takes
// its envs over from the same wave of step k through the hand-off word (HAND kernels above).  What a stream's barrier
// between two launches costs — the drain of one launch and the dispatch of the next, 2.7 of the step's 5.0 us at 65,536
// envs — is then covered by the other stream's launch.  Whole ring cycles of an even period replay two cycle graphs (ring
// slots 0, 2, ... / 1, 3, ..., each with its own tick word advanced by its own last node); what is left over runs the ordinary
// way behind the join.  Calls shorter than XV_ANYMDP_PIPE_GRAPH_MIN steps take the ordinary path altogether: the second
// hipGraphLaunch reaches the device ~12-25 us after the first and the join costs a cross-queue wait, so a 20-step burst
// (~110 us) loses what the overlap gains (measured 5.9-6.4 vs 5.55 us per step; issuing the steps as plain launches
// alternately on the two streams is host-bound: 5.7; both chains as the branches of ONE graph: 6.8, and 4.3 instead of 3.7
// on long calls; profiles/r05_c_*, r05_d_burst_timeline.txt, r05_n_*).  Same launch ticks, same
// results as the ordinary path (tests/test_gpu_chains.py).
#define XV_ANYMDP_PIPE_GRAPH_MIN 64      // calls of at least this many steps are overlapped
static bool anymdp_pipe_setup(xv_anymdp* h) {
  if (h->side && h->side_for != h->eng->stream) {      // the engine moved to another stream: choose again
    (void)hipStreamSynchronize(h->side);
    anymdp_pipe_drop_graphs(h);
    (void)hipStreamDestroy(h->side);
    h->side = nullptr;
  }
  if (!h->side) {
    // the side stream is chosen by measurement (xv_pipe.h): a ping-pong of chained launches over the two streams must cost
    // about what the same chain costs on one stream — concurrent hardware queues, not two that take turns
    h->side_for = h->eng->stream;
    if (!xv_pipe_pick_side_stream(h->eng->stream, &h->side, nullptr, nullptr)) { h->side = nullptr; return false; }
    if (!h->side_ev[0] && (hipEventCreateWithFlags(&h->side_ev[0], hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&h->side_ev[1], hipEventDisableTiming) != hipSuccess))
      return false;
  }
  if (!h->d_ptick && hipMalloc(&h->d_ptick, XV_PIPE_DEPTH_MAX * sizeof(uint64_t)) != hipSuccess) return false;   // the graphs' tick words
  if (!h->gate.d_seen && !xv_pipe_gate_create(&h->gate)) return false;
  if (!h->d_snap && hipMalloc(&h->d_snap, (size_t)h->a.n_env * sizeof(uint2)) != hipSuccess) { h->d_snap = nullptr; return false; }
  if (!h->d_snap_w) {
    if (hipMalloc(&h->d_snap_w, 4 * sizeof(uint32_t)) != hipSuccess) { h->d_snap_w = nullptr; return false; }
    if (hipMemsetAsync(h->d_snap_w, 0, 4 * sizeof(uint32_t), h->eng->stream) != hipSuccess) return false;
  }
  return true;
}
// the replay kernel for this handle's layout and search (HAND kernels exist for the fence and bucket searches only)
static bool anymdp_launch_replay(xv_anymdp* h, int eff, const AnyMDPStepIO& io, int period, int cycles, int mode, uint64_t t0) {
  const int bk = eff == XV_ANYMDP_SEARCH_BUCKET ? h->a.bfmt : 0;
  AnyMDPArgs a = h->a;
  a.seed = h->eng->seed; a.gid_base = h->eng->env_id_base; a.tick = t0; a.tick_dev = nullptr;
  const dim3 grid(xv_div_up(h->a.n_env, 256)), block(256);
  uint32_t* h_fell = h->gate.d_issued + 1;
#define XV_REPLAY(GV, BKV) \
  hipLaunchKernelGGL((anymdp_replay_kernel<GV, BKV>), grid, block, 0, h->eng->stream, a, io, period, cycles, mode, \
                     (const uint2*)h->d_snap, h->d_snap_w, h->a.err, h_fell)
#define XV_REPLAY_G(BKV)                                                                 \
  do {                                                                                   \
    if (h->a.G == 1) XV_REPLAY(1, BKV); else if (h->a.G == 2) XV_REPLAY(2, BKV);          \
    else if (h->a.G == 3) XV_REPLAY(3, BKV); else if (h->a.G == 4) XV_REPLAY(4, BKV);     \
    else XV_REPLAY(5, BKV);                                                               \
  } while (0)
  if (bk == 1) {      // 7 cuts per line: G <= 3
    if (h->a.G == 1) XV_REPLAY(1, 1); else if (h->a.G == 2) XV_REPLAY(2, 1); else XV_REPLAY(3, 1);
  } else if (bk == 2) {
    XV_REPLAY_G(2);
  } else {
    XV_REPLAY_G(0);
  }
#undef XV_REPLAY_G
#undef XV_REPLAY
  return hipGetLastError() == hipSuccess;
}
// streams 2 .. depth - 1 (xv_pipe_depth() > 2): each accepted beside the engine's stream by the same timed trial
static bool anymdp_pipe_setup_deep(xv_anymdp* h, int depth) {
  XvPipeGate& G = h->gate;
  if (G.side_n_for != h->eng->stream) {
    for (int i = 0; i < 2; ++i)
      if (G.side_n[i]) { (void)hipStreamSynchronize(G.side_n[i]); (void)hipStreamDestroy(G.side_n[i]); G.side_n[i] = nullptr; }
    G.side_n_for = h->eng->stream;
  }
  for (int i = 0; i < depth - 2; ++i) {
    // (beside the engine's stream AND beside the first side stream; a fourth stream is tried against those two only)
    if (!G.side_n[i] && !xv_pipe_pick_side_stream(h->eng->stream, &G.side_n[i], nullptr, nullptr, h->side)) { G.side_n[i] = nullptr; return false; }
    if (!G.ev_n[i] && hipEventCreateWithFlags(&G.ev_n[i], hipEventDisableTiming) != hipSuccess) return false;
  }
  return true;
}

// -> cycles per graph the call uses (graphs built / reused), 0: this call is not overlapped, -1: failure
static int anymdp_pipe_graphs(xv_anymdp* h, int D, int period, int cycles, size_t stride, const int32_t* actions, int32_t* obs,
                              float* reward, float* reward_gt, uint8_t* terminated, uint8_t* truncated, int32_t* final_obs, int mode) {
  const int eff = anymdp_effective_search(h);
  const void* ptrs[7] = {actions, obs, reward, reward_gt, terminated, truncated, final_obs};
  auto& K = h->pipe_key;
  const bool same = h->pgraph_exec[0] && h->pgraph_exec[1] && K.period == period && K.mode == mode && K.search == eff &&
                    K.stride == stride && K.bucket == (const void*)h->a.bucket && K.seed == h->eng->seed &&
                    K.gid_base == h->eng->env_id_base && memcmp(K.ptrs, ptrs, sizeof(ptrs)) == 0;
  const bool same_d = same && h->gate.depth == D;
  const int U = xv_pipe_pick_unroll(period, cycles, same_d ? h->gate.unroll[0] : 0, D);
  if (U == 0) return 0;
  if (same_d && U == h->gate.unroll[0]) return U;
  (void)hipStreamSynchronize(h->side);
  for (int i = 0; i < 2; ++i) if (h->gate.side_n[i]) (void)hipStreamSynchronize(h->gate.side_n[i]);
  (void)hipStreamSynchronize(h->eng->stream);
  anymdp_pipe_drop_graphs(h);
  if (D > 2 && !anymdp_pipe_setup_deep(h, D)) return -1;
  if (!xv_pipe_gate_sync(&h->gate)) return -1;      // (every stream of the handle has drained)
  for (int q = 0; q < D; ++q) {
    hipGraph_t* gr = q < 2 ? &h->pgraph[q] : &h->gate.graph_n[q - 2];
    hipGraphExec_t* ge = q < 2 ? &h->pgraph_exec[q] : &h->gate.exec_n[q - 2];
    if (hipGraphCreate(gr, 0) != hipSuccess) return -1;
    hipGraphNode_t prev = nullptr;
    if (!anymdp_add_head_node(h, *gr, &prev, q, period, U)) return -1;
    if (!anymdp_add_chain(h, *gr, &prev, h->d_ptick + q, eff, period, stride, actions, obs, reward, reward_gt,
                          terminated, truncated, final_obs, mode, q, D, true, U))
      return -1;
    if (hipGraphInstantiate(ge, *gr, nullptr, nullptr, 0) != hipSuccess) {
      *ge = nullptr;
      return -1;
    }
  }
  h->gate.depth = D;
  K.period = period; K.mode = mode; K.search = eff; K.stride = stride; K.bucket = (const void*)h->a.bucket;
  K.seed = h->eng->seed; K.gid_base = h->eng->env_id_base;
  memcpy(K.ptrs, ptrs, sizeof(ptrs));
  h->gate.unroll[0] = U;
  return U;
}

// the whole ring cycles of a call, overlapped; *issued = steps issued (0: the caller takes the ordinary path for all of it).
// -> XV_OK, or an error when a cycle went out in part (the streams are joined either way)
// what the replay of an overlapped call needs (ring slot 0 of the call's buffers)
struct AnyMDPReplay {
  AnyMDPStepIO io;      // MDP steps (xv_anymdp_step_many)
  int ring_period, mode, eff;
  bool tokens;          // token steps (xv_anymdp_step_tokens_many): `tio` instead of `io`
  AnyMDPTokIO tio;
};
static int anymdp_pipe_launch(xv_anymdp* h, hipGraphExec_t* ex, int cycles, int period, int* issued, int depth = 2,
                              const AnyMDPReplay* rp = nullptr);

static int anymdp_pipe_run(xv_anymdp* h, int n_steps, int period, size_t stride, const int32_t* actions, int32_t* obs,
                           float* reward, float* reward_gt, uint8_t* terminated, uint8_t* truncated, int32_t* final_obs,
                           int mode, int* issued) {
  *issued = 0;
  const int cycles = n_steps / period;
  static const int min_steps = getenv("XV_ANYMDP_PIPE_MIN_STEPS") ? atoi(getenv("XV_ANYMDP_PIPE_MIN_STEPS")) : XV_ANYMDP_PIPE_GRAPH_MIN;
  if (cycles <= 0 || period % 2 != 0 || n_steps < min_steps) return XV_OK;
  if (xv_pipe_backoff_step(&h->backoff, &h->gate)) return XV_OK;      // a recent call was replayed: one stream for a while
  // two or three launches resident at once, or the one-stream path (xv_pipe.h)
  int D = xv_pipe_choose_depth(anymdp_graph_step_fn(h, anymdp_effective_search(h), true), 256,
                               (size_t)xv_div_up(h->a.n_env, 256), h->eng->device);
  if (D > period) D = period;      // the steps in flight write distinct ring slots
  if (D > 2 && xv_pipe_pick_unroll(period, cycles, 0, D) == 0) D = 2;      // too few ring cycles for a three-stream graph set
  if (D < 2) return XV_OK;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;      // not inside a stream capture: the set-up synchronises
  if (hipStreamIsCapturing(h->eng->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return XV_OK; }
  const bool set_up = anymdp_pipe_setup(h);
  if (set_up && D > 2 && !anymdp_pipe_setup_deep(h, D)) { (void)hipGetLastError(); D = 2; }      // no third stream qualifies: two
  const int U = set_up ? anymdp_pipe_graphs(h, D, period, cycles, stride, actions, obs, reward, reward_gt, terminated,
                                            truncated, final_obs, mode) : -1;
  if (U < 0) {
    (void)hipGetLastError();
    h->pipe_failed = true;
    return XV_OK;
  }
  if (U == 0) return XV_OK;      // too short for the graphs this handle holds: one stream
  AnyMDPReplay rp;
  memset(&rp, 0, sizeof(rp));
  rp.io = AnyMDPStepIO{actions, nullptr, nullptr, nullptr, obs, reward, reward_gt, terminated, truncated, final_obs, nullptr, nullptr, 0.0f};
  rp.ring_period = period; rp.mode = mode; rp.eff = anymdp_effective_search(h); rp.tokens = false;
  return anymdp_pipe_launch(h, h->pgraph_exec, cycles / U, period * U, issued, h->gate.depth, &rp);
}

static __global__ __launch_bounds__(256) void anymdp_pipe_open_kernel(uint2* sr, int n, uint32_t tag, uint64_t* tick_words, int n_words,
                                                                      uint64_t tick_base, uint2* snap, uint32_t* snap_w, uint32_t* err,
                                                                      int bad_tag_env, int repairable) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    uint2 r = sr[i];
    r.x &= (1u << XV_ANYMDP_SR_TAG_SHIFT) - 1u;
    snap[i] = r;      // what the replay restores should a hand-off of this call expire (anymdp_replay_kernel)
    r.x |= (i == bad_tag_env ? ((tag + 77u) & 0x3FFFu) : tag) << XV_ANYMDP_SR_TAG_SHIFT;      // bad_tag_env: test hook, -1 otherwise
    sr[i] = r;
  }
  if (i < n_words) tick_words[i] = tick_base;
  // the error word as it is at entry; the HANDOFF bit leaves the word for the time of the call (a wave that finds it set takes
  // it for "this call has failed": xv_hand_aborted) and returns with the replay kernel
  if (i == 0 && repairable) { const uint32_t e = *err; snap_w[0] = e; snap_w[1] = 0u; snap_w[2] = 0u; *err = e & ~(uint32_t)XV_DEVERR_HANDOFF; }
}

// `cycles` replays of the cycle graphs ex[0] (engine's stream) / ex[1] (side stream) / the deep ones: tags and tick words, fork,
// launches, join
static int anymdp_pipe_launch(xv_anymdp* h, hipGraphExec_t* ex, int cycles, int period, int* issued, int depth, const AnyMDPReplay* rp) {
  hipStream_t st = h->eng->stream;
  const int test_bad_tag = getenv("XV_PIPE_TEST_BAD_TAG") ? atoi(getenv("XV_PIPE_TEST_BAD_TAG")) : 0;   // tests: env 0's tag is wrong
  const int test_fail = getenv("XV_PIPE_TEST_FAIL") ? atoi(getenv("XV_PIPE_TEST_FAIL")) : 0;            // tests: see anymdp_test_fail_kernel
  h->fell_seen = __atomic_load_n(h->gate.h_issued + 1, __ATOMIC_ACQUIRE);
  const int deep = depth > 2 ? depth - 2 : 0;      // streams beyond the engine's and `side`
  const uint64_t t0 = h->eng->tick;
  // one launch opens the call: every record gets the tag of the first step (whatever ran since the last overlapped call
  // wrote the tag bits as 0), the graphs' tick words the base their head nodes advance by `period` first
  hipLaunchKernelGGL(anymdp_pipe_open_kernel, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, st, h->a.sr, h->a.n_env,
                     XV_ANYMDP_SR_TAG(t0), h->d_ptick, XV_PIPE_DEPTH_MAX, t0 - (uint64_t)period, h->d_snap, h->d_snap_w,
                     h->a.err, (rp && test_bad_tag) ? 0 : -1, rp ? 1 : 0);
  bool ok = hipGetLastError() == hipSuccess;
  // fork: the side stream starts behind what the engine's stream holds now (the caller's actions, a reset, the tags above)
  ok = ok && hipEventRecord(h->side_ev[0], st) == hipSuccess && hipStreamWaitEvent(h->side, h->side_ev[0], 0) == hipSuccess;
  for (int i = 0; i < deep; ++i) ok = ok && hipStreamWaitEvent(h->gate.side_n[i], h->side_ev[0], 0) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    h->pipe_failed = true; h->ptick_valid = false;
    return XV_OK;
  }
  h->ptick_valid = false;      // until the cycles have gone out
  int k = 0;
  bool broken = false;
  for (int c = 0; c < cycles; ++c) {
    // both halves or neither: an even half without its odd half leaves the next even launch waiting (bounded, flagged)
    // (the even half starts with the cycle gate: it runs once both halves are enqueued, however long the host takes)
    if (hipGraphLaunch(ex[0], st) != hipSuccess) break;
    xv_pipe_test_stall(c);
    if (hipGraphLaunch(ex[1], h->side) != hipSuccess) { broken = true; xv_pipe_gate_release(&h->gate); break; }
    for (int i = 0; i < deep && !broken; ++i) {
      if (i == 0) xv_pipe_test_stall(c, 1);
      if (hipGraphLaunch(h->gate.exec_n[i], h->gate.side_n[i]) != hipSuccess) broken = true;
    }
    if (broken) { xv_pipe_gate_release(&h->gate); break; }
    xv_pipe_gate_release(&h->gate);
    k += period;
    h->eng->tick = t0 + (uint64_t)k;
  }
  if (k < cycles * period) { (void)hipGetLastError(); h->pipe_failed = true; }
  bool joined = hipEventRecord(h->side_ev[1], h->side) == hipSuccess && hipStreamWaitEvent(st, h->side_ev[1], 0) == hipSuccess;
  for (int i = 0; i < deep; ++i)
    joined = hipEventRecord(h->gate.ev_n[i], h->gate.side_n[i]) == hipSuccess && hipStreamWaitEvent(st, h->gate.ev_n[i], 0) == hipSuccess && joined;
  *issued = k;
  if (broken || !joined) {
    (void)hipGetLastError();
    h->pipe_failed = true;
    xv_set_error("xv_anymdp_step_many: an overlapped ring cycle could be issued only in part; the envs' states are undefined");
    return XV_ERR_HIP;
  }
  if (rp && k > 0) {
    // behind the join: should a hand-off of this call have expired, the call is replayed from its entry state on this
    // stream (anymdp_replay_kernel: a nearly empty launch otherwise) — an expiry costs time, never results
    if (test_fail) {
      const size_t n_obs = (size_t)rp->ring_period * (size_t)h->a.n_env * (size_t)(rp->tokens ? h->d_obs : 1);
      hipLaunchKernelGGL(anymdp_test_fail_kernel, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, st, h->a.sr, h->a.n_env, h->a.err,
                         rp->tokens ? rp->tio.obs : rp->io.obs, n_obs);
    }
    bool replay_ok;
    if (rp->tokens) {
      AnyMDPArgs a = h->a;
      a.seed = h->eng->seed; a.gid_base = h->eng->env_id_base; a.tick = t0; a.tick_dev = nullptr;
      AnyMDPTokArgs KA{h->obs_cdf, h->n_obs, h->d_obs, h->d_act, h->obs_bucket};
      const dim3 grid(xv_div_up(h->a.n_env, 256)), block(256);
      uint32_t* h_fell = h->gate.d_issued + 1;
#define XV_TOK_REPLAY(FV, PV) \
  hipLaunchKernelGGL((anymdp_tok_replay_kernel<FV, PV>), grid, block, 0, st, a, KA, rp->tio, rp->ring_period, k, rp->mode, \
                     (const uint2*)h->d_snap, h->d_snap_w, h->a.err, h_fell)
      if (h->a.bfmt == 1) { if (h->d_obs > 1) XV_TOK_REPLAY(1, true); else XV_TOK_REPLAY(1, false); }
      else { if (h->d_obs > 1) XV_TOK_REPLAY(2, true); else XV_TOK_REPLAY(2, false); }
#undef XV_TOK_REPLAY
      replay_ok = hipGetLastError() == hipSuccess;
    } else {
      replay_ok = anymdp_launch_replay(h, rp->eff, rp->io, rp->ring_period, k / rp->ring_period, rp->mode, t0);
    }
    if (!replay_ok) {
      (void)hipGetLastError();
      h->pipe_failed = true;      // (this call's hand-offs are flagged as before; the next calls take the one-stream path)
    }
  }
  if (k > 0) { h->ptick_value = h->eng->tick; h->ptick_valid = true; h->graph_used_last = true; h->pipe_used_last = true; }
  return XV_OK;
}

// One handle per device may have the overlap on at a time, and never a view.  Two overlapped calls in flight at once can
// deadlock on the hardware queues: streams are mapped onto a few queues, each in order, and a launch that waits for a wave
// of a launch queued BEHIND another handle's waiting launch never gets it (measured: two views, each overlapped, 4.5 us
// per step instead of 3.6; four, their waits ran into the bound — scripts/devtools/probe_views_overlap.py).
// (the slot is the device's, shared with the other families' overlapped paths: engine.hip, xv_device_overlap_acquire)
extern "C" int xv_anymdp_set_step_many_overlap(xv_anymdp* h, int on) {
  XV_CHECK_ARG(h != nullptr && (on == 0 || on == 1));
  const int dev = h->eng->device;
  XV_CHECK_ARG(dev >= 0 && dev < 64);
  if (on) {
    if (h->parent != nullptr) {
      xv_set_error("xv_anymdp_set_step_many_overlap: not on a view (overlap the parent's step_many instead)");
      return XV_ERR_UNSUPPORTED;
    }
    if (!xv_device_overlap_acquire(dev, h)) {
      xv_set_error("xv_anymdp_set_step_many_overlap: another handle on device %d has the overlap on; one at a time", dev);
      return XV_ERR_UNSUPPORTED;
    }
    h->pipe_failed = false;
    xv_pipe_backoff_reset(&h->backoff, &h->gate);
    h->backoff_mixed.len = 0;      // (mixed.hip starts it over against its own replay counter)
  } else {
    xv_device_overlap_release(dev, h);
  }
  h->overlap = on;
  return XV_OK;
}

// 1: the last xv_anymdp_step_many overlapped its ring cycles, 0: it did not (off, odd period, per-lane search, device tick,
// graph mode 0), -1: the overlapped path failed on this handle and is no longer tried, -2: the last call overlapped, a
// hand-off expired and the call was replayed on one stream (results are right; meaningful once the stream has drained)
extern "C" int xv_anymdp_step_many_overlap_state(xv_anymdp* h) {
  if (!h) return 0;
  if (h->pipe_failed) return -1;
  if (h->overlap && h->pipe_used_last && h->gate.h_issued &&
      __atomic_load_n(h->gate.h_issued + 1, __ATOMIC_ACQUIRE) != h->fell_seen)
    return -2;
  return (h->overlap && h->pipe_used_last) ? 1 : 0;
}

extern "C" int xv_anymdp_step_many(xv_anymdp* h, int n_steps, int period, const int32_t* actions,
                                   int32_t* obs, float* reward, float* reward_gt, uint8_t* terminated,
                                   uint8_t* truncated, int32_t* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && n_steps > 0 && period > 0);
  XV_CHECK_ARG(actions && obs && reward && reward_gt && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  const size_t n = (size_t)h->a.n_env;
  int k = 0;
  // whole ring cycles: replay the graph
  const int cycles = n_steps / period;
  h->pipe_used_last = false;
  if (h->overlap && h->graph_mode != 0 && !h->eng->dev_tick && !h->pipe_failed &&
      anymdp_effective_search(h) != XV_ANYMDP_SEARCH_BINARY) {
    XV_HIP(hipSetDevice(h->eng->device));
    h->graph_used_last = false;
    const int rc = anymdp_pipe_run(h, n_steps, period, n, actions, obs, reward, reward_gt, terminated, truncated, final_obs,
                                   autoreset_mode, &k);
    if (rc != XV_OK) return rc;
  }
  if (k == 0 && anymdp_many_prepare(h, n_steps, period, n, actions, obs, reward, reward_gt, terminated, truncated, final_obs,
                          autoreset_mode, false))
    for (int c = 0; c < cycles && anymdp_many_cycle(h, period); ++c) k += period;
  for (; k < n_steps; ++k) {
    const int rc = anymdp_many_plain(h, k, period, n, actions, obs, reward, reward_gt, terminated, truncated, final_obs,
                                     autoreset_mode);
    if (rc != XV_OK) return rc;
  }
  return XV_OK;
}

// ONE graph for a ring cycle of all chains: K branches (the views' chains of `period` step nodes, all reading the parent's
// tick word) joined by the node that advances the word.  Kept on the parent.
static bool anymdp_ensure_chain_graph(xv_anymdp* p, xv_anymdp* const* views, int n_views, int period, const int32_t* actions,
                                      int32_t* obs, float* reward, float* reward_gt, uint8_t* terminated, uint8_t* truncated,
                                      int32_t* final_obs, int mode) {
  const int eff = anymdp_effective_search(p);
  const void* ptrs[7] = {actions, obs, reward, reward_gt, terminated, truncated, final_obs};
  auto& K = p->cgraph_key;
  bool same = p->cgraph_exec && K.period == period && K.mode == mode && K.search == eff && K.n_views == n_views &&
              K.bucket == (const void*)p->a.bucket && memcmp(K.ptrs, ptrs, sizeof(ptrs)) == 0;
  for (int v = 0; same && v < n_views; ++v) same = K.views[v] == (const void*)views[v];
  if (same) return true;
  anymdp_drop_chain_graph(p);
  if (!p->d_tick && hipMalloc(&p->d_tick, sizeof(uint64_t)) != hipSuccess) return false;
  if (hipGraphCreate(&p->cgraph, 0) != hipSuccess) return false;
  hipGraphNode_t tails[XV_ANYMDP_MAX_CHAINS];
  for (int v = 0; v < n_views; ++v) {
    xv_anymdp* h = views[v];
    const size_t lo = (size_t)h->view_lo;
    tails[v] = nullptr;
    if (!anymdp_add_chain(h, p->cgraph, &tails[v], p->d_tick, eff, period, (size_t)p->a.n_env, actions + lo, obs + lo,
                          reward + lo, reward_gt + lo, terminated + lo, truncated + lo, final_obs ? final_obs + lo : nullptr,
                          mode))
      return false;
  }
  if (!anymdp_add_tick_node(p->cgraph, tails, n_views, p->d_tick, period)) return false;
  if (hipGraphInstantiate(&p->cgraph_exec, p->cgraph, nullptr, nullptr, 0) != hipSuccess) { p->cgraph_exec = nullptr; return false; }
  K.period = period; K.mode = mode; K.search = eff; K.n_views = n_views; K.bucket = (const void*)p->a.bucket;
  memcpy(K.ptrs, ptrs, sizeof(ptrs));
  for (int v = 0; v < n_views; ++v) K.views[v] = (const void*)views[v];
  return true;
}

// xv_anymdp_step_many with the envs stepped as K independent chains: views[c] (xv_anymdp_view) covers a contiguous range
// of the parent's envs, the K ranges tile it in order.  Step k of chain c depends on step k - 1 of chain c only — so the
// chains run on their own streams (how = 0: each replays its own cycle graph, or plain launches when graphs are off) or as
// the K branches of one graph on the parent's stream (how = 1), and the launch-to-launch gap of one chain (an empty launch
// of this grid is 2.7 of the step's 5.0 us) is covered by the other chains' table lines in flight.  Same launch ticks and
// the same per-env Philox counters as one chain: every output, the env records and the parent's tick afterwards equal
// xv_anymdp_step_many's bit for bit.  Arrays are the parent's [period][parent n_env] rings.  Stream order: the chains start
// behind what the parent's stream holds at the call and the parent's stream waits for all of them before it goes on.
// Host tick only (device-tick engines: XV_ERR_UNSUPPORTED).
extern "C" int xv_anymdp_step_many_chains(xv_anymdp* p, xv_anymdp* const* views, int n_views, int how, int n_steps, int period,
                                          const int32_t* actions, int32_t* obs, float* reward, float* reward_gt,
                                          uint8_t* terminated, uint8_t* truncated, int32_t* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(p && views && n_views >= 1 && n_views <= XV_ANYMDP_MAX_CHAINS && (how == 0 || how == 1));
  XV_CHECK_ARG(n_steps > 0 && period > 0);
  XV_CHECK_ARG(actions && obs && reward && reward_gt && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  int covered = 0;
  bool dev_tick = p->eng->dev_tick;
  for (int v = 0; v < n_views; ++v) {
    XV_CHECK_ARG(views[v] != nullptr && views[v]->parent == p && views[v]->view_lo == covered);
    covered += views[v]->a.n_env;
    dev_tick = dev_tick || views[v]->eng->dev_tick;
  }
  XV_CHECK_ARG(covered == p->a.n_env);
  if (dev_tick) {
    xv_set_error("xv_anymdp_step_many_chains: needs the host tick on the parent's and the views' engines");
    return XV_ERR_UNSUPPORTED;
  }
  XV_HIP(hipSetDevice(p->eng->device));
  const uint64_t t0 = p->eng->tick;
  const size_t N = (size_t)p->a.n_env;
  const int cycles = n_steps / period;
  p->graph_used_last = false;
  for (int v = 0; v < n_views; ++v) {      // the views follow the parent: search setting and launch tick
    views[v]->search = p->search;
    views[v]->graph_mode = p->graph_mode;
    views[v]->eng->tick = t0;
  }
  int k = 0;
  if (how == 1) {
    bool ok = cycles > 0 && period > 1 && p->graph_mode != 0 && !p->graph_failed &&
              anymdp_ensure_chain_graph(p, views, n_views, period, actions, obs, reward, reward_gt, terminated, truncated,
                                        final_obs, autoreset_mode);
    if (ok && !(p->d_tick_valid && p->d_tick_value == t0)) {
      hipLaunchKernelGGL(anymdp_set_tick_kernel, dim3(1), dim3(1), 0, p->eng->stream, p->d_tick, t0);
      ok = hipGetLastError() == hipSuccess;
    }
    for (int c = 0; ok && c < cycles; ++c) {
      ok = hipGraphLaunch(p->cgraph_exec, p->eng->stream) == hipSuccess;
      if (ok) {
        k += period;
        p->eng->tick = t0 + (uint64_t)k;
        p->d_tick_value = p->eng->tick;
        p->d_tick_valid = true;
        p->graph_used_last = true;
      }
    }
    if (!ok && cycles > 0 && period > 1 && p->graph_mode != 0) { (void)hipGetLastError(); p->d_tick_valid = false; }
    for (; k < n_steps; ++k) {           // what is left of the call: the parent's own launches
      const int rc = anymdp_many_plain(p, k, period, N, actions, obs, reward, reward_gt, terminated, truncated, final_obs,
                                       autoreset_mode);
      if (rc != XV_OK) return rc;
    }
    for (int v = 0; v < n_views; ++v) views[v]->eng->tick = t0 + (uint64_t)n_steps;
    return XV_OK;
  }
  // how = 0: fork
  if (!p->chain_ev) XV_HIP(hipEventCreateWithFlags(&p->chain_ev, hipEventDisableTiming));
  XV_HIP(hipEventRecord(p->chain_ev, p->eng->stream));
  bool graph[XV_ANYMDP_MAX_CHAINS];
  bool all_graph = true;
  for (int v = 0; v < n_views; ++v) {
    xv_anymdp* h = views[v];
    if (h->eng->stream != p->eng->stream) XV_HIP(hipStreamWaitEvent(h->eng->stream, p->chain_ev, 0));
    const size_t lo = (size_t)h->view_lo;
    graph[v] = anymdp_many_prepare(h, n_steps, period, N, actions + lo, obs + lo, reward + lo, reward_gt + lo, terminated + lo,
                                   truncated + lo, final_obs ? final_obs + lo : nullptr, autoreset_mode, true);
    all_graph = all_graph && graph[v];
  }
  int rc = XV_OK;
  if (all_graph) {
    for (int c = 0; c < cycles && all_graph; ++c) {
      for (int v = 0; v < n_views; ++v) all_graph = anymdp_many_cycle(views[v], period) && all_graph;
      if (all_graph) k += period;
    }
    p->graph_used_last = k > 0;
  }
  // a graph launch that failed mid-cycle leaves the chains at different steps: bring every chain to the furthest one
  for (int v = 0; v < n_views; ++v) k = (int)(views[v]->eng->tick - t0) > k ? (int)(views[v]->eng->tick - t0) : k;
  for (int v = 0; v < n_views && rc == XV_OK; ++v) {
    xv_anymdp* h = views[v];
    const size_t lo = (size_t)h->view_lo;
    for (int kk = (int)(h->eng->tick - t0); kk < k && rc == XV_OK; ++kk)
      rc = anymdp_many_plain(h, kk, period, N, actions + lo, obs + lo, reward + lo, reward_gt + lo, terminated + lo,
                             truncated + lo, final_obs ? final_obs + lo : nullptr, autoreset_mode);
  }
  for (; k < n_steps && rc == XV_OK; ++k)
    for (int v = 0; v < n_views && rc == XV_OK; ++v) {
      xv_anymdp* h = views[v];
      const size_t lo = (size_t)h->view_lo;
      rc = anymdp_many_plain(h, k, period, N, actions + lo, obs + lo, reward + lo, reward_gt + lo, terminated + lo,
                             truncated + lo, final_obs ? final_obs + lo : nullptr, autoreset_mode);
    }
  // join (also after an error: the parent's stream must not run ahead of launches already issued)
  for (int v = 0; v < n_views; ++v) {
    xv_anymdp* h = views[v];
    if (h->eng->stream == p->eng->stream) continue;
    if (!h->chain_ev && hipEventCreateWithFlags(&h->chain_ev, hipEventDisableTiming) != hipSuccess) { rc = rc == XV_OK ? XV_ERR_HIP : rc; continue; }
    if (hipEventRecord(h->chain_ev, h->eng->stream) != hipSuccess ||
        hipStreamWaitEvent(p->eng->stream, h->chain_ev, 0) != hipSuccess)
      rc = rc == XV_OK ? XV_ERR_HIP : rc;
  }
  if (rc == XV_ERR_HIP) xv_set_error("xv_anymdp_step_many_chains: joining the chains failed: %s", hipGetErrorString(hipGetLastError()));
  if (rc == XV_OK) p->eng->tick = t0 + (uint64_t)n_steps;
  return rc;
}

extern "C" int xv_anymdp_rollout(xv_anymdp* h, int T, const int32_t* actions, int32_t* obs, float* reward,
                                 float* reward_gt, uint8_t* terminated, uint8_t* truncated,
                                 int32_t* final_obs) {
  XV_CHECK_ARG(h && T > 0 && actions && obs && reward && reward_gt && terminated && truncated);
  anymdp_bind_rng(h, (uint64_t)T);
  AnyMDPStepIO io{actions, nullptr, nullptr, nullptr, obs, reward, reward_gt, terminated, truncated, final_obs, nullptr, nullptr, 0.0f};
  return anymdp_launch_step<false>(h, io, T, XV_AUTORESET_SAME_STEP);
}

extern "C" int xv_anymdp_rollout_teacher(xv_anymdp* h, int T, const uint8_t* greedy, float epsilon, int32_t* actions_out,
                                         int32_t* obs, float* reward, float* reward_gt, uint8_t* terminated,
                                         uint8_t* truncated, int32_t* final_obs) {
  XV_CHECK_ARG(h && T > 0 && greedy && actions_out && obs && reward && reward_gt && terminated && truncated);
  XV_CHECK_ARG(epsilon >= 0.0f && epsilon <= 1.0f);
  anymdp_bind_rng(h, (uint64_t)T);
  AnyMDPStepIO io{nullptr, nullptr, nullptr, nullptr, obs, reward, reward_gt, terminated, truncated, final_obs, greedy,
                  actions_out, epsilon};
  // T == 1 must still take the rollout instantiation (it is the one that honours the teacher fields per step)
  return anymdp_launch_step<false>(h, io, T, XV_AUTORESET_SAME_STEP);
}

extern "C" int xv_anymdp_solve(xv_anymdp* h, double gamma, double tol, int max_iter, double* q_out,
                               uint8_t* greedy_out, int32_t* iters_out) {
  XV_CHECK_ARG(h != nullptr && (q_out || greedy_out));
  XV_CHECK_ARG(gamma > 0.0 && gamma < 1.0 && tol > 0.0 && max_iter > 0);
  const AnyMDPArgs& a = h->a;
  const int SA = a.S * a.A, Sp = (a.S + 63) & ~63;
  const bool reg = a.S <= 64 && SA <= 512;
  const int threads = 512;
  const size_t lds = sizeof(double) * ((size_t)SA + Sp + threads + (reg ? 0 : SA));
  if (lds > 150 * 1024) {
    xv_set_error("xv_anymdp_solve: S*A = %d does not fit the workgroup's LDS", SA);
    return XV_ERR_UNSUPPORTED;
  }
  if (reg) {
    hipLaunchKernelGGL(anymdp_solve_kernel<true>, dim3(a.n_task), dim3(threads), lds, h->eng->stream, a, gamma, tol,
                       max_iter, q_out, greedy_out, iters_out);
  } else {
    if (lds > 48 * 1024)
      XV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&anymdp_solve_kernel<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(anymdp_solve_kernel<false>, dim3(a.n_task), dim3(threads), lds, h->eng->stream, a, gamma, tol,
                       max_iter, q_out, greedy_out, iters_out);
  }
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_anymdp_set_search(xv_anymdp* h, int search) {
  XV_CHECK_ARG(h != nullptr);
  XV_CHECK_ARG(search == XV_ANYMDP_SEARCH_AUTO || search == XV_ANYMDP_SEARCH_BINARY ||
               search == XV_ANYMDP_SEARCH_FENCE || search == XV_ANYMDP_SEARCH_BUCKET);
  if (search == XV_ANYMDP_SEARCH_BUCKET && h->a.bucket == nullptr) {
    xv_set_error("xv_anymdp_set_search: BUCKET needs xv_anymdp_build_buckets first");
    return XV_ERR_UNSUPPORTED;
  }
  if (search == XV_ANYMDP_SEARCH_FENCE && !h->fast) {
    xv_set_error("xv_anymdp_set_search: FENCE needs s0_max <= 4, observation ids < 65536 and max_steps < 2^27");
    return XV_ERR_UNSUPPORTED;
  }
  h->search = search;
  return XV_OK;
}

// observation bucket lines for the current observation model and NBK (no-op without either); frees stale ones
static int anymdp_build_obs_buckets(xv_anymdp* h) {
  if (h->obs_bucket) {
    XV_HIP(hipStreamSynchronize(h->eng->stream));
    (void)hipFree(h->obs_bucket);
    h->obs_bucket = nullptr;
  }
  h->census.obs_lines = 0; h->census.obs_lines_dirty = 0; h->census.obs_p_fallback = 0.0;
  if (!h->obs_cdf || !h->a.bucket || h->a.NBK <= 0) return XV_OK;
  if (h->n_obs > 256) return XV_OK;                                    // symbol ids are bytes in the lines: the per-lane kernel serves
  const size_t n_rows = (size_t)h->a.n_task * h->d_obs * h->a.S;
  if (n_rows * (size_t)h->a.NBK >= (1ull << 32)) return XV_OK;      // 32-bit line index: the per-lane kernel serves
  // budget: the lines may take what is free minus 2 GiB of headroom for the caller (1,024 tasks x 4 tokens x S = 256 x 16
  // buckets are 2 GiB); beyond that the per-lane kernel serves (xv_anymdp_token_kernel reports which one runs)
  size_t free_b = 0, total_b = 0;
  const size_t need = n_rows * (size_t)h->a.NBK * 128;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need + ((size_t)2 << 30) > free_b) return XV_OK;
  uint4* b = nullptr;
  unsigned long long* d_cen = nullptr;
  if (hipMalloc(&b, need) != hipSuccess || hipMalloc(&d_cen, 3 * sizeof(unsigned long long)) != hipSuccess) {
    (void)hipGetLastError();
    if (b) (void)hipFree(b);
    return XV_OK;                                                  // not fatal: the per-lane kernel serves
  }
  hipError_t r = hipMemsetAsync(d_cen, 0, 3 * sizeof(unsigned long long), h->eng->stream);
  const size_t chunk = (size_t)1 << 22;
  for (size_t r0 = 0; r == hipSuccess && r0 < n_rows; r0 += chunk) {
    const size_t nr = n_rows - r0 < chunk ? n_rows - r0 : chunk;
    hipLaunchKernelGGL(anymdp_build_obs_cutlines_kernel, dim3((unsigned)((nr * (size_t)h->a.NBK + 255) / 256)), dim3(256), 0,
                       h->eng->stream, h->obs_cdf + r0 * (size_t)h->n_obs, nr, h->n_obs, h->a.NBK, b + r0 * (size_t)h->a.NBK * 8, d_cen);
    r = hipGetLastError();
  }
  unsigned long long cen[3] = {0, 0, 0};
  if (r == hipSuccess) r = hipMemcpyAsync(cen, d_cen, sizeof(cen), hipMemcpyDeviceToHost, h->eng->stream);
  if (r == hipSuccess) r = hipStreamSynchronize(h->eng->stream);
  (void)hipFree(d_cen);
  if (r != hipSuccess) {
    (void)hipFree(b);
    xv_set_error("xv_anymdp: building the observation bucket lines failed: %s", hipGetErrorString(r));
    return XV_ERR_HIP;
  }
  h->obs_bucket = b;
  h->census.obs_lines = (uint64_t)(n_rows * (size_t)h->a.NBK);
  h->census.obs_lines_dirty = (uint64_t)cen[0];
  h->census.obs_p_fallback = cen[2] ? (double)cen[1] / 68719476736.0 / (double)cen[2] : 0.0;
  return XV_OK;
}

// runs the cut-line builder over every row: writes the lines into `b` (nullptr: census only) and fills `out`
static int anymdp_run_cutlines(xv_anymdp* h, int n_bucket, uint4* b, xv_anymdp_bucket_census* out) {
  const size_t n_rows = (size_t)h->a.n_task * h->a.S * h->a.A;
  const int fmt = (h->a.S <= 256 && h->max_obs >= 0 && h->max_obs <= 255) ? 1 : 2;
  unsigned long long* d_cen = nullptr;
  XV_HIP(hipMalloc(&d_cen, 3 * sizeof(unsigned long long)));
  hipError_t r = hipMemsetAsync(d_cen, 0, 3 * sizeof(unsigned long long), h->eng->stream);
  const size_t chunk = (size_t)1 << 22;      // rows per launch: a launch holds fewer than 2^32 threads
  for (size_t r0 = 0; r == hipSuccess && r0 < n_rows; r0 += chunk) {
    const size_t nr = n_rows - r0 < chunk ? n_rows - r0 : chunk;
    const dim3 grid((unsigned)((nr * (size_t)n_bucket + 255) / 256)), block(256);
    if (fmt == 1) hipLaunchKernelGGL(anymdp_build_cutlines_kernel<1>, grid, block, 0, h->eng->stream, h->a, b, r0, nr, n_bucket, d_cen);
    else hipLaunchKernelGGL(anymdp_build_cutlines_kernel<2>, grid, block, 0, h->eng->stream, h->a, b, r0, nr, n_bucket, d_cen);
    r = hipGetLastError();
  }
  unsigned long long cen[3] = {0, 0, 0};
  if (r == hipSuccess) r = hipMemcpyAsync(cen, d_cen, sizeof(cen), hipMemcpyDeviceToHost, h->eng->stream);
  if (r == hipSuccess) r = hipStreamSynchronize(h->eng->stream);
  (void)hipFree(d_cen);
  if (r != hipSuccess) {
    xv_set_error("xv_anymdp_build_buckets: building the lines failed: %s", hipGetErrorString(r));
    return XV_ERR_HIP;
  }
  memset(out, 0, sizeof(*out));
  out->n_bucket = n_bucket;
  out->format = fmt;
  out->cuts_per_line = fmt == 1 ? 7 : 6;
  out->built = b != nullptr;
  out->lines = (uint64_t)(n_rows * (size_t)n_bucket);
  out->lines_dirty = (uint64_t)cen[0];
  out->live_rows = (uint64_t)cen[2];
  out->p_fallback = cen[2] ? (double)cen[1] / 68719476736.0 / (double)cen[2] : 0.0;
  out->fallbacks_per_launch = out->p_fallback * (double)h->a.n_env;
  out->auto_limit = anymdp_auto_fallback_limit(h);
  out->auto_uses_bucket = out->fallbacks_per_launch <= out->auto_limit;
  out->bytes = (double)n_rows * (double)n_bucket * 128.0;
  return XV_OK;
}

static int anymdp_buckets_supported(const xv_anymdp* h, int n_bucket) {
  if (!h->fast) {
    xv_set_error("xv_anymdp_build_buckets: needs the fence layout (s0_max <= 4, observation ids < 65536, max_steps < 2^27)");
    return XV_ERR_UNSUPPORTED;
  }
  const size_t n_rows = (size_t)h->a.n_task * h->a.S * h->a.A;
  if (n_rows * (size_t)n_bucket >= (1ull << 32)) {
    xv_set_error("xv_anymdp_build_buckets: %zu bucket lines exceed the 32-bit line index", n_rows * (size_t)n_bucket);
    return XV_ERR_UNSUPPORTED;
  }
  return XV_OK;
}

extern "C" int xv_anymdp_probe_buckets(xv_anymdp* h, int n_bucket, xv_anymdp_bucket_census* out) {
  XV_CHECK_ARG(h != nullptr && out != nullptr && (n_bucket == 16 || n_bucket == 32 || n_bucket == 64));
  XV_HIP(hipSetDevice(h->eng->device));
  const int rc = anymdp_buckets_supported(h, n_bucket);
  if (rc != XV_OK) return rc;
  return anymdp_run_cutlines(h, n_bucket, nullptr, out);
}

extern "C" int xv_anymdp_bucket_census_get(xv_anymdp* h, xv_anymdp_bucket_census* out) {
  XV_CHECK_ARG(h != nullptr && out != nullptr);
  *out = h->census;
  return XV_OK;
}

extern "C" int xv_anymdp_effective_search(xv_anymdp* h) {
  return h ? anymdp_effective_search(h) : XV_ERR_INVALID;
}

#define XV_ANYMDP_NO_VIEWS(h)                                                                                   \
  do {                                                                                                           \
    if ((h)->parent != nullptr || (h)->n_views > 0) {                                                            \
      xv_set_error("%s: not on a view, and not while views of the handle exist (they borrow its lines)", __func__); \
      return XV_ERR_UNSUPPORTED;                                                                                 \
    }                                                                                                            \
  } while (0)

extern "C" int xv_anymdp_build_buckets(xv_anymdp* h, int n_bucket) {
  XV_CHECK_ARG(h != nullptr && (n_bucket == 0 || n_bucket == 16 || n_bucket == 32 || n_bucket == 64));
  XV_ANYMDP_NO_VIEWS(h);
  XV_HIP(hipSetDevice(h->eng->device));
  if (h->bucket_rw) {
    XV_HIP(hipStreamSynchronize(h->eng->stream));
    // an instantiated step_many graph holds the old lines' address and count in its kernel arguments: drop it with them
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
    if (h->side) XV_HIP(hipStreamSynchronize(h->side));
    for (int i = 0; i < 2; ++i) if (h->gate.side_n[i]) XV_HIP(hipStreamSynchronize(h->gate.side_n[i]));
    anymdp_pipe_drop_graphs(h);
    (void)hipFree(h->bucket_rw);
    h->bucket_rw = nullptr; h->a.bucket = nullptr; h->a.NBK = 0; h->a.bfmt = 0;
    memset(&h->census, 0, sizeof(h->census));
    if (h->search == XV_ANYMDP_SEARCH_BUCKET) h->search = XV_ANYMDP_SEARCH_AUTO;
  }
  if (n_bucket == 0) return anymdp_build_obs_buckets(h);   // frees the observation lines too
  int rc = anymdp_buckets_supported(h, n_bucket);
  if (rc != XV_OK) return rc;
  const size_t bytes = (size_t)h->a.n_task * h->a.S * h->a.A * (size_t)n_bucket * 128;
  uint4* b = nullptr;
  if (hipMalloc(&b, bytes) != hipSuccess) {
    (void)hipGetLastError();
    xv_set_error("xv_anymdp_build_buckets: cannot allocate %.1f GiB of bucket lines", (double)bytes / (double)(1ull << 30));
    return XV_ERR_NOMEM;
  }
  xv_anymdp_bucket_census cen;
  rc = anymdp_run_cutlines(h, n_bucket, b, &cen);
  if (rc != XV_OK) {
    (void)hipFree(b);
    return rc;
  }
  h->bucket_rw = b; h->a.bucket = b; h->a.NBK = n_bucket; h->a.bfmt = cen.format;
  h->census = cen;
  return anymdp_build_obs_buckets(h);
}

extern "C" int xv_anymdp_get_state(xv_anymdp* h, int32_t* inner_state, int32_t* steps, uint8_t* need_reset) {
  XV_CHECK_ARG(h != nullptr);
  hipLaunchKernelGGL(anymdp_get_state_kernel, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream, h->a,
                     inner_state, steps, need_reset);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_anymdp_set_state(xv_anymdp* h, const int32_t* inner_state, const int32_t* steps,
                                   const uint8_t* need_reset) {
  XV_CHECK_ARG(h != nullptr);
  hipLaunchKernelGGL(anymdp_set_state_kernel, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream, h->a,
                     inner_state, steps, need_reset);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_anymdp_transition_gt(xv_anymdp* h, const int32_t* action, double* out) {
  XV_CHECK_ARG(h && action && out);
  const size_t total = (size_t)h->a.n_env * h->a.S;
  hipLaunchKernelGGL(anymdp_tgt_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->eng->stream,
                     h->a, action, out);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

// ---- POMDP / MTPOMDP entry points ----
extern "C" int xv_anymdp_set_observation_model(xv_anymdp* h, int n_obs, int d_obs, int d_act, const double* obs_cdf) {
  XV_CHECK_ARG(h && obs_cdf && n_obs >= 1 && d_obs >= 1 && d_obs <= 64 && d_act >= 1 && d_act <= 64);
  XV_ANYMDP_NO_VIEWS(h);
  XV_HIP(hipSetDevice(h->eng->device));   // the observation bucket lines are allocated and built on the engine's device
  h->obs_cdf = obs_cdf; h->n_obs = n_obs; h->d_obs = d_obs; h->d_act = d_act;
  return anymdp_build_obs_buckets(h);   // beside existing transition bucket lines
}

// which kernel xv_anymdp_step_tokens launches now: 1 = the cooperative kernel on transition + observation bucket lines
// (bucket search in effect and both line sets built), 0 = the per-lane kernel
static inline bool anymdp_tok_coop(const xv_anymdp* h) {
  return anymdp_effective_search(h) == XV_ANYMDP_SEARCH_BUCKET && h->a.NBK > 0 && h->obs_bucket != nullptr;
}
extern "C" int xv_anymdp_token_kernel(xv_anymdp* h) { return h ? (anymdp_tok_coop(h) ? 1 : 0) : XV_ERR_INVALID; }

template <bool INJECT>
static int anymdp_tok_launch_step(xv_anymdp* h, const AnyMDPTokIO& io, int mode) {
  AnyMDPTokArgs K{h->obs_cdf, h->n_obs, h->d_obs, h->d_act, h->obs_bucket};
  const dim3 grid(xv_div_up(h->a.n_env, 256)), block(256);
  if (anymdp_tok_coop(h)) {   // bucket search: the cooperative kernel
    const bool pair = K.d_obs > 1;
    if (h->a.bfmt == 1) {
      if (pair) hipLaunchKernelGGL((anymdp_tok_step_coop_kernel<INJECT, 1, true>), grid, block, 0, h->eng->stream, h->a, K, io, mode);
      else hipLaunchKernelGGL((anymdp_tok_step_coop_kernel<INJECT, 1, false>), grid, block, 0, h->eng->stream, h->a, K, io, mode);
    } else {
      if (pair) hipLaunchKernelGGL((anymdp_tok_step_coop_kernel<INJECT, 2, true>), grid, block, 0, h->eng->stream, h->a, K, io, mode);
      else hipLaunchKernelGGL((anymdp_tok_step_coop_kernel<INJECT, 2, false>), grid, block, 0, h->eng->stream, h->a, K, io, mode);
    }
  } else {
    hipLaunchKernelGGL(anymdp_tok_step_kernel<INJECT>, grid, block, 0, h->eng->stream, h->a, K, io, mode);
  }
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_anymdp_step_tokens(xv_anymdp* h, const int32_t* action, int32_t* obs, float* reward,
                                     float* reward_gt, uint8_t* terminated, uint8_t* truncated, int32_t* final_obs,
                                     int autoreset_mode) {
  XV_CHECK_ARG(h && h->obs_cdf && action && obs && reward && reward_gt && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  anymdp_bind_rng(h, 1);
  AnyMDPTokIO io{action, nullptr, nullptr, nullptr, nullptr, nullptr, obs, reward, reward_gt, terminated, truncated, final_obs};
  return anymdp_tok_launch_step<false>(h, io, autoreset_mode);
}

// xv_anymdp_step_tokens that also writes info["steps"] and the terminated | truncated mask from the same launch (as
// xv_anymdp_step_info does for the MDP step): a Python-level step() of a POMDP is then ONE launch
extern "C" int xv_anymdp_step_tokens_info(xv_anymdp* h, const int32_t* action, int32_t* obs, float* reward, float* reward_gt,
                                          uint8_t* terminated, uint8_t* truncated, int32_t* final_obs, int32_t* steps,
                                          uint8_t* done, int autoreset_mode) {
  XV_CHECK_ARG(h && h->obs_cdf && action && obs && reward && reward_gt && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  anymdp_bind_rng(h, 1);
  AnyMDPTokIO io{action, nullptr, nullptr, nullptr, nullptr, nullptr, obs, reward, reward_gt, terminated, truncated, final_obs,
                 steps, done};
  return anymdp_tok_launch_step<false>(h, io, autoreset_mode);
}

// n_steps token steps issued from C over ring buffers: step k reads actions slot k % period ([period][n_env][d_act]) and
// writes slot k % period of the outputs (obs / final_obs [period][n_env][d_obs], the others [period][n_env]); equals
// n_steps calls of xv_anymdp_step_tokens (a Python / ctypes loop costs more per call than the 10-20 us kernel)
static void* anymdp_tok_hand_fn(const xv_anymdp* h) {      // the HAND instantiation of the cooperative token kernel for this handle
  const bool pair = h->d_obs > 1;
  return h->a.bfmt == 1 ? (pair ? reinterpret_cast<void*>(&anymdp_tok_step_coop_kernel<false, 1, true, true>)
                                : reinterpret_cast<void*>(&anymdp_tok_step_coop_kernel<false, 1, false, true>))
                        : (pair ? reinterpret_cast<void*>(&anymdp_tok_step_coop_kernel<false, 2, true, true>)
                                : reinterpret_cast<void*>(&anymdp_tok_step_coop_kernel<false, 2, false, true>));
}

// the two cycle graphs of the overlapped token step: HAND instantiations of the cooperative kernel, ring slots q, q + 2, ...
// -> cycles per graph (built / reused), 0: this call is not overlapped, -1: failure
static int anymdp_tok_pipe_graphs(xv_anymdp* h, int D, int period, int cycles, const int32_t* action, int32_t* obs, float* reward,
                                  float* reward_gt, uint8_t* terminated, uint8_t* truncated, int32_t* final_obs, int mode) {
  const void* ptrs[7] = {action, obs, reward, reward_gt, terminated, truncated, final_obs};
  auto& K = h->tpipe_key;
  const bool same = h->tgraph_exec[0] && h->tgraph_exec[1] && K.period == period && K.mode == mode && K.fmt == h->a.bfmt &&
                    K.d_obs == h->d_obs && K.d_act == h->d_act && K.bucket == (const void*)h->a.bucket &&
                    K.obs_bucket == (const void*)h->obs_bucket && K.seed == h->eng->seed && K.gid_base == h->eng->env_id_base &&
                    memcmp(K.ptrs, ptrs, sizeof(ptrs)) == 0;
  const bool same_d = same && h->gate.depth == D;
  const int U = xv_pipe_pick_unroll(period, cycles, same_d ? h->gate.unroll[1] : 0, D);
  if (U == 0) return 0;
  if (same_d && U == h->gate.unroll[1]) return U;
  (void)hipStreamSynchronize(h->side);
  for (int i = 0; i < 2; ++i) if (h->gate.side_n[i]) (void)hipStreamSynchronize(h->gate.side_n[i]);
  (void)hipStreamSynchronize(h->eng->stream);
  anymdp_pipe_drop_graphs(h);
  if (D > 2 && !anymdp_pipe_setup_deep(h, D)) return -1;
  if (!xv_pipe_gate_sync(&h->gate)) return -1;
  void* fn = anymdp_tok_hand_fn(h);
  const size_t n = (size_t)h->a.n_env, da = (size_t)h->d_act, dob = (size_t)h->d_obs;
  const dim3 grid(xv_div_up(h->a.n_env, 256)), block(256);
  for (int q = 0; q < D; ++q) {
    hipGraph_t* gr = q < 2 ? &h->tgraph[q] : &h->gate.graph_n[q - 2];
    hipGraphExec_t* ge = q < 2 ? &h->tgraph_exec[q] : &h->gate.exec_n[q - 2];
    if (hipGraphCreate(gr, 0) != hipSuccess) return -1;
    hipGraphNode_t prev = nullptr;
    if (!anymdp_add_head_node(h, *gr, &prev, q, period, U)) return -1;
    for (int g = q; g < U * period; g += D) {      // step g of the graph set: stream g % D, ring slot g % period, tick base + g
      const int j = g % period;
      AnyMDPArgs a = h->a;
      a.seed = h->eng->seed; a.gid_base = h->eng->env_id_base;
      a.tick = (uint64_t)g; a.tick_dev = h->d_ptick + q;
      AnyMDPTokArgs KA{h->obs_cdf, h->n_obs, h->d_obs, h->d_act, h->obs_bucket};
      const size_t o = (size_t)j * n;
      AnyMDPTokIO io{action + o * da, nullptr, nullptr, nullptr, nullptr, nullptr, obs + o * dob, reward + o, reward_gt + o,
                     terminated + o, truncated + o, final_obs ? final_obs + o * dob : nullptr};
      int md = mode;
      void* params[] = {&a, &KA, &io, &md};
      hipKernelNodeParams np;
      memset(&np, 0, sizeof(np));
      np.func = fn; np.gridDim = grid; np.blockDim = block; np.kernelParams = params;
      hipGraphNode_t node;
      if (hipGraphAddKernelNode(&node, *gr, prev ? &prev : nullptr, prev ? 1 : 0, &np) != hipSuccess) return -1;
      prev = node;
    }
    if (hipGraphInstantiate(ge, *gr, nullptr, nullptr, 0) != hipSuccess) { *ge = nullptr; return -1; }
  }
  h->gate.depth = D;
  K.period = period; K.mode = mode; K.fmt = h->a.bfmt; K.d_obs = h->d_obs; K.d_act = h->d_act;
  K.bucket = (const void*)h->a.bucket; K.obs_bucket = (const void*)h->obs_bucket;
  K.seed = h->eng->seed; K.gid_base = h->eng->env_id_base;
  memcpy(K.ptrs, ptrs, sizeof(ptrs));
  h->gate.unroll[1] = U;
  return U;
}

// n_steps token steps issued from C over ring buffers: step k reads actions slot k % period ([period][n_env][d_act]) and
// writes slot k % period of the outputs (obs / final_obs [period][n_env][d_obs], the others [period][n_env]); equals
// n_steps calls of xv_anymdp_step_tokens (a Python / ctypes loop costs more per call than the 10-20 us kernel).
// With xv_anymdp_set_step_many_overlap on, whole cycles of an even period (calls of >= 64 steps, cooperative kernel, host
// tick) are issued on two streams with the hand-off through the env records, as xv_anymdp_step_many's.
extern "C" int xv_anymdp_step_tokens_many(xv_anymdp* h, int n_steps, int period, const int32_t* action, int32_t* obs,
                                          float* reward, float* reward_gt, uint8_t* terminated, uint8_t* truncated,
                                          int32_t* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && h->obs_cdf && n_steps > 0 && period > 0);
  XV_CHECK_ARG(action && obs && reward && reward_gt && terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  const size_t n = (size_t)h->a.n_env, da = (size_t)h->d_act, dob = (size_t)h->d_obs;
  int k = 0;
  h->pipe_used_last = false;
  const int cycles = n_steps / period;
  if (h->overlap && !h->eng->dev_tick && !h->pipe_failed && anymdp_tok_coop(h) && cycles > 0 && period % 2 == 0 &&
      n_steps >= XV_ANYMDP_PIPE_GRAPH_MIN && !xv_pipe_backoff_step(&h->backoff, &h->gate) && hipSetDevice(h->eng->device) == hipSuccess &&
      xv_pipe_choose_depth(anymdp_tok_hand_fn(h), 256, (size_t)xv_div_up(h->a.n_env, 256), h->eng->device) >= 2) {
    int D = xv_pipe_choose_depth(anymdp_tok_hand_fn(h), 256, (size_t)xv_div_up(h->a.n_env, 256), h->eng->device);
    if (D > period) D = period;      // the steps in flight write distinct ring slots
    if (D > 2 && xv_pipe_pick_unroll(period, cycles, 0, D) == 0) D = 2;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(h->eng->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    if (!capturing) {
      const bool set_up = anymdp_pipe_setup(h);
      if (set_up && D > 2 && !anymdp_pipe_setup_deep(h, D)) { (void)hipGetLastError(); D = 2; }
      const int U = set_up ? anymdp_tok_pipe_graphs(h, D, period, cycles, action, obs, reward, reward_gt, terminated,
                                                    truncated, final_obs, autoreset_mode) : -1;
      if (U > 0) {
        AnyMDPReplay rp;
        memset(&rp, 0, sizeof(rp));
        rp.ring_period = period; rp.mode = autoreset_mode; rp.eff = 0; rp.tokens = true;
        rp.tio = AnyMDPTokIO{action, nullptr, nullptr, nullptr, nullptr, nullptr, obs, reward, reward_gt, terminated, truncated, final_obs};
        const int rc = anymdp_pipe_launch(h, h->tgraph_exec, cycles / U, period * U, &k, h->gate.depth, &rp);
        if (rc != XV_OK) return rc;
      } else if (U < 0) {
        (void)hipGetLastError();
        h->pipe_failed = true;
      }
    } else {
      (void)hipGetLastError();
    }
  }
  for (; k < n_steps; ++k) {
    const size_t o = (size_t)(k % period) * n;
    anymdp_bind_rng(h, 1);
    AnyMDPTokIO io{action + o * da, nullptr, nullptr, nullptr, nullptr, nullptr, obs + o * dob, reward + o, reward_gt + o,
                   terminated + o, truncated + o, final_obs ? final_obs + o * dob : nullptr};
    const int rc = anymdp_tok_launch_step<false>(h, io, autoreset_mode);
    if (rc != XV_OK) return rc;
  }
  return XV_OK;
}

extern "C" int xv_anymdp_step_tokens_injected(xv_anymdp* h, const int32_t* action, const double* u, const float* z,
                                              const double* u_obs, const double* u_reset, const double* u_obs_reset,
                                              int32_t* obs, float* reward, float* reward_gt, uint8_t* terminated,
                                              uint8_t* truncated, int32_t* final_obs, int autoreset_mode) {
  XV_CHECK_ARG(h && h->obs_cdf && action && u && z && u_obs && u_reset && u_obs_reset && obs && reward && reward_gt &&
               terminated && truncated);
  XV_CHECK_ARG(autoreset_mode >= 0 && autoreset_mode <= 2);
  anymdp_bind_rng(h, 0);
  AnyMDPTokIO io{action, u, z, u_obs, u_reset, u_obs_reset, obs, reward, reward_gt, terminated, truncated, final_obs};
  return anymdp_tok_launch_step<true>(h, io, autoreset_mode);
}

extern "C" int xv_anymdp_reset_tokens(xv_anymdp* h, const uint8_t* mask, int32_t* obs) {
  XV_CHECK_ARG(h && h->obs_cdf);
  anymdp_bind_rng(h, 1);
  AnyMDPTokArgs K{h->obs_cdf, h->n_obs, h->d_obs, h->d_act, h->obs_bucket};
  AnyMDPTokIO io{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, obs, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(anymdp_tok_reset_kernel<false>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                     h->a, K, io, mask);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_anymdp_reset_tokens_injected(xv_anymdp* h, const uint8_t* mask, const double* u_reset,
                                               const double* u_obs_reset, int32_t* obs) {
  XV_CHECK_ARG(h && h->obs_cdf && u_reset && u_obs_reset);
  anymdp_bind_rng(h, 0);
  AnyMDPTokArgs K{h->obs_cdf, h->n_obs, h->d_obs, h->d_act, h->obs_bucket};
  AnyMDPTokIO io{nullptr, nullptr, nullptr, nullptr, u_reset, u_obs_reset, obs, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(anymdp_tok_reset_kernel<true>, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream,
                     h->a, K, io, mask);
  XV_LAUNCH_CHECK();
  return XV_OK;
}
#endif   // XV_KERNELS_ONLY
