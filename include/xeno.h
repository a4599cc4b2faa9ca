/* xeno.h — C-ABI of libxeno_hip.so, the MI355X batched environment-step engine.
 *
 * The reference (FutureAGI/Xenoverse) is pure Python and has no FFI: its boundary for this path is the
 * duck-typed trio  env.set_task(task_dict) -> env.reset() -> env.step(action)  on one environment
 * object per instance.  This header is the batched C form of that trio: one handle
 * holds N environment instances of one family, `*_create` is `set_task` for the whole batch, `*_reset`
 * and `*_step` advance all N at once on the GPU.  Each entry point cites the reference lines whose
 * results it reproduces.  The Python classes in xenoverse_amd/ bind these symbols with ctypes and put
 * the gymnasium VectorEnv surface on top (INTEGRATION.md shows the binding).
 *
 * Conventions
 *  - every symbol is extern "C"; no C++ or torch type crosses the boundary;
 *  - return value 0 = XV_OK, negative = error; text via xv_last_error() (thread-local);
 *  - every array argument is a CALLER-OWNED DEVICE pointer (e.g. a PyTorch-ROCm tensor's data_ptr())
 *    that must stay alive until the handle is destroyed (task tables) or the call's work has completed
 *    on the stream (per-step arrays).  Table pointers are borrowed, not copied: 32 GiB of transition
 *    tables are never duplicated;
 *  - every call is asynchronous and ordered on the engine's HIP stream; the caller synchronises
 *    (xv_engine_sync, or its own stream/event);
 *  - a handle is not thread-safe; distinct engines are independent (there is no global RNG state,
 *    unlike the reference's process-wide numpy.random stream);
 *  - device-side range errors (action out of range, stepping a terminated env with auto-reset off)
 *    do not trap: they set bits in a sticky per-engine error word read with xv_engine_error_flags.
 */
#ifndef XENO_H_
#define XENO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: AnyMDP rows are records of 128-byte lines (fence line + 7-entry blocks), completed in place by
 *    xv_anymdp_create; xv_maze_tables carries the texture-library sizes; Acrobot family; command-table and
 *    graph-replay switches
 * 4: xv_linds_rollout, xv_cartpole_rollout, xv_acrobot_rollout; the maze teachers (xv_maze_agent_*);
 *    xv_maze_set_typing; AnyMDP bucket search (xv_anymdp_build_buckets)
 * 5-7: xv_linds_step_many, xv_mixed_step[_many], device observation-model sampler, xv_rccl_comm_count, engine timing events,
 *    xv_anymdp_step_tokens_many
 * 8: device tick (xv_engine_set_device_tick / _tick_batch / _set_stream: a step is capturable in a hipGraph); AnyMDP bucket
 *    lines list chosen cuts of the row's CDF, with a census (xv_anymdp_probe_buckets, xv_anymdp_bucket_census_get,
 *    xv_anymdp_effective_search, xv_anymdp_token_kernel); xv_anymdp_step_info / xv_linds_step_info (steps and the done mask
 *    from the step launch); xv_mixed_supported; xv_anymdp_sample_tasks up to 256 states; xv_maze_set_precision accepts
 *    XV_MAZE_FILTER_EXACT_DIRECT (a new value of an existing argument: no bump)
 * 9: xv_anymdp_step_tokens_info (the POMDP / multi-token step writes steps and the done mask itself); xv_cartpole_step_info,
 *    xv_acrobot_step_info (the done mask from the step launch); xv_maze_set_raycast_mapping
 * 10: overlapped step_many (xv_anymdp_set_step_many_overlap / _overlap_state), sub-batch views (xv_anymdp_view,
 *    xv_anymdp_step_many_chains), xv_anymdp_build_rows, xv_pack_rollout_f32 / xv_unpack_rollout_f32, XV_DEVERR_HANDOFF
 * 11: the overlap switch also covers xv_mixed_step_many; xv_mixed_step_many_overlap_state; xv_engine_probe_side_streams
 * 12: an expired hand-off of an overlapped xv_anymdp_step_many / xv_mixed_step_many is repaired (the call is replayed from its
 *    entry state on one stream; the *_overlap_state functions return -2 for such a call); every cycle graph starts with the
 *    cycle gate; launches in flight are sized against 3/4 of the device, two at most beside an RCCL communicator;
 *    xv_mixed_io grew five nullable outputs (steps / done masks from the fused launch) */
#define XV_ABI_VERSION 12

/* return codes */
#define XV_OK 0
#define XV_ERR_INVALID (-1)  /* bad argument (null pointer, size out of the supported range) */
#define XV_ERR_HIP (-2)      /* a HIP runtime call failed; xv_last_error() has hipGetErrorString */
#define XV_ERR_UNSUPPORTED (-3)
#define XV_ERR_NOMEM (-4)

/* auto-reset modes (gymnasium >= 1.1 AutoresetMode).  The reference never
 * auto-resets, which is XV_AUTORESET_DISABLED. */
#define XV_AUTORESET_DISABLED 0
#define XV_AUTORESET_NEXT_STEP 1
#define XV_AUTORESET_SAME_STEP 2

/* sticky device error bits (xv_engine_error_flags) */
#define XV_DEVERR_ACTION_RANGE 1u   /* reference: `assert action < self.na` (anymdp_env.py:97) */
#define XV_DEVERR_STEP_TERMINAL 2u  /* reference: raise "given an terminated state" (anymdp_env.py:95-96) */
#define XV_DEVERR_NONFINITE 4u      /* a float state left the finite range */
#define XV_DEVERR_HANDOFF 8u        /* overlapped xv_anymdp_step_many: a wave's bounded wait for the step before it expired */

#define XV_STREAM_OWN ((void*)(intptr_t)-1)

typedef struct xv_engine xv_engine;
typedef struct xv_anymdp xv_anymdp;
typedef struct xv_linds xv_linds;
typedef struct xv_cartpole xv_cartpole;
typedef struct xv_acrobot xv_acrobot;
typedef struct xv_maze xv_maze;

/* ------------------------------------------------------------------------------------------------
 * engine: one device + one stream + one Philox key
 * ---------------------------------------------------------------------------------------------- */
int xv_abi_version(void);
const char* xv_last_error(void);

/* `hip_stream` is a hipStream_t: NULL is the device's default (null) stream — what PyTorch uses unless told
 * otherwise — and XV_STREAM_OWN makes the engine create (and own) a non-blocking stream.
 * `seed` is the Philox4x32-10 key; `env_id_base` is added to the local env index to form the counter's
 * env word, so that a batch sharded over ranks draws the same numbers as the unsharded batch
 * (SURVEY.md §8(e)).  Replaces the reference's per-reset reseeding of numpy's global RandomState from
 * OS entropy (anymdp_env.py:87, linds_env.py:114, utils/random_nn.py:9-16). */
int xv_engine_create(int device, uint64_t seed, uint64_t env_id_base, void* hip_stream, xv_engine** out);
int xv_engine_destroy(xv_engine* e);
int xv_engine_sync(xv_engine* e);
void* xv_engine_stream(xv_engine* e);
/* reads (and with clear!=0 resets) the sticky device error word; synchronises the stream */
int xv_engine_error_flags(xv_engine* e, int clear, uint32_t* out_flags);
/* the launch counter that forms the Philox counter's tick word; get/set makes runs resumable */
int xv_engine_get_tick(xv_engine* e, uint64_t* out_tick);
int xv_engine_set_tick(xv_engine* e, uint64_t tick);
/* Device tick mode (ABI 8; what makes a step CAPTURABLE in a hipGraph / torch.cuda.graph).  By default the launch tick is a
 * host counter handed to every stochastic kernel as an argument: a captured launch would replay the same draws for ever.
 * With the device tick on, the counter lives in device memory: every stochastic entry point first launches a one-thread
 * kernel that advances it and then its own kernel, which reads it — nothing host-side remains in the arguments, no
 * allocation and no synchronisation happen in a step call, and a replayed capture draws fresh numbers exactly as the same
 * calls issued eagerly would (same Philox counters: bit-identical trajectories, tested).  xv_engine_get_tick reads the
 * device word back (synchronises) in this mode; xv_anymdp_step_many issues plain launches.  The policy -> step loop the
 * reference runs per env object (anymdp/test_utils.py:45-57) becomes one graph replay per vector step (or per `unroll`
 * steps): xenoverse_amd/capture.py. */
int xv_engine_set_device_tick(xv_engine* e, int on);
int xv_engine_device_tick(xv_engine* e);          /* 1 on, 0 off */
/* device tick mode, for a capture that unrolls several steps: between (e, 1) and (e, 0) stochastic launches read
 * *tick + 0, + 1, + 2, ... and advance nothing; (e, 0) advances the word once by what the batch consumed — one tick
 * node per unrolled capture instead of one per step.  Same draws as without the batch. */
int xv_engine_tick_batch(xv_engine* e, int on);
/* re-points an engine created on a caller's stream at another stream of the same device (stream capture runs on a side
 * stream: the launches of a captured step must go there).  XV_ERR_UNSUPPORTED for an engine that owns its stream. */
int xv_engine_set_stream(xv_engine* e, void* hip_stream);
/* Two timing events on the engine's own stream (slot 0 = start, 1 = stop), for callers that time a burst of launches
   without a second event API: record is stream-ordered and asynchronous; done polls without blocking (*done = 1 once
   everything ahead of the record has finished); elapsed needs both slots finished (XV_ERR_HIP otherwise). */
int xv_engine_event_record(xv_engine* e, int slot);
int xv_engine_event_done(xv_engine* e, int slot, int* done);
int xv_engine_event_elapsed_ms(xv_engine* e, float* ms);
/* Diagnostic for the overlapped step_many paths: the side-stream candidates tried beside this engine's stream, in order (a
   ping-pong of chained one-thread launches over the two streams against the same chain on one stream, us per launch; a
   negative two_stream_us: a bounded wait expired, the streams did not run side by side).  accepted[i] = 1 for the
   candidate the overlapped paths would keep (the trial stops there).  Rows up to max_rows (6 are tried at most). */
int xv_engine_probe_side_streams(xv_engine* e, int max_rows, int* priority, float* two_stream_us, float* one_stream_us,
                                 int* accepted, int* n_rows);

/* Philox4x32-10 known-answer hook: fills out[4*n] on the device from ctr[4*n], key[2] (device ptrs). */
int xv_philox4x32_10(xv_engine* e, const uint32_t* ctr, const uint32_t* key, uint32_t* out, int n);

/* Rollout records for the trajectory all-gather (the one exchange step of the path): an AnyMDP env-step packed into ONE
 * 8-byte word — bits 0-15 observation id, 16-23 action, 24 terminated, 25 truncated, 32-63 reward (fp32 bits).
 * n records; arrays are device pointers; launched on `hip_stream` (NULL = the default stream).  The reference
 * counterpart is the per-step tuple its rollout loops append (anymdp/test_utils.py:42-60). */
int xv_pack_rollout(void* hip_stream, size_t n, const int32_t* obs, const int32_t* action, const float* reward,
                    const uint8_t* terminated, const uint8_t* truncated, uint64_t* out);
int xv_unpack_rollout(void* hip_stream, size_t n, const uint64_t* rec, int32_t* obs, int32_t* action, float* reward,
                      uint8_t* terminated, uint8_t* truncated);
/* The same for the families whose observation is a float vector (ABI 10; the mixed batch of BASELINE config 5 gathers all
 * three families): one env-step = obs_dim + 2 32-bit words — obs_dim fp32 observation values (LinDS: the 16 padded outputs
 * linds_env.py:83-91 returns; CartPole: the 4 state values of random_cartpole.py:52-61), the reward (fp32 bits), and a flag
 * word: bit 0 terminated, bit 1 truncated, bits 8-31 the discrete action (`action` nullable: 0).  n records. */
int xv_pack_rollout_f32(void* hip_stream, size_t n, int obs_dim, const float* obs, const float* reward,
                        const uint8_t* terminated, const uint8_t* truncated, const int32_t* action, uint32_t* out);
int xv_unpack_rollout_f32(void* hip_stream, size_t n, int obs_dim, const uint32_t* rec, float* obs, float* reward,
                          uint8_t* terminated, uint8_t* truncated, int32_t* action);

/* The all-gather of finished rollout chunks over RCCL directly (no torch.distributed needed):
 * every rank contributes `bytes_per_rank` bytes at `local` and receives all ranks' chunks, rank-major, at `global`
 * (device pointers; ncclAllGather of a uint8 payload on the engine's stream, asynchronous like every other call).
 * Communicator: rank 0 calls xv_rccl_unique_id (128 bytes, host buffer) and hands the id to the other ranks by any
 * channel; every rank then calls xv_rccl_comm_create (collective).  librccl.so is opened on first use: on a host
 * without it these entry points return XV_ERR_UNSUPPORTED and everything else works.  The reference has no counterpart
 * (one env object per process, no multi-device path); stepping itself never communicates. */
#define XV_RCCL_ID_BYTES 128
int xv_rccl_unique_id(void* out128);
int xv_rccl_comm_create(xv_engine* e, int world, int rank, const void* id128, void** comm_out);
int xv_rccl_comm_destroy(void* comm);
int xv_rccl_comm_count(void* comm, int* count_out);   /* ncclCommCount: how many ranks the communicator really spans */
int xv_rollout_allgather(xv_engine* e, void* rccl_comm, const void* local, void* global, size_t bytes_per_rank);

/* ------------------------------------------------------------------------------------------------
 * AnyMDP — reference: xenoverse/anymdp/anymdp_env.py
 *   set_task  :32-79   -> xv_anymdp_create   (tables prepared host-side: see xenoverse_amd/anymdp)
 *   reset     :81-90   -> xv_anymdp_reset
 *   step      :112-132 -> xv_anymdp_step / _step_injected   (single_step :92-110, get_observation :145-159)
 *
 * Table layout (device memory, borrowed for the life of the handle):
 *   rows       [n_task][S][A] row records of XV_ANYMDP_ROW_LINES(S) = 1 + NB lines of 128 bytes; NB = ceil(S/7) rounded
 *              up to a multiple of G = ceil(ceil(S/7)/16) (G = 1 up to S = 112, 2 up to 224, 3 up to 336, 4 up to 448, 5 up to 512).
 *                line 0      FENCE: double[16]; fence[k] = the CDF entry of the last next-state of block group k
 *                            (G consecutive blocks) for k < NB/G - 1, 2.0 beyond.  -- written by xv_anymdp_create --
 *                line 1 + k  BLOCK k: 7 entries of 16 bytes {double cdf; float reward; float reward_noise} for next
 *                            states 7k..7k+6, then 16 bytes {uint16 obs[7]; uint8 term_bits; uint8 0}: observation id
 *                            and terminal flag of the same next states.  -- metadata written by xv_anymdp_create --
 *              cdf = entries of the inclusive CDF of transition[s,a,:], i.e. cumsum(row)/cumsum(row)[-1] computed on
 *              the host in fp64 exactly as numpy.random.choice does (entries >= S hold 2.0; xv_anymdp_create gives them the reward pair of entry S-1, what a clamped s' gets;
 *              rows of terminal states, all-zero in the reference, hold 1.0); reward pair = {reward[s,a,s'],
 *              reward_noise[s,a,s']}.  The caller fills the ENTRIES (xenoverse_amd.anymdp.tables.to_blocked builds them
 *              from the reference's task arrays); xv_anymdp_create completes fence and metadata IN PLACE from the
 *              entries, state_map and term_mask, so the two 128-byte lines that decide s' also carry its reward,
 *              observation and termination: a step reads two table lines and follows no dependent gather.
 *   state_map  int32  [n_task][S]         inner state -> observation id (task["state_mapping"])
 *   term_mask  uint64 [n_task][(S+63)/64] bit s set <=> s in task["s_e"]
 *   s0_cdf     double [n_task][s0_max]    inclusive CDF of task["s_0_prob"] (padded with 1.0)
 *   s0_ids     int32  [n_task][s0_max]    task["s_0"] (padded by repeating the last id)
 *   max_steps  int32  [n_task]            ceil(task["max_steps"]): `steps >= max_steps` on integer steps
 *   env_task   int32  [n_env]             env -> task index
 * Supported: 2 <= S <= 512, 2 <= A <= 64, 1 <= s0_max <= 256.
 * ---------------------------------------------------------------------------------------------- */
/* blocks per row: ceil(S/7) rounded up to a multiple of G = ceil(blocks/16); lines per row: one more (the fence) */
#define XV_ANYMDP_ROW_G(S) (((((S) + 6) / 7) + 15) / 16)
#define XV_ANYMDP_ROW_BLOCKS(S) ((((S) + 6) / 7 + XV_ANYMDP_ROW_G(S) - 1) / XV_ANYMDP_ROW_G(S) * XV_ANYMDP_ROW_G(S))
#define XV_ANYMDP_ROW_LINES(S) (1 + XV_ANYMDP_ROW_BLOCKS(S))
int xv_anymdp_create(xv_engine* e, int n_env, int n_task, int S, int A, int s0_max,
                     void* rows, const int32_t* state_map,
                     const uint64_t* term_mask, const double* s0_cdf, const int32_t* s0_ids,
                     const int32_t* max_steps, const int32_t* env_task, xv_anymdp** out);
int xv_anymdp_destroy(xv_anymdp* h);

/* reset the envs whose mask byte is non-zero (mask == NULL: all).  obs[n_env] is written for reset
 * envs only.  Initial state = s0_ids[upper_bound(s0_cdf, u)] (anymdp_env.py:89), u from Philox. */
int xv_anymdp_reset(xv_anymdp* h, const uint8_t* mask, int32_t* obs);
/* parity hook: same, with the uniform draw supplied by the caller (u[n_env], fp64 in [0,1)) */
int xv_anymdp_reset_injected(xv_anymdp* h, const uint8_t* mask, const double* u, int32_t* obs);

/* one vector step.  action[n_env] in; obs, reward, reward_gt (info["reward_gt"]), terminated,
 * truncated out; final_obs (nullable) receives the pre-reset observation for SAME_STEP.
 *   steps += 1; truncated = steps >= max_steps                      (anymdp_env.py:113-114)
 *   s' = upper_bound(cdf[s,a,:], u)  == numpy.random.choice(n, p=T[s,a])   (:99-100)
 *   r_gt = R[s,a,s'];  r = r_gt + sigma[s,a,s'] * z                  (:103-105)
 *   terminated = s' in s_e                                           (:107-108)
 *   obs = state_mapping[s']                                          (:146-148)
 * u is a 53-bit uniform built from two Philox words exactly as numpy's random_sample builds it from two
 * MT19937 words; z is a Box-Muller normal from the other two words. */
int xv_anymdp_step(xv_anymdp* h, const int32_t* action, int32_t* obs, float* reward, float* reward_gt,
                   uint8_t* terminated, uint8_t* truncated, int32_t* final_obs, int autoreset_mode);
/* xv_anymdp_step that also writes, from the same launch, info["steps"] (the env's step counter after the step: 0 for an env the
 * call restarted) and the terminated | truncated mask — steps int32[n_env], done uint8[n_env], each nullable.  A Python-level
 * step() otherwise pays a second launch (xv_anymdp_get_state) and an elementwise op for them: ~8 us of host time per call. */
int xv_anymdp_step_info(xv_anymdp* h, const int32_t* action, int32_t* obs, float* reward, float* reward_gt,
                        uint8_t* terminated, uint8_t* truncated, int32_t* final_obs, int32_t* steps, uint8_t* done,
                        int autoreset_mode);
/* parity hook: the three random inputs are supplied per env: u[n_env] fp64 (transition draw),
 * z[n_env] fp32 (standard normal), u_reset[n_env] fp64 (initial-state draw used if the env resets) */
int xv_anymdp_step_injected(xv_anymdp* h, const int32_t* action, const double* u, const float* z,
                            const double* u_reset, int32_t* obs, float* reward, float* reward_gt,
                            uint8_t* terminated, uint8_t* truncated, int32_t* final_obs,
                            int autoreset_mode);

/* n_steps back-to-back vector steps, one kernel launch each, issued from C (a Python loop over
 * xv_anymdp_step is host-bound at ~6 us per launch).  Step k reads actions[k % period] and writes slot
 * k % period of the [period][n_env] output arrays — the rollout-chunk ring a learner consumes.
 * Reference counterpart: the per-step rollout loop, anymdp/test_utils.py:42-60. */
int xv_anymdp_step_many(xv_anymdp* h, int n_steps, int period, const int32_t* actions, int32_t* obs,
                        float* reward, float* reward_gt, uint8_t* terminated, uint8_t* truncated,
                        int32_t* final_obs, int autoreset_mode);
/* Whole ring cycles of xv_anymdp_step_many can be replayed from an instantiated hipGraph (period kernel nodes that
 * read the launch tick from device memory + a tick update; built on first use, rebuilt when the arrays, period or
 * mode change).  Same kernels, same results (parity-tested).  mode: 0 plain launches, 1 graph, 2 auto (default):
 * graph for n_env <= 8,192, where the stream's launch rate is the limiter (7-9 % faster at 1,024-4,096 envs), and for
 * calls of n_steps <= 128, where one submission instead of n_steps shortens the burst (20 steps of 65,536 envs: 154
 * instead of 173 us); long runs of large batches are not launch-bound and plain launches are 2-3 % faster there. */
int xv_anymdp_set_step_many_graph(xv_anymdp* h, int mode);
/* 1: the last xv_anymdp_step_many replayed the graph, 0: plain launches, -1: graph construction or launch failed (plain
 * launches used) */
int xv_anymdp_step_many_graph_state(xv_anymdp* h);
/* Overlapped xv_anymdp_step_many (ABI 10; off by default).  on = 1: whole ring cycles of an EVEN period are issued as two
 * cycle graphs — ring slots 0, 2, ... on the engine's stream, 1, 3, ... on a side stream the handle owns — or as three
 * (slots 0, 3, ... / 1, 4, ... / 2, 5, ... on three streams: three steps in flight) where three launches fit on the device
 * together and the ring holds at least three slots, with no dependency between the streams: step k + 1 is dispatched while step k runs, and each of its waves takes its 64 envs over from the
 * same wave of step k through a tag in the envs' 8-byte records (one agent-scope store hands an env on; an env depends on
 * its own previous step only, anymdp_env.py:92-132).  The drain-and-dispatch gap between two dependent launches of one stream
 * (2.7 of the 5.0 us of a 65,536-env step) is covered by the other stream.  The engine's stream waits for the side stream
 * before the call's remainder and whatever follows.  Same launch ticks, same results as plain launches.  The side stream is
 * chosen by measurement at the first overlapped call (xv_engine_probe_side_streams); each cycle's even half starts behind a
 * gate the host opens once both halves are enqueued; calls whose two launches could not be resident together take the
 * one-stream path.  A wave's wait is bounded — 2^20 polls AND 2 s of wall clock (a suspended wave does not poll: time the
 * device spends on another process does not count) — then the wave goes on and XV_DEVERR_HANDOFF is set in the engine's
 * error word.  Such a call is REPAIRED (ABI 12): its opening kernel keeps every env record and the error word as they were at
 * entry, and behind the join a replay kernel — a nearly empty launch when no hand-off expired — restores them and re-runs the
 * whole call on the engine's stream as the fused roll-out does (same ticks, every ring slot rewritten), clears the bit and
 * drops the error bits the failed attempt raised from wrong states: an expiry costs time (seconds), never results, and
 * never a hang (xv_anymdp_step_tokens_many's overlapped path likewise).  Needs the fence or bucket search and the
 * host tick; otherwise, and for odd periods, step_many behaves as without it.
 * One handle per device at a time, never a view (XV_ERR_UNSUPPORTED): two overlapped calls in flight can block each other
 * on the hardware queues their streams share.  Not inside a stream capture (the call takes the one-stream path).
 * The launches in flight are sized against 3/4 of what the empty device holds of the kernel (a neighbour kernel of the same
 * process takes slots), and while the process holds an RCCL communicator on the device at most two are in flight.
 * xv_anymdp_step_many_overlap_state: 1 the last call overlapped, 0 it did not, -1 the path failed and is no longer tried,
 * -2 the last call overlapped, a hand-off expired and the call was replayed (results are right; read it once the stream has
 * drained). */
int xv_anymdp_set_step_many_overlap(xv_anymdp* h, int on);
int xv_anymdp_step_many_overlap_state(xv_anymdp* h);

/* The device half of set_task for RAW task tensors (ABI 10): transition / reward / reward_noise, fp64 [n_task][S][A][S] as the
 * reference's task dicts hold them (anymdp/task_sampler.py:46-50, set by AnyMDPEnv.set_task, anymdp_env.py:32-79), device
 * pointers -> rows_out, the [n_task][S][A][XV_ANYMDP_ROW_LINES(S)] row records xv_anymdp_create takes (fence line and block
 * metadata zero: create completes them).  The CDF of a row is formed as numpy.random.choice forms it every step
 * (anymdp_env.py:99-100: sequential fp64 accumulate, divided by its last entry; all-zero rows -> 1.0), bit for bit what the
 * host builder (xenoverse_amd/anymdp/tables.py) produces; rewards and noise are rounded to fp32.  term_mask: [n_task][ceil(S/64)]
 * (states in s_e).  The reference's row check, (sum(row) - 1)^2 < 1e-6 unless the state is in s_e (anymdp_env.py:66-71), is
 * made on the way: *bad_row (device word, set it to ~0 first) receives the smallest failing row index row_index_base +
 * t * S * A + s * A + a (row_index_base: the first row of this call when the tasks arrive in chunks). */
int xv_anymdp_build_rows(xv_engine* e, int n_task, int S, int A, const double* transition, const double* reward,
                         const double* reward_noise, const uint64_t* term_mask, void* rows_out, unsigned long long* bad_row,
                         unsigned long long row_index_base);

/* A VIEW of envs [env_lo, env_lo + n_env) of `parent` as a handle of its own (ABI 10).  The reference steps one env object
 * at a time and its batched loop iterates independent envs (anymdp/anymdp_env.py:92-132, anymdp/test_utils.py:42-60): any
 * subset of a vector step can run on its own.  The view borrows the parent's tables, env records and bucket / observation
 * lines and launches on the stream of `e`, whose seed must equal the parent engine's and whose env_id_base must be the
 * parent's + env_lo — an env then has the same Philox counters through either handle, so a view stepped with launch tick t
 * writes exactly what the parent stepped with tick t writes for those envs.  Every entry point that takes an xv_anymdp
 * takes a view (step, step_info, step_many, rollout, reset, get/set_state, token steps ...).  While views exist the parent
 * refuses xv_anymdp_build_buckets, xv_anymdp_set_observation_model and xv_anymdp_destroy (XV_ERR_UNSUPPORTED /
 * XV_ERR_INVALID); destroy the views first.  Device errors of a view's launches land in ITS engine's error word. */
int xv_anymdp_view(xv_anymdp* parent, xv_engine* e, int env_lo, int n_env, xv_anymdp** out);
/* xv_anymdp_step_many with the envs stepped as n_views INDEPENDENT chains (n_views <= 16).  views[c] are views of `parent`
 * that tile its envs in order.  Step k of chain c waits for step k - 1 of chain c only, so the launch-to-launch gap of one
 * chain (an empty launch of a 65,536-env grid is 2.7 of the step's 5.0 us) is covered by the table lines other chains
 * have in flight.  how = 0: every chain on its view's stream (its own cycle graph, or plain launches when the parent's
 * step_many graph mode is 0); how = 1: ONE graph on the parent's stream whose branches are the chains.  Arrays are the
 * parent's [period][parent n_env] rings.  Same launch ticks, same per-env Philox counters: outputs, env records and the
 * parent's tick afterwards equal xv_anymdp_step_many's bit for bit (tests/test_gpu_chains.py).  Stream order as for one
 * call on the parent's stream: the chains start behind what that stream holds and it waits for all of them.  Needs the
 * host tick (XV_ERR_UNSUPPORTED on device-tick engines). */
int xv_anymdp_step_many_chains(xv_anymdp* parent, xv_anymdp* const* views, int n_views, int how, int n_steps, int period,
                               const int32_t* actions, int32_t* obs, float* reward, float* reward_gt, uint8_t* terminated,
                               uint8_t* truncated, int32_t* final_obs, int autoreset_mode);

/* fused rollout: T vector steps in one launch with pre-generated actions[T][n_env] (open-loop / random
 * policy data collection); outputs are [T][n_env].  Bit-identical to T calls of xv_anymdp_step with
 * SAME_STEP auto-reset.  final_obs nullable. */
int xv_anymdp_rollout(xv_anymdp* h, int T, const int32_t* actions, int32_t* obs, float* reward,
                      float* reward_gt, uint8_t* terminated, uint8_t* truncated, int32_t* final_obs);

/* POMDP / multi-token POMDP (task_type "POMDP", "MTPOMDP"; anymdp_env.py:39-44,116-128,148-157;
 * task_sampler.py:67-118).  Observations are d_obs categorical draws from observation_transition[k][state]
 * (n_obs symbols each); an action is d_act tokens applied in turn by single_step, rewards summed, stopping at
 * the first termination.  obs_cdf: double [n_task][d_obs][S][n_obs] inclusive CDF rows (host fp64, as
 * numpy.random.choice forms them).  After this call use the *_tokens entry points:
 *   action int32[n_env][d_act];  obs / final_obs int32[n_env][d_obs]
 *   injected draws: u double[d_act][n_env], z float[d_act][n_env] (one pair per token), u_obs double[d_obs][n_env]
 *   (observation after the step), u_reset double[n_env], u_obs_reset double[d_obs][n_env] (used if the env resets) */
int xv_anymdp_set_observation_model(xv_anymdp* h, int n_obs, int d_obs, int d_act, const double* obs_cdf);
int xv_anymdp_reset_tokens(xv_anymdp* h, const uint8_t* mask, int32_t* obs);
int xv_anymdp_reset_tokens_injected(xv_anymdp* h, const uint8_t* mask, const double* u_reset,
                                    const double* u_obs_reset, int32_t* obs);
int xv_anymdp_step_tokens(xv_anymdp* h, const int32_t* action, int32_t* obs, float* reward, float* reward_gt,
                          uint8_t* terminated, uint8_t* truncated, int32_t* final_obs, int autoreset_mode);
/* xv_anymdp_step_tokens that also writes info["steps"] (int32[n_env], the counter after the step) and the terminated | truncated
 * mask (uint8[n_env]) from the same launch, each nullable — as xv_anymdp_step_info for the MDP step (ABI 9) */
int xv_anymdp_step_tokens_info(xv_anymdp* h, const int32_t* action, int32_t* obs, float* reward, float* reward_gt,
                               uint8_t* terminated, uint8_t* truncated, int32_t* final_obs, int32_t* steps, uint8_t* done,
                               int autoreset_mode);
/* n_steps token steps issued from C over [period][...] ring buffers (step k on slot k % period), as xv_anymdp_step_many
 * does for the scalar step; equals n_steps calls of xv_anymdp_step_tokens. */
int xv_anymdp_step_tokens_many(xv_anymdp* h, int n_steps, int period, const int32_t* action, int32_t* obs, float* reward,
                               float* reward_gt, uint8_t* terminated, uint8_t* truncated, int32_t* final_obs,
                               int autoreset_mode);
int xv_anymdp_step_tokens_injected(xv_anymdp* h, const int32_t* action, const double* u, const float* z,
                                   const double* u_obs, const double* u_reset, const double* u_obs_reset,
                                   int32_t* obs, float* reward, float* reward_gt, uint8_t* terminated,
                                   uint8_t* truncated, int32_t* final_obs, int autoreset_mode);

/* How s' = upper_bound(cdf row, u) is evaluated; both modes return the same index (parity-tested).
 *   AUTO    FENCE when available (s0_max <= 4, observation ids < 65536, max_steps < 2^27), else BINARY
 *   BINARY  per-lane binary search over the row's CDF entries in global memory (any S), then dependent reads of
 *           the reward pair, observation id and terminal flag
 *   FENCE   the row's fence line selects the G 128-byte blocks (one for S <= 112) that are read, by 8 lanes each,
 *           coalesced, and counted with a compare + ballot + popcount; reward pair, observation id and terminal
 *           flag come out of the same block; per-env reset records replace the per-task s_0 tables */
#define XV_ANYMDP_SEARCH_AUTO 0
#define XV_ANYMDP_SEARCH_BINARY 1
#define XV_ANYMDP_SEARCH_FENCE 3
/*   BUCKET  (after xv_anymdp_build_buckets) one line in one dependent level: the row's probability axis is cut into
 *           n_bucket equal buckets and bucket line (row, k) answers every u of bucket k by itself, so the line that
 *           contains s' is named by the row and the env's own uniform, floor(u * n_bucket), with no fence read.  A line
 *           lists up to 7 CUTS of the row's CDF chosen for its bucket (6 when S > 256 or an observation id > 255): the
 *           next states a draw of the bucket can return are grouped in order, a group of one state answers exactly, a
 *           group that lumps a run of states (chosen so that the lumped probability is least: they are the ones that
 *           are practically never drawn) sends the draw's wave through the FENCE path for that step.  Identical
 *           results; costs n_task * S * A * n_bucket * 128 bytes of HBM.
 *   AUTO    (round 4) BUCKET when the lines are built and their census (below) expects no more draws per launch that a
 *           line cannot answer than `auto_limit` (0.5 when the rows miss every cache, 0.2 when the fence lines of all
 *           tasks stay cache resident: there the fence search is nearly as fast and a fall-back costs relatively more),
 *           else FENCE when available, else BINARY: xv_anymdp_effective_search tells. */
#define XV_ANYMDP_SEARCH_BUCKET 4
int xv_anymdp_set_search(xv_anymdp* h, int search);
int xv_anymdp_effective_search(xv_anymdp* h);   /* XV_ANYMDP_SEARCH_BINARY | _FENCE | _BUCKET: what a step launches now */
/* builds (n_bucket = 16 | 32 | 64) or frees (0) the engine-owned bucket lines of this handle's rows (FENCE layout, any
 * S <= 512).  XV_ERR_NOMEM when they do not fit.  Synchronises the stream (the census is read back). */
int xv_anymdp_build_buckets(xv_anymdp* h, int n_bucket);
/* What the bucket lines of these rows can and cannot answer.  p_fallback: the share of draws (uniform over the rows of
 * non-terminal states and over u) that land in a lumped group or beyond a line's last cut; one such draw costs its
 * wave — and, with one wave per SIMD, the launch — two more dependent lines.  AUTO uses the lines iff
 * fallbacks_per_launch = p_fallback * n_env <= auto_limit.  Synthetic dense bands: 0.  Rows of the reference's sampler
 * (task_sampler_utils.py:65-175), 16 buckets: ~5e-8 (the consecutive-entry lines of rounds 2-3: 2.7e-2). */
typedef struct {
  int32_t n_bucket, format;          /* format 1: 7 cuts per line (S <= 256, observation ids <= 255); 2: 6 cuts */
  int32_t cuts_per_line, built;      /* built 0: xv_anymdp_probe_buckets (nothing allocated) */
  int32_t auto_uses_bucket, reserved;
  uint64_t lines, lines_dirty, live_rows;
  double p_fallback, fallbacks_per_launch, bytes;
  double auto_limit;                 /* AUTO uses the lines iff fallbacks_per_launch <= auto_limit */
  /* POMDP / multi-token tasks: the observation bucket lines built beside them (14 cuts of obs_cdf[t][k][s][:] per line,
   * n_obs <= 256; 0 lines: not built, the per-lane token kernel serves) and the share of observation draws they cannot
   * answer (those take a per-lane search of the row) */
  uint64_t obs_lines, obs_lines_dirty;
  double obs_p_fallback;
} xv_anymdp_bucket_census;
int xv_anymdp_probe_buckets(xv_anymdp* h, int n_bucket, xv_anymdp_bucket_census* out);   /* census without the memory */
int xv_anymdp_bucket_census_get(xv_anymdp* h, xv_anymdp_bucket_census* out);             /* of the lines that are built */
/* 1: xv_anymdp_step_tokens launches the cooperative kernel (bucket search in effect, observation bucket lines built:
 * they are built beside the transition lines when they fit the free memory minus 2 GiB), 0: the per-lane kernel */
int xv_anymdp_token_kernel(xv_anymdp* h);

/* fused teacher rollout: like xv_anymdp_rollout, but the action of every step comes from a per-task greedy table
 * greedy uint8[n_task][S] (argmax_a Q[inner_state], the policy of AnyMDPSolverOpt, anymdp_solver_opt.py:38-51) and is
 * replaced by a uniform action with probability epsilon; the actions taken are written to actions_out[T][n_env]. */
int xv_anymdp_rollout_teacher(xv_anymdp* h, int T, const uint8_t* greedy, float epsilon, int32_t* actions_out,
                              int32_t* obs, float* reward, float* reward_gt, uint8_t* terminated,
                              uint8_t* truncated, int32_t* final_obs);

/* Value iteration for every task of the handle, one workgroup per task (the reference's ground-truth teacher,
 * anymdp_solver_opt.py:30-51, for whole task batches): synchronous sweeps
 *   Q[s,a] <- sum_s' T[s,a,s'] (R[s,a,s'] + gamma max_a' Q[s',a'])   until rms(Q_new - Q) <= tol or max_iter sweeps,
 * T recovered from the row records (rows of terminal states are zero), R the tables' fp32 rewards.
 * q_out double[n_task][S][A], greedy_out uint8[n_task][S] (first argmax_a, the table xv_anymdp_rollout_teacher
 * reads), iters_out int32[n_task]; each nullable, at least one of the first two given. */
int xv_anymdp_solve(xv_anymdp* h, double gamma, double tol, int max_iter, double* q_out, uint8_t* greedy_out,
                    int32_t* iters_out);

/* HOST function (host pointers, no device work): value iteration in the reference's own order of operations —
 * update_value_matrix (solver.py:57-82, max_iteration = -1): damped Gauss-Seidel sweeps over (s, a) on the value
 * matrix in place, V(s') = max_a (is_greedy) or numpy.mean_a of the CURRENT matrix, damping 1, .8, .64, .512, .5, ...,
 * until rms(old - new) <= 1e-4; fp64, every product and sum rounded separately, numpy's pairwise summation for the
 * means.  The seed-compatible task sampler (task_sampler.py:15-65, task_sampler_utils.py:177-256) feeds these values
 * back into the reward tensor, so they have to match to the bit.  t_mat, r_mat double[ns][na][ns]; vm double[ns][na]
 * holds the starting point and receives the result; sweeps_out nullable. */
int xv_anymdp_value_iteration_gs(const double* t_mat, const double* r_mat, int ns, int na, double gamma,
                                 int is_greedy, double* vm, int32_t* sweeps_out);
/* Order of the np.mean reductions inside it (process-wide): 0 = NumPy's pairwise summation — the reference run as plain
 * Python, which produced the golden fixtures (default) — 1 = one sequential loop, what a numba-compiled
 * update_value_matrix does (solver.py:57 is @njit); mode 1 is not pinned by any fixture (numba is absent here). */
int xv_anymdp_value_iteration_set_summation(int mode);

/* Task sampler on the device: AnyMDPTaskSampler's generative model and acceptance test (task_sampler.py:15-65,
 * task_sampler_utils.py:65-256, solver.py:84-148) for n_cand candidate tasks per launch, one workgroup per candidate,
 * candidate index cand_base + i.  Every random quantity is Philox4x32-10(counter = {candidate, index, purpose}, key =
 * seed): same distributions as the reference, reproducible per (seed, candidate), NOT NumPy's stream (the
 * seed-compatible sampler is host code).  Value iteration: synchronous sweeps to rms update <= 1e-4.
 * status[i]: 0 accepted | 1 terminal rewards not repairable in 5 rounds (sample_mdp returns None) | 2 value gap between
 * optimal and uniform policy < 2 | 3 long-run occupancy too concentrated (gini <= 0.70 or entropy <= 0.35) | 4 no
 * convergence.  Accepted candidates write slot i of the step engine's tables (rows / state_map / term_mask / s0_cdf /
 * s0_ids / max_steps, exactly what xv_anymdp_create takes; pass env_task entries that point at accepted slots); the
 * dense fp64 tensors and `info` (nullable) are written for every candidate.  Supported: 8 <= S <= 64 with S * A <= 512
 * (the candidate's transition tensor in registers) and S <= 256 with S * A <= 4096 (round 4: rows in a transposed global
 * scratch, allocated stream-ordered per call; term_mask then has ceil(S / 64) words per candidate). */
typedef struct {
  int32_t status, goal, n_s0, repair_rounds;
  int32_t s0[4];
  int32_t sweeps[8];       /* value-iteration sweeps: repair rounds 0..4, acceptance greedy [5] and uniform [6] */
  int32_t band_lo[256], band_hi[256], state_map[256];
  uint8_t s_e[256];
  double max_steps, gini, ent, gap_min;
  double s0_prob[4];
} xv_anymdp_cand_info;
int xv_anymdp_sample_tasks(xv_engine* e, uint64_t seed, int64_t cand_base, int n_cand, int S, int A, int s0_max,
                           void* rows, int32_t* state_map, uint64_t* term_mask, double* s0_cdf, int32_t* s0_ids,
                           int32_t* max_steps, double* transition, double* reward, double* reward_noise,
                           xv_anymdp_cand_info* info, int32_t* status);

/* env.inner_state / env.steps accessors (anymdp_env.py:138-143); device int32[n_env] each, nullable */
int xv_anymdp_get_state(xv_anymdp* h, int32_t* inner_state, int32_t* steps, uint8_t* need_reset);
int xv_anymdp_set_state(xv_anymdp* h, const int32_t* inner_state, const int32_t* steps,
                        const uint8_t* need_reset);
/* info["transition_gt"] = transition_obs[state, action] (anymdp_env.py:130) is a 8*S-byte row per env
 * and is produced on request only: out[n_env][S] double, the pmf of (current inner state, action[i])
 * scattered to observation ids. */
int xv_anymdp_transition_gt(xv_anymdp* h, const int32_t* action, double* out);

/* Synthetic task tables written directly in device memory (bench / stress configs whose tables exceed
 * host memory: 65,536 distinct tasks of S=64, A=8 are 44 GiB).  Writes the row ENTRIES (see "rows").  Deterministic in (seed, task index);
 * restated bit-for-bit by oracle/xeno_oracle.c: xo_anymdp_synth.  Band-limited rows as produced by the
 * reference sampler's sample_transition (task_sampler_utils.py:65-175), uniform weights. */
int xv_anymdp_synth_tasks(xv_engine* e, uint64_t seed, int64_t task_index_base, int n_task, int S, int A,
                          int s0_max, void* rows, int32_t* state_map, uint64_t* term_mask,
                          double* s0_cdf, int32_t* s0_ids, int32_t* max_steps);

/* ------------------------------------------------------------------------------------------------
 * LinDS — reference: xenoverse/linds/linds_env.py
 *   set_task + build_dynamics_matrices :40-76 -> xv_linds_create (ZOH discretisation done on the host in
 *                                                fp64: xenoverse_amd/linds/tables.py)
 *   reset :108-131                            -> xv_linds_reset
 *   step  :133-169 (dynamics :78-80, get_observation :83-91, get_inner_cmd :93-98,
 *                   RandomFourier.__call__ utils/random_nn.py:362-368)  -> xv_linds_step
 *
 * fp32 on the device (north_star: float dynamics within 1e-5 rel of the fp64 reference).  Batch-wide padded
 * dims: NS in {16, 32} (state), NA in {8, 16} (= pad_action_dim), NO in {16, 32} (= pad_observation_dim =
 * pad_command_dim); smaller tasks are zero-padded, which is exact.  Matrices are stored transposed (k-major).
 * The engine keeps the env state in its own layout (the accumulator fragments of the matrix kernel, csrc/linds.hip);
 * xv_linds_get_state / _set_state speak component-major float[NS][n_env] in the caller's env order.
 * ---------------------------------------------------------------------------------------------- */
#define XV_LINDS_KMAX 6
typedef struct xv_linds_tables {
  const float* phiT;        /* [n_task][NS][NS]  phiT[k][j] = Phi[j][k],  Phi = e^{A dt} */
  const float* gamT;        /* [n_task][NA][NS]  gamT[k][j] = Gamma[j][k] */
  const float* cT;          /* [n_task][NS][NO]  cT[k][j]   = ld_C[j][k] */
  const float* xt;          /* [n_task][NS]      ld_X * dt */
  const float* y0;          /* [n_task][NO]      ld_Y */
  const float* valid;       /* [n_task][NO]      target_valid as 0/1 */
  const float* cmd0;        /* [n_task][NO]      static command */
  const float* four_coef;   /* [n_task][KMAX][NO][2]  Fourier coefficients (sin, cos) */
  const double* four_omega; /* [n_task][KMAX]    Fourier orders */
  const double* four_period;/* [n_task]          RandomFourier.max_steps (1000) */
  const float* scal;        /* [n_task][8]  action_cost, reward_base, terminate_punish, reward_factor,
                                            noise_drift*dt, dt, 0, 0 */
  const int32_t* ints;      /* [n_task][4]  max_steps, target_delay, n_init, n_fourier_terms (0: static) */
  const float* init;        /* [n_task][NI][NS]  initial_states */
} xv_linds_tables;

int xv_linds_create(xv_engine* e, int n_env, int n_task, int NS, int NA, int NO, int NI,
                    const xv_linds_tables* tables, const int32_t* env_task, xv_linds** out);
int xv_linds_destroy(xv_linds* h);
/* kernel selection (results are identical, bit for bit on the state/observation path):
 *   MFMA    one wave per tile of 16 envs that share a task, x' = Phi x + Gamma a and y = C x' on
 *           v_mfma_f32_16x16x4_f32.  When the caller's env -> task map does not put 16 envs of one task side by side,
 *           xv_linds_create orders the engine's own state by task (16-slot tiles per task, the last one padded) and the
 *           kernel reaches actions / outputs through a slot -> env index: any map, same results
 *   SCALAR  one lane per env, task matrices as scalar-cache broadcast operands, waterfall over the tasks of a
 *           wave: any env -> task mapping (the independent second implementation the parity tests compare with)
 *   AUTO    MFMA */
#define XV_LINDS_PATH_AUTO 0
#define XV_LINDS_PATH_MFMA 1
#define XV_LINDS_PATH_SCALAR 2
int xv_linds_set_path(xv_linds* h, int path);
/* get_inner_cmd (linds_env.py:93-98) depends on (task, integer time) only; xv_linds_create tabulates it per task
 * ([max_steps + 2 + delay][live columns] floats — columns past the last non-zero target_valid are identically zero and not stored — with the kernels' own evaluation code, if the table fits 2 GiB) and a step
 * reads two rows instead of evaluating 2 x NO x n_fourier sin/cos pairs; likewise the observation and tracking error of
 * every initial state ([n_task][NI][NO+4] floats), read by restarting envs.  enable = 0 evaluates both directly (same
 * bits; parity-tested).  Returns XV_ERR_UNSUPPORTED when enable = 1 and no table was built. */
int xv_linds_set_command_table(xv_linds* h, int enable);
/* reset: x = initial_states[k], k uniform (linds_env.py:117 uses random.choice); obs = C x + Y; command =
 * cmd(0); error = ||(obs - cmd) * valid||.  obs/cmd float[n_env][NO], error float[n_env]; all nullable. */
int xv_linds_reset(xv_linds* h, const uint8_t* mask, float* obs, float* cmd, float* error);
int xv_linds_reset_injected(xv_linds* h, const uint8_t* mask, const int32_t* init_index, float* obs,
                            float* cmd, float* error);
/* one vector step.  action float[n_env][NA] (raw, unclipped: the action cost uses it as given, :164);
 * outputs obs/cmd float[n_env][NO] (info["command"] = cmd(steps)), reward, error (info["error"]), flags;
 * final_obs float[n_env][NO] nullable (SAME_STEP): the rows of envs that FINISH in this call receive their last
 * observation, every other row is left untouched (round 3: 64 B per env-step that ~93 % of the envs never needed) —
 * read it under terminated | truncated.  Process noise: NS Box-Muller normals per env from Philox (one call per four
 * state components' lane group: csrc/philox.h xv_box_muller16). */
int xv_linds_step(xv_linds* h, const float* action, float* obs, float* reward, uint8_t* terminated,
                  uint8_t* truncated, float* cmd, float* error, float* final_obs, int autoreset_mode);
/* n_steps vector steps issued back to back from C (the reference counterpart is the caller's loop around step(),
 * linds/test.py): step k takes its actions from slot k % period of action float[period][n_env][NA] and writes slot
 * k % period of every output ([period][n_env][NO] / [period][n_env]); equals n_steps calls of xv_linds_step. */
int xv_linds_step_many(xv_linds* h, int n_steps, int period, const float* action, float* obs, float* reward,
                       uint8_t* terminated, uint8_t* truncated, float* cmd, float* error, float* final_obs,
                       int autoreset_mode);
/* xv_linds_step that also writes, from the same launch, info["steps"] (the counter after the step: 0 for an env the call
 * restarted) and the terminated | truncated mask — steps int32[n_env], done uint8[n_env], each nullable.  Matrix kernel only
 * (XV_ERR_UNSUPPORTED on XV_LINDS_PATH_SCALAR). */
int xv_linds_step_info(xv_linds* h, const float* action, float* obs, float* reward, uint8_t* terminated, uint8_t* truncated,
                       float* cmd, float* error, float* final_obs, int32_t* steps, uint8_t* done, int autoreset_mode);
/* parity hook: z float[NS][n_env] standard normals, init_index int32[n_env] (initial state used on reset) */
int xv_linds_step_injected(xv_linds* h, const float* action, const float* z, const int32_t* init_index,
                           float* obs, float* reward, uint8_t* terminated, uint8_t* truncated, float* cmd,
                           float* error, float* final_obs, int autoreset_mode);
/* Fused roll-out: T steps in one launch (SAME_STEP auto-reset, free-running noise), the state resident in registers
 * between the steps; equals T calls of xv_linds_step bit for bit (step t draws with the tick the t-th call would use).
 * action float[T][n_env][NA]; obs float[T][n_env][NO], reward float[T][n_env], terminated / truncated uint8[T][n_env];
 * cmd, error, final_obs as in xv_linds_step with a leading T, nullable.  The reference counterpart is the loop its
 * users write around step() (linds/test.py). */
int xv_linds_rollout(xv_linds* h, int T, const float* action, float* obs, float* reward, uint8_t* terminated,
                     uint8_t* truncated, float* cmd, float* error, float* final_obs);
/* env.state accessor (:185-187): x float[NS][n_env], steps int32[n_env], need_reset uint8[n_env]; nullable */
int xv_linds_get_state(xv_linds* h, float* x, int32_t* steps, uint8_t* need_reset);
int xv_linds_set_state(xv_linds* h, const float* x, const int32_t* steps, const uint8_t* need_reset);

/* ------------------------------------------------------------------------------------------------
 * CartPole — reference: xenoverse/metacontrol/random_cartpole.py (set_task :46-50, step :52-61, reset
 * :63-75).  The physics is gymnasium's CartPoleEnv.step (third-party dependency, `gymnasium>=1.0.0` in the
 * reference's setup.py:42, not vendored and not installed here): restated from the public gymnasium 1.x
 * source equations — PARITY UNPINNED.  State double[4][n_env] (x, x_dot, theta, theta_dot) and all of the Euler update
 * in fp64, as gymnasium keeps `self.state`; only the returned observations are float32 (its np.array(..., float32)).
 *   params double[n_task][4] = gravity, masscart, masspole, length (sample_cartpole :13-29)
 *   reset_scale double[4]    = reset_bounds_scale (registered default [0.45, 0.90, 0.13, 1.0])
 *   frameskip               = physics sub-steps per step (registered default 1)
 *   max_steps               <= 0: never truncates, as the reference (no TimeLimit registered)
 * ---------------------------------------------------------------------------------------------- */
int xv_cartpole_create(xv_engine* e, int n_env, int n_task, int frameskip, int max_steps, const double* params,
                       const double* reset_scale, const int32_t* env_task, xv_cartpole** out);
int xv_cartpole_destroy(xv_cartpole* h);
int xv_cartpole_reset(xv_cartpole* h, const uint8_t* mask, float* obs /*[n_env][4]*/);
int xv_cartpole_reset_injected(xv_cartpole* h, const uint8_t* mask, const double* u /*[4][n_env] in [0,1)*/,
                               float* obs);
int xv_cartpole_step(xv_cartpole* h, const int32_t* action, float* obs, float* reward, uint8_t* terminated,
                     uint8_t* truncated, float* final_obs, int autoreset_mode);
/* the same, and the terminated | truncated mask from the same launch (done uint8[n_env], nullable; ABI 9) */
int xv_cartpole_step_info(xv_cartpole* h, const int32_t* action, float* obs, float* reward, uint8_t* terminated,
                          uint8_t* truncated, float* final_obs, uint8_t* done, int autoreset_mode);
/* T steps in one launch, the state in registers between them: action int32[T][n_env], outputs with a leading T
 * (obs float[T][n_env][4] ...); equals T calls of xv_cartpole_step (step t draws with the tick the t-th call would) */
int xv_cartpole_rollout(xv_cartpole* h, int T, const int32_t* action, float* obs, float* reward, uint8_t* terminated,
                        uint8_t* truncated, float* final_obs, int autoreset_mode);
int xv_cartpole_step_injected(xv_cartpole* h, const int32_t* action, const double* u_reset, float* obs,
                              float* reward, uint8_t* terminated, uint8_t* truncated, float* final_obs,
                              int autoreset_mode);
int xv_cartpole_get_state(xv_cartpole* h, double* state /*[4][n_env]*/, int32_t* steps, uint8_t* need_reset);
int xv_cartpole_set_state(xv_cartpole* h, const double* state, const int32_t* steps, const uint8_t* need_reset);

/* ------------------------------------------------------------------------------------------------
 * Acrobot — reference: xenoverse/metacontrol/random_acrobot.py (_dsdt :58-96, _terminal :98-101, set_task
 * :103-106, step :108-117 = `frameskip` repeats of gymnasium's AcrobotEnv.step, reset :119-130).  The integrator
 * around _dsdt (rk4 over [0, 0.2], wrap to [-pi, pi], velocity bounds 4 pi / 9 pi, torques {-1, 0, +1}, "book"
 * dynamics, no torque noise) is gymnasium's: third-party, not vendored, PARITY UNPINNED for that part; _dsdt and
 * _terminal are pinned to the reference's own code (tests/golden/acrobot_dsdt.npz).
 *   params double[n_task][7] = link_length_1, link_length_2, link_mass_1, link_mass_2, link_com_1, link_com_2,
 *                              gravity (sample_acrobot :14-39)
 *   reset_scale double[4] (device), scale_is_vector: reset_bounds_scale was given as a list (the reset state is then
 *   a float64 product) or as a scalar (float32 product, and the reset observation uses float32 cos/sin, as numpy does)
 *   fp64 state [4][n_env] = theta1, theta2, dtheta1, dtheta2; obs float[n_env][6] = cos/sin(theta1), cos/sin(theta2),
 *   dtheta1, dtheta2; action in {0, 1, 2}; reward = -1 per sub-step until the terminating one; max_steps <= 0: no
 *   truncation (the reference registers no TimeLimit).  Injected reset draws u double[4][n_env] in [0, 1).
 * ---------------------------------------------------------------------------------------------- */
int xv_acrobot_create(xv_engine* e, int n_env, int n_task, int frameskip, int max_steps, const double* params,
                      const double* reset_scale, int scale_is_vector, const int32_t* env_task, xv_acrobot** out);
int xv_acrobot_destroy(xv_acrobot* h);
int xv_acrobot_reset(xv_acrobot* h, const uint8_t* mask, float* obs /*[n_env][6]*/);
int xv_acrobot_reset_injected(xv_acrobot* h, const uint8_t* mask, const double* u /*[4][n_env] in [0,1)*/, float* obs);
int xv_acrobot_step(xv_acrobot* h, const int32_t* action, float* obs, float* reward, uint8_t* terminated,
                    uint8_t* truncated, float* final_obs, int autoreset_mode);
int xv_acrobot_step_info(xv_acrobot* h, const int32_t* action, float* obs, float* reward, uint8_t* terminated,
                         uint8_t* truncated, float* final_obs, uint8_t* done, int autoreset_mode);   /* as xv_cartpole_step_info */
/* T steps in one launch, as xv_cartpole_rollout: action int32[T][n_env], obs float[T][n_env][6] ... */
int xv_acrobot_rollout(xv_acrobot* h, int T, const int32_t* action, float* obs, float* reward, uint8_t* terminated,
                       uint8_t* truncated, float* final_obs, int autoreset_mode);
int xv_acrobot_step_injected(xv_acrobot* h, const int32_t* action, const double* u_reset, float* obs, float* reward,
                             uint8_t* terminated, uint8_t* truncated, float* final_obs, int autoreset_mode);
int xv_acrobot_get_state(xv_acrobot* h, double* state /*[4][n_env]*/, int32_t* steps, uint8_t* need_reset);
int xv_acrobot_set_state(xv_acrobot* h, const double* state, const int32_t* steps, const uint8_t* need_reset);

/* ------------------------------------------------------------------------------------------------
 * MazeWorld — reference: xenoverse/mazeworld/envs
 *   MazeBase.set_task            maze_base.py:23-52                 -> xv_maze_create (+ host tables)
 *   MazeBase.reset               maze_base.py:83-105                -> xv_maze_reset
 *   MazeWorldEnvBase.step        maze_env.py:50-66, do_action maze_continuous_3d.py:49-62,
 *     vector_move_with_collision dynamics.py:48-123,158-187, evaluation_rule maze_base.py:54-81,107-119
 *                                                                   -> xv_maze_step (move + rules kernel)
 *   update_observation           maze_continuous_3d.py:96-113, maze_view / DDA_2D / interpolate
 *                                ray_caster_utils.py:47-320         -> ray-cast kernel (frames)
 * Pose is fp64 as in the reference.  Frames are uint8[n_env][W][H][3] (the reference's (W,H,3) layout).
 * ---------------------------------------------------------------------------------------------- */
#define XV_MAZE_LMAX 15
typedef struct xv_maze_tables {
  const int8_t* walls;      /* [n_task][NG][NG] cell_walls, pitch NG (cells beyond a task's n are walls) */
  const int32_t* texts;     /* [n_task][NG][NG] cell_texts (wall texture id) */
  const int8_t* landmarks;  /* [n_task][NG][NG] cell_landmarks (-1: none) */
  const int32_t* ints;      /* [n_task][8]: n, start_i, start_j, ground_text, ceiling_text, n_landmarks, 0, 0 */
  const double* dbl;        /* [n_task][8]: cell_size, wall_height, agent_height, fol_angle, step_reward,
                                            goal_reward, collision_reward, tan(fol_angle/2) (host fp64) */
  const int32_t* commands;  /* [n_task][n_cmd] commands_sequence */
  const int32_t* lm_coord;  /* [n_task][XV_MAZE_LMAX][2] landmarks_coordinates */
  const float* tex_walls;   /* [n][256][256][3] float32 in [0,255], as MazeTaskManager loads them */
  const float* tex_grounds;
  const float* tex_ceilings;
  int32_t n_tex_walls;      /* library sizes.  When every texel is an integer in [0,255] (decoded 8-bit images, as the */
  int32_t n_tex_grounds;    /* reference's are) the engine keeps a packed RGBX-byte copy and the ray-caster reads the  */
  int32_t n_tex_ceilings;   /* four y-taps of a filter row with one 16-byte load; results are identical.              */
} xv_maze_tables;

#define XV_MAZE_ACTION_CONTINUOUS 0 /* action = double[n_env][2] (turn_rate, walk_speed) */
#define XV_MAZE_ACTION_DISCRETE16 1 /* action = int32[n_env], DEFAULT_ACTION_SPACE_16 (dynamics.py:16-27) */
#define XV_MAZE_ACTION_DISCRETE32 2 /* action = int32[n_env], DEFAULT_ACTION_SPACE_32 (dynamics.py:29-46) */

int xv_maze_create(xv_engine* e, int n_env, int n_task, int NG, int n_cmd, int max_steps, int W, int H,
                   int command_in_observation, double collision_dist, double visibility_3D,
                   const xv_maze_tables* tables, const int32_t* env_task, xv_maze** out);
int xv_maze_destroy(xv_maze* h);
/* frames (nullable) uint8[n_env][W][H][3]; command_rgb (nullable) float[n_env][3] = info["command"] */
int xv_maze_reset(xv_maze* h, const uint8_t* mask, uint8_t* frames, float* command_rgb);
/* one vector step: move + rules, then the frame of every env.  final_frames (nullable): the pre-reset frame of
 * envs that ended this step (SAME_STEP); rendering it costs a second ray-cast launch. */
int xv_maze_step(xv_maze* h, const void* action, int action_mode, uint8_t* frames, float* reward,
                 uint8_t* terminated, uint8_t* truncated, float* command_rgb, uint8_t* final_frames,
                 int autoreset_mode);
/* state accessors (device pointers, each nullable): pos double[2][n_env], ori double[n_env],
 * grid int32[2][n_env], steps/cmd_idx/cmd_age int32[n_env], need_reset uint8[n_env], collision double[n_env] */
int xv_maze_get_state(xv_maze* h, double* pos, double* ori, int32_t* grid, int32_t* steps, int32_t* cmd_idx,
                      int32_t* cmd_age, uint8_t* need_reset, double* collision);
int xv_maze_set_state(xv_maze* h, const double* pos, const double* ori, const int32_t* steps,
                      const int32_t* cmd_idx, const int32_t* cmd_age, const uint8_t* need_reset);
/* render the current state only (no step) */
int xv_maze_render(xv_maze* h, uint8_t* frames, float* command_rgb);
/* Texture filter of the ray-caster (interpolate, ray_caster_utils.py:123-140).  EXACT (default): the reference's
 * float64/float32 typing, frames identical to the oracle's.  F32: the same 4x4 distance-weighted taps evaluated in
 * float32 — opt-in, ~3x less arithmetic per pixel; frames stay within SURVEY.md M5's budget against the reference
 * (+-1 level on <= 0.5 % of the values; a value moves only where the exact colour lies within ~1e-4 of an integer). */
/* The move / collision / rules kernel (results identical; AUTO is the default): LANE_PER_ENV walks the 100 sub-steps
 * of an env in one lane; NINE_LANES / THREE_LANES give an env the lanes of its 3x3 wall neighbourhood (one cell or one
 * row of cells each), evaluate the position-independent part of all sub-steps (heading, sin / cos, displacement) in
 * parallel first and keep only the position chain sequential (dynamics.py:98-123,158-187); AUTO picks nine or three
 * lanes by the batch size.  NINE_LANES_COMPACT (what AUTO uses from 10,240 envs up): an env whose action has no walk
 * speed and whose 3x3 neighbourhood holds no wall within collision distance keeps its position through all sub-steps
 * (every displacement is +-0 and every push-out force exactly 0) — a first kernel sorts the batch into envs to walk
 * and envs that stand still, the nine-lane kernel walks the first kind and its spare workgroups finish the second
 * (heading recurrence + rules) beside them (10 of the 16
 * Discrete16 actions only turn). */
#define XV_MAZE_MOVE_LANE_PER_ENV 0
#define XV_MAZE_MOVE_NINE_LANES 1
#define XV_MAZE_MOVE_THREE_LANES 2
#define XV_MAZE_MOVE_AUTO 3
#define XV_MAZE_MOVE_NINE_LANES_COMPACT 4
int xv_maze_set_move_kernel(xv_maze* h, int kernel);
#define XV_MAZE_FILTER_EXACT 0
#define XV_MAZE_FILTER_F32 1
#define XV_MAZE_FILTER_EXACT_DIRECT 2   /* EXACT's bytes with every pixel filtered in the reference's typing directly (EXACT
                                          speculates in float64 sums and re-runs a pixel whose byte is not certain) */
int xv_maze_set_precision(xv_maze* h, int filter);
/* Which lanes paint which pixels in the ray caster (same bytes either way).  COLUMNS: a lane paints its column top to bottom.
 * ROWS: per batch of columns the lanes first work out what each column hands its pixels (wall hit, ray direction; 80 bytes per
 * column in LDS), then every wave paints 64 rows of one column at a time — wall pixels of a column share their four texture
 * rows, so the texture path sees fewer distinct lines per load.  AUTO (default): ROWS on packed (integer-valued) texture
 * libraries, COLUMNS otherwise (float-valued textures take the general path, which has no ROWS form). */
#define XV_MAZE_MAP_AUTO 0
#define XV_MAZE_MAP_COLUMNS 1
#define XV_MAZE_MAP_ROWS 2
int xv_maze_set_raycast_mapping(xv_maze* h, int mapping);
/* Which typing of the reference's ray-caster source the frames follow.  NUMPY2 (default): its @njit functions run as plain
 * Python under NumPy >= 2 (Python floats are weak, so DDA_2D and the wall-column geometry stay in the float32 of the
 * per-column tables) — the typing the golden frames were generated with.  NUMBA: the types numba infers for the same
 * source (float32 op float64 -> float64: DDA_2D and the wall-column geometry in float64).  The two differ on 0.06 % of
 * the frame values (tests/test_oracle_maze.py); NUMBA is unpinned (numba is not installed in the build container). */
#define XV_MAZE_TYPING_NUMPY2 0
#define XV_MAZE_TYPING_NUMBA 1
int xv_maze_set_typing(xv_maze* h, int typing);

/* ---------------------------------------------------------------------------------------------
 * MazeWorld rule-based teachers — reference: xenoverse/mazeworld/agents
 *   AgentBase.__init__ / update_common_info / valid_neighbors   agent_base.py:10-107
 *   SmartSLAMAgent.update_cost_map / policy / retrieve_path / exploration / navigate_landmarks_navigate /
 *     path_to_action                                            smart_slam_agent.py:105-231
 *   OracleAgent (long-term memory of ones)                      oracle_agent.py
 *   search_optimal_action                                       envs/dynamics.py:126-156
 *   maze_view's cell_exposed (what the agent remembers)         envs/ray_caster_utils.py:47-115,250-255
 * One agent per env of an xv_maze, all decisions of a batch in one launch (csrc/maze_agent.hip).  An env that starts an
 * episode (steps == 0) gets a new agent, as the reference's loops construct one after reset().
 * ------------------------------------------------------------------------------------------- */
typedef struct xv_maze_agent xv_maze_agent;
/* short_term_memory_size (3), memory_keep_ratio (1.0): the AgentBase keyword arguments; oracle_agent != 0: OracleAgent;
 * n_actions 16 | 32: the env's Discrete16 / Discrete32 table (maze_env.list_actions); keep_cost_map != 0 keeps the
 * _cost_map of the last decision readable through xv_maze_agent_get (n_env * NG * NG doubles). */
int xv_maze_agent_create(xv_maze* env, int short_term_memory_size, double memory_keep_ratio, int oracle_agent,
                         int n_actions, int keep_cost_map, xv_maze_agent** out);
int xv_maze_agent_destroy(xv_maze_agent* g);
/* agent.step(observation, r) for every env: update_common_info + policy on the env's present state.
 * action int32[n_env]: index into the Discrete table, ready for xv_maze_step.  exposed_inject: NULL = the cells are
 * exposed as maze_view does it (the W columns' DDA, each listed cell with probability 0.05, draws keyed by
 * (seed, env id, tick, column, cell)); else uint8[n_env][NG][NG] = maze_core._cell_exposed handed in (parity hook). */
int xv_maze_agent_act(xv_maze_agent* g, const uint8_t* exposed_inject, int32_t* action);
/* state of the last decision, each nullable: _mask_info uint8[n_env][NG][NG], _cost_map double[n_env][NG][NG],
 * path int32[n_env][5] = {len(path), path[0], path[1]}, cell_exposed uint8[n_env][NG][NG] */
int xv_maze_agent_get(xv_maze_agent* g, uint8_t* mask, double* cost, int32_t* path, uint8_t* exposed);

/* Observation models of AnyPOMDPTaskSampler / MultiTokensAnyPOMDPTaskSampler (task_sampler.py:78-87, :103-117) for the
 * tasks task_base .. task_base + n_task - 1 on the device: per (task, observation token) scipy.sparse.random(S, n_obs,
 * min(density, maximum_distribution / n_obs)) — exactly round(density * S * n_obs) cells without replacement, U[0,1) values
 * — with a 1 in a random column of every empty row, rows normalised, written as the inclusive row CDFs
 * obs_cdf double[n_task][d_obs][S][n_obs] that xv_anymdp_set_observation_model takes.  Counter-based draws (its own stream,
 * reproducible per (seed, task, token), independent of the batch split); the reference's distribution, not NumPy's
 * stream — the seeded host samplers give that. */
int xv_anymdp_sample_observation_model(xv_engine* e, uint64_t seed, int64_t task_base, int n_task, int S, int n_obs,
                                       int d_obs, double density, double maximum_distribution, double* obs_cdf);

/* ------------------------------------------------------------------------------------------------
 * Mixed task batch (BASELINE.json config 5: anymdp + linds + metacontrol envs in one batch; the reference steps one env
 * object per call, its users loop over heterogeneous envs in Python): ONE kernel launch advances all three families by
 * one vector step — the families' own step bodies share a grid — where three launches cost three launch latencies.
 * Results are bit for bit those of xv_anymdp_step + xv_linds_step + xv_cartpole_step on the same handles (each handle
 * keeps its own engine tick; the three engines must share one device and HIP stream).  Field meanings as in the
 * families' step calls; *_final_obs nullable.  XV_ERR_UNSUPPORTED when no fused instantiation fits the handles (AnyMDP on
 * the per-lane binary search or S > 112, LinDS on the scalar path or pads other than (16|32, 8, 16)): step separately.
 * xv_mixed_step_many: n_steps steps from C over ring buffers, step k on slot k % period of every array.
 * ---------------------------------------------------------------------------------------------- */
typedef struct xv_mixed_io {
  const int32_t* a_action; int32_t* a_obs; float* a_reward; float* a_reward_gt; uint8_t* a_terminated; uint8_t* a_truncated;
  int32_t* a_final_obs;
  const float* l_action; float* l_obs; float* l_reward; uint8_t* l_terminated; uint8_t* l_truncated; float* l_cmd;
  float* l_error; float* l_final_obs;
  const int32_t* c_action; float* c_obs; float* c_reward; uint8_t* c_terminated; uint8_t* c_truncated; float* c_final_obs;
  /* ABI 12, xv_mixed_step only (each nullable; xv_mixed_step_many ignores them): info["steps"] and the terminated | truncated
   * mask of the same step, written by the step launch itself as xv_anymdp_step_info / xv_linds_step_info / xv_cartpole_step_info
   * do — a host that wants them needs no further launch */
  int32_t* a_steps; uint8_t* a_done; int32_t* l_steps; uint8_t* l_done; uint8_t* c_done;
} xv_mixed_io;
int xv_mixed_step(xv_anymdp* a, xv_linds* l, xv_cartpole* c, const xv_mixed_io* io, int autoreset_mode);
/* 1: xv_mixed_step has a fused instantiation for these three handles as they are configured now, 0: step them separately */
int xv_mixed_supported(xv_anymdp* a, xv_linds* l, xv_cartpole* c);
int xv_mixed_step_many(xv_anymdp* a, xv_linds* l, xv_cartpole* c, const xv_mixed_io* ring, int n_steps, int period,
                       int autoreset_mode);
/* Overlapped xv_mixed_step_many: with xv_anymdp_set_step_many_overlap(a, 1) on the AnyMDP handle, calls of >= 64 steps over
 * an even period issue the ring's even slots on the engines' stream and the odd slots on a side stream (two hipGraphs), so
 * the launch of step k + 1 runs under step k; every wave takes its envs over from the same wave of the step before (AnyMDP:
 * tag in the env record; LinDS and CartPole: one word per wave).  Same launch ticks, same results, bit for bit.  Needs
 * the three handles on three engines of their own (host ticks, one stream); otherwise, and for the steps beyond the last
 * whole ring cycle, the ordinary loop runs.  Waits are bounded as for xv_anymdp_step_many, and an expired one is repaired the
 * same way (ABI 12): the three families' states and error words are kept at the call's entry and the call is replayed in one
 * launch behind the join.
 * xv_mixed_step_many_overlap_state: 1 the last call with this AnyMDP handle was overlapped, 0 it was not, -1 the overlapped
 * path failed on this device (streams do not run concurrently, graph build) and is no longer tried, -2 the last call
 * overlapped, a hand-off expired and the call was replayed (results are right). */
int xv_mixed_step_many_overlap_state(xv_anymdp* a);

#ifdef __cplusplus
}
#endif
#endif /* XENO_H_ */
