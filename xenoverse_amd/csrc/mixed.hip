// mixed.hip — one launch per vector step for a MIXED task batch (BASELINE.json config 5: per GPU 16,384 anymdp + 8,192
// linds + 8,192 cartpole envs).
//
// The families' step kernels are 4-6 us each at those sizes — about the cost of launching anything — so stepping them
// one after the other costs three launch latencies per vector step (14-17 us measured), and separate streams cost more
// in cross-stream waits than the overlap returns (HISTORY.md 5.1).  Here the three families share ONE grid: workgroups
// [0, nbA) run the AnyMDP step body, [nbA, nbA + nbL) the LinDS matrix body, the rest the CartPole body — the very
// functions the families' own kernels wrap (anymdp_step_body / linds_step_mfma_body / cartpole_step_body, called with the
// workgroup's index inside its family), so every env gets bit for bit what xv_anymdp_step / xv_linds_step /
// xv_cartpole_step would give it.  Each family keeps its handle, its engine tick and its Philox stream; the handles must
// share one HIP stream.  The reference has no counterpart: it steps one env object per Python call.
#define XV_KERNELS_ONLY
#include "anymdp.hip"
#include "cartpole.hip"
#include "linds.hip"

template <int AG, int ABK, int LNS, int LNO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
void mixed_step_kernel(AnyMDPArgs A, AnyMDPStepIO aio, int nbA, LinDSArgs L, LinDSStepIO lio, int nbL, CartPoleArgs C, CartPoleIO cio,
                       int mode) {
  const int b = (int)blockIdx.x;
  if (b < nbA) {
    anymdp_step_body<false, AG, false, false, ABK>(A, aio, 1, mode, b);
  } else if (b < nbA + nbL) {
    linds_step_mfma_body<LNS, 8, LNO, false>(L, lio, mode, b - nbA);
  } else {
    cartpole_step_body<false>(C, cio, mode, 1, b - nbA - nbL);
  }
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
                   io->a_truncated, io->a_final_obs, nullptr, nullptr, 0.0f};
  LinDSStepIO lio{io->l_action, nullptr, nullptr, io->l_obs, io->l_reward, io->l_terminated, io->l_truncated, io->l_cmd,
                  io->l_error, io->l_final_obs};
  CartPoleIO cio{io->c_action, nullptr, io->c_obs, io->c_reward, io->c_terminated, io->c_truncated, io->c_final_obs};
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

// n_steps fused vector steps issued from C over ring buffers: step k uses slot k % period of every [period][...] array
extern "C" int xv_mixed_step_many(xv_anymdp* a, xv_linds* l, xv_cartpole* c, const xv_mixed_io* ring, int n_steps, int period,
                                  int autoreset_mode) {
  XV_CHECK_ARG(a && l && c && ring && n_steps > 0 && period > 0);
  const size_t na = (size_t)a->a.n_env, nl = (size_t)l->a.n_env, nc = (size_t)c->a.n_env;
  const size_t LA = (size_t)l->a.NA, LO = (size_t)l->a.NO;
  for (int k = 0; k < n_steps; ++k) {
    const size_t s = (size_t)(k % period);
    xv_mixed_io io = *ring;
    io.a_action += s * na; io.a_obs += s * na; io.a_reward += s * na; io.a_reward_gt += s * na;
    io.a_terminated += s * na; io.a_truncated += s * na;
    if (io.a_final_obs) io.a_final_obs += s * na;
    io.l_action += s * nl * LA; io.l_obs += s * nl * LO; io.l_reward += s * nl; io.l_terminated += s * nl;
    io.l_truncated += s * nl; io.l_cmd += s * nl * LO; io.l_error += s * nl;
    if (io.l_final_obs) io.l_final_obs += s * nl * LO;
    io.c_action += s * nc; io.c_obs += s * nc * 4; io.c_reward += s * nc; io.c_terminated += s * nc; io.c_truncated += s * nc;
    if (io.c_final_obs) io.c_final_obs += s * nc * 4;
    const int rc = xv_mixed_step(a, l, c, &io, autoreset_mode);
    if (rc != XV_OK) return rc;
  }
  return XV_OK;
}
