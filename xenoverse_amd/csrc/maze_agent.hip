// maze_agent.hip — the MazeWorld rule-based teacher on the device: SmartSLAMAgent / OracleAgent for every env of a batch.
//
// Reproduces xenoverse/mazeworld/agents: agent_base.py:10-107 (memory of exposed cells, valid_neighbors,
// update_common_info), smart_slam_agent.py:105-231 (update_cost_map, policy, navigate_landmarks_navigate, exploration,
// retrieve_path, path_to_action), oracle_agent.py, the 5x5 exploration convolution (smart_slam_agent.py:12-20 over
// utils/tools.py:9-34), envs/dynamics.py:126-156 (search_optimal_action) and the cell_exposed output of the ray caster
// (ray_caster_utils.py:47-115,250-255).  Checked against oracle/xeno_oracle_agent.c, which is pinned to trajectories of
// the reference agent (tests/golden/agent_*.npz).
//
// One workgroup per env, everything of one decision in LDS:
//   exposure   the W columns' DDA (the ray caster's float32 walk, no painting) marks cells in an LDS bitmap
//   memory     bitmaps in HBM: `stm_size` short-term maps as a ring + the long-term map; mask = OR of all
//   cost map   the reference relaxes cells from a FIFO queue (label correcting).  The map it ends with is the greatest
//              fixed point below the start values of  c[n] = min(c[n], min_o fl(c[o] + w(o, n)))  — with positive
//              weights and a rounded addition that is monotone in c[o] this fixed point does not depend on the order
//              of relaxations, so here all cells pull from their 8 neighbours in parallel, in place, until a sweep
//              changes nothing: the same doubles as the queue (tested bit for bit against the oracle's queue).
//   target     commanded landmark if remembered, else argmin of cost - exploration weight (first index on ties)
//   path       descent over the cost map (sequential, a few dozen cells, lane 0); only its first two cells are used
//   action     search_optimal_action: one lane per candidate action, first minimum
#include "maze_common.h"

#include <cmath>
#include <cstring>
#include <new>

#define AG_STM_MAX 8

struct AgentArgs {
  uint32_t* stm;        // [n_env][AG_STM_MAX][WORDS]
  int32_t* stm_state;   // [n_env][2]: entries held, ring head (oldest)
  uint32_t* ltm;        // [n_env][WORDS]
  uint32_t* mask;       // [n_env][WORDS]   _mask_info of the last decision
  uint32_t* exposed;    // [n_env][WORDS]   cell_exposed of the last decision
  double* cost;         // [n_env][NG*NG]   _cost_map of the last decision; nullptr unless asked for at create
  int32_t* path;        // [n_env][5]: len(path), path[0], path[1]
  int stm_size, oracle_agent, na, words;
  double keep_ratio;
  double act_cost[32];  // 1e-4 * (a0 ** 2 + a1 ** 2), host pow() as CPython's float ** 2
  uint64_t seed, gid_base, tick;
  const uint64_t* tick_dev;   // device tick mode of the engine (xv_launch_tick)
};

struct xv_maze_agent {
  xv_maze* env;
  AgentArgs a;
};

__device__ __forceinline__ bool ag_bit(const uint32_t* m, int c) { return (m[c >> 5] >> (c & 31)) & 1u; }

// class byte of a cell: bit 0 known (in _mask_info), bit 1 wall (god_info < 0)
#define AG_KNOWN 1
#define AG_WALL 2
#define AG_EFF 4     // retrieve_path's eff_targets: the agent's cell and the known cells it can step to

__device__ const int AG_NB[8][2] = {{-1, 0}, {1, 0}, {0, 1}, {0, -1}, {-1, -1}, {-1, 1}, {1, -1}, {1, 1}};   // agent_base.py:27

// agent_base.py:48-71 for one offset: may the agent go from (cx, cy) to (cx + dx, cy + dy)?
__device__ __forceinline__ bool ag_valid(const uint8_t* cls, int n, int NG, int cx, int cy, int dx, int dy, bool mask_included) {
  const int nx = cx + dx, ny = cy + dy;
  if (nx < 0 || nx >= n || ny < 0 || ny >= n) return false;
  const int c = cls[nx * NG + ny];
  if (!(c & AG_KNOWN) && !mask_included) return false;
  if ((c & AG_WALL) && (c & AG_KNOWN)) return false;
  if (dx * dy == 0) return true;
  const int a = cls[nx * NG + cy], b = cls[cx * NG + ny];
  return !(a & AG_WALL) && !(b & AG_WALL) && (a & AG_KNOWN) && (b & AG_KNOWN);
}

__global__ __launch_bounds__(256) void maze_agent_kernel(MazeArgs P, AgentArgs A, const uint8_t* exposed_inject,
                                                         int32_t* action) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int e = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
  const int NG = P.NG, G2 = NG * NG, WORDS = A.words;
  const size_t N = (size_t)P.n_env;
  const int t = P.env_task[e];
  const int32_t* in = P.T.ints + (size_t)t * 8;
  const double* db = P.T.dbl + (size_t)t * 8;
  const int n = in[0];
  const double cell_size = db[0];
  const int8_t* walls = P.T.walls + (size_t)t * G2;
  const int8_t* lmk = P.T.landmarks + (size_t)t * G2;

  double* cost = reinterpret_cast<double*>(lds);                         // [G2]
  uint32_t* ex = reinterpret_cast<uint32_t*>(cost + G2);                 // [WORDS]
  uint32_t* mk = ex + WORDS;                                             // [WORDS]
  uint8_t* cls = reinterpret_cast<uint8_t*>(mk + WORDS);                 // [G2]
  uint8_t* inm = cls + G2;                                               // [G2] bit q: the step from c - NB[q] to c is valid
  int16_t* nxt = reinterpret_cast<int16_t*>(inm + G2 + (G2 & 1));        // [G2] retrieve_path's choice at each cell, or -1
  __shared__ int sh_changed, sh_goal, sh_len, sh_p[4];
  __shared__ double sh_u[256];
  __shared__ int sh_i[256];
  __shared__ double sh_acost[32];

  const double pe0 = P.pos[e], pe1 = P.pos[N + e], ori = P.ori[e];
  const int cx = P.grid[e], cy = P.grid[N + e];
  const uint64_t gid = A.gid_base + (uint64_t)e;

  // ---- cell_exposed of the present pose: ray_caster_utils.py:47-115,250-255 ----
  for (int w = tid; w < WORDS; w += nth) ex[w] = 0u;
  __syncthreads();
  if (exposed_inject != nullptr) {
    const uint8_t* src = exposed_inject + (size_t)e * G2;
    for (int c = tid; c < G2; c += nth)
      if (src[c]) atomicOr(&ex[c >> 5], 1u << (c & 31));
  } else {
    const int W = P.W;
    const float pos0 = (float)pe0, pos1 = (float)pe1;
    const double l_focal = 0.20, half_h = db[7] * l_focal, pixel_size = 2.0 * half_h / W;
    const double pixel_factor = pixel_size / l_focal;
    const double s_ori = sin(ori), c_ori = cos(ori);
    const float cs_f = (float)cell_size, eps_f = (float)1.0e-8, vis_f = (float)P.visibility;
    const float vis06 = (float)(P.visibility * 0.60);
    for (int d_h = tid; d_h < W; d_h += nth) {
      double tan_hp = (-0.5 - W / 2.0) * pixel_factor;
      for (int q = 0; q <= d_h; ++q) tan_hp += pixel_factor;      // the reference accumulates column by column (:170-177)
      const double cos_hp = sqrt(1.0 / (1.0 + tan_hp * tan_hp));
      const double sin_hp = tan_hp * cos_hp;
      const float so = (float)(sin_hp * c_ori + cos_hp * s_ori);
      const float co = (float)(cos_hp * c_ori - sin_hp * s_ori);
      const int i0 = (int)(pos0 / cs_f), j0 = (int)(pos1 / cs_f);
      const float c_sign = co < 0 ? -1.0f : 1.0f, s_sign = so < 0 ? -1.0f : 1.0f;
      const float ddx = fabsf(co) < eps_f ? fabsf(cs_f / eps_f) : fabsf(cs_f / co);
      const float ddy = fabsf(so) < eps_f ? fabsf(cs_f / eps_f) : fabsf(cs_f / so);
      const float d_x = co > 0 ? ((float)((i0 + 1) * cell_size) - pos0) : ((float)(i0 * cell_size) - pos0);
      const float d_y = so > 0 ? ((float)((j0 + 1) * cell_size) - pos1) : ((float)(j0 * cell_size) - pos1);
      float sdx = fabsf(co) < eps_f ? c_sign * (d_x / eps_f) : d_x / co;
      float sdy = fabsf(so) < eps_f ? s_sign * (d_y / eps_f) : d_y / so;
      const int di = co > 0 ? 1 : -1, dj = so > 0 ? 1 : -1;
      int hi = i0, hj = j0, k = 0;
      float hit_dist = 0.0f;
      xv_u32x4 w4 = xv_u32x4{0, 0, 0, 0};
      auto expose = [&](int ci, int cj) {
        if ((k & 3) == 0) w4 = xv_env_draw_sub(A.seed, gid, xv_launch_tick(A.tick, A.tick_dev), XV_DRAW_EXPOSE, 64u * (uint32_t)d_h + (uint32_t)(k >> 2));
        const uint32_t w = (k & 3) == 0 ? w4.x : ((k & 3) == 1 ? w4.y : ((k & 3) == 2 ? w4.z : w4.w));
        const bool hit = (double)w * (1.0 / 4294967296.0) < 0.05;                      // random.random() < 0.05 (:254)
        if (hit && ci >= 0 && ci < n && cj >= 0 && cj < n) atomicOr(&ex[(ci * NG + cj) >> 5], 1u << ((ci * NG + cj) & 31));
        if (k < 255) ++k;
      };
      expose(i0, j0);
      while (hit_dist < vis_f) {
        const bool xstep = sdx < sdy;
        if (xstep) { hi += di; sdy -= sdx; hit_dist += sdx; }
        else { hj += dj; sdx -= sdy; hit_dist += sdy; }
        if (hi < 0 || hi >= n) { if (hj < 0 || hj >= n) break; }
        else {
          if (hit_dist <= vis06) expose(hi, hj);
          if (hj >= 0 && hj < n && walls[hi * NG + hj] > 0) break;
        }
        if (xstep) sdx = ddx; else sdy = ddy;
      }
    }
  }
  __syncthreads();

  // ---- memory: agent_base.py:73-86.  A new episode (steps == 0) starts with a new agent (:14-40) ----
  uint32_t* stm = A.stm + (size_t)e * AG_STM_MAX * WORDS;
  uint32_t* ltm = A.ltm + (size_t)e * WORDS;
  int held = A.stm_state[2 * e], head = A.stm_state[2 * e + 1];
  const bool fresh = P.steps[e] == 0;
  if (fresh) { held = 0; head = 0; }
  const bool pop = held >= A.stm_size;                 // the list grows past its size: the eldest map retires
  const int slot = pop ? head : held;
  for (int w = tid; w < WORDS; w += nth) {
    uint32_t lt = fresh ? (A.oracle_agent ? 0xFFFFFFFFu : 0u) : ltm[w];
    const uint32_t now = ex[w];
    if (pop) {
      uint32_t old = A.stm_size > 0 ? stm[(size_t)slot * WORDS + w] : now;
      if (A.keep_ratio < 1.0 && old != 0u) {           // rand(nx, ny) < memory_keep_ratio, one draw per cell (:80)
        uint32_t keep = 0u;
        for (int b = 0; b < 32; b += 4) {
          const xv_u32x4 d = xv_env_draw_sub(A.seed, gid, xv_launch_tick(A.tick, A.tick_dev), XV_DRAW_KEEP, (uint32_t)((w * 32 + b) >> 2));
          keep |= ((double)d.x * (1.0 / 4294967296.0) < A.keep_ratio ? 1u : 0u) << b;
          keep |= ((double)d.y * (1.0 / 4294967296.0) < A.keep_ratio ? 1u : 0u) << (b + 1);
          keep |= ((double)d.z * (1.0 / 4294967296.0) < A.keep_ratio ? 1u : 0u) << (b + 2);
          keep |= ((double)d.w * (1.0 / 4294967296.0) < A.keep_ratio ? 1u : 0u) << (b + 3);
        }
        old &= keep;
      }
      lt |= old;
    }
    if (A.stm_size > 0) stm[(size_t)slot * WORDS + w] = now;
    uint32_t m = lt;
    const int cnt = pop ? A.stm_size : held + 1;
    for (int q = 0; q < cnt; ++q) m |= (q == slot) ? now : stm[(size_t)q * WORDS + w];
    ltm[w] = lt;
    mk[w] = m;
    A.mask[(size_t)e * WORDS + w] = m;
    A.exposed[(size_t)e * WORDS + w] = now;
  }
  if (tid == 0) {
    A.stm_state[2 * e] = pop ? A.stm_size : held + 1;
    A.stm_state[2 * e + 1] = pop && A.stm_size > 0 ? (head + 1) % A.stm_size : head;
  }
  __syncthreads();
  for (int c = tid; c < G2; c += nth) {
    const int x = c / NG, y = c - x * NG;
    int v = 0;
    if (x < n && y < n) {
      const int god = 1 - (int)walls[c] + (int)lmk[c];      // agent_base.py:23
      v = (ag_bit(mk, c) ? AG_KNOWN : 0) | (god < 0 ? AG_WALL : 0);
    }
    cls[c] = (uint8_t)v;
    cost[c] = 1.0e+6;                                        // smart_slam_agent.py:108
  }
  if (tid < 32) sh_acost[tid] = A.act_cost[tid];
  __syncthreads();

  // ---- update_cost_map: smart_slam_agent.py:105-142 ----
  const double gf0 = pe0 / cell_size, gf1 = pe1 / cell_size;   // get_loc_grid_float, maze_base.py:225-228
  if (tid < 9) {
    const int dx = tid == 0 ? 0 : AG_NB[tid - 1][0], dy = tid == 0 ? 0 : AG_NB[tid - 1][1];
    if (tid == 0 || ag_valid(cls, n, NG, cx, cy, dx, dy, false)) {
      const int i = cx + dx, j = cy + dy;
      const double d0 = (i + 0.5) - gf0, d1 = (j + 0.5) - gf1;
      const double dist = sqrt(d0 * d0 + d1 * d1);
      const double o = 1.0 - (d0 / (dist + 1.0e-3) * cos(ori) + d1 / (dist + 1.0e-3) * sin(ori));
      const double ori_cost = 20.0 * o * fmin(dist, 0.01);
      cost[i * NG + j] = dist + ori_cost;
    }
  }
  __syncthreads();
  // edges are fixed while the costs settle: valid_neighbors(center = o) lists o -> c (agent_base.py:48-71), one bit per
  // direction; the start cells the agent can step to directly (retrieve_path's eff_targets, :177-179) get a flag
  for (int c = tid; c < G2; c += nth) {
    const int x = c / NG, y = c - x * NG;
    int m = 0;
    if (x < n && y < n)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int ox = x - AG_NB[q][0], oy = y - AG_NB[q][1];
        if (ox >= 0 && ox < n && oy >= 0 && oy < n && ag_valid(cls, n, NG, ox, oy, AG_NB[q][0], AG_NB[q][1], true)) m |= 1 << q;
      }
    inm[c] = (uint8_t)m;
  }
  __syncthreads();
  if (tid < 9) {
    const int dx = tid == 0 ? 0 : AG_NB[tid - 1][0], dy = tid == 0 ? 0 : AG_NB[tid - 1][1];
    if (tid == 0 || ag_valid(cls, n, NG, cx, cy, dx, dy, false)) cls[(cx + dx) * NG + cy + dy] |= AG_EFF;
  }
  const double W_DIAG = sqrt(2.0);
  for (int sweep = 0; sweep < 4 * G2; ++sweep) {
    if (tid == 0) sh_changed = 0;
    __syncthreads();
    bool any = false;
    for (int c = tid; c < G2; c += nth) {
      const int m = inm[c];
      if (m == 0) continue;
      const int x = c / NG, y = c - x * NG;
      const bool known = cls[c] & AG_KNOWN;
      const double w_s = known ? 1.0 : 10 + 1.0, w_d = known ? W_DIAG : 10 + W_DIAG;
      double best = cost[c];
      bool better = false;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (m & (1 << q)) {
          const double cand = cost[(x - AG_NB[q][0]) * NG + (y - AG_NB[q][1])] + (q < 4 ? w_s : w_d);
          if (best > cand) { best = cand; better = true; }
        }
      if (better) { cost[c] = best; any = true; }
    }
    if (any) sh_changed = 1;
    __syncthreads();
    if (!sh_changed) break;
    __syncthreads();
  }
  // retrieve_path's choice at every cell (:186-199): among the valid steps to cells cheaper than this one (and below 1e4),
  // the first strictly smallest in neighbour order
  for (int c = tid; c < G2; c += nth) {
    const int x = c / NG, y = c - x * NG;
    int best = -1;
    if (x < n && y < n) {
      double min_cost = cost[c];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int nx = x + AG_NB[q][0], ny = y + AG_NB[q][1];
        if (nx < 0 || nx >= n || ny < 0 || ny >= n) continue;
        if (!(inm[nx * NG + ny] & (1 << q))) continue;         // the step c -> (nx, ny) is the edge into (nx, ny) from c
        const double cv = cost[nx * NG + ny];
        if (cv > 1.0e+4) continue;
        if (cv < min_cost) { min_cost = cv; best = nx * NG + ny; }
      }
    }
    nxt[c] = (int16_t)best;
  }
  __syncthreads();

  // ---- target: navigate_landmarks_navigate (:223-230), else exploration (:213-221) ----
  const int idxc = P.cmd_idx[e] < P.n_cmd ? P.cmd_idx[e] : P.n_cmd - 1;
  const int command = P.T.commands[(size_t)t * P.n_cmd + idxc];
  if (tid == 0) sh_goal = 0x7FFFFFFF;
  __syncthreads();
  for (int c = tid; c < G2; c += nth) {
    const int x = c / NG, y = c - x * NG;
    if (x < n && y < n && (1 - (int)walls[c] + (int)lmk[c]) == command + 1 && (cls[c] & AG_KNOWN)) atomicMin(&sh_goal, c);
  }
  __syncthreads();
  int goal = sh_goal;
  if (goal == 0x7FFFFFFF) {
    double bu = 0.0;
    int bi = 0x7FFFFFFF;
    for (int c = tid; c < G2; c += nth) {
      const int x = c / NG, y = c - x * NG;
      if (x >= n || y >= n) continue;
      int unk = 0;
      for (int i = x - 2; i <= x + 2; ++i)
        for (int j = y - 2; j <= y + 2; ++j)
          if (i >= 0 && i < n && j >= 0 && j < n) unk += (cls[i * NG + j] & AG_KNOWN) ? 0 : 1;
      const double wht = (double)(unk + ((cls[c] & AG_KNOWN) ? 0 : 999));   // 5x5 ones, 1000 in the middle, zero padding
      const double u = cost[c] - wht;
      if (bi == 0x7FFFFFFF || u < bu) { bu = u; bi = c; }     // ascending c per lane: first minimum of the lane
    }
    sh_u[tid] = bu; sh_i[tid] = bi;
    __syncthreads();
    if (tid == 0) {
      double u = 0.0;
      int bi2 = 0x7FFFFFFF;
      for (int q = 0; q < nth; ++q) {
        if (sh_i[q] == 0x7FFFFFFF) continue;
        if (bi2 == 0x7FFFFFFF || sh_u[q] < u || (sh_u[q] == u && sh_i[q] < bi2)) { u = sh_u[q]; bi2 = sh_i[q]; }
      }
      sh_goal = !(u >= 0) ? bi2 : -1;                         // numpy.min(utility) >= 0 -> no target
    }
    __syncthreads();
    goal = sh_goal;
  }

  // ---- retrieve_path :171-219 (lane 0), path = [cur_grid] when there is no target (:153-154) ----
  if (tid == 0) {
    int len = 1, a0 = cx, a1 = cy, b0 = -1, b1 = -1, c0 = -1, c1 = -1;
    if (goal >= 0) {
      const int gx = goal / NG, gy = goal - gx * NG;
      a0 = gx; a1 = gy;
      int sel = goal;
      const int cur_cell = cx * NG + cy;
      while (sel != cur_cell) {
        if (cls[sel] & AG_EFF) break;                         // a cell the agent can step to directly
        const int p = nxt[sel];
        if (p < 0) break;
        sel = p;
        c0 = b0; c1 = b1; b0 = a0; b1 = a1; a0 = sel / NG; a1 = sel - a0 * NG;
        ++len;
      }
      if (len > 2) {
        const double dx = a0 + 0.5 - gf0, dy = a1 + 0.5 - gf1;
        const double ds = sqrt(dx * dx + dy * dy);
        const double dx2 = b0 + 0.5 - gf0, dy2 = b1 + 0.5 - gf1;
        const double ds2 = sqrt(dx2 * dx2 + dy2 * dy2);
        if (ds + cost[a0 * NG + a1] > ds2 + cost[b0 * NG + b1] && ds < 0.2) {   // del path[0]
          --len;
          a0 = b0; a1 = b1; b0 = c0; b1 = c1;
        }
      }
    }
    sh_len = len; sh_p[0] = a0; sh_p[1] = a1; sh_p[2] = b0; sh_p[3] = b1;
    int32_t* po = A.path + (size_t)e * 5;
    po[0] = len; po[1] = a0; po[2] = a1; po[3] = len > 1 ? b0 : -1; po[4] = len > 1 ? b1 : -1;
  }
  __syncthreads();

  // ---- path_to_action :158-169 -> search_optimal_action, dynamics.py:126-156 ----
  if (tid < A.na) {
    const double t10 = sh_p[0] + 0.5 - gf0, t11 = sh_p[1] + 0.5 - gf1;
    const bool two = sh_len > 1;
    const double t20 = sh_p[2] + 0.5 - gf0, t21 = sh_p[3] + 0.5 - gf1;
    const double a0 = A.na == 16 ? MZ_ACT16[tid][0] : MZ_ACT32[tid][0];
    const double a1 = A.na == 16 ? MZ_ACT16[tid][1] : MZ_ACT32[tid][1];
    const double tr = a0 * MZ_PI, ws = a1, dt = 1.0;
    const double d_theta = tr * dt, arc = ws * dt;
    const double c_theta = cos(ori), s_theta = sin(ori), c_dt = cos(0.5 * d_theta), s_dt = sin(0.5 * d_theta);
    const double n_ori = mz_angle_norm(ori + d_theta);
    double dx, dy;
    if (fabs(d_theta) < 1.0e-8) { dx = c_theta * arc; dy = s_theta * arc; }
    else {
      const double rad = ws / tr, offset = 2.0 * s_dt * rad;
      const double c_n = c_theta * c_dt - s_theta * s_dt, s_n = c_theta * s_dt + s_theta * c_dt;
      dx = c_n * offset; dy = s_n * offset;
    }
    const double e0 = dx - t10, e1 = dy - t11;
    const double dist_loss = e0 * e0 + e1 * e1;
    const double dist = sqrt(dist_loss);
    double cst = dist_loss;
    cst += sh_acost[tid];
    const double delta1 = mz_angle_norm(atan2(t11, t10) - n_ori);
    double delta2 = delta1;
    if (two) delta2 = mz_angle_norm(atan2(t21, t20) - n_ori);
    const double f = fmin(dist / 0.2, 1.0);
    cst += delta1 * delta1 * f + delta2 * delta2 * (1 - f);
    sh_u[tid] = cst;
  }
  __syncthreads();
  if (tid == 0) {
    int best = 0;
    for (int k = 1; k < A.na; ++k)
      if (sh_u[k] < sh_u[best]) best = k;
    action[e] = best;
  }
  if (A.cost != nullptr)
    for (int c = tid; c < G2; c += nth) A.cost[(size_t)e * G2 + c] = cost[c];
}

__global__ __launch_bounds__(256) void maze_agent_unpack_kernel(const uint32_t* bits, uint8_t* out, int n_env, int G2, int words) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_env * G2) return;
  const size_t e = i / G2;
  const int c = (int)(i - e * G2);
  out[i] = (uint8_t)((bits[e * words + (c >> 5)] >> (c & 31)) & 1u);
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" int xv_maze_agent_create(xv_maze* env, int short_term_memory_size, double memory_keep_ratio, int oracle_agent,
                                    int n_actions, int keep_cost_map, xv_maze_agent** out) {
  XV_CHECK_ARG(out != nullptr);
  *out = nullptr;
  XV_CHECK_ARG(env != nullptr && short_term_memory_size >= 0 && short_term_memory_size <= AG_STM_MAX);
  XV_CHECK_ARG(n_actions == 16 || n_actions == 32);
  XV_HIP(hipSetDevice(env->eng->device));
  xv_maze_agent* g = new (std::nothrow) xv_maze_agent();
  if (!g) {
    xv_set_error("xv_maze_agent_create: out of host memory");
    return XV_ERR_NOMEM;
  }
  g->env = env;
  AgentArgs& a = g->a;
  memset(&a, 0, sizeof(a));
  const MazeArgs& m = env->a;
  const size_t n = (size_t)m.n_env, G2 = (size_t)m.NG * m.NG;
  a.words = (int)((G2 + 31) / 32);
  a.stm_size = short_term_memory_size; a.oracle_agent = oracle_agent; a.na = n_actions; a.keep_ratio = memory_keep_ratio;
  static const double act16[16][2] = {{0.0, 0.5}, {0.05, 0.0}, {-0.05, 0.0}, {0.1, 0.0}, {-0.1, 0.0}, {0.2, 0.0},
                                      {-0.2, 0.0}, {0.3, 0.0}, {-0.3, 0.0}, {0.5, 0.0}, {-0.5, 0.0}, {0.0, 1.0},
                                      {0.05, 1.0}, {-0.05, 1.0}, {0.10, 1.0}, {-0.10, 1.0}};
  static const double act32[32][2] = {
      {0.0, 0.2}, {0.02, 0.0}, {-0.02, 0.0}, {0.05, 0.0}, {-0.05, 0.0}, {0.1, 0.0}, {-0.1, 0.0}, {0.2, 0.0},
      {-0.2, 0.0}, {0.3, 0.0}, {-0.3, 0.0}, {0.4, 0.0}, {-0.4, 0.0}, {0.5, 0.0}, {-0.5, 0.0}, {0.0, 0.5},
      {0.0, 1.0}, {0.02, 0.5}, {0.02, 1.0}, {-0.02, 0.5}, {-0.02, 1.0}, {0.05, 0.5}, {0.05, 1.0}, {-0.05, 0.5},
      {-0.05, 1.0}, {0.10, 0.5}, {0.10, 1.0}, {-0.10, 0.5}, {-0.10, 1.0}, {0.0, -0.2}, {0.1, -0.2}, {-0.1, -0.2}};
  for (int k = 0; k < n_actions; ++k) {
    const double a0 = n_actions == 16 ? act16[k][0] : act32[k][0], a1 = n_actions == 16 ? act16[k][1] : act32[k][1];
    a.act_cost[k] = 1.0e-4 * (std::pow(a0, 2.0) + std::pow(a1, 2.0));
  }
  const size_t wb = (size_t)a.words * 4;
  hipError_t r = hipMalloc(&a.stm, n * AG_STM_MAX * wb);
  if (r == hipSuccess) r = hipMalloc(&a.stm_state, n * 8);
  if (r == hipSuccess) r = hipMalloc(&a.ltm, n * wb);
  if (r == hipSuccess) r = hipMalloc(&a.mask, n * wb);
  if (r == hipSuccess) r = hipMalloc(&a.exposed, n * wb);
  if (r == hipSuccess) r = hipMalloc(&a.path, n * 20);
  if (r == hipSuccess && keep_cost_map) r = hipMalloc(&a.cost, n * G2 * 8);
  if (r == hipSuccess) r = hipMemsetAsync(a.stm, 0, n * AG_STM_MAX * wb, env->eng->stream);
  if (r == hipSuccess) r = hipMemsetAsync(a.stm_state, 0, n * 8, env->eng->stream);
  if (r == hipSuccess) r = hipMemsetAsync(a.ltm, oracle_agent ? 0xFF : 0, n * wb, env->eng->stream);
  if (r == hipSuccess) r = hipMemsetAsync(a.mask, 0, n * wb, env->eng->stream);
  if (r == hipSuccess) r = hipMemsetAsync(a.exposed, 0, n * wb, env->eng->stream);
  if (r == hipSuccess) r = hipMemsetAsync(a.path, 0, n * 20, env->eng->stream);
  if (r != hipSuccess) {
    xv_set_error("xv_maze_agent_create: device allocation failed: %s", hipGetErrorString(r));
    void* ps[] = {a.stm, a.stm_state, a.ltm, a.mask, a.exposed, a.path, a.cost};
    for (void* p : ps) if (p) (void)hipFree(p);
    delete g;
    return XV_ERR_HIP;
  }
  *out = g;
  return XV_OK;
}

extern "C" int xv_maze_agent_destroy(xv_maze_agent* g) {
  if (!g) return XV_OK;
  (void)hipSetDevice(g->env->eng->device);
  (void)hipStreamSynchronize(g->env->eng->stream);
  AgentArgs& a = g->a;
  void* ps[] = {a.stm, a.stm_state, a.ltm, a.mask, a.exposed, a.path, a.cost};
  for (void* p : ps) if (p) (void)hipFree(p);
  delete g;
  return XV_OK;
}

extern "C" int xv_maze_agent_act(xv_maze_agent* g, const uint8_t* exposed_inject, int32_t* action) {
  XV_CHECK_ARG(g != nullptr && action != nullptr);
  xv_engine* eng = g->env->eng;
  AgentArgs& a = g->a;
  a.seed = eng->seed; a.gid_base = eng->env_id_base;
  const XvTickBind tb = xv_engine_bind_tick(eng, 1);
  a.tick = tb.tick; a.tick_dev = tb.tick_dev;
  const MazeArgs& m = g->env->a;
  const size_t G2 = (size_t)m.NG * m.NG;
  const size_t lds = G2 * 8 + (size_t)a.words * 8 + 2 * G2 + 2 + 2 * G2 + 16;
  // small mazes: one wave per env (its barriers cost nothing and four times as many envs are resident per CU)
  const int threads = G2 <= 1024 ? 64 : 256;
  hipLaunchKernelGGL(maze_agent_kernel, dim3(m.n_env), dim3(threads), lds, eng->stream, m, a, exposed_inject, action);
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_maze_agent_get(xv_maze_agent* g, uint8_t* mask, double* cost, int32_t* path, uint8_t* exposed) {
  XV_CHECK_ARG(g != nullptr);
  xv_engine* eng = g->env->eng;
  const AgentArgs& a = g->a;
  const MazeArgs& m = g->env->a;
  const int G2 = m.NG * m.NG;
  const size_t tot = (size_t)m.n_env * G2;
  if (mask)
    hipLaunchKernelGGL(maze_agent_unpack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, eng->stream, a.mask, mask,
                       m.n_env, G2, a.words);
  if (exposed)
    hipLaunchKernelGGL(maze_agent_unpack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, eng->stream, a.exposed,
                       exposed, m.n_env, G2, a.words);
  if (cost) {
    if (!a.cost) {
      xv_set_error("xv_maze_agent_get: the agent was created without keep_cost_map");
      return XV_ERR_INVALID;
    }
    XV_HIP(hipMemcpyAsync(cost, a.cost, tot * 8, hipMemcpyDeviceToDevice, eng->stream));
  }
  if (path) XV_HIP(hipMemcpyAsync(path, a.path, (size_t)m.n_env * 20, hipMemcpyDeviceToDevice, eng->stream));
  XV_LAUNCH_CHECK();
  return XV_OK;
}
