// maze.hip — MazeWorld batched move/collision + rules kernel and first-person ray-cast kernel for gfx950.
//
// Reproduces xenoverse/mazeworld/envs: dynamics.py:48-123,158-187 (unicycle arc + soft wall push-out in 100
// sub-steps of 0.01), maze_continuous_3d.py:49-62 (do_action), maze_base.py:54-119,212-223 (command / reward
// rules), ray_caster_utils.py:47-320 (DDA_2D, interpolate, maze_view).  Same operations in the same types and
// order as oracle/xeno_oracle.c (mz_*), which reproduces the reference's frames bit for bit.
//
//   move kernel : one lane per env, pose in fp64 registers.  The agent moves < 1 unit per step and cells are
//                 >= 1.5 units, so every 3x3 wall neighbourhood of the 100 sub-steps lies in the 5x5 cells
//                 around the starting cell: they are fetched ONCE into a 25-bit register mask; the sub-step
//                 loop touches no memory.
//   ray-cast    : one workgroup per env frame, one lane per screen column d_h (the reference's per-column
//                 tables become per-lane registers), ONE loop over rows d_v: each pixel is filtered once, by
//                 the stage (wall / floor / ceiling) whose store survives in the reference.  The frame is
//                 staged in LDS as bytes, in chunks of rows, and leaves with coalesced 16-byte stores; the
//                 landmark overlays read-modify-write there, as the reference does on its int32 array.
//                 Integer-valued texture libraries are read from a packed RGBX-byte copy (16 B per filter row).
#include "maze_common.h"

#include <cstring>

static const size_t MAZE_LDS_CHUNK_MAX = 50176;   // 256 columns x (64 rows x 3 + 4) bytes: three workgroups per 160-KiB CU
static inline int maze_rc_threads(int W) { return W <= 64 ? 64 : (W <= 128 ? 128 : 256); }
#define MZ_TEX_PITCH 260   // 256 texels + 3 wrapped ones (+1 pad): the 4 y-taps of a filter row never wrap
// Round 4: a second packed copy with PAIRS of filter rows interleaved — texel (x, y) at word ((x >> 1) * PITCH + y) * 2 +
// (x & 1) — so that the four y-taps of rows 2m and 2m + 1 are 32 contiguous bytes and a 4 x 4 window lies in 2 (x even) or 3
// such spans instead of 4 rows 1 KB apart.  The fp32 filter, which waits on its texture fetches, reads this copy (64 x 64:
// 1.09 -> 0.98 ms, 256 x 256: 11.95 -> 11.24 ms); the exact filter, which is bound by fp64 issue and pays for the extra
// selects (17.1 -> 18.0 ms at 256 x 256), keeps the row-major copy (profiles/r04_j_ab_tex_pairs.txt).

// ray_caster_utils.py:11-25
__device__ const float MZ_LANDMARK_RGB[XV_MAZE_LMAX][3] = {
    {0, 255, 0}, {255, 0, 0}, {0, 0, 255}, {0, 255, 255}, {255, 0, 255}, {255, 255, 0}, {128, 128, 255},
    {128, 255, 128}, {255, 128, 128}, {0, 96, 128}, {96, 0, 128}, {0, 128, 96}, {96, 128, 0}, {128, 96, 0},
    {128, 0, 96}};

// a / b for a divisor b that is reused: hipcc expands an fp64 division into div_scale / rcp / two Newton steps on
// the reciprocal / q0 = a*y / r = fma(-b,q0,a) / q = fma(r,y,q0) / div_fixup, and the scale and fixup steps only act
// on operands near the ends of the exponent range.  Hoisting the refined reciprocal y leaves three instructions per
// quotient with the same correctly rounded result (scripts/devtools/check_div.hip: 0 mismatches against hipcc's own
// division in 1.3e10 quotients over the operand ranges used here).
struct MzDivisor { double b, y; };
__device__ __forceinline__ MzDivisor mz_divisor(double b) {
  double y = __builtin_amdgcn_rcp(b);
  y = __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
  y = __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
  return {b, y};
}
__device__ __forceinline__ double mz_div(double a, const MzDivisor& r) {   // == a / r.b (see above)
  const double q0 = a * r.y;
  return __builtin_fma(__builtin_fma(-r.b, q0, a), r.y, q0);
}

// 1 / b to ~2^-40 or better (v_rcp_f64, accurate to at least 2^-20 — the ISA guide says about 2^-24 —, and ONE Newton step, which
// squares the error): for the SPECULATED filter only, whose bound has room for it (MZ_SPEC_KAPPA) — never for a value the
// reference computes by a division
__device__ __forceinline__ double mz_rcp_spec(double b) {
  const double y = __builtin_amdgcn_rcp(b);
  return __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
}

// dynamics.py:56-69
__device__ __forceinline__ double mz_nearest_point(double p0, double p1, double l10, double l11, double l20,
                                                   double l21, double& n0, double& n1) {
  double u0 = l20 - l10, u1 = l21 - l11;
  const double edge = sqrt(u0 * u0 + u1 * u1);
  const double m = edge > 1.0e-6 ? edge : 1.0e-6;
  u0 /= m; u1 /= m;
  const double d1 = (p0 - l10) * u0 + (p1 - l11) * u1;
  if (d1 > edge) { n0 = l20; n1 = l21; }
  else if (d1 < 0) { n0 = l10; n1 = l11; }
  else { n0 = l10 + d1 * u0; n1 = l11 + d1 * u1; }
  const double a = p0 - n0, b = p1 - n1;
  return sqrt(a * a + b * b);
}

// dynamics.py:71-96
__device__ __forceinline__ void mz_collision_force(double v0, double v1, double cell_size, double eff /* col_dist / cell_size */,
                                                   double& f0, double& f1) {
  const double dist = sqrt(v0 * v0 + v1 * v1);
  f0 = 0.0; f1 = 0.0;
  if (dist > 0.708 + eff) return;
  if (fabs(v0) < 0.5 && fabs(v1) < 0.5) {
    const double s = 0.50 / (dist > 1.0e-6 ? dist : 1.0e-6) * (0.708 + eff - dist) * cell_size;
    f0 = s * v0; f1 = s * v1;
    return;
  }
  const bool x_pos = v0 + v1 > 0, y_pos = v1 - v0 > 0;
  double n0, n1, d;
  if (x_pos && y_pos) d = mz_nearest_point(v0, v1, 0.5, 0.5, -0.5, 0.5, n0, n1);
  else if (!x_pos && y_pos) d = mz_nearest_point(v0, v1, -0.5, 0.5, -0.5, -0.5, n0, n1);
  else if (!x_pos && !y_pos) d = mz_nearest_point(v0, v1, -0.5, -0.5, 0.5, -0.5, n0, n1);
  else d = mz_nearest_point(v0, v1, 0.5, -0.5, 0.5, 0.5, n0, n1);
  if (eff < d) return;
  double o0 = v0 - n0, o1 = v1 - n1;
  const double on = sqrt(o0 * o0 + o1 * o1);
  const double inv = 1.0 / (on > 1.0e-6 ? on : 1.0e-6);
  o0 *= inv; o1 *= inv;
  const double s = 0.50 * (eff - d) * cell_size;
  f0 = s * o0; f1 = s * o1;
}

// mz_collision_force for a point already known to lie within eff (+1e-9) of the wall cell's box in Chebyshev
// distance.  Outside the box the reference's first test, |v| > 0.708 + eff -> 0, cannot fire there and change a result:
// a point within eff of the box is at most sqrt(0.5) + eff < 0.708 + eff from its centre, and farther points already get
// zero from the edge test — so |v| (a square root nothing else uses) is not formed.  Same values otherwise.
__device__ __forceinline__ void mz_collision_force_near(double v0, double v1, double cell_size, double eff, double& f0,
                                                        double& f1) {
  f0 = 0.0; f1 = 0.0;
  if (fabs(v0) < 0.5 && fabs(v1) < 0.5) {
    const double dist = sqrt(v0 * v0 + v1 * v1);
    if (dist > 0.708 + eff) return;
    const double s = 0.50 / (dist > 1.0e-6 ? dist : 1.0e-6) * (0.708 + eff - dist) * cell_size;
    f0 = s * v0; f1 = s * v1;
    return;
  }
  // nearest_point (dynamics.py:56-69) on the facing edge of the unit box.  The four edges differ in their end points
  // only, so the end points are selected and the arithmetic runs once for the whole wave instead of once per quadrant
  // present in it; and every edge has length sqrt(1) = 1 exactly, so the reference's normalisation `line_vec /
  // max(1e-6, edge)` returns line_vec itself: neither the square root nor the two divisions are formed.  Same values.
  const bool x_pos = v0 + v1 > 0, y_pos = v1 - v0 > 0;
  const double l10 = x_pos ? 0.5 : -0.5;                                               // x of the first end point
  const double l11 = y_pos ? 0.5 : -0.5;                                               // y of the first end point
  const double l20 = (x_pos == y_pos) ? (x_pos ? -0.5 : 0.5) : l10;                    // second end point
  const double l21 = (x_pos == y_pos) ? l11 : (x_pos ? 0.5 : -0.5);
  const double u0 = l20 - l10, u1 = l21 - l11, edge = 1.0;
  const double d1 = (v0 - l10) * u0 + (v1 - l11) * u1;
  double n0, n1;
  if (d1 > edge) { n0 = l20; n1 = l21; }
  else if (d1 < 0) { n0 = l10; n1 = l11; }
  else { n0 = l10 + d1 * u0; n1 = l11 + d1 * u1; }
  const double a = v0 - n0, b = v1 - n1;
  const double d = sqrt(a * a + b * b);
  if (eff < d) return;
  double o0 = v0 - n0, o1 = v1 - n1;
  const double on = sqrt(o0 * o0 + o1 * o1);
  const double inv = 1.0 / (on > 1.0e-6 ? on : 1.0e-6);
  o0 *= inv; o1 *= inv;
  const double s = 0.50 * (eff - d) * cell_size;
  f0 = s * o0; f1 = s * o1;
}

__device__ __forceinline__ void mz_reset_env(const MazeArgs& P, int e, int t) {   // maze_base.py:83-105
  const int32_t* in = P.T.ints + (size_t)t * 8;
  const double cs = P.T.dbl[(size_t)t * 8];
  const size_t N = (size_t)P.n_env;
  P.grid[e] = in[1]; P.grid[N + e] = in[2];
  P.pos[e] = in[1] * cs + 0.5 * cs;        // get_cell_center :215-218
  P.pos[N + e] = in[2] * cs + 0.5 * cs;
  P.ori[e] = 0.0;
  P.cmd_idx[e] = 0; P.cmd_age[e] = 0; P.steps[e] = 0; P.need_reset[e] = 0; P.collision[e] = 0.0;
}

__global__ __launch_bounds__(256) void maze_reset_kernel(MazeArgs P, const uint8_t* mask) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= P.n_env) return;
  if (mask && !mask[e]) return;
  mz_reset_env(P, e, P.env_task[e]);
}

// action -> (turn_rate, walk_speed): maze_env.py:151-162, maze_continuous_3d.py:49-52
__device__ __forceinline__ void mz_decode_action(const void* action, int action_mode, int e, double& turn_rate,
                                                 double& walk_speed, uint32_t& err) {
  double tr, ws;
  if (action_mode == XV_MAZE_ACTION_CONTINUOUS) {
    const double* a = (const double*)action;
    tr = a[2 * (size_t)e]; ws = a[2 * (size_t)e + 1];
  } else {
    int a = ((const int32_t*)action)[e];
    const int na = action_mode == XV_MAZE_ACTION_DISCRETE16 ? 16 : 32;
    if (a < 0 || a >= na) { err |= XV_DEVERR_ACTION_RANGE; a = a < 0 ? 0 : na - 1; }
    if (action_mode == XV_MAZE_ACTION_DISCRETE16) { tr = MZ_ACT16[a][0]; ws = MZ_ACT16[a][1]; }
    else { tr = MZ_ACT32[a][0]; ws = MZ_ACT32[a][1]; }
  }
  turn_rate = (tr < -1.0 ? -1.0 : (tr > 1.0 ? 1.0 : tr)) * MZ_PI;
  walk_speed = ws < -1.0 ? -1.0 : (ws > 1.0 ? 1.0 : ws);
}

// what follows the 100 sub-steps for one env: get_loc_grid (maze_base.py:220-223), evaluation_rule (:107-119), the
// state and output stores, auto-reset
__device__ __forceinline__ void mz_finish_move(const MazeArgs& P, int e, int t, double p0, double p1, double ori, double coll,
                                               uint32_t err, int mode, float* reward, uint8_t* terminated, uint8_t* truncated) {
  const size_t N = (size_t)P.n_env;
  const double* db = P.T.dbl + (size_t)t * 8;
  const double cell_size = db[0];
  if (!(fabs(p0) <= 1.0e300) || !(fabs(p1) <= 1.0e300)) err |= XV_DEVERR_NONFINITE;
  const int g0i = (int)(p0 / cell_size), g1i = (int)(p1 / cell_size);
  const int steps = P.steps[e] + 1;
  int age = P.cmd_age[e] + 1, idx = P.cmd_idx[e];
  const int cmd = P.T.commands[(size_t)t * P.n_cmd + (idx < P.n_cmd ? idx : P.n_cmd - 1)];
  const int32_t* lc = P.T.lm_coord + ((size_t)t * XV_MAZE_LMAX + cmd) * 2;
  const bool at_goal = (idx < P.n_cmd) && lc[0] == g0i && lc[1] == g1i;
  // instant_rewards is a float32 array holding goal_reward at the active command's cell (:61-69, :96)
  const float r = (at_goal ? (float)db[5] : 0.0f) + (float)db[4];
  int term = 0;
  if (at_goal || age >= 500) {   // reach_goal() or step_limits() -> refresh_command (:54-70)
    idx += 1; age = 0;
    if (idx > P.n_cmd - 1) term = 1;
  }
  const int trunc = (steps > P.max_steps - 1) ? 1 : 0;   // :212-213
  P.pos[e] = p0; P.pos[N + e] = p1; P.ori[e] = ori; P.collision[e] = coll;
  P.grid[e] = g0i; P.grid[N + e] = g1i;
  P.steps[e] = steps; P.cmd_age[e] = age; P.cmd_idx[e] = idx;
  reward[e] = r; terminated[e] = (uint8_t)term; truncated[e] = (uint8_t)trunc;
  if (term || trunc) {
    if (mode == XV_AUTORESET_SAME_STEP) {
      P.fin_pose[e] = p0; P.fin_pose[N + e] = p1; P.fin_pose[2 * N + e] = ori;
      P.fin_cmd[e] = idx; P.fin_flag[e] = 1;
      mz_reset_env(P, e, t);
    } else if (mode == XV_AUTORESET_NEXT_STEP) {
      P.need_reset[e] = 1;
    }
  }
  if (err) atomicOr(P.err, err);
}

__global__ __launch_bounds__(256) void maze_step_kernel(MazeArgs P, const void* action, int action_mode,
                                                        float* reward, uint8_t* terminated, uint8_t* truncated,
                                                        int mode) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= P.n_env) return;
  const size_t N = (size_t)P.n_env;
  const int t = P.env_task[e];
  const int32_t* in = P.T.ints + (size_t)t * 8;
  const double* db = P.T.dbl + (size_t)t * 8;
  P.fin_flag[e] = 0;
  if (mode == XV_AUTORESET_NEXT_STEP && P.need_reset[e]) {
    mz_reset_env(P, e, t);
    reward[e] = 0.0f; terminated[e] = 0; truncated[e] = 0;
    return;
  }
  double turn_rate, walk_speed;
  uint32_t err = 0;
  mz_decode_action(action, action_mode, e, turn_rate, walk_speed, err);

  // ---- vector_move_with_collision: dynamics.py:158-187 ----
  const int n = in[0], NG = P.NG;
  const double cell_size = db[0], col_dist = P.collision_dist;
  double p0 = P.pos[e], p1 = P.pos[N + e], ori = P.ori[e], coll = 0.0;
  // walls of the 5x5 cells around the starting cell, one bit each (out-of-range cells: no wall, as :180)
  const int ci = (int)(p0 / cell_size), cj = (int)(p1 / cell_size);
  const int8_t* walls = P.T.walls + (size_t)t * NG * NG;
  uint32_t patch = 0;
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      const int wi = ci + a - 2, wj = cj + b - 2;
      if (wi > -1 && wi < n && wj > -1 && wj < n && walls[wi * NG + wj] > 0) patch |= 1u << (a * 5 + b);
    }
  const double t_prec = 0.01, delta_t = 1.0;
  const int iteration = (int)(delta_t / t_prec);
  bool left_patch = false;
  // The reference evaluates cos/sin of the heading and of the half turn in every sub-step.  All full sub-steps
  // share dt = t_prec, so the half-turn pair is evaluated once; the heading pair is re-evaluated only when the
  // heading changed (it does not when turn_rate == 0).  Same arguments, same functions, same values.
  const MzDivisor R_cs = mz_divisor(cell_size);
  const double eff_cd = col_dist / cell_size;
  const double rad = turn_rate != 0.0 ? walk_speed / turn_rate : 0.0;
  double c_dt_full, s_dt_full, c_t, s_t, ori_cached = ori;
  sincos(0.5 * (turn_rate * t_prec), &s_dt_full, &c_dt_full);
  sincos(ori, &s_t, &c_t);
  for (int it = 0; it < iteration + 1; ++it) {
    const double rem = delta_t - it * t_prec;
    const double dt = rem < t_prec ? rem : t_prec;
    if (dt < 1.0e-8) continue;
    // vector_move_no_collision: dynamics.py:98-123
    const double d_theta = turn_rate * dt, arc = walk_speed * dt;
    double c_dt = c_dt_full, s_dt = s_dt_full;
    if (dt != t_prec) sincos(0.5 * d_theta, &s_dt, &c_dt);
    if (ori != ori_cached) { sincos(ori, &s_t, &c_t); ori_cached = ori; }
    const double n_ori = mz_angle_norm(ori + d_theta);
    double dx, dy;
    if (fabs(d_theta) < 1.0e-8) { dx = c_t * arc; dy = s_t * arc; }
    else {
      const double off = 2.0 * s_dt * rad;
      const double c_n = c_t * c_dt - s_t * s_dt, s_n = c_t * s_dt + s_t * c_dt;
      dx = c_n * off; dy = s_n * off;
    }
    ori = n_ori;
    const double e0 = p0 + dx, e1 = p1 + dy;
    const double c0 = mz_div(e0, R_cs), c1 = mz_div(e1, R_cs);
    const int b0 = (int)c0, b1 = (int)c1;
    double f0 = 0.0, f1 = 0.0;
    for (int i = -1; i < 2; ++i)
      for (int j = -1; j < 2; ++j) {
        const int a = b0 + i - ci + 2, b = b1 + j - cj + 2;
        bool wall;
        if (a >= 0 && a < 5 && b >= 0 && b < 5) wall = (patch >> (a * 5 + b)) & 1u;
        else {   // cannot happen for cell_size >= 1 (|move| <= 1); kept exact by falling back to memory
          left_patch = true;
          const int wi = b0 + i, wj = b1 + j;
          wall = wi > -1 && wi < n && wj > -1 && wj < n && walls[wi * NG + wj] > 0;
        }
        if (wall) {
          double g0, g1;
          mz_collision_force(c0 - floor(c0) - (double)(float)(i + 0.5), c1 - floor(c1) - (double)(float)(j + 0.5),
                             cell_size, eff_cd, g0, g1);
          f0 += g0; f1 += g1;
        }
      }
    p0 = f0 + e0; p1 = f1 + e1;
    coll += sqrt(f0 * f0 + f1 * f1);
  }
  (void)left_patch;
  mz_finish_move(P, e, t, p0, p1, ori, coll, err, mode, reward, terminated, truncated);
}

// ------------------------------------------------------------------------------------------------
// The same step, re-arranged for the machine: 9 lanes per env, 7 envs per wave.
//
// A step is 100 sub-steps that look sequential, but only the POSITION is: the heading of sub-step k, its sin / cos and
// therefore its collision-free displacement (dx_k, dy_k) depend on the action alone (dynamics.py:98-123 reads `ori`,
// never `pos`).  So
//   phase 1  every lane walks the cheap heading recurrence (two rounded additions per sub-step, in order) and the 9
//            lanes of an env evaluate sin / cos / displacement for every ninth sub-step in parallel -> LDS;
//   phase 2  the sequential part is what is left: p + d_k, the cell coordinates, and the push-out of the 3x3
//            neighbourhood — one cell per lane (the lane index IS the reference's (i, j) loop position), the nine
//            forces summed in that loop order after an exchange through LDS that is skipped while no lane of the wave
//            touches a wall.
// Same operations on the same operands in the same order as maze_step_kernel above -> identical results (tested), with
// the dependent chain per sub-step cut from ~6,000 to a few hundred cycles and 9x the lanes: 16,384 envs are 2,341
// waves instead of 256.
// ------------------------------------------------------------------------------------------------
constexpr int MZ_SUBMAX = 104;
constexpr int MZ_NSUB = 100;     // sub-steps of a move that are not skipped (see the note in maze_step9_kernel)

// L lanes per env (9: one neighbour cell each; 3: one row of three cells each), 64 / L envs per wave.  Which one is
// faster is a matter of filling the chip: every lane of an env repeats the position arithmetic, so more lanes per env
// mean more waves issuing the same instructions (xv_maze_step picks L from the batch size).
// `list` / `count` (nullable; maze_move_sort_kernel): walk only the envs list[0 .. count[0]); the grid is sized for
// every env, and the workgroups past the walked ones finish the count[1] envs that stand still (list[n_env - 1 - k]).
template <int L>
__global__ __launch_bounds__(64) void maze_step9_kernel(MazeArgs P, const void* action, int action_mode, float* reward,
                                                        uint8_t* terminated, uint8_t* truncated, int mode,
                                                        const int32_t* list, const int32_t* count) {
  constexpr int MZ_EPW = 64 / L, CPL = 9 / L;      // envs per wave, cells per lane
  __shared__ double2 s_d[MZ_EPW + 1][MZ_SUBMAX];
  __shared__ double2 s_g[(MZ_EPW + 1) * 9];
  const int lane = threadIdx.x, q = lane / L, cell = lane - L * q;   // `cell`: this lane's index within its env
  const int n_walk = list ? count[0] : P.n_env;
  if ((int)blockIdx.x * MZ_EPW >= n_walk) {   // uniform: a workgroup past the walked envs
    // finishes 64 of the envs that stand still (listed from the end of `list`): heading recurrence, rules, stores —
    // a few microseconds beside the walking workgroups' ~85
    if (list == nullptr) return;
    const int k = ((int)blockIdx.x - (n_walk + MZ_EPW - 1) / MZ_EPW) * 64 + lane;
    if (k >= count[1]) return;
    const int e = list[P.n_env - 1 - k];
    double turn_rate, walk_speed;
    uint32_t err = 0;
    mz_decode_action(action, action_mode, e, turn_rate, walk_speed, err);
    const double d_theta = turn_rate * 0.01;
    double ori = P.ori[e];
    for (int c = 0; c < MZ_NSUB; ++c) ori = mz_angle_norm(ori + d_theta);
    P.fin_flag[e] = 0;
    mz_finish_move(P, e, P.env_task[e], P.pos[e], P.pos[(size_t)P.n_env + e], ori, 0.0, err, mode, reward, terminated,
                   truncated);
    return;
  }
  const int slot = blockIdx.x * MZ_EPW + q;
  const bool active = q < MZ_EPW && slot < n_walk;
  // idle lanes shadow a real env (the workgroup's first when a list is walked) and store nothing
  const int e = list ? list[active ? slot : blockIdx.x * MZ_EPW] : (active ? slot : P.n_env - 1);
  const size_t N = (size_t)P.n_env;
  const int t = P.env_task[e];
  const int32_t* in = P.T.ints + (size_t)t * 8;
  const double* db = P.T.dbl + (size_t)t * 8;
  const bool lead = active && cell == 0;
  const bool resetting = mode == XV_AUTORESET_NEXT_STEP && P.need_reset[e];
  double turn_rate, walk_speed;
  uint32_t err = 0;
  mz_decode_action(action, action_mode, e, turn_rate, walk_speed, err);
  const int n = in[0], NG = P.NG;
  const double cell_size = db[0], col_dist = P.collision_dist;
  double p0 = P.pos[e], p1 = P.pos[N + e], ori = P.ori[e], coll = 0.0;
  const int ci = (int)(p0 / cell_size), cj = (int)(p1 / cell_size);
  const int8_t* walls = P.T.walls + (size_t)t * NG * NG;
  uint32_t patch = 0;
  {   // all 25 bytes requested at once from clamped addresses (a branch per cell would be 25 dependent round trips)
    int8_t wv[25];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const int wi = ci + a - 2, wj = cj + b - 2;
        const int ri = wi < 0 ? 0 : (wi >= NG ? NG - 1 : wi), rj = wj < 0 ? 0 : (wj >= NG ? NG - 1 : wj);
        wv[a * 5 + b] = walls[ri * NG + rj];
      }
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const int wi = ci + a - 2, wj = cj + b - 2;
        if (wi > -1 && wi < n && wj > -1 && wj < n && wv[a * 5 + b] > 0) patch |= 1u << (a * 5 + b);
      }
  }
  const double t_prec = 0.01;
  const MzDivisor R_cs = mz_divisor(cell_size);
  const double eff_cd = col_dist / cell_size;
  const double rad = turn_rate != 0.0 ? walk_speed / turn_rate : 0.0;
  // ---- phase 1: headings of all sub-steps (sequential, cheap), displacements of this lane's sub-steps ----
  {
    double c_dt_full, s_dt_full, c_t = 1.0, s_t = 0.0, ori_k = ori, ori_cached = 0.0;
    bool have = false;
    sincos(0.5 * (turn_rate * t_prec), &s_dt_full, &c_dt_full);
    // groups of L sub-steps: all lanes walk the recurrence together and each keeps the heading of ITS sub-step of
    // the group (a select); the expensive part then runs once per group with every lane busy, not once per sub-step
    // The reference's loop runs i = 0 .. 100 with t_res = min(delta_t - i * t_prec, t_prec) and skips t_res < 1e-8
    // (dynamics.py:166-169).  With delta_t = 1 and t_prec = 0.01 in fp64: 1 - i * 0.01 >= 0.01 for every i <= 99 (the
    // smallest, i = 99, is 0.010000000000000009) and exactly 0 for i = 100 — i.e. MZ_NSUB = 100 sub-steps of t_res = t_prec,
    // the 101st skipped.  So the step length is a constant of the loop, not a per-iteration computation.
    const double d_theta = turn_rate * t_prec, arc = walk_speed * t_prec;
    for (int base = 0; base < MZ_NSUB; base += L) {
      double my_ori = 0.0;
      bool mine = false;
      for (int c = 0; c < L && base + c < MZ_NSUB; ++c) {
        if (c == cell) { my_ori = ori_k; mine = true; }
        ori_k = mz_angle_norm(ori_k + d_theta);
      }
      if (mine) {
        const double c_dt = c_dt_full, s_dt = s_dt_full;
        if (!have || my_ori != ori_cached) { sincos(my_ori, &s_t, &c_t); ori_cached = my_ori; have = true; }
        double dx, dy;
        if (fabs(d_theta) < 1.0e-8) { dx = c_t * arc; dy = s_t * arc; }
        else {
          const double off = 2.0 * s_dt * rad;
          const double c_n = c_t * c_dt - s_t * s_dt, s_n = c_t * s_dt + s_t * c_dt;
          dx = c_n * off; dy = s_n * off;
        }
        s_d[q][base + cell] = make_double2(dx, dy);
      }
    }
    ori = ori_k;
  }
  __syncthreads();
  // ---- phase 2: positions (sequential), CPL neighbour cells per lane ----
  for (int it = 0; it < MZ_NSUB; ++it) {
    const double2 d = s_d[q][it];
    const double e0 = p0 + d.x, e1 = p1 + d.y;
    const double c0 = mz_div(e0, R_cs), c1 = mz_div(e1, R_cs);
    const int b0 = (int)c0, b1 = (int)c1;
    const double fr0 = c0 - floor(c0), fr1 = c1 - floor(c1);
    double g0[CPL], g1[CPL];
    bool any = false;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int idx = cell * CPL + k;                       // the reference's (i, j) loop position
      const int ni = idx / 3 - 1, nj = idx - 3 * (idx / 3) - 1;
      const int a = b0 + ni - ci + 2, b = b1 + nj - cj + 2;
      bool wall;
      if (a >= 0 && a < 5 && b >= 0 && b < 5) wall = (patch >> (a * 5 + b)) & 1u;
      else {   // cannot happen for cell_size >= 1 (|move| <= 1); kept exact by falling back to memory
        const int wi = b0 + ni, wj = b1 + nj;
        wall = wi > -1 && wi < n && wj > -1 && wj < n && walls[wi * NG + wj] > 0;
      }
      g0[k] = 0.0; g1[k] = 0.0;
      if (wall) {
        // exact-zero filter: the force vanishes unless the agent is within eff of the wall cell's box — the distance
        // to the facing edge is at least the Chebyshev distance max(|v0|, |v1|) - 0.5, and the reference returns zero
        // for eff < distance (dynamics.py:93-94).  The 1e-9 guard keeps the rounding of that distance on the safe
        // side; inside the band the reference's own arithmetic decides.
        const double v0 = fr0 - (double)(float)(ni + 0.5), v1 = fr1 - (double)(float)(nj + 0.5);
        const double cheb = __builtin_fmax(fabs(v0), fabs(v1)) - 0.5;
        if (!(cheb > eff_cd + 1.0e-9)) mz_collision_force_near(v0, v1, cell_size, eff_cd, g0[k], g1[k]);
      }
      any = any || g0[k] != 0.0 || g1[k] != 0.0;
    }
    double f0 = 0.0, f1 = 0.0;
    if (__ballot(any) != 0ull) {   // some env of this wave touches a wall: sum the nine forces in (i, j) order
      double2* gq = s_g + 9 * q;
#pragma unroll
      for (int k = 0; k < CPL; ++k) gq[cell * CPL + k] = make_double2(g0[k], g1[k]);
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 9; ++c) { f0 += gq[c].x; f1 += gq[c].y; }
      __syncthreads();
      // the collision measure only grows where a force exists (+ sqrt(0) = + 0 otherwise): inside the wave-uniform
      // branch, so that the contact-free sub-steps — most of them — do not pay an fp64 square root that hipcc had
      // if-converted into every iteration (round 3: ~25 of ~85 instructions of a contact-free sub-step)
      if (f0 != 0.0 || f1 != 0.0) coll += sqrt(f0 * f0 + f1 * f1);
    }
    p0 = f0 + e0; p1 = f1 + e1;
  }
  if (!lead) return;
  P.fin_flag[e] = 0;
  if (resetting) {
    mz_reset_env(P, e, t);
    reward[e] = 0.0f; terminated[e] = 0; truncated[e] = 0;
    return;
  }
  mz_finish_move(P, e, t, p0, p1, ori, coll, err, mode, reward, terminated, truncated);
}

// ------------------------------------------------------------------------------------------------
// Which envs have to be walked at all.  An env whose action has no walk speed gets a displacement of +-0 in every
// sub-step (dynamics.py:98-123: arc = walk_speed * dt = 0, or offset = 2 s_dt * walk_speed / turn_rate = 0), so e = p + d = p;
// if at p none of the nine cells pushes (each is no wall, or farther than the collision distance by the zero filter of
// maze_step9_kernel, or inside the band with a force of exactly 0 from the same mz_collision_force_near on the same
// operands) the force sum is exactly 0 and p = 0 + e = p: the
// env stands still through all 100 sub-steps, collision 0, and only its heading turns.  10 of the 16 Discrete16 actions
// are turns.  One thread per env sorts the batch: envs to walk are listed from the front of `list`, envs that stand still
// from its end (one atomic per wave and kind; order does not matter, envs are independent).  count[2 w], count[2 w + 1]
// are this step's counters, the other pair is zeroed for the next step.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maze_move_sort_kernel(MazeArgs P, const void* action, int action_mode, int mode,
                                                             int32_t* list, int32_t* count, int w) {
  const int e_raw = blockIdx.x * blockDim.x + threadIdx.x;
  if (e_raw == 0) { count[2 * (w ^ 1)] = 0; count[2 * (w ^ 1) + 1] = 0; }
  const bool valid = e_raw < P.n_env;
  const int e = valid ? e_raw : P.n_env - 1;
  const size_t N = (size_t)P.n_env;
  const int t = P.env_task[e];
  const int32_t* in = P.T.ints + (size_t)t * 8;
  const double* db = P.T.dbl + (size_t)t * 8;
  const bool resetting = mode == XV_AUTORESET_NEXT_STEP && P.need_reset[e];
  double turn_rate, walk_speed;
  uint32_t err = 0;
  mz_decode_action(action, action_mode, e, turn_rate, walk_speed, err);
  const int n = in[0], NG = P.NG;
  const double cell_size = db[0];
  const double p0 = P.pos[e], p1 = P.pos[N + e];
  bool still = valid && !resetting && walk_speed == 0.0 && p0 > 0.0 && p1 > 0.0;   // (p > 0: p + -0 and 0 + p keep p's bits)
  {
    const int8_t* walls = P.T.walls + (size_t)t * NG * NG;
    const MzDivisor R_cs = mz_divisor(cell_size);
    const double eff_cd = P.collision_dist / cell_size;
    const double c0 = mz_div(p0, R_cs), c1 = mz_div(p1, R_cs);
    const int b0 = (int)c0, b1 = (int)c1;
    const double fr0 = c0 - floor(c0), fr1 = c1 - floor(c1);
    int8_t wv[9];
#pragma unroll
    for (int idx = 0; idx < 9; ++idx) {   // all nine bytes requested at once
      const int wi = b0 + idx / 3 - 1, wj = b1 + idx % 3 - 1;
      const int ri = wi < 0 ? 0 : (wi >= NG ? NG - 1 : wi), rj = wj < 0 ? 0 : (wj >= NG ? NG - 1 : wj);
      wv[idx] = walls[ri * NG + rj];
    }
#pragma unroll
    for (int idx = 0; idx < 9; ++idx) {
      const int ni = idx / 3 - 1, nj = idx % 3 - 1;
      const int wi = b0 + ni, wj = b1 + nj;
      const bool wall = wi > -1 && wi < n && wj > -1 && wj < n && wv[idx] > 0;
      const double v0 = fr0 - (double)(float)(ni + 0.5), v1 = fr1 - (double)(float)(nj + 0.5);
      const double cheb = __builtin_fmax(fabs(v0), fabs(v1)) - 0.5;
      if (still && wall && !(cheb > eff_cd + 1.0e-9)) {
        // inside the band the reference's own arithmetic decides (an agent a wall has pushed out rests exactly at the
        // collision distance, where the force is 0 again): the same function on the same operands as the walking kernel
        double g0, g1;
        mz_collision_force_near(v0, v1, cell_size, eff_cd, g0, g1);
        if (g0 != 0.0 || g1 != 0.0) still = false;
      }
    }
  }
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  const unsigned long long mw = __ballot(valid && !still), ms = __ballot(valid && still);
  int base_w = 0, base_s = 0;
  if (lane == 0) {
    if (mw) base_w = atomicAdd(count + 2 * w, __popcll(mw));
    if (ms) base_s = atomicAdd(count + 2 * w + 1, __popcll(ms));
  }
  base_w = __shfl(base_w, 0);
  base_s = __shfl(base_s, 0);
  if (valid && !still) list[base_w + __popcll(mw & below)] = e;
  if (valid && still) list[P.n_env - 1 - (base_s + __popcll(ms & below))] = e;
}

__device__ __forceinline__ uint8_t mz_clip_u8(double v) {   // numpy.clip(v, 0, 255) -> int32 -> uint8
  return (uint8_t)(int)__builtin_fmin(__builtin_fmax(v, 0.0), 255.0);   // v_max/v_min_f64: no NaNs reach here
}

// The 16 packed texels of the 4 x 4 filter window, q[row xx + 1][tap yy + 1], rows ib - 1 .. ib + 2 (mod 256), taps
// (jb - 1) & 255 ... + 3 (rows are padded: no wrap in y).
template <bool PAIRS>
__device__ __forceinline__ void mz_fetch_window(const void* __restrict__ texv, int ib, int jb, uint32_t (&q)[4][4]) {
  const uint32_t* tb = static_cast<const uint32_t*>(texv);
  const int ys = (jb - 1) & 255;
  if (PAIRS) {
  // spans of 8 words = {row 2m tap t, row 2m + 1 tap t : t = 0..3}.  x0 even: rows x0, x0 + 1 in span 0, x0 + 2, x0 + 3 in span
  // 1; x0 odd: x0 is the odd row of span 0, x0 + 1, x0 + 2 fill span 1, x0 + 3 is the even row of span 2.  The third span
  // of an even x0 is span 1 again (same address: no further request).
  const int x0 = (ib - 1) & 255;
  const int odd = x0 & 1;
  const int p0 = x0 >> 1, p1 = ((x0 + 2) & 255) >> 1, p2 = ((x0 + 3) & 255) >> 1;
  const uint4* s0 = reinterpret_cast<const uint4*>(tb + ((size_t)p0 * MZ_TEX_PITCH + ys) * 2);
  const uint4* s1 = reinterpret_cast<const uint4*>(tb + ((size_t)p1 * MZ_TEX_PITCH + ys) * 2);
  const uint4* s2 = reinterpret_cast<const uint4*>(tb + ((size_t)p2 * MZ_TEX_PITCH + ys) * 2);
  const uint4 a0 = s0[0], b0 = s0[1], a1 = s1[0], b1 = s1[1];   // dword-aligned 16-byte loads
  uint4 a2 = s2[0], b2 = s2[1];
  // only words x and z of the third span are used: left alone, hipcc narrows its two 16-byte loads to FOUR dword loads, and
  // every load instruction is one L1 access per lane (the lanes' windows share no lines) — 8.2 accesses per pixel at 0.96 per
  // CU and cycle; with the span fetched whole 6.2 at 0.77, the 64 x 64 ray cast 0.99 -> 0.94 ms (profiles/r06_u_*, r06_v_*).
  // What the fetch costs beyond that is LINES, not instructions: the row-major copy (four loads, four lines per window) and a
  // pair copy held in both phases (four loads, two lines, no selects — but twice the bytes in the 4-MB L2s) both run 1.09-1.10 ms
  // (r06_w_*).  Word y is made "used" through a zero the compiler cannot see (one v_and_or_b32 per load): a 12-byte load each,
  // and no wait at this point (an empty asm taking the words as operands would force one, which the prefetching loop cannot have)
  uint32_t z0;
  asm("s_mov_b32 %0, 0" : "=s"(z0));
  a2.x |= a2.y & z0;
  b2.x |= b2.y & z0;
  // row 0: span 0, member `odd`
  q[0][0] = odd ? a0.y : a0.x; q[0][1] = odd ? a0.w : a0.z; q[0][2] = odd ? b0.y : b0.x; q[0][3] = odd ? b0.w : b0.z;
  // row 1: even -> span 0 member 1; odd -> span 1 member 0
  q[1][0] = odd ? a1.x : a0.y; q[1][1] = odd ? a1.z : a0.w; q[1][2] = odd ? b1.x : b0.y; q[1][3] = odd ? b1.z : b0.w;
  // row 2: span 1, member `odd`
  q[2][0] = odd ? a1.y : a1.x; q[2][1] = odd ? a1.w : a1.z; q[2][2] = odd ? b1.y : b1.x; q[2][3] = odd ? b1.w : b1.z;
  // row 3: even -> span 1 member 1; odd -> span 2 member 0
  q[3][0] = odd ? a2.x : a1.y; q[3][1] = odd ? a2.z : a1.w; q[3][2] = odd ? b2.x : b1.y; q[3][3] = odd ? b2.z : b1.w;
  } else {
#pragma unroll
    for (int xx = 0; xx < 4; ++xx) {
      const int xv = (ib - 1 + xx) & 255;
      const uint4 v = *reinterpret_cast<const uint4*>(tb + (size_t)xv * MZ_TEX_PITCH + ys);   // dword-aligned 16-byte load
      q[xx][0] = v.x; q[xx][1] = v.y; q[xx][2] = v.z; q[xx][3] = v.w;
    }
  }
}

// interpolate, ray_caster_utils.py:123-140 (see oracle mz_interpolate for the typing).
// PACKED: the texture is the engine's RGBX-byte copy with padded rows: the four y-taps of filter row x are the
// 16 contiguous bytes at [x & 255][(jb - 1) & 255 ...], one global_load_dwordx4 instead of 12 dword loads.  Texel
// values are the same integers, so every operation below sees the same operands as the float path.
//
// The weight 1 - 10*dist/d2 divides by the same d2 for all 16 taps (d2 in [1e-8, ~1e3], 10*dist 0 or in
// [~1e-40, 1e4]): mz_div, 3 instructions per tap instead of 14.
// dist >= 0 and d2 > 0 make the reference's upper clamp (wht > 1 -> 1) unreachable; the lower one is a v_max_f64.
// W: the 4 x 4 window is already in registers (`win`, mz_fetch_window); otherwise PACKED says which copy `texv` is.
template <bool PACKED, bool W = false>
__device__ __forceinline__ void mz_interpolate(const void* __restrict__ texv, double i, double j, double d,
                                               double px, double py, double (&out)[3], const uint32_t (*win)[4] = nullptr) {
  double d2 = d * d;
  if (d2 < 1.0e-8) d2 = 1.0e-8;
  const MzDivisor D2 = mz_divisor(d2);
  const int ib = (int)i, jb = (int)j;
  double sum_wht = 0.0;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
  double bb[4];
#pragma unroll
  for (int yy = -1; yy < 3; ++yy) {
    const double b = ((double)(jb + yy) - j) * py;
    bb[yy + 1] = b * b;
  }
  uint32_t qw[4][4];
  if (PACKED && !W) mz_fetch_window<false>(texv, ib, jb, qw);
#pragma unroll
  for (int xx = -1; xx < 3; ++xx) {
    const int x = ib + xx;
    const double a = ((double)x - i) * px;
    const double aa = a * a;
    const int xv = x & 255;   // python's non-negative x % 256
#pragma unroll
    for (int yy = -1; yy < 3; ++yy) {
      const double dist = aa + bb[yy + 1];
      const double wht = __builtin_fmax(1.0 - mz_div(10 * dist, D2), 0.01);
      sum_wht += wht;
      float t0, t1, t2;
      if (PACKED || W) {
        const uint32_t p = W ? win[xx + 1][yy + 1] : qw[xx + 1][yy + 1];
        t0 = (float)(p & 0xFFu); t1 = (float)((p >> 8) & 0xFFu); t2 = (float)((p >> 16) & 0xFFu);
      } else {
        const float* tp = static_cast<const float*>(texv) + ((size_t)xv * 256 + ((jb + yy) & 255)) * 3;
        t0 = tp[0]; t1 = tp[1]; t2 = tp[2];
      }
      s0 = (float)((double)s0 + wht * (double)t0);
      s1 = (float)((double)s1 + wht * (double)t1);
      s2 = (float)((double)s2 + wht * (double)t2);
    }
  }
  const MzDivisor SW = mz_divisor(sum_wht);   // in [0.16, 16]
  out[0] = mz_div((double)s0, SW); out[1] = mz_div((double)s1, SW); out[2] = mz_div((double)s2, SW);
}

// The exact filter, speculated (the default on packed textures, XV_MAZE_FILTER_EXACT).  What the reference's typing costs
// is the float32 accumulator: every tap rounds three running sums to float32 and widens them again (3 x 6 of the tap's 26
// fp64-rate instructions), and the weight is a true division.  mz_interpolate_spec runs the same taps on the same window
// with float64 sums (fma), the weight as fma(-dist, 10 / d2, 1): 13 instructions per tap.  Its colour c' differs from the
// reference's c by a PROVEN bound, and the pixel byte is floor(clip(L (A + B c))), monotone in c — so the byte is certain
// unless an integer lies within the bound of L (A + B c'); only then the lane re-runs mz_interpolate on the window it
// already holds (a second loop per 64 rows; one pixel in ~300 on the bench's synthetic textures, whose flat areas put many
// colours within the bound of an integer — scripts/devtools/probe_spec_redo.py).  Same bytes as the plain exact filter, always.
//   weights: w_ref = max(fl(1 - fl(fl(10 dist) / d2)), .01), w' = max(fl(1 - dist fl(10 / d2)), .01): both within 2^-52 of the
//     real max(1 - 10 dist / d2, .01) where that is not clamped on both sides, so |w' - w_ref| <= 2^-51 <= 4.5e-14 w_ref
//     (w >= .01); sums of positive terms keep relative errors: S' = sum w' t and sw' = sum w' are within 5e-14 (incl. their
//     own 16 roundings of 2^-53) of sum w_ref t and sum w_ref, and the reference's float64 sum_wht within 16 * 2^-53 of it.
//   float32 chain of the reference: s <- fl32(fl64(s + fl64(w t))), |fl32(x) - x| <= 2^-24 |x| (normal range: a nonzero
//     term is >= .01), partial sums <= the final one: |s_ref - sum w_ref t| <= 16 * 2^-24 (1 + 1e-6) sum w_ref t.
//   so |c' - c| <= (9.5368e-7 * 1.000001 + 2e-13) c < 9.6e-7 c'(1 + 1e-6); v = fl(L fl(A + fl(B c))) with L, B >= 0 moves by
//     at most L B |c' - c| + 3 roundings of values < 2^9 (< 1e-12): KAPPA = 9.7e-7 and an absolute 1e-9 cover both.
// Round 6 (XV_MAZE_SPEC32, the default): the pixel loop is bound by VALU ISSUE — every class of instruction, not the fp64 pipe:
// gfx950 issues v_fma_f64 at the rate of v_fma_f32, which is why the fp32 filter was only 17 % faster (counters and the class count
// of the loop: profiles/r06_*raycast*).  So the speculation is cut in INSTRUCTIONS, 13 -> 9 per tap:
//   * weight: cx = fma(-aa, k10, 1) per window row, w' = max(fma(-bb, k10, cx), .01) — one fma per tap instead of add + fma.  Two
//     roundings of values in [-2, 1] (below -2 both this and the reference's weight are clamped): pre-clamp value within 2^-51 of
//     the real one, the reference's within 2^-52, max() is 1-Lipschitz: |w' - w_ref| <= 3 * 2^-52 <= 7e-14 w_ref (w >= .01).
//   * the three colour sums run in float32: t = v_cvt_f32_ubyteN (one instruction instead of bit-field extract + v_cvt_f64_u32),
//     wf = fl32(w'), s <- fl32(fma(wf, t, s)) (two channels in one v_pk_fma_f32).  |wf - w'| <= 2^-24 w'; every term is >= 0, so
//     each of the 16 roundings is at most 2^-24 of the final sum: |s' - sum w' t| <= 17 * 2^-24 (1 + 2e-6) sum w' t = 1.0133e-6.
//     The weight sum runs in the same float32 chain (sw' <- fl32(fma(wf, 1, sw')), the second lane of the v_pk_fma_f32 that sums
//     the third channel): |sw' - sum w'| <= 17 * 2^-24 (1 + 2e-6) sum w' likewise.
//   * 10 / d2 and 1 / sw' come from v_rcp_f64 + one Newton step (mz_rcp_spec: within 1e-12; the weights then move by < 1e-10 of
//     themselves, the colour by 1e-12): 12 instructions fewer than two correctly rounded divisor set-ups and three quotients.
//   so c' = s' / sw' has |c' - c| <= (2 * 1.0133e-6 + 9.5368e-7 * 1.000001 + 3e-10) c < 2.981e-6 c'(1 + 4e-6): KAPPA = 3.0e-6 (0.97e-6
//     for the float64 sums), and about one pixel in 100 instead of one in 300 is re-run in the reference's typing.
//   * byte: v' = fma(L B, c', L A) instead of L (A + B c') (three roundings of values < 2^9 either way: inside the absolute 1e-9),
//     bound e = fma(v', KAPPA, 1e-9) >= L B c' KAPPA + 1e-9 (L, A, B >= 0).  0 <= v' <= 255 (L <= 1, A + B = 1, c' <= 255 (1 + 3e-6);
//     a v' above 255 by that much truncates to 255 as the reference's clip does): no clamp instructions.
// Same bytes as the plain exact filter, always: tests/test_gpu_maze.py (golden frames, the direct filter), tests/soak_maze.py.
#ifndef XV_MAZE_SPEC32
#define XV_MAZE_SPEC32 1
#endif
#if XV_MAZE_SPEC32
#define MZ_SPEC_KAPPA 3.0e-6
#else
#define MZ_SPEC_KAPPA 9.7e-7
#endif
typedef float mz_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mz_interpolate_spec(const uint32_t (&qw)[4][4], double i, double j, double d, double ps,
                                                    double (&out)[3]) {
  double d2 = d * d;
  if (d2 < 1.0e-8) d2 = 1.0e-8;
#if XV_MAZE_SPEC32
  const double k10 = 10.0 * mz_rcp_spec(d2);         // 10 / d2 within 1e-12: the weights move by < 1e-12 absolute, 1e-10 of themselves
#else
  const double k10 = mz_div(10.0, mz_divisor(d2));   // == 10 / d2, correctly rounded
#endif
  const int ib = (int)i, jb = (int)j;
#if XV_MAZE_SPEC32
  double nbb[4];
  mz_f2 s01 = {0.0f, 0.0f}, s2w = {0.0f, 0.0f};      // (s0, s1), (s2, sw)
#pragma unroll
  for (int yy = -1; yy < 3; ++yy) {
    const double b = ((double)(jb + yy) - j) * ps;
    nbb[yy + 1] = -(b * b);
  }
#pragma unroll
  for (int xx = -1; xx < 3; ++xx) {
    const double a = ((double)(ib + xx) - i) * ps;
    const double cx = __builtin_fma(-(a * a), k10, 1.0);
#pragma unroll
    for (int yy = -1; yy < 3; ++yy) {
      const double wht = __builtin_fmax(__builtin_fma(nbb[yy + 1], k10, cx), 0.01);
      const float wf = (float)wht;
      const uint32_t p = qw[xx + 1][yy + 1];
      const mz_f2 t01 = {(float)(p & 0xFFu), (float)((p >> 8) & 0xFFu)};   // v_cvt_f32_ubyte0 / 1
      const mz_f2 t2w = {(float)((p >> 16) & 0xFFu), 1.0f};
      s01 = __builtin_elementwise_fma(mz_f2{wf, wf}, t01, s01);           // v_pk_fma_f32
      s2w = __builtin_elementwise_fma(mz_f2{wf, wf}, t2w, s2w);
    }
  }
  const double isw = mz_rcp_spec((double)s2w.y);      // (1e-12 of the colour: inside the bound's slack)
  out[0] = (double)s01.x * isw; out[1] = (double)s01.y * isw; out[2] = (double)s2w.x * isw;
#else
  double sw = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0, bb[4];
#pragma unroll
  for (int yy = -1; yy < 3; ++yy) {
    const double b = ((double)(jb + yy) - j) * ps;
    bb[yy + 1] = b * b;
  }
#pragma unroll
  for (int xx = -1; xx < 3; ++xx) {
    const double a = ((double)(ib + xx) - i) * ps;
    const double aa = a * a;
#pragma unroll
    for (int yy = -1; yy < 3; ++yy) {
      const double wht = __builtin_fmax(__builtin_fma(-(aa + bb[yy + 1]), k10, 1.0), 0.01);
      sw += wht;
      const uint32_t p = qw[xx + 1][yy + 1];
      s0 = __builtin_fma(wht, (double)(p & 0xFFu), s0);
      s1 = __builtin_fma(wht, (double)((p >> 8) & 0xFFu), s1);
      s2 = __builtin_fma(wht, (double)((p >> 16) & 0xFFu), s2);
    }
  }
  const MzDivisor SW = mz_divisor(sw);
  out[0] = mz_div(s0, SW); out[1] = mz_div(s1, SW); out[2] = mz_div(s2, SW);
#endif
}
// byte of v' = L (A + B c') and whether the reference's byte could differ (an integer within the bound of v', or no number)
__device__ __forceinline__ uint8_t mz_spec_byte(double L, double A, double B, double c, bool& doubt) {
#if XV_MAZE_SPEC32
  const double v = __builtin_fma(L * B, c, L * A);      // (L B and L A are per pixel: the compiler shares them between the channels)
  const double e = __builtin_fma(v, MZ_SPEC_KAPPA, 1.0e-9);
#else
  const double t = B * c, v = L * (A + t);
  const double e = __builtin_fma(L * t, MZ_SPEC_KAPPA, 1.0e-9);
#endif
  const double fr = v - __builtin_floor(v);
  doubt = doubt || !(fr > e && fr < 1.0 - e);
#if XV_MAZE_SPEC32
  return (uint8_t)(int)v;      // 0 <= v < 256 (see above); a doubtful or non-finite v is re-run by the caller anyway
#else
  return mz_clip_u8(v);
#endif
}

// Opt-in fp32 variant of the filter above (xv_maze_set_precision(XV_MAZE_FILTER_F32)): same 4x4 taps, same weights
// 1 - 10 dist / d2 clamped at 0.01, same normalisation, evaluated in float32 with one reciprocal per pixel instead of
// in the reference's float64 typing: 10 fp32 instructions per tap instead of ~23 fp64-rate ones.  The colour differs
// from the exact filter by ~1e-4 of a level, i.e. a frame value moves by one level where the exact colour sits that
// close to an integer: within SURVEY.md M5's budget (+-1 LSB on <= 0.5 % of the values; measured by the tests).
template <bool PACKED>
__device__ __forceinline__ void mz_interpolate_f32(const void* __restrict__ texv, double i, double j, double d,
                                                   double ps, double (&out)[3]) {
  const int ib = (int)i, jb = (int)j;
  const float fi = (float)((double)ib - i), fj = (float)((double)jb - j), p = (float)ps;
  const float d2 = fmaxf((float)(d * d), 1.0e-8f);
  const float k10 = 10.0f / d2;
  float sw = 0.0f, s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, bb[4];
#pragma unroll
  for (int yy = -1; yy < 3; ++yy) {
    const float b = (fj + (float)yy) * p;
    bb[yy + 1] = b * b;
  }
  uint32_t qw[4][4];
  if (PACKED) mz_fetch_window<true>(texv, ib, jb, qw);   // the pair-interleaved copy
#pragma unroll
  for (int xx = -1; xx < 3; ++xx) {
    const float a = (fi + (float)xx) * p, aa = a * a;
    const int xv = (ib + xx) & 255;
#pragma unroll
    for (int yy = -1; yy < 3; ++yy) {
      const float wht = fmaxf(fmaf(-(aa + bb[yy + 1]), k10, 1.0f), 0.01f);
      sw += wht;
      float t0, t1, t2;
      if (PACKED) {
        const uint32_t px = qw[xx + 1][yy + 1];
        t0 = (float)(px & 0xFFu); t1 = (float)((px >> 8) & 0xFFu); t2 = (float)((px >> 16) & 0xFFu);
      } else {
        const float* tp = static_cast<const float*>(texv) + ((size_t)xv * 256 + ((jb + yy) & 255)) * 3;
        t0 = tp[0]; t1 = tp[1]; t2 = tp[2];
      }
      s0 = fmaf(wht, t0, s0); s1 = fmaf(wht, t1, s1); s2 = fmaf(wht, t2, s2);
    }
  }
  const float inv = 1.0f / sw;
  out[0] = (double)(s0 * inv); out[1] = (double)(s1 * inv); out[2] = (double)(s2 * inv);
}

// texel (x, y) of library entry k -> packed RGBX word, rows padded with 3 wrapped texels: at [k][x][y], or — PAIRS — at
// [k][x >> 1][y][x & 1]
template <bool PAIRS>
__global__ __launch_bounds__(256) void maze_pack_tex_kernel(const float* tex, uint32_t* pk, int n_tex) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n_tex * 256 * MZ_TEX_PITCH;
  if (idx >= total) return;
  int y;
  size_t kx;
  if (PAIRS) {
    const size_t per = (size_t)256 * MZ_TEX_PITCH, k = idx / per, w = idx % per;
    const int b = (int)(w & 1), m = (int)((w >> 1) / MZ_TEX_PITCH);
    y = (int)((w >> 1) % MZ_TEX_PITCH);
    kx = k * 256 + (size_t)(2 * m + b);
  } else {
    y = (int)(idx % MZ_TEX_PITCH);
    kx = idx / MZ_TEX_PITCH;
  }
  const float* tp = tex + (kx * 256 + (size_t)(y & 255)) * 3;
  pk[idx] = (uint32_t)tp[0] | ((uint32_t)tp[1] << 8) | ((uint32_t)tp[2] << 16);
}

// are all texels integers in [0, 255]?
__global__ __launch_bounds__(256) void maze_tex_integral_kernel(const float* tex, size_t n, int* not_integral) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float v = tex[idx];
  if (!(v >= 0.0f && v <= 255.0f && v == floorf(v))) atomicOr(not_integral, 1);
}

// One workgroup per frame, one lane per column.  FINAL: render the stored pre-reset pose of flagged envs.
// The frame is built in LDS in chunks of P.HC rows ([column][HC*3 + 4] bytes: the pad makes the per-lane byte
// writes bank-conflict free) and each chunk leaves with 16-byte stores; a 64x64 frame is one chunk, the registered
// 256x256 frame four chunks of 64 rows (49 KiB of LDS, three workgroups per CU).
// NB (xv_maze_set_typing(XV_MAZE_TYPING_NUMBA)): DDA_2D and the wall-column geometry in float64, the types numba infers
// for the reference's source (float32 table entry op float64 -> float64); default float32 = the same source run as plain
// Python under NumPy 2, which the golden frames were made with (oracle/mz_wall_stage.inc holds both, REAL = float / double).
// FILT: 0 / 3 = the exact filter, speculated when PACKED (mz_interpolate_spec), reading the pair-interleaved (0) or the
// row-major (3) texture copy; 1 = the opt-in fp32 filter; 2 = the exact filter evaluated directly for every pixel
// (XV_MAZE_FILTER_EXACT_DIRECT: what the speculated one is tested against); 5 / 6 = 3 / 1 with the pixel loop on the ROWS of
// one column per wave (xv_maze_set_raycast_mapping; maze_launch_render picks).  The launcher picks 0 for frames up to 128 x 128
// and 3 beyond (16,384 envs, scripts/runs_r04/gpu_t.sh: 64 x 64 1.09 ms pairs / 1.16 ms rows, 256 x 256 14.6 / 14.1 ms;
// direct 1.29 / 17.4 ms).
// What a column of the frame hands to its pixels (the wall the column's ray hit and the ray's direction): in registers when a
// lane paints its own column, in LDS when the lanes of a wave paint the ROWS of one column (FILT 5 / 6)
struct MzColumn {
  int v_s, v_e, text_id, ti;                                        // wall rows [v_s, v_e), wall texture, its texel row (0 .. 255)
  double L, a_far, a_near, ratio, co, so, rcos_b, rcos_y;             // 80 bytes
};
__device__ __forceinline__ double mz_shfl_f64(double v, int src) {
  return __hiloint2double(__shfl(__double2hiint(v), src), __shfl(__double2loint(v), src));
}
__device__ __forceinline__ MzColumn mz_column_of_lane(const MzColumn& m, int src) {   // all lanes of the wave call this
  MzColumn r;
  r.v_s = __shfl(m.v_s, src); r.v_e = __shfl(m.v_e, src); r.text_id = __shfl(m.text_id, src); r.ti = __shfl(m.ti, src);
  r.L = mz_shfl_f64(m.L, src); r.a_far = mz_shfl_f64(m.a_far, src);
  r.a_near = mz_shfl_f64(m.a_near, src); r.ratio = mz_shfl_f64(m.ratio, src); r.co = mz_shfl_f64(m.co, src);
  r.so = mz_shfl_f64(m.so, src); r.rcos_b = mz_shfl_f64(m.rcos_b, src); r.rcos_y = mz_shfl_f64(m.rcos_y, src);
  return r;
}
// Pixels whose byte the speculated filter could not settle (about one in 100) are re-run in the reference's typing.  They sit
// unevenly on the lanes: a wave that lets every lane re-run its own runs max-over-lanes rounds of the ~600-instruction exact
// filter per block of 64 x 64 pixels (3 - 4 rounds for ~40 items: a tenth of the block's time).  So the items of a block are
// SPREAD over the wave first: compacted into a per-wave list in LDS (each round the lanes that still hold an item append their
// lowest one, ranked by a ballot), then item idx goes to lane idx % 64 — one round for up to 64 items.  What does not fit the list
// stays on its lane (the loop behind the spread).
#define MZ_REDO_LIST 64      // (items of a block beyond it stay on their lanes; 4 waves x 64 x 2 B of LDS)
#ifndef XV_MAZE_REDO_SPREAD
#define XV_MAZE_REDO_SPREAD 1
#endif
// the frame flush as nontemporal stores: a launch writes 201 MB (64 x 64) / 3.2 GB (256 x 256) of frames through the 4-MB L2s that
// hold the texels; as streaming lines they displace fewer of them (L2 misses 5.7e6 -> 4.0e6 per 64 x 64 launch, 0.943 -> 0.927 ms,
// 10.58 -> 10.36 ms at 256 x 256: profiles/r06_y_*)
#ifndef XV_MAZE_NT_FLUSH
#define XV_MAZE_NT_FLUSH 1
#endif
// Columns mapping, speculated filter (FILT 0 / 3 on packed textures): the window of the NEXT pixel is requested before the current
// one is filtered.  That takes 206 registers, i.e. two waves per SIMD instead of three — and still wins: 0.925 -> 0.885 ms at
// 64 x 64 (two pixels ahead, 241 registers: 0.90; three waves per SIMD with the 29 spills that forces: 1.04).  The rows mapping
// does not take it: 10.40 -> 10.52 ms at 256 x 256, where issue is 86 % busy already (profiles/r06_zz*)
#ifndef XV_MAZE_ROWS_HALF
#define XV_MAZE_ROWS_HALF 1
#endif
#ifndef XV_MAZE_PREFETCH
#define XV_MAZE_PREFETCH 1
#endif
#define MZ_RC_WAVES_OF(FILT, PACKED) ((XV_MAZE_PREFETCH && (PACKED) && ((FILT) == 0 || (FILT) == 3)) ? 2 : XV_MAZE_RC_WAVES)
#ifndef XV_MAZE_RC_WAVES
#define XV_MAZE_RC_WAVES 3   // waves per SIMD the register allocation aims at (2: 1.12 ms at 64 x 64, same at 256 x 256)
#endif
template <bool FINAL, bool PACKED, int FILT, bool NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MZ_RC_WAVES_OF(FILT, PACKED), MZ_RC_WAVES_OF(FILT, PACKED)))) void maze_raycast_kernel(MazeArgs P, uint8_t* frames, float* command_rgb) {
  using RT = typename std::conditional<NB, double, float>::type;
  constexpr bool F32 = FILT == 1 || FILT == 6, SPEC = (FILT == 0 || FILT == 3 || FILT == 5) && PACKED;
  constexpr bool PP = F32 || (SPEC && FILT == 0);   // which packed copy the pixels read
  constexpr bool ROWS = (FILT == 5 || FILT == 6) && PACKED;        // lanes = rows of one column in the pixel loop
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  __shared__ uint16_t redo_list[4][MZ_REDO_LIST];      // per wave: (lane | index << 6) of the pixels to re-run, see above
  const int e = blockIdx.x;
  if (FINAL && !P.fin_flag[e]) return;   // block-uniform
  const int W = P.W, H = P.H, NG = P.NG;
  const size_t N = (size_t)P.n_env;
  const int t = P.env_task[e];
  const int32_t* in = P.T.ints + (size_t)t * 8;
  const double* db = P.T.dbl + (size_t)t * 8;
  const int n = in[0];
  const double cell_size = db[0], ceil_height = db[1], vision_height = db[2];
  const double visibility = P.visibility, l_focal = 0.20, text_size = 1.0;
  const int8_t* walls = P.T.walls + (size_t)t * NG * NG;
  const int8_t* transp = P.T.landmarks + (size_t)t * NG * NG;
  const int32_t* texts = P.T.texts + (size_t)t * NG * NG;
  const void* ground = PACKED ? (const void*)((PP ? P.pp_grounds : P.pk_grounds) + (size_t)in[3] * 256 * MZ_TEX_PITCH)
                              : (const void*)(P.T.tex_grounds + (size_t)in[3] * 256 * 256 * 3);
  const void* ceil_t = PACKED ? (const void*)((PP ? P.pp_ceilings : P.pk_ceilings) + (size_t)in[4] * 256 * MZ_TEX_PITCH)
                              : (const void*)(P.T.tex_ceilings + (size_t)in[4] * 256 * 256 * 3);
  const double pe0 = FINAL ? P.fin_pose[e] : P.pos[e];
  const double pe1 = FINAL ? P.fin_pose[N + e] : P.pos[N + e];
  const double ori = FINAL ? P.fin_pose[2 * N + e] : P.ori[e];
  const int cmd_idx_e = FINAL ? P.fin_cmd[e] : P.cmd_idx[e];
  const float pos0 = (float)pe0, pos1 = (float)pe1;   // maze_continuous_3d.py:97: pose cast to float32

  const double half_h = db[7] * l_focal;               // numpy.tan(vision_angle_h / 2) * l_focal (host fp64)
  const double half_v = half_h * H / W;
  const double pixel_size = 2.0 * half_h / W;
  const double s_ori = sin(ori), c_ori = cos(ori);
  const double pixel_factor = pixel_size / l_focal;
  const double percell = cell_size / text_size;
  const double tps = text_size / 256;
  const size_t fsz = (size_t)W * H * 3;
  uint8_t* dst = frames + (size_t)e * fsz;
  const int HC = P.HC, cstride = HC * 3 + 4;
  // per-row table {distance to the floor/ceiling point, light} (:182-186, :216-219): rows only, shared by all columns
  // rows mapping: the frame chunk holds NSUB columns at a time (the per-column table all of the pass's), see maze_launch_render
  const int NSUB = ROWS ? P.NSUB : (int)blockDim.x;
  double2* rowtab = reinterpret_cast<double2*>(lds + (((size_t)NSUB * cstride + 15) & ~(size_t)15));
  for (int d_v = threadIdx.x; d_v < H; d_v += blockDim.x) {
    const bool is_floor = d_v > H / 2;
    const double v_screen = is_floor ? (d_v + 0.5) * pixel_size - half_v : half_v - (d_v + 0.5) * pixel_size;
    double distance = (is_floor ? vision_height : ceil_height - vision_height) / v_screen * l_focal;
    double light = v_screen / l_focal;
    light = light > 1.0 ? 1.0 : light;
    if (d_v == H / 2) distance = __builtin_huge_val();   // the middle row belongs to neither loop
    rowtab[d_v] = make_double2(distance, light);
  }
  __syncthreads();
  const MzDivisor R_vis = mz_divisor(visibility), R_cs = mz_divisor(cell_size), R_lf = mz_divisor(l_focal);
  const int idxc = cmd_idx_e < P.n_cmd ? cmd_idx_e : P.n_cmd - 1;
  const int cmd = P.T.commands[(size_t)t * P.n_cmd + idxc];

  // quirk (i), SURVEY.md M5: the wall stage filters with the eff_distance left over by the LAST floor/ceiling
  // pixel the reference painted: last ceiling row within visibility (else last floor row), column W-1.
  // It depends on the pose-free screen geometry only, so every lane derives it (no exchange).
  const double cmh = ceil_height - vision_height;
  const float cs_f = (float)cell_size;
  for (int g0 = 0; g0 < W; g0 += blockDim.x) {
    // lanes past the last column repeat it (no divergence); the flush below copies real columns only
    const int d_h = min(g0 + (int)threadIdx.x, W - 1);
    // ---- per-column tables :170-177 (the reference accumulates tan_hp column by column) ----
    double tan_hp = (-0.5 - W / 2.0) * pixel_factor, tan_acc = tan_hp;
    for (int q = 0; q < W; ++q) { tan_acc += pixel_factor; if (q == d_h) tan_hp = tan_acc; }
    const double cos_hp = sqrt(1.0 / (1.0 + tan_hp * tan_hp));
    const double sin_hp = tan_hp * cos_hp;
    const float so32 = (float)(sin_hp * c_ori + cos_hp * s_ori);
    const float co32 = (float)(cos_hp * c_ori - sin_hp * s_ori);
    const RT so = (RT)so32, co = (RT)co32;
    const float cos_hp_f = (float)cos_hp;
    const float cos_last = (float)sqrt(1.0 / (1.0 + tan_acc * tan_acc));
    double eff_stale = 0.0;
    {
      bool found = false;
      for (int d_v = H / 2 - 1; d_v >= 0 && !found; --d_v) {
        const double distance = rowtab[d_v].x;
        if (!(distance > visibility)) { eff_stale = distance / (double)cos_last; found = true; }
      }
      for (int d_v = H / 2 + 1; d_v <= H - 1 && !found; ++d_v) {
        const double distance = rowtab[d_v].x;
        if (!(distance > visibility)) { eff_stale = distance / (double)cos_last; found = true; }
      }
    }
    uint8_t* col = lds + (size_t)threadIdx.x * cstride;
    // ---- DDA_2D :47-115, in RT ----
    const int i0 = (int)(pos0 / cs_f), j0 = (int)(pos1 / cs_f);
    const RT cs_r = (RT)cell_size, eps_r = (RT)1.0e-8, vis_r = (RT)visibility, lf_r = (RT)l_focal;
    const RT c_sign = co < 0 ? (RT)-1 : (RT)1, s_sign = so < 0 ? (RT)-1 : (RT)1;
    const RT ddx = xv_abs(co) < eps_r ? xv_abs(cs_r / eps_r) : xv_abs(cs_r / co);
    const RT ddy = xv_abs(so) < eps_r ? xv_abs(cs_r / eps_r) : xv_abs(cs_r / so);
    const RT d_x = co > 0 ? ((RT)((i0 + 1) * cell_size) - (RT)pos0) : ((RT)(i0 * cell_size) - (RT)pos0);
    const RT d_y = so > 0 ? ((RT)((j0 + 1) * cell_size) - (RT)pos1) : ((RT)(j0 * cell_size) - (RT)pos1);
    RT sdx = xv_abs(co) < eps_r ? c_sign * (d_x / eps_r) : d_x / co;
    RT sdy = xv_abs(so) < eps_r ? s_sign * (d_y / eps_r) : d_y / so;
    const int di = co > 0 ? 1 : -1, dj = so > 0 ? 1 : -1;
    int hi = i0, hj = j0, hit_side = 0, n_tr = 0;
    RT hit_dist = 0;
    // landmark cells crossed by this ray: at most one per DDA step within visibility; 16 slots cover any maze
    RT tr_dist[16];
    int tr_id[16];
    while (hit_dist < vis_r) {
      int crossed;
      if (sdx < sdy) {
        hi += di; sdy -= sdx; hit_dist += sdx;
        crossed = (hi >= 0 && hi < NG && hj >= 0 && hj < NG) ? transp[hi * NG + hj] : -1;
        if (crossed > -1 && n_tr < 16) {
#pragma unroll
          for (int q = 0; q < 16; ++q) if (q == n_tr) { tr_dist[q] = hit_dist; tr_id[q] = crossed; }
          ++n_tr;
        }
        if (hi < 0 || hi >= n) { if (hj < 0 || hj >= n) { hit_dist = (RT)1.0e+6; break; } }
        else if (hj >= 0 && hj < NG && walls[hi * NG + hj] > 0) { hit_side = 0; break; }
        sdx = ddx;
      } else {
        hj += dj; sdx -= sdy; hit_dist += sdy;
        crossed = (hi >= 0 && hi < NG && hj >= 0 && hj < NG) ? transp[hi * NG + hj] : -1;
        if (crossed > -1 && n_tr < 16) {
#pragma unroll
          for (int q = 0; q < 16; ++q) if (q == n_tr) { tr_dist[q] = hit_dist; tr_id[q] = crossed; }
          ++n_tr;
        }
        if (hi < 0 || hi >= n) { if (hj < 0 || hj >= n) { hit_dist = (RT)1.0e+6; break; } }
        else if (hj >= 0 && hj < NG && walls[hi * NG + hj] > 0) { hit_side = 1; break; }
        sdy = ddy;
      }
    }
    // ---- wall column parameters :258-298 ----
    RT alpha_w = (RT)2 * hit_dist / vis_r - (RT)1;
    alpha_w = alpha_w < 0 ? (RT)0 : alpha_w;
    alpha_w = alpha_w > 1 ? (RT)1 : alpha_w;
    const bool in_grid = hi >= 0 && hi < NG && hj >= 0 && hj < NG;
    const int text_id = in_grid ? texts[hi * NG + hj] : 0;
    const RT hit_pt_x = hit_dist * co + (RT)pos0, hit_pt_y = hit_dist * so + (RT)pos1;
    RT local_h;
    float light_w;
    if (hit_side == 0) { local_h = hit_pt_y / cs_r; local_h -= xv_floor(local_h); light_w = fabsf(co32); }
    else { local_h = hit_pt_x / cs_r; local_h -= xv_floor(local_h); light_w = fabsf(so32); }
    RT ratio = hit_dist * (RT)cos_hp_f / lf_r;
    if (xv_abs(ratio) < eps_r) ratio = ratio > 0 ? eps_r : -eps_r;
    const RT top_v = (RT)cmh / ratio, bot_v = (RT)vision_height / ratio;
    int v_s = (int)((half_v - (double)top_v) / pixel_size), v_e = (int)((half_v + (double)bot_v) / pixel_size);
    v_s = v_s < 0 ? 0 : v_s;
    v_e = v_e > H ? H : v_e;
    const void* wt = PACKED ? (const void*)((PP ? P.pp_walls : P.pk_walls) + (size_t)text_id * 256 * MZ_TEX_PITCH)
                            : (const void*)(P.T.tex_walls + (size_t)text_id * 256 * 256 * 3);
    const double eff_ps_w = eff_stale * pixel_size / l_focal;
    float wall_ti;
    {
      RT d_i = local_h * (RT)percell;
      d_i -= xv_floor(d_i);
      wall_ti = (float)(int)((RT)256 * d_i);
    }
    const double a_far_w = (double)(RT)(alpha_w * (RT)1), a_near_w = (double)((RT)1 - alpha_w);
    const MzDivisor R_cos = mz_divisor((double)cos_hp_f);

    // ---- one pass over the column.  The reference paints floor (:180-211), ceiling (:214-244) and then the wall
    // segment [v_s, v_e) over them (:258-298); a floor/ceiling pixel under the wall is a dead store, so each pixel
    // is filtered once with the parameters of the stage that owns it.  Every lane runs exactly H iterations and
    // there is a single copy of the 16-tap filter.  Unpainted pixels keep FAR_RGB = 1 (:165-166).
    // what pixel d_v of the column shows: the texture, the filter position (f_i, f_j), the footprint f_d and the shading
    // v = L (A + B c) of its colour c
    const MzColumn me = {v_s, v_e, text_id, (int)wall_ti, (double)light_w, a_far_w, a_near_w, (double)ratio,
                         (double)co, (double)so, R_cos.b, R_cos.y};
    auto pixel = [&](const MzColumn& C, int d_v, const void*& tx, double& f_i, double& f_j, double& f_d, double& L, double& A,
                     double& B) -> bool {
      bool paint = false;
      tx = PACKED ? (const void*)((PP ? P.pp_walls : P.pk_walls) + (size_t)C.text_id * 256 * MZ_TEX_PITCH)
                  : (const void*)(P.T.tex_walls + (size_t)C.text_id * 256 * 256 * 3);
      f_i = 0.0; f_j = 0.0; f_d = eff_ps_w; L = C.L; A = C.a_far; B = C.a_near;
      if (d_v >= C.v_s && d_v < C.v_e) {
        const double local_v = (half_v - (d_v + 0.5) * pixel_size) * C.ratio + vision_height;
        double d_j = local_v / text_size;
        d_j -= floor(d_j);
        f_i = (double)C.ti;
        f_j = (double)(int)(256 * d_j);
        paint = true;
      } else {
        const bool is_floor = d_v > H / 2;
        const double2 dl = rowtab[d_v];
        const double distance = dl.x, light = dl.y;
        if (!(distance > visibility)) {
          const MzDivisor R_c = {C.rcos_b, C.rcos_y};
          const double eff = mz_div(distance, R_c);
          double alpha = mz_div(2.0 * eff, R_vis) - 1.0;
          alpha = __builtin_fmin(__builtin_fmax(alpha, 0.0), 1.0);
          if (is_floor) alpha *= light;   // :189, the floor only
          const double hit_x = eff * C.co + (double)pos0, hit_y = eff * C.so + (double)pos1;
          const double fi = mz_div(hit_x, R_cs), fj = mz_div(hit_y, R_cs);
          double d_i = fi - floor(fi), d_j = fj - floor(fj);
          const int i = (int)fi, j = (int)fj;
          if (i < n && i >= 0 && j < n && j >= 0) {
            d_i *= percell; d_j *= percell;
            d_i -= floor(d_i); d_j -= floor(d_j);
            f_i = d_i * 256; f_j = d_j * 256;
            f_d = mz_div(eff * pixel_size, R_lf);
            tx = is_floor ? ground : ceil_t;
            L = light; A = alpha * 1.0; B = 1.0 - alpha;
            paint = true;
          }
        }
      }
      return paint;
    };
    // redo: the wave's items out of the lanes' masks, `exact(src_lane, j, have)` for each (all lanes call it; `have`: this lane
    // holds an item)
    auto redo_spread = [&](unsigned long long& redo, auto&& exact) {
      const int wvi = threadIdx.x >> 6, ln = threadIdx.x & 63;
      uint16_t* list = redo_list[wvi];
      int total = 0;
      for (;;) {
        const unsigned long long holders = __ballot(redo != 0ull);
        if (holders == 0ull) break;
        const int nh = __popcll(holders);
        if (total + nh > MZ_REDO_LIST) break;
        if (redo != 0ull) {
          const int j = __builtin_ctzll(redo);
          redo &= redo - 1ull;
          list[total + __popcll(holders & ((1ull << ln) - 1ull))] = (uint16_t)(ln | (j << 6));
        }
        total += nh;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own LDS writes, before its lanes read each other's
      for (int base = 0; base < total; base += 64) {
        const int idx = base + ln;
        const bool have = idx < total;
        const uint32_t it = list[have ? idx : 0];
        exact((int)(it & 63u), (int)(it >> 6), have);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the list is free again
    };
    MzColumn* colp = reinterpret_cast<MzColumn*>(rowtab + H);   // ROWS: the columns of this batch, behind the row table
    if (ROWS) colp[threadIdx.x] = me;
    for (int c0 = 0; c0 < H; c0 += HC) {
      const int c1 = min(c0 + HC, H);
      // SPEC: blocks of 64 rows; the pixels whose byte the speculated filter could not settle are noted in `redo` and
      // filtered in the reference's typing by a second loop (rare: the two filters never share a register allocation)
      for (int h0 = 0; h0 < min((int)blockDim.x, W - g0); h0 += NSUB) {      // (one sub-pass unless the rows mapping splits the columns)
      const int nsub = min(NSUB, min((int)blockDim.x, W - g0) - h0);
      if (ROWS) {
        // the lanes of a wave paint 64 ROWS of one column at a time (wave w takes columns h0 + w, h0 + w + nw, ...): wall pixels of a
        // column read the same four texture rows, the rows of a ray's floor / ceiling pixels neighbouring texels
        __syncthreads();
        const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6, ln = threadIdx.x & 63;
        for (int r0 = c0; r0 < c1; r0 += 64) {
          const bool act = r0 + ln < c1;
          const int d_v = act ? r0 + ln : c1 - 1;
          unsigned long long redo = 0ull;
          int k = 0;
          for (int cl = wv; cl < nsub; cl += nw, ++k) {
            const MzColumn C = colp[h0 + cl];
            const void* tx;
            double f_i, f_j, f_d, L, A, B;
            const bool paint = pixel(C, d_v, tx, f_i, f_j, f_d, L, A, B);
            uint8_t* px = lds + (size_t)cl * cstride + (d_v - c0) * 3;
            uint8_t b0 = 1, b1 = 1, b2 = 1;
            if (paint) {
              double c[3];
              if (SPEC) {
                uint32_t qw[4][4];
                mz_fetch_window<PP>(tx, (int)f_i, (int)f_j, qw);
                mz_interpolate_spec(qw, f_i, f_j, f_d, tps, c);
                bool doubt = false;
                b0 = mz_spec_byte(L, A, B, c[0], doubt);
                b1 = mz_spec_byte(L, A, B, c[1], doubt);
                b2 = mz_spec_byte(L, A, B, c[2], doubt);
                redo |= (unsigned long long)(doubt && act) << k;
              } else {
                mz_interpolate_f32<PACKED>(tx, f_i, f_j, f_d, tps, c);
                b0 = mz_clip_u8(L * (A + B * c[0]));
                b1 = mz_clip_u8(L * (A + B * c[1]));
                b2 = mz_clip_u8(L * (A + B * c[2]));
              }
            }
            if (act) { px[0] = b0; px[1] = b1; px[2] = b2; }
          }
          if (SPEC && XV_MAZE_REDO_SPREAD) {
            redo_spread(redo, [&](int src_lane, int kk, bool have) {
              if (!have) return;
              const int dv2 = r0 + src_lane, cc = wv + kk * nw;      // (only lanes with a row of this block recorded items)
              const MzColumn C = colp[h0 + cc];
              const void* tx;
              double f_i, f_j, f_d, L, A, B, c[3];
              (void)pixel(C, dv2, tx, f_i, f_j, f_d, L, A, B);
              uint32_t qw[4][4];
              mz_fetch_window<PP>(tx, (int)f_i, (int)f_j, qw);
              mz_interpolate<true, true>(nullptr, f_i, f_j, f_d, tps, tps, c, qw);
              uint8_t* px = lds + (size_t)cc * cstride + (dv2 - c0) * 3;
              px[0] = mz_clip_u8(L * (A + B * c[0]));
              px[1] = mz_clip_u8(L * (A + B * c[1]));
              px[2] = mz_clip_u8(L * (A + B * c[2]));
            });
          }
          if (SPEC) {
            while (redo) {   // per lane: the pixels whose byte the speculation could not settle, in the reference's typing
              const int kk = __builtin_ctzll(redo);
              redo &= redo - 1ull;
              const int cc = wv + kk * nw;
              const MzColumn C = colp[h0 + cc];
              const void* tx;
              double f_i, f_j, f_d, L, A, B, c[3];
              (void)pixel(C, d_v, tx, f_i, f_j, f_d, L, A, B);
              uint32_t qw[4][4];
              mz_fetch_window<PP>(tx, (int)f_i, (int)f_j, qw);
              mz_interpolate<true, true>(nullptr, f_i, f_j, f_d, tps, tps, c, qw);
              uint8_t* px = lds + (size_t)cc * cstride + (d_v - c0) * 3;
              px[0] = mz_clip_u8(L * (A + B * c[0]));
              px[1] = mz_clip_u8(L * (A + B * c[1]));
              px[2] = mz_clip_u8(L * (A + B * c[2]));
            }
          }
        }
        __syncthreads();
      } else
      for (int r0 = c0; r0 < c1; r0 += SPEC ? 64 : HC) {
        const int r1 = SPEC ? min(r0 + 64, c1) : c1;
        unsigned long long redo = 0ull;
#if XV_MAZE_PREFETCH
        if (SPEC) {
          // software pipeline: the window of pixel k + 1 is requested before pixel k is filtered, so a wave's texel round trip
          // (an L2 miss for at least one of its 64 lanes in nearly every row) runs under its own arithmetic.  Unpainted pixels
          // fetch and filter the wall texture at (0, 0) — valid addresses, bytes discarded — so no branch separates the stages.
          struct PxGeo { double f_i, f_j, f_d, L, A, B; bool paint; };
          auto stage1 = [&](int d_v, PxGeo& g, uint32_t (&qw)[4][4]) {
            const void* tx;
            g.paint = pixel(me, d_v, tx, g.f_i, g.f_j, g.f_d, g.L, g.A, g.B);
            mz_fetch_window<PP>(tx, (int)g.f_i, (int)g.f_j, qw);
          };
          auto stage2 = [&](int d_v, const PxGeo& g, const uint32_t (&qw)[4][4]) {
            double c[3];
            mz_interpolate_spec(qw, g.f_i, g.f_j, g.f_d, tps, c);
            bool doubt = false;
            const uint8_t b0 = mz_spec_byte(g.L, g.A, g.B, c[0], doubt);
            const uint8_t b1 = mz_spec_byte(g.L, g.A, g.B, c[1], doubt);
            const uint8_t b2 = mz_spec_byte(g.L, g.A, g.B, c[2], doubt);
            uint8_t* px = col + (d_v - c0) * 3;
            px[0] = g.paint ? b0 : (uint8_t)1; px[1] = g.paint ? b1 : (uint8_t)1; px[2] = g.paint ? b2 : (uint8_t)1;
            redo |= (unsigned long long)(doubt && g.paint) << (d_v - r0);
          };
          PxGeo gA, gB;
          uint32_t qA[4][4], qB[4][4];
          stage1(r0, gA, qA);
          for (int d_v = r0; d_v < r1; d_v += 2) {
            stage1(min(d_v + 1, r1 - 1), gB, qB);
            stage2(d_v, gA, qA);
            stage1(min(d_v + 2, r1 - 1), gA, qA);
            if (d_v + 1 < r1) stage2(d_v + 1, gB, qB);
          }
        } else
#endif
        for (int d_v = r0; d_v < r1; ++d_v) {
          const void* tx;
          double f_i, f_j, f_d, L, A, B;
          const bool paint = pixel(me, d_v, tx, f_i, f_j, f_d, L, A, B);
          uint8_t* px = col + (d_v - c0) * 3;
          if (paint) {
            double c[3];
            if (SPEC) {
              uint32_t qw[4][4];
              mz_fetch_window<PP>(tx, (int)f_i, (int)f_j, qw);
              mz_interpolate_spec(qw, f_i, f_j, f_d, tps, c);
              bool doubt = false;
              px[0] = mz_spec_byte(L, A, B, c[0], doubt);
              px[1] = mz_spec_byte(L, A, B, c[1], doubt);
              px[2] = mz_spec_byte(L, A, B, c[2], doubt);
              redo |= (unsigned long long)doubt << (d_v - r0);
            } else {
              if (F32) mz_interpolate_f32<PACKED>(tx, f_i, f_j, f_d, tps, c);
              else mz_interpolate<PACKED>(tx, f_i, f_j, f_d, tps, tps, c);
              px[0] = mz_clip_u8(L * (A + B * c[0]));
              px[1] = mz_clip_u8(L * (A + B * c[1]));
              px[2] = mz_clip_u8(L * (A + B * c[2]));
            }
          } else {
            px[0] = 1; px[1] = 1; px[2] = 1;
          }
        }
        if (SPEC && XV_MAZE_REDO_SPREAD) {
          redo_spread(redo, [&](int src_lane, int j, bool have) {
            const MzColumn C = mz_column_of_lane(me, src_lane);      // the item's column lives in its lane's registers
            if (!have) return;
            const int dv2 = r0 + j;
            const void* tx;
            double f_i, f_j, f_d, L, A, B, c[3];
            (void)pixel(C, dv2, tx, f_i, f_j, f_d, L, A, B);
            uint32_t qw[4][4];
            mz_fetch_window<PP>(tx, (int)f_i, (int)f_j, qw);
            mz_interpolate<true, true>(nullptr, f_i, f_j, f_d, tps, tps, c, qw);
            uint8_t* px = lds + (size_t)((threadIdx.x & ~63u) + (unsigned)src_lane) * cstride + (dv2 - c0) * 3;
            px[0] = mz_clip_u8(L * (A + B * c[0]));
            px[1] = mz_clip_u8(L * (A + B * c[1]));
            px[2] = mz_clip_u8(L * (A + B * c[2]));
          });
        }
        if (SPEC) {
          while (redo) {
            const int d_v = r0 + __builtin_ctzll(redo);
            redo &= redo - 1ull;
            const void* tx;
            double f_i, f_j, f_d, L, A, B, c[3];
            (void)pixel(me, d_v, tx, f_i, f_j, f_d, L, A, B);
            uint32_t qw[4][4];
            mz_fetch_window<PP>(tx, (int)f_i, (int)f_j, qw);
            mz_interpolate<true, true>(nullptr, f_i, f_j, f_d, tps, tps, c, qw);
            uint8_t* px = col + (d_v - c0) * 3;
            px[0] = mz_clip_u8(L * (A + B * c[0]));
            px[1] = mz_clip_u8(L * (A + B * c[1]));
            px[2] = mz_clip_u8(L * (A + B * c[2]));
          }
        }
      }
      // (rows mapping: the lanes whose column is in this sub-pass; its pixels are at column (thread - h0) of the chunk)
      const bool mine = !ROWS || ((int)threadIdx.x >= h0 && (int)threadIdx.x < h0 + nsub);
      uint8_t* ocol = ROWS ? lds + (size_t)(mine ? (int)threadIdx.x - h0 : 0) * cstride : col;
      // ---- transparent landmark overlays, far to near :301-318 ----
      for (int q = mine ? n_tr - 1 : -1; q >= 0; --q) {
        RT hd = 0;
        int lid = 0;
#pragma unroll
        for (int z = 0; z < 16; ++z) if (z == q) { hd = tr_dist[z]; lid = tr_id[z]; }
        RT r2 = hd * (RT)cos_hp_f / lf_r;
        if (xv_abs(r2) < eps_r) r2 = r2 > 0 ? eps_r : -eps_r;
        const RT tv = (RT)cmh / r2, bv = (RT)vision_height / r2;
        int s2 = (int)((half_v - (double)tv) / pixel_size), e2 = (int)((half_v + (double)bv) / pixel_size);
        s2 = s2 < c0 ? c0 : s2;
        e2 = e2 > c1 ? c1 : e2;
        RT a2 = (RT)2 * hd / vis_r - (RT)1;
        a2 = a2 < 0 ? (RT)0 : a2;
        a2 = a2 > 1 ? (RT)1 : a2;
        RT tint[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) tint[c] = (RT)0.30 * (((RT)1 - a2) * (RT)MZ_LANDMARK_RGB[lid][c] + a2 * (RT)1);
        for (int d_v = s2; d_v < e2; ++d_v) {
          uint8_t* px = ocol + (d_v - c0) * 3;
          px[0] = mz_clip_u8((1.0 - 0.30) * (double)px[0] + (double)tint[0]);
          px[1] = mz_clip_u8((1.0 - 0.30) * (double)px[1] + (double)tint[1]);
          px[2] = mz_clip_u8((1.0 - 0.30) * (double)px[2] + (double)tint[2]);
        }
      }
      // ---- command bar, maze_continuous_3d.py:23-29,102-107 (its x range is derived from H, as there) ----
      if (P.command_in_observation) {
        const int sx = (int)(0.25 * H), sy = (int)(0.10 * H), ex = (int)(0.25 * H + 0.50 * H), ey = (int)(0.10 * H + 0.05 * W);
        if (mine && d_h >= sx && d_h < ex)
          for (int y = max(sy, c0); y < ey && y < c1; ++y)
            for (int c = 0; c < 3; ++c) ocol[(y - c0) * 3 + c] = (uint8_t)(int)MZ_LANDMARK_RGB[cmd][c];
      }
      __syncthreads();
      // ---- chunk out: column k's rows [c0, c1) are (c1 - c0) * 3 contiguous bytes of the frame ----
      {
        const int ncols = nsub, run = (c1 - c0) * 3;
        uint8_t* gdst = dst + ((size_t)(g0 + h0) * H + c0) * 3;
        if (((H * 3) & 15) == 0 && ((c0 * 3) & 15) == 0 && (run & 15) == 0) {
          const int vpr = run >> 4;
          for (int k = threadIdx.x; k < ncols * vpr; k += blockDim.x) {
            const int c = k / vpr, v = k - c * vpr;
            const uint32_t* sp = reinterpret_cast<const uint32_t*>(lds + (size_t)c * cstride + v * 16);
            uint4 val;
            val.x = sp[0]; val.y = sp[1]; val.z = sp[2]; val.w = sp[3];
#if XV_MAZE_NT_FLUSH
            typedef uint32_t mz_u4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(mz_u4{val.x, val.y, val.z, val.w}, reinterpret_cast<mz_u4*>(gdst + (size_t)c * H * 3 + v * 16));
#else
            *reinterpret_cast<uint4*>(gdst + (size_t)c * H * 3 + v * 16) = val;
#endif
          }
        } else {
          for (int k = threadIdx.x; k < ncols * run; k += blockDim.x) {
            const int c = k / run, v = k - c * run;
            gdst[(size_t)c * H * 3 + v] = lds[(size_t)c * cstride + v];
          }
        }
      }
      __syncthreads();
      }      // h0
    }
  }
  if (command_rgb && threadIdx.x < 3) command_rgb[(size_t)e * 3 + threadIdx.x] = MZ_LANDMARK_RGB[cmd][threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" int xv_maze_create(xv_engine* e, int n_env, int n_task, int NG, int n_cmd, int max_steps, int W, int H,
                              int command_in_observation, double collision_dist, double visibility_3D,
                              const xv_maze_tables* tables, const int32_t* env_task, xv_maze** out) {
  XV_CHECK_ARG(out != nullptr);
  *out = nullptr;
  XV_CHECK_ARG(e && tables && env_task && n_env > 0 && n_task > 0);
  XV_CHECK_ARG(NG >= 3 && NG <= 64 && n_cmd >= 1 && W >= 2 && H >= 2 && W <= 1024 && H <= 1024);
  XV_CHECK_ARG(tables->walls && tables->texts && tables->landmarks && tables->ints && tables->dbl &&
               tables->commands && tables->lm_coord && tables->tex_walls && tables->tex_grounds &&
               tables->tex_ceilings);
  XV_HIP(hipSetDevice(e->device));
  xv_maze* h = new (std::nothrow) xv_maze();
  if (!h) {
    xv_set_error("xv_maze_create: out of host memory");
    return XV_ERR_NOMEM;
  }
  h->eng = e;
  MazeArgs& a = h->a;
  memset(&a, 0, sizeof(a));
  a.T = *tables; a.env_task = env_task;
  a.n_env = n_env; a.n_task = n_task; a.NG = NG; a.n_cmd = n_cmd; a.max_steps = max_steps; a.W = W; a.H = H;
  a.command_in_observation = command_in_observation;
  a.collision_dist = collision_dist; a.visibility = visibility_3D;
  a.err = e->d_err;
  const size_t n = (size_t)n_env;
  hipError_t m = hipMalloc(&a.pos, 16 * n);
  if (m == hipSuccess) m = hipMalloc(&a.ori, 8 * n);
  if (m == hipSuccess) m = hipMalloc(&a.grid, 8 * n);
  if (m == hipSuccess) m = hipMalloc(&a.steps, 4 * n);
  if (m == hipSuccess) m = hipMalloc(&a.cmd_idx, 4 * n);
  if (m == hipSuccess) m = hipMalloc(&a.cmd_age, 4 * n);
  if (m == hipSuccess) m = hipMalloc(&a.need_reset, n);
  if (m == hipSuccess) m = hipMalloc(&a.collision, 8 * n);
  if (m == hipSuccess) m = hipMalloc(&a.fin_pose, 24 * n);
  if (m == hipSuccess) m = hipMalloc(&a.fin_cmd, 4 * n);
  if (m == hipSuccess) m = hipMalloc(&a.fin_flag, n);
  if (m == hipSuccess) m = hipMalloc(&h->move_list, 4 * n);
  if (m == hipSuccess) m = hipMalloc(&h->move_count, 4 * sizeof(int32_t));
  if (m == hipSuccess) m = hipMemsetAsync(h->move_count, 0, 4 * sizeof(int32_t), e->stream);
  if (m == hipSuccess) m = hipMemsetAsync(a.need_reset, 1, n, e->stream);
  if (m == hipSuccess) m = hipMemsetAsync(a.fin_flag, 0, n, e->stream);
  if (m != hipSuccess) {
    xv_set_error("xv_maze_create: device allocation failed: %s", hipGetErrorString(m));
    void* ps[] = {a.pos, a.ori, a.grid, a.steps, a.cmd_idx, a.cmd_age, a.need_reset, a.collision, a.fin_pose,
                  a.fin_cmd, a.fin_flag, h->move_list, h->move_count};
    for (void* p : ps) if (p) (void)hipFree(p);
    delete h;
    return XV_ERR_HIP;
  }
  hipLaunchKernelGGL(maze_reset_kernel, dim3(xv_div_up(n_env, 256)), dim3(256), 0, e->stream, a,
                     (const uint8_t*)nullptr);
  XV_HIP(hipMemsetAsync(a.need_reset, 1, n, e->stream));
  // packed byte textures when the libraries are integer-valued (they are for decoded 8-bit images)
  if (tables->n_tex_walls > 0 && tables->n_tex_grounds > 0 && tables->n_tex_ceilings > 0) {
    int* d_flag = nullptr;
    int h_flag = 1;
    XV_HIP(hipMalloc(&d_flag, sizeof(int)));
    XV_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), e->stream));
    const float* libs[3] = {tables->tex_walls, tables->tex_grounds, tables->tex_ceilings};
    const int cnt[3] = {tables->n_tex_walls, tables->n_tex_grounds, tables->n_tex_ceilings};
    for (int k = 0; k < 3; ++k) {
      const size_t nt = (size_t)cnt[k] * 256 * 256 * 3;
      hipLaunchKernelGGL(maze_tex_integral_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, e->stream, libs[k],
                         nt, d_flag);
    }
    XV_HIP(hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    XV_HIP(hipStreamSynchronize(e->stream));
    XV_HIP(hipFree(d_flag));
    if (h_flag == 0) {
      uint32_t* pk[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // row-major copies, then pair-interleaved
      bool ok = true;
      for (int k = 0; k < 6 && ok; ++k) {
        const size_t words = (size_t)cnt[k % 3] * 256 * MZ_TEX_PITCH + 4;
        ok = hipMalloc(&pk[k], words * sizeof(uint32_t)) == hipSuccess;
        if (ok) {
          const dim3 grid((unsigned)((words + 255) / 256)), block(256);
          if (k < 3) hipLaunchKernelGGL(maze_pack_tex_kernel<false>, grid, block, 0, e->stream, libs[k], pk[k], cnt[k]);
          else hipLaunchKernelGGL(maze_pack_tex_kernel<true>, grid, block, 0, e->stream, libs[k - 3], pk[k], cnt[k - 3]);
        }
      }
      if (ok) {
        a.pk_walls = pk[0]; a.pk_grounds = pk[1]; a.pk_ceilings = pk[2];
        a.pp_walls = pk[3]; a.pp_grounds = pk[4]; a.pp_ceilings = pk[5];
      } else {
        for (int k = 0; k < 6; ++k) if (pk[k]) (void)hipFree(pk[k]);
        (void)hipGetLastError();
      }
    }
  }
  // ray-caster LDS chunk: the whole column if it fits in MAZE_LDS_CHUNK_MAX, else a multiple of 16 rows
  {
    const int threads = maze_rc_threads(W);
    int hc = H;
    if ((size_t)threads * (H * 3 + 4) > MAZE_LDS_CHUNK_MAX) {
      hc = (int)((MAZE_LDS_CHUNK_MAX / threads - 4) / 3) & ~15;
      if (hc < 16) hc = 16;
    }
    if (getenv("XV_MAZE_HC") && atoi(getenv("XV_MAZE_HC")) >= 16) hc = std::min(H, atoi(getenv("XV_MAZE_HC")) & ~15);   // devtools A/B
    a.HC = hc;
  }
  XV_LAUNCH_CHECK();
  *out = h;
  return XV_OK;
}

extern "C" int xv_maze_destroy(xv_maze* h) {
  if (!h) return XV_OK;
  (void)hipSetDevice(h->eng->device);
  (void)hipStreamSynchronize(h->eng->stream);
  MazeArgs& a = h->a;
  void* ps[] = {a.pos, a.ori, a.grid, a.steps, a.cmd_idx, a.cmd_age, a.need_reset, a.collision, a.fin_pose,
                a.fin_cmd, a.fin_flag, (void*)a.pk_walls, (void*)a.pk_grounds, (void*)a.pk_ceilings, (void*)a.pp_walls,
                (void*)a.pp_grounds, (void*)a.pp_ceilings, h->move_list, h->move_count};
  for (void* p : ps) if (p) (void)hipFree(p);
  delete h;
  return XV_OK;
}

static int maze_launch_render(xv_maze* h, uint8_t* frames, float* command_rgb, bool final) {
  const MazeArgs& a = h->a;
  const int threads = maze_rc_threads(a.W);
  // which lanes paint what (xv_maze_set_raycast_mapping): AUTO = rows of a column on packed textures, fp32 and exact filter, every
  // frame size (round 4: exact 14.1 -> 13.4 ms at 256 x 256 but 1.08 -> 1.19 ms at 64 x 64, which stayed on columns until the
  // half-pass chunk gave the rows mapping its third wave per SIMD — round 6, 16,384 frames, columns (pair copy, prefetch) against
  // rows: 32 x 32 0.466 / 0.409 ms, 64 x 64 0.895 / 0.81, 96 x 96 2.27 / 2.23, 128 x 128 3.48 / 3.34, 192 x 192 8.14 / 5.74:
  // profiles/r06_zz27_*)
  const int rows_map = h->raycast_mapping == XV_MAZE_MAP_COLUMNS ? 0 : (h->raycast_mapping == XV_MAZE_MAP_ROWS ? 3 : 2);
  float* crgb = final ? nullptr : command_rgb;
  const bool packed = a.pk_walls != nullptr;
  // FILT of the kernel: 0 / 3 the speculated exact filter on the pair / row-major texture copy, 1 fp32, 2 direct, 5 / 6 = 3 / 1 on rows
  const int filt0 = h->filter != XV_MAZE_FILTER_EXACT ? h->filter : (packed && (size_t)a.W * a.H > 128 * 128 ? 3 : 0);
  int filt = !packed || !rows_map ? filt0 : (filt0 == 1 ? 6 : ((filt0 == 3 || filt0 == 0) ? 5 : filt0));
  if (packed && h->filter == XV_MAZE_FILTER_EXACT && getenv("XV_MAZE_FILT")) {      // devtools A/B: 0 / 3 / 5, all the same bytes
    const int f = atoi(getenv("XV_MAZE_FILT"));
    if (f == 0 || f == 3 || f == 5) filt = f;
  }
  // Rows mapping: the frame chunk holds HALF the pass's columns at a time (two sub-passes per 64-row chunk, the per-column table
  // stays whole), which is what lets a THIRD wave per SIMD be resident: at 256 columns 50,176 instead of 79,872 B of LDS per
  // workgroup (three workgroups per CU instead of two: 10.40 -> 9.70 ms per 16,384 frames of 256 x 256, 2.8 instead of 1.9 waves
  // per SIMD, issue 92 % busy; at 54,272 B — the per-column table at its former 96 bytes — the third workgroup does not become
  // resident; 112 / 96 / 64 columns per sub-pass: 10.10 / 10.13 / 10.30 ms, profiles/r06_zz17_*), at 64 columns 12.9 instead of
  // 19 KB per one-wave workgroup (0.885 -> 0.805 ms at 64 x 64, profiles/r06_zz26_*)
  MazeArgs ka = a;
  ka.NSUB = (filt >= 5 && XV_MAZE_ROWS_HALF) ? threads / 2 : threads;
  const size_t lds_bytes = (((size_t)ka.NSUB * (a.HC * 3 + 4) + 15) & ~(size_t)15) + (size_t)a.H * 16 +
                           (filt >= 5 ? (size_t)threads * sizeof(MzColumn) : 0);
#define MAZE_RC(F, K, Q, B) \
  hipLaunchKernelGGL((maze_raycast_kernel<F, K, Q, B>), dim3(a.n_env), dim3(threads), lds_bytes, h->eng->stream, ka, frames, crgb)
#define MAZE_RC2(F, K)                                                                  \
  do {                                                                                  \
    if (h->typing_numba) { if (filt == 1) MAZE_RC(F, K, 1, true); else if (filt == 2) MAZE_RC(F, K, 2, true);                          \
                           else if (filt == 3) MAZE_RC(F, K, 3, true); else if (filt == 5) MAZE_RC(F, K, 5, true);                     \
                           else if (filt == 6) MAZE_RC(F, K, 6, true); else MAZE_RC(F, K, 0, true); }                                  \
    else { if (filt == 1) MAZE_RC(F, K, 1, false); else if (filt == 2) MAZE_RC(F, K, 2, false);                                        \
           else if (filt == 3) MAZE_RC(F, K, 3, false); else if (filt == 5) MAZE_RC(F, K, 5, false);                                   \
           else if (filt == 6) MAZE_RC(F, K, 6, false); else MAZE_RC(F, K, 0, false); }                                               \
  } while (0)
  if (final) { if (packed) MAZE_RC2(true, true); else MAZE_RC2(true, false); }
  else { if (packed) MAZE_RC2(false, true); else MAZE_RC2(false, false); }
#undef MAZE_RC2
#undef MAZE_RC
  XV_LAUNCH_CHECK();
  return XV_OK;
}

extern "C" int xv_maze_set_precision(xv_maze* h, int filter) {
  XV_CHECK_ARG(h != nullptr && (filter == XV_MAZE_FILTER_EXACT || filter == XV_MAZE_FILTER_F32 || filter == XV_MAZE_FILTER_EXACT_DIRECT));
  h->filter = filter;
  return XV_OK;
}

extern "C" int xv_maze_set_raycast_mapping(xv_maze* h, int mapping) {
  XV_CHECK_ARG(h != nullptr && (mapping == XV_MAZE_MAP_AUTO || mapping == XV_MAZE_MAP_COLUMNS || mapping == XV_MAZE_MAP_ROWS));
  h->raycast_mapping = mapping;
  return XV_OK;
}

extern "C" int xv_maze_set_typing(xv_maze* h, int typing) {
  XV_CHECK_ARG(h != nullptr && (typing == XV_MAZE_TYPING_NUMPY2 || typing == XV_MAZE_TYPING_NUMBA));
  h->typing_numba = typing == XV_MAZE_TYPING_NUMBA;
  return XV_OK;
}

extern "C" int xv_maze_set_move_kernel(xv_maze* h, int kernel) {
  XV_CHECK_ARG(h != nullptr && (kernel == XV_MAZE_MOVE_LANE_PER_ENV || kernel == XV_MAZE_MOVE_NINE_LANES ||
                                kernel == XV_MAZE_MOVE_THREE_LANES || kernel == XV_MAZE_MOVE_AUTO ||
                                kernel == XV_MAZE_MOVE_NINE_LANES_COMPACT));
  h->move_lanes9 = kernel != XV_MAZE_MOVE_LANE_PER_ENV;
  h->move_lanes = (kernel == XV_MAZE_MOVE_NINE_LANES || kernel == XV_MAZE_MOVE_NINE_LANES_COMPACT) ? 9
                  : (kernel == XV_MAZE_MOVE_THREE_LANES ? 3 : 0);
  h->move_compact = kernel == XV_MAZE_MOVE_AUTO ? -1 : (kernel == XV_MAZE_MOVE_NINE_LANES_COMPACT ? 1 : 0);
  return XV_OK;
}

extern "C" int xv_maze_render(xv_maze* h, uint8_t* frames, float* command_rgb) {
  XV_CHECK_ARG(h != nullptr && frames != nullptr);
  return maze_launch_render(h, frames, command_rgb, false);
}

extern "C" int xv_maze_reset(xv_maze* h, const uint8_t* mask, uint8_t* frames, float* command_rgb) {
  XV_CHECK_ARG(h != nullptr);
  hipLaunchKernelGGL(maze_reset_kernel, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream, h->a, mask);
  XV_LAUNCH_CHECK();
  if (frames) return maze_launch_render(h, frames, command_rgb, false);
  return XV_OK;
}

extern "C" int xv_maze_step(xv_maze* h, const void* action, int action_mode, uint8_t* frames, float* reward,
                            uint8_t* terminated, uint8_t* truncated, float* command_rgb, uint8_t* final_frames,
                            int autoreset_mode) {
  XV_CHECK_ARG(h && action && reward && terminated && truncated);
  XV_CHECK_ARG(action_mode >= 0 && action_mode <= 2 && autoreset_mode >= 0 && autoreset_mode <= 2);
  // lanes per env: enough waves to give every SIMD work, no more (the lanes of an env repeat the position arithmetic)
  // (measured at 16,384 envs: 169 us with nine lanes, 225 us with three, 269 us with the lane-per-env kernel)
  const int lanes = h->move_lanes > 0 ? h->move_lanes : (h->a.n_env <= 32768 ? 9 : 3);
  // a one-thread-per-env kernel sorts the batch into envs to walk and envs that cannot leave their position; the
  // nine-lane kernel walks the first kind and its spare workgroups finish the second.  Measured (uniform Discrete16 actions): 16,384 envs 138 -> 99 us, a batch of turning envs 50 -> 11 us; but
  // 6,144 envs 84 -> 96 us and 16,384 envs that all walk 140 -> 148 us: the sorting launch costs ~10 us and only pays
  // when the unsorted walk has more than one wave per SIMD (7 envs per wave, 1,024 SIMDs), hence AUTO's threshold;
  // continuous actions have a walk speed of exactly 0 by accident only: AUTO does not sort them
  const bool compact = h->move_lanes9 && lanes == 9 && h->move_list != nullptr &&
                       (h->move_compact >= 0 ? h->move_compact == 1
                                              : (h->a.n_env >= 10240 && action_mode != XV_MAZE_ACTION_CONTINUOUS));
  if (compact) {
    const int w = h->move_word;
    h->move_word ^= 1;
    hipLaunchKernelGGL(maze_move_sort_kernel, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, h->eng->stream, h->a, action,
                       action_mode, autoreset_mode, h->move_list, h->move_count, w);
    hipLaunchKernelGGL(maze_step9_kernel<9>, dim3(xv_div_up(h->a.n_env, 7) + 1), dim3(64), 0, h->eng->stream, h->a, action,
                       action_mode, reward, terminated, truncated, autoreset_mode, (const int32_t*)h->move_list,
                       (const int32_t*)(h->move_count + 2 * w));
  } else if (h->move_lanes9 && lanes == 9)
    hipLaunchKernelGGL(maze_step9_kernel<9>, dim3(xv_div_up(h->a.n_env, 7)), dim3(64), 0, h->eng->stream, h->a, action,
                       action_mode, reward, terminated, truncated, autoreset_mode, (const int32_t*)nullptr,
                       (const int32_t*)nullptr);
  else if (h->move_lanes9)
    hipLaunchKernelGGL(maze_step9_kernel<3>, dim3(xv_div_up(h->a.n_env, 21)), dim3(64), 0, h->eng->stream, h->a, action,
                       action_mode, reward, terminated, truncated, autoreset_mode, (const int32_t*)nullptr,
                       (const int32_t*)nullptr);
  else
    hipLaunchKernelGGL(maze_step_kernel, dim3(xv_div_up(h->a.n_env, 64)), dim3(64), 0, h->eng->stream, h->a, action,
                       action_mode, reward, terminated, truncated, autoreset_mode);
  XV_LAUNCH_CHECK();
  if (final_frames && autoreset_mode == XV_AUTORESET_SAME_STEP) {
    const int rc = maze_launch_render(h, final_frames, nullptr, true);
    if (rc != XV_OK) return rc;
  }
  if (frames) return maze_launch_render(h, frames, command_rgb, false);
  return XV_OK;
}

extern "C" int xv_maze_get_state(xv_maze* h, double* pos, double* ori, int32_t* grid, int32_t* steps,
                                 int32_t* cmd_idx, int32_t* cmd_age, uint8_t* need_reset, double* collision) {
  XV_CHECK_ARG(h != nullptr);
  const size_t n = (size_t)h->a.n_env;
  hipStream_t s = h->eng->stream;
  if (pos) XV_HIP(hipMemcpyAsync(pos, h->a.pos, 16 * n, hipMemcpyDeviceToDevice, s));
  if (ori) XV_HIP(hipMemcpyAsync(ori, h->a.ori, 8 * n, hipMemcpyDeviceToDevice, s));
  if (grid) XV_HIP(hipMemcpyAsync(grid, h->a.grid, 8 * n, hipMemcpyDeviceToDevice, s));
  if (steps) XV_HIP(hipMemcpyAsync(steps, h->a.steps, 4 * n, hipMemcpyDeviceToDevice, s));
  if (cmd_idx) XV_HIP(hipMemcpyAsync(cmd_idx, h->a.cmd_idx, 4 * n, hipMemcpyDeviceToDevice, s));
  if (cmd_age) XV_HIP(hipMemcpyAsync(cmd_age, h->a.cmd_age, 4 * n, hipMemcpyDeviceToDevice, s));
  if (need_reset) XV_HIP(hipMemcpyAsync(need_reset, h->a.need_reset, n, hipMemcpyDeviceToDevice, s));
  if (collision) XV_HIP(hipMemcpyAsync(collision, h->a.collision, 8 * n, hipMemcpyDeviceToDevice, s));
  return XV_OK;
}

// _agent_grid follows _agent_loc (get_loc_grid, maze_base.py:220-223) when a pose is handed in
__global__ __launch_bounds__(256) void maze_regrid_kernel(MazeArgs P) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= P.n_env) return;
  const double cell_size = P.T.dbl[(size_t)P.env_task[e] * 8];
  P.grid[e] = (int)(P.pos[e] / cell_size);
  P.grid[(size_t)P.n_env + e] = (int)(P.pos[(size_t)P.n_env + e] / cell_size);
}

extern "C" int xv_maze_set_state(xv_maze* h, const double* pos, const double* ori, const int32_t* steps,
                                 const int32_t* cmd_idx, const int32_t* cmd_age, const uint8_t* need_reset) {
  XV_CHECK_ARG(h != nullptr);
  const size_t n = (size_t)h->a.n_env;
  hipStream_t s = h->eng->stream;
  if (pos) {
    XV_HIP(hipMemcpyAsync(h->a.pos, pos, 16 * n, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(maze_regrid_kernel, dim3(xv_div_up(h->a.n_env, 256)), dim3(256), 0, s, h->a);
    XV_LAUNCH_CHECK();
  }
  if (ori) XV_HIP(hipMemcpyAsync(h->a.ori, ori, 8 * n, hipMemcpyDeviceToDevice, s));
  if (steps) XV_HIP(hipMemcpyAsync(h->a.steps, steps, 4 * n, hipMemcpyDeviceToDevice, s));
  if (cmd_idx) XV_HIP(hipMemcpyAsync(h->a.cmd_idx, cmd_idx, 4 * n, hipMemcpyDeviceToDevice, s));
  if (cmd_age) XV_HIP(hipMemcpyAsync(h->a.cmd_age, cmd_age, 4 * n, hipMemcpyDeviceToDevice, s));
  if (need_reset) XV_HIP(hipMemcpyAsync(h->a.need_reset, need_reset, n, hipMemcpyDeviceToDevice, s));
  return XV_OK;
}
