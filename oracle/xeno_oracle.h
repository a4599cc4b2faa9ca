/* xeno_oracle.h — CPU restatement of the Xenoverse hot path (TEST INFRASTRUCTURE, not product).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.  The
 * product (xenoverse_amd/, libxeno_hip.so) never does and has no CPU fallback.
 *
 * Every function works on the same struct-of-arrays layout as the device ABI (include/xeno.h), with host
 * pointers, one plain scalar loop over envs, so that a parity test is "same inputs, memcmp the outputs".
 * Pinned against the reference by tests/golden/ (made by oracle/gen_golden.py, which imports
 * /root/reference in the build container).
 */
#ifndef XENO_ORACLE_H_
#define XENO_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Philox4x32-10 (Salmon et al., SC'11); KATs in tests/test_oracle_philox.py */
void xo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* counter convention shared with the device: {gid_lo, gid_hi, tick_lo, purpose | tick_hi<<8} */
void xo_env_draw(uint64_t seed, uint64_t gid, uint64_t tick, uint32_t purpose, uint32_t out[4]);
/* numpy random_sample construction from two 32-bit words: ((a>>5)*2^26 + (b>>6)) / 2^53 */
double xo_u53(uint32_t a, uint32_t b);
void xo_env_draw_sub(uint64_t seed, uint64_t gid, uint64_t tick, uint32_t purpose, uint32_t sub, uint32_t out[4]);
/* Box-Muller pair from two words (float): z0 = r cos(2 pi u2), z1 = r sin(2 pi u2) */
void xo_box_muller(uint32_t a, uint32_t b, float* z0, float* z1);
void xo_box_muller16(uint32_t w, float* z0, float* z1);

typedef struct {
  int n_env, n_task, S, A, s0_max;
  const double* cdf;        /* [n_task][S][A][S] */
  const float* rs;          /* [n_task][S][A][S][2] */
  const int32_t* state_map; /* [n_task][S] */
  const uint64_t* term_mask;/* [n_task][(S+63)/64] */
  const double* s0_cdf;     /* [n_task][s0_max] */
  const int32_t* s0_ids;    /* [n_task][s0_max] */
  const int32_t* max_steps; /* [n_task] */
  const int32_t* env_task;  /* [n_env] */
  /* env state */
  int32_t* state;           /* [n_env] inner state */
  int32_t* steps;           /* [n_env] */
  uint8_t* need_reset;      /* [n_env] */
  uint32_t err_flags;
  uint32_t gid_stride;      /* free-running draws: env i draws as global env gid_base + i * gid_stride (0 = 1): lets a
                             * scattered subset of a large device batch be checked env for env */
} xo_anymdp;

/* first index j with cdf[j] > u (numpy.searchsorted side='right'), clamped to n-1 */
int xo_upper_bound(const double* cdf, int n, double u);

void xo_anymdp_reset_injected(xo_anymdp* h, const uint8_t* mask, const double* u, int32_t* obs);
void xo_anymdp_step_injected(xo_anymdp* h, const int32_t* action, const double* u, const float* z,
                             const double* u_reset, int32_t* obs, float* reward, float* reward_gt,
                             uint8_t* terminated, uint8_t* truncated, int32_t* final_obs, int mode);
/* free-running: draws u, z, u_reset with xo_env_draw(seed, gid_base+i, tick, purpose) as the device does */
void xo_anymdp_reset(xo_anymdp* h, uint64_t seed, uint64_t gid_base, uint64_t tick, const uint8_t* mask,
                     int32_t* obs);
void xo_anymdp_step(xo_anymdp* h, uint64_t seed, uint64_t gid_base, uint64_t tick, const int32_t* action,
                    int32_t* obs, float* reward, float* reward_gt, uint8_t* terminated,
                    uint8_t* truncated, int32_t* final_obs, int mode);
/* multi-threaded twin of xo_anymdp_step used only for the cpu_baseline timing (OpenMP over envs) */
void xo_anymdp_step_mt(xo_anymdp* h, uint64_t seed, uint64_t gid_base, uint64_t tick,
                       const int32_t* action, int32_t* obs, float* reward, float* reward_gt,
                       uint8_t* terminated, uint8_t* truncated, int32_t* final_obs, int mode,
                       int n_threads);
void xo_anymdp_transition_gt(const xo_anymdp* h, const int32_t* action, double* out);
void xo_anymdp_synth(uint64_t seed, int64_t task_index_base, int n_task, int S, int A, int s0_max,
                     double* cdf, float* rs, int32_t* state_map, uint64_t* term_mask, double* s0_cdf,
                     int32_t* s0_ids, int32_t* max_steps);
/* POMDP / MTPOMDP (anymdp_env.py:116-128,148-157): d_act action tokens per step, d_obs observation draws */
typedef struct {
  xo_anymdp* m;
  int n_obs, d_obs, d_act;
  const double* obs_cdf; /* [n_task][d_obs][S][n_obs] */
} xo_anymdp_tok;
void xo_anymdp_tok_reset_injected(xo_anymdp_tok* h, const uint8_t* mask, const double* u_reset,
                                  const double* u_obs_reset, int32_t* obs);
void xo_anymdp_tok_step_injected(xo_anymdp_tok* h, const int32_t* action, const double* u, const float* z,
                                 const double* u_obs, const double* u_reset, const double* u_obs_reset,
                                 int32_t* obs, float* reward, float* reward_gt, uint8_t* terminated,
                                 uint8_t* truncated, int32_t* final_obs, int mode);
void xo_anymdp_tok_reset(xo_anymdp_tok* h, uint64_t seed, uint64_t gid_base, uint64_t tick, const uint8_t* mask,
                         int32_t* obs);
void xo_anymdp_tok_step(xo_anymdp_tok* h, uint64_t seed, uint64_t gid_base, uint64_t tick, const int32_t* action,
                        int32_t* obs, float* reward, float* reward_gt, uint8_t* terminated, uint8_t* truncated,
                        int32_t* final_obs, int mode);
int xo_max_threads(void);
/* xo_anymdp_synth for the tasks task_index_base + k * task_stride, k < n_task */
void xo_anymdp_synth_strided(uint64_t seed, int64_t task_index_base, int64_t task_stride, int n_task, int S, int A,
                             int s0_max, double* cdf, float* rs, int32_t* state_map, uint64_t* term_mask, double* s0_cdf,
                             int32_t* s0_ids, int32_t* max_steps);

/* ---- task-sampler arithmetic (xeno_oracle_sampler.c) ---- */
double xo_np_pairwise_sum(const double* a, int64_t n);
int xo_update_value_matrix(const double* t_mat, const double* r_mat, int ns, int na, double gamma, double* vm,
                           int is_greedy);
/* value iteration for every task of an oracle handle, as xv_anymdp_solve runs it (xeno_oracle_sampler.c part 3) */
void xo_anymdp_solve(const xo_anymdp* h, double gamma, double tol, int max_iter, double* q_out, uint8_t* greedy_out,
                     int32_t* iters_out);
/* one candidate of the device task sampler (counter-based draws); see xeno_oracle_sampler.c part 2 */
typedef struct {
  int32_t status, goal, n_s0, repair_rounds;
  int32_t s0[4];
  int32_t sweeps[8];       /* repair rounds 0..4, then acceptance greedy [5] and uniform [6] */
  int32_t band_lo[256], band_hi[256], state_map[256];
  uint8_t s_e[256];
  double max_steps, gini, ent, gap_min;
  double s0_prob[4];
} xo_cand_info;
void xo_anymdp_sample_observation_model(uint64_t seed, int64_t task_base, int n_task, int S, int n_obs, int d_obs, double density,
                                        double maximum_distribution, double* obs_cdf);
int xo_anymdp_sample_candidate(uint64_t seed, uint64_t cand, int ns, int na, double* T, double* R, double* noise,
                               xo_cand_info* info);

/* ---------------------------------------------------------------------------------------------
 * LinDS — reference: linds/linds_env.py.  fp32 arithmetic in the device's fixed operation order.
 * Batch-wide padded dims NS (state), NA (= pad_action_dim), NO (= pad_observation_dim = pad_command_dim).
 * Matrices are stored transposed (k-major), as the device reads them.
 * ------------------------------------------------------------------------------------------- */
#define XO_LINDS_KMAX 6
typedef struct {
  int n_env, n_task, NS, NA, NO, NI;
  const float* phiT;   /* [n_task][NS][NS]  phiT[k][j] = Phi[j][k] */
  const float* gamT;   /* [n_task][NA][NS]  gamT[k][j] = Gamma[j][k] */
  const float* cT;     /* [n_task][NS][NO]  cT[k][j]   = C[j][k] */
  const float* xt;     /* [n_task][NS]      X * dt */
  const float* y0;     /* [n_task][NO]      ld_Y */
  const float* valid;  /* [n_task][NO]      target_valid (0/1) */
  const float* cmd0;   /* [n_task][NO]      static command */
  const float* four_coef;    /* [n_task][KMAX][NO][2] */
  const double* four_omega;  /* [n_task][KMAX] */
  const double* four_period; /* [n_task] RandomFourier.max_steps */
  const float* scal;   /* [n_task][8]: action_cost, reward_base, terminate_punish, reward_factor,
                                       noise_drift*dt, dt, 0, 0 */
  const int32_t* ints; /* [n_task][4]: max_steps, target_delay, n_init, four_n (0 = static target) */
  const float* init;   /* [n_task][NI][NS] initial states */
  const int32_t* env_task;
  float* x;            /* [NS][n_env] component-major state */
  int32_t* steps;
  uint8_t* need_reset;
  uint32_t err_flags;
} xo_linds;

/* command at integer time t (linds_env.py:93-98, utils/random_nn.py:362-368), times target_valid */
void xo_linds_cmd(const xo_linds* h, int task, int t, float* out /*[NO]*/);
void xo_linds_reset_injected(xo_linds* h, const uint8_t* mask, const int32_t* init_index, float* obs,
                             float* cmd, float* error);
void xo_linds_step_injected(xo_linds* h, const float* action /*[n_env][NA]*/, const float* z /*[NS][n_env]*/,
                            const int32_t* init_index, float* obs /*[n_env][NO]*/, float* reward,
                            uint8_t* terminated, uint8_t* truncated, float* cmd /*[n_env][NO]*/,
                            float* error, float* final_obs /*nullable*/, int mode);
void xo_linds_reset(xo_linds* h, uint64_t seed, uint64_t gid_base, uint64_t tick, const uint8_t* mask,
                    float* obs, float* cmd, float* error);
void xo_linds_step(xo_linds* h, uint64_t seed, uint64_t gid_base, uint64_t tick, const float* action,
                   float* obs, float* reward, uint8_t* terminated, uint8_t* truncated, float* cmd,
                   float* error, float* final_obs, int mode, int n_threads);
/* k order of the observation product y = C x (the order an MFMA 32x32x2 chain visits k when x' is consumed
 * straight from the accumulator layout of the previous product): fills ord[NS], returns NS */
int xo_linds_yorder(int NS, int* ord);

/* ---------------------------------------------------------------------------------------------
 * CartPole — reference: metacontrol/random_cartpole.py (set_task :46-50, step :52-61, reset :63-75) over
 * gymnasium's CartPoleEnv.step (third-party, not vendored: PARITY UNPINNED, restated from the public
 * gymnasium 1.x source equations).  fp32 state, component-major [4][n_env].
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int n_env, n_task, frameskip, max_steps; /* max_steps <= 0: never truncates (the reference registers no TimeLimit) */
  const double* params;    /* [n_task][4]: gravity, masscart, masspole, length */
  const double* reset_scale; /* [4] reset_bounds_scale */
  const int32_t* env_task;
  double* state;           /* [4][n_env]: x, x_dot, theta, theta_dot (float64, as gymnasium keeps it) */
  int32_t* steps;
  uint8_t* need_reset;
  uint32_t err_flags;
} xo_cartpole;
void xo_cartpole_reset_injected(xo_cartpole* h, const uint8_t* mask, const double* u /*[4][n_env] in [0,1)*/,
                                float* obs /*[n_env][4]*/);
void xo_cartpole_step_injected(xo_cartpole* h, const int32_t* action, const double* u_reset, float* obs,
                               float* reward, uint8_t* terminated, uint8_t* truncated, float* final_obs, int mode);
void xo_cartpole_reset(xo_cartpole* h, uint64_t seed, uint64_t gid_base, uint64_t tick, const uint8_t* mask, float* obs);
void xo_cartpole_step(xo_cartpole* h, uint64_t seed, uint64_t gid_base, uint64_t tick, const int32_t* action,
                      float* obs, float* reward, uint8_t* terminated, uint8_t* truncated, float* final_obs, int mode);

/* ---------------------------------------------------------------------------------------------
 * Acrobot — reference: metacontrol/random_acrobot.py (_dsdt :58-96, _terminal :98-101, set_task :103-106,
 * step :108-117, reset :119-130) over gymnasium's AcrobotEnv.step / rk4 / wrap / bound (third-party, not vendored:
 * that part is PARITY UNPINNED, restated from the public gymnasium 1.x source).  The reference's own _dsdt and
 * _terminal ARE pinned (tests/golden/acrobot_dsdt.npz, generated by calling them).  fp64 state, [4][n_env].
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int n_env, n_task, frameskip, max_steps; /* max_steps <= 0: never truncates (the reference registers no TimeLimit) */
  int scale_is_vector;      /* reset_bounds_scale given as a list (float64 product) or a scalar (float32 product) */
  const double* params;     /* [n_task][7]: link_length_1, link_length_2, link_mass_1, link_mass_2, link_com_1,
                               link_com_2, gravity (sample_acrobot :14-39) */
  const double* reset_scale;/* [4] */
  const int32_t* env_task;
  double* state;            /* [4][n_env]: theta1, theta2, dtheta1, dtheta2 */
  uint8_t* fresh;           /* state still holds the float32 reset values (the reference's obs is then float32 math) */
  int32_t* steps;
  uint8_t* need_reset;
  uint32_t err_flags;
} xo_acrobot;
void xo_acrobot_dsdt(const double prm[7], const double s_aug[5], double out[5]);
int xo_acrobot_terminal(const double prm[7], const double s[4]);
void xo_acrobot_reset_injected(xo_acrobot* h, const uint8_t* mask, const double* u /*[4][n_env] in [0,1)*/,
                               float* obs /*[n_env][6]*/);
void xo_acrobot_step_injected(xo_acrobot* h, const int32_t* action, const double* u_reset, float* obs,
                              float* reward, uint8_t* terminated, uint8_t* truncated, float* final_obs, int mode);
void xo_acrobot_reset(xo_acrobot* h, uint64_t seed, uint64_t gid_base, uint64_t tick, const uint8_t* mask, float* obs);
void xo_acrobot_step(xo_acrobot* h, uint64_t seed, uint64_t gid_base, uint64_t tick, const int32_t* action,
                     float* obs, float* reward, uint8_t* terminated, uint8_t* truncated, float* final_obs, int mode);

/* ---------------------------------------------------------------------------------------------
 * MazeWorld — reference: mazeworld/envs/dynamics.py (move/collision), maze_base.py (rules),
 * maze_continuous_3d.py (do_action, update_observation), ray_caster_utils.py (DDA_2D, interpolate, maze_view).
 * Pose in fp64 as the reference; ray-caster with the reference's mixed f32/f64 typing (see xeno_oracle.c).
 * ------------------------------------------------------------------------------------------- */
#define XO_MAZE_LMAX 15
typedef struct {
  int n_env, n_task, NG, n_cmd, max_steps, W, H, command_in_observation;
  double collision_dist, visibility;
  const int8_t* walls;       /* [n_task][NG][NG] cell_walls (pitch NG) */
  const int32_t* texts;      /* [n_task][NG][NG] cell_texts */
  const int8_t* landmarks;   /* [n_task][NG][NG] cell_landmarks (-1: none) */
  const int32_t* ints;       /* [n_task][8]: n, start_i, start_j, ground_text, ceiling_text, n_landmarks, 0, 0 */
  const double* dbl;         /* [n_task][8]: cell_size, wall_height, agent_height, fol_angle, step_reward,
                                             goal_reward, collision_reward, 0 */
  const int32_t* commands;   /* [n_task][n_cmd] commands_sequence */
  const int32_t* lm_coord;   /* [n_task][LMAX][2] landmarks_coordinates */
  const float* tex_walls;    /* [n_tex][256][256][3] */
  const float* tex_grounds;
  const float* tex_ceilings;
  const int32_t* env_task;
  double* pos;               /* [2][n_env] */
  double* ori;               /* [n_env] */
  int32_t* grid;             /* [2][n_env] */
  int32_t* steps;
  int32_t* cmd_idx;
  int32_t* cmd_age;          /* _commands_exists */
  uint8_t* need_reset;
  double* collision;         /* [n_env] accumulated |force| of the last move (info only) */
} xo_maze;
void xo_maze_reset(xo_maze* h, const uint8_t* mask);
/* action: (turn_rate, walk_speed) double[n_env][2] as handed to do_action (maze_continuous_3d.py:49) */
void xo_maze_step(xo_maze* h, const double* action, float* reward, uint8_t* terminated, uint8_t* truncated,
                  int mode);
/* frames uint8[n_env][W][H][3] (maze_view + astype uint8); command RGB float[n_env][3] (info["command"]) */
void xo_maze_render(const xo_maze* h, uint8_t* frames, float* command_rgb, int n_threads);
/* typing_f64 != 0: DDA_2D and the wall-column geometry in float64, as numba types the reference's source (unpinned: numba
 * is not installed here; used to measure the distance between the two typings) */
void xo_maze_render_typed(const xo_maze* h, uint8_t* frames, float* command_rgb, int n_threads, int typing_f64);
/* cell_exposed of maze_view (ray_caster_utils.py:153,250-255): uint8[n_env][NG][NG]; see xeno_oracle.c */
void xo_maze_expose(const xo_maze* h, uint64_t seed, uint64_t gid_base, uint64_t tick, double prob, uint8_t* exposed);

/* SmartSLAMAgent / OracleAgent (mazeworld/agents/agent_base.py:10-107, smart_slam_agent.py:105-231, oracle_agent.py):
 * the rule-based teacher.  One agent per env, a new one whenever the env starts an episode (steps == 0). */
#define XO_AGENT_STM_MAX 8
typedef struct {
  const xo_maze* env;
  int stm_size;              /* short_term_memory_size (3) */
  int oracle_agent;          /* OracleAgent: long-term memory all ones */
  double keep_ratio;         /* memory_keep_ratio (1.0) */
  int na;                    /* 16 or 32 */
  const double* actions;     /* [na][2] maze_env.list_actions: (turn_rate / PI, walk_speed) */
  uint8_t* stm;              /* [n_env][STM_MAX][NG*NG], oldest first */
  int32_t* stm_len;          /* [n_env] */
  uint8_t* ltm;              /* [n_env][NG*NG] */
  uint8_t* mask;             /* [n_env][NG*NG] _mask_info after the last act */
  double* cost;              /* [n_env][NG*NG] _cost_map after the last act */
  int32_t* path;             /* [n_env][5]: len(path), path[0], path[1] (-1 if absent) after the last act */
} xo_maze_agent;
/* agent.step(): update_common_info + policy.  exposed uint8[n_env][NG*NG] = maze_core._cell_exposed of the present
 * observation; u_keep double[n_env][NG*NG] in [0,1) or NULL (only read when keep_ratio < 1).  action int32[n_env] */
void xo_maze_agent_act(xo_maze_agent* a, const uint8_t* exposed, const double* u_keep, int32_t* action);
/* search_optimal_action (dynamics.py:126-156), exposed for unit tests; targ2 NULL = None */
int xo_maze_search_action(double ori, const double targ1[2], const double* targ2, const double* actions, int na);

/* one move, exposed for unit tests: dynamics.py:158-187 */
void xo_maze_move(double* ori, double pos[2], double turn_rate, double walk_speed, const int8_t* walls, int n,
                  int NG, double cell_size, double col_dist, double* collision);

#ifdef __cplusplus
}
#endif
#endif
