// maze_common.h — what maze.hip (move / rules / ray-cast) and maze_agent.hip (the rule-based teacher) share: the kernel
// argument block, the engine-side handle and the reference's action tables.
#pragma once
#include "philox.h"
#include "xv_common.h"

#include <type_traits>

#define MZ_PI 3.1415926     // dynamics.py:7-8, the reference's own truncated constants
#define MZ_TPI 6.2831852

struct MazeArgs {
  xv_maze_tables T;
  const int32_t* env_task;
  double* pos;       // [2][n_env]
  double* ori;
  int32_t* grid;     // [2][n_env]
  int32_t* steps;
  int32_t* cmd_idx;
  int32_t* cmd_age;
  uint8_t* need_reset;
  double* collision;
  int HC;   // rows per LDS chunk of the ray-caster
  int NSUB; // columns of that chunk (rows mapping; set per launch by maze_launch_render)
  // packed RGBX-byte copies of the texture libraries ([n][256][MZ_TEX_PITCH] uint32), nullptr if not integral
  const uint32_t* pk_walls;
  const uint32_t* pk_grounds;
  const uint32_t* pk_ceilings;
  // the same texels with PAIRS of filter rows interleaved ([n][128][MZ_TEX_PITCH][2]): what the fp32 filter fetches (round 4)
  const uint32_t* pp_walls;
  const uint32_t* pp_grounds;
  const uint32_t* pp_ceilings;
  // pose of envs that ended this step, kept for the optional final frame
  double* fin_pose;  // [3][n_env]
  int32_t* fin_cmd;  // [n_env]
  uint8_t* fin_flag; // [n_env]
  uint32_t* err;
  int n_env, n_task, NG, n_cmd, max_steps, W, H, command_in_observation;
  double collision_dist, visibility;
};

struct xv_maze {
  xv_engine* eng;
  MazeArgs a;
  int filter = 0;            // xv_maze_set_precision (XV_MAZE_FILTER_*)
  int raycast_mapping = 0;   // xv_maze_set_raycast_mapping (XV_MAZE_MAP_*)
  bool typing_numba = false; // xv_maze_set_typing
  bool move_lanes9 = true;   // xv_maze_set_move_kernel
  int move_lanes = 0;        // 0: by batch size; 3 or 9: forced (xv_maze_set_move_kernel)
  int move_compact = -1;     // -1: by batch size; 0 / 1: forced (xv_maze_set_move_kernel)
  int32_t* move_list = nullptr;   // [n_env] envs the nine-lane kernel has to walk this step (maze_move_sort_kernel)
  int32_t* move_count = nullptr;  // [2][2] {envs to walk, envs standing still}; the two pairs alternate between steps
  int move_word = 0;
};

static __device__ const double MZ_ACT16[16][2] = {{0.0, 0.5}, {0.05, 0.0}, {-0.05, 0.0}, {0.1, 0.0}, {-0.1, 0.0}, {0.2, 0.0},
                                           {-0.2, 0.0}, {0.3, 0.0}, {-0.3, 0.0}, {0.5, 0.0}, {-0.5, 0.0}, {0.0, 1.0},
                                           {0.05, 1.0}, {-0.05, 1.0}, {0.10, 1.0}, {-0.10, 1.0}};
static __device__ const double MZ_ACT32[32][2] = {
    {0.0, 0.2}, {0.02, 0.0}, {-0.02, 0.0}, {0.05, 0.0}, {-0.05, 0.0}, {0.1, 0.0}, {-0.1, 0.0}, {0.2, 0.0},
    {-0.2, 0.0}, {0.3, 0.0}, {-0.3, 0.0}, {0.4, 0.0}, {-0.4, 0.0}, {0.5, 0.0}, {-0.5, 0.0}, {0.0, 0.5},
    {0.0, 1.0}, {0.02, 0.5}, {0.02, 1.0}, {-0.02, 0.5}, {-0.02, 1.0}, {0.05, 0.5}, {0.05, 1.0}, {-0.05, 0.5},
    {-0.05, 1.0}, {0.10, 0.5}, {0.10, 1.0}, {-0.10, 0.5}, {-0.10, 1.0}, {0.0, -0.2}, {0.1, -0.2}, {-0.1, -0.2}};

__device__ __forceinline__ double mz_angle_norm(double t) {   // dynamics.py:48-54
  while (t > MZ_PI) t -= MZ_TPI;
  while (t < -MZ_PI) t += MZ_TPI;
  return t;
}


__device__ __forceinline__ float xv_abs(float v) { return fabsf(v); }
__device__ __forceinline__ double xv_abs(double v) { return fabs(v); }
__device__ __forceinline__ float xv_floor(float v) { return floorf(v); }
__device__ __forceinline__ double xv_floor(double v) { return floor(v); }
