/* xeno_oracle_agent.c — CPU restatement of the MazeWorld rule-based teacher.  TEST INFRASTRUCTURE ONLY (see
 * xeno_oracle.h): nothing under xenoverse_amd/ links or calls this.
 *
 * Follows xenoverse/mazeworld/agents/agent_base.py:10-107 (memory, valid_neighbors, update_common_info),
 * smart_slam_agent.py:105-231 (update_cost_map, policy, path_to_action, retrieve_path, exploration,
 * navigate_landmarks_navigate), oracle_agent.py (long-term memory of ones), utils/tools.py:9-34 (conv2d_numpy with the
 * 5x5 exploration kernel of smart_slam_agent.py:12-20) and envs/dynamics.py:126-156 (search_optimal_action) with
 * :98-123 (vector_move_no_collision) and :48-54 (angle_normalization).  The cost map is computed with the reference's
 * own FIFO label-correcting loop, statement by statement in meaning (a queue of cells, relax the valid neighbours of the
 * popped cell), so that the device's parallel relaxation is checked against the reference's order of operations.
 * Pinned by tests/golden/maze_agent_*.npz (trajectories of the reference's SmartSLAMAgent / OracleAgent). */
#include "xeno_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define AG_PI 3.1415926 /* dynamics.py:7-8 */
#define AG_TPI 6.2831852

static const int AG_NB[8][2] = {{-1, 0}, {1, 0}, {0, 1}, {0, -1}, {-1, -1}, {-1, 1}, {1, -1}, {1, 1}}; /* agent_base.py:27 */

typedef struct {
  int n, NG;
  const int8_t* walls;
  const int8_t* lm;
  const uint8_t* mask;
} ag_view;

static inline int ag_god(const ag_view* v, int x, int y) { /* agent_base.py:23 */
  return 1 - (int)v->walls[x * v->NG + y] + (int)v->lm[x * v->NG + y];
}
static inline int ag_mask(const ag_view* v, int x, int y) { return v->mask[x * v->NG + y] != 0; }

/* agent_base.py:48-71; returns the count, offsets in out */
static int ag_valid_neighbors(const ag_view* v, int cx, int cy, int self_included, int mask_included, int out[9][2]) {
  int c = 0;
  if (self_included) { out[c][0] = 0; out[c][1] = 0; ++c; }
  for (int q = 0; q < 8; ++q) {
    const int dx = AG_NB[q][0], dy = AG_NB[q][1], nx = cx + dx, ny = cy + dy;
    if (nx < 0 || nx >= v->n || ny < 0 || ny >= v->n) continue;
    if (!ag_mask(v, nx, ny) && !mask_included) continue;
    if (ag_god(v, nx, ny) < 0 && ag_mask(v, nx, ny)) continue;
    if (dx * dy == 0) { out[c][0] = dx; out[c][1] = dy; ++c; }
    else if (ag_god(v, nx, cy) > -1 && ag_god(v, cx, ny) > -1 && ag_mask(v, nx, cy) && ag_mask(v, cx, ny)) {
      out[c][0] = dx; out[c][1] = dy; ++c;
    }
  }
  return c;
}

static double ag_angle_norm(double t) { /* dynamics.py:48-54 */
  while (t > AG_PI) t -= AG_TPI;
  while (t < -AG_PI) t += AG_TPI;
  return t;
}

/* dynamics.py:126-156 */
int xo_maze_search_action(double ori, const double targ1[2], const double* targ2, const double* actions, int na) {
  int best = 0;
  double best_cost = 0.0;
  for (int k = 0; k < na; ++k) {
    const double a0 = actions[2 * k], a1 = actions[2 * k + 1];
    const double tr = a0 * AG_PI, ws = a1, dt = 1.0;
    /* vector_move_no_collision :98-123 */
    const double d_theta = tr * dt, arc = ws * dt;
    const double c_theta = cos(ori), s_theta = sin(ori), c_dt = cos(0.5 * d_theta), s_dt = sin(0.5 * d_theta);
    const double n_ori = ag_angle_norm(ori + d_theta);
    double dx, dy;
    if (fabs(d_theta) < 1.0e-8) { dx = c_theta * arc; dy = s_theta * arc; }
    else {
      const double rad = ws / tr, offset = 2.0 * s_dt * rad;
      const double c_n = c_theta * c_dt - s_theta * s_dt, s_n = c_theta * s_dt + s_theta * c_dt;
      dx = c_n * offset; dy = s_n * offset;
    }
    const double e0 = dx - targ1[0], e1 = dy - targ1[1];
    const double dist_loss = e0 * e0 + e1 * e1;
    const double dist = sqrt(dist_loss);
    double cost = dist_loss;
    cost += 1.0e-4 * (pow(a0, 2.0) + pow(a1, 2.0)); /* Python float ** 2 */
    const double targ1_ang = atan2(targ1[1], targ1[0]);
    const double delta1 = ag_angle_norm(targ1_ang - n_ori);
    double delta2 = delta1;
    if (targ2) delta2 = ag_angle_norm(atan2(targ2[1], targ2[0]) - n_ori);
    const double f = fmin(dist / 0.2, 1.0);
    cost += delta1 * delta1 * f + delta2 * delta2 * (1 - f);
    if (k == 0 || cost < best_cost) { best = k; best_cost = cost; } /* numpy.argmin: first minimum */
  }
  return best;
}

typedef struct { int* d; size_t head, tail, cap; } ag_queue;
static void ag_put(ag_queue* q, int v) {
  if (q->tail == q->cap) {
    if (q->head > 0) { memmove(q->d, q->d + q->head, (q->tail - q->head) * sizeof(int)); q->tail -= q->head; q->head = 0; }
    if (q->tail == q->cap) { q->cap = q->cap ? 2 * q->cap : 1024; q->d = (int*)realloc(q->d, q->cap * sizeof(int)); }
  }
  q->d[q->tail++] = v;
}

/* smart_slam_agent.py:105-142 */
static void ag_update_cost_map(const ag_view* v, int cx, int cy, const double gf[2], double ori, double* cost) {
  const int NG = v->NG, n = v->n;
  for (int x = 0; x < n; ++x)
    for (int y = 0; y < n; ++y) cost[x * NG + y] = 1.0e+6;
  ag_queue q = {0, 0, 0, 0};
  int nb[9][2];
  int c = ag_valid_neighbors(v, cx, cy, 1, 0, nb);
  for (int k = 0; k < c; ++k) {
    const int i = nb[k][0] + cx, j = nb[k][1] + cy;
    const double d0 = (i + 0.5) - gf[0], d1 = (j + 0.5) - gf[1];
    const double dist = sqrt(d0 * d0 + d1 * d1);
    const double o = 1.0 - (d0 / (dist + 1.0e-3) * cos(ori) + d1 / (dist + 1.0e-3) * sin(ori));
    const double ori_cost = 20.0 * o * fmin(dist, 0.01);
    cost[i * NG + j] = dist + ori_cost;
    ag_put(&q, i * NG + j);
  }
  while (q.head < q.tail) {
    const int o = q.d[q.head++], ox = o / NG, oy = o % NG;
    c = ag_valid_neighbors(v, ox, oy, 0, 1, nb);
    for (int k = 0; k < c; ++k) {
      const int dx = nb[k][0], dy = nb[k][1], nx = ox + dx, ny = oy + dy;
      if (nx >= n || nx < 0 || ny >= n || ny < 0) continue;
      const int c_type = ag_god(v, nx, ny), m_type = ag_mask(v, nx, ny);
      const double dist_cost = sqrt((double)(dx * dx + dy * dy));
      double w;
      if (c_type < 0 && m_type > 0) continue;
      else if (m_type < 1) w = 10 + dist_cost;
      else w = dist_cost;
      if (cost[nx * NG + ny] > cost[o] + w) {
        cost[nx * NG + ny] = cost[o] + w;
        ag_put(&q, nx * NG + ny);
      }
    }
  }
  free(q.d);
}

/* smart_slam_agent.py:171-219; returns len(path) and its first two entries (only those are used, :158-169).  The path
 * grows at the front, so its first three entries are the last three cells found. */
static int ag_retrieve_path(const ag_view* v, const double* cost, int gx, int gy, int cx, int cy, const double gf[2],
                            int p0[2], int p1[2]) {
  const int NG = v->NG, n = v->n;
  int len = 1;
  int a[2] = {gx, gy}, b[2] = {-1, -1}, c3[2] = {-1, -1}; /* path[0], path[1], path[2] */
  double cur = cost[gx * NG + gy];
  int sx = gx, sy = gy;
  int eff[9][2], nb[9][2];
  const int ne = ag_valid_neighbors(v, cx, cy, 1, 0, eff);
  while (sx != cx || sy != cy) {
    int flag = 0;
    for (int k = 0; k < ne; ++k)
      if (sx == cx + eff[k][0] && sy == cy + eff[k][1]) flag = 1;
    if (flag) break;
    double min_cost = cur;
    int mx = -1, my = -1;
    const int c = ag_valid_neighbors(v, sx, sy, 0, 1, nb);
    for (int k = 0; k < c; ++k) {
      const int nx = sx + nb[k][0], ny = sy + nb[k][1];
      if (nx < 0 || nx > n - 1 || ny < 0 || ny > n - 1) continue;
      if (cost[nx * NG + ny] > 1.0e+4) continue;
      if (cost[nx * NG + ny] < min_cost) { min_cost = cost[nx * NG + ny]; mx = nx; my = ny; }
    }
    if (mx > -1) {
      sx = mx; sy = my;
      c3[0] = b[0]; c3[1] = b[1];
      b[0] = a[0]; b[1] = a[1];
      a[0] = sx; a[1] = sy;
      ++len;
      cur = cost[sx * NG + sy];
    } else break; /* "[WARNING] Unexpected error in path retrieving" */
  }
  if (len > 2) { /* :209-218 */
    const double dx = a[0] + 0.5 - gf[0], dy = a[1] + 0.5 - gf[1];
    const double ds = sqrt(dx * dx + dy * dy);
    const double dx2 = b[0] + 0.5 - gf[0], dy2 = b[1] + 0.5 - gf[1];
    const double ds2 = sqrt(dx2 * dx2 + dy2 * dy2);
    if (ds + cost[a[0] * NG + a[1]] > ds2 + cost[b[0] * NG + b[1]] && ds < 0.2) { /* del path[0] */
      --len;
      a[0] = b[0]; a[1] = b[1];
      b[0] = c3[0]; b[1] = c3[1];
    }
  }
  p0[0] = a[0]; p0[1] = a[1]; p1[0] = b[0]; p1[1] = b[1];
  return len;
}

void xo_maze_agent_act(xo_maze_agent* A, const uint8_t* exposed, const double* u_keep, int32_t* action) {
  const xo_maze* h = A->env;
  const int N = h->n_env, NG = h->NG, G2 = NG * NG;
  for (int e = 0; e < N; ++e) {
    const int t = h->env_task[e];
    const int32_t* in = h->ints + (size_t)t * 8;
    const double* db = h->dbl + (size_t)t * 8;
    const int n = in[0];
    const double cell_size = db[0];
    uint8_t* stm = A->stm + (size_t)e * XO_AGENT_STM_MAX * G2;
    uint8_t* ltm = A->ltm + (size_t)e * G2;
    uint8_t* mask = A->mask + (size_t)e * G2;
    double* cost = A->cost + (size_t)e * G2;
    if (h->steps[e] == 0) { /* a new episode gets a new agent (agent_base.py:14-40) */
      A->stm_len[e] = 0;
      memset(ltm, A->oracle_agent ? 1 : 0, (size_t)G2);
    }
    /* ---- update_common_info :73-95 ---- */
    memcpy(stm + (size_t)A->stm_len[e] * G2, exposed + (size_t)e * G2, (size_t)G2);
    A->stm_len[e] += 1;
    if (A->stm_len[e] > A->stm_size) {
      for (int k = 0; k < G2; ++k) {
        const int keep = A->keep_ratio >= 1.0 ? 1 : (u_keep[(size_t)e * G2 + k] < A->keep_ratio);
        ltm[k] = (uint8_t)(ltm[k] || (stm[k] && keep));
      }
      memmove(stm, stm + G2, (size_t)(A->stm_len[e] - 1) * G2);
      A->stm_len[e] -= 1;
    }
    for (int k = 0; k < G2; ++k) {
      int m = ltm[k] != 0;
      for (int q = 0; q < A->stm_len[e]; ++q) m = m || stm[(size_t)q * G2 + k];
      mask[k] = (uint8_t)m;
    }
    const ag_view v = {n, NG, h->walls + (size_t)t * G2, h->landmarks + (size_t)t * G2, mask};
    const double ori = h->ori[e];
    const double loc[2] = {h->pos[e], h->pos[(size_t)N + e]};
    const int cx = h->grid[e], cy = h->grid[(size_t)N + e];
    const double gf[2] = {loc[0] / cell_size, loc[1] / cell_size}; /* maze_base.py:225-228 */
    const int idx = h->cmd_idx[e] < h->n_cmd ? h->cmd_idx[e] : h->n_cmd - 1;
    const int command = h->commands[(size_t)t * h->n_cmd + idx];

    /* ---- policy :144-156 ---- */
    ag_update_cost_map(&v, cx, cy, gf, ori, cost);
    int have = 0, gx = -1, gy = -1;
    for (int x = 0; x < n && !have; ++x) /* navigate_landmarks_navigate :223-230 */
      for (int y = 0; y < n && !have; ++y)
        if (ag_god(&v, x, y) == command + 1 && ag_mask(&v, x, y)) { have = 1; gx = x; gy = y; }
    if (!have) { /* exploration :213-221 */
      double best = 0.0;
      for (int x = 0; x < n; ++x)
        for (int y = 0; y < n; ++y) {
          double wht = 0.0; /* conv2d_numpy, 5x5 ones with 1000 in the middle, zero padding */
          for (int i = x - 2; i <= x + 2; ++i)
            for (int j = y - 2; j <= y + 2; ++j)
              if (i >= 0 && i < n && j >= 0 && j < n)
                wht += (double)(1 - (int)ag_mask(&v, i, j)) * ((i == x && j == y) ? 1000.0 : 1.0);
          const double u = cost[x * NG + y] - wht;
          if ((x == 0 && y == 0) || u < best) { best = u; gx = x; gy = y; }
        }
      have = !(best >= 0);
    }
    int len = 1, p0[2] = {cx, cy}, p1[2] = {-1, -1};
    if (have) {
      len = ag_retrieve_path(&v, cost, gx, gy, cx, cy, gf, p0, p1);
    }
    /* ---- path_to_action :158-169 ---- */
    const double targ1[2] = {p0[0] + 0.5 - gf[0], p0[1] + 0.5 - gf[1]};
    const double targ2[2] = {p1[0] + 0.5 - gf[0], p1[1] + 0.5 - gf[1]};
    action[e] = xo_maze_search_action(ori, targ1, len > 1 ? targ2 : NULL, A->actions, A->na);
    A->path[(size_t)e * 5] = len;
    A->path[(size_t)e * 5 + 1] = p0[0]; A->path[(size_t)e * 5 + 2] = p0[1];
    A->path[(size_t)e * 5 + 3] = len > 1 ? p1[0] : -1; A->path[(size_t)e * 5 + 4] = len > 1 ? p1[1] : -1;
  }
}
