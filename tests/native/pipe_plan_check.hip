// host-only check of the planning helpers of csrc/xv_pipe.h (built and run by tests/test_pipe_plan.py; no GPU needed:
// nothing here calls the HIP runtime)
#include <cstdio>

#include "xv_pipe.h"

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++fails; printf("FAIL %s: ", #cond); printf(__VA_ARGS__); printf("\n"); } } while (0)

int main() {
  const int periods[] = {2, 4, 6, 8, 10, 16, 20, 32, 64, 128, 200};
  for (int depth = 2; depth <= 4; ++depth)
    for (int period : periods)
      for (int n_steps = 1; n_steps <= 3000; n_steps += (n_steps < 300 ? 1 : 37)) {
        const int cycles = n_steps / period;
        const int U = xv_pipe_pick_unroll(period, cycles, 0, depth);
        if (U == 0) {
          // nothing fits: no U <= cycles with whole steps per stream
          bool any = false;
          const int big = (period >= XV_PIPE_GRAPH_STEPS_BIG ? 1 : XV_PIPE_GRAPH_STEPS_BIG / period) * (depth == 2 ? 1 : depth);
          for (int u = 1; u <= cycles && u <= big; ++u) any = any || (u * period) % depth == 0;
          CHECK(!any || cycles == 0, "depth %d period %d steps %d: a graph set fits but none was chosen", depth, period, n_steps);
          continue;
        }
        CHECK(U >= 1 && U <= cycles, "depth %d period %d steps %d: U = %d", depth, period, n_steps, U);
        CHECK((U * period) % depth == 0, "depth %d period %d: U = %d leaves a ragged stream", depth, period, U);
        CHECK(U * period / depth <= (period > XV_PIPE_GRAPH_STEPS_BIG ? period : XV_PIPE_GRAPH_STEPS_BIG),
              "depth %d period %d: %d steps per stream", depth, period, U * period / depth);
        // sticky: what was chosen is kept for the same call, and for any call it serves within 3 % of the best
        CHECK(xv_pipe_pick_unroll(period, cycles, U, depth) == U, "depth %d period %d steps %d: not stable", depth, period, n_steps);
        for (int other = cycles; other <= cycles * 4 && other > 0; other += (cycles > 3 ? cycles / 3 : 1)) {
          const int V = xv_pipe_pick_unroll(period, other, U, depth);
          CHECK(V > 0 && (V * period) % depth == 0 && V <= other, "depth %d period %d: %d after %d", depth, period, V, U);
          if (V != U) {
            const double kept = xv_pipe_unroll_cost(period, other, U, depth), best = xv_pipe_unroll_cost(period, other, V, depth);
            CHECK(kept > 1.03 * best, "depth %d period %d cycles %d: rebuilt %d -> %d for %.1f against %.1f", depth, period, other, U, V,
                  kept, best);
          }
        }
      }
  // the documented cases
  CHECK(xv_pipe_pick_unroll(32, 62, 0, 2) == 4, "2,000 steps, ring of 32, two streams: %d", xv_pipe_pick_unroll(32, 62, 0, 2));
  CHECK(xv_pipe_pick_unroll(32, 62, 0, 3) % 3 == 0, "2,000 steps, three streams: %d", xv_pipe_pick_unroll(32, 62, 0, 3));
  CHECK(xv_pipe_pick_unroll(32, 2, 0, 3) == 0 && xv_pipe_pick_unroll(32, 2, 0, 2) == 2, "64 steps: three streams cannot, two can");
  CHECK(xv_pipe_pick_unroll(8, 8, 0, 2) == 8, "ring of 8, 64 steps: %d", xv_pipe_pick_unroll(8, 8, 0, 2));
  CHECK(xv_pipe_pick_unroll(32, 0, 0, 2) == 0, "no whole cycle");
  printf("%s (%d failures)\n", fails ? "FAILED" : "ok", fails);
  return fails ? 1 : 0;
}
