/* host build of xenoverse_amd/csrc/acrobot_wrap.h for tests/test_host_acrobot_wrap.py */
#include "../../xenoverse_amd/csrc/acrobot_wrap.h"

double wrap_fast(double x, int* stuck) { return xv_acrobot_wrap(x, stuck); }

double wrap_loop(double x) { /* gymnasium acrobot.wrap(x, -pi, pi), literally */
  const double m = -3.141592653589793, M = 3.141592653589793;
  const double diff = M - m;
  while (x > M) x = x - diff;
  while (x < m) x = x + diff;
  return x;
}

/* loop == NULL: the plain loop is not run (arguments for which it would take hours) */
void wrap_both(const double* x, int n, double* fast, double* loop, int* stuck) {
  for (int i = 0; i < n; ++i) {
    int s = 0;
    fast[i] = wrap_fast(x[i], &s);
    stuck[i] = s;
    if (loop) loop[i] = s ? x[i] : wrap_loop(x[i]);
  }
}
