// anymdp_cutline.h — which next states a bucket line of the AnyMDP step engine lists (round 4).
//
// A bucket line (row, k) serves every draw u in [lo, hi) = [k / NBK, (k + 1) / NBK) of the categorical
// s' = #{cdf <= u}  (numpy.random.choice, anymdp_env.py:99-100) from ONE 128-byte read.  Rounds 2-3 stored the 7
// CONSECUTIVE entries that start at #{cdf <= lo}; on the reference sampler's rows (task_sampler_utils.py:65-175: a few
// large probabilities among many of 1e-8 ... 1e-300) 2-3 % of the draws found their line exhausted by entries that are
// practically never drawn, and the whole wave left for the two-line fence search.  A line now holds K CUTS instead:
//
//     the next states whose interval meets [lo, hi) — "live" ones only, cdf[j] > cdf[j-1]; a state of probability zero
//     cannot be drawn and is skipped — are partitioned, in order, into at most K groups (+ an unlisted tail).  A group is
//     PURE (one live state j: its interval [cut before, cdf[j]) answers s' = j exactly) or DIRTY (a run of states lumped
//     together: a draw that lands there takes the fence search).  Unit c of the line = {cut_c = cdf of the group's last
//     state, reward pair of its state}; c = #{cut <= u} names the group.  The partition minimises the probability mass of
//     the dirty groups and the tail (a dynamic programme over the live states, linear in their number).
//
// On the golden 64x8 task and five more sampled with the reference's sampler the mass that still needs the fence search is
// 5e-8 of the draws (K = 7, 16 buckets; 2.7e-2 with consecutive entries): 0.004 draws per launch of 65,536 envs.
//
// Plain C++ (no HIP types): tests/test_host_cutline.py compiles this header for the host and checks every line against
// a brute-force search.  The device kernel (anymdp.hip: anymdp_build_cutlines_kernel) calls the same function.
#pragma once
#ifndef XV_HD
#define XV_HD __host__ __device__
#endif

#define XV_CUT_MAXK 14    // cuts per line: transition lines 7 (narrow metadata: S <= 256 and observation ids <= 255) or 6
                          // (wide); observation lines 14 (8-byte cuts only, one-byte symbol ids)
#define XV_CUT_MAXL 64    // live states considered per bucket; a longer run ends in the unlisted tail

struct XvCutLine {
  double cut[XV_CUT_MAXK];   // non-decreasing; 2.0 (never <= u) for unused groups and for the virtual entry past the row
  int state[XV_CUT_MAXK];    // s' of a pure group (clamped to S - 1, as the search clamps); 0 for dirty / unused groups
  int entry[XV_CUT_MAXK];    // row entry whose reward pair the unit carries (= state)
  unsigned dirty;            // bit c: group c is a lumped run -> fence search
  int n_groups;
  double dirty_mass;         // probability (within [lo, hi)) of a draw this line cannot answer
};

// CDF: callable int -> double, the inclusive CDF entry of next state j < S (non-decreasing).
template <class CDF>
XV_HD inline void xv_cutline_build(const CDF& cdf, int S, double lo, double hi, int K, XvCutLine& out) {
  for (int c = 0; c < XV_CUT_MAXK; ++c) { out.cut[c] = 2.0; out.state[c] = 0; out.entry[c] = 0; }
  out.dirty = 0u; out.n_groups = 0; out.dirty_mass = 0.0;
  // I = #{cdf <= lo}: the first state a draw of this bucket can return
  int I = 0, n = S;
  while (n > 0) {
    const int half = n >> 1;
    if (cdf(I + half) <= lo) { I += half + 1; n -= half + 1; } else n = half;
  }
  // live states from I on, up to the one whose interval reaches hi.  Entry S is virtual: cdf 2.0, state S - 1 — what a
  // clamped s' gets when u >= cdf[S-1] (a caller-supplied row whose last entry stays below 1)
  unsigned short jj[XV_CUT_MAXL];
  double pre[XV_CUT_MAXL];      // min(cdf[j], hi) - lo: the mass of [lo, hi) up to and including this state
  int m = 0;
  bool more = false;
  double prevc = lo;
  for (int j = I; j <= S; ++j) {
    if (prevc >= hi) break;
    const double c = j < S ? cdf(j) : 2.0;
    if (c > prevc) {
      if (m < XV_CUT_MAXL) { jj[m] = (unsigned short)j; pre[m] = (c < hi ? c : hi) - lo; ++m; } else more = true;
      prevc = c;
    }
  }
  auto emit = [&](int q, int j, bool dirty) {
    out.cut[q] = j < S ? cdf(j) : 2.0;
    if (dirty) { out.dirty |= 1u << q; }
    else { out.state[q] = j < S ? j : S - 1; out.entry[q] = out.state[q]; }
  };
  // a count that names an unused group (cut 2.0) must never be taken for an answer: unused groups are dirty
  auto close = [&](int used) { out.n_groups = used; out.dirty |= ((1u << K) - 1u) & ~((1u << used) - 1u); };
  if (!more && m <= K) {
    for (int q = 0; q < m; ++q) emit(q, jj[q], false);
    close(m);
    return;
  }
  // f[g][b]: least dirty mass with g groups closed or open over the states seen so far; b = 1: the last group is a dirty
  // run (a further state may join it at no extra group).  dec[i]: how the best f after state i was reached.
  const double INF = 1.0e300, total = hi - lo;
  double f[XV_CUT_MAXK + 1][2];
  unsigned dec[XV_CUT_MAXL];            // bit g: the pure group g - 1 follows a dirty run; bit 16 + g: a dirty run opens group g - 1
  for (int g = 0; g <= K; ++g) f[g][0] = f[g][1] = INF;
  f[0][0] = 0.0;
  double best = total;          // nothing listed: every draw of the bucket is beyond the line
  int bi = -1, bg = 0, bb = 0;
  for (int i = 0; i < m; ++i) {
    const double w = pre[i] - (i ? pre[i - 1] : 0.0);
    double nf[XV_CUT_MAXK + 1][2];
    unsigned d = 0;
    for (int g = 0; g <= K; ++g) nf[g][0] = nf[g][1] = INF;
    for (int g = 1; g <= K; ++g) {
      // state i alone in group g - 1 (pure): after a pure group or after a dirty run
      if (f[g - 1][1] < f[g - 1][0]) { nf[g][0] = f[g - 1][1]; d |= 1u << g; } else nf[g][0] = f[g - 1][0];
      // state i in a dirty run: joins the open run (same g) or opens group g - 1 after a pure group
      const double ext = f[g][1], neu = f[g - 1][0];
      if (neu < ext) { nf[g][1] = neu + w; d |= 1u << (16 + g); } else if (ext < INF) nf[g][1] = ext + w;
    }
    dec[i] = d;
    for (int g = 0; g <= K; ++g) { f[g][0] = nf[g][0]; f[g][1] = nf[g][1]; }
    const double tail = total - pre[i];      // the states after i (and any beyond the cap) stay unlisted
    for (int g = 1; g <= K; ++g)
      for (int b = 0; b < 2; ++b)
        if (f[g][b] < INF && f[g][b] + tail < best) { best = f[g][b] + tail; bi = i; bg = g; bb = b; }
  }
  out.dirty_mass = best > 0.0 ? best : 0.0;
  close(bi < 0 ? 0 : bg);
  // walk the decisions back; the first state met of a group (from the end) is its last state: its cdf is the cut
  int g = bg, b = bb, open = -1;
  for (int i = bi; i >= 0; --i) {
    const unsigned d = dec[i];
    const int q = g - 1;
    if (b == 0) {
      emit(q, jj[i], false);
      b = (d >> g) & 1u;
      g -= 1;
      open = -1;
    } else {
      if (open != q) { emit(q, jj[i], true); open = q; }
      if ((d >> (16 + g)) & 1u) { b = 0; g -= 1; open = -1; }
    }
  }
}

// what the step kernel computes from a line: c = #{cut <= u}; -1: the draw is beyond the line (c == K, or a dirty group)
XV_HD inline int xv_cutline_resolve(const XvCutLine& L, int K, double u) {
  int c = 0;
  for (int q = 0; q < K; ++q) c += L.cut[q] <= u ? 1 : 0;
  if (c >= K || ((L.dirty >> c) & 1u)) return -1;
  return L.state[c];
}
