// Developer study (CPU): how stable is the elimination order Eigen's rule picks from one Newton step of a
// dense QP to the next?  Compiled INTO a private build of the oracle (-include this file, see
// tools/ldlt_order_study.py); per thread, the order of each factorisation is compared with the order of
// the factorisation before it.  A new QP is recognised by the caller resetting through fbo_obs_new_qp()
// being impossible from inside the solve, so the first factorisation of a solve is recognised by K's
// (0,0) entry changing sign of nothing - instead the study runs ONE QP per call and one thread, and the
// Python side calls fbo_obs_mark() between QPs.
#pragma once
#include <vector>
#include <cstring>
namespace fbo_obs {
struct State {
  std::vector<int> prev;
  bool have_prev = false;
  long long factorisations = 0, same_as_previous = 0, first_of_qp = 0;
  long long prefix_hist[65] = {0};  // length of the common prefix with the previous order
  long long zfirst = 0;              // orders that take all of the leading block first
  int nz = 0;
};
inline State& st() { static State s; return s; }
template <class L>
inline void observe(const L& f) {
  State& s = st();
  const int n = f.n;
  std::vector<int> perm(n);
  for (int i = 0; i < n; i++) perm[i] = i;
  for (int k = 0; k < n; k++) std::swap(perm[k], perm[f.transp[k]]);  // perm[k] = original index eliminated at step k
  s.factorisations++;
  bool zf = true;
  for (int k = 0; k < s.nz && k < n; k++) zf = zf && perm[k] < s.nz;
  s.zfirst += zf;
  if (s.have_prev && (int)s.prev.size() == n) {
    int c = 0;
    while (c < n && perm[c] == s.prev[c]) c++;
    s.prefix_hist[c > 64 ? 64 : c]++;
    s.same_as_previous += c == n;
  } else {
    s.first_of_qp++;
  }
  s.prev = perm;
  s.have_prev = true;
}
// line search: trials evaluated per Newton step (1 = the full step accepted at once; the loop leaves
// after max_linesearch_iters trials whether or not the last one was accepted)
struct LsState { long long hist[32] = {0}; int cur = 0; };
inline LsState& ls() { static LsState s; return s; }
inline void ls_event(int step_done) {
  LsState& s = ls();
  if (!step_done) { s.cur++; return; }   // called once per trial, after it (accepted trials leave before: cur = trials - 1 then)
  s.hist[s.cur > 31 ? 31 : s.cur]++;
  s.cur = 0;
}
}  // namespace fbo_obs
#define FBO_LDLT_OBSERVER(f) fbo_obs::observe(f)
#define FBO_LINESEARCH_OBSERVER(done) fbo_obs::ls_event(done)
extern "C" {
inline void fbo_obs_dummy() {}
}
