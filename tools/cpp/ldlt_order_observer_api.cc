// extern "C" face of the study's counters (linked into the private oracle build only)
#include "ldlt_order_observer.h"
extern "C" {
void fbo_obs_mark(int nz) { fbo_obs::st().have_prev = false; fbo_obs::st().nz = nz; }
void fbo_obs_read(long long* out) {  // factorisations, same, first_of_qp, zfirst, prefix_hist[65]
  fbo_obs::State& s = fbo_obs::st();
  out[0] = s.factorisations; out[1] = s.same_as_previous; out[2] = s.first_of_qp; out[3] = s.zfirst;
  for (int i = 0; i < 65; i++) out[4 + i] = s.prefix_hist[i];
}
void fbo_obs_read_linesearch(long long* out) { for (int i = 0; i < 32; i++) out[i] = fbo_obs::ls().hist[i]; }
}
