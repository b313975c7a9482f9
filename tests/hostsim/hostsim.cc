// TEST INFRASTRUCTURE ONLY - single-threaded host compilation of the device solver logic: the flat-vector
// headers of fbstab_amd/csrc instantiated for a workgroup of ONE thread (Ctx<1>) and compiled by g++ against
// tests/hostsim/shim/hip/hip_runtime.h, the stand-in for the device environment (lane exchanges return the
// caller's value, barriers are nothing).  It lets the CPU test-suite check the kernels' arithmetic against the
// oracle where no GPU exists.  It is built by tests/ only, is not part of libfbstab_hip.so, and the product
// never loads it: the HIP library has no CPU execution path, and its headers carry no host branch.
#define FB_DENSE_NO_MFMA 1  // (the K assembly on the matrix cores needs four wavefronts; the scalar loop is the same sum)
#include <cstring>
#include <vector>

#include "../../fbstab_amd/csrc/fb_algorithm.h"
#include "../../fbstab_amd/csrc/fb_dense.h"
#include "../../fbstab_amd/csrc/fb_mpc.h"

using namespace fbk;
typedef Ctx<1> C1;

extern "C" {

int hostsim_mpc_solve(int N, int nx, int nu, int nc, const double* Q, const double* R,
                      const double* S, const double* q, const double* r, const double* A,
                      const double* B, const double* c, const double* E, const double* L,
                      const double* d, const double* x0, double* z, double* l, double* v,
                      double* y, const fbstab_options_t* opts, fbstab_solver_out_t* out) {
  MpcLayout lay;
  lay.init(N, nx, nu, nc, 1);
  std::vector<double> lds(lay.lds_doubles, 0.0), ws(lay.ws_doubles, 0.0);
  MpcData D = {Q, R, S, q, r, A, B, c, E, L, d, x0};
  C1 ctx;
  ctx.tid = 0;
  ctx.red = nullptr;
  MpcProblem<C1> p;
  p.bind(lay, D, z, l, v, y, lds.data(), ws.data());
  fbstab_options_t o = *opts;
  fbstab_options_validate(&o);
  Solver<MpcProblem<C1>, C1> s(p, ctx, o);
  s.solve(out);
  return 0;
}

int hostsim_dense_solve(int nz, int nl, int nv, const double* H, const double* f,
                        const double* G, const double* h, const double* A, const double* b,
                        double* z, double* l, double* v, double* y,
                        const fbstab_options_t* opts, fbstab_solver_out_t* out) {
  DenseLayout lay;
  lay.init(nz, nl, nv, 1);
  std::vector<double> lds(lay.lds_doubles, 0.0);
  DenseData D = {H, f, G, h, A, b};
  C1 ctx;
  ctx.tid = 0;
  ctx.red = nullptr;
  DenseProblem<C1> p;
  p.bind(lay, D, z, l, v, y, lds.data());
  fbstab_options_t o = *opts;
  fbstab_options_validate(&o);
  Solver<DenseProblem<C1>, C1> s(p, ctx, o);
  s.solve(out);
  return 0;
}

}  // extern "C"

extern "C" int hostsim_mpc_newton_refined(int N, int nx, int nu, int nc, const double* Q, const double* R,
                                          const double* S, const double* q, const double* r, const double* A,
                                          const double* B, const double* c, const double* E, const double* L,
                                          const double* d, const double* x0, const double* z, const double* l,
                                          const double* v, const double* zb, const double* lb, const double* vb,
                                          double sigma, double alpha, int refine_sweeps, double* out);

// Component probe: one Newton step of the device logic at (x, xbar, sigma).
// out: dz,dl,dv,adz,wz,wl,rz,rl concatenated; returns 0 ok / 2 factor failure.
extern "C" int hostsim_mpc_newton(int N, int nx, int nu, int nc, const double* Q,
                                  const double* R, const double* S, const double* q,
                                  const double* r, const double* A, const double* B,
                                  const double* c, const double* E, const double* L,
                                  const double* d, const double* x0, const double* z,
                                  const double* l, const double* v, const double* zb,
                                  const double* lb, const double* vb, double sigma,
                                  double alpha, double* out) {
  return hostsim_mpc_newton_refined(N, nx, nu, nc, Q, R, S, q, r, A, B, c, E, L, d, x0, z, l, v, zb, lb, vb, sigma,
                                    alpha, 0, out);
}

// The same with `refine_sweeps` refinement sweeps behind the step (MpcProblem::refine_step); out then has
// two more entries behind rl: linear_residual2() before and after.
extern "C" int hostsim_mpc_newton_refined(int N, int nx, int nu, int nc, const double* Q,
                                          const double* R, const double* S, const double* q,
                                          const double* r, const double* A, const double* B,
                                          const double* c, const double* E, const double* L,
                                          const double* d, const double* x0, const double* z,
                                          const double* l, const double* v, const double* zb,
                                          const double* lb, const double* vb, double sigma,
                                          double alpha, int refine_sweeps, double* out) {
  MpcLayout lay;
  lay.init(N, nx, nu, nc, 1);
  std::vector<double> lds(lay.lds_doubles, 0.0), ws(lay.ws_doubles, 0.0);
  std::vector<double> uz(z, z + lay.nz), ul(l, l + lay.nl), uv(v, v + lay.nv), uy(lay.nv);
  MpcData D = {Q, R, S, q, r, A, B, c, E, L, d, x0};
  C1 ctx;
  ctx.tid = 0;
  ctx.red = nullptr;
  MpcProblem<C1> p;
  p.bind(lay, D, uz.data(), ul.data(), uv.data(), uy.data(), lds.data(), ws.data());
  p.load_guess(ctx);
  for (int i = 0; i < lay.nz; i++) p.zb[i] = zb[i];
  for (int i = 0; i < lay.nl; i++) p.lb[i] = lb[i];
  for (int i = 0; i < lay.nv; i++) p.vb[i] = vb[i];
  p.residual(ctx);
  const bool ok = p.newton_step(ctx, sigma, alpha);
  if (ok && refine_sweeps > 0) {
    out[2 * lay.nz + 2 * lay.nl + 2 * lay.nv + lay.nz + lay.nl] = p.linear_residual2(ctx, sigma);
    for (int k = 0; k < refine_sweeps; k++) p.refine_step(ctx, sigma);
    out[2 * lay.nz + 2 * lay.nl + 2 * lay.nv + lay.nz + lay.nl + 1] = p.linear_residual2(ctx, sigma);
  }
  double* o = out;
  auto put = [&](const double* s, int n) { std::memcpy(o, s, n * sizeof(double)); o += n; };
  put(p.dz, lay.nz); put(p.dl, lay.nl); put(p.dv, lay.nv); put(p.adz, lay.nv);
  put(p.wz, lay.nz); put(p.wl, lay.nl); put(p.rz, lay.nz); put(p.rl, lay.nl);
  return ok ? 0 : 2;
}
