// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_linalg.h header).
//
// extern "C" surface of the CPU oracle, loaded with ctypes by tests/, by
// __graft_entry__.smoke() and by bench.py's cpu_baseline leg — never by the
// product library.
//
// Built twice from this one file (oracle/Makefile):
//   oracle/liboracle.so            the restated loop (oracle_algorithm.h)
//   oracle/_ref/libfbstab_ref.so   -DFBO_USE_REFERENCE_ALGORITHM: the loop is
//       the reference's own FBstabAlgorithm<> template, compiled from
//       /root/reference/fbstab/fbstab_algorithm.h (+ -impl.h, tools/utilities.h)
//       where they lie, instantiated over the oracle components.  Those headers
//       are the part of the reference that compiles without Eigen
//       (SURVEY.md F5); nothing is copied and nothing is stubbed.
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "oracle_algorithm.h"

#ifdef FBO_USE_REFERENCE_ALGORITHM
#include <memory>
#include "fbstab/fbstab_algorithm.h"
#include "tools/output_stream.h"
#endif

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

thread_local std::string g_last_error;

#ifdef FBO_USE_REFERENCE_ALGORITHM
fbstab::AlgorithmParameters ToRef(const fbstab_options_t& o) {
  fbstab::AlgorithmParameters p;
  p.sigma0 = o.sigma0;
  p.sigma_max = o.sigma_max;
  p.sigma_min = o.sigma_min;
  p.alpha = o.alpha;
  p.beta = o.beta;
  p.eta = o.eta;
  p.delta = o.delta;
  p.gamma = o.gamma;
  p.abs_tol = o.abs_tol;
  p.rel_tol = o.rel_tol;
  p.stall_tol = o.stall_tol;
  p.infeas_tol = o.infeas_tol;
  p.inner_tol_max = o.inner_tol_max;
  p.inner_tol_min = o.inner_tol_min;
  p.max_newton_iters = o.max_newton_iters;
  p.max_prox_iters = o.max_prox_iters;
  p.max_inner_iters = o.max_inner_iters;
  p.max_linesearch_iters = o.max_linesearch_iters;
  p.check_feasibility = o.check_feasibility != 0;
  p.nonmonotone_linesearch = o.nonmonotone_linesearch != 0;
  p.display_level = static_cast<fbstab::Display>(o.display_level);
  return p;
}
// An output stream of the reference's own kind (tools/output_stream.h:15-37)
// that collects what the reference prints instead of sending it to stdout.
class CaptureOutput : public fbstab::OutputStream<CaptureOutput> {
 public:
  explicit CaptureOutput(std::string* sink) : sink_(sink) {}

 protected:
  void PrintImplementation(const char* message) const {
    if (sink_) sink_->append(message); else printf("%s", message);
  }
  friend class fbstab::OutputStream<CaptureOutput>;

 private:
  std::string* sink_;
};

fbstab_solver_out_t FromRef(const fbstab::SolverOut& s) {
  fbstab_solver_out_t o;
  o.eflag = static_cast<int>(s.eflag);
  o.pad_ = 0;
  o.residual = s.residual;
  o.newton_iters = s.newton_iters;
  o.prox_iters = s.prox_iters;
  o.solve_time = s.solve_time;
  o.initial_residual = s.initial_residual;
  return o;
}
#endif

// One solver workspace (the analogue of an FBstabDense / FBstabMpc object:
// fbstab_dense.cc:18-42, fbstab_mpc.cc:61-89).
template <class LinearSolver>
struct Workspace {
  fbo::FullVariable x1, x2, x3, x4;
  fbo::FullResidual r1, r2;
  fbo::FullFeasibility feas;
  LinearSolver ls;
  fbo::Vec z, l, v, y;
#ifdef FBO_USE_REFERENCE_ALGORITHM
  fbstab::FBstabAlgorithm<fbo::FullVariable, fbo::FullResidual, LinearSolver,
                          fbo::FullFeasibility>
      alg;
#else
  fbo::Algorithm<LinearSolver> alg;
#endif
  template <class... LsArgs>
  Workspace(int nz, int nl, int nv, LsArgs... a)
      : x1(nz, nl, nv), x2(nz, nl, nv), x3(nz, nl, nv), x4(nz, nl, nv),
        r1(nz, nl, nv), r2(nz, nl, nv), feas(nz, nl, nv), ls(a...),
        z(nz), l(nl), v(nv), y(nv),
        alg(&x1, &x2, &x3, &x4, &r1, &r2, &ls, &feas) {}

  void SetOptions(const fbstab_options_t* o) {
#ifdef FBO_USE_REFERENCE_ALGORITHM
    fbstab::AlgorithmParameters p = ToRef(*o);
    alg.UpdateParameters(&p);
#else
    alg.UpdateParameters(o);
#endif
  }

  // Display capture for the *_solve_display entry points: the reference build
  // collects the text the reference prints, the restated build the records.
  std::string* text_sink = nullptr;
  std::vector<double>* trace_sink = nullptr;

  template <class DataT>
  fbstab_solver_out_t Solve(const DataT& data, double* zp, double* lp,
                            double* vp, double* yp) {
    std::copy(zp, zp + z.size(), z.begin());
    std::copy(lp, lp + l.size(), l.begin());
    std::copy(vp, vp + v.size(), v.begin());
    const auto t0 = std::chrono::high_resolution_clock::now();
#ifdef FBO_USE_REFERENCE_ALGORITHM
    CaptureOutput os(text_sink);
    fbstab_solver_out_t out = FromRef(alg.Solve(data, &z, &l, &v, &y, os));
#else
    alg.SetTraceSink(trace_sink);
    fbstab_solver_out_t out = alg.Solve(data, &z, &l, &v, &y);
#endif
    const auto t1 = std::chrono::high_resolution_clock::now();
    out.solve_time = std::chrono::duration<double>(t1 - t0).count();
    std::copy(z.begin(), z.end(), zp);
    std::copy(l.begin(), l.end(), lp);
    std::copy(v.begin(), v.end(), vp);
    std::copy(y.begin(), y.end(), yp);
    return out;
  }
};

typedef Workspace<fbo::DenseCholeskySolver> DenseWs;
typedef Workspace<fbo::RiccatiLinearSolver> MpcWs;

void FailOut(fbstab_solver_out_t* o) {
  std::memset(o, 0, sizeof(*o));
  o->eflag = -1;
  o->solve_time = -1.0;
}

}  // namespace

extern "C" {

const char* fbo_last_error() { return g_last_error.c_str(); }

int fbo_uses_reference_algorithm() {
#ifdef FBO_USE_REFERENCE_ALGORITHM
  return 1;
#else
  return 0;
#endif
}

// Batched dense solve.  Each array holds `batch` problems, problem k at
// base + k*stride (strides in doubles, in the order H,f,G,h,A,b,z,l,v,y).
// Returns the number of problems whose solve threw (message of the last one in
// fbo_last_error()); their out[k].eflag is -1.
int fbo_dense_solve_batch(int batch, int nz, int nl, int nv, const double* H,
                          const double* f, const double* G, const double* h,
                          const double* A, const double* b, double* z,
                          double* l, double* v, double* y,
                          const long long* strides,
                          const fbstab_options_t* opts,
                          fbstab_solver_out_t* out, int nthreads) {
  int failures = 0;
  std::string err;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
  {
    std::unique_ptr<DenseWs> ws;
    try {
      ws.reset(new DenseWs(nz, nl, nv, nz, nl, nv));
      ws->SetOptions(opts);
    } catch (const std::exception& e) {
#ifdef _OPENMP
#pragma omp critical
#endif
      err = e.what();
    }
#ifdef _OPENMP
#pragma omp for schedule(dynamic)
#endif
    for (int k = 0; k < batch; k++) {
      if (!ws) {
        FailOut(&out[k]);
#ifdef _OPENMP
#pragma omp atomic
#endif
        failures++;
        continue;
      }
      try {
        fbo::DenseData data(H + k * strides[0], f + k * strides[1],
                            G + k * strides[2], h + k * strides[3],
                            A + k * strides[4], b + k * strides[5], nz, nl, nv);
        out[k] = ws->Solve(data, z + k * strides[6], l + k * strides[7],
                           v + k * strides[8], y + k * strides[9]);
      } catch (const std::exception& e) {
        FailOut(&out[k]);
#ifdef _OPENMP
#pragma omp critical
#endif
        {
          err = e.what();
          failures++;
        }
      }
    }
  }
  g_last_error = err;
  return failures;
}

// Batched MPC solve; strides in the order Q,R,S,q,r,A,B,c,E,L,d,x0,z,l,v,y.
int fbo_mpc_solve_batch(int batch, int N, int nx, int nu, int nc,
                        const double* Q, const double* R, const double* S,
                        const double* q, const double* r, const double* A,
                        const double* B, const double* c, const double* E,
                        const double* L, const double* d, const double* x0,
                        double* z, double* l, double* v, double* y,
                        const long long* strides, const fbstab_options_t* opts,
                        fbstab_solver_out_t* out, int nthreads) {
  int failures = 0;
  std::string err;
  const int nz = (N + 1) * (nx + nu), nl = (N + 1) * nx, nv = (N + 1) * nc;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
  {
    std::unique_ptr<MpcWs> ws;
    try {
      ws.reset(new MpcWs(nz, nl, nv, N, nx, nu, nc));
      ws->SetOptions(opts);
    } catch (const std::exception& e) {
#ifdef _OPENMP
#pragma omp critical
#endif
      err = e.what();
    }
#ifdef _OPENMP
#pragma omp for schedule(dynamic)
#endif
    for (int k = 0; k < batch; k++) {
      if (!ws) {
        FailOut(&out[k]);
#ifdef _OPENMP
#pragma omp atomic
#endif
        failures++;
        continue;
      }
      try {
        fbo::MpcData data(Q + k * strides[0], R + k * strides[1],
                          S + k * strides[2], q + k * strides[3],
                          r + k * strides[4], A + k * strides[5],
                          B + k * strides[6], c + k * strides[7],
                          E + k * strides[8], L + k * strides[9],
                          d + k * strides[10], x0 + k * strides[11], N, nx, nu,
                          nc);
        out[k] = ws->Solve(data, z + k * strides[12], l + k * strides[13],
                           v + k * strides[14], y + k * strides[15]);
      } catch (const std::exception& e) {
        FailOut(&out[k]);
#ifdef _OPENMP
#pragma omp critical
#endif
        {
          err = e.what();
          failures++;
        }
      }
    }
  }
  g_last_error = err;
  return failures;
}


}  // extern "C"

// One QP with its display captured.  The reference-algorithm build returns
// the text the reference's own Print* functions produce at opts->display_level
// (fbstab_algorithm-impl.h:411-541) in `text` (NUL-terminated, truncated to
// text_cap) and no records; the restated build returns the records
// (fbstab_trace_record_t, 8 doubles each, at most trace_cap of them) and an
// empty text.  *trace_count is the number of records produced.
template <class Ws, class DataT>
static int solve_display(Ws* ws, const DataT& data, double* z, double* l, double* v, double* y,
                         const fbstab_options_t* opts, fbstab_solver_out_t* out, char* text,
                         int text_cap, double* trace, int trace_cap, int* trace_count) {
  std::string txt;
  std::vector<double> rec;
  try {
    ws->SetOptions(opts);
    ws->text_sink = &txt;
    ws->trace_sink = &rec;
    *out = ws->Solve(data, z, l, v, y);
  } catch (const std::exception& e) {
    FailOut(out);
    g_last_error = e.what();
    return 1;
  }
  if (text && text_cap > 0) {
    const size_t n = std::min(txt.size(), (size_t)text_cap - 1);
    std::memcpy(text, txt.data(), n);
    text[n] = 0;
  }
  const int nrec = (int)(rec.size() / 8);
  if (trace_count) *trace_count = nrec;
  if (trace) std::copy(rec.begin(), rec.begin() + 8 * (size_t)std::min(nrec, trace_cap), trace);
  return 0;
}

extern "C" {

int fbo_dense_solve_display(int nz, int nl, int nv, const double* H, const double* f,
                            const double* G, const double* h, const double* A, const double* b,
                            double* z, double* l, double* v, double* y,
                            const fbstab_options_t* opts, fbstab_solver_out_t* out, char* text,
                            int text_cap, double* trace, int trace_cap, int* trace_count) {
  try {
    DenseWs ws(nz, nl, nv, nz, nl, nv);
    fbo::DenseData data(H, f, G, h, A, b, nz, nl, nv);
    return solve_display(&ws, data, z, l, v, y, opts, out, text, text_cap, trace, trace_cap,
                         trace_count);
  } catch (const std::exception& e) {
    FailOut(out);
    g_last_error = e.what();
    return 1;
  }
}

int fbo_mpc_solve_display(int N, int nx, int nu, int nc, const double* Q, const double* R,
                          const double* S, const double* q, const double* r, const double* A,
                          const double* B, const double* c, const double* E, const double* L,
                          const double* d, const double* x0, double* z, double* l, double* v,
                          double* y, const fbstab_options_t* opts, fbstab_solver_out_t* out,
                          char* text, int text_cap, double* trace, int trace_cap,
                          int* trace_count) {
  try {
    const int nz = (N + 1) * (nx + nu), nl = (N + 1) * nx, nv = (N + 1) * nc;
    MpcWs ws(nz, nl, nv, N, nx, nu, nc);
    fbo::MpcData data(Q, R, S, q, r, A, B, c, E, L, d, x0, N, nx, nu, nc);
    return solve_display(&ws, data, z, l, v, y, opts, out, text, text_cap, trace, trace_cap,
                         trace_count);
  } catch (const std::exception& e) {
    FailOut(out);
    g_last_error = e.what();
    return 1;
  }
}

}  // extern "C"

// ---- component-level entry points (golden-vector tests) --------------------

// which: 0 gemvH, 1 gemvA, 2 gemvG, 3 gemvAT, 4 gemvGT, 5 axpyf, 6 axpyh,
// 7 axpyb (for axpy* x is ignored and b is ignored).
static int data_op(const fbo::Data& data, int which, const double* x, int nxin,
                   double a, double b, double* y, int nyio) {
  try {
    fbo::Vec xv(x, x + nxin), yv(y, y + nyio);
    switch (which) {
      case 0: data.gemvH(xv, a, b, &yv); break;
      case 1: data.gemvA(xv, a, b, &yv); break;
      case 2: data.gemvG(xv, a, b, &yv); break;
      case 3: data.gemvAT(xv, a, b, &yv); break;
      case 4: data.gemvGT(xv, a, b, &yv); break;
      case 5: data.axpyf(a, &yv); break;
      case 6: data.axpyh(a, &yv); break;
      case 7: data.axpyb(a, &yv); break;
      default: throw std::runtime_error("bad op");
    }
    std::copy(yv.begin(), yv.end(), y);
    return 0;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return 1;
  }
}

extern "C" {

int fbo_mpc_data_op(int N, int nx, int nu, int nc, const double* Q,
                    const double* R, const double* S, const double* q,
                    const double* r, const double* A, const double* B,
                    const double* c, const double* E, const double* L,
                    const double* d, const double* x0, int which,
                    const double* x, int nxin, double a, double b, double* y,
                    int nyio) {
  try {
    fbo::MpcData data(Q, R, S, q, r, A, B, c, E, L, d, x0, N, nx, nu, nc);
    return data_op(data, which, x, nxin, a, b, y, nyio);
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return 1;
  }
}

int fbo_dense_data_op(int nz, int nl, int nv, const double* H, const double* f,
                      const double* G, const double* h, const double* A,
                      const double* b_, int which, const double* x, int nxin,
                      double a, double b, double* y, int nyio) {
  try {
    fbo::DenseData data(H, f, G, h, A, b_, nz, nl, nv);
    return data_op(data, which, x, nxin, a, b, y, nyio);
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return 1;
  }
}

}  // extern "C"

// Generic component probe shared by both problem classes.
// Inputs: x=(z,l,v), xbar=(zb,lb,vb) (y's are initialised as b - A z), sigma,
// alpha, and a residual r=(rz,rl,rv) for the linear solve.
// Outputs (any may be null):
//   x_y, xbar_y        constraint margins
//   inner[nz+nl+nv]    InnerResidual(x,xbar,sigma)
//   natural[nz+nl+nv]  NaturalResidual(x)
//   pnr[nz+nl+nv]      PenalizedNaturalResidual(x)
//   dx[nz+nl+nv+nv]    Newton step (dz,dl,dv,dy) = LinearSolver(x,xbar,sigma).Solve(r)
//   gamma[nv], mus[nv] PFB gradient vectors of the linear solver
//   feas               FeasibilityStatus of x at tolerance feas_tol
// Returns 0 ok, 1 exception, 2 factorisation failure.
template <class DataT, class LsT>
static int probe(const DataT& data, LsT* ls, int nz, int nl, int nv,
                 const double* z, const double* l, const double* v,
                 const double* zb, const double* lb, const double* vb,
                 double sigma, double alpha, const double* r, double* x_y,
                 double* xbar_y, double* inner, double* natural, double* pnr,
                 double* dx, double* gamma, double* mus, int* feas,
                 double feas_tol) {
  fbo::FullVariable x(nz, nl, nv), xb(nz, nl, nv), d(nz, nl, nv);
  x.LinkData(&data);
  xb.LinkData(&data);
  d.LinkData(&data);
  std::copy(z, z + nz, x.z().begin());
  std::copy(l, l + nl, x.l().begin());
  std::copy(v, v + nv, x.v().begin());
  std::copy(zb, zb + nz, xb.z().begin());
  std::copy(lb, lb + nl, xb.l().begin());
  std::copy(vb, vb + nv, xb.v().begin());
  x.InitializeConstraintMargin();
  xb.InitializeConstraintMargin();
  if (x_y) std::copy(x.y().begin(), x.y().end(), x_y);
  if (xbar_y) std::copy(xb.y().begin(), xb.y().end(), xbar_y);
  fbo::FullResidual res(nz, nl, nv);
  res.LinkData(&data);
  res.SetAlpha(alpha);
  auto dump = [&](double* o) {
    std::copy(res.z().begin(), res.z().end(), o);
    std::copy(res.l().begin(), res.l().end(), o + nz);
    std::copy(res.v().begin(), res.v().end(), o + nz + nl);
  };
  if (inner) {
    res.InnerResidual(x, xb, sigma);
    dump(inner);
  }
  if (natural) {
    res.NaturalResidual(x);
    dump(natural);
  }
  if (pnr) {
    res.PenalizedNaturalResidual(x);
    dump(pnr);
  }
  if (dx) {
    ls->LinkData(&data);
    ls->SetAlpha(alpha);
    if (!ls->Initialize(x, xb, sigma)) return 2;
    std::copy(r, r + nz, res.z().begin());
    std::copy(r + nz, r + nz + nl, res.l().begin());
    std::copy(r + nz + nl, r + nz + nl + nv, res.v().begin());
    if (!ls->Solve(res, &d)) return 2;
    std::copy(d.z().begin(), d.z().end(), dx);
    std::copy(d.l().begin(), d.l().end(), dx + nz);
    std::copy(d.v().begin(), d.v().end(), dx + nz + nl);
    std::copy(d.y().begin(), d.y().end(), dx + nz + nl + nv);
    if (gamma) std::copy(ls->gamma_.begin(), ls->gamma_.end(), gamma);
    if (mus) std::copy(ls->mus_.begin(), ls->mus_.end(), mus);
  }
  if (feas) {
    fbo::FullFeasibility fc(nz, nl, nv);
    fc.LinkData(&data);
    *feas = static_cast<int>(fc.CheckFeasibility(x, feas_tol));
  }
  return 0;
}

extern "C" {

int fbo_mpc_probe(int N, int nx, int nu, int nc, const double* Q,
                  const double* R, const double* S, const double* q,
                  const double* r_, const double* A, const double* B,
                  const double* c, const double* E, const double* L,
                  const double* d, const double* x0, const double* z,
                  const double* l, const double* v, const double* zb,
                  const double* lb, const double* vb, double sigma,
                  double alpha, const double* r, double* x_y, double* xbar_y,
                  double* inner, double* natural, double* pnr, double* dx,
                  double* gamma, double* mus, int* feas, double feas_tol) {
  try {
    fbo::MpcData data(Q, R, S, q, r_, A, B, c, E, L, d, x0, N, nx, nu, nc);
    fbo::RiccatiLinearSolver ls(N, nx, nu, nc);
    return probe(data, &ls, data.nz(), data.nl(), data.nv(), z, l, v, zb, lb,
                 vb, sigma, alpha, r, x_y, xbar_y, inner, natural, pnr, dx,
                 gamma, mus, feas, feas_tol);
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return 1;
  }
}

int fbo_dense_probe(int nz, int nl, int nv, const double* H, const double* f,
                    const double* G, const double* h, const double* A,
                    const double* b, const double* z, const double* l,
                    const double* v, const double* zb, const double* lb,
                    const double* vb, double sigma, double alpha,
                    const double* r, double* x_y, double* xbar_y,
                    double* inner, double* natural, double* pnr, double* dx,
                    double* gamma, double* mus, int* feas, double feas_tol) {
  try {
    fbo::DenseData data(H, f, G, h, A, b, nz, nl, nv);
    fbo::DenseCholeskySolver ls(nz, nl, nv);
    return probe(data, &ls, nz, nl, nv, z, l, v, zb, lb, vb, sigma, alpha, r,
                 x_y, xbar_y, inner, natural, pnr, dx, gamma, mus, feas,
                 feas_tol);
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return 1;
  }
}

int fbo_num_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
