// MPC problem policy for the batched FBstab kernel: one workgroup owns one
// stage-structured QP
//   min sum_i 1/2 [x;u]'[Q S';S R][x;u] + q'x + r'u
//   s.t. x(i+1) = A x + B u + c,  x(0) = x0,  E x + L u + d <= 0
// in the reference's MatrixSequence layout (fbstab/fbstab_mpc.h:67-81,
// tools/matrix_sequence.h:81-83).
//
// What it implements, and the reference code each part answers to:
//   * implicit block products with H, G, A (mpc_data.cc:17-238) and the
//     f/h/b vectors (mpc_data.cc:240-289), evaluated stage by stage from an
//     LDS tile of that stage's matrices (coalesced HBM/L2 loads);
//   * the Newton step of RiccatiLinearSolver (riccati_linear_solver.cc:77-344):
//     barrier-augmented stage Hessians, the matrix recursion and the
//     forward/backward vector recursions, fused into ONE forward sweep over
//     the stages (factor + forward substitution) and ONE backward sweep;
//   * the infeasibility certificates of FullFeasibility
//     (full_feasibility.cc:25-88).
//
// Differences from the reference that change rounding but not mathematics:
//   * L(i) is inverted explicitly once per stage (each column by an independent forward
//     substitution) and applied as h = inv(L)'(inv(L) theta); M and SG are NOT (round 5): AM, SM, P and
//     the vectors tx, tu, u, x come from substitutions with the factors themselves, as in the reference
//     (riccati_linear_solver.cc:149-175, :241-249, :299-325) - products with explicitly inverted M, SG
//     are forward stable only, and on stages wider than the inputs can reach (nx > N nu) they left
//     4.6e-6 in the Newton system where the reference leaves 1e-7;
//   * the forward vector recursion runs inside the factor sweep, and the
//     backward sweep reuses tx = inv(M) h and tu = inv(SG)(SM tx + ru) from it
//     (the reference recomputes them, riccati_linear_solver.cc:299-312);
//   * dv = rv/mus + Gamma .* (A dz)  ==  (rv + gamma .* A dz) ./ mus.
#pragma once

#include <type_traits>

#include "fb_common.h"

namespace fbk {

// Pointers to one QP's problem data (already offset to that QP).
struct MpcData {
  const double *Q, *R, *S, *q, *r, *A, *B, *c, *E, *L, *d, *x0;
};

// Sizes and derived offsets, identical on host (workspace sizing) and device.
struct MpcLayout {
  int N, nx, nu, nc, ns, nz, nl, nv;
  // per-stage factor record (doubles) kept in global scratch for the backward sweep
  int f_minv, f_sm, f_am, f_p, f_sginv, f_linv, f_tx, f_tu, f_th, f_stride;
  // iterate vectors in global scratch (offsets in doubles)
  long v_z, v_l, v_v, v_y, v_zb, v_lb, v_vb, v_yb, v_dz, v_dl, v_dv, v_adz, v_rz, v_rl,
      v_wz, v_wl, v_gam, v_rvm, v_fac, ws_doubles;
  // LDS carve (offsets in doubles)
  int t_q, t_r, t_s, t_a, t_b, t_e, t_l;                  // data tile
  int s_z, s_l, s_ln, s_v;                                // vector slices of the stage
  int w_sb, w_rb, w_linv, w_m, w_minv, w_am, w_sm, w_sg, w_sginv, w_pp, w_p, w_ln;
  int w_gam, w_rvm, w_r1, w_r2, w_th, w_thp, w_h, w_tx, w_tu, w_t1, w_t2, w_lp, w_out;
  int w_red, lds_doubles;
  // wglobal != 0: the carve above does not fit the LDS and lives at v_carve of the
  // workgroup's global scratch instead (MpcProblem<C, true>); launch_lds_doubles is what
  // a launch asks for (the whole carve, or the reduction scratch alone)
  int wglobal, launch_lds_doubles;
  long v_carve;

  __host__ __device__ void init(int N_, int nx_, int nu_, int nc_, int nthreads) {
    N = N_; nx = nx_; nu = nu_; nc = nc_;
    ns = nx + nu;
    nz = (N + 1) * ns;
    nl = (N + 1) * nx;
    nv = (N + 1) * nc;
    int o = 0;
    f_minv = o; o += nx * nx;
    f_sm = o; o += nu * nx;
    f_am = o; o += nx * nx;
    f_p = o; o += nx * nu;
    f_sginv = o; o += nu * nu;
    f_linv = o; o += nx * nx;
    f_tx = o; o += nx;
    f_tu = o; o += nu;
    f_th = o; o += nx;
    f_stride = (o + 1) & ~1;
    long g = 0;
    v_z = g; g += nz;  v_l = g; g += nl;  v_v = g; g += nv;  v_y = g; g += nv;
    v_zb = g; g += nz; v_lb = g; g += nl; v_vb = g; g += nv; v_yb = g; g += nv;
    v_dz = g; g += nz; v_dl = g; g += nl; v_dv = g; g += nv; v_adz = g; g += nv;
    v_rz = g; g += nz; v_rl = g; g += nl; v_wz = g; g += nz; v_wl = g; g += nl;
    v_gam = g; g += nv; v_rvm = g; g += nv;
    g = (g + 1) & ~1L;
    v_fac = g; g += (long)f_stride * (N + 1);
    ws_doubles = (g + 15) & ~15L;  // 128-byte multiple per workgroup slot
    int s = 0;
    t_q = s; s += nx * nx;  t_r = s; s += nu * nu;  t_s = s; s += nu * nx;
    t_a = s; s += nx * nx;  t_b = s; s += nx * nu;  t_e = s; s += nc * nx;
    t_l = s; s += nc * nu;
    s_z = s; s += ns;  s_l = s; s += nx;  s_ln = s; s += nx;  s_v = s; s += nc;
    w_sb = s; s += nu * nx;  w_rb = s; s += nu * nu;
    w_linv = s; s += nx * nx;  w_m = s; s += nx * nx;  w_minv = s; s += nx * nx;
    w_am = s; s += nx * nx;  w_sm = s; s += nu * nx;  w_sg = s; s += nu * nu;
    w_sginv = s; s += nu * nu;  w_pp = s; s += nx * nu;  w_p = s; s += nx * nu;
    w_ln = s; s += nx * nx;
    w_gam = s; s += nc;  w_rvm = s; s += nc;  w_r1 = s; s += ns;  w_r2 = s; s += nx;
    w_th = s; s += nx;  w_thp = s; s += nx;  w_h = s; s += nx;  w_tx = s; s += nx;
    w_tu = s; s += nu;  w_t1 = s; s += ns;  w_t2 = s; s += ns;  w_lp = s; s += nx;
    w_out = s; s += ns + nx + nc;
    w_red = s; s += kMaxReduce * ((nthreads + 63) / 64);
    lds_doubles = (s + 1) & ~1;
    wglobal = (long)lds_doubles * 8 > 160 * 1024 ? 1 : 0;
    launch_lds_doubles = lds_doubles;
    v_carve = 0;
    if (wglobal) {
      v_carve = ws_doubles;
      ws_doubles += (lds_doubles + 15) & ~15L;
      launch_lds_doubles = (kMaxReduce * ((nthreads + 63) / 64) + ns + nx + nc + 17) & ~1;
    }
  }
};

// WGLOBAL: the stage tile and the work matrices of the Newton step - everything the
// layout carves out of the LDS except the reduction scratch - live in the workgroup's
// GLOBAL scratch instead: the instance for stages whose matrices do not fit the 160 KB
// of LDS (the reference allocates on the heap for any size, fbstab_mpc.cc:61-89).  One
// wavefront per QP either way: its global-memory operations stay in program order, so
// the same synchronisation serves both.
template <class C, bool WGLOBAL = false>
struct MpcProblem {
  typedef typename std::conditional<WGLOBAL, double*, lds_ptr>::type mptr;
  static constexpr bool kFusedTrial = false;  // see fb_algorithm.h
  static constexpr bool kOwnVectorOps = false;  // the Solver loops over the flat vectors below
  static constexpr bool kCanRefine = true;      // linear_residual2() / refine_step() below
  MpcLayout lay;
  MpcData D;
  double *uz, *ul, *uv, *uy;  // caller's (z,l,v,y) for this QP
  mptr lds;  // base of the stage tile / work-matrix carve (LDS, or global scratch: WGLOBAL)
  double* ws;  // this workgroup's global scratch
  int nz, nl, nv;
  double *z, *l, *v, *y, *zb, *lb, *vb, *yb, *dz, *dl, *dv, *adz, *rz, *rl, *wz, *wl;
  double *gam, *rvm, *fac;

  FB_DEV void bind(const MpcLayout& L_, const MpcData& D_, double* uz_, double* ul_,
                   double* uv_, double* uy_, mptr lds_, double* ws_) {
    lay = L_; D = D_; uz = uz_; ul = ul_; uv = uv_; uy = uy_; lds = lds_; ws = ws_;
    nz = lay.nz; nl = lay.nl; nv = lay.nv;
    z = ws + lay.v_z; l = ws + lay.v_l; v = ws + lay.v_v; y = ws + lay.v_y;
    zb = ws + lay.v_zb; lb = ws + lay.v_lb; vb = ws + lay.v_vb; yb = ws + lay.v_yb;
    dz = ws + lay.v_dz; dl = ws + lay.v_dl; dv = ws + lay.v_dv; adz = ws + lay.v_adz;
    rz = ws + lay.v_rz; rl = ws + lay.v_rl; wz = ws + lay.v_wz; wl = ws + lay.v_wl;
    gam = ws + lay.v_gam; rvm = ws + lay.v_rvm; fac = ws + lay.v_fac;
  }

  // ---- small helpers -------------------------------------------------------
  template <class Src>
  FB_DEV void copy_in(const C& c, mptr dst, Src src, int n) const {
    for (int i = c.tid; i < n; i += C::nt) dst[i] = src[i];
  }

  // Load stage i's matrices into the LDS tile (coalesced, stage-contiguous).
  FB_DEV void load_tile(const C& c, int i) const {
    const int nx = lay.nx, nu = lay.nu, nc = lay.nc;
    copy_in(c, lds + lay.t_q, D.Q + (long)i * nx * nx, nx * nx);
    copy_in(c, lds + lay.t_r, D.R + (long)i * nu * nu, nu * nu);
    copy_in(c, lds + lay.t_s, D.S + (long)i * nu * nx, nu * nx);
    copy_in(c, lds + lay.t_e, D.E + (long)i * nc * nx, nc * nx);
    copy_in(c, lds + lay.t_l, D.L + (long)i * nc * nu, nc * nu);
    if (i < lay.N) {
      copy_in(c, lds + lay.t_a, D.A + (long)i * nx * nx, nx * nx);
      copy_in(c, lds + lay.t_b, D.B + (long)i * nx * nu, nx * nu);
    }
  }

  // Stage slices of a (z,l,v)-shaped vector triple into LDS.
  FB_DEV void load_slices(const C& c, int i, const double* zz, const double* ll,
                          const double* vv) const {
    copy_in(c, lds + lay.s_z, zz + (long)i * lay.ns, lay.ns);
    if (ll) {
      copy_in(c, lds + lay.s_l, ll + (long)i * lay.nx, lay.nx);
      if (i < lay.N) copy_in(c, lds + lay.s_ln, ll + (long)(i + 1) * lay.nx, lay.nx);
    }
    if (vv) copy_in(c, lds + lay.s_v, vv + (long)i * lay.nc, lay.nc);
  }

  // (H zz)_i[r] from the tile (mpc_data.cc:28-63); r in [0, ns).
  FB_DEV double tile_Hz(int r) const {
    const int nx = lay.nx, nu = lay.nu;
    mptr zx = lds + lay.s_z;
    mptr zu = zx + nx;
    double s = 0.0;
    if (r < nx) {
      mptr Q = lds + lay.t_q;
      mptr S = lds + lay.t_s;
      for (int k = 0; k < nx; k++) s += Q[r + k * nx] * zx[k];
      for (int k = 0; k < nu; k++) s += S[k + r * nu] * zu[k];
    } else {
      const int ru = r - nx;
      mptr S = lds + lay.t_s;
      mptr R = lds + lay.t_r;
      for (int k = 0; k < nx; k++) s += S[ru + k * nu] * zx[k];
      for (int k = 0; k < nu; k++) s += R[ru + k * nu] * zu[k];
    }
    return s;
  }
  // (G' ll)_i[r] (mpc_data.cc:171-198).
  FB_DEV double tile_GTl(int i, int r) const {
    const int nx = lay.nx;
    mptr li = lds + lay.s_l;
    mptr ln = lds + lay.s_ln;
    double s = 0.0;
    if (r < nx) {
      s = -li[r];
      if (i < lay.N) {
        mptr A = lds + lay.t_a;
        for (int k = 0; k < nx; k++) s += A[k + r * nx] * ln[k];
      }
    } else if (i < lay.N) {
      mptr B = lds + lay.t_b;
      const int ru = r - nx;
      for (int k = 0; k < nx; k++) s += B[k + ru * nx] * ln[k];
    }
    return s;
  }
  // (A' vv)_i[r] (mpc_data.cc:217-237).
  FB_DEV double tile_ATv(int r) const {
    const int nx = lay.nx, nc = lay.nc;
    mptr vi = lds + lay.s_v;
    double s = 0.0;
    if (r < nx) {
      mptr E = lds + lay.t_e + r * nc;
      for (int k = 0; k < nc; k++) s += E[k] * vi[k];
    } else {
      mptr L = lds + lay.t_l + (r - nx) * nc;
      for (int k = 0; k < nc; k++) s += L[k] * vi[k];
    }
    return s;
  }
  // (A zz)_i[k] = E x + L u (mpc_data.cc:84-104).
  FB_DEV double tile_Az(int k) const {
    const int nx = lay.nx, nu = lay.nu, nc = lay.nc;
    mptr zx = lds + lay.s_z;
    mptr E = lds + lay.t_e;
    mptr L = lds + lay.t_l;
    double s = 0.0;
    for (int j = 0; j < nx; j++) s += E[k + j * nc] * zx[j];
    for (int j = 0; j < nu; j++) s += L[k + j * nc] * zx[nx + j];
    return s;
  }
  // A_i x_i + B_i u_i, row r (the stage-(i+1) block of G z before "- x(i+1)",
  // mpc_data.cc:127-152).
  FB_DEV double tile_ABz(int r) const {
    const int nx = lay.nx, nu = lay.nu;
    mptr zx = lds + lay.s_z;
    mptr A = lds + lay.t_a;
    mptr B = lds + lay.t_b;
    double s = 0.0;
    for (int j = 0; j < nx; j++) s += A[r + j * nx] * zx[j];
    for (int j = 0; j < nu; j++) s += B[r + j * nx] * zx[nx + j];
    return s;
  }

  // ---- policy interface ------------------------------------------------------
  // ||(f,h,b)||_2 (mpc_data.h:88-97).
  FB_DEV double forcing_norm(const C& c) const {
    double s[1] = {0.0};
    for (int i = c.tid; i < (lay.N + 1) * lay.nx; i += C::nt) s[0] += D.q[i] * D.q[i];
    for (int i = c.tid; i < (lay.N + 1) * lay.nu; i += C::nt) s[0] += D.r[i] * D.r[i];
    for (int i = c.tid; i < (lay.N + 1) * lay.nc; i += C::nt) s[0] += D.d[i] * D.d[i];
    for (int i = c.tid; i < lay.nx; i += C::nt) s[0] += D.x0[i] * D.x0[i];
    for (int i = c.tid; i < lay.N * lay.nx; i += C::nt) s[0] += D.c[i] * D.c[i];
    c.sum(s);
    return sqrt(s[0]);
  }

  FB_DEV int num_primal_dual() const { return nz + nl + nv; }
  FB_DEV double bvec(int i) const { return -D.d[i]; }  // b = -d (mpc_data.cc:276-289)

  // x <- caller's guess; y = b - A z (impl:334-347, full_variable.cc:47-53).
  FB_DEV void load_guess(const C& c) const {
    for (int i = c.tid; i < nz; i += C::nt) z[i] = uz[i];
    for (int i = c.tid; i < nl; i += C::nt) l[i] = ul[i];
    for (int i = c.tid; i < nv; i += C::nt) v[i] = uv[i];
    c.sync();
    for (int i = 0; i <= lay.N; i++) {
      load_tile(c, i);
      load_slices(c, i, z, nullptr, nullptr);
      c.sync();
      for (int k = c.tid; k < lay.nc; k += C::nt)
        y[(long)i * lay.nc + k] = bvec(i * lay.nc + k) - tile_Az(k);
      c.sync();
    }
  }

  // Natural residual blocks at x: rz = Hz + f + G'l + A'v, rl = h - Gz
  // (full_residual.cc:79-91).
  FB_DEV void residual(const C& c) const {
    const int nx = lay.nx, ns = lay.ns;
    for (int i = 0; i <= lay.N; i++) {
      load_tile(c, i);
      load_slices(c, i, z, l, v);
      c.sync();
      for (int r = c.tid; r < ns + nx; r += C::nt) {
        if (r < ns) {
          const double f = r < nx ? D.q[(long)i * nx + r] : D.r[(long)i * lay.nu + (r - nx)];
          rz[(long)i * ns + r] = f + tile_Hz(r) + tile_GTl(i, r) + tile_ATv(r);
        } else {
          const int rr = r - ns;
          // block 0: h0 - (Gz)0 = -x0 + x(0); block i+1: -c(i) - (A x + B u - x(i+1))
          if (i == 0) rl[rr] = -D.x0[rr] + (lds + lay.s_z)[rr];
          if (i < lay.N)
            rl[(long)(i + 1) * nx + rr] =
                -D.c[(long)i * nx + rr] - (tile_ABz(rr) - z[(long)(i + 1) * ns + rr]);
        }
      }
      c.sync();
    }
  }

  // Infeasibility certificates for dx = (dz,dl,dv) (full_feasibility.cc:25-88).
  FB_DEV int feasibility(const C& c, double tol) const {
    const int nx = lay.nx, ns = lay.ns, nc = lay.nc;
    double mx[5] = {-1e300, 0.0, 0.0, 0.0, 0.0};  // max(A dz), |G dz|, |H dz|, |dz|, |A'dv+G'dl|
    double sm[2] = {0.0, 0.0};                    // f'dz, b'dv + h'dl
    double ul[1] = {0.0};                         // max(|dv|,|dl|)
    for (int i = 0; i <= lay.N; i++) {
      load_tile(c, i);
      load_slices(c, i, dz, dl, dv);
      c.sync();
      for (int r = c.tid; r < ns + nx + nc; r += C::nt) {
        if (r < ns) {
          const double dzr = (lds + lay.s_z)[r];
          const double f = r < nx ? D.q[(long)i * nx + r] : D.r[(long)i * lay.nu + (r - nx)];
          mx[2] = fmax(mx[2], fabs(tile_Hz(r)));
          mx[3] = fmax(mx[3], fabs(dzr));
          mx[4] = fmax(mx[4], fabs(tile_ATv(r) + tile_GTl(i, r)));
          sm[0] += f * dzr;
        } else if (r < ns + nx) {
          const int rr = r - ns;
          const double dli = (lds + lay.s_l)[rr];
          ul[0] = fmax(ul[0], fabs(dli));
          if (i == 0) {
            mx[1] = fmax(mx[1], fabs((lds + lay.s_z)[rr]));  // (G dz)_0 = -dx(0)
            sm[1] += -D.x0[rr] * dli;
          } else {
            sm[1] += -D.c[(long)(i - 1) * nx + rr] * dli;
          }
          if (i < lay.N)
            mx[1] = fmax(mx[1], fabs(tile_ABz(rr) - dz[(long)(i + 1) * ns + rr]));
        } else {
          const int k = r - ns - nx;
          const double dvk = (lds + lay.s_v)[k];
          mx[0] = fmax(mx[0], tile_Az(k));
          ul[0] = fmax(ul[0], fabs(dvk));
          sm[1] += bvec(i * nc + k) * dvk;
        }
      }
      c.sync();
    }
    c.max(mx);
    c.sum(sm);
    c.max(ul);
    const double d1 = mx[0], d2 = mx[1], d3 = mx[2], w = mx[3], p1 = mx[4];
    const double d4 = sm[0], p2 = sm[1], u = ul[0];
    bool dual_feasible = true, primal_feasible = true;
    if ((d1 <= w * tol) && (d2 <= tol * w) && (d3 <= tol * w) && (d4 < 0) && (w > 1e-14))
      dual_feasible = false;
    if ((p1 <= tol * u) && (p2 < 0)) primal_feasible = false;
    if (primal_feasible && dual_feasible) return kFeasible;
    if (primal_feasible && !dual_feasible) return kDualInfeasible;
    if (!primal_feasible && dual_feasible) return kPrimalInfeasible;
    return kBothInfeasible;
  }

  // ---- triangular solves with a lower factor Lo (n x n, column-major) ----------------------------
  // X <- X inv(Lo)' for the m x n matrix X (Eigen: Lo.transpose().solveInPlace<OnTheRight>(X)): one row
  // per thread, x[c] = (x[c] - sum_{k<c} x[k] Lo[c][k]) / Lo[c][c].  Caller syncs afterwards.
  FB_DEV void solve_right_t(const C& c, mptr X, int m, mptr Lo, int n) const {
    for (int r = c.tid; r < m; r += C::nt) {
      for (int cc = 0; cc < n; cc++) {
        double s = X[r + cc * m];
        for (int k = 0; k < cc; k++) s -= X[r + k * m] * Lo[cc + k * n];
        X[r + cc * m] = s / Lo[cc + cc * n];
      }
    }
  }
  // x <- inv(Lo) x (forward substitution, column-oriented: the entries below a solved one take its
  // contribution side by side).  Synchronised on return.
  FB_DEV void solve_lower(const C& c, mptr Lo, int n, mptr x) const {
    for (int k = 0; k < n; k++) {
      const double xk = x[k] / Lo[k + k * n];
      c.sync();
      if (c.tid == 0) x[k] = xk;
      for (int r = k + 1 + c.tid; r < n; r += C::nt) x[r] -= Lo[r + k * n] * xk;
      c.sync();
    }
  }
  // x <- inv(Lo)' x (back substitution with the transpose).  Synchronised on return.
  FB_DEV void solve_lower_t(const C& c, mptr Lo, int n, mptr x) const {
    for (int k = n - 1; k >= 0; k--) {
      const double xk = x[k] / Lo[k + k * n];
      c.sync();
      if (c.tid == 0) x[k] = xk;
      for (int r = c.tid; r < k; r += C::nt) x[r] -= Lo[k + r * n] * xk;
      c.sync();
    }
  }

  // ---- dense micro-kernels on LDS matrices (n <= 64 <= NT) -------------------
  // In-place lower Cholesky, one thread per row (left-looking: column j is
  // finished from the already final columns < j).  Workgroup-uniform result.
  FB_DEV bool chol(const C& c, mptr A, int n) const {
    for (int j = 0; j < n; j++) {
      for (int r = j + c.tid; r < n; r += C::nt) {
        double t = A[r + j * n];
        for (int k = 0; k < j; k++) t -= A[r + k * n] * A[j + k * n];
        A[r + j * n] = t;
      }
      c.sync();
      const double d = A[j + j * n];
      if (!(d > 0.0)) return false;
      const double sd = sqrt(d);
      c.sync();
      for (int r = j + c.tid; r < n; r += C::nt) A[r + j * n] = (r == j) ? sd : A[r + j * n] / sd;
      c.sync();
    }
    return true;
  }
  // X = inv(Lo) for lower-triangular Lo (column c by thread c); the strict
  // upper triangle of X is zeroed.  Caller syncs afterwards.
  FB_DEV void tri_inv(const C& c, mptr Lo, mptr X, int n) const {
    for (int cc = c.tid; cc < n; cc += C::nt) {
      for (int r = 0; r < cc; r++) X[r + cc * n] = 0.0;
      X[cc + cc * n] = 1.0 / Lo[cc + cc * n];
      for (int r = cc + 1; r < n; r++) {
        double s = 0.0;
        for (int k = cc; k < r; k++) s += Lo[r + k * n] * X[k + cc * n];
        X[r + cc * n] = -s / Lo[r + r * n];
      }
    }
  }

  // The backward recursion (:267-327) from the factors and (tx, tu, theta) of every stage in `fac`
  // (stage N still resident in LDS): the step's z block to oz, its l block to ol.
  FB_DEV void backward_sweep(const C& c, double* oz, double* ol) const {
    const int N = lay.N, nx = lay.nx, nu = lay.nu, ns = lay.ns;
    mptr Linv = lds + lay.w_linv; mptr M = lds + lay.w_m;  // (M, SG: the Cholesky factors themselves)
    mptr AM = lds + lay.w_am; mptr SM = lds + lay.w_sm; mptr SG = lds + lay.w_sg; mptr P = lds + lay.w_p;
    mptr th = lds + lay.w_th; mptr tx = lds + lay.w_tx; mptr tu = lds + lay.w_tu;
    mptr t1 = lds + lay.w_t1; mptr t2 = lds + lay.w_t2; mptr lp = lds + lay.w_lp;
    for (int i = N; i >= 0; i--) {
      if (i < N) {
        const double* F = fac + (long)i * lay.f_stride;
        copy_in(c, M, F + lay.f_minv, nx * nx);
        copy_in(c, Linv, F + lay.f_linv, nx * nx);
        copy_in(c, AM, F + lay.f_am, nx * nx);
        copy_in(c, SM, F + lay.f_sm, nu * nx);
        copy_in(c, P, F + lay.f_p, nu * nx);
        copy_in(c, SG, F + lay.f_sginv, nu * nu);
        copy_in(c, tx, F + lay.f_tx, nx);
        copy_in(c, tu, F + lay.f_tu, nu);
        copy_in(c, th, F + lay.f_th, nx);
        c.sync();
      }
      // a = tu + P' l(i+1)   (a = tu at the terminal stage)
      for (int r = c.tid; r < nu; r += C::nt) {
        double s = tu[r];
        if (i < N)
          for (int k = 0; k < nx; k++) s += P[k + r * nx] * lp[k];
        t2[r] = s;
      }
      c.sync();
      // u = inv(SG)' a: back substitution (:307-309)
      solve_lower_t(c, SG, nu, t2);
      for (int r = c.tid; r < nu; r += C::nt) t1[nx + r] = t2[r];
      c.sync();
      // b = tx + SM' u + AM' l(i+1)
      for (int r = c.tid; r < nx; r += C::nt) {
        double s = tx[r];
        for (int k = 0; k < nu; k++) s += SM[k + r * nu] * t1[nx + k];
        if (i < N)
          for (int k = 0; k < nx; k++) s += AM[k + r * nx] * lp[k];
        t2[r] = s;
      }
      c.sync();
      // x = -inv(M)' b: back substitution (:315-320)
      solve_lower_t(c, M, nx, t2);
      for (int r = c.tid; r < nx; r += C::nt) t1[r] = -t2[r];
      c.sync();
      // w = inv(L)(theta + x);  l = -inv(L)' w
      for (int r = c.tid; r < nx; r += C::nt) {
        double s = 0.0;
        for (int k = 0; k <= r; k++) s += Linv[r + k * nx] * (th[k] + t1[k]);
        t2[r] = s;
      }
      c.sync();
      for (int r = c.tid; r < ns + nx; r += C::nt) {
        if (r < ns) {
          oz[(long)i * ns + r] = t1[r];
        } else {
          const int rr = r - ns;
          double s = 0.0;
          for (int k = rr; k < nx; k++) s += Linv[k + rr * nx] * t2[k];
          lp[rr] = -s;
          ol[(long)i * nx + rr] = -s;
        }
      }
      c.sync();
    }

  }

  // dv, A dz and W = (H dz + G'dl + A'dv, -G dz) from (dz, dl) (:329-341 + the residual increment of
  // fb_algorithm.h): the third block row of the Newton system holds exactly for whatever dz is.
  FB_DEV void post_sweep(const C& c) const {
    const int N = lay.N, nx = lay.nx, nc = lay.nc, ns = lay.ns;
    for (int i = 0; i <= N; i++) {
      load_tile(c, i);
      load_slices(c, i, dz, dl, nullptr);
      c.sync();
      for (int k = c.tid; k < nc; k += C::nt) {
        const long g = (long)i * nc + k;
        const double a = tile_Az(k);
        const double d = rvm[g] + gam[g] * a;
        adz[g] = a;
        dv[g] = d;
        (lds + lay.s_v)[k] = d;
      }
      c.sync();
      for (int r = c.tid; r < ns + nx; r += C::nt) {
        if (r < ns) {
          wz[(long)i * ns + r] = tile_Hz(r) + tile_GTl(i, r) + tile_ATv(r);
        } else {
          const int rr = r - ns;
          if (i == 0) wl[rr] = (lds + lay.s_z)[rr];  // -(G dz)_0 = dx(0)
          if (i < N)
            wl[(long)(i + 1) * nx + rr] = -(tile_ABz(rr) - dz[(long)(i + 1) * ns + rr]);
        }
      }
      c.sync();
    }
  }

  // ---- one refinement of the step (same rule as the record kernels') ---------
  // Squared norm of what the linear solve left of the Newton system in its z and l block rows at the
  // step (dz, dl, dv): minus the inner residual's affine blocks at x + dx (full_residual.cc:52-66).
  FB_DEV double linear_residual2(const C& c, double sigma) const {
    double s[1] = {0.0};
    for (int i = c.tid; i < nz; i += C::nt) {
      const double r = (rz[i] + wz[i]) + sigma * ((z[i] + dz[i]) - zb[i]);
      s[0] += r * r;
    }
    for (int i = c.tid; i < nl; i += C::nt) {
      const double r = (rl[i] + wl[i]) + sigma * ((l[i] + dl[i]) - lb[i]);
      s[0] += r * r;
    }
    c.sum(s);
    return s[0];
  }
  // V ddx = r - V dx with the factors newton_step() left in `fac` (this kernel inverts its triangular
  // factors explicitly too: forward stable only, see the header), dx += ddx: the vector recursions of
  // riccati_linear_solver.cc:212-327 once more, on the residual.  dv and W follow from the refined
  // (dz, dl) as before.
  FB_DEV void refine_step(const C& c, double sigma) const {
    const int N = lay.N, nx = lay.nx, nu = lay.nu, ns = lay.ns;
    mptr Linv = lds + lay.w_linv; mptr M = lds + lay.w_m;
    mptr AM = lds + lay.w_am; mptr SM = lds + lay.w_sm; mptr SG = lds + lay.w_sg; mptr P = lds + lay.w_p;
    mptr r1 = lds + lay.w_r1;
    mptr th = lds + lay.w_th; mptr thp = lds + lay.w_thp; mptr hh = lds + lay.w_h;
    mptr tx = lds + lay.w_tx; mptr tu = lds + lay.w_tu;
    mptr t1 = lds + lay.w_t1;
    for (int r = c.tid; r < nx; r += C::nt) thp[r] = 0.0;
    c.sync();
    for (int i = 0; i <= N; i++) {
      double* F = fac + (long)i * lay.f_stride;
      copy_in(c, M, F + lay.f_minv, nx * nx);
      copy_in(c, Linv, F + lay.f_linv, nx * nx);
      copy_in(c, SM, F + lay.f_sm, nu * nx);
      copy_in(c, SG, F + lay.f_sginv, nu * nu);
      if (i < N) {
        copy_in(c, AM, F + lay.f_am, nx * nx);
        copy_in(c, P, F + lay.f_p, nu * nx);
      }
      for (int r = c.tid; r < ns + nx; r += C::nt) {
        if (r < ns) {
          const long g = (long)i * ns + r;
          r1[r] = -((rz[g] + wz[g]) + sigma * ((z[g] + dz[g]) - zb[g]));
        } else {
          const int rr = r - ns;
          const long g = (long)i * nx + rr;
          th[rr] = thp[rr] + ((rl[g] + wl[g]) + sigma * ((l[g] + dl[g]) - lb[g]));
        }
      }
      c.sync();
      for (int r = c.tid; r < nx; r += C::nt) {  // w = inv(L) theta
        double s = 0.0;
        for (int k = 0; k <= r; k++) s += Linv[r + k * nx] * th[k];
        t1[r] = s;
      }
      c.sync();
      for (int r = c.tid; r < nx; r += C::nt) {  // h = inv(L)' w - rx
        double s = -r1[r];
        for (int k = r; k < nx; k++) s += Linv[k + r * nx] * t1[k];
        hh[r] = s;
        tx[r] = s;
      }
      c.sync();
      solve_lower(c, M, nx, tx);  // tx = inv(M) h
      for (int r = c.tid; r < nu; r += C::nt) {  // t2 = SM tx + ru
        double s = r1[nx + r];
        for (int k = 0; k < nx; k++) s += SM[r + k * nu] * tx[k];
        tu[r] = s;
      }
      c.sync();
      solve_lower(c, SG, nu, tu);  // tu = inv(SG) t2
      for (int r = c.tid; r < nx; r += C::nt) {
        F[lay.f_tx + r] = tx[r];
        F[lay.f_th + r] = th[r];
        if (i < N) {  // theta(i+1) partial = P tu + AM tx
          double s = 0.0;
          for (int k = 0; k < nu; k++) s += P[r + k * nx] * tu[k];
          for (int k = 0; k < nx; k++) s += AM[r + k * nx] * tx[k];
          thp[r] = s;
        }
      }
      for (int r = c.tid; r < nu; r += C::nt) F[lay.f_tu + r] = tu[r];
      c.sync();
    }
    // the correction to (wz, wl) - the forward recursion above has read them, the post sweep rewrites them
    backward_sweep(c, wz, wl);
    for (int i = c.tid; i < nz; i += C::nt) dz[i] += wz[i];
    for (int i = c.tid; i < nl; i += C::nt) dl[i] += wl[i];
    c.sync();
    post_sweep(c);
  }

  // ---- the Newton step --------------------------------------------------------
  // Solves V(x,xbar,sigma) dx = -R(x,xbar,sigma) (abstract_components.h:276-288)
  // and produces dz, dl, dv, adz = A dz, wz = H dz + G'dl + A'dv, wl = -G dz.
  // Returns false iff a Cholesky pivot was not positive
  // (riccati_linear_solver.cc:131-136).
  FB_DEV bool newton_step(const C& c, double sigma, double alpha) const {
    const int N = lay.N, nx = lay.nx, nu = lay.nu, nc = lay.nc, ns = lay.ns;
    mptr tQ = lds + lay.t_q; mptr tR = lds + lay.t_r; mptr tS = lds + lay.t_s;
    mptr tA = lds + lay.t_a; mptr tB = lds + lay.t_b; mptr tE = lds + lay.t_e;
    mptr tL = lds + lay.t_l;
    mptr Sb = lds + lay.w_sb; mptr Rb = lds + lay.w_rb;
    mptr Linv = lds + lay.w_linv; mptr M = lds + lay.w_m;
    mptr AM = lds + lay.w_am; mptr SM = lds + lay.w_sm; mptr SG = lds + lay.w_sg;
    mptr PP = lds + lay.w_pp; mptr P = lds + lay.w_p;
    mptr Ln = lds + lay.w_ln;
    mptr Gam = lds + lay.w_gam; mptr Rvm = lds + lay.w_rvm;
    mptr r1 = lds + lay.w_r1; mptr r2 = lds + lay.w_r2;
    mptr th = lds + lay.w_th; mptr thp = lds + lay.w_thp; mptr hh = lds + lay.w_h;
    mptr tx = lds + lay.w_tx; mptr tu = lds + lay.w_tu;
    mptr t1 = lds + lay.w_t1; mptr t2 = lds + lay.w_t2;

    // Base case L(0) = sqrt(sigma) I  =>  inv(L(0)) = I / sqrt(sigma)
    // (riccati_linear_solver.cc:127).
    {
      const double is = 1.0 / sqrt(sigma);
      for (int idx = c.tid; idx < nx * nx; idx += C::nt)
        Linv[idx] = (idx % nx == idx / nx) ? is : 0.0;
      for (int r = c.tid; r < nx; r += C::nt) thp[r] = 0.0;
    }
    c.sync();

    // ============ forward sweep: factor + forward substitution ==============
    for (int i = 0; i <= N; i++) {
      double* F = fac + (long)i * lay.f_stride;
      load_tile(c, i);
      // PFB gradient of the stage's constraints (riccati_linear_solver.cc:91-99)
      for (int k = c.tid; k < nc; k += C::nt) {
        const long g = (long)i * nc + k;
        const double vk = v[g];
        const double ys = y[g] + sigma * (vk - vb[g]);
        double g0, g1;
        pfb_gradient(ys, vk, alpha, &g0, &g1);
        const double mu = g1 + sigma * g0;
        const double G_ = g0 / mu;
        const double rm = -pfb(ys, vk, alpha) / mu;  // (-rv)/mus
        Gam[k] = G_;
        Rvm[k] = rm;
        gam[g] = G_;
        rvm[g] = rm;
      }
      c.sync();
      // Barrier-augmented Hessians (lower triangles; :101-123), the matrix to
      // factor QQ = Qbar + inv(L L') (:142-145), and the eliminated right-hand
      // sides r1 = -rz_inner - A'(rv/mus), r2 = rl_inner (:222-225).
      {
        const int nq = nx * nx, nr = nu * nu, nsx = nu * nx;
        for (int idx = c.tid; idx < nq + nr + nsx + ns + nx; idx += C::nt) {
          if (idx < nq) {
            const int r = idx % nx, cc = idx / nx;
            if (r >= cc) {
              double s = tQ[idx] + (r == cc ? sigma : 0.0);
              for (int k = 0; k < nc; k++) s += Gam[k] * tE[k + r * nc] * tE[k + cc * nc];
              for (int k = r; k < nx; k++) s += Linv[k + r * nx] * Linv[k + cc * nx];
              M[idx] = s;
            }
          } else if (idx < nq + nr) {
            const int e = idx - nq;
            const int r = e % nu, cc = e / nu;
            if (r >= cc) {
              double s = tR[e] + (r == cc ? sigma : 0.0);
              for (int k = 0; k < nc; k++) s += Gam[k] * tL[k + r * nc] * tL[k + cc * nc];
              Rb[e] = s;
            }
          } else if (idx < nq + nr + nsx) {
            const int e = idx - nq - nr;
            const int r = e % nu, cc = e / nu;
            double s = tS[e];
            for (int k = 0; k < nc; k++) s += Gam[k] * tL[k + r * nc] * tE[k + cc * nc];
            Sb[e] = s;
          } else if (idx < nq + nr + nsx + ns) {
            const int r = idx - nq - nr - nsx;
            const long g = (long)i * ns + r;
            double s = -(rz[g] + sigma * (z[g] - zb[g]));
            mptr col = r < nx ? tE + r * nc : tL + (r - nx) * nc;
            for (int k = 0; k < nc; k++) s -= col[k] * Rvm[k];
            r1[r] = s;
          } else {
            const int r = idx - nq - nr - nsx - ns;
            const long g = (long)i * nx + r;
            const double r2v = rl[g] + sigma * (l[g] - lb[g]);
            r2[r] = r2v;
            th[r] = thp[r] + r2v;  // theta(i) (:231, :252-254)
          }
        }
      }
      c.sync();
      if (!chol(c, M, nx)) return false;
      // w = inv(L) theta  (first half of h = inv(L L') theta - rx, :233-236,:257-261)
      for (int r = c.tid; r < nx; r += C::nt) {
        double s = 0.0;
        for (int k = 0; k <= r; k++) s += Linv[r + k * nx] * th[k];
        t1[r] = s;
      }
      c.sync();
      // AM = A inv(M)', SM = Sbar inv(M)' (:149-161) by substitution with M, a row per thread;
      // h = inv(L)' w - rx.
      {
        const int na = (i < N) ? nx * nx : 0;
        for (int idx = c.tid; idx < na + nu * nx + nx; idx += C::nt) {
          if (idx < na) {
            AM[idx] = tA[idx];
          } else if (idx < na + nu * nx) {
            SM[idx - na] = Sb[idx - na];
          } else {
            const int r = idx - na - nu * nx;
            double s = -r1[r];
            for (int k = r; k < nx; k++) s += Linv[k + r * nx] * t1[k];
            hh[r] = s;
            tx[r] = s;
          }
        }
      }
      c.sync();
      if (i < N) solve_right_t(c, AM, nx, M, nx);
      solve_right_t(c, SM, nu, M, nx);
      // tx = inv(M) h (:241-243): forward substitution
      solve_lower(c, M, nx, tx);
      // SG = chol(Rbar - SM SM') (:163-165)
      for (int idx = c.tid; idx < nu * nu; idx += C::nt) {
        const int r = idx % nu, cc = idx / nu;
        if (r >= cc) {
          double s = Rb[idx];
          for (int k = 0; k < nx; k++) s -= SM[r + k * nu] * SM[cc + k * nu];
          SG[idx] = s;
        }
      }
      c.sync();
      if (!chol(c, SG, nu)) return false;
      // PP = AM SM' - B (:169-170); t2 = SM tx + ru (:247-248)
      {
        const int np = (i < N) ? nx * nu : 0;
        for (int idx = c.tid; idx < np + nu; idx += C::nt) {
          if (idx < np) {
            const int r = idx % nx, cc = idx / nx;
            double s = -tB[idx];
            for (int k = 0; k < nx; k++) s += AM[r + k * nx] * SM[cc + k * nu];
            PP[idx] = s;
          } else {
            const int r = idx - np;
            double s = r1[nx + r];
            for (int k = 0; k < nx; k++) s += SM[r + k * nu] * tx[k];
            t2[r] = s;
          }
        }
      }
      c.sync();
      // P = PP inv(SG)' (:171-175); tu = inv(SG) t2 (:249): substitutions with SG
      {
        const int np = (i < N) ? nx * nu : 0;
        for (int idx = c.tid; idx < np + nu; idx += C::nt) {
          if (idx < np) P[idx] = PP[idx];
          else tu[idx - np] = t2[idx - np];
        }
      }
      c.sync();
      if (i < N) solve_right_t(c, P, nx, SG, nu);
      solve_lower(c, SG, nu, tu);
      // Save what the backward sweep needs: the factors M and SG themselves.
      for (int idx = c.tid; idx < nx * nx; idx += C::nt) {
        F[lay.f_minv + idx] = M[idx];
        F[lay.f_linv + idx] = Linv[idx];
        if (i < N) F[lay.f_am + idx] = AM[idx];
      }
      for (int idx = c.tid; idx < nu * nx; idx += C::nt) {
        F[lay.f_sm + idx] = SM[idx];
        if (i < N) F[lay.f_p + idx] = P[idx];
      }
      for (int idx = c.tid; idx < nu * nu; idx += C::nt) F[lay.f_sginv + idx] = SG[idx];
      for (int r = c.tid; r < nx; r += C::nt) {
        F[lay.f_tx + r] = tx[r];
        F[lay.f_th + r] = th[r];
      }
      for (int r = c.tid; r < nu; r += C::nt) F[lay.f_tu + r] = tu[r];
      if (i < N) {
        // L(i+1) = chol(sigma I + P P' + AM AM') (:177-183);
        // theta(i+1) partial = P tu + AM tx (:252-253).
        for (int idx = c.tid; idx < nx * nx + nx; idx += C::nt) {
          if (idx < nx * nx) {
            const int r = idx % nx, cc = idx / nx;
            if (r >= cc) {
              double s = (r == cc) ? sigma : 0.0;
              for (int k = 0; k < nu; k++) s += P[r + k * nx] * P[cc + k * nx];
              for (int k = 0; k < nx; k++) s += AM[r + k * nx] * AM[cc + k * nx];
              Ln[idx] = s;
            }
          } else {
            const int r = idx - nx * nx;
            double s = 0.0;
            for (int k = 0; k < nu; k++) s += P[r + k * nx] * tu[k];
            for (int k = 0; k < nx; k++) s += AM[r + k * nx] * tx[k];
            thp[r] = s;
          }
        }
        c.sync();
        if (!chol(c, Ln, nx)) return false;
        tri_inv(c, Ln, Linv, nx);  // becomes inv(L(i+1)) for the next stage
        c.sync();
      }
    }

    backward_sweep(c, dz, dl);
    post_sweep(c);
    return true;
  }

  // ---- results ---------------------------------------------------------------
  FB_DEV void write_x(const C& c) const {
    for (int i = c.tid; i < nz; i += C::nt) uz[i] = z[i];
    for (int i = c.tid; i < nl; i += C::nt) ul[i] = l[i];
    for (int i = c.tid; i < nv; i += C::nt) { uv[i] = v[i]; uy[i] = y[i]; }
  }
  FB_DEV void write_xbar(const C& c) const {
    for (int i = c.tid; i < nz; i += C::nt) uz[i] = zb[i];
    for (int i = c.tid; i < nl; i += C::nt) ul[i] = lb[i];
    for (int i = c.tid; i < nv; i += C::nt) { uv[i] = vb[i]; uy[i] = yb[i]; }
  }
  // x <- dx = xi - xk with dx.y = xi.y - xk.y + b (impl:202-210,
  // full_variable.cc:55-65).
  FB_DEV void write_certificate(const C& c) const {
    for (int i = c.tid; i < nz; i += C::nt) uz[i] = dz[i];
    for (int i = c.tid; i < nl; i += C::nt) ul[i] = dl[i];
    for (int i = c.tid; i < nv; i += C::nt) {
      uv[i] = dv[i];
      uy[i] = (y[i] - yb[i]) + bvec(i);
    }
  }
};

}  // namespace fbk
