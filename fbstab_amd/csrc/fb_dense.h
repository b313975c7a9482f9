// Dense problem policy for the batched FBstab kernel: one workgroup owns one
// dense QP  min 1/2 z'Hz + f'z  s.t. Gz = h, Az <= b  with H (nz x nz),
// G (nl x nz), A (nv x nz) column-major (fbstab/fbstab_dense.h:55-64).
//
// Reference code answered to:
//   * DenseData products and axpys            dense_data.cc:12-41
//   * DenseCholeskySolver::Initialize/Solve   dense_cholesky_solver.cc:32-127
//     (E = H + sigma I + A'Gamma A, K = [E .; G -sigma I], pivoted LDL' with
//     symmetric pivoting on the largest |diagonal| as Eigen::LDLT does,
//     pseudo-inverse of D in the solve)
//   * FullFeasibility::CheckFeasibility       full_feasibility.cc:25-88
//
// K and every iterate vector live in LDS for the whole solve; H, G, A are read
// from HBM/L2 where they are used.  When K = (nz+nl)^2 doubles does not fit
// beside the vectors (nz + nl > ~140) the KGLOBAL instance keeps K in a
// per-workgroup global scratch instead (L2-resident at these sizes), the vectors
// stay in LDS, and the generic multi-wavefront factorisation runs unchanged.  The LDL' here is right-looking (rank-1
// trailing updates spread over the workgroup) where Eigen's is left-looking;
// pivot order is the same rule, summation order differs (rounding only).
#pragma once

#include <float.h>

#include <type_traits>

#include "fb_common.h"
#include "fb_mpc.h"  // FB_LDS, lds_ptr

namespace fbk {

struct DenseData {
  const double *H, *f, *G, *h, *A, *b;
};

struct DenseLayout {
  int nz, nl, nv, nk;
  // LDS carve (offsets in doubles)
  int o_k, o_rhs, o_z, o_l, o_v, o_y, o_zb, o_lb, o_vb, o_yb, o_dz, o_dl, o_dv, o_adz,
      o_rz, o_rl, o_wz, o_wl, o_gam, o_rvm, o_perm, o_red, o_slot, lds_doubles;
  // A (nv x nz) is constant over the whole solve and used by every phase:
  // when it fits it is kept in LDS with an odd leading dimension lda (column
  // walks by different lanes then hit different banks).  a_lds == 0: read A
  // from global memory instead.
  int o_a, lda, a_lds;
  // k_global != 0: K lives in global scratch (k_doubles per workgroup), not at o_k
  int k_global;
  long k_doubles;
  // v_global != 0 (with k_global): the iterate vectors do not fit the LDS either and live
  // behind K in the workgroup's global scratch (v_doubles of them; the o_* offsets then
  // count from there) - the LDS holds the reduction scratch alone.  The reference
  // allocates on the heap for any size (fbstab_dense.cc:18-42).
  int v_global;
  long v_doubles;
  // wave != 0 (nz + nl <= 64, K in LDS): the factorisation and the substitutions run
  // on ONE wavefront with the matrix rows in registers (ldlt_rows) - the first
  // wavefront of the workgroup; the others wait at a barrier.  The K region then
  // also holds the multipliers (nk per pivot); one row buffer and the per-row
  // pivot bookkeeping sit behind it.
  int wave, o_rowbuf, o_dpiv, o_ord;

  __host__ __device__ void init(int nz_, int nl_, int nv_, int nthreads) {
    wave = nz_ + nl_ <= 64;
    carve(nz_, nl_, nv_, nthreads, 0, 0);
    if ((long)lds_doubles * 8 > 160 * 1024 || k_doubles > (1L << 30)) {
      wave = 0;
      carve(nz_, nl_, nv_, nthreads, 1, 0);
      if ((long)lds_doubles * 8 > 160 * 1024) carve(nz_, nl_, nv_, nthreads, 1, 1);
    }
  }

  __host__ __device__ void carve(int nz_, int nl_, int nv_, int nthreads, int kg, int vg) {
    nz = nz_; nl = nl_; nv = nv_; nk = nz + nl;
    k_doubles = (long)nk * nk;
    k_global = kg;
    v_global = vg;
    v_doubles = 0;
    long s = 0;
    o_k = 0;
    o_rowbuf = o_dpiv = o_ord = 0;
    if (!k_global) s += k_doubles;  // (wave: the multipliers of step k at k * nk + row)
    if (wave) {
      o_rowbuf = s; s += 64;
      o_dpiv = s; s += 64;
      o_ord = s; s += 34;  // 64 ints and the factorisation's verdict
    }
    o_rhs = s; s += nk;
    o_z = s; s += nz;  o_l = s; s += nl;  o_v = s; s += nv;  o_y = s; s += nv;
    o_zb = s; s += nz; o_lb = s; s += nl; o_vb = s; s += nv; o_yb = s; s += nv;
    o_dz = s; s += nz; o_dl = s; s += nl; o_dv = s; s += nv; o_adz = s; s += nv;
    o_rz = s; s += nz; o_rl = s; s += nl; o_wz = s; s += nz; o_wl = s; s += nl;
    o_gam = s; s += nv; o_rvm = s; s += nv;
    o_perm = s; s += (nk + 1) / 2 + 1;  // nk ints
    if (v_global) {  // everything so far sits in global scratch; the LDS starts here
      v_doubles = (s + 15) & ~15L;
      s = 0;
    }
    o_red = s; s += kMaxReduce * ((nthreads + 63) / 64);
    o_slot = s; s += 2;  // (the queue's hand-out word of multi-wavefront workgroups)
    s = (s + 1) & ~1L;
    lda = nv | 1;
    o_a = s;
    // two workgroups per CU must still fit
    a_lds = !k_global && (s + (long)lda * nz) * 8 <= 80 * 1024 ? 1 : 0;
    if (a_lds) s += lda * nz;
    s = (s + 1) & ~1L;
    lds_doubles = s > (1L << 28) ? (1 << 28) : (int)s;  // (callers reject what exceeds the LDS)
  }
};

template <class C, bool KGLOBAL = false, bool VGLOBAL = false>
struct DenseProblem {
  typedef typename std::conditional<KGLOBAL, double*, lds_ptr>::type kptr;
  // the iterate vectors: LDS, or the workgroup's global scratch (DenseLayout::v_global)
  typedef typename std::conditional<VGLOBAL, double*, lds_ptr>::type vptr;
  typedef typename std::conditional<VGLOBAL, int*, FB_LDS int*>::type iptr;
  static constexpr bool kFusedTrial = false;  // see fb_algorithm.h
  static constexpr bool kOwnVectorOps = false;  // the Solver loops over the flat vectors below
  DenseLayout lay;
  DenseData D;
  double *uz, *ul, *uv, *uy;
  lds_ptr lds;
  int nz, nl, nv;
  vptr z, l, v, y, zb, lb, vb, yb, dz, dl, dv, adz, rz, rl, wz, wl;
  vptr gam, rvm, rhs;
  lds_ptr Al;
  kptr K;
  iptr perm;

  FB_DEV void bind(const DenseLayout& L_, const DenseData& D_, double* uz_, double* ul_,
                   double* uv_, double* uy_, lds_ptr lds_, double* k_scratch = nullptr) {
    lay = L_; D = D_; uz = uz_; ul = ul_; uv = uv_; uy = uy_; lds = lds_;
    nz = lay.nz; nl = lay.nl; nv = lay.nv;
    if constexpr (KGLOBAL) K = k_scratch;
    else K = lds + lay.o_k;
    vptr vb_;  // base of the vector carve
    if constexpr (VGLOBAL) vb_ = k_scratch + lay.k_doubles;
    else vb_ = lds;
    rhs = vb_ + lay.o_rhs;
    z = vb_ + lay.o_z; l = vb_ + lay.o_l; v = vb_ + lay.o_v; y = vb_ + lay.o_y;
    zb = vb_ + lay.o_zb; lb = vb_ + lay.o_lb; vb = vb_ + lay.o_vb; yb = vb_ + lay.o_yb;
    dz = vb_ + lay.o_dz; dl = vb_ + lay.o_dl; dv = vb_ + lay.o_dv; adz = vb_ + lay.o_adz;
    rz = vb_ + lay.o_rz; rl = vb_ + lay.o_rl; wz = vb_ + lay.o_wz; wl = vb_ + lay.o_wl;
    gam = vb_ + lay.o_gam; rvm = vb_ + lay.o_rvm;
    perm = (iptr)(vb_ + lay.o_perm);
    Al = lds + lay.o_a;
  }

  // dot of column j of a column-major m-row matrix with an LDS vector
  FB_DEV double col_dot(const double* M, int m, int j, vptr x) const {
    const double* col = M + (long)j * m;
    double s = 0.0;
    for (int k = 0; k < m; k++) s += col[k] * x[k];
    return s;
  }
  // dot of row i of a column-major m x n matrix with an LDS vector
  FB_DEV double row_dot(const double* M, int m, int n, int i, vptr x) const {
    double s = 0.0;
    for (int k = 0; k < n; k++) s += M[i + (long)k * m] * x[k];
    return s;
  }

  // (A x)_i and (A' x)_j from the LDS copy of A when present
  FB_DEV double A_row_dot(int i, vptr x) const {
    if (lay.a_lds) {
      double s = 0.0;
      for (int k = 0; k < nz; k++) s += Al[i + k * lay.lda] * x[k];
      return s;
    }
    return row_dot(D.A, nv, nz, i, x);
  }
  FB_DEV double A_col_dot(int j, vptr x) const {
    if (lay.a_lds) {
      lds_ptr col = Al + j * lay.lda;
      double s = 0.0;
      for (int k = 0; k < nv; k++) s += col[k] * x[k];
      return s;
    }
    return col_dot(D.A, nv, j, x);
  }

  FB_DEV double forcing_norm(const C& c) const {  // dense_data.h:72-73
    double s[1] = {0.0};
    for (int i = c.tid; i < nz; i += C::nt) s[0] += D.f[i] * D.f[i];
    for (int i = c.tid; i < nl; i += C::nt) s[0] += D.h[i] * D.h[i];
    for (int i = c.tid; i < nv; i += C::nt) s[0] += D.b[i] * D.b[i];
    c.sum(s);
    return sqrt(s[0]);
  }
  FB_DEV int num_primal_dual() const { return nz + nl + nv; }
  FB_DEV double bvec(int i) const { return D.b[i]; }

  FB_DEV void load_guess(const C& c) const {
    FB_WAVE_TIMER(9);
    for (int i = c.tid; i < nz; i += C::nt) z[i] = uz[i];
    for (int i = c.tid; i < nl; i += C::nt) l[i] = ul[i];
    for (int i = c.tid; i < nv; i += C::nt) v[i] = uv[i];
    if (lay.a_lds) {
      for (int e = c.tid; e < nv * nz; e += C::nt) Al[(e % nv) + (e / nv) * lay.lda] = D.A[e];
    }
    c.sync();
    for (int i = c.tid; i < nv; i += C::nt) y[i] = D.b[i] - A_row_dot(i, z);
    c.sync();
  }

  // rz = Hz + f + G'l + A'v ; rl = h - Gz (full_residual.cc:79-91)
  FB_DEV void residual(const C& c) const {
    FB_WAVE_TIMER(15);
    for (int i = c.tid; i < nz + nl; i += C::nt) {
      if (i < nz) {
        rz[i] = D.f[i] + row_dot(D.H, nz, nz, i, z) + col_dot(D.G, nl, i, l) + A_col_dot(i, v);
      } else {
        const int j = i - nz;
        rl[j] = D.h[j] - row_dot(D.G, nl, nz, j, z);
      }
    }
    c.sync();
  }

  FB_DEV int feasibility(const C& c, double tol) const {  // full_feasibility.cc:25-88
    FB_WAVE_TIMER(16);
    double mx[5] = {-1e300, 0.0, 0.0, 0.0, 0.0};
    double sm[2] = {0.0, 0.0};
    double ul_[1] = {0.0};
    for (int i = c.tid; i < nz + nl + nv; i += C::nt) {
      if (i < nz) {
        mx[2] = fmax(mx[2], fabs(row_dot(D.H, nz, nz, i, dz)));
        mx[3] = fmax(mx[3], fabs(dz[i]));
        mx[4] = fmax(mx[4], fabs(A_col_dot(i, dv) + col_dot(D.G, nl, i, dl)));
        sm[0] += D.f[i] * dz[i];
      } else if (i < nz + nl) {
        const int j = i - nz;
        mx[1] = fmax(mx[1], fabs(row_dot(D.G, nl, nz, j, dz)));
        ul_[0] = fmax(ul_[0], fabs(dl[j]));
        sm[1] += D.h[j] * dl[j];
      } else {
        const int j = i - nz - nl;
        mx[0] = fmax(mx[0], A_row_dot(j, dz));
        ul_[0] = fmax(ul_[0], fabs(dv[j]));
        sm[1] += D.b[j] * dv[j];
      }
    }
    c.max(mx);
    c.sum(sm);
    c.max(ul_);
    const double d1 = mx[0], d2 = mx[1], d3 = mx[2], w = mx[3], p1 = mx[4];
    const double d4 = sm[0], p2 = sm[1], u = ul_[0];
    bool dual_feasible = true, primal_feasible = true;
    if ((d1 <= w * tol) && (d2 <= tol * w) && (d3 <= tol * w) && (d4 < 0) && (w > 1e-14))
      dual_feasible = false;
    if ((p1 <= tol * u) && (p2 < 0)) primal_feasible = false;
    if (primal_feasible && dual_feasible) return kFeasible;
    if (primal_feasible && !dual_feasible) return kDualInfeasible;
    if (!primal_feasible && dual_feasible) return kPrimalInfeasible;
    return kBothInfeasible;
  }

  // The factorisation and the substitutions are chains of ~6 dependent phases
  // per pivot with little work in each (n = nz + nl ~ 60): spread over four
  // wavefronts every phase costs a workgroup barrier.  They run on the first
  // wavefront alone, where a phase boundary is an LDS fence, while the others
  // wait at one barrier.
  template <class F>
  FB_DEV void on_first_wave(const C& c, F&& f) const {
    if constexpr (C::nt > 64) {
      if (c.tid < 64) {
        Ctx<64> w;
        w.tid = c.tid;
        w.red = c.red;
        f(w);
      }
      c.sync();
    } else {
      f(c);
    }
  }
  FB_DEV bool ldlt(const C& c) const {
    {
      // A NaN on the diagonal (an iterate that overflowed) ends Eigen's factorisation: its
      // pivot search compares false against it and the pivot it takes is invalid over a
      // non-zero column - NumericalIssue, the reference throws (impl:263-267).  (The
      // searches below would find NO pivot among NaNs.)
      double bad[1] = {0.0};
      for (int i = c.tid; i < lay.nk; i += C::nt) {
        const double d = K[i + (long)i * lay.nk];
        if (d != d) bad[0] = 1.0;
      }
      c.max(bad);
      if (bad[0] != 0.0) return false;
    }
    // (the one-wavefront and the fused factorisations need a wavefront: a workgroup narrower than one -
    //  tests/hostsim's single thread - takes the general loop)
    if constexpr (!KGLOBAL && C::nt >= 64) {
      if (lay.wave) {
        // (the verdict travels through LDS: the other wavefronts wait at the barrier)
        FB_LDS int* okw = (FB_LDS int*)(lds + lay.o_ord) + 64;
        on_first_wave(c, [&](const auto& w) {
          const bool ok = ldlt_rows(w);
          if (w.tid == 0) *okw = ok ? 1 : 0;
          w.sync();
        });
        return *okw != 0;
      }
    }
    if constexpr (!KGLOBAL && C::nt > 64) {
      if (lay.nk <= 64) return ldlt_fused(c);
    }
    // (running the whole factorisation on one wavefront was measured too: 27.3 ms
    // against 23.3 on config 2 - the trailing update wants the four of them)
    return ldlt_impl(c);
  }

  // Pivoted LDL' for n <= 64 on ONE wavefront with the matrix in registers: lane t
  // holds row t of the (full, symmetric) trailing matrix, Kr[j] = K[t][j].  Same
  // pivot rule as Eigen::LDLT (largest |diagonal| of what is left, the first maximum
  // wins), but nothing is swapped: eliminated rows and columns simply drop out of
  // the pivot search, and the elimination order is kept as a list (perm[k] = row
  // eliminated at step k).  A step is: the pivot row goes to LDS once (so that every
  // lane can fetch "its" entry of it, the multiplier column), then every lane updates
  // its whole row with the pivot row broadcast through v_readlane into scalar
  // registers - no barrier, no LDS round trip per column.  The multipliers of step k
  // are stored at Lm[n k + t] for the substitutions (ldlt_solve_rows); the tracked
  // diagonal feeds the next search.  dense_cholesky_solver.cc:52-79 calls
  // Eigen::LDLT::compute: right-looking here, left-looking there - the same
  // factorisation up to rounding.
  // a[p] for a wavefront-uniform p: a tree of scalar branches ends in one move (the
  // array lives in registers, which cannot be indexed at run time)
  template <int LO, int HI>
  static FB_DEV double pick(const double (&a)[64], int p) {
    if constexpr (HI - LO == 1) {
      return a[LO];
    } else {
      constexpr int MID = (LO + HI) / 2;
      return p < MID ? pick<LO, MID>(a, p) : pick<MID, HI>(a, p);
    }
  }
  FB_DEV bool ldlt_rows(const Ctx<64>& c) const {
    const int n = lay.nk, t = c.tid;
    const bool in = t < n;
    double Kr[64];
#pragma unroll
    for (int j = 0; j < 64; j++) {
      const int jj = j < n ? j : 0, tt = in ? t : 0;
      const double a = jj <= tt ? K[tt + jj * n] : K[jj + tt * n];  // lower triangle is what was assembled
      Kr[j] = (in && j < n) ? a : 0.0;
    }
    double dg = in ? K[t + t * n] : 0.0;
    c.sync();  // K has been read: its region now takes the multipliers
    lds_ptr Lm = lds + lay.o_k;
    lds_ptr rowbuf = lds + lay.o_rowbuf;
    bool alive = in;
    int ord = 64;       // step at which this row was eliminated
    double dpiv = 0.0;  // its pivot
    bool found_zero_pivot = false;
    for (int k = 0; k < n; k++) {
      double best = alive ? fabs(dg) : -1.0;
      int p = alive ? t : 64;
      C::wave_argmax_first(best, p);
      p = __builtin_amdgcn_readfirstlane(p);
      const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dg), p),
                                        __builtin_amdgcn_readlane(__double2loint(dg), p));
      if (t == 0) perm[k] = p;
      const bool valid = fabs(d) > 0.0;
      if (found_zero_pivot && valid) return false;
      if (!valid) found_zero_pivot = true;
      typedef double dbl2 __attribute__((ext_vector_type(2)));
      if (t == p) {
        // the pivot row goes to LDS once: every lane fetches "its" entry of it (the
        // multiplier column, by symmetry) and reads the row back as broadcasts
#pragma unroll
        for (int j = 0; j < 64; j += 2) {
          dbl2 v2 = {Kr[j], Kr[j + 1]};
          *reinterpret_cast<FB_LDS dbl2*>(rowbuf + j) = v2;
        }
        alive = false;
        ord = k;
        dpiv = d;
      }
      c.sync();
      const double colp = rowbuf[t];  // K[p][t], for K[t][p]
      const double l = (alive && valid) ? colp * (1.0 / d) : 0.0;
      if (in) Lm[n * k + t] = l;
      if (valid) {
        const double nl_ = -l;
#pragma unroll
        for (int j0 = 0; j0 < 64; j0 += 8) {
          dbl2 sj[4];
#pragma unroll
          for (int u = 0; u < 4; u++) sj[u] = *reinterpret_cast<FB_LDS const dbl2*>(rowbuf + j0 + 2 * u);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 4; u++) {
            Kr[j0 + 2 * u] = fma(nl_, sj[u][0], Kr[j0 + 2 * u]);
            Kr[j0 + 2 * u + 1] = fma(nl_, sj[u][1], Kr[j0 + 2 * u + 1]);
          }
        }
        dg = fma(nl_, colp, dg);
      }
      c.sync();  // rowbuf is rewritten by the next pivot
    }
    (lds + lay.o_dpiv)[t] = dpiv;
    ((FB_LDS int*)(lds + lay.o_ord))[t] = ord;
    c.sync();
    return true;
  }

  // rhs <- K^{-1} rhs with the factors of ldlt_rows (P' L^{-T} D^{+} L^{-1} P of
  // dense_cholesky_solver.cc:112 with the permutation implicit in the elimination
  // order).  Lane t owns entry t of the right-hand side throughout.
  FB_DEV void ldlt_solve_rows(const Ctx<64>& c) const {
    const int n = lay.nk, t = c.tid;
    const bool in = t < n;
    lds_ptr Lm = lds + lay.o_k;
    const double dpiv = (lds + lay.o_dpiv)[t];
    const int ord = ((FB_LDS int*)(lds + lay.o_ord))[t];
    double x = in ? rhs[t] : 0.0;
    // L y = b: step k hands the entry of the row eliminated at step k to all later rows
    for (int k0 = 0; k0 < n - 1; k0 += 8) {
      double lk[8];
#pragma unroll
      for (int u = 0; u < 8; u++) lk[u] = (in && k0 + u < n - 1) ? Lm[n * (k0 + u) + t] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int k = k0 + u < n ? k0 + u : n - 1;
        const int p = __builtin_amdgcn_readfirstlane(perm[k]);
        const double xk = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), p),
                                           __builtin_amdgcn_readlane(__double2loint(x), p));
        x = fma(-lk[u], xk, x);  // the multiplier is 0 for rows eliminated up to step k
      }
    }
    x = (in && fabs(dpiv) > DBL_MIN) ? x / dpiv : 0.0;  // pseudo-inverse of D
    // L' w = y: the entry of the row eliminated at step k goes to the rows eliminated
    // before it, each of which reads the multiplier it gave that row at its own step
    for (int k0 = n - 1; k0 > 0; k0 -= 8) {
      double lk[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int k = k0 - u > 0 ? k0 - u : 0;
        const int p = __builtin_amdgcn_readfirstlane(perm[k]);
        lk[u] = (k0 - u > 0 && ord < k) ? Lm[n * ord + p] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int k = k0 - u > 0 ? k0 - u : 0;
        const int p = __builtin_amdgcn_readfirstlane(perm[k]);
        const double xk = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), p),
                                           __builtin_amdgcn_readlane(__double2loint(x), p));
        x = fma(-lk[u], xk, x);
      }
    }
    c.sync();
    if (in) rhs[t] = x;
    c.sync();
  }

  // ldlt_impl for n <= 64 on a multi-wavefront workgroup with its phases fused:
  // the scaling of column k-1 rides along with the pivot search of step k (they
  // touch different entries), the search itself is DPP + v_readlane in the first
  // wavefront, and barriers that only separated reads are gone: two per pivot
  // (three with a swap) instead of five.
  FB_DEV bool ldlt_fused(const C& c) const {
    const int n = lay.nk;
    bool found_zero_pivot = false;
    double id_prev = 0.0;  // 1 / d of the previous pivot, 0 = nothing to scale
    for (int k = 0; k < n; k++) {
      if (id_prev != 0.0)
        for (int i = k + c.tid; i < n; i += C::nt) K[i + (k - 1) * n] *= id_prev;
      if (c.tid < 64) {
        int p = k + c.tid;
        double best = p < n ? fabs(K[p + p * n]) : -1.0;
        if (p >= n) p = n;
        C::wave_argmax_first(best, p);  // largest |diagonal|, the first maximum wins
        if (c.tid == 0) perm[k] = p;
      }
      c.sync();
      const int p = perm[k];
      if (p != k) {
        const int s = n - p - 1;
        for (int j = c.tid; j < k; j += C::nt) {
          const double t = K[k + j * n];
          K[k + j * n] = K[p + j * n];
          K[p + j * n] = t;
        }
        for (int i = c.tid; i < s; i += C::nt) {
          const double t = K[(p + 1 + i) + k * n];
          K[(p + 1 + i) + k * n] = K[(p + 1 + i) + p * n];
          K[(p + 1 + i) + p * n] = t;
        }
        for (int i = k + 1 + c.tid; i < p; i += C::nt) {
          const double t = K[i + k * n];
          K[i + k * n] = K[p + i * n];
          K[p + i * n] = t;
        }
        if (c.tid == 0) {
          const double t = K[k + k * n];
          K[k + k * n] = K[p + p * n];
          K[p + p * n] = t;
        }
        c.sync();
      }
      const double d = K[k + k * n];
      const bool valid = fabs(d) > 0.0;
      if (found_zero_pivot && valid) return false;
      if (!valid) found_zero_pivot = true;
      id_prev = 0.0;
      if (n - k - 1 > 0 && valid) {
        // trailing update of the lower triangle: K(i,j) -= K(i,k) K(j,k) / d
        const double id = 1.0 / d;
        constexpr int TW = 16, TH = C::nt / TW;
        const int ti = c.tid % TW, tj = c.tid / TW;
        for (int j = k + 1 + tj; j < n; j += TH) {
          const double ljk = K[j + k * n] * id;
          for (int i = j + ti; i < n; i += TW) K[i + j * n] -= K[i + k * n] * ljk;
        }
        id_prev = id;
        c.sync();
      }
    }
    return true;
  }
  FB_DEV void ldlt_solve(const C& c) const {
    on_first_wave(c, [&](const auto& w) {
      typedef typename std::remove_cv<typename std::remove_reference<decltype(w)>::type>::type W;
      if constexpr (W::nt == 64) {
        if constexpr (!KGLOBAL) {
          if (lay.wave) { ldlt_solve_rows(w); return; }
        }
        if (lay.nk <= 64) ldlt_solve_wave(w);
        else ldlt_solve_impl(w);
      } else {
        ldlt_solve_impl(w);
      }
    });
  }

  // One-wavefront substitutions for n <= 64 (same formulas as ldlt_solve_impl):
  // the right-hand side stays in registers, lane t owning entry t, the solved
  // entry is handed round by v_readlane, and the L columns are fetched from LDS
  // eight steps ahead of their use.
  FB_DEV void ldlt_solve_wave(const Ctx<64>& c) const {
    const int n = lay.nk, t = c.tid;
    const bool in = t < n;
    // the transpositions composed into one index map: (P b)[t] = b[pi[t]]
    int pi = t;
    for (int k = 0; k < n; k++) {
      const int p = __builtin_amdgcn_readfirstlane(perm[k]);
      const int vk = __builtin_amdgcn_readlane(pi, k), vp = __builtin_amdgcn_readlane(pi, p);
      pi = t == k ? vp : (t == p ? vk : pi);
    }
    double x = in ? rhs[pi] : 0.0;
    const double dg = in ? K[t + t * n] : 1.0;
    // L y = P b (unit lower L, column k below the diagonal)
    for (int k0 = 0; k0 < n - 1; k0 += 8) {
      double lk[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int k = k0 + u;
        lk[u] = (k < n - 1 && t > k && in) ? K[t + k * n] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int k = k0 + u < n ? k0 + u : n - 1;
        const double xk = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), k),
                                           __builtin_amdgcn_readlane(__double2loint(x), k));
        x = fma(-lk[u], xk, x);  // lk = 0 outside the column
      }
    }
    x = fabs(dg) > DBL_MIN ? x / dg : 0.0;  // pseudo-inverse of D
    // L' w = y (row k of L left of the diagonal)
    for (int k0 = n - 1; k0 > 0; k0 -= 8) {
      double lk[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int k = k0 - u;
        lk[u] = (k > 0 && t < k) ? K[k + t * n] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int k = k0 - u > 0 ? k0 - u : 0;
        const double xk = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), k),
                                           __builtin_amdgcn_readlane(__double2loint(x), k));
        x = fma(-lk[u], xk, x);
      }
    }
    c.sync();
    if (in) rhs[pi] = x;  // P' w
    c.sync();
  }

  // Pivoted LDL' of the lower triangle of K in place (Eigen::LDLT semantics).
  template <class W>
  FB_DEV bool ldlt_impl(const W& c) const {
    const int n = lay.nk;
    bool found_zero_pivot = false;
    for (int k = 0; k < n; k++) {
      // largest |diagonal| in the trailing corner; the first maximum wins
      double best = -1.0;
      int p = n;
      for (int i = k + c.tid; i < n; i += W::nt) {
        const double a = fabs(K[i + i * n]);
        if (a > best) { best = a; p = i; }  // i ascending: first maximum kept
      }
      if (W::nt > 64 && n - k <= 64) {
        // every candidate sits in the first wavefront: it decides, the others
        // read the result after the barrier below
        if (c.tid < 64) {
          W::wave_argmax_first(best, p);
          if (c.tid == 0) perm[k] = p;
        }
        c.sync();
        p = perm[k];
      } else
      {
        c.argmax_first(&best, &p);
        if (c.tid == 0) perm[k] = p;
      }
      if (p != k) {
        c.sync();
        const int s = n - p - 1;
        for (int j = c.tid; j < k; j += W::nt) {
          const double t = K[k + j * n];
          K[k + j * n] = K[p + j * n];
          K[p + j * n] = t;
        }
        for (int i = c.tid; i < s; i += W::nt) {
          const double t = K[(p + 1 + i) + k * n];
          K[(p + 1 + i) + k * n] = K[(p + 1 + i) + p * n];
          K[(p + 1 + i) + p * n] = t;
        }
        for (int i = k + 1 + c.tid; i < p; i += W::nt) {
          const double t = K[i + k * n];
          K[i + k * n] = K[p + i * n];
          K[p + i * n] = t;
        }
        if (c.tid == 0) {
          const double t = K[k + k * n];
          K[k + k * n] = K[p + p * n];
          K[p + p * n] = t;
        }
      }
      c.sync();
      const double d = K[k + k * n];
      const bool valid = fabs(d) > 0.0;
      if (found_zero_pivot && valid) return false;
      if (!valid) found_zero_pivot = true;
      const int rs = n - k - 1;
      if (rs > 0 && valid) {
        // trailing update of the lower triangle: K(i,j) -= K(i,k) K(j,k) / d
        const double id = 1.0 / d;
        // 2-D sweep of the trailing block without integer divisions
        constexpr int TW = W::nt >= 256 ? 16 : (W::nt >= 64 ? 8 : 1);
        constexpr int TH = W::nt / TW;
        const int ti = c.tid % TW, tj = c.tid / TW;
        for (int j = k + 1 + tj; j < n; j += TH) {
          const double ljk = K[j + k * n] * id;
          for (int i = j + ti; i < n; i += TW) K[i + j * n] -= K[i + k * n] * ljk;
        }
        c.sync();
        for (int i = k + 1 + c.tid; i < n; i += W::nt) K[i + k * n] *= id;
      }
      c.sync();
    }
    return true;
  }

  // rhs <- K^{-1} rhs using P' L^{-T} D^{+} L^{-1} P.
  template <class W>
  FB_DEV void ldlt_solve_impl(const W& c) const {
    const int n = lay.nk;
    if (c.tid == 0) {
      for (int k = 0; k < n; k++) {
        const int p = perm[k];
        if (p != k) { const double t = rhs[k]; rhs[k] = rhs[p]; rhs[p] = t; }
      }
    }
    c.sync();
    for (int k = 0; k < n - 1; k++) {
      const double xk = rhs[k];
      for (int i = k + 1 + c.tid; i < n; i += W::nt) rhs[i] -= K[i + k * n] * xk;
      c.sync();
    }
    for (int i = c.tid; i < n; i += W::nt) {
      const double d = K[i + i * n];
      rhs[i] = fabs(d) > DBL_MIN ? rhs[i] / d : 0.0;
    }
    c.sync();
    for (int k = n - 1; k > 0; k--) {
      const double xk = rhs[k];
      for (int j = c.tid; j < k; j += W::nt) rhs[j] -= K[k + j * n] * xk;
      c.sync();
    }
    if (c.tid == 0) {
      for (int k = n - 1; k >= 0; k--) {
        const int p = perm[k];
        if (p != k) { const double t = rhs[k]; rhs[k] = rhs[p]; rhs[p] = t; }
      }
    }
    c.sync();
  }

  FB_DEV bool newton_step(const C& c, double sigma, double alpha) const {
    FB_WAVE_LAP_DECL;
    const int n = lay.nk;
    // PFB gradients (dense_cholesky_solver.cc:54-61)
    for (int i = c.tid; i < nv; i += C::nt) {
      const double ys = y[i] + sigma * (v[i] - vb[i]);
      double g0, g1;
      pfb_gradient(ys, v[i], alpha, &g0, &g1);
      const double mu = g1 + sigma * g0;
      gam[i] = g0 / mu;
      rvm[i] = -pfb(ys, v[i], alpha) / mu;
    }
    c.sync();
    FB_WAVE_LAP(10);
    // K = [H + sigma I + A'Gamma A  .; G  -sigma I] (lower; :52-69) and the
    // eliminated right-hand side (:98-104).
#if !defined(FB_DENSE_NO_MFMA)
    if (lay.a_lds && (C::nt & 63) == 0 && nz <= 64 && (nv & 3) == 0) {
      // E = H + sigma I + A' Gamma A on the matrix cores: one QP per workgroup, so
      // v_mfma_f64_16x16x4 fits - the 16x16 tiles of the lower triangle of the
      // (padded) 64x64 product are dealt to the four wavefronts, each tile
      // accumulates over nv/4 steps with both operands read from the LDS copy
      // of A (A operand lane l: A[k0 + l/16][16 I + l%16], B operand the same
      // entry of block column J times Gamma).  Accumulation order differs from
      // the scalar loop below (rounding only).
      typedef double d4 __attribute__((ext_vector_type(4)));
      const int wave = c.tid >> 6, lane = c.tid & 63;
      const int kq = lane >> 4, ij = lane & 15;
      const int nt16 = (nz + 15) >> 4;
      const int ntiles = nt16 * (nt16 + 1) / 2;
      for (int t0 = wave; t0 < ntiles; t0 += C::nt / 64) {
        // tile index -> (I, J), I >= J (row-major over the lower triangle)
        int I = 0, rem = t0;
        while (rem > I) { rem -= I + 1; I++; }
        const int J = rem;
        const int ci = 16 * I + ij < nz ? 16 * I + ij : nz - 1;  // padded columns re-read the last one
        const int cj = 16 * J + ij < nz ? 16 * J + ij : nz - 1;
        lds_ptr ai = Al + ci * lay.lda + kq;
        lds_ptr aj = Al + cj * lay.lda + kq;
        vptr gk = gam + kq;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < nv; k0 += 4)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[k0], gk[k0] * aj[k0], acc, 0, 0, 0);
        // D layout: column = lane % 16, row = lane / 16 + 4 * register
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int i = 16 * I + kq + 4 * q, j = 16 * J + ij;
          if (i < nz && j <= i) K[i + j * n] = D.H[i + (long)j * nz] + (i == j ? sigma : 0.0) + acc[q];
        }
      }
      for (int e = c.tid; e < nl * n + n; e += C::nt) {
        if (e < nl * n) {
          const int i = nz + e % nl, j = e / nl;  // rows of [G  -sigma I]
          if (j < nz) K[i + j * n] = D.G[(i - nz) + (long)j * nl];
          else if (i >= j) K[i + j * n] = (i == j) ? -sigma : 0.0;
        } else {
          const int j = e - nl * n;
          if (j < nz) {
            rhs[j] = -(rz[j] + sigma * (z[j] - zb[j])) - A_col_dot(j, rvm);
          } else {
            const int q = j - nz;
            rhs[j] = rl[q] + sigma * (l[q] - lb[q]);
          }
        }
      }
    } else
#endif
    if (lay.a_lds) {
      // E = H + sigma I + A' Gamma A (lower), A from LDS: a task is row i and a
      // chunk of JC columns, so that Gamma_k A[k][i] is reused JC times
      constexpr int JC = 6;
      const int nch = (nz + JC - 1) / JC;
      for (int t = c.tid; t < nz * nch; t += C::nt) {
        const int i = t / nch, j0 = (t % nch) * JC;
        if (j0 > i) continue;
        double acc[JC];
#pragma unroll
        for (int m = 0; m < JC; m++) acc[m] = 0.0;
        lds_ptr ai = Al + i * lay.lda;
        lds_ptr aj = Al + j0 * lay.lda;
        int off[JC];  // columns past nz re-read column j0; their sums are discarded
#pragma unroll
        for (int m = 0; m < JC; m++) off[m] = (j0 + m < nz) ? m * lay.lda : 0;
        for (int k = 0; k < nv; k++) {
          const double g = gam[k] * ai[k];
#pragma unroll
          for (int m = 0; m < JC; m++) acc[m] = fma(g, aj[k + off[m]], acc[m]);
        }
#pragma unroll
        for (int m = 0; m < JC; m++) {
          const int j = j0 + m;
          if (j <= i && j < nz) K[i + j * n] = D.H[i + (long)j * nz] + (i == j ? sigma : 0.0) + acc[m];
        }
      }
      for (int e = c.tid; e < nl * n + n; e += C::nt) {
        if (e < nl * n) {
          const int i = nz + e % nl, j = e / nl;  // rows of [G  -sigma I]
          if (j < nz) K[i + j * n] = D.G[(i - nz) + (long)j * nl];
          else if (i >= j) K[i + j * n] = (i == j) ? -sigma : 0.0;
        } else {
          const int j = e - nl * n;
          if (j < nz) {
            rhs[j] = -(rz[j] + sigma * (z[j] - zb[j])) - A_col_dot(j, rvm);
          } else {
            const int q = j - nz;
            rhs[j] = rl[q] + sigma * (l[q] - lb[q]);
          }
        }
      }
    } else
    for (int e = c.tid; e < n * n + n; e += C::nt) {
      if (e < n * n) {
        const int i = e % n, j = e / n;
        if (i < j) continue;
        double s;
        if (i < nz) {
          s = D.H[i + (long)j * nz] + (i == j ? sigma : 0.0);
          const double* ai = D.A + (long)i * nv;
          const double* aj = D.A + (long)j * nv;
          for (int k = 0; k < nv; k++) s += gam[k] * ai[k] * aj[k];
        } else if (j < nz) {
          s = D.G[(i - nz) + (long)j * nl];
        } else {
          s = (i == j) ? -sigma : 0.0;
        }
        K[e] = s;
      } else {
        const int j = e - n * n;
        if (j < nz) {
          rhs[j] = -(rz[j] + sigma * (z[j] - zb[j])) - col_dot(D.A, nv, j, rvm);
        } else {
          const int q = j - nz;
          rhs[j] = rl[q] + sigma * (l[q] - lb[q]);
        }
      }
    }
    c.sync();
    FB_WAVE_LAP(11);
    if (!ldlt(c)) return false;
    FB_WAVE_LAP(12);
    ldlt_solve(c);
    FB_WAVE_LAP(13);
    for (int i = c.tid; i < n; i += C::nt) {
      if (i < nz) dz[i] = rhs[i];
      else dl[i - nz] = rhs[i];
    }
    c.sync();
    // dv = rv/mus + Gamma .* (A dz) (:114-121); adz = A dz (dy = b - A dz, :124)
    for (int i = c.tid; i < nv; i += C::nt) {
      const double a = A_row_dot(i, dz);
      adz[i] = a;
      dv[i] = rvm[i] + gam[i] * a;
    }
    c.sync();
    // W = (H dz + G'dl + A'dv, -G dz)
    for (int i = c.tid; i < nz + nl; i += C::nt) {
      if (i < nz)
        wz[i] = row_dot(D.H, nz, nz, i, dz) + col_dot(D.G, nl, i, dl) + A_col_dot(i, dv);
      else
        wl[i - nz] = -row_dot(D.G, nl, nz, i - nz, dz);
    }
    c.sync();
    FB_WAVE_LAP(14);
    return true;
  }

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
