// One-wavefront dense policy: ONE 64-lane wavefront owns one dense QP
//   min 1/2 z'Hz + f'z  s.t. Gz = h, Az <= b      (fbstab/fbstab_dense.h:55-64)
// with nz + nl <= 64, for EVERY phase of the solve, so that eight QPs are in
// flight on a CU (two wavefronts per SIMD) and hide each other's latencies.
// The four-wavefront policy of fb_dense.h keeps K, A and the iterates of a QP
// in 80 KB of LDS: two QPs per CU, and three of its four wavefronts wait at a
// barrier while the first one factors.  Here a QP holds ~19 KB of LDS:
//
//   * the KKT matrix  K = [H + sigma I + A'Gamma A  G'; G  -sigma I]
//     (dense_cholesky_solver.cc:52-69) lives in REGISTERS, lane t holding row t
//     of the full symmetric matrix (64 doubles);
//   * A' Gamma A is accumulated on the matrix cores (v_mfma_f64_16x16x4, the ten
//     16x16 tiles of the lower triangle), operands read straight from the caller's
//     column-major A: a trip takes sixteen rows and a lane holds two adjacent row
//     PAIRS of its column (16 bytes each; the four lanes that share a column cover
//     64 contiguous bytes per access) - the contraction runs over the rows of A in
//     a permuted order, which the sum does not care about - so that no transposed
//     copy of A exists (40 KB per QP at 50/10/100: a third of the scratch a resident
//     QP used to hold); the tiles reach the row layout through a 64 x 16 staging
//     panel in LDS, one column block at a time;
//   * LDL' by Eigen::LDLT's pivoting rule (largest |diagonal| of what is left, the first
//     maximum wins; dense_cholesky_solver.cc:70-79), the DEFAULT since round 4: it
//     eliminates in place without swapping anything and keeps its multipliers in a 64 x 64
//     global scratch.  Same elimination order as the reference, same iteration counts on
//     every QP tried ("which order is the default");
//   * LDL' in the NATURAL order, unrolled over its steps and fused with both
//     substitutions (factor_solve_static, round 3), a per-handle option
//     (fbstab_hip_dense_set_factorisation: NATURAL, or AUTO = natural until a QP shows
//     itself ill-conditioned): K is quasi-definite, every elimination order factors it,
//     and a compile-time order makes every register index an immediate - no pivot search,
//     no pick of a run-time column, half the trailing updates, the multipliers used where
//     they are formed and the backward sweep a lane-local dot product; 1.6 x faster per
//     launch, differently ordered rounding errors (a zero, denormal, infinite or NaN pivot
//     still goes to the pivoted path, whose verdict is Eigen's);
//   * right-hand side and solution of the substitutions stay in a register
//     (lane t owns entry t), solved entries are handed round by v_readlane.
//
// Reference code answered to: DenseData products (dense_data.cc:12-41),
// DenseCholeskySolver::Initialize / Solve (dense_cholesky_solver.cc:32-127),
// FullFeasibility::CheckFeasibility (full_feasibility.cc:25-88).  The pivoted
// path is the same factorisation as Eigen's up to rounding (right-looking here,
// left-looking there; the pivot order is the same rule); the natural-order path
// solves the same systems with a different rounding, which shows where the
// answer is not unique - the multipliers of dual-degenerate QPs (tests/
// test_gpu_parity.py, _unique_duals) - and where a convergence test is decided by the
// last digits: about one per cent of heavily degenerate QPs take a different number of
// iterations (profiles/r04_a_dense_order_choice.txt).
#pragma once

#include <float.h>

#include "fb_dense.h"
#include "fb_row16.h"  // dpp_mov

namespace fbk {


// Sub-phase cycles of the factorisation loop (diagnostic builds): summed in registers,
// one atomic per factorisation (an atomic per lap would be most of what is measured).
// (-DFB_DW_NO_INNER_LAPS keeps the phase laps of newton_step only: the laps inside the unrolled
// factorisation cost it registers, tools/dense_variant.sh var_dstamp -DFB_CLOCKSTAMP -DFB_DW_NO_INNER_LAPS -> tools/_build/)
#if (defined(FB_STAMP) || defined(FB_CLOCKSTAMP)) && !defined(FB_DW_NO_INNER_LAPS)
#define FB_DW_LAPS_DECL long long dw_acc_[4] = {0, 0, 0, 0}; long long dw_t_ = __builtin_readcyclecounter()
#define FB_DW_LAP(i) do { const long long n_ = __builtin_readcyclecounter(); dw_acc_[i] += n_ - dw_t_; dw_t_ = n_; } while (0)
#define FB_DW_LAPS_FLUSH(base) do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 4; i_++) atomicAdd(&g_stamps[(base) + i_], (unsigned long long)dw_acc_[i_]); } while (0)
#else
#define FB_DW_LAPS_DECL
#define FB_DW_LAP(i)
#define FB_DW_LAPS_FLUSH(base)
#endif

struct DenseWaveLayout {
  static constexpr int kLd = 17;  // leading dimension of the staging panel (odd: rows hit different banks)
  int nz, nl, nv, nk;
  // Elimination order of the LDL' (fbstab_hip_dense_set_factorisation; see factor_solve_static):
  // 1 = Eigen's rule at every step (the reference's order; the default), 2 = natural order
  // whenever it exists, 0 = natural order until the QP shows itself ill-conditioned, then
  // Eigen's rule for the rest of its solve (`sticky`): a step whose natural-order pivots
  // span more than spread_bits binary orders of magnitude is factored again, and a QP with
  // at least nz - nl "active" inequality rows at an iterate - rows whose barrier weight
  // Gamma = gamma / mu exceeds 2^-act_bits / sigma - switches before that step (act_bits = 0:
  // no such test).
  int order = 1;
  int spread_bits = 32;
  int sticky = 1;
  int act_bits = 20;
  // LDS carve (offsets in doubles)
  int o_z, o_l, o_v, o_y, o_zb, o_lb, o_vb, o_yb, o_dz, o_dl, o_dv, o_adz, o_rz, o_rl, o_wz, o_wl,
      o_gam, o_rvm, o_rowbuf, o_stage, lds_doubles;
  // global scratch of one workgroup (doubles): the multipliers, H in accumulator
  // layout (40 x 64) and G' (Gt[64 q + t] = G[q][t])
  long o_lg, o_hd, o_gt, ws_doubles;

  __host__ __device__ void init(int nz_, int nl_, int nv_) {
    nz = nz_; nl = nl_; nv = nv_; nk = nz + nl;
    int s = 0;
    o_z = s; s += nz;  o_l = s; s += nl;  o_v = s; s += nv;  o_y = s; s += nv;
    o_zb = s; s += nz; o_lb = s; s += nl; o_vb = s; s += nv; o_yb = s; s += nv;
    o_dz = s; s += nz; o_dl = s; s += nl; o_dv = s; s += nv; o_adz = s; s += nv;
    o_rz = s; s += nz; o_rl = s; s += nl; o_wz = s; s += nz; o_wl = s; s += nl;
    o_gam = s; s += nv; o_rvm = s; s += nv;
    s = (s + 1) & ~1;
    o_rowbuf = s; s += 64;
    o_stage = s; s += 64 * kLd;
    lds_doubles = (s + 1) & ~1;
    o_lg = 0;
    o_hd = o_lg + 64 * 64;
    o_gt = o_hd + 40 * 64;
    ws_doubles = o_gt + 64 * (long)(nl > 0 ? nl : 1);
  }
  // nz + nl <= 64 and the iterate vectors fit a share of the LDS that leaves room
  // for at least four workgroups per CU
  __host__ __device__ bool fits() const { return nk <= 64 && nk >= 1 && (long)lds_doubles * 8 <= 40 * 1024; }
};

struct DenseWave {
  typedef Ctx<64> C;
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  typedef double d4 __attribute__((ext_vector_type(4)));
  static constexpr bool kFusedTrial = false;    // see fb_algorithm.h
  static constexpr bool kOwnVectorOps = false;  // the Solver loops over the flat LDS vectors below
  static constexpr int LD = DenseWaveLayout::kLd;
  DenseWaveLayout lay;
  DenseData D;
  double *uz, *ul, *uv, *uy;
  // global scratch: the multipliers Lg[64 k + t], the lower triangle of H as the MFMA
  // accumulators hold it (Hd[64 (4 tile + q) + lane]) and G'
  double *Lg, *Hd, *Gt;
  int* nfallback;  // Newton steps of this launch that went to the pivoted path behind a natural-order attempt
  mutable bool went_pivoted;  // (wavefront-uniform) this QP has been handed to the pivoted path before
  int nz, nl, nv;
  lds_ptr z, l, v, y, zb, lb, vb, yb, dz, dl, dv, adz, rz, rl, wz, wl;
  lds_ptr gam, rvm, rowbuf, stage;

  FB_DEV void bind(const DenseWaveLayout& L_, const DenseData& D_, double* uz_, double* ul_, double* uv_,
                   double* uy_, lds_ptr lds, double* ws, int* nfallback_ = nullptr) {
    lay = L_; D = D_; uz = uz_; ul = ul_; uv = uv_; uy = uy_;
    nfallback = nfallback_;
    went_pivoted = false;
    nz = lay.nz; nl = lay.nl; nv = lay.nv;
    z = lds + lay.o_z; l = lds + lay.o_l; v = lds + lay.o_v; y = lds + lay.o_y;
    zb = lds + lay.o_zb; lb = lds + lay.o_lb; vb = lds + lay.o_vb; yb = lds + lay.o_yb;
    dz = lds + lay.o_dz; dl = lds + lay.o_dl; dv = lds + lay.o_dv; adz = lds + lay.o_adz;
    rz = lds + lay.o_rz; rl = lds + lay.o_rl; wz = lds + lay.o_wz; wl = lds + lay.o_wl;
    gam = lds + lay.o_gam; rvm = lds + lay.o_rvm;
    rowbuf = lds + lay.o_rowbuf; stage = lds + lay.o_stage;
    Lg = ws + lay.o_lg;
    Hd = ws + lay.o_hd;
    Gt = ws + lay.o_gt;
  }

  // The wavefront's own global stores (Lg, Hd, Gt) become visible to its other lanes:
  // one L1 serves the whole CU and is written through, what is needed is that the
  // stores have left the wavefront (s_waitcnt vmcnt(0)).
  static FB_DEV void global_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }

  // sum_k M[i + k m] x[k], k < n: row i of a column-major matrix with m rows (lanes
  // i, i+1, ... read consecutive words) times an LDS vector (broadcast reads).  Two
  // wavefronts per SIMD do not hide a trip to L2: the loads go out U at a time.
  template <int U = 16>
  static FB_DEV double row_dot(const double* M, int m, int n, int i, lds_ptr x) {
    const double* p = M + i;
    double s0 = 0.0, s1 = 0.0;
    int k = 0;
    for (; k + U <= n; k += U) {
      double a[U];
#pragma unroll
      for (int u = 0; u < U; u++) a[u] = p[(long)(k + u) * m];
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (u & 1) s1 = fma(a[u], x[k + u], s1);
        else s0 = fma(a[u], x[k + u], s0);
      }
    }
    if (k < n) {  // remainder: clamped loads, zero weights
      double a[U];
#pragma unroll
      for (int u = 0; u < U; u++) a[u] = p[(long)(k + u < n ? k + u : n - 1) * m];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const double w = k + u < n ? x[k + u < n ? k + u : n - 1] : 0.0;
        if (u & 1) s1 = fma(a[u], w, s1);
        else s0 = fma(a[u], w, s0);
      }
    }
    return s0 + s1;
  }
  // sum_k M[k + j m] x[k], k < m: column j (only used for the small G, nl rows)
  static FB_DEV double col_dot(const double* M, int m, int j, lds_ptr x) {
    const double* col = M + (long)j * m;
    double s = 0.0;
    for (int k = 0; k < m; k++) s = fma(col[k], x[k], s);
    return s;
  }
  FB_DEV double A_row_dot(int i, lds_ptr x) const { return row_dot(D.A, nv, nz, i, x); }   // (A x)_i
  // (A'x)_j = sum_k A[k + j nv] x[k]: column j is contiguous, the lane walks it sixteen
  // entries (one cache line) at a time.  Only the passes that run once per proximal
  // iteration use it (residual, feasibility); the Newton step takes A'(rv/mu) from the
  // matrix-core operands.
  FB_DEV double A_col_dot(int j, lds_ptr x) const {
    const double* p = D.A + (long)j * nv;
    double s0 = 0.0, s1 = 0.0;
    constexpr int U = 16;
    for (int k = 0; k < nv; k += U) {
      double a[U];
#pragma unroll
      for (int u = 0; u < U; u++) a[u] = p[k + u < nv ? k + u : nv - 1];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const double w = k + u < nv ? x[k + u < nv ? k + u : nv - 1] : 0.0;
        if (u & 1) s1 = fma(a[u], w, s1);
        else s0 = fma(a[u], w, s0);
      }
    }
    return s0 + s1;
  }

  FB_DEV double forcing_norm(const C& c) const {  // dense_data.h:72-73
    double s[1] = {0.0};
    for (int i = c.tid; i < nz; i += 64) s[0] += D.f[i] * D.f[i];
    for (int i = c.tid; i < nl; i += 64) s[0] += D.h[i] * D.h[i];
    for (int i = c.tid; i < nv; i += 64) s[0] += D.b[i] * D.b[i];
    c.sum(s);
    return sqrt(s[0]);
  }
  FB_DEV int num_primal_dual() const { return nz + nl + nv; }
  FB_DEV double bvec(int i) const { return D.b[i]; }

  // x <- caller's guess, y = b - A z (impl:334-347, full_variable.cc:47-53), H and G' in
  // the layouts the Newton step reads them in
  FB_DEV void load_guess(const C& c) const {
    FB_WAVE_TIMER(9);
    for (int i = c.tid; i < nz; i += 64) z[i] = uz[i];
    for (int i = c.tid; i < nl; i += 64) l[i] = ul[i];
    for (int i = c.tid; i < nv; i += 64) v[i] = uv[i];
    {
      // H as the accumulators of assemble() hold it: lane l, register q of tile (I, J)
      // <-> entry (16 I + l/16 + 4 q, 16 J + l%16), rows and columns past nz clamped
      // (never used).  Sixteen cache lines per load here, once per QP, instead of once
      // per Newton iteration.
      const int kq = c.tid >> 4, ij = c.tid & 15;
#pragma unroll
      for (int I = 0; I < 4; I++) {
#pragma unroll
        for (int J = 0; J <= I; J++) {
          const int cj = 16 * J + ij < nz ? 16 * J + ij : nz - 1;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int r = 16 * I + kq + 4 * q;
            Hd[64 * (4 * tile_of(I, J) + q) + c.tid] = D.H[(r < nz ? r : nz - 1) + (long)cj * nz];
          }
        }
      }
      for (int q = 0; q < nl; q++) Gt[64 * q + c.tid] = c.tid < nz ? D.G[q + (long)c.tid * nl] : 0.0;
    }
    global_fence();
    c.sync();
    for (int i = c.tid; i < nv; i += 64) y[i] = D.b[i] - A_row_dot(i, z);
    c.sync();
  }

  // rz = Hz + f + G'l + A'v ; rl = h - Gz (full_residual.cc:79-91)
  FB_DEV void residual(const C& c) const {
    FB_WAVE_TIMER(15);
    const int i = c.tid;
    if (i < nz) {
      rz[i] = D.f[i] + row_dot(D.H, nz, nz, i, z) + col_dot(D.G, nl, i, l) + A_col_dot(i, v);
    } else if (i < nz + nl) {
      const int j = i - nz;
      rl[j] = D.h[j] - row_dot(D.G, nl, nz, j, z);
    }
    c.sync();
  }

  FB_DEV int feasibility(const C& c, double tol) const {  // full_feasibility.cc:25-88
    FB_WAVE_TIMER(16);
    double mx[5] = {-1e300, 0.0, 0.0, 0.0, 0.0};
    double sm[2] = {0.0, 0.0};
    double ul_[1] = {0.0};
    const int i = c.tid;
    if (i < nz) {
      mx[2] = fabs(row_dot(D.H, nz, nz, i, dz));
      mx[3] = fabs(dz[i]);
      mx[4] = fabs(A_col_dot(i, dv) + col_dot(D.G, nl, i, dl));
      sm[0] = D.f[i] * dz[i];
    } else if (i < nz + nl) {
      const int j = i - nz;
      mx[1] = fabs(row_dot(D.G, nl, nz, j, dz));
      ul_[0] = fabs(dl[j]);
      sm[1] = D.h[j] * dl[j];
    }
    for (int j = c.tid; j < nv; j += 64) {
      mx[0] = fmax(mx[0], A_row_dot(j, dz));
      ul_[0] = fmax(ul_[0], fabs(dv[j]));
      sm[1] += D.b[j] * dv[j];
    }
    c.max(mx);
    c.sum(sm);
    c.max(ul_);
    const double d1 = mx[0], d2 = mx[1], d3 = mx[2], w = mx[3], p1 = mx[4];
    const double d4_ = sm[0], p2 = sm[1], u = ul_[0];
    bool dual_feasible = true, primal_feasible = true;
    if ((d1 <= w * tol) && (d2 <= tol * w) && (d3 <= tol * w) && (d4_ < 0) && (w > 1e-14))
      dual_feasible = false;
    if ((p1 <= tol * u) && (p2 < 0)) primal_feasible = false;
    if (primal_feasible && dual_feasible) return kFeasible;
    if (primal_feasible && !dual_feasible) return kDualInfeasible;
    if (!primal_feasible && dual_feasible) return kPrimalInfeasible;
    return kBothInfeasible;
  }

  static FB_DEV double lane_of(double x, int p) {  // x of lane p (p wavefront-uniform)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), p),
                            __builtin_amdgcn_readlane(__double2loint(x), p));
  }

  // ---- K into registers -----------------------------------------------------------
  // Kr[c] = K[t][c] for this lane's row t, dg = K[t][t].  E = H + sigma I + A'Gamma A
  // is taken from its lower triangle (what the reference assembles and Eigen's LDLT
  // reads), mirrored, so that the rows held here are exactly symmetric.
  static constexpr int tile_of(int I, int J) { return I * (I + 1) / 2 + J; }  // I >= J
  // (Lane id and sizes made opaque per call: everything below that depends on them
  // only - 64 columns' worth of masks, addresses and even the H and G entries - is
  // invariant over the whole solve, and the optimiser would otherwise hoist all of
  // it out of the Newton loop and spill it: 1000 registers.)
  // H in accumulator layout (Hd), requested by the caller a phase ahead
  FB_DEV void load_hd(int t, d4 (&hd)[10]) const {
#pragma unroll
    for (int i = 0; i < 10; i++) {
#pragma unroll
      for (int q = 0; q < 4; q++) hd[i][q] = Hd[64 * (4 * i + q) + t];
    }
  }
  FB_DEV void assemble(const C& c, const d4 (&hd)[10], double sigma, double (&Kr)[64], double* dg_out,
                       double* atr_out) const {
    int t = c.tid;
    int nz = this->nz, nl = this->nl, nv = this->nv;
    asm volatile("" : "+v"(t), "+s"(nz), "+s"(nl), "+s"(nv));
    const int kq = t >> 4, ij = t & 15;
    const int nt16 = (nz + 15) >> 4;
    FB_DW_LAPS_DECL;
    int col[4];
#pragma unroll
    for (int I = 0; I < 4; I++) col[I] = 16 * I + ij < nz ? 16 * I + ij : nz - 1;  // padded columns re-read the last one
    // the accumulators start from H + sigma I (lower triangle; what lies above the
    // diagonal or past nz is never used): lane l, register q <-> entry
    // (16 I + l/16 + 4 q, 16 J + l%16) of tile (I, J)
    d4 acc[10];
#pragma unroll
    for (int I = 0; I < 4; I++) {
#pragma unroll
      for (int J = 0; J <= I; J++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          double h = hd[tile_of(I, J)][q];
          if (I == J && kq + 4 * q == ij) h += sigma;
          acc[tile_of(I, J)][q] = h;
        }
      }
    }
    // Sixteen rows of A per trip, four k-steps.  Lane (kq, ij) holds four rows of column
    // 16 I + ij for every tile row I, and step q takes element q: A operand of tile row I =
    // A[k][16 I + ij], B operand of tile column J = the same entry of block column J times
    // Gamma_k.  Rows past nv re-read the last rows with weight zero.  The next trip's operands are
    // requested before this trip's products.  The same operands give A'(rv/mu) of the
    // right-hand side (dense_cholesky_solver.cc:98-100): lane (kq, ij) sums its rows of
    // column 16 I + ij, the four partial sums meet below.
    FB_DW_LAP(0);
    double part[4] = {0.0, 0.0, 0.0, 0.0};
    const double* acol[4];
#pragma unroll
    for (int I = 0; I < 4; I++) acol[I] = D.A + (long)col[I] * nv;
    struct Ops {
      d4 a[4];
      double g[4], rm[4];
    };
    auto load_ops = [&](int k0, Ops& o) {
      // two row pairs per lane: rows k0 + 2 kq, + 1 (the four kq of a column cover 64
      // contiguous bytes with one 16-byte access each) and the same pair eight rows on
      typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
      if (nv >= 2) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int kb = k0 + 8 * h + 2 * kq;
          int kc = kb, skip = 0;
          if (kb + 2 > nv) { kc = nv - 2; skip = kb - kc; }  // past the end: the last pair, weight zero
#pragma unroll
          for (int I = 0; I < 4; I++) {
            const dbl2 t2 = *reinterpret_cast<const d2u*>(acol[I] + kc);
            o.a[I][2 * h] = t2[0];
            o.a[I][2 * h + 1] = t2[1];
          }
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const bool mine = q >= skip;
            const double gq = gam[kc + q], rq = rvm[kc + q];
            o.g[2 * h + q] = mine ? gq : 0.0;
            o.rm[2 * h + q] = mine ? rq : 0.0;
          }
        }
      } else {  // a single row in all: one trip, element by element
        const int kb = k0 + 4 * kq;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int kk = q < nv ? q : nv - 1;
#pragma unroll
          for (int I = 0; I < 4; I++) o.a[I][q] = acol[I][kk];
          const bool mine = kb == 0 && q < nv;
          const double gq = gam[kk], rq = rvm[kk];
          o.g[q] = mine ? gq : 0.0;
          o.rm[q] = mine ? rq : 0.0;
        }
      }
    };
    auto use_ops = [&](const Ops& o) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        double bb[4];
#pragma unroll
        for (int I = 0; I < 4; I++) {
          bb[I] = o.g[q] * o.a[I][q];
          part[I] = fma(o.a[I][q], o.rm[q], part[I]);
        }
#pragma unroll
        for (int I = 0; I < 4; I++) {
          if (I < nt16) {
#pragma unroll
            for (int J = 0; J <= I; J++)
              acc[tile_of(I, J)] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[I][q], bb[J], acc[tile_of(I, J)], 0, 0, 0);
          }
        }
      }
    };
    {
      Ops oa, ob;
      load_ops(0, oa);
      for (int k0 = 0; k0 < nv; k0 += 32) {
        if (k0 + 16 < nv) load_ops(k0 + 16, ob);
        use_ops(oa);
        if (k0 + 32 < nv) load_ops(k0 + 32, oa);
        if (k0 + 16 < nv) use_ops(ob);
      }
    }
    {
      // (A' rv/mu)[t]: column t = 16 I + ij with I = t / 16 = kq of this lane
#pragma unroll
      for (int I = 0; I < 4; I++) {
        part[I] += __shfl_xor(part[I], 16, 64);
        part[I] += __shfl_xor(part[I], 32, 64);
      }
      *atr_out = kq == 0 ? part[0] : (kq == 1 ? part[1] : (kq == 2 ? part[2] : part[3]));
    }
    // D layout of a tile (R, C): lane l, register q hold E[16 R + l/16 + 4 q][16 C + l%16].
    // Column block P of all 64 rows goes through the staging panel S[row][16]:
    // tiles (R, P), R > P, as they are; the diagonal tile from its lower half, mirrored;
    // the rows above it from the transposes of tiles (P, R), R < P.
    FB_DW_LAP(1);
    lds_ptr S = stage;
    double dgE = 0.0;
#pragma unroll
    for (int P = 0; P < 4; P++) {
      if (P < nt16) {
#pragma unroll
        for (int R = 0; R < 4; R++) {
          if (R > P) {
            if (R < nt16) {
#pragma unroll
              for (int q = 0; q < 4; q++) S[(16 * R + kq + 4 * q) * LD + ij] = acc[tile_of(R, P)][q];
            }
          } else if (R == P) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int r = kq + 4 * q;
              const double e = acc[tile_of(P, P)][q];
              if (r >= ij) S[(16 * P + r) * LD + ij] = e;
              if (r > ij) S[(16 * P + ij) * LD + r] = e;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; q++) S[(16 * R + ij) * LD + kq + 4 * q] = acc[tile_of(P, R)][q];
          }
        }
        c.sync();
#pragma unroll
        for (int cc = 0; cc < 16; cc++) Kr[16 * P + cc] = S[t * LD + cc];
        if ((t >> 4) == P) dgE = S[t * LD + (t & 15)];
        c.sync();
      } else {
#pragma unroll
        for (int cc = 0; cc < 16; cc++) Kr[16 * P + cc] = 0.0;
      }
    }
    // G, G' and -sigma I round the E block (dense_cholesky_solver.cc:62-69).  One load
    // per column, all of them issued before the first is used (clamped addresses, no
    // branch in between: a branch per column made each load a round trip of its own);
    // the lane's kind selects afterwards.
    FB_DW_LAP(2);
    const int n = nz + nl;
    const bool tz = t < nz, tg = t >= nz && t < n;
    if (nl > 0) {  // (uniform)
      const int q = tg ? t - nz : 0;
      const int tt = tz ? t : 0;
      const double* Gq = D.G + q;              // row q of G (lanes that hold a row of [G -sigma I])
      const double* Gtt = Gt + tt;             // G'[t][.] (lanes that hold a row of E): Gt[64 q + t]
      auto load_g = [&](auto C0, double (&g)[16]) {
        constexpr int c0 = decltype(C0)::value;
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const int cidx = c0 + u;
          const long off_q = (long)(cidx < nz ? cidx : 0) * nl;
          const long off_t = 64 * (long)(cidx >= nz && cidx < n ? cidx - nz : 0);
          const double* pq = Gq + off_q;
          const double* pt = Gtt + off_t;
          g[u] = *(cidx < nz ? pq : pt);
        }
      };
      auto use_g = [&](auto C0, const double (&g)[16]) {
        constexpr int c0 = decltype(C0)::value;
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const int cidx = c0 + u;
          const double e = tz ? Kr[cidx] : (tg ? g[u] : 0.0);                   // cidx < nz
          const double r = tz ? g[u] : ((tg && cidx == t) ? -sigma : 0.0);       // nz <= cidx < n
          Kr[cidx] = cidx < nz ? e : (cidx < n ? r : 0.0);
        }
      };
      // (two chunks of sixteen columns in flight)
      double ga[16], gb[16];
      load_g(std::integral_constant<int, 0>{}, ga);
      load_g(std::integral_constant<int, 16>{}, gb);
      use_g(std::integral_constant<int, 0>{}, ga);
      load_g(std::integral_constant<int, 32>{}, ga);
      use_g(std::integral_constant<int, 16>{}, gb);
      load_g(std::integral_constant<int, 48>{}, gb);
      use_g(std::integral_constant<int, 32>{}, ga);
      use_g(std::integral_constant<int, 48>{}, gb);
    } else {
#pragma unroll
      for (int cidx = 0; cidx < 64; cidx++) Kr[cidx] = (tz && cidx < nz) ? Kr[cidx] : 0.0;
    }
    FB_DW_LAP(3);
    FB_DW_LAPS_FLUSH(19);
    *dg_out = tz ? dgE : (tg ? -sigma : 0.0);
  }

  // ---- pivoted LDL' on the register-held rows ---------------------------------------
  // Nothing is swapped: eliminated rows and columns drop out of the pivot search, the
  // elimination order is a list (lane k keeps perm[k], each row the step `ord` at
  // which it went and its pivot).  A step: pivot search; every lane picks ITS entry
  // of column p out of its own registers (= its entry of the pivot row, by symmetry)
  // through a tree of scalar branches - the registers cannot be indexed at run time,
  // and the alternative, the pivot lane alone writing its 64 values to LDS, keeps the
  // LDS busy for 32 instructions that seven other wavefronts wait behind; the column
  // goes to LDS with ONE store and comes back as the pivot row, kLdsCols columns as
  // 16-byte broadcasts and the rest through v_readlane (the LDS and the vector pipe
  // share the work).  Returns false where Eigen reports failure.
  // out = a[p], p wavefront-uniform.  Real branches: the empty asm keeps the
  // optimiser from turning the tree into selects of loads, i.e. into a copy of the
  // array in scratch memory that every update then has to write through.
  template <int LO, int HI>
  static FB_DEV void pick(const double (&a)[64], int p, double& out) {
    if constexpr (HI - LO == 1) {
      out = a[LO];
      asm volatile("" : "+v"(out));
    } else {
      constexpr int MID = (LO + HI) / 2;
      if (p < MID) pick<LO, MID>(a, p, out);
      else pick<MID, HI>(a, p, out);
    }
  }
  // sum over the wavefront, the same bits in every lane
  static FB_DEV double wave_sum(double v) {
    v += dpp_mov<0x128>(v);  // row_ror:8
    v += dpp_mov<0x124>(v);  // row_ror:4
    v += dpp_mov<0x122>(v);  // row_ror:2
    v += dpp_mov<0x121>(v);  // row_ror:1
    const double r0 = lane_of(v, 0), r1 = lane_of(v, 16), r2 = lane_of(v, 32), r3 = lane_of(v, 48);
    return (r0 + r1) + (r2 + r3);
  }
  // largest value over the wavefront, the same bits in every lane
  static FB_DEV double wave_max(double v) {
    v = fmax(v, dpp_mov<0x128>(v));  // row_ror:8
    v = fmax(v, dpp_mov<0x124>(v));  // row_ror:4
    v = fmax(v, dpp_mov<0x122>(v));  // row_ror:2
    v = fmax(v, dpp_mov<0x121>(v));  // row_ror:1
    const double r0 = lane_of(v, 0), r1 = lane_of(v, 16), r2 = lane_of(v, 32), r3 = lane_of(v, 48);
    return fmax(fmax(r0, r1), fmax(r2, r3));
  }
#ifndef FB_DW_W_FROM_THE_SYSTEM
#define FB_DW_W_FROM_THE_SYSTEM 1
#endif
#ifndef FB_DW_LDS_COLS
#define FB_DW_LDS_COLS 48
#endif
  static constexpr int kLdsCols = FB_DW_LDS_COLS;  // even
#ifndef FB_DW_LDS_BATCH
#define FB_DW_LDS_BATCH 24
#endif
  static constexpr int kLdsBatch = FB_DW_LDS_BATCH;  // columns per batch of LDS reads (even)
  FB_DEV bool factor(const C& c, double (&Kr)[64], double dg, int* ord_o, double* dpiv_o, int* permv_o) const {
    int n = lay.nk, t = c.tid;
    asm volatile("" : "+s"(n), "+v"(t));
    bool alive = t < n;
    int ord = 64, permv = 0;
    double dpiv = 0.0;
    bool found_zero_pivot = false;
    // A NaN on the diagonal (an iterate that overflowed: the matrix is NaN wherever a NaN
    // Gamma reaches) ends Eigen's factorisation - its pivot search compares false against
    // it, the pivot it then takes is invalid over a non-zero column: NumericalIssue, and the
    // reference throws (impl:263-267).  Below, NaN marks the rows that are gone.
    if (__ballot(alive && dg != dg) != 0ull) return false;
    FB_DW_LAPS_DECL;
    for (int k = 0; k < n; k++) {
      // largest |diagonal| of what is left; the first maximum wins (Eigen's maxCoeff)
      // (rows that are gone carry a NaN: v_max_f64 passes over it, it equals nothing,
      // and all-ones is an inline constant - a -1.0 lived in a register the allocator
      // spilled, and its reload in here waited for the multiplier store of the step
      // before with every other vector memory operation)
      const double mag = __hiloint2double(alive ? (__double2hiint(dg) & 0x7fffffff) : -1, alive ? __double2loint(dg) : -1);
      const double best = wave_max(mag);
      const unsigned long long hit = __ballot(mag == best);
      const int p = __builtin_ctzll(hit);
      const double d = lane_of(dg, p);
      if (t == k) permv = p;
      const bool valid = fabs(d) > 0.0;
      if (found_zero_pivot && valid) return false;
      if (!valid) found_zero_pivot = true;
      FB_DW_LAP(0);
      double colp;  // K[t][p] = K[p][t]
      pick<0, 64>(Kr, p, colp);
      FB_DW_LAP(1);
      if (t == p) {
        alive = false;
        ord = k;
        dpiv = d;
      }
      rowbuf[t] = colp;
      const double lm = (alive && valid) ? colp * (1.0 / d) : 0.0;
      Lg[64 * k + t] = lm;
      c.sync();
      FB_DW_LAP(2);
      if (valid) {
        const double nl_ = -lm;
        // the row's LDS part is requested kLdsBatch columns at a time; the v_readlane
        // columns run while the first batch is on its way
        constexpr int NB = kLdsBatch / 2;
        dbl2 sj[NB];
#pragma unroll
        for (int u = 0; u < NB; u++)
          if (2 * u < kLdsCols) sj[u] = *reinterpret_cast<FB_LDS const dbl2*>(rowbuf + 2 * u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = kLdsCols; j < 64; j++) Kr[j] = fma(nl_, lane_of(colp, j), Kr[j]);
        dg = fma(nl_, colp, dg);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j0 = 0; j0 < kLdsCols; j0 += kLdsBatch) {
#pragma unroll
          for (int u = 0; u < NB; u++) {
            if (j0 + 2 * u < kLdsCols) {
              Kr[j0 + 2 * u] = fma(nl_, sj[u][0], Kr[j0 + 2 * u]);
              Kr[j0 + 2 * u + 1] = fma(nl_, sj[u][1], Kr[j0 + 2 * u + 1]);
            }
          }
          if (j0 + kLdsBatch < kLdsCols) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NB; u++)
              if (j0 + kLdsBatch + 2 * u < kLdsCols)
                sj[u] = *reinterpret_cast<FB_LDS const dbl2*>(rowbuf + j0 + kLdsBatch + 2 * u);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      c.sync();  // (the next column is stored behind these reads)
      FB_DW_LAP(3);
    }
    FB_DW_LAPS_FLUSH(0);
    *ord_o = ord;
    *dpiv_o = dpiv;
    *permv_o = permv;
    global_fence();  // the multipliers are read back by other lanes
    return true;
  }

  // ---- LDL' in the natural order, fused with both substitutions (round 3) -------------
  // K is quasi-definite for sigma > 0: the leading block H + sigma I + A'Gamma A is
  // positive definite, the Schur complement behind it negative definite, so LDL' exists
  // for EVERY symmetric permutation and the natural order is the block elimination an
  // implementation without a pivoting library would write down (Cholesky of the leading
  // block, then of sigma I + G E^-1 G').  Eigen's rule (largest remaining |diagonal|,
  // dense_cholesky_solver.cc:70-79) eliminates the same leading block first - the
  // diagonal of the trailing one is -sigma - in a different order inside it; the two
  // factorisations solve the same system to rounding.  With the order known at compile
  // time the whole step list unrolls and every register index is an immediate:
  //   * no pivot search, no pick of a run-time column out of the registers;
  //   * step k updates columns k + 1 .. 63 only (the pivoted loop has to sweep all 64);
  //   * the multipliers are used where they are formed: the right-hand side is
  //     eliminated along with the matrix (Gaussian elimination of the augmented column),
  //     nothing goes to the global scratch and the forward sweep with its 63 loads is gone;
  //   * what a finished step leaves in lane k - row k of D L' in registers k + 1 .. 63,
  //     frozen from then on - is what the backward sweep needs: lane t takes its own dot
  //     product, one v_readlane pair and one FMA per solved entry, instead of a sum over
  //     the wavefront per entry.
  // The pivot row reaches the other lanes as in the pivoted loop: the column (= the row,
  // by symmetry) goes to LDS with one store, the first kRlCols columns behind the pivot
  // come back through v_readlane - column k + 1 first, the next pivot and its reciprocal
  // hang on it -, the rest as 16-byte LDS broadcasts.  Rows and columns n .. 63 are an
  // identity block (their pivots read as 1).
  // Returns false - and leaves K and x destroyed - if a pivot is zero, denormal, infinite
  // or NaN: the caller assembles K again and takes the pivoted path, whose verdict on such
  // a matrix is Eigen's.
#ifndef FB_DW_STATIC_ORDER
#define FB_DW_STATIC_ORDER 1
#endif
#ifndef FB_DW_RL_COLS
#define FB_DW_RL_COLS 8
#endif
  static constexpr int kRlCols = FB_DW_RL_COLS;
#ifndef FB_DW_ST_BATCH
#define FB_DW_ST_BATCH 8
#endif
#ifndef FB_DW_RCP_STEPS
#define FB_DW_RCP_STEPS 3
#endif
  static constexpr int kStBatch = FB_DW_ST_BATCH;  // 16-byte LDS reads in flight
  // 1 / d: hardware seed and three Newton steps, without the scaling and fix-up of the
  // IEEE sequence (the exponent of d is checked by the caller).  Two steps are what the
  // IEEE sequence takes before its final correction of the quotient and gave the same
  // bits on every test; the third is insurance against a seed at the low end of its
  // specification and costs nothing measurable (gpurun_out/r03_z).
  static FB_DEV double rcp_nr(double d) {
    double r = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int i = 0; i < FB_DW_RCP_STEPS; i++) r = fma(fma(-d, r, 1.0), r, r);
    return r;
  }
  FB_DEV bool factor_solve_static(const C& c, double (&Kr)[64], double& x) const {
    int n = lay.nk, t = c.tid;
    asm volatile("" : "+s"(n), "+v"(t));
    unsigned emin = 0x7ffu, emax = 0u;
    double dinv = 0.0;
    auto pivot = [&](auto K_, double& r) {  // reciprocal of pivot k, its exponent noted
      constexpr int k = decltype(K_)::value;
      const double dk = lane_of(Kr[k], k);
      const double d = k < n ? dk : 1.0;
      const unsigned e = ((unsigned)__double2hiint(d) >> 20) & 0x7ffu;
      emin = e < emin ? e : emin;  // (identity padding counts as a pivot of one; a predicate on
      emax = e > emax ? e : emax;  //  k < n here cost the kernel 770 more spilled registers)
      r = rcp_nr(d);
    };
    double r;
    pivot(std::integral_constant<int, 0>{}, r);
    FB_DW_LAPS_DECL;
    auto step = [&](auto K_) {
      constexpr int k = decltype(K_)::value;
      const double colk = Kr[k];  // K[t][k] = K[k][t]
      const double nlm = t > k ? -(colk * r) : 0.0;
      dinv = t == k ? r : dinv;
      rowbuf[t] = colk;
      x = fma(nlm, lane_of(x, k), x);
      Kr[k + 1] = fma(nlm, lane_of(colk, k + 1), Kr[k + 1]);
      pivot(std::integral_constant<int, k + 1>{}, r);
      // first column that comes from LDS: even, at least kRlCols behind the pivot
      constexpr int jl = (k + 1 + kRlCols + 1) & ~1;
      constexpr int jA = jl < 64 ? jl : 64;
      sfor<k + 2, jA>([&](auto J_) {
        constexpr int j = decltype(J_)::value;
        Kr[j] = fma(nlm, lane_of(colk, j), Kr[j]);
      });
      if constexpr (jA < 64) {
        __builtin_amdgcn_wave_barrier();  // (LDS keeps a wavefront's accesses in order)
        constexpr int NP = (64 - jA) / 2;  // pairs
        dbl2 sj[kStBatch];
        auto request = [&](auto P0) {
          constexpr int p0 = decltype(P0)::value;
          sfor<0, kStBatch>([&](auto U_) {
            constexpr int u = decltype(U_)::value;
            if constexpr (p0 + u < NP) sj[u] = *reinterpret_cast<FB_LDS const dbl2*>(rowbuf + jA + 2 * (p0 + u));
          });
        };
        request(std::integral_constant<int, 0>{});
        sfor<0, (NP + kStBatch - 1) / kStBatch>([&](auto B_) {
          constexpr int p0 = decltype(B_)::value * kStBatch;
          dbl2 cur[kStBatch];
          sfor<0, kStBatch>([&](auto U_) { cur[decltype(U_)::value] = sj[decltype(U_)::value]; });
          if constexpr (p0 + kStBatch < NP) request(std::integral_constant<int, p0 + kStBatch>{});
          sfor<0, kStBatch>([&](auto U_) {
            constexpr int u = decltype(U_)::value;
            if constexpr (p0 + u < NP) {
              constexpr int j = jA + 2 * (p0 + u);
              Kr[j] = fma(nlm, cur[u][0], Kr[j]);
              Kr[j + 1] = fma(nlm, cur[u][1], Kr[j + 1]);
            }
          });
        });
        __builtin_amdgcn_wave_barrier();  // (the next column is stored behind these reads)
      }
    };
    // (sixteen steps at a time: a block whose rows are all identity padding is skipped -
    // small problems do not pay for sixty-three steps)
    sfor<0, 16>(step);
    if (n > 16) {
      sfor<16, 32>(step);
      if (n > 32) {
        sfor<32, 48>(step);
        if (n > 48) sfor<48, 63>(step);
      }
    }
    dinv = t == 63 ? r : dinv;
    FB_DW_LAP(3);
    // A pivot that is zero, denormal, infinite or NaN; or (order chosen per step) pivots that
    // span more than spread_bits binary orders of magnitude: K is then so ill-conditioned
    // that the iteration's course depends on HOW the rounding errors of the solve fall, and
    // the caller takes the reference's elimination order.
    if (emin == 0u || emax == 0x7ffu || (lay.order == 0 && (int)(emax - emin) > lay.spread_bits)) {
      FB_DW_LAPS_FLUSH(0);
      return false;
    }
    // D L' w = y: lane t holds row t of D L' (columns t + 1 ..) and 1 / d_t
    x *= dinv;
    auto back = [&](auto J_) {
      constexpr int j = 63 - decltype(J_)::value;
      const double u = t < j ? Kr[j] * dinv : 0.0;
      x = fma(-u, lane_of(x, j), x);
    };
    if (n > 48) sfor<0, 16>(back);   // j = 63 .. 48 (columns of identity padding are zero in the rows above them)
    if (n > 32) sfor<16, 32>(back);  // j = 47 .. 32
    if (n > 16) sfor<32, 48>(back);  // j = 31 .. 16
    sfor<48, 63>(back);              // j = 15 .. 1
    FB_DW_LAP(1);
    FB_DW_LAPS_FLUSH(0);
    return true;
  }

  // x <- K^{-1} x with the factors above (P' L^{-T} D^{+} L^{-1} P of
  // dense_cholesky_solver.cc:112 with the permutation implicit in the elimination
  // order).  Lane t owns entry t.  A multiplier is zero wherever its row was no
  // longer (or not yet) part of the step, so neither sweep needs a predicate.
  FB_DEV double substitute(int t, double x, int ord, double dpiv, int permv) const {
    // (t arrives opaque: the 126 load addresses are invariant over the whole solve, and
    // the optimiser would otherwise form them once per QP and keep them in scratch.)
    // Both sweeps are chains of dependent steps whose multipliers do not depend on the
    // chain: the loads go out before the first step (the matrix registers are free by
    // now).  Steps past the problem's n run with zero multipliers (rows n.. of the
    // scratch are zero from its allocation on; lanes n.. hold perm = 0).
    const int n = lay.nk;
    double lk[64];
    FB_DW_LAPS_DECL;
    // L y = b: step k hands the entry of the row eliminated at step k to all later rows
#pragma unroll
    for (int k = 0; k < 63; k++) lk[k] = Lg[64 * k + t];
    FB_DW_LAP(0);
#pragma unroll
    for (int k = 0; k < 63; k++) {
      const int p = __builtin_amdgcn_readlane(permv, k);
      x = fma(-lk[k], lane_of(x, p), x);
    }
    FB_DW_LAP(1);
    x = (t < n && fabs(dpiv) > DBL_MIN) ? x / dpiv : 0.0;  // pseudo-inverse of D
    // L' w = y, one row of L' per step: w(p_j) = y(p_j) - sum_t Lg[j][t] w(t) over the
    // rows t eliminated after step j (the multipliers of all others are zero) - a sum
    // over the wavefront per step, but row j of the multipliers is what lane t already
    // read for the forward sweep.  (The column form - the solved entry handed to the
    // rows eliminated before it - needs Lg[ord(t)][p_k], a different cache line in
    // every lane: those 63 loads took five times as long as the rest of the solve.)
    FB_DW_LAP(2);
#pragma unroll
    for (int j = 62; j >= 0; j--) {
      const double s = wave_sum(lk[j] * x);
      const int p = __builtin_amdgcn_readlane(permv, j);
      x = t == p ? x - s : x;
    }
    FB_DW_LAP(3);
    FB_DW_LAPS_FLUSH(4);
    return x;
  }

  FB_DEV bool newton_step(const C& c, double sigma, double alpha) const {
    FB_WAVE_LAP_DECL;
    int t = c.tid;
    asm volatile("" : "+v"(t));  // (see assemble)
    const int n = lay.nk;
    d4 hd[10];
    load_hd(t, hd);  // (on its way while the gradients are formed)
    // PFB gradients (dense_cholesky_solver.cc:54-61)
    int nact = 0;  // rows with Gamma sigma > 2^-act_bits (wavefront-uniform)
    const double act_thr = __hiloint2double((1023 - lay.act_bits) << 20, 0);
    for (int i0 = 0; i0 < nv; i0 += 64) {
      const int i = i0 + t;
      bool act = false;
      if (i < nv) {
        const double ys = y[i] + sigma * (v[i] - vb[i]);
        double g0, g1;
        pfb_gradient(ys, v[i], alpha, &g0, &g1);
        const double mu = g1 + sigma * g0;
        const double g = g0 / mu;
        gam[i] = g;
        rvm[i] = -pfb(ys, v[i], alpha) / mu;
        act = g * sigma > act_thr;
      }
      nact += __popcll(__ballot(act));
    }
    if (lay.order == 0 && lay.act_bits > 0 && nact + nl >= nz) went_pivoted = true;
    c.sync();
    FB_WAVE_LAP(10);
    double Kr[64], dg, atr;
    assemble(c, hd, sigma, Kr, &dg, &atr);
    // eliminated right-hand side (dense_cholesky_solver.cc:98-104), entry t in lane t
    double x = 0.0;
    double rhs0 = 0.0;  // this row's right-hand side of the full Newton system (before the elimination of dv)
    if (t < nz) { rhs0 = -(rz[t] + sigma * (z[t] - zb[t])); x = rhs0 - atr; }
    else if (t < n) { rhs0 = rl[t - nz] + sigma * (l[t - nz] - lb[t - nz]); x = rhs0; }
    FB_WAVE_LAP(11);
#if FB_DW_STATIC_ORDER
    bool solved = false;
    {
      // (a NaN on the diagonal - an overflowed iterate - goes straight to the pivoted
      // path, which answers it the way Eigen does)
      const double x0 = x;
      const bool take_static = lay.order != 1 && !(lay.order == 0 && lay.sticky != 0 && went_pivoted) &&
                               __ballot(t < n && dg != dg) == 0ull;
      if (take_static) solved = factor_solve_static(c, Kr, x);
      if (!take_static && went_pivoted && nfallback != nullptr && t == 0) atomicAdd(nfallback + 1, 1);
      if (take_static && !solved) {
        went_pivoted = true;
        if (nfallback != nullptr && t == 0) atomicAdd(nfallback, 1);
        x = x0;
        load_hd(t, hd);
        assemble(c, hd, sigma, Kr, &dg, &atr);
      }
    }
    FB_WAVE_LAP(12);
    if (!solved)
#endif
    {
      int ord, permv;
      double dpiv;
      if (!factor(c, Kr, dg, &ord, &dpiv, &permv)) return false;
      FB_WAVE_LAP(12);
      x = substitute(t, x, ord, dpiv, permv);
    }
    FB_WAVE_LAP(13);
    if (t < nz) dz[t] = x;
    else if (t < n) dl[t - nz] = x;
#if FB_DW_W_FROM_THE_SYSTEM
    // W = (H dz + G'dl + A'dv, -G dz), the increment of the natural residual's (z, l)
    // rows along the step (the line search evaluates its trial points from it), is what
    // the rows of the Newton system just solved leave of their right-hand sides:
    //   (H + sigma I) dz + G'dl + A'dv = rhs_z,   G dz - sigma dl = rhs_l
    // - no pass over H, G' and A' (160 loads per lane and Newton step), at the price of
    // the solve's own residual in W, which is of the size of the rounding error of the
    // products it replaces.
    if (t < nz) wz[t] = rhs0 - sigma * x;
    else if (t < n) wl[t - nz] = -(rhs0 + sigma * x);
#endif
    c.sync();
    // dv = rv/mus + Gamma .* (A dz) (:114-121); adz = A dz (dy = b - A dz, :124): rows t
    // and t + 64 of A against dz (and, for the explicit W, row t of H or of G in the same
    // loop), the loads of ten entries in flight together.  Rows past the end read row 0
    // and are dropped.
    [[maybe_unused]] double hdz = 0.0;  // (H dz)_t, t < nz; (G dz)_(t - nz), nz <= t < n
    if (nv <= 128) {
      const bool r0 = t < nv, r1 = t + 64 < nv;
      const double* a0 = D.A + (r0 ? t : 0);
      const double* a1 = D.A + (r1 ? t + 64 : 0);
      [[maybe_unused]] const double* hr = t < nz ? D.H + t : (nl > 0 ? D.G + (t < n ? t - nz : 0) : D.H);
      [[maybe_unused]] const long hs = t < nz ? nz : (nl > 0 ? nl : nz);
      double s0 = 0.0, s1 = 0.0;
      struct Chunk { double x0[10], x1[10], xh[10]; };
      auto load_chunk = [&](int k0, Chunk& ch) {
#pragma unroll
        for (int u = 0; u < 10; u++) {
          const int k = k0 + u < nz ? k0 + u : nz - 1;
          ch.x0[u] = a0[(long)k * nv];
          ch.x1[u] = a1[(long)k * nv];
#if !FB_DW_W_FROM_THE_SYSTEM
          ch.xh[u] = hr[(long)k * hs];
#endif
        }
      };
      auto use_chunk = [&](int k0, const Chunk& ch) {
#pragma unroll
        for (int u = 0; u < 10; u++) {
          const double d = k0 + u < nz ? dz[k0 + u < nz ? k0 + u : nz - 1] : 0.0;
          s0 = fma(ch.x0[u], d, s0);
          s1 = fma(ch.x1[u], d, s1);
#if !FB_DW_W_FROM_THE_SYSTEM
          hdz = fma(ch.xh[u], d, hdz);
#endif
        }
      };
      // (the next ten entries' thirty loads go out before these ten are used)
      Chunk ca, cb;
      load_chunk(0, ca);
      for (int k0 = 0; k0 < nz; k0 += 20) {
        if (k0 + 10 < nz) load_chunk(k0 + 10, cb);
        use_chunk(k0, ca);
        if (k0 + 20 < nz) load_chunk(k0 + 20, ca);
        if (k0 + 10 < nz) use_chunk(k0 + 10, cb);
      }
      if (r0) { adz[t] = s0; dv[t] = rvm[t] + gam[t] * s0; }
      if (r1) { adz[t + 64] = s1; dv[t + 64] = rvm[t + 64] + gam[t + 64] * s1; }
    } else {
      for (int i = t; i < nv; i += 64) {
        const double a = A_row_dot(i, dz);
        adz[i] = a;
        dv[i] = rvm[i] + gam[i] * a;
      }
#if !FB_DW_W_FROM_THE_SYSTEM
      if (t < nz) hdz = row_dot(D.H, nz, nz, t, dz);
      else if (t < n) hdz = row_dot(D.G, nl, nz, t - nz, dz);
#endif
    }
    c.sync();
#if !FB_DW_W_FROM_THE_SYSTEM
    if (t < nz) {
      double gdl = 0.0;  // (G'dl)_t from the transposed copy
      for (int q = 0; q < nl; q++) gdl = fma(Gt[64 * q + t], dl[q], gdl);
      wz[t] = hdz + gdl + A_col_dot(t, dv);
    } else if (t < n) {
      wl[t - nz] = -hdz;
    }
    c.sync();
#endif
    FB_WAVE_LAP(14);
    return true;
  }

  FB_DEV void write_x(const C& c) const {
    for (int i = c.tid; i < nz; i += 64) uz[i] = z[i];
    for (int i = c.tid; i < nl; i += 64) ul[i] = l[i];
    for (int i = c.tid; i < nv; i += 64) { uv[i] = v[i]; uy[i] = y[i]; }
  }
  FB_DEV void write_xbar(const C& c) const {
    for (int i = c.tid; i < nz; i += 64) uz[i] = zb[i];
    for (int i = c.tid; i < nl; i += 64) ul[i] = lb[i];
    for (int i = c.tid; i < nv; i += 64) { uv[i] = vb[i]; uy[i] = yb[i]; }
  }
  FB_DEV void write_certificate(const C& c) const {
    for (int i = c.tid; i < nz; i += 64) uz[i] = dz[i];
    for (int i = c.tid; i < nl; i += 64) ul[i] = dl[i];
    for (int i = c.tid; i < nv; i += 64) {
      uv[i] = dv[i];
      uy[i] = (y[i] - yb[i]) + bvec(i);
    }
  }
};


}  // namespace fbk
