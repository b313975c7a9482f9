// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the
// shipped product; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may build, load or call it (as the checker / the timed CPU
// baseline, never as the GPU product path).
//
// Minimal column-major dense linear algebra used by the CPU restatement of
// dliaomcp/fbstab.  The reference delegates these to Eigen 3.4.0 (pinned at
// tools/eigen/repository.bzl:8-9, not vendored, absent from this image), so the
// Eigen routines the hot path calls are restated here from Eigen 3.4's
// published behaviour:
//   * Eigen::LLT  (unblocked in-place lower Cholesky, reads the lower triangle
//     only, fails iff a pivot is <= 0)            -> llt_inplace_lower()
//     call sites: riccati_linear_solver.cc:146,165,182,193,205
//   * Eigen::LDLT (in-place lower LDL' with symmetric pivoting on the largest
//     |diagonal| entry, pseudo-inverse of D in solve) -> Ldlt
//     call sites: dense_cholesky_solver.cc:72,112
//   * triangularView<Lower>().solveInPlace (left / OnTheRight, plain /
//     transposed)                                   -> trsv_* / trsm_right_lt()
#pragma once

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <limits>
#include <vector>

namespace fbo {

typedef std::vector<double> Vec;

// Column-major element access helper: A(i,j) with leading dimension ld.
#define FBO_AT(A, ld, i, j) ((A)[(size_t)(i) + (size_t)(j) * (size_t)(ld)])

inline double dot(const double* a, const double* b, int n) {
  double s = 0.0;
  for (int i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}

inline double norm2(const double* a, int n) { return std::sqrt(dot(a, a, n)); }
inline double norm2(const Vec& a) { return norm2(a.data(), (int)a.size()); }

inline double inf_norm(const Vec& a) {
  double s = 0.0;
  for (size_t i = 0; i < a.size(); i++) s = std::max(s, std::fabs(a[i]));
  return s;
}

// y += a * M * x, M is m x n column-major.
inline void gemv_n(const double* M, int m, int n, double a, const double* x,
                   double* y) {
  for (int j = 0; j < n; j++) {
    const double ax = a * x[j];
    const double* col = M + (size_t)j * m;
    for (int i = 0; i < m; i++) y[i] += col[i] * ax;
  }
}

// y += a * M' * x, M is m x n column-major (y has n entries).
inline void gemv_t(const double* M, int m, int n, double a, const double* x,
                   double* y) {
  for (int j = 0; j < n; j++) {
    y[j] += a * dot(M + (size_t)j * m, x, m);
  }
}

// In-place lower Cholesky, unblocked (Eigen llt_inplace<Lower>::unblocked).
// Reads and writes only the lower triangle.  Returns -1 on success or the
// index of the first non-positive pivot.
inline int llt_inplace_lower(double* A, int n, int ld) {
  for (int k = 0; k < n; k++) {
    const int rs = n - k - 1;
    double x = FBO_AT(A, ld, k, k);
    for (int j = 0; j < k; j++) x -= FBO_AT(A, ld, k, j) * FBO_AT(A, ld, k, j);
    if (!(x > 0.0)) return k;
    x = std::sqrt(x);
    FBO_AT(A, ld, k, k) = x;
    if (k > 0 && rs > 0) {
      for (int j = 0; j < k; j++) {
        const double akj = FBO_AT(A, ld, k, j);
        for (int i = k + 1; i < n; i++)
          FBO_AT(A, ld, i, k) -= FBO_AT(A, ld, i, j) * akj;
      }
    }
    if (rs > 0) {
      for (int i = k + 1; i < n; i++) FBO_AT(A, ld, i, k) /= x;
    }
  }
  return -1;
}

// Solve L x = b in place (L lower, non-unit diagonal).
inline void trsv_l(const double* L, int n, int ld, double* x) {
  for (int i = 0; i < n; i++) {
    double s = x[i];
    for (int j = 0; j < i; j++) s -= FBO_AT(L, ld, i, j) * x[j];
    x[i] = s / FBO_AT(L, ld, i, i);
  }
}

// Solve L' x = b in place.
inline void trsv_lt(const double* L, int n, int ld, double* x) {
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int j = i + 1; j < n; j++) s -= FBO_AT(L, ld, j, i) * x[j];
    x[i] = s / FBO_AT(L, ld, i, i);
  }
}

// Solve L X = B in place for an n x m right-hand side.
inline void trsm_left_l(const double* L, int n, int ld, double* B, int m,
                        int ldb) {
  for (int c = 0; c < m; c++) trsv_l(L, n, ld, B + (size_t)c * ldb);
}
inline void trsm_left_lt(const double* L, int n, int ld, double* B, int m,
                         int ldb) {
  for (int c = 0; c < m; c++) trsv_lt(L, n, ld, B + (size_t)c * ldb);
}

// Solve X L' = B in place, B is m x n (triangularView<Lower>().transpose()
// .solveInPlace<OnTheRight>).  Row r of X solves L x_r' = b_r'.
inline void trsm_right_lt(const double* L, int n, int ld, double* B, int m,
                          int ldb) {
  for (int c = 0; c < n; c++) {
    for (int k = 0; k < c; k++) {
      const double lck = FBO_AT(L, ld, c, k);
      for (int r = 0; r < m; r++)
        FBO_AT(B, ldb, r, c) -= FBO_AT(B, ldb, r, k) * lck;
    }
    const double d = FBO_AT(L, ld, c, c);
    for (int r = 0; r < m; r++) FBO_AT(B, ldb, r, c) /= d;
  }
}

// Restatement of Eigen::LDLT<MatrixXd, Lower>: compute() + solve().
struct Ldlt {
  int n = 0;
  std::vector<double> m;    // packed n x n, lower holds L (unit) and D on diag
  std::vector<int> transp;  // transpositions
  std::vector<double> temp;
  bool ok = false;

  explicit Ldlt(int n_ = 0) { resize(n_); }
  void resize(int n_) {
    n = n_;
    m.assign((size_t)n * n, 0.0);
    transp.assign(n, 0);
    temp.assign(n, 0.0);
  }

  // A is n x n column-major; only its lower triangle is referenced.
  bool compute(const double* A) {
    std::copy(A, A + (size_t)n * n, m.begin());
    ok = unblocked();
#ifdef FBO_LDLT_OBSERVER  // studies of the elimination order (tools/cpp/ldlt_order_observer.h); never in liboracle.so
    FBO_LDLT_OBSERVER(*this);
#endif
    return ok;
  }

  bool unblocked() {
    double* mat = m.data();
    const int size = n;
    bool found_zero_pivot = false;
    bool ret = true;
    if (size <= 1) {
      for (int i = 0; i < size; i++) transp[i] = i;
      return true;
    }
    for (int k = 0; k < size; k++) {
      // Largest |diagonal| entry in the trailing corner (first maximum wins).
      int big = k;
      double best = std::fabs(FBO_AT(mat, n, k, k));
      for (int i = k + 1; i < size; i++) {
        const double a = std::fabs(FBO_AT(mat, n, i, i));
        if (a > best) {
          best = a;
          big = i;
        }
      }
      transp[k] = big;
      if (k != big) {
        const int s = size - big - 1;
        for (int j = 0; j < k; j++)
          std::swap(FBO_AT(mat, n, k, j), FBO_AT(mat, n, big, j));
        for (int i = 0; i < s; i++)
          std::swap(FBO_AT(mat, n, big + 1 + i, k),
                    FBO_AT(mat, n, big + 1 + i, big));
        std::swap(FBO_AT(mat, n, k, k), FBO_AT(mat, n, big, big));
        for (int i = k + 1; i < big; i++) {
          const double tmp = FBO_AT(mat, n, i, k);
          FBO_AT(mat, n, i, k) = FBO_AT(mat, n, big, i);
          FBO_AT(mat, n, big, i) = tmp;
        }
      }
      const int rs = size - k - 1;
      if (k > 0) {
        for (int j = 0; j < k; j++)
          temp[j] = FBO_AT(mat, n, j, j) * FBO_AT(mat, n, k, j);
        double s = 0.0;
        for (int j = 0; j < k; j++) s += FBO_AT(mat, n, k, j) * temp[j];
        FBO_AT(mat, n, k, k) -= s;
        if (rs > 0) {
          for (int j = 0; j < k; j++) {
            const double tj = temp[j];
            for (int i = k + 1; i < size; i++)
              FBO_AT(mat, n, i, k) -= FBO_AT(mat, n, i, j) * tj;
          }
        }
      }
      const double akk = FBO_AT(mat, n, k, k);
      const bool pivot_is_valid = std::fabs(akk) > 0.0;
      if (k == 0 && !pivot_is_valid) {
        for (int j = 0; j < size; j++) {
          transp[j] = j;
          for (int i = j + 1; i < size; i++)
            ret = ret && (FBO_AT(mat, n, i, j) == 0.0);
        }
        return ret;
      }
      if (rs > 0 && pivot_is_valid) {
        for (int i = k + 1; i < size; i++) FBO_AT(mat, n, i, k) /= akk;
      } else if (rs > 0) {
        for (int i = k + 1; i < size; i++)
          ret = ret && (FBO_AT(mat, n, i, k) == 0.0);
      }
      if (found_zero_pivot && pivot_is_valid)
        ret = false;
      else if (!pivot_is_valid)
        found_zero_pivot = true;
    }
    return ret;
  }

  // x <- A^{-1} x using P' L^{-T} D^{+} L^{-1} P.
  void solve_inplace(double* x) const {
    const double* mat = m.data();
    for (int k = 0; k < n; k++)
      if (transp[k] != k) std::swap(x[k], x[transp[k]]);
    for (int i = 0; i < n; i++) {  // unit lower solve
      double s = x[i];
      for (int j = 0; j < i; j++) s -= FBO_AT(mat, n, i, j) * x[j];
      x[i] = s;
    }
    const double tol = (std::numeric_limits<double>::min)();
    for (int i = 0; i < n; i++) {
      const double d = FBO_AT(mat, n, i, i);
      if (std::fabs(d) > tol)
        x[i] /= d;
      else
        x[i] = 0.0;
    }
    for (int i = n - 1; i >= 0; i--) {  // unit upper solve (L')
      double s = x[i];
      for (int j = i + 1; j < n; j++) s -= FBO_AT(mat, n, j, i) * x[j];
      x[i] = s;
    }
    for (int k = n - 1; k >= 0; k--)
      if (transp[k] != k) std::swap(x[k], x[transp[k]]);
  }
};

}  // namespace fbo
