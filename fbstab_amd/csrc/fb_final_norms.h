// The three component norms of the summary block the reference prints at
// Display::FINAL - its DEFAULT display level - (PrintFinal,
// fbstab_algorithm-impl.h:493-541: |rz| |rl| |rv| of the penalised natural residual
// rk_, full_residual.cc:99-109, and the stopping tolerance
// abs_tol + rel_tol (1 + ||(f, h, b)||), impl:137), evaluated at the point a batch
// solve has just returned: one wavefront per QP, straight from the caller's arrays in
// the reference's layouts (tools/matrix_sequence.h:81-83), whatever kernel solved the
// batch.  SolverOut carries the norm of the three blocks only; with this pass the
// default-constructed FBstabMpc / FBstabDense of the facade run the batch kernels.
//
// Exits whose printed residual belongs to a point the solve does not return (the
// infeasibility certificates, impl:204-212: rk_ is that of x(k); the proximal
// iteration limit, impl:219-223: rk_ is one iteration stale) are the caller's business
// (include/fbstab/: the traced solve).
#pragma once

#include "../../include/fbstab_hip.h"
#include "fb_common.h"

namespace fbk {


FB_DEV double wave_sum(double x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
  return x;
}

struct MpcNormArgs {
  const double* base[12];  // Q R S q r A B c E L d x0 (FBSTAB_MPC_*)
  long long stride[12];
  const double* x[4];      // z l v y
  long long xstride[4];
  int N, nx, nu, nc;
};

// norms[4 q ..] = {|rz|, |rl|, |rv|, tolerance}; one 64-thread workgroup per QP.
// rz = H z + f + G'l + A'v, rl = h - G z, rv = alpha min(y, v) + (1 - alpha) max(0, y) max(0, v)
// with the implicit H, G, A, f, h, b of mpc_data.cc:17-289.
__global__ __launch_bounds__(64) void fbstab_mpc_final_norms_kernel(MpcNormArgs a, fbstab_options_t opts,
                                                                    double* norms, int batch) {
  const long q = blockIdx.x;
  if (q >= batch) return;
  const int t = threadIdx.x, N = a.N, nx = a.nx, nu = a.nu, nc = a.nc, ns = nx + nu;
  const double *Q = a.base[FBSTAB_MPC_Q] + q * a.stride[FBSTAB_MPC_Q], *R = a.base[FBSTAB_MPC_R] + q * a.stride[FBSTAB_MPC_R],
               *S = a.base[FBSTAB_MPC_S] + q * a.stride[FBSTAB_MPC_S], *pq = a.base[FBSTAB_MPC_q] + q * a.stride[FBSTAB_MPC_q],
               *pr = a.base[FBSTAB_MPC_r] + q * a.stride[FBSTAB_MPC_r], *A = a.base[FBSTAB_MPC_A] + q * a.stride[FBSTAB_MPC_A],
               *B = a.base[FBSTAB_MPC_B] + q * a.stride[FBSTAB_MPC_B], *pc = a.base[FBSTAB_MPC_c] + q * a.stride[FBSTAB_MPC_c],
               *E = a.base[FBSTAB_MPC_E] + q * a.stride[FBSTAB_MPC_E], *L = a.base[FBSTAB_MPC_L] + q * a.stride[FBSTAB_MPC_L],
               *pd = a.base[FBSTAB_MPC_d] + q * a.stride[FBSTAB_MPC_d], *px0 = a.base[FBSTAB_MPC_x0] + q * a.stride[FBSTAB_MPC_x0];
  const double *z = a.x[0] + q * a.xstride[0], *l = a.x[1] + q * a.xstride[1], *v = a.x[2] + q * a.xstride[2],
               *y = a.x[3] + q * a.xstride[3];
  double sz = 0.0, sl = 0.0, sv = 0.0, sw = 0.0;
  for (int i = 0; i <= N; i++) {
    const double *zi = z + (long)i * ns, *li = l + (long)i * nx, *vi = v + (long)i * nc;
    const double *Qi = Q + (long)i * nx * nx, *Ri = R + (long)i * nu * nu, *Si = S + (long)i * nu * nx;
    const double *Ei = E + (long)i * nc * nx, *Li = L + (long)i * nc * nu;
    for (int r = t; r < ns; r += 64) {
      double acc;
      if (r < nx) {  // state row: Q x + S'u + q - l(i) + A'l(i+1) + E'v
        acc = pq[(long)i * nx + r] - li[r];
        for (int c = 0; c < nx; c++) acc = fma(Qi[r + c * nx], zi[c], acc);
        for (int c = 0; c < nu; c++) acc = fma(Si[c + r * nu], zi[nx + c], acc);
        if (i < N) {
          const double* Ai = A + (long)i * nx * nx;
          for (int j = 0; j < nx; j++) acc = fma(Ai[j + r * nx], li[nx + j], acc);
        }
        for (int k = 0; k < nc; k++) acc = fma(Ei[k + r * nc], vi[k], acc);
      } else {  // input row: S x + R u + r + B'l(i+1) + L'v
        const int ru = r - nx;
        acc = pr[(long)i * nu + ru];
        for (int c = 0; c < nx; c++) acc = fma(Si[ru + c * nu], zi[c], acc);
        for (int c = 0; c < nu; c++) acc = fma(Ri[ru + c * nu], zi[nx + c], acc);
        if (i < N) {
          const double* Bi = B + (long)i * nx * nu;
          for (int j = 0; j < nx; j++) acc = fma(Bi[j + ru * nx], li[nx + j], acc);
        }
        for (int k = 0; k < nc; k++) acc = fma(Li[k + ru * nc], vi[k], acc);
      }
      sz = fma(acc, acc, sz);
    }
    for (int r = t; r < nx; r += 64) {  // rl(i) = h(i) - (G z)(i)
      double acc;
      if (i == 0) {
        acc = -px0[r] + zi[r];
        sw = fma(px0[r], px0[r], sw);
      } else {
        const double *Ap = A + (long)(i - 1) * nx * nx, *Bp = B + (long)(i - 1) * nx * nu, *zp = zi - ns;
        double g = -zi[r];
        for (int c = 0; c < nx; c++) g = fma(Ap[r + c * nx], zp[c], g);
        for (int c = 0; c < nu; c++) g = fma(Bp[r + c * nx], zp[nx + c], g);
        const double cc = pc[(long)(i - 1) * nx + r];
        acc = -cc - g;
        sw = fma(cc, cc, sw);
      }
      sl = fma(acc, acc, sl);
      const double qq = pq[(long)i * nx + r];
      sw = fma(qq, qq, sw);
    }
    for (int r = t; r < nu; r += 64) {
      const double rr = pr[(long)i * nu + r];
      sw = fma(rr, rr, sw);
    }
    for (int k = t; k < nc; k += 64) {
      const double p = pnr(y[(long)i * nc + k], vi[k], opts.alpha);
      sv = fma(p, p, sv);
      const double dd = pd[(long)i * nc + k];
      sw = fma(dd, dd, sw);
    }
  }
  sz = wave_sum(sz);
  sl = wave_sum(sl);
  sv = wave_sum(sv);
  sw = wave_sum(sw);
  if (t == 0) {
    double* o = norms + 4 * q;
    o[0] = sqrt(sz);
    o[1] = sqrt(sl);
    o[2] = sqrt(sv);
    o[3] = opts.abs_tol + opts.rel_tol * (1.0 + sqrt(sw));
  }
}

struct DenseNormArgs {
  const double* base[6];  // H f G h A b (FBSTAB_DENSE_*)
  long long stride[6];
  const double* x[4];
  long long xstride[4];
  int nz, nl, nv;
};

// The dense analogue (dense_data.cc:12-41): column-major H (nz x nz), G (nl x nz), A (nv x nz).
__global__ __launch_bounds__(64) void fbstab_dense_final_norms_kernel(DenseNormArgs a, fbstab_options_t opts,
                                                                      double* norms, int batch) {
  const long q = blockIdx.x;
  if (q >= batch) return;
  const int t = threadIdx.x, nz = a.nz, nl = a.nl, nv = a.nv;
  const double *H = a.base[FBSTAB_DENSE_H] + q * a.stride[FBSTAB_DENSE_H], *f = a.base[FBSTAB_DENSE_f] + q * a.stride[FBSTAB_DENSE_f],
               *G = a.base[FBSTAB_DENSE_G] + q * a.stride[FBSTAB_DENSE_G], *h = a.base[FBSTAB_DENSE_h] + q * a.stride[FBSTAB_DENSE_h],
               *A = a.base[FBSTAB_DENSE_A] + q * a.stride[FBSTAB_DENSE_A], *b = a.base[FBSTAB_DENSE_b] + q * a.stride[FBSTAB_DENSE_b];
  const double *z = a.x[0] + q * a.xstride[0], *l = a.x[1] + q * a.xstride[1], *v = a.x[2] + q * a.xstride[2],
               *y = a.x[3] + q * a.xstride[3];
  double sz = 0.0, sl = 0.0, sv = 0.0, sw = 0.0;
  for (int r = t; r < nz; r += 64) {
    double acc = f[r];
    for (int c = 0; c < nz; c++) acc = fma(H[r + (long)c * nz], z[c], acc);
    for (int j = 0; j < nl; j++) acc = fma(G[j + (long)r * nl], l[j], acc);
    for (int k = 0; k < nv; k++) acc = fma(A[k + (long)r * nv], v[k], acc);
    sz = fma(acc, acc, sz);
    sw = fma(f[r], f[r], sw);
  }
  for (int j = t; j < nl; j += 64) {
    double acc = h[j];
    for (int c = 0; c < nz; c++) acc = fma(-G[j + (long)c * nl], z[c], acc);
    sl = fma(acc, acc, sl);
    sw = fma(h[j], h[j], sw);
  }
  for (int k = t; k < nv; k += 64) {
    const double p = pnr(y[k], v[k], opts.alpha);
    sv = fma(p, p, sv);
    sw = fma(b[k], b[k], sw);
  }
  sz = wave_sum(sz);
  sl = wave_sum(sl);
  sv = wave_sum(sv);
  sw = wave_sum(sw);
  if (t == 0) {
    double* o = norms + 4 * q;
    o[0] = sqrt(sz);
    o[1] = sqrt(sl);
    o[2] = sqrt(sv);
    o[3] = opts.abs_tol + opts.rel_tol * (1.0 + sqrt(sw));
  }
}


}  // namespace fbk
