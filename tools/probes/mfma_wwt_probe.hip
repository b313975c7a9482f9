// Developer probe: Pi+ = W W' of the record kernel's forward stage (four QPs per
// wavefront, lane r of a 16-lane row holding row r of its QP's 12 x 16 W) done two ways
//   DPP : the kernel's broadcast-FMA stream (16 x 12 v_mov_b64_dpp + 16 x 12 v_fma_f64)
//   MFMA: v_mfma_f64_16x16x4 per QP.  The contraction index of that instruction runs
//         over the four 16-lane rows, i.e. over the four QPs, so the operands of QP q
//         are first gathered from row q into all four rows (a 4 x 4 transpose of
//         row blocks between four registers: two v_permlane16_swap + two
//         v_permlane32_swap per 32-bit half) and the results scattered back the same way.
// Prints the largest difference between the two and the cycles of each per stage.
// Standalone: hipcc --offload-arch=gfx950 -O3 -o mfma_wwt_probe mfma_wwt_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int J>
__device__ __forceinline__ double bc(double x) {
  return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + J, 0xf, 0xf, true);
}

// out[q].row[k] = in[k].row[q] for the four 16-lane rows of a wavefront (32-bit values)
__device__ __forceinline__ void transpose4(unsigned (&v)[4]) {
  auto a = __builtin_amdgcn_permlane16_swap(v[0], v[1], false, false);  // [v0.r0 v1.r0 v0.r2 v1.r2], [v0.r1 v1.r1 v0.r3 v1.r3]
  auto b = __builtin_amdgcn_permlane16_swap(v[2], v[3], false, false);
  auto c = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);  // [a0.r0 a0.r1 b0.r0 b0.r1], [a0.r2 a0.r3 b0.r2 b0.r3]
  auto d = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
  v[0] = c[0]; v[2] = c[1]; v[1] = d[0]; v[3] = d[1];
}
__device__ __forceinline__ void transpose4d(double (&x)[4]) {
  unsigned lo[4], hi[4];
#pragma unroll
  for (int i = 0; i < 4; i++) { lo[i] = __double2loint(x[i]); hi[i] = __double2hiint(x[i]); }
  transpose4(lo);
  transpose4(hi);
#pragma unroll
  for (int i = 0; i < 4; i++) x[i] = __hiloint2double(hi[i], lo[i]);
}

template <int K, int C>
__device__ __forceinline__ void row_k(const double (&W)[16], double (&P)[12]) {
  if constexpr (C < 12) {
    P[C] = fma(W[K], bc<C>(W[K]), P[C]);
    row_k<K, C + 1>(W, P);
  }
}
template <int K>
__device__ __forceinline__ void all_k(const double (&W)[16], double (&P)[12]) {
  if constexpr (K < 16) {
    row_k<K, 0>(W, P);
    all_k<K + 1>(W, P);
  }
}
__device__ __forceinline__ void wwt_dpp(const double (&W)[16], double (&P)[12]) {
#pragma unroll
  for (int c = 0; c < 12; c++) P[c] = 0.0;
  all_k<0>(W, P);
}

__device__ __forceinline__ void wwt_mfma(const double (&W)[16], double (&P)[12]) {
  d4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; q++) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int m = 0; m < 4; m++) {
    // registers 4m..4m+3 of every row -> operand of QP q: row kk holds register 4m+kk of row q
    double op[4] = {W[4 * m], W[4 * m + 1], W[4 * m + 2], W[4 * m + 3]};
    transpose4d(op);
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[q], op[q], acc[q], 0, 0, 0);
  }
  // D of QP q: lane (g, j), register v = (W W')[4 v + g][j]; back to lane j of row q
#pragma unroll
  for (int v = 0; v < 3; v++) {
    double o[4] = {acc[0][v], acc[1][v], acc[2][v], acc[3][v]};
    transpose4d(o);  // o[g].row[q] = acc[q][v].row[g]
#pragma unroll
    for (int g = 0; g < 4; g++) P[4 * v + g] = o[g];
  }
}

__global__ void check(double* err) {
  const int lane = threadIdx.x;
  double W[16], Pa[12], Pb[12];
  for (int k = 0; k < 16; k++) W[k] = (lane & 15) < 12 ? sin(0.37 * lane + 1.3 * k) : 0.0;
  wwt_dpp(W, Pa);
  wwt_mfma(W, Pb);
  double e = 0.0;
  for (int c = 0; c < 12; c++) e = fmax(e, fabs(Pa[c] - Pb[c]));
  err[lane] = e;
  err[64 + lane] = Pa[3];
  err[128 + lane] = Pb[3];
}

// The same two products with 256 independent FMAs of "other work" per stage: behind the
// DPP product, or sixteen of them after each of the sixteen matrix instructions (a
// wavefront issues in order: a matrix instruction that finds the pipe busy - 64 cycles
// per v_mfma_f64_16x16x4 - holds back everything behind it, so overlap has to be
// written into the instruction stream).
__device__ __forceinline__ void other16(double (&o)[16], double a) {
#pragma unroll
  for (int k = 0; k < 16; k++) o[k] = fma(o[k], a, 1e-3);
}
__device__ __forceinline__ void wwt_mfma_interleaved(const double (&W)[16], double (&P)[12], double (&o)[16], double a) {
  d4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; q++) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int m = 0; m < 4; m++) {
    double op[4] = {W[4 * m], W[4 * m + 1], W[4 * m + 2], W[4 * m + 3]};
    transpose4d(op);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[q], op[q], acc[q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      other16(o, a);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int v = 0; v < 3; v++) {
    double t[4] = {acc[0][v], acc[1][v], acc[2][v], acc[3][v]};
    transpose4d(t);
#pragma unroll
    for (int g = 0; g < 4; g++) P[4 * v + g] = t[g];
  }
}
template <int MODE>
__global__ void rate_with_work(double* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  double W[16], P[12], o[16];
  for (int k = 0; k < 16; k++) { W[k] = 1e-3 * (lane + 7 * k); o[k] = 0.5 + 1e-3 * k; }
  const double a = 0.999;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {
      wwt_dpp(W, P);
#pragma unroll
      for (int r = 0; r < 16; r++) other16(o, a);
    } else {
      wwt_mfma_interleaved(W, P, o, a);
    }
#pragma unroll
    for (int c = 0; c < 12; c++) W[c] = fma(1e-9, P[c], W[c]);
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0.0;
  for (int k = 0; k < 16; k++) s += W[k] + o[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
__global__ void rate(double* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  double W[16], P[12];
  for (int k = 0; k < 16; k++) W[k] = 1e-3 * (lane + 7 * k);
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) wwt_dpp(W, P); else wwt_mfma(W, P);
#pragma unroll
    for (int c = 0; c < 12; c++) W[c] = fma(1e-9, P[c], W[c]);  // the next product depends on this one
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0.0;
  for (int k = 0; k < 16; k++) s += W[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  double* derr; hipMalloc(&derr, 192 * sizeof(double));
  check<<<1, 64>>>(derr);
  std::vector<double> e(192);
  hipMemcpy(e.data(), derr, 192 * sizeof(double), hipMemcpyDeviceToHost);
  double mx = 0.0;
  for (int l = 0; l < 64; l++) mx = fmax(mx, e[l]);
  printf("max |W W' (DPP) - W W' (MFMA)| over the wavefront: %.3e  (lane 17: %.12f vs %.12f)\n", mx, e[64 + 17], e[128 + 17]);
  double* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096 * 8);
  const int iters = 2000;
  for (int waves = 1; waves <= 2; waves++)
    for (int m = 0; m < 2; m++) {
      double cc = 0;
      for (int rep = 0; rep < 2; rep++) {
        if (m == 0) rate<0><<<dim3(256), dim3(64 * 4), 0, 0>>>(out, cyc, iters);
        else rate<1><<<dim3(256), dim3(64 * 4), 0, 0>>>(out, cyc, iters);
        long long c[1]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        cc = (double)c[0];
      }
      printf("%-5s one wavefront per SIMD: %8.1f cycles per W W' of four QPs (dependent products back to back)\n",
             m == 0 ? "DPP" : "MFMA", cc / iters);
      if (waves == 1) continue;
    }
  for (int m = 0; m < 2; m++) {
    double cc = 0;
    for (int rep = 0; rep < 2; rep++) {
      if (m == 0) rate_with_work<0><<<dim3(256), dim3(64 * 4), 0, 0>>>(out, cyc, iters);
      else rate_with_work<1><<<dim3(256), dim3(64 * 4), 0, 0>>>(out, cyc, iters);
      long long c[1]; (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
      cc = (double)c[0];
    }
    printf("%-5s + 256 independent FMAs (%s): %8.1f cycles\n", m == 0 ? "DPP" : "MFMA",
           m == 0 ? "behind the product" : "sixteen behind each matrix instruction", cc / iters);
  }
  return 0;
}
