// Developer probe: v_mfma_f64_4x4x4_4b_f64 on gfx950 - FOUR independent 4 x 4 x 4 products
// per instruction, one per 16-lane row of the wavefront, i.e. one per QP of the record
// kernel's layout (the 16 x 16 x 4 form contracts ACROSS the four rows and is useless for a
// product that belongs to one QP: DESIGN.md 4.1, "the matrix cores on the Riccati stage").
//   1. the lane <-> element map of the A, B and C / D operands inside a block;
//   2. cycles per instruction: dependent through the accumulator, independent, and with
//      independent v_fma_f64 written between the matrix instructions (does the vector pipe
//      issue while the matrix pipe works, for ONE wavefront per SIMD?);
//   3. the broadcast-FMA stream that does the same arithmetic today (4 v_mov_b64_dpp +
//      4 v_fma_f64 per 4 x 4 x 4 per row), for comparison.
// Standalone: hipcc --offload-arch=gfx950 -O3 -o mfma_4x4_probe mfma_4x4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

// out[(la * 16 + lb) * 64 + lane] = D of lane `lane` when A = 1 on lane la only, B = 1 on lane lb only
__global__ void map_kernel(double* out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 16; la++)
    for (int lb = 0; lb < 16; lb++) {
      const double a = (lane & 15) == la ? 1.0 : 0.0;
      const double b = (lane & 15) == lb ? 1.0 : 0.0;
      const double d = MFMA4(a, b, 0.0);
      out[(la * 16 + lb) * 64 + lane] = d;
    }
}
// blocks are independent: A of row 0 only must not reach rows 1..3
__global__ void block_kernel(double* out) {
  const int lane = threadIdx.x;
  const double a = lane < 16 ? 1.0 : 0.0, b = 1.0;
  out[lane] = MFMA4(a, b, 0.0);
}
// EXEC: a masked-off row keeps its accumulator, the others are computed
__global__ void exec_kernel(double* out) {
  const int lane = threadIdx.x;
  double d = -7.0;
  if ((lane >> 4) != 2) d = MFMA4(1.0, 1.0, d);
  out[lane] = d;
}

template <int MODE>
__global__ void rate(double* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
  double c[8], f[8];
  for (int k = 0; k < 8; k++) { c[k] = k * 0.125 + lane; f[k] = 0.5 * k; }
  asm volatile("s_nop 4");
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {  // 8 dependent
#pragma unroll
      for (int k = 0; k < 8; k++) c[0] = MFMA4(a, b, c[0]);
    } else if (MODE == 1) {  // 8 independent
#pragma unroll
      for (int k = 0; k < 8; k++) c[k] = MFMA4(a, b, c[k]);
    } else if (MODE == 2) {  // 8 independent, 2 independent FMAs behind each
#pragma unroll
      for (int k = 0; k < 8; k++) {
        c[k] = MFMA4(a, b, c[k]);
        __builtin_amdgcn_sched_barrier(0);
        f[k] = fma(f[k], a, b);
        f[(k + 4) & 7] = fma(f[(k + 4) & 7], b, a);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 3) {  // 8 independent, 4 independent FMAs behind each
#pragma unroll
      for (int k = 0; k < 8; k++) {
        c[k] = MFMA4(a, b, c[k]);
        __builtin_amdgcn_sched_barrier(0);
        f[k] = fma(f[k], a, b);
        f[(k + 2) & 7] = fma(f[(k + 2) & 7], b, a);
        f[(k + 4) & 7] = fma(f[(k + 4) & 7], b, a);
        f[(k + 6) & 7] = fma(f[(k + 6) & 7], a, a);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 4) {  // the same arithmetic as 8 x (4x4x4 per row) by broadcast + FMA: 32 pairs
#pragma unroll
      for (int k = 0; k < 8; k++) {
        double t0_ = __builtin_amdgcn_update_dpp(0.0, b, 0x150 + 0, 0xf, 0xf, true);
        double t1_ = __builtin_amdgcn_update_dpp(0.0, b, 0x150 + 1, 0xf, 0xf, true);
        double t2_ = __builtin_amdgcn_update_dpp(0.0, b, 0x150 + 2, 0xf, 0xf, true);
        double t3_ = __builtin_amdgcn_update_dpp(0.0, b, 0x150 + 3, 0xf, 0xf, true);
        c[k] = fma(a, t0_, c[k]);
        c[(k + 1) & 7] = fma(a, t1_, c[(k + 1) & 7]);
        c[(k + 2) & 7] = fma(a, t2_, c[(k + 2) & 7]);
        c[(k + 3) & 7] = fma(a, t3_, c[(k + 3) & 7]);
      }
    } else if (MODE == 5) {  // 8 plain independent FMAs (the vector pipe's own rate)
#pragma unroll
      for (int k = 0; k < 8; k++) f[k] = fma(f[k], a, b);
    } else if (MODE == 6) {  // dependent chain, 4 independent FMAs behind each (latency hiding within one wave)
#pragma unroll
      for (int k = 0; k < 8; k++) {
        c[0] = MFMA4(a, b, c[0]);
        __builtin_amdgcn_sched_barrier(0);
        f[k] = fma(f[k], a, b);
        f[(k + 2) & 7] = fma(f[(k + 2) & 7], b, a);
        f[(k + 4) & 7] = fma(f[(k + 4) & 7], b, a);
        f[(k + 6) & 7] = fma(f[(k + 6) & 7], a, a);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int k = 0; k < 8; k++) s += c[k] + f[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, sizeof(double) * 256 * 64 + (1 << 22));
  (void)hipMalloc(&cyc, 4096 * 8);
  map_kernel<<<1, 64>>>(out);
  std::vector<double> h(256 * 64);
  (void)hipMemcpy(h.data(), out, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
  // For every (la, lb): which lanes of block 0 received a 1?  D[i][j] = sum_k A[i][k] B[k][j]:
  // a single (la, lb) pair contributes to exactly one output iff k(la) == k(lb).
  printf("pairs (la, lb) -> output lane (block 0), '.' = no contribution\n     lb:");
  for (int lb = 0; lb < 16; lb++) printf("%3d", lb);
  printf("\n");
  for (int la = 0; la < 16; la++) {
    printf("la %2d:   ", la);
    for (int lb = 0; lb < 16; lb++) {
      int hit = -1, nhit = 0;
      for (int l = 0; l < 16; l++)
        if (h[(la * 16 + lb) * 64 + l] != 0.0) { hit = l; nhit++; }
      if (nhit == 0) printf("  .");
      else if (nhit == 1) printf("%3d", hit);
      else printf("  *");
    }
    printf("\n");
  }
  block_kernel<<<1, 64>>>(out);
  double hb[64];
  (void)hipMemcpy(hb, out, sizeof(hb), hipMemcpyDeviceToHost);
  printf("A = 1 on row 0 only, B = 1: D lane 0 %.1f, lane 16 %.1f, lane 32 %.1f, lane 48 %.1f (blocks %s)\n", hb[0], hb[16],
         hb[32], hb[48], (hb[16] == 0.0 && hb[32] == 0.0 && hb[48] == 0.0) ? "independent" : "NOT independent");
  exec_kernel<<<1, 64>>>(out);
  (void)hipMemcpy(hb, out, sizeof(hb), hipMemcpyDeviceToHost);
  printf("row 2 masked off: D lane 0 %.1f, lane 32 %.1f (kept -7: %s), lane 48 %.1f\n", hb[0], hb[32],
         hb[32] == -7.0 ? "yes" : "NO", hb[48]);
  const int iters = 2000;
  const char* names[7] = {"8 dependent 4x4x4", "8 independent 4x4x4", "8 independent + 2 FMA each", "8 independent + 4 FMA each",
                          "32 bcast + 32 FMA (same arithmetic as 8 MFMA)", "8 independent v_fma_f64",
                          "8 dependent 4x4x4 + 4 FMA each"};
  for (int mode = 0; mode < 7; mode++) {
    for (int waves = 1; waves <= 2; waves++) {  // one or two wavefronts per SIMD (1024 or 2048 wavefronts, 256 CUs x 4 SIMDs)
      const int blocks = 1024 * waves;
      switch (mode) {
        case 0: rate<0><<<blocks, 64>>>(out, cyc, iters); break;
        case 1: rate<1><<<blocks, 64>>>(out, cyc, iters); break;
        case 2: rate<2><<<blocks, 64>>>(out, cyc, iters); break;
        case 3: rate<3><<<blocks, 64>>>(out, cyc, iters); break;
        case 4: rate<4><<<blocks, 64>>>(out, cyc, iters); break;
        case 5: rate<5><<<blocks, 64>>>(out, cyc, iters); break;
        case 6: rate<6><<<blocks, 64>>>(out, cyc, iters); break;
      }
      (void)hipDeviceSynchronize();
      std::vector<long long> hc(blocks);
      (void)hipMemcpy(hc.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
      double mean = 0;
      for (long long v : hc) mean += (double)v;
      mean /= blocks;
      printf("%-48s %d wavefront(s)/SIMD: %8.1f cycles per group of 8 (%.1f per MFMA / pair-quad / FMA)\n", names[mode], waves,
             mean / iters, mean / iters / 8);
    }
  }
  return 0;
}
