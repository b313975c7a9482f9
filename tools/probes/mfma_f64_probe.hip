// Developer probe: operand layout and issue/latency cycles of
// v_mfma_f64_4x4x4f64 (4 blocks of 4x4x4) on gfx950, next to v_fma_f64 and
// v_mov_b64_dpp.  Standalone: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void layout_probe(int* out) {
  // one-hot A at lane la, one-hot B at lane lb (block 0): record the lane of block 0 that receives 1
  const int lane = threadIdx.x;
  for (int la = 0; la < 16; la++)
    for (int lb = 0; lb < 16; lb++) {
      double a = lane == la ? 1.0 : 0.0;
      double b = lane == lb ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) out[la * 16 + lb] = m ? __ffsll((long long)m) - 1 : -1;
    }
}

template <int MODE>
__global__ void rate_probe(double* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {  // 8 independent MFMA chains
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
      c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
      c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
    } else if (MODE == 1) {  // one dependent MFMA chain (8 per iteration)
#pragma unroll
      for (int k = 0; k < 8; k++) c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    } else if (MODE == 2) {  // 8 independent FMA chains
      c0 = fma(a, b, c0); c1 = fma(a, b, c1); c2 = fma(a, b, c2); c3 = fma(a, b, c3);
      c4 = fma(a, b, c4); c5 = fma(a, b, c5); c6 = fma(a, b, c6); c7 = fma(a, b, c7);
    } else if (MODE == 3) {  // dependent FMA chain
#pragma unroll
      for (int k = 0; k < 8; k++) c0 = fma(a, b, c0);
    } else if (MODE == 4) {  // fma + 64-bit DPP row broadcast pairs, 8 independent
      c0 = fma(a, __builtin_amdgcn_update_dpp(0.0, c1, 0x150, 0xf, 0xf, true), c0);
      c1 = fma(a, __builtin_amdgcn_update_dpp(0.0, c2, 0x151, 0xf, 0xf, true), c1);
      c2 = fma(a, __builtin_amdgcn_update_dpp(0.0, c3, 0x152, 0xf, 0xf, true), c2);
      c3 = fma(a, __builtin_amdgcn_update_dpp(0.0, c4, 0x153, 0xf, 0xf, true), c3);
      c4 = fma(a, __builtin_amdgcn_update_dpp(0.0, c5, 0x154, 0xf, 0xf, true), c4);
      c5 = fma(a, __builtin_amdgcn_update_dpp(0.0, c6, 0x155, 0xf, 0xf, true), c5);
      c6 = fma(a, __builtin_amdgcn_update_dpp(0.0, c7, 0x156, 0xf, 0xf, true), c6);
      c7 = fma(a, __builtin_amdgcn_update_dpp(0.0, c0, 0x157, 0xf, 0xf, true), c7);
    } else if (MODE == 5) {  // 16x16x4 f64, 2 independent chains of 4 results
      // handled in rate16 below
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef double double4_ __attribute__((ext_vector_type(4)));
template <int DEP>
__global__ void rate16(double* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
  double4_ c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (DEP) {
#pragma unroll
      for (int k = 0; k < 4; k++) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double4_ s = c0 + c1 + c2 + c3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  int* dmap; hipMalloc(&dmap, 256 * sizeof(int));
  layout_probe<<<1, 64>>>(dmap);
  std::vector<int> map(256);
  hipMemcpy(map.data(), dmap, 256 * sizeof(int), hipMemcpyDeviceToHost);
  printf("D lane for (A one-hot lane la [row], B one-hot lane lb [col]); -1 = no product\n");
  for (int la = 0; la < 16; la++) { for (int lb = 0; lb < 16; lb++) printf("%3d", map[la * 16 + lb]); printf("\n"); }
  double* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096 * 8);
  const int iters = 4000;
  const char* names[] = {"mfma4x4x4 x8 indep", "mfma4x4x4 x8 dep", "fma x8 indep", "fma x8 dep", "fma+dpp64 x8"};
  for (int waves = 1; waves <= 8; waves *= 2) {
    auto run = [&](int mode) {
      dim3 g(256), b(64 * waves);
      switch (mode) {
        case 0: rate_probe<0><<<g, b>>>(out, cyc, iters); break;
        case 1: rate_probe<1><<<g, b>>>(out, cyc, iters); break;
        case 2: rate_probe<2><<<g, b>>>(out, cyc, iters); break;
        case 3: rate_probe<3><<<g, b>>>(out, cyc, iters); break;
        case 4: rate_probe<4><<<g, b>>>(out, cyc, iters); break;
        case 5: rate16<0><<<g, b>>>(out, cyc, iters); break;
        case 6: rate16<1><<<g, b>>>(out, cyc, iters); break;
      }
      long long c[4]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
      return (double)c[0];
    };
    for (int m = 0; m < 5; m++) { run(m); double c = run(m); printf("waves/WG=%d %-22s %.2f cyc/op (clock units)\n", waves, names[m], c / (iters * 8.0)); }
    { run(5); double c = run(5); printf("waves/WG=%d %-22s %.2f cyc/op\n", waves, "mfma16x16x4 x4 indep", c / (iters * 4.0)); }
    { run(6); double c = run(6); printf("waves/WG=%d %-22s %.2f cyc/op\n", waves, "mfma16x16x4 x4 dep", c / (iters * 4.0)); }
  }
  // clock calibration: wall_clock64 vs readcyclecounter
  return 0;
}
