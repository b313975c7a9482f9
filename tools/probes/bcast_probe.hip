// Developer probe: cost (shader cycles per wave instruction, steady state) of the
// ways to hand one lane's f64 to the other lanes of its 16-lane row on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o bcast_probe bcast_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define DPP64(x, ctrl) __builtin_amdgcn_update_dpp(0.0, (x), (ctrl), 0xf, 0xf, true)

__device__ __forceinline__ double dpp32x2(double x, int) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad64(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x00, 0xf, 0xf, true);  // quad_perm [0,0,0,0]
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x00, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void probe(double* out, long long* cyc, int iters) {
  __shared__ double sm[1024];
  const int lane = threadIdx.x & 63;
  const int row = lane >> 4;
  double a = 1.0 + lane * 1e-3;
  double c[8];
  for (int k = 0; k < 8; k++) c[k] = k * 0.125 + lane;
  for (int k = threadIdx.x; k < 1024; k += blockDim.x) sm[k] = k;
  __syncthreads();
  __attribute__((address_space(3))) double* lp = (__attribute__((address_space(3))) double*)sm + row * 48 + (threadIdx.x >> 6) * 200;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {  // 8 dpp64 row_newbcast movs only
#pragma unroll
      for (int k = 0; k < 8; k++) c[k] = DPP64(c[k], 0x150 + 3) + 0.0 * 0;  // mov only
    } else if (MODE == 1) {  // 8 x (2 dpp32 movs)
#pragma unroll
      for (int k = 0; k < 8; k++) c[k] = dpp32x2(c[k], 0);
    } else if (MODE == 2) {  // 8 dpp64 movs of one source then 8 fmas
      double b[8];
#pragma unroll
      for (int k = 0; k < 8; k++) b[k] = DPP64(c[(k + 1) & 7], 0x153);
#pragma unroll
      for (int k = 0; k < 8; k++) c[k] = fma(a, b[k], c[k]);
    } else if (MODE == 3) {  // 8 LDS broadcast reads (b64) then 8 fmas
      double b[8];
#pragma unroll
      for (int k = 0; k < 8; k++) b[k] = lp[k];
#pragma unroll
      for (int k = 0; k < 8; k++) c[k] = fma(a, b[k], c[k]);
      asm volatile("" ::: "memory");
    } else if (MODE == 4) {  // 4 LDS b128 reads then 8 fmas
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2 b[4];
#pragma unroll
      for (int k = 0; k < 4; k++) b[k] = *(__attribute__((address_space(3))) d2*)(lp + 2 * k);
#pragma unroll
      for (int k = 0; k < 4; k++) { c[2 * k] = fma(a, b[k][0], c[2 * k]); c[2 * k + 1] = fma(a, b[k][1], c[2 * k + 1]); }
      asm volatile("" ::: "memory");
    } else if (MODE == 5) {  // quad_perm 2x32
#pragma unroll
      for (int k = 0; k < 8; k++) c[k] = quad64(c[k]);
    } else if (MODE == 6) {  // readlane x2 + fma with sgpr
#pragma unroll
      for (int k = 0; k < 8; k++) {
        int lo = __builtin_amdgcn_readlane(__double2loint(c[(k + 1) & 7]), k);
        int hi = __builtin_amdgcn_readlane(__double2hiint(c[(k + 1) & 7]), k);
        c[k] = fma(a, __hiloint2double(hi, lo), c[k]);
      }
    } else if (MODE == 7) {  // LDS write b64 + barrier-free readback (own row) : transpose cost
#pragma unroll
      for (int k = 0; k < 8; k++) lp[k * 17 + (lane & 15)] = c[k];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 8; k++) c[k] = lp[(lane & 15) * 17 + k];
      asm volatile("" ::: "memory");
    } else if (MODE == 8) {  // dependent pair chain: dpp64 -> fma -> dpp64 -> fma ...
#pragma unroll
      for (int k = 0; k < 8; k++) c[0] = fma(a, DPP64(c[0], 0x153), c[0]);
    } else if (MODE == 9) {  // 8 global loads b64 + fma? (skipped)
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int k = 0; k < 8; k++) s += c[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  double* out; long long* cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 4096 * 8);
  const int iters = 4000;
  const char* names[] = {"dpp64 bcast mov x8", "2x dpp32 bcast x8", "8 dpp64 then 8 fma", "8 lds b64 rd + 8 fma",
                         "4 lds b128 rd + 8 fma", "quad_perm 2x32 x8", "2 readlane + fma x8", "lds wr8+rd8 transpose",
                         "dep chain dpp64->fma x8"};
  for (int waves = 4; waves <= 16; waves *= 2) {
    for (int m = 0; m < 9; m++) {
      dim3 g(256), b(64 * waves);
      double cc = 0;
      for (int rep = 0; rep < 2; rep++) {
        switch (m) {
          case 0: probe<0><<<g, b>>>(out, cyc, iters); break;
          case 1: probe<1><<<g, b>>>(out, cyc, iters); break;
          case 2: probe<2><<<g, b>>>(out, cyc, iters); break;
          case 3: probe<3><<<g, b>>>(out, cyc, iters); break;
          case 4: probe<4><<<g, b>>>(out, cyc, iters); break;
          case 5: probe<5><<<g, b>>>(out, cyc, iters); break;
          case 6: probe<6><<<g, b>>>(out, cyc, iters); break;
          case 7: probe<7><<<g, b>>>(out, cyc, iters); break;
          case 8: probe<8><<<g, b>>>(out, cyc, iters); break;
        }
        long long c[1]; (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        cc = (double)c[0];
      }
      printf("waves/SIMD=%d %-26s %8.2f cycles per iteration\n", waves / 4, names[m], cc / iters);
    }
  }
  return 0;
}
