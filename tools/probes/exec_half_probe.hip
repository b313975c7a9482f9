// Developer probe (gfx950): does a vector instruction of a wavefront whose EXEC mask leaves whole 16- or 32-lane
// groups empty take fewer cycles?  (If it did, a wavefront of the record kernel with idle rows could run its
// Newton step under a narrower EXEC.)  A loop of FP64 FMAs / fused broadcast-FMAs, dependent and independent,
// timed with s_memtime under four masks: all 64 lanes, lanes 0-31, lanes 0-15, lanes 0-15 + 32-47.
#include <hip/hip_runtime.h>
#include <cstdio>

#define FMAC_BC(acc, y, x, J) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(y), "v"(x))

template <int MODE>
__global__ void rate(double* out, long long* cyc, int iters, unsigned long long mask) {
  const int lane = threadIdx.x & 63;
  double a = 1.0 + lane * 1e-3, y = 1.0 - lane * 1e-3;
  double c[8];
  for (int k = 0; k < 8; k++) c[k] = k * 0.125 + lane;
  long long t0 = 0, t1 = 0;
  if ((mask >> lane) & 1ull) {
    asm volatile("s_nop 4");
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
      if (MODE == 0) {  // 8 independent v_fma_f64
#pragma unroll
        for (int k = 0; k < 8; k++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[k]) : "v"(a), "v"(y));
      } else if (MODE == 1) {  // 8 dependent v_fma_f64
#pragma unroll
        for (int k = 0; k < 8; k++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[0]) : "v"(a), "v"(y));
      } else if (MODE == 2) {  // 8 independent fused broadcast-FMAs
        FMAC_BC(c[0], y, a, 0); FMAC_BC(c[1], y, a, 1); FMAC_BC(c[2], y, a, 2); FMAC_BC(c[3], y, a, 3);
        FMAC_BC(c[4], y, a, 4); FMAC_BC(c[5], y, a, 5); FMAC_BC(c[6], y, a, 6); FMAC_BC(c[7], y, a, 7);
      } else {  // 8 independent 32-bit moves (a non-FP64 vector instruction)
#pragma unroll
        for (int k = 0; k < 8; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(((int*)c)[2 * k]) : "v"(lane));
      }
    }
    t1 = __builtin_readcyclecounter();
  }
  double s = 0;
  for (int k = 0; k < 8; k++) s += c[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + y;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  double* out; long long* cyc; (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 4096 * 8);
  const int iters = 20000;
  const char* names[] = {"8 independent v_fma_f64", "8 dependent v_fma_f64", "8 independent v_fmac_f64_dpp", "8 independent v_add_u32"};
  const unsigned long long masks[] = {~0ull, 0xffffffffull, 0xffffull, 0x0000ffff0000ffffull};
  const char* mnames[] = {"all 64 lanes", "lanes 0-31", "lanes 0-15", "lanes 0-15 and 32-47"};
  for (int m = 0; m < 4; m++)
    for (int k = 0; k < 4; k++) {
      double cc = 0;
      for (int rep = 0; rep < 2; rep++) {
        switch (m) {
          case 0: rate<0><<<1, 64>>>(out, cyc, iters, masks[k]); break;
          case 1: rate<1><<<1, 64>>>(out, cyc, iters, masks[k]); break;
          case 2: rate<2><<<1, 64>>>(out, cyc, iters, masks[k]); break;
          case 3: rate<3><<<1, 64>>>(out, cyc, iters, masks[k]); break;
        }
        long long c[1]; (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        cc = (double)c[0];
      }
      printf("%-30s %-22s %7.2f cycles per instruction (one wavefront alone on its SIMD)\n", names[m], mnames[k], cc / iters / 8);
    }
  return 0;
}
