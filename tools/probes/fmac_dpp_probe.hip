// Developer probe: v_fmac_f64_dpp (FMA with a row_newbcast operand) on gfx950:
// correctness of the lane selection and cycles per instruction next to the
// v_mov_b64_dpp + v_fma_f64 pair it replaces.
#include <hip/hip_runtime.h>
#include <cstdio>

#define FMAC_BC(acc, y, x, J) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(y), "v"(x))

__global__ void check(double* out) {
  const int lane = threadIdx.x;
  double y = 100.0 + lane, x = 2.0, acc = 1000.0 * lane;
  asm volatile("s_nop 4");
  FMAC_BC(acc, y, x, 5);  // acc += (lane 5 of my row).y * x
  out[lane] = acc;
}

template <int MODE>
__global__ void rate(double* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  double a = 1.0 + lane * 1e-3, y = 1.0 - lane * 1e-3;
  double c[8];
  for (int k = 0; k < 8; k++) c[k] = k * 0.125 + lane;
  asm volatile("s_nop 4");
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {  // 8 independent fmac_dpp, same broadcast source
      FMAC_BC(c[0], y, a, 0); FMAC_BC(c[1], y, a, 1); FMAC_BC(c[2], y, a, 2); FMAC_BC(c[3], y, a, 3);
      FMAC_BC(c[4], y, a, 4); FMAC_BC(c[5], y, a, 5); FMAC_BC(c[6], y, a, 6); FMAC_BC(c[7], y, a, 7);
    } else if (MODE == 1) {  // dependent chain through the accumulator
      FMAC_BC(c[0], y, a, 0); FMAC_BC(c[0], y, a, 1); FMAC_BC(c[0], y, a, 2); FMAC_BC(c[0], y, a, 3);
      FMAC_BC(c[0], y, a, 4); FMAC_BC(c[0], y, a, 5); FMAC_BC(c[0], y, a, 6); FMAC_BC(c[0], y, a, 7);
    } else if (MODE == 2) {  // mov_dpp + fma pairs, movs 4 ahead (the current code shape)
      double t[8];
#pragma unroll
      for (int k = 0; k < 8; k++) t[k] = __builtin_amdgcn_update_dpp(0.0, y, 0x153, 0xf, 0xf, true);
#pragma unroll
      for (int k = 0; k < 8; k++) c[k] = fma(a, t[k], c[k]);
    } else if (MODE == 3) {  // broadcast source produced right before (hazard distance check)
      double yy = y * a;
      asm volatile("s_nop 1");
      FMAC_BC(c[0], yy, a, 0); FMAC_BC(c[1], yy, a, 1); FMAC_BC(c[2], yy, a, 2); FMAC_BC(c[3], yy, a, 3);
      FMAC_BC(c[4], yy, a, 4); FMAC_BC(c[5], yy, a, 5); FMAC_BC(c[6], yy, a, 6); FMAC_BC(c[7], yy, a, 7);
      y = yy * 0.999;
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int k = 0; k < 8; k++) s += c[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + y;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  double* out; long long* cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 4096 * 8);
  check<<<1, 64>>>(out);
  double h[64]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) { double want = 1000.0 * l + (100.0 + (l & ~15) + 5) * 2.0; if (h[l] != want) bad++; }
  printf("lane selection: %s (lane 17 got %.1f, want %.1f)\n", bad ? "WRONG" : "ok", h[17], 17000.0 + (100 + 16 + 5) * 2.0);
  const int iters = 4000;
  const char* names[] = {"8 indep fmac_dpp", "8 dependent fmac_dpp", "8 mov_dpp then 8 fma", "mul; nop; 8 fmac_dpp"};
  for (int waves = 4; waves <= 8; waves *= 2)
    for (int m = 0; m < 4; m++) {
      dim3 g(256), b(64 * waves);
      double cc = 0;
      for (int rep = 0; rep < 2; rep++) {
        switch (m) {
          case 0: rate<0><<<g, b>>>(out, cyc, iters); break;
          case 1: rate<1><<<g, b>>>(out, cyc, iters); break;
          case 2: rate<2><<<g, b>>>(out, cyc, iters); break;
          case 3: rate<3><<<g, b>>>(out, cyc, iters); break;
        }
        long long c[1]; (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        cc = (double)c[0];
      }
      printf("waves/SIMD=%d %-24s %8.2f cycles per iteration\n", waves / 4, names[m], cc / iters);
    }
  return 0;
}
