// Developer probe: does the shader clock drop when every CU runs FP64 FMAs?
// Times a fixed dependent-free FMA loop in wave 0 with s_memtime and with the
// 100 MHz s_memrealtime, for grids of 1 .. 2048 waves.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, long long* t, int iters) {
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  long long m0 = __builtin_readcyclecounter();
  long long r0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
    c0 = fma(a, b, c0); c1 = fma(a, b, c1); c2 = fma(a, b, c2); c3 = fma(a, b, c3);
    c4 = fma(a, b, c4); c5 = fma(a, b, c5); c6 = fma(a, b, c6); c7 = fma(a, b, c7);
  }
  long long m1 = __builtin_readcyclecounter();
  long long r1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = m1 - m0; t[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
  double* out; long long* t; hipMalloc(&out, 1 << 24); hipMalloc(&t, 1 << 20);
  const int iters = 400000;  // ~10 ms
  int grids[] = {1, 256, 1024, 2048, 4096};
  for (int g : grids) {
    for (int rep = 0; rep < 2; rep++) {
      k<<<g, 64>>>(out, t, iters);
      long long h[2]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
      if (rep) printf("waves=%5d  memtime ticks/fma=%.3f  realtime(100MHz) ns/fma=%.3f  memtime MHz=%.1f\n", g,
                      (double)h[0] / (iters * 8.0), (double)h[1] * 10.0 / (iters * 8.0), (double)h[0] / ((double)h[1] * 0.01));
    }
  }
  return 0;
}
