// Developer probe (gfx950): what one wavefront ALONE on its SIMD can issue - the regime of the record kernel
// (493 registers: one wavefront per SIMD).  Cycles per instruction (s_memtime) of unrolled streams of
// independent instructions of one class (read-modify-write forms only: a stream that rewrites its destinations
// without reading them measures something else), of an FP64 / integer pair, and of the FP64 stream with two and
// four wavefronts on the SIMD (the slowest wavefront's time: the arbiter favours the oldest).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8_(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
// 64 instructions (groups) per trip of the loop: the taken branch at its end costs ~28 cycles
#define REP8(X) REP8_(X) REP8_(X) REP8_(X) REP8_(X) REP8_(X) REP8_(X) REP8_(X) REP8_(X)
#define FMA64(k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[k]) : "v"(a), "v"(y));
#define MUL64(k) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c[k]) : "v"(a), "v"(y));
#define ADD64(k) asm volatile("v_add_f64 %0, %1, %0" : "+v"(c[k]) : "v"(a));
#define FMACD(k) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(c[k]) : "v"(y), "v"(a));
#define FMA32(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[k]) : "v"(fa), "v"(fy));
#define MOV32(k) asm volatile("v_mov_b32 %0, %1" : "=v"(u[k]) : "v"(lane));
#define MOV64(k) asm volatile("v_mov_b64 %0, %1" : "=v"(c[k]) : "v"(a));
#define CND32(k) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[k]) : "v"(lane), "v"(lane2));
#define ACCW(k) asm volatile("v_accvgpr_write_b32 a" #k ", %0" ::"v"(lane));
#define ADDU(k) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[k]) : "v"(lane));
// mixes: one FP64 FMA followed by one / two / three 32-bit instructions
#define MIX1(k) FMA64(k) ADDU(k)
#define MIX2(k) FMA64(k) ADDU(k) MOV32(k)
#define MIX3(k) FMA64(k) ADDU(k) MOV32(k) CND32(k)

template <int MODE>
__global__ void rate(double* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63, lane2 = lane ^ 5;
  double a = 1.0 + lane * 1e-3, y = 1.0 - lane * 1e-3;
  float fa = 1.0f + lane * 1e-3f, fy = 1.0f - lane * 1e-3f;
  double c[8];
  float f[8];
  int u[8];
  for (int k = 0; k < 8; k++) { c[k] = k * 0.125 + lane; f[k] = k * 0.25f; u[k] = k + lane; }
  asm volatile("s_nop 4");
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) { REP8(FMA64) }
    else if (MODE == 1) { REP8(MUL64) }
    else if (MODE == 2) { REP8(ADD64) }
    else if (MODE == 3) { REP8(FMACD) }
    else if (MODE == 4) { REP8(FMA32) }
    else if (MODE == 5) { REP8(MOV32) }
    else if (MODE == 6) { REP8(MOV64) }
    else if (MODE == 7) { REP8(CND32) }
    else if (MODE == 8) { REP8(ACCW) }
    else if (MODE == 9) { REP8(ADDU) }
    else if (MODE == 10) { REP8(MIX1) }
    else if (MODE == 11) { REP8(MIX2) }
    else if (MODE == 12) { REP8(MIX3) }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int k = 0; k < 8; k++) s += c[k] + f[k] + u[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + y;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

static double g_last_ms = 0.0;  // the last launch by the host's events: ticks of s_memtime per microsecond come out of it
template <int MODE>
static double run(double* out, long long* cyc, int iters, int waves_per_simd) {
  long long c[16];
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    (void)hipEventRecord(e0, 0);
    rate<MODE><<<1, 256 * waves_per_simd>>>(out, cyc, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); g_last_ms = ms;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  long long worst = 0;  // the slowest wavefront of the workgroup (the SIMD's arbiter favours the oldest one)
  for (int w = 0; w < 4 * waves_per_simd; w++) worst = c[w] > worst ? c[w] : worst;
  return (double)worst / iters;
}

int main() {
  double* out; long long* cyc; (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 4096 * 8);
  const int it = 4000;
  printf("one wavefront per SIMD (a workgroup of 256), cycles per instruction of an unrolled stream of independent instructions:\n");
  { const double t = run<0>(out, cyc, 40000, 1); printf("   (s_memtime: %.0f ticks per microsecond of the launch - %.0f ticks in %.3f ms)\n", t * 40000 / (g_last_ms * 1e3), t * 40000, g_last_ms); }
  printf("   v_fma_f64            %6.2f\n", run<0>(out, cyc, it, 1) / 64);
  printf("   v_add_f64            %6.2f\n", run<2>(out, cyc, it, 1) / 64);
  printf("   v_fmac_f64_dpp       %6.2f\n", run<3>(out, cyc, it, 1) / 64);
  printf("   v_fma_f32            %6.2f\n", run<4>(out, cyc, it, 1) / 64);
  printf("   v_accvgpr_write_b32  %6.2f\n", run<8>(out, cyc, it, 1) / 64);
  printf("   v_add_u32            %6.2f\n", run<9>(out, cyc, it, 1) / 64);
  printf("mix, cycles per PAIR (one v_fma_f64 and one v_add_u32: does the integer instruction fit into the FP64 one's shadow?):\n");
  printf("   fma_f64 + add_u32    %6.2f\n", run<10>(out, cyc, it, 1) / 64);
  printf("v_fma_f64 stream with more wavefronts on the SIMD, cycles per instruction of the SLOWEST wavefront:\n");
  printf("   two per SIMD         %6.2f\n", run<0>(out, cyc, it, 2) / 64);
  printf("   four per SIMD        %6.2f\n", run<0>(out, cyc, it, 4) / 64);
  printf("v_fmac_f64_dpp stream, two / four per SIMD: %6.2f / %6.2f\n", run<3>(out, cyc, it, 2) / 64, run<3>(out, cyc, it, 4) / 64);
  return 0;
}
