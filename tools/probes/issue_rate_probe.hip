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

// round 6: DEPENDENT streams - every FMA into one accumulator, into two, into four (the stream above goes round eight)
#define FMA64D1(k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[0]) : "v"(a), "v"(y));
#define FMA64D2(k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[k % 2]) : "v"(a), "v"(y));
#define FMA64D4(k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[k % 4]) : "v"(a), "v"(y));
// the K row of the record kernel's bounds path (fb_mpc_r16.h, k_bounds): broadcast, product, FMA into ONE accumulator
#define TRI(k)                                                                                                   \
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #k " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(t[k]) : "v"(y)); \
  asm volatile("v_mul_f64 %0, %1, %0" : "+v"(t[k]) : "v"(c[k]));                                                 \
  asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(sacc) : "v"(t[k]), "v"(c[k]));
// the same arithmetic with the eight products formed ahead of the chain
#define TRI_A(k) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #k " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(t[k]) : "v"(y));
#define TRI_B(k) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(t[k]) : "v"(c[k]));
#define TRI_C(k) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(sacc) : "v"(t[k]), "v"(c[k]));
// one LDS round trip: write, wait, read back, wait
#define LDSRT(k) asm volatile("ds_write_b64 %1, %0\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(c[k]) : "v"(laddr) : "memory");

template <int MODE>
__global__ void rate(double* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63, lane2 = lane ^ 5;
  double a = 1.0 + lane * 1e-3, y = 1.0 - lane * 1e-3;
  float fa = 1.0f + lane * 1e-3f, fy = 1.0f - lane * 1e-3f;
  double c[8];
  float f[8];
  int u[8];
  double t[8], sacc = 0.5;
  __shared__ double lbuf[512];
  const unsigned laddr = (unsigned)(unsigned long)(lbuf + threadIdx.x);
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
    else if (MODE == 13) { REP8(FMA64D1) }
    else if (MODE == 14) { REP8(FMA64D2) }
    else if (MODE == 15) { REP8(FMA64D4) }
    else if (MODE == 16) { REP8(TRI) }
    else if (MODE == 17) { for (int g = 0; g < 8; g++) { REP8_(TRI_A) REP8_(TRI_B) REP8_(TRI_C) } }
    else if (MODE == 18) { REP8(LDSRT) }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int k = 0; k < 8; k++) s += c[k] + f[k] + u[k] + (MODE == 16 || MODE == 17 ? t[k] : 0.0);
  s += sacc;
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
  printf("DEPENDENT v_fma_f64 streams, one wavefront per SIMD, cycles per instruction: into ONE accumulator %6.2f, alternating two %6.2f, round four %6.2f (round eight: the first line)\n",
         run<13>(out, cyc, it, 1) / 64, run<14>(out, cyc, it, 1) / 64, run<15>(out, cyc, it, 1) / 64);
  printf("broadcast (v_mov_b64_dpp) -> v_mul_f64 -> v_fmac_f64 into one accumulator, cycles per TRIPLE: as a chain %6.2f, eight products formed ahead of the chain %6.2f\n",
         run<16>(out, cyc, it, 1) / 64, run<17>(out, cyc, it, 1) / 64);
  printf("LDS round trip (ds_write_b64, wait, ds_read_b64, wait), cycles: %6.1f\n", run<18>(out, cyc, it, 1) / 64);
  return 0;
}
