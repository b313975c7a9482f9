// Developer probe: what SQ_INSTS_VALU counts on gfx950.  Five one-wavefront kernels, each a loop
// of 1000 trips over sixteen instructions of ONE kind (inline assembly, so the count is exact):
// v_fma_f64, v_mov_b64_dpp (row_newbcast), v_readlane_b32, v_cndmask_b32, v_accvgpr_write + read.
// Run under  rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace  and compare the counter of each
// dispatch with 16,000 (+ the loop's own handful): tools/isa_ledger.py sums mnemonics, and its
// total has to be compared with a counter whose definition is not documented per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x

__global__ void k_fma(double* out) {
  double a = threadIdx.x, b = 1.0000001, c = 0.5;
  for (int i = 0; i < 1000; i++) asm volatile(REP16("v_fma_f64 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(b), "v"(c));
  out[threadIdx.x] = a;
}
__global__ void k_dpp(double* out) {
  double a = threadIdx.x, d = 0.0;
  for (int i = 0; i < 1000; i++)
    asm volatile("s_nop 1\n\t" REP16("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") : "+v"(d) : "v"(a));
  out[threadIdx.x] = d;
}
__global__ void k_readlane(double* out) {
  int a = threadIdx.x, s = 0;
  for (int i = 0; i < 1000; i++) asm volatile("s_nop 1\n\t" REP16("v_readlane_b32 %0, %1, 5\n\t") : "+s"(s) : "v"(a));
  out[threadIdx.x] = s;
}
__global__ void k_cndmask(double* out) {
  int a = threadIdx.x, b = 7, d = 0;
  for (int i = 0; i < 1000; i++) asm volatile(REP16("v_cndmask_b32 %0, %1, %2, vcc\n\t") : "+v"(d) : "v"(a), "v"(b) : "vcc");
  out[threadIdx.x] = d;
}
__global__ void k_acc(double* out) {
  int a = threadIdx.x;
  for (int i = 0; i < 1000; i++)
    asm volatile(REP16("v_accvgpr_write_b32 a0, %0\n\ts_nop 1\n\tv_accvgpr_read_b32 %0, a0\n\t") : "+v"(a) : : "a0");
  out[threadIdx.x] = a;
}
int main() {
  double* out;
  (void)hipMalloc(&out, 64 * sizeof(double));
  k_fma<<<1, 64>>>(out);
  k_dpp<<<1, 64>>>(out);
  k_readlane<<<1, 64>>>(out);
  k_cndmask<<<1, 64>>>(out);
  k_acc<<<1, 64>>>(out);
  (void)hipDeviceSynchronize();
  printf("done: expect 16000 per kernel (32000 for the accvgpr pair) plus loop overhead\n");
  return 0;
}
