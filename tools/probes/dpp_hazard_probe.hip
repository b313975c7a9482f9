// Developer probe (gfx950): the wait-state rules tools/check_dpp_hazards.py enforces for the hand-written
// v_fmac_f64_dpp instructions, tried on the hardware.  For each rule the producing instruction, N wait states
// (N = 0 .. 6) and the fused instruction sit in ONE asm block, so that nothing but the N s_nop stands between
// them; the result of every lane is compared with the same sequence at 8 wait states.  A row "differs" says
// that the hardware does NOT interlock the pair at that distance (the rule is necessary there); a row of
// "same" says the distance is safe on this chip (the rule may still be what the ISA document requires).
//   rule 1: a VALU instruction writes the register pair that the DPP operand reads      (checker: >= 2)
//   rule 2: a VALU instruction (v_cmpx) writes EXEC                                     (checker: >= 5)
//   rule 3: a transcendental instruction writes the register pair the DPP operand reads (checker: >= 2, and
//           >= 1 for any operand of the fused instruction)
// Build: make -C tools/probes dpp_hazard_probe ; run on the GPU box: tools/probes/dpp_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define FMAC "v_fmac_f64_dpp %[acc], %[yy], %[one] row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"

// (the asm strings must be literals: one kernel body per distance, picked by the preprocessor)
#define RULE1(NAME, WAIT)                                                                        \
  __global__ void NAME(const double* in, double* out) {                                          \
    const int lane = threadIdx.x;                                                                \
    double y = in[lane], acc = 0.25 * lane, yy = -7.0 - lane, one = 1.0, two = 2.0;              \
    asm volatile("s_nop 7\n\t"                                                                   \
                 "v_mul_f64 %[yy], %[y], %[two]\n\t" WAIT FMAC "s_nop 7"                         \
                 : [yy] "+v"(yy), [acc] "+v"(acc)                                                \
                 : [y] "v"(y), [two] "v"(two), [one] "v"(one));                                  \
    out[lane] = acc;                                                                             \
    out[64 + lane] = yy;                                                                         \
  }
// the same with a single-pass producer (v_mov_b64: the FP64 multiply above occupies the pipe for two passes)
#define RULE1B(NAME, WAIT)                                                                       \
  __global__ void NAME(const double* in, double* out) {                                          \
    const int lane = threadIdx.x;                                                                \
    double y = in[lane], acc = 0.25 * lane, yy = -7.0 - lane, one = 1.0;                         \
    asm volatile("s_nop 7\n\t"                                                                   \
                 "v_mov_b64 %[yy], %[y]\n\t" WAIT FMAC "s_nop 7"                                 \
                 : [yy] "+v"(yy), [acc] "+v"(acc)                                                \
                 : [y] "v"(y), [one] "v"(one));                                                  \
    out[lane] = acc;                                                                             \
    out[64 + lane] = yy;                                                                         \
  }
#define RULE3(NAME, WAIT)                                                                        \
  __global__ void NAME(const double* in, double* out) {                                          \
    const int lane = threadIdx.x;                                                                \
    double y = in[lane], acc = 0.25 * lane, yy = -7.0 - lane, one = 1.0;                         \
    asm volatile("s_nop 7\n\t"                                                                   \
                 "v_rcp_f64 %[yy], %[y]\n\t" WAIT FMAC "s_nop 7"                                 \
                 : [yy] "+v"(yy), [acc] "+v"(acc)                                                \
                 : [y] "v"(y), [one] "v"(one));                                                  \
    out[lane] = acc;                                                                             \
    out[64 + lane] = yy;                                                                         \
  }
// EXEC <- lanes 7 .. 15 of every row; the broadcast source (lane 5 of the row) is then switched off
#define RULE2(NAME, WAIT)                                                                        \
  __global__ void NAME(const double* in, double* out) {                                          \
    const int lane = threadIdx.x;                                                                \
    const int r = lane & 15;                                                                     \
    double yy = in[lane], acc = 0.25 * lane, one = 1.0;                                          \
    unsigned long long sv;                                                                       \
    asm volatile("s_mov_b64 %[sv], exec\n\t"                                                     \
                 "s_nop 7\n\t"                                                                   \
                 "v_cmpx_le_i32_e32 vcc, 7, %[r]\n\t" WAIT FMAC "s_nop 7\n\t"                    \
                 "s_mov_b64 exec, %[sv]\n\t"                                                     \
                 "s_nop 7"                                                                       \
                 : [sv] "=&s"(sv), [acc] "+v"(acc)                                               \
                 : [yy] "v"(yy), [one] "v"(one), [r] "v"(r)                                      \
                 : "vcc");                                                                       \
    out[lane] = acc;                                                                             \
    out[64 + lane] = yy;                                                                         \
  }

#define W0 ""
#define W1 "s_nop 0\n\t"
#define W2 "s_nop 1\n\t"
#define W3 "s_nop 2\n\t"
#define W4 "s_nop 3\n\t"
#define W5 "s_nop 4\n\t"
#define W6 "s_nop 5\n\t"
#define W8 "s_nop 7\n\t"

#define ALL(R, P) R(P##0, W0) R(P##1, W1) R(P##2, W2) R(P##3, W3) R(P##4, W4) R(P##5, W5) R(P##6, W6) R(P##8, W8)
ALL(RULE1, r1_)
ALL(RULE1B, r1b_)
ALL(RULE2, r2_)
ALL(RULE3, r3_)

typedef void (*kern_t)(const double*, double*);

static int run_rule(const char* name, kern_t (&k)[8], const double* din, double* dout) {
  const int dist[8] = {0, 1, 2, 3, 4, 5, 6, 8};
  double ref[128], got[128];
  k[7]<<<1, 64>>>(din, dout);
  (void)hipMemcpy(ref, dout, sizeof(ref), hipMemcpyDeviceToHost);
  int first_safe = -1;
  printf("%s\n", name);
  for (int i = 0; i < 7; i++) {
    int worst = 0;
    for (int rep = 0; rep < 200; rep++) {
      k[i]<<<1, 64>>>(din, dout);
      (void)hipMemcpy(got, dout, sizeof(got), hipMemcpyDeviceToHost);
      int bad = 0;
      for (int l = 0; l < 64; l++) bad += memcmp(&got[l], &ref[l], 8) != 0;
      if (bad > worst) worst = bad;
    }
    printf("   %d wait state%s: %s", dist[i], dist[i] == 1 ? " " : "s", worst ? "DIFFERS" : "same as at 8");
    if (worst) printf(" (%d lanes at worst of 200 launches)", worst);
    printf("\n");
    if (!worst && first_safe < 0) first_safe = dist[i];
    if (worst) first_safe = -1;
  }
  printf("   -> every distance from %d on gives the result of 8 wait states\n", first_safe);
  return first_safe;
}

int main() {
  double h[64];
  for (int l = 0; l < 64; l++) h[l] = 1.5 + 0.37 * l;
  double *din, *dout;
  (void)hipMalloc(&din, sizeof(h));
  (void)hipMalloc(&dout, 128 * sizeof(double));
  (void)hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  kern_t k1[8] = {r1_0, r1_1, r1_2, r1_3, r1_4, r1_5, r1_6, r1_8};
  kern_t k1b[8] = {r1b_0, r1b_1, r1b_2, r1b_3, r1b_4, r1b_5, r1b_6, r1b_8};
  kern_t k2[8] = {r2_0, r2_1, r2_2, r2_3, r2_4, r2_5, r2_6, r2_8};
  kern_t k3[8] = {r3_0, r3_1, r3_2, r3_3, r3_4, r3_5, r3_6, r3_8};
  const int s1 = run_rule("rule 1: v_mul_f64 writes the DPP source pair, then v_fmac_f64_dpp (checker requires >= 2)", k1, din, dout);
  const int s1b = run_rule("rule 1, single-pass producer: v_mov_b64 writes the DPP source pair, then v_fmac_f64_dpp (checker requires >= 2)", k1b, din, dout);
  const int s2 = run_rule("rule 2: v_cmpx writes EXEC, then v_fmac_f64_dpp (checker requires >= 5)", k2, din, dout);
  const int s3 = run_rule("rule 3: v_rcp_f64 writes the DPP source pair, then v_fmac_f64_dpp (checker requires >= 2)", k3, din, dout);
  // sanity of the reference itself: rule 1 at 8 wait states must be acc0 + 2 y(lane 5 of the row)
  r1_8<<<1, 64>>>(din, dout);
  double ref[128];
  (void)hipMemcpy(ref, dout, sizeof(ref), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) bad += ref[l] != 0.25 * l + 2.0 * h[(l & ~15) + 5];
  printf("reference sequence (8 wait states) against the arithmetic: %s\n", bad ? "WRONG" : "ok");
  printf("summary: hardware-safe distances %d (%d) / %d / %d against the checker's 2 / 5 / 2: the checker is %s\n", s1, s1b, s2, s3,
         (s1 >= 0 && s1 <= 2 && s1b >= 0 && s1b <= 2 && s2 >= 0 && s2 <= 5 && s3 >= 0 && s3 <= 2 && !bad) ? "at least as strict as this chip needs" : "NOT COVERING what this chip shows");
  return 0;
}
