// Record-based 16-lane MPC policy: FOUR QPs per 64-wide wavefront (one per
// 16-lane DPP row), every array of a QP kept in ONE stage-major, lane-major
// workspace so that each sweep over the horizon touches one contiguous record
// per stage through a single base pointer.
//
// Mathematics: the block form of RiccatiLinearSolver documented at the top of
// fb_row16.h (riccati_linear_solver.cc:77-344), the residual / feasibility /
// variable algebra of full_residual.cc:49-109, full_feasibility.cc:25-88,
// full_variable.cc:47-83 and the implicit block products of mpc_data.cc:17-289.
// Where the data lives and who runs the proximal-level passes:
//
//   * Stage record i of a QP = kSlots slots of 16 doubles, lane r of the row
//     owning element r of every slot:
//       iterate vectors  ez rz zb lb dz wz el rl dl wl+ | v y | vb | yb | dv adz gam rvm
//                        (ez = z - zbar, el = l - lbar: the DISPLACEMENT from the proximal centre, which is
//                        all the sweeps and the line search need of z and l - sigma (z - zbar) - so that the
//                        (zbar lbar) pair stays out of the Newton step and the trial passes; z = zbar + ez
//                        is formed by the passes of the proximal level)
//       constants        f h b                        (mpc_data.cc:240-289)
//       factor record    inv(Lc) (rows and columns folded into one triangle
//                        pair), inv(Pi), t, theta
//     and, in a region of its own, the matrix copy of the stage: row r of
//     [Q S';S R], column r of [E L], row r and column r of [A B], built once per
//     QP, shared between neighbouring stages with identical matrices and kept in
//     LDS while in use.
//     Slots are interleaved in pairs ((s>>1)*32 + 2r + (s&1)): one 16-byte
//     access per lane moves two slots and every memory instruction of a row
//     covers 256 contiguous bytes.  Lanes without an element hold zeros, so the
//     sweeps carry no lane predicates on loads (and smaller problems run padded).
//   * residual, feasibility, norms, the variable updates and the I/O passes are
//     written for this layout too (fused into two passes per proximal
//     iteration) instead of borrowing the workgroup-generic LDS-tile code: they
//     cost a fraction of a Newton step, which is what lets the rows of a
//     wavefront run out of step (Solver::solve_stream).
//   * the barrier Hessian C'Gamma C is accumulated with DPP broadcasts; LDS
//     holds the matrix copy in use (6.5 KB per QP) and transposes inv(Lc)
//     (forward) and C (backward, for A dz) in 2.7 KB more.
//   * the backward sweep forms W'dl = inv(Lc)([A B]'dl) from the already needed
//     [A B]'dl and the row triangle of inv(Lc) instead of reading W by columns.
#pragma once

#include "fb_row16.h"

namespace fbk {


// Batch descriptors as the kernel receives them (one base + stride per array).
struct MpcBatchPtrs {
  const double* base[12];
  long long stride[12];
  int nx, nu, nc;  // the problem's own sizes (<= the kernel instance's)
};
struct VarBatchPtrs {
  double* base[4];
  long long stride[4];
};

// EXACT: the problem has exactly the instance's shape (compile-time strides in
// the passes over the caller's arrays); otherwise it may be smaller and runs
// zero-padded.
// KEEP: instance for FBSTAB_HIP_KEEP_MATRICES (QP q lives in slot q from call to
// call; with `reuse` set the matrix copies of the previous call are still valid).
template <int NX, int NU, int NC, bool EXACT = true, bool KEEP = false, int RQ = 1>
struct MpcR16 {
  typedef CtxRow<RQ> C;
  // lanes per QP (RQ = 1: one 16-lane DPP row, four QPs per wavefront; RQ = 2: a row
  // pair, two QPs per wavefront - fb_row16.h), QPs per wavefront
  static constexpr int LPQ = 16 * RQ, kQpPerWave = 64 / LPQ;
  typedef double dbl2 __attribute__((ext_vector_type(2)));
#ifndef FB_R16_ASM_IMAGES
#define FB_R16_ASM_IMAGES 1
#endif
  static constexpr bool kFusedTrial = true;
  static constexpr bool kOwnVectorOps = true;
  static constexpr int NS = NX + NU;
  static constexpr int NSP = RQ == 1 ? 16 : ((NS + 1) & ~1);  // slots of a row of a stage matrix
  static constexpr int KS = (NC + LPQ - 1) / LPQ;  // constraint slots per lane
  static_assert(NS <= LPQ, "stage width must fit the QP's lanes");

  // ---- stage record ----------------------------------------------------------
  // (DL, WLN): WLN(i) = wl(i + 1), the l-block increment the backward sweep forms at
  // stage i for block i + 1 - kept with dl(i) so that the sweep writes whole pairs;
  // wl(0) = dx(0) is not stored (readers walk the stages upwards and carry it).
  static constexpr int sZ = 0, sRZ = 1, sZB = 2, sLB = 3, sDZ = 4, sWZ = 5, sL = 6, sRL = 7,
                       sDL = 8, sWLN = 9;
  static constexpr int sV = 10;              // pairs (V_s, Y_s)
  static constexpr int sVB = sV + 2 * KS;             // KS slots vbar (every sweep reads them)
  static constexpr int sYB = sVB + ((KS + 1) & ~1);   // KS slots ybar (results only)
  static constexpr int sDV = sYB + ((KS + 1) & ~1);   // pairs (DV_s, ADZ_s)
  // The barrier terms gamma = g0 / mu and rv / mu of the Newton system (riccati_linear_solver.cc:91-99):
  // on the row-pair instances they travel in the record from the forward to the backward sweep; the one-row
  // instances form them AGAIN in the backward sweep from (v, y, vbar), which it reads anyway - 8 slots of
  // traffic per stage less, 90 instructions more.  (Rounds 1-2 measured that at -6 %; with the fused
  // broadcast-FMAs the kernel sits nearer its memory wall and it is +1.1 % pipelined, +1 % one launch at a
  // time - gpurun_out/r05_j; same bits either way: the same function of the same stored values.)
#ifndef FB_R16_RECOMPUTE_GAMMA
#define FB_R16_RECOMPUTE_GAMMA 1  // 0: never; 1: on the one-row instances; 2: everywhere
#endif
  static constexpr bool kStoreGamma = !(FB_R16_RECOMPUTE_GAMMA == 2 || (FB_R16_RECOMPUTE_GAMMA == 1 && RQ == 1));
  static constexpr int sGAM = sDV + 2 * KS;  // (kStoreGamma) pairs (GAM_s, RVM_s)
  static constexpr int sF = sGAM + (kStoreGamma ? 2 * KS : 0), sH = sF + 1;
  static constexpr int sB = sH + 1;          // KS slots
  // factor record, packed: the lower triangles of inv(Lc) (NS (NS + 1) / 2 values,
  // row-major: element (j, c <= j) at j (j + 1) / 2 + c) and of the symmetric inv(Pi)
  // (NX (NX + 1) / 2 values), each cut into slots of 16 consecutive elements - the
  // pipelined solver is bound by the bytes of this record going out in the forward
  // sweep and back in the backward sweep.  Lanes get at their own row and column
  // through a linear image of the triangle in LDS (tri_*).  t rides behind the
  // slots of inv(Lc), theta behind those of inv(Pi): each group is written at
  // one time.
  static constexpr int tri(int j) { return j * (j + 1) / 2; }
  static constexpr int kXTri = tri(NS), kPTri = tri(NX);
  static constexpr int nXs = (kXTri + LPQ - 1) / LPQ, nPs = (kPTri + LPQ - 1) / LPQ;  // slots of each triangle
  static constexpr int fX = (sB + KS + 1) & ~1;          // nXs slots, then t
  static constexpr int fP = fX + ((nXs + 2) & ~1);       // nPs slots, then theta
  static constexpr int kSlots = fP + ((nPs + 2) & ~1);
  static constexpr int kRec = LPQ * kSlots;  // doubles per stage
  // Matrix copy ("pack"), a region of its own: stage i's copy sits at
  // pack + i * kPack.  A stage whose matrices are bitwise identical to the
  // previous stage's reads that stage's copy instead (poff[i], the offset of
  // the copy stage i uses): for a time-invariant plant - the common case in
  // MPC, e.g. OcpGenerator::CopyOverHorizon - the sweeps then stream two or
  // three cache-resident 8 KB copies per QP rather than N + 1 from HBM.
  // Time-varying data simply gets poff[i] = i * kPack.
  static constexpr int pK = 0;                         // row r of [Q S'; S R], 16 slots
  static constexpr int pC = pK + NSP;                  // column r of C = [E L], NC slots
  static constexpr int pABr = pC + ((NC + 1) & ~1);    // row r of [A B], 16 slots
  static constexpr int pABc = pABr + NSP;              // column r of [A B], NX slots
  static constexpr int kPackSlots = (pABc + NX + 1) & ~1;
  static constexpr int kPack = LPQ * kPackSlots;
  // poff[N+1] ints, the flag word and the two scale words of the costate form (below)
  static constexpr long hdr_doubles(int N) { return ((N + 5) / 2 + 15) & ~15L; }
  static constexpr long ws_doubles(int N) {
    return hdr_doubles(N) + (long)(kRec + kPack) * (N + 1) + (2 * (NC - LPQ * (KS - 1)) <= LPQ ? kRec : 0);  // (+ the spare record)
  }

  static constexpr int off(int slot) { return (slot >> 1) * 2 * LPQ + (slot & 1); }

  // LDS of a workgroup (one wavefront):
  //   the matrix-copy area   the copy each QP's current stage uses, minus the [A B] columns (the
  //                          pair-interleaved image of HBM, the QPs' images interleaved pair by pair:
  //                          kPackPair): loaded when poff[i] changes, i.e. three to five times per sweep
  //                          pair for a time-invariant plant, and read by every pass with 16-byte loads;
  //   one region per QP      C as [col][k] (A z products) or the triangle images (sweeps); odd strides;
  //                          behind it the solver loop's parked scalars;
  //   the tables of matrix-copy offsets (lpo).
  static constexpr int CS = NC | 1;
  // the triangle images: inv(Lc) at 0, inv(Pi) behind it, one dump word for the
  // lanes that have no element to write
  static constexpr int kXl = 0, kPl = LPQ * nXs, kDump = kPl + LPQ * nPs;
  static constexpr int TS = (kDump + 16 + LPQ - 1) / LPQ;  // LPQ * TS doubles hold them
  static constexpr int kPackLdsSlots = pABc;        // K, C, [A B] rows
  static constexpr int kLdsDoubles = LPQ * (CS > TS ? CS : TS);  // a QP's own region: transpose buffer / triangle images
  // region stride: 16 doubles mod 32 (bank placement of neighbouring QPs) and at least
  // 24 spare doubles behind the images for the solver loop's parked scalars
  static constexpr int kLdsBase = ((kLdsDoubles + 31) & ~31) + 16;
  static constexpr int kLdsPerRow = kLdsBase - kLdsDoubles >= 24 ? kLdsBase : kLdsBase + 32;
  static constexpr bool kPackInLds = true;
  static constexpr int kPackLds = kPackInLds ? LPQ * kPackLdsSlots : 0;  // doubles of one QP's image
  // The images of a wavefront's QPs share ONE area in front of the QPs' own regions, interleaved pair by
  // pair: pair p of QP q at p * kPackPair + q * 2 LPQ, lane r at + 2 r.  A slot pair of all the wavefront's
  // lanes is then 1 KiB of consecutive LDS in lane order - what an LDS-DMA load writes
  // (global_load_lds_dwordx4: wave-uniform base + lane x 16 bytes; stage_pack_dma below).
  // Round 6 (FB_R16_ABC_FROM_LDS, the default): the images are laid out per QP instead - pair p of QP q at
  // q * kPackQp + p * kPackPair, lane r at + 2 r - with a pair stride of 2 LPQ + 2 doubles: PADDED, so that lane r
  // can read COLUMN r of a row-held matrix out of the image (slot pABr + r, entry j: the lanes' addresses are
  // (r >> 1) kPackPair + (r & 1) + 2 j apart - with the stride a multiple of the 64 banks all sixteen would
  // hit two bank pairs, with 2 LPQ + 2 they fall on 16 different ones).  That is what lets the backward sweep
  // take the columns of [A B] from the rows it has staged anyway instead of reading the 12 column slots of
  // the matrix copy from memory: for a plant whose matrices change with the stage - 31 copies per sweep, a
  // workload at the HBM's limit with eight launches in flight - 12 of the 116 slots a stage pair moves.
  // (The interleaved layout stays behind FB_R16_PACK_DMA: the LDS-DMA load writes a wavefront's lanes in order.)
#ifndef FB_R16_ABC_FROM_LDS
#define FB_R16_ABC_FROM_LDS 1
#endif
#ifndef FB_R16_PACK_DMA
#define FB_R16_PACK_DMA 0
#endif
  static constexpr bool kAbcFromLds = FB_R16_ABC_FROM_LDS != 0 && FB_R16_PACK_DMA == 0 && kPackInLds;
  static constexpr int kPackPair = kAbcFromLds ? 2 * LPQ + 2 : 2 * LPQ * kQpPerWave;  // doubles between consecutive slot pairs of a QP's image
  // The rows of [A B] are NX: lanes r >= NX hold zeros in those slot pairs (and the sweeps rely on reading them:
  // W, Pi+ and T come out zero there without a select).  Where it buys a workgroup per CU the pairs of [A B] are
  // therefore TRIMMED (FB_R16_TRIM_AB, round 6): 2 NX + 2 doubles apart, lanes r < NX at + 2 r and the two
  // doubles of padding - kept zero: every lane r >= NX stores its zeros there and reads them back from there -
  // at + 2 NX.  <24,8,16>: 57,600 -> 53,504 bytes (N = 30), three workgroups per CU (160 KB) instead of two; the column
  // reads of the backward sweep stay conflict-free (pair stride 50 doubles: 100 dwords = 36 mod 64, sixteen
  // lane pairs on sixteen different bank quads).  Row-pair instances only, and only where a workgroup is gained:
  // <24,8,32> stays at two, <18,5,10> has its four - no other instance's code changes.
#ifndef FB_R16_TRIM_AB
#define FB_R16_TRIM_AB 1
#endif
  static constexpr int kAbPairTrim = 2 * NX + 2;
  static constexpr int pack_qp(int ab_pair) { return (pABr / 2) * kPackPair + (NSP / 2) * ab_pair; }
  static constexpr int wgs_per_cu(int pack_qp_doubles) {  // by LDS (160 KB a CU), at most the four SIMDs' one wavefront each
    const int per = 163840 / ((kQpPerWave * (pack_qp_doubles + kLdsPerRow)) * 8 + 512);  // (+ the offset tables: 8 (N + 1) bytes a QP)
    return per > 4 ? 4 : per;
  }
  static constexpr bool kTrimAb = FB_R16_TRIM_AB != 0 && kAbcFromLds && RQ == 2 && NX < LPQ &&
                                  wgs_per_cu(pack_qp(kAbPairTrim)) > wgs_per_cu(pack_qp(kPackPair));
  static constexpr int kAbPair = kTrimAb ? kAbPairTrim : kPackPair;  // doubles between consecutive slot pairs of [A B]'s rows
  // ... and where leaving the rows of K = [Q S'; S R] OUT of the image buys another workgroup they are read from the
  // matrix copy in memory instead (FB_R16_K_FROM_MEMORY, round 6: the `if constexpr (kKinLds)` at the five places K is read).  A sweep stages a stage's copy once and
  // reads K from it once, so a plant whose matrices change with the stage moves the same bytes either way, and a
  // time-invariant one finds its few copies in L2.  <24,8,16>: 53,504 -> 36,608 bytes, FOUR workgroups per CU - every
  // SIMD; <24,8,32>: 68,608 -> 51,712, three instead of two.
#ifndef FB_R16_K_FROM_MEMORY
#define FB_R16_K_FROM_MEMORY 1
#endif
  static constexpr bool kKinLds = !(FB_R16_K_FROM_MEMORY != 0 && kAbcFromLds && RQ == 2 &&
                                    wgs_per_cu(pack_qp(kAbPair) - (pC / 2) * kPackPair) > wgs_per_cu(pack_qp(kAbPair)));
  static constexpr int kLdsFirstPair = kKinLds ? 0 : pC / 2;  // the image starts with this slot pair of the matrix copy
  // where slot pair `pr` of a QP's image starts, in doubles
  static constexpr int pair_at(int pr) {
    return pr < pABr / 2 ? (pr - kLdsFirstPair) * kPackPair : (pABr / 2 - kLdsFirstPair) * kPackPair + (pr - pABr / 2) * kAbPair;
  }
  // this lane's place in the pairs of [A B] relative to its place in the others (+ 2 r): lanes r >= NX share the padding
  static FB_DEV int ab_lane_shift() {
    if constexpr (kTrimAb) {
      const int r = threadIdx.x & (LPQ - 1);
      return r > NX ? 2 * (NX - r) : 0;
    } else {
      return 0;
    }
  }
  static constexpr int kPackQp = kAbcFromLds ? pack_qp(kAbPair) - kLdsFirstPair * kPackPair : 2 * LPQ;  // doubles between the images of two QPs
  static constexpr int kPackArea = kAbcFromLds ? kQpPerWave * kPackQp : kQpPerWave * kPackLds;  // doubles of the wavefront's area
  // this lane's view of the matrix copy in use
  typedef typename std::conditional<kPackInLds, lds_ptr, const double*>::type pk_ptr;


  // ---- the triangle images in LDS, hand-scheduled (instance <12,4,20>, one row per QP) ----
  // Element (j, c) of a packed lower triangle sits at tri(j) + c of a linear image; lane r
  // writes its column (elements (j, r), j >= r) or row (elements (r, c), c <= r) and
  // reads them back the other way round.  Left to the compiler each of these predicated
  // accesses is five to fifteen instructions (address select against a dump word, masks
  // read back from spilled scalars with v_readlane, the LDS base fetched from an AGPR
  // every time, a branch round every store) and the reads come back one at a time, each
  // behind its own s_waitcnt because their registers are reused.  Here the predicate
  // lives in EXEC: the columns are walked in the order in which the set of lanes that
  // take part only shrinks, so ONE v_cmpx per element narrows EXEC and ONE ds
  // instruction with an immediate offset does the access; all reads of a block are in
  // flight together and waited for once.  EXEC is restored before the block ends, and
  // five wait states follow (the compiler's hazard recognizer does not look inside the
  // block: a DPP instruction of its own must not sit in the shadow of the last v_cmpx).
  static constexpr bool kAsmImages = FB_R16_ASM_IMAGES && RQ == 1 && NS == 16 && NX == 12;
  // The row-pair instances (stages up to 32 wide) SUBSTITUTE with Lc where the one-row instances multiply
  // with its explicit inverse (fb_row16.h, subst_rows): the factor record then holds Lc itself - rows of
  // the packed triangle, the diagonal as its reciprocal - and t = inv(Lc) g, s = t - inv(Lc) u and
  // [dx; du] = inv(Lc)' s are the reference's three triangular solves (riccati_linear_solver.cc:241-249,
  // :299-325 do them blockwise with M and SG).  On 16-wide stages the explicit inverse stays two orders
  // under the tolerance (and is what the headline's time is made of); on stages wider than that, with
  // nx > N nu, it left up to 4.6e-6 where the oracle leaves 1e-7 - the one deviation round 4's fuzz found
  // (tests/test_gpu_components.py::test_one_step_qp_...).
#ifndef FB_R16_SUBST
#define FB_R16_SUBST 2  // substitute on the instances with at least this many rows per QP (1: all, 3: none)
#endif
  static constexpr bool kSubst = RQ >= FB_R16_SUBST;
  static_assert(!(kSubst && kAsmImages), "the hand-written image blocks move the inverse's columns");
  static FB_DEV unsigned lds_addr(lds_ptr p) { return (unsigned)(unsigned long)p; }
#define FB_IMG_WP(cc, off) "v_cmpx_le_i32_e32 vcc, " #cc ", %[ro]\n\tds_write_b64 %[rb], %[p" #cc "] offset:" #off "\n\t"
#define FB_IMG_WX(j, off) "v_cmpx_ge_i32_e32 vcc, " #j ", %[ro]\n\tds_write_b64 %[rb], %[c" #j "] offset:" #off "\n\t"
#define FB_IMG_RR(j, off) "v_cmpx_le_i32_e32 vcc, " #j ", %[ro]\n\tds_read_b64 %[r" #j "], %[rbr] offset:" #off "\n\t"
#define FB_IMG_RC(j, off) "v_cmpx_ge_i32_e32 vcc, " #j ", %[ro]\n\tds_read_b64 %[c" #j "], %[rbc] offset:" #off "\n\t"
#define FB_IMG_RS(s, off) "ds_read_b64 %[s" #s "], %[rbc] offset:" #off "\n\t"
#define FB_IMG_RPB(cc, off) "ds_read_b64 %[p" #cc "], %[rbpc] offset:" #off "\n\t"
#define FB_IMG_RPA(cc, off) "v_cmpx_le_i32_e32 vcc, " #cc ", %[ro]\n\tds_read_b64 %[p" #cc "], %[rbpr] offset:" #off "\n\t"
  // Tr[kPl + tri(r) + cc] = Pv[cc] for cc <= r, lanes r < 12 (forward sweep)
  static FB_DEV void img_write_pinv(lds_ptr row_r, int ro, const double (&Pv)[12]) {
    unsigned long long sv;
    const unsigned rb = lds_addr(row_r);
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_gt_i32_e32 vcc, 12, %[ro]\n\t"
        FB_IMG_WP(0, 0) FB_IMG_WP(1, 8) FB_IMG_WP(2, 16) FB_IMG_WP(3, 24) FB_IMG_WP(4, 32) FB_IMG_WP(5, 40)
        FB_IMG_WP(6, 48) FB_IMG_WP(7, 56) FB_IMG_WP(8, 64) FB_IMG_WP(9, 72) FB_IMG_WP(10, 80) FB_IMG_WP(11, 88)
        "s_mov_b64 exec, %[sv]\n\t"
        "s_nop 4"
        : [sv] "=&s"(sv)
        : [rb] "v"(rb), [ro] "v"(ro), [p0] "v"(Pv[0]), [p1] "v"(Pv[1]), [p2] "v"(Pv[2]), [p3] "v"(Pv[3]),
          [p4] "v"(Pv[4]), [p5] "v"(Pv[5]), [p6] "v"(Pv[6]), [p7] "v"(Pv[7]), [p8] "v"(Pv[8]), [p9] "v"(Pv[9]),
          [p10] "v"(Pv[10]), [p11] "v"(Pv[11])
        : "memory", "vcc");
  }
  // Tr[kXl + tri(j) + r] = XC[j] for j >= r (forward sweep); col_r = Tr + kXl + r
  static FB_DEV void img_write_x(lds_ptr col_r, int ro, const double (&XC)[16]) {
    unsigned long long sv;
    const unsigned rb = lds_addr(col_r);
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        FB_IMG_WX(15, 960) FB_IMG_WX(14, 840) FB_IMG_WX(13, 728) FB_IMG_WX(12, 624) FB_IMG_WX(11, 528)
        FB_IMG_WX(10, 440) FB_IMG_WX(9, 360) FB_IMG_WX(8, 288) FB_IMG_WX(7, 224) FB_IMG_WX(6, 168) FB_IMG_WX(5, 120)
        FB_IMG_WX(4, 80) FB_IMG_WX(3, 48) FB_IMG_WX(2, 24) FB_IMG_WX(1, 8) FB_IMG_WX(0, 0)
        "s_mov_b64 exec, %[sv]\n\t"
        "s_nop 4"
        : [sv] "=&s"(sv)
        : [rb] "v"(rb), [ro] "v"(ro), [c0] "v"(XC[0]), [c1] "v"(XC[1]), [c2] "v"(XC[2]), [c3] "v"(XC[3]),
          [c4] "v"(XC[4]), [c5] "v"(XC[5]), [c6] "v"(XC[6]), [c7] "v"(XC[7]), [c8] "v"(XC[8]), [c9] "v"(XC[9]),
          [c10] "v"(XC[10]), [c11] "v"(XC[11]), [c12] "v"(XC[12]), [c13] "v"(XC[13]), [c14] "v"(XC[14]),
          [c15] "v"(XC[15])
        : "memory", "vcc");
  }
  // XR[j] = Tr[kXl + tri(r) + j] for j <= r, zero beyond (XR arrives zeroed), and the
  // nine packed slots Xp[s] = Tr[kXl + 16 s + r] (forward sweep)
  static FB_DEV void img_read_xrow_slots(lds_ptr row_r, lds_ptr col_r, int ro, double (&XR)[16], double (&Xp)[10]) {
    unsigned long long sv;
    const unsigned rbr = lds_addr(row_r), rbc = lds_addr(col_r);
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        FB_IMG_RS(0, 0) FB_IMG_RS(1, 128) FB_IMG_RS(2, 256) FB_IMG_RS(3, 384) FB_IMG_RS(4, 512) FB_IMG_RS(5, 640)
        FB_IMG_RS(6, 768) FB_IMG_RS(7, 896) FB_IMG_RS(8, 1024)
        FB_IMG_RR(0, 0) FB_IMG_RR(1, 8) FB_IMG_RR(2, 16) FB_IMG_RR(3, 24) FB_IMG_RR(4, 32) FB_IMG_RR(5, 40)
        FB_IMG_RR(6, 48) FB_IMG_RR(7, 56) FB_IMG_RR(8, 64) FB_IMG_RR(9, 72) FB_IMG_RR(10, 80) FB_IMG_RR(11, 88)
        FB_IMG_RR(12, 96) FB_IMG_RR(13, 104) FB_IMG_RR(14, 112) FB_IMG_RR(15, 120)
        "s_mov_b64 exec, %[sv]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_nop 4"
        : [sv] "=&s"(sv), [r0] "+v"(XR[0]), [r1] "+v"(XR[1]), [r2] "+v"(XR[2]), [r3] "+v"(XR[3]), [r4] "+v"(XR[4]),
          [r5] "+v"(XR[5]), [r6] "+v"(XR[6]), [r7] "+v"(XR[7]), [r8] "+v"(XR[8]), [r9] "+v"(XR[9]),
          [r10] "+v"(XR[10]), [r11] "+v"(XR[11]), [r12] "+v"(XR[12]), [r13] "+v"(XR[13]), [r14] "+v"(XR[14]),
          [r15] "+v"(XR[15]), [s0] "=&v"(Xp[0]), [s1] "=&v"(Xp[1]), [s2] "=&v"(Xp[2]), [s3] "=&v"(Xp[3]),
          [s4] "=&v"(Xp[4]), [s5] "=&v"(Xp[5]), [s6] "=&v"(Xp[6]), [s7] "=&v"(Xp[7]), [s8] "=&v"(Xp[8])
        : [rbr] "v"(rbr), [rbc] "v"(rbc), [ro] "v"(ro)
        : "memory", "vcc");
  }
  // Backward sweep: row r and column r of inv(Lc), row r of the symmetric inv(Pi), one
  // block each, requested just before their products (all three at once are 88 registers
  // that the sweep does not have: the allocator answered with a hundred AGPR copies).
  // The destinations arrive zeroed.  XR[j] = Tr[kXl + tri(r) + j], j <= r.
  static FB_DEV void img_read_xrow(lds_ptr xrow_r, int ro, double (&XR)[16]) {
    unsigned long long sv;
    const unsigned rbr = lds_addr(xrow_r);
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        FB_IMG_RR(0, 0) FB_IMG_RR(1, 8) FB_IMG_RR(2, 16) FB_IMG_RR(3, 24) FB_IMG_RR(4, 32) FB_IMG_RR(5, 40)
        FB_IMG_RR(6, 48) FB_IMG_RR(7, 56) FB_IMG_RR(8, 64) FB_IMG_RR(9, 72) FB_IMG_RR(10, 80) FB_IMG_RR(11, 88)
        FB_IMG_RR(12, 96) FB_IMG_RR(13, 104) FB_IMG_RR(14, 112) FB_IMG_RR(15, 120)
        "s_mov_b64 exec, %[sv]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_nop 4"
        : [sv] "=&s"(sv), [r0] "+v"(XR[0]), [r1] "+v"(XR[1]), [r2] "+v"(XR[2]), [r3] "+v"(XR[3]), [r4] "+v"(XR[4]),
          [r5] "+v"(XR[5]), [r6] "+v"(XR[6]), [r7] "+v"(XR[7]), [r8] "+v"(XR[8]), [r9] "+v"(XR[9]),
          [r10] "+v"(XR[10]), [r11] "+v"(XR[11]), [r12] "+v"(XR[12]), [r13] "+v"(XR[13]), [r14] "+v"(XR[14]),
          [r15] "+v"(XR[15])
        : [rbr] "v"(rbr), [ro] "v"(ro)
        : "memory", "vcc");
  }
  // XC[j] = Tr[kXl + tri(j) + r], j >= r
  static FB_DEV void img_read_xcol(lds_ptr xcol_r, int ro, double (&XC)[16]) {
    unsigned long long sv;
    const unsigned rbc = lds_addr(xcol_r);
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        FB_IMG_RC(15, 960) FB_IMG_RC(14, 840) FB_IMG_RC(13, 728) FB_IMG_RC(12, 624) FB_IMG_RC(11, 528)
        FB_IMG_RC(10, 440) FB_IMG_RC(9, 360) FB_IMG_RC(8, 288) FB_IMG_RC(7, 224) FB_IMG_RC(6, 168) FB_IMG_RC(5, 120)
        FB_IMG_RC(4, 80) FB_IMG_RC(3, 48) FB_IMG_RC(2, 24) FB_IMG_RC(1, 8) FB_IMG_RC(0, 0)
        "s_mov_b64 exec, %[sv]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_nop 4"
        : [sv] "=&s"(sv), [c0] "+v"(XC[0]), [c1] "+v"(XC[1]), [c2] "+v"(XC[2]), [c3] "+v"(XC[3]), [c4] "+v"(XC[4]),
          [c5] "+v"(XC[5]), [c6] "+v"(XC[6]), [c7] "+v"(XC[7]), [c8] "+v"(XC[8]), [c9] "+v"(XC[9]),
          [c10] "+v"(XC[10]), [c11] "+v"(XC[11]), [c12] "+v"(XC[12]), [c13] "+v"(XC[13]), [c14] "+v"(XC[14]),
          [c15] "+v"(XC[15])
        : [rbc] "v"(rbc), [ro] "v"(ro)
        : "memory", "vcc");
  }
  // Pv[cc] = Tr[kPl + (cc <= r ? tri(r) + cc : tri(cc) + r)] on the lanes r < 12 (the
  // second form is read first, the first over it where cc <= r: LDS operations of a
  // wavefront complete in order)
  static FB_DEV void img_read_pinv(lds_ptr prow_r, lds_ptr pcol_r, int ro, double (&Pv)[12]) {
    unsigned long long sv;
    const unsigned rbpr = lds_addr(prow_r), rbpc = lds_addr(pcol_r);
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_gt_i32_e32 vcc, 12, %[ro]\n\t"
        FB_IMG_RPB(0, 0) FB_IMG_RPB(1, 8) FB_IMG_RPB(2, 24) FB_IMG_RPB(3, 48) FB_IMG_RPB(4, 80) FB_IMG_RPB(5, 120)
        FB_IMG_RPB(6, 168) FB_IMG_RPB(7, 224) FB_IMG_RPB(8, 288) FB_IMG_RPB(9, 360) FB_IMG_RPB(10, 440)
        FB_IMG_RPB(11, 528)
        FB_IMG_RPA(0, 0) FB_IMG_RPA(1, 8) FB_IMG_RPA(2, 16) FB_IMG_RPA(3, 24) FB_IMG_RPA(4, 32) FB_IMG_RPA(5, 40)
        FB_IMG_RPA(6, 48) FB_IMG_RPA(7, 56) FB_IMG_RPA(8, 64) FB_IMG_RPA(9, 72) FB_IMG_RPA(10, 80) FB_IMG_RPA(11, 88)
        "s_mov_b64 exec, %[sv]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_nop 4"
        : [sv] "=&s"(sv), [p0] "+v"(Pv[0]), [p1] "+v"(Pv[1]), [p2] "+v"(Pv[2]), [p3] "+v"(Pv[3]), [p4] "+v"(Pv[4]),
          [p5] "+v"(Pv[5]), [p6] "+v"(Pv[6]), [p7] "+v"(Pv[7]), [p8] "+v"(Pv[8]), [p9] "+v"(Pv[9]),
          [p10] "+v"(Pv[10]), [p11] "+v"(Pv[11])
        : [rbpr] "v"(rbpr), [rbpc] "v"(rbpc), [ro] "v"(ro)
        : "memory", "vcc");
  }

  // ---- state -------------------------------------------------------------------
  double* rec;   // this row's records, lane offset included
  double* pack;  // this row's matrix copies, lane offset included
  int* poff;     // per stage: offset (doubles) of the copy it reads (the slot's own table, kept
                 // between calls for FBSTAB_HIP_KEEP_MATRICES)
  // ... and its working copy in LDS: the passes read it every stage, and a global
  // load there would make each stage wait for every memory operation in flight
  // (loads return in order: the wait for the offset drains the prefetched records
  // and the stores of the previous stage with it).
  typedef FB_LDS int* lds_iptr;
  lds_iptr lpo;
  // ints of LDS per row for the table (N + 1 offsets and the flag word)
  static constexpr int lpo_ints(int N) { return (N + 3) & ~1; }
  int lds_off;   // offset of the copy currently in LDS (-1: none)
  bool reuse = false;  // (KEEP instances) the slot's matrix copies are those of this QP already
  // Every constraint row of every stage has at most one nonzero (bounds on single
  // states / inputs, the usual MPC constraints): C'Gamma C is then diagonal and
  // the forward sweep adds it as such.  The skipped products are exact zeros, so
  // the result equals the general path bit for bit.  Detected while packing.
  bool bounds = false;
  // The costate step dl(i) of the backward sweep, two forms of the same quantity:
  //   (a) dl(i) = -inv(Pi_i)(theta_i + dx_i), the reference's (riccati_linear_solver.cc:321-325);
  //   (b) dl(i) from the state rows of the Newton system's first block row
  //       (abstract_components.h:276-288),
  //         dl(i) = [(H + sigma I) dz(i)]_x + [A B]'dl(i+1) + [A'dv(i)]_x + rz_x + sigma (z - zbar)_x,
  //       which the pair (dz, dl) of (a) satisfies identically.
  // (b) needs neither inv(Pi) nor theta in the backward sweep: they stay out of the record
  // (6 slots written and 6 read per stage and Newton step) and the sweeps lose the LDS image
  // of inv(Pi) - 8 % more QP/s on the BASELINE workload.  Its rounding error is that of the
  // sum's largest terms, eps |K| |dz| with |K| <= |H| + 1/sigma + |C'Gamma C| and
  // Gamma <= 1/sigma, against eps |inv(Pi)| |theta + dx| <= eps |dz| / sigma for (a): the
  // same class as long as |H| and the column sums of C'C stay within a small factor of
  // 1/sigma and 1.  That is what `rowdl` records (choose_costate_form): bound constraints
  // with entries of order one - the usual MPC constraints - and rows with a few entries of
  // order one (nonzeros per row x column sum of C'C <= 8) take (b), everything else (a);
  // e.g. the reference's servo-motor problem (output constraint rows with entries 1280,
  // |C'Gamma C| ~ 1e6 / sigma) keeps (a), where (b) was measured to shift an iteration
  // count.  Both forms give the oracle's iteration counts on every test problem they
  // serve.
  bool rowdl = false;
  float cmax2 = 0.f, hmax = 0.f;  // max_r sum_k C[k][r]^2 and max_r sum_c |H[r][c]| over the stages (load_guess)
  int nzmax = 0;                  // the largest number of nonzeros in a constraint row of any stage (load_guess)
  lds_ptr lds;  // this QP's own LDS region (transpose buffer / triangle images, parked scalars)
  lds_ptr lpk;  // this QP's image in the wavefront's matrix-copy area (its lanes' offsets not included)
  const MpcBatchPtrs* data;  // kernel arguments (uniform)
  const VarBatchPtrs* var;
  long q;  // QP index
  int N;
  // The problem's own sizes, nx <= NX, nu <= NU, nc <= NC: a smaller problem runs
  // zero-padded (states in lanes [0, nx), inputs in lanes [NX, NX + nu),
  // constraints k < nc).  Padded rows and columns are exact zeros everywhere;
  // their pivots are sigma (+ 1/sigma), their steps are zero, and only the passes
  // that touch the caller's arrays (load_guess, write_out, forcing_norm, the
  // test probe) know about it.  Read from the launch descriptor where needed
  // rather than held in registers across the sweeps.
  FB_DEV int prob_nx() const { return EXACT ? NX : data->nx; }
  FB_DEV int prob_nu() const { return EXACT ? NU : data->nu; }
  FB_DEV int prob_nc() const { return EXACT ? NC : data->nc; }
  // Step length of an accepted but not yet applied Newton step (0 = none); the
  // next forward sweep applies it stage by stage, everything else flushes first.
  double pend_t;

  FB_DEV void bind(double* ws_row, lds_ptr lds_row, lds_ptr lpk_row, lds_iptr lpo_row, const MpcBatchPtrs* d,
                   const VarBatchPtrs* x, long q_, int N_, int lane) {
    lpk = lpk_row;
    lds_off = -1;
    poff = reinterpret_cast<int*>(ws_row);
    lpo = lpo_row;
    pack = ws_row + hdr_doubles(N_) + 2 * lane;
    rec = pack + (long)kPack * (N_ + 1);
    lds = lds_row;
    data = d;
    var = x;
    q = q_;
    N = N_;
    pend_t = 0.0;
  }
  // A row that never gets a QP (batch < row slots of the grid) still takes part in the
  // cooperative passes of its wavefront - with its own LDS region as scratch
  // (close_subproblem_coop) - so what those read is set before the first fetch.
  FB_DEV void bind_idle(lds_ptr lds_row, lds_ptr lpk_row, lds_iptr lpo_row, int N_) {
    lpk = lpk_row;
    lds_off = -1;
    poff = nullptr;
    lpo = lpo_row;
    pack = rec = nullptr;
    lds = lds_row;
    data = nullptr;
    var = nullptr;
    q = -1;
    N = N_;
    pend_t = 0.0;
  }
  FB_DEV int num_primal_dual() const { return (N + 1) * (2 * prob_nx() + prob_nu() + prob_nc()); }

  // ---- record access -------------------------------------------------------------
  static FB_DEV double ld(const double* R, int slot) { return R[off(slot)]; }
  static FB_DEV void st(double* R, int slot, double v) { R[off(slot)] = v; }
  static FB_DEV dbl2 ld2(const double* R, int even_slot) {
    return *reinterpret_cast<const dbl2*>(R + off(even_slot));
  }
  static FB_DEV void st2(double* R, int even_slot, double a, double b) {
    dbl2 t = {a, b};
    *reinterpret_cast<dbl2*>(R + off(even_slot)) = t;
  }
  // The LAST constraint slot of a lane holds an entry on the first kTail lanes only (NC = 20: 4 of 16).  The
  // other lanes' entries are zero from load_guess() on and stay zero - every pass computes zeros there.  In
  // the sweeps and the trial pass those lanes therefore take the slot from ONE spare record per QP (record
  // N + 1, zeroed by load_guess(), cache-resident) instead of the stage's own: the same instructions, the same
  // values in every lane bit for bit, and three quarters of the slot's bytes stay out of HBM.  (With the
  // lanes masked out of the accesses instead - `if (tail_lane())` - the counters showed the same bytes and
  // the headline lost 7 %: five conditional regions per stage pair cut the sweeps' basic blocks and the
  // waits at their joins, gpurun_out/r05_t1.)
#ifndef FB_R16_TAIL_CUT
#define FB_R16_TAIL_CUT 1
#endif
  static constexpr int kTail = NC - LPQ * (KS - 1);
  static constexpr bool kTailCut = FB_R16_TAIL_CUT != 0 && 2 * kTail <= LPQ;
  static constexpr int kSpareRecords = kTailCut ? 1 : 0;
  static FB_DEV bool tail_lane() { return (int)(threadIdx.x & (LPQ - 1)) < kTail; }
  // where this lane finds the last constraint slot of the stage whose record is R (Rd: the spare record)
  template <class P>
  static FB_DEV P tail_of(P R, P Rd) {
    if constexpr (kTailCut) return tail_lane() ? R : Rd;
    else return R;
  }
  static FB_DEV void zero_spare(double* R0, int N_) {
    if constexpr (kTailCut) {
      double* const Rd = spare_of(R0, N_);
      st2(Rd, sV + 2 * (KS - 1), 0.0, 0.0);
      st2(Rd, sDV + 2 * (KS - 1), 0.0, 0.0);
      if constexpr (kStoreGamma) st2(Rd, sGAM + 2 * (KS - 1), 0.0, 0.0);
    }
  }
  static FB_DEV const double* spare_of(const double* R0, int N_) { return R0 + (long)(N_ + 1) * kRec; }
  static FB_DEV double* spare_of(double* R0, int N_) { return R0 + (long)(N_ + 1) * kRec; }
  // slots [S0, S0 + CNT) into out[0..CNT); S0 even
  template <int S0, int CNT, int NOUT>
  static FB_DEV void ldv(const double* R, double (&out)[NOUT]) {
    static_assert((S0 & 1) == 0, "slot ranges start on a pair");
    sfor<0, (CNT + 1) / 2>([&](auto P_) {
      constexpr int pr = decltype(P_)::value;
      const dbl2 t = *reinterpret_cast<const dbl2*>(R + (S0 / 2 + pr) * 2 * LPQ);
      out[2 * pr] = t[0];
      if constexpr (2 * pr + 1 < CNT) out[2 * pr + 1] = t[1];
    });
  }
  template <int S0, int CNT, int NIN>
  static FB_DEV void stv(double* R, const double (&in)[NIN]) {
    static_assert((S0 & 1) == 0, "slot ranges start on a pair");
    sfor<0, CNT / 2>([&](auto P_) {
      constexpr int pr = decltype(P_)::value;
      dbl2 t = {in[2 * pr], in[2 * pr + 1]};
      *reinterpret_cast<dbl2*>(R + (S0 / 2 + pr) * 2 * LPQ) = t;
    });
    if constexpr (CNT & 1) R[(S0 / 2 + CNT / 2) * 2 * LPQ] = in[CNT - 1];
  }

  // Makes the matrix copy at offset `off` the one resident in LDS (row-uniform).
  template <bool L = kPackInLds>
  FB_DEV auto pack_view(const C& c) const {
    if constexpr (L) return lpk + 2 * c.tid;
    else return static_cast<const double*>(pack);
  }
  FB_DEV void stage_pack(const C& c, pk_ptr& view, int off) { stage_pack_s(c, pack, view, lds_off, off); }
  static FB_DEV void stage_pack_s(const C&, const double* pack0, const double*& view, int& cur, int off) {
    view = pack0 + off;
    cur = off;
  }
  static FB_DEV void stage_pack_s(const C& c, const double* pack0, lds_ptr& view, int& cur, int off) {
    if (off == cur) return;
    cur = off;
    lds_ptr dst = view;
    const double* src = pack0 + off;
    c.sync();  // earlier readers of the previous copy
    [[maybe_unused]] const int ab = ab_lane_shift();
    constexpr int kPairs = kPackLdsSlots / 2, kChunk = 13;
    sfor<0, (kPairs - kLdsFirstPair + kChunk - 1) / kChunk>([&](auto Ch) {
      constexpr int c0 = kLdsFirstPair + decltype(Ch)::value * kChunk;
      constexpr int cn = c0 + kChunk < kPairs ? kChunk : kPairs - c0;
      dbl2 t[kChunk];
      sfor<0, cn>([&](auto I) { t[decltype(I)::value] = *reinterpret_cast<const dbl2*>(src + (c0 + decltype(I)::value) * 2 * LPQ); });
      sfor<0, cn>([&](auto I) {
        constexpr int pr = c0 + decltype(I)::value;
        // (kTrimAb: in the pairs of [A B] the lanes r >= NX - zeros in the matrix copy - all store to the padding)
        *reinterpret_cast<FB_LDS dbl2*>(dst + pair_at(pr) + (pr >= pABr / 2 ? ab : 0)) = t[decltype(I)::value];
      });
    });
    c.sync();
  }
  // The same by LDS-DMA (round 5; build knob FB_R16_PACK_DMA, OFF): global_load_lds_dwordx4 moves 16 bytes
  // per lane from the lane's own address to LDS at a wave-uniform base + lane x 16 - one slot pair of the
  // wavefront's area per instruction, no register in between, nothing waits.  The sweeps of the Newton step
  // issue it for the NEXT stage's copy as soon as the current stage has read the image for the last time
  // and wait for it at the top of the next stage, so that a plant whose matrices change from stage to
  // stage (31 stagings per sweep instead of three to five) does not pay a trip to memory and back per
  // stage.  Built, correct (same checksums, 161 parity tests) - and measured: the time-varying workload
  // +1.8 % one launch at a time, +-0 with eight in flight (427-432 k either way), the headline +-0
  // (gpurun_out/r05_v).  That workload is bound by the BYTES of its 31 matrix copies per QP (LABNOTES
  // Part II), which a different way of fetching them does not change.  Left in as a knob, not the default.
  // Lp: this lane's view; the lanes of a QP whose copy does not change sit the call out (EXEC).
#ifndef FB_R16_PACK_DMA
#define FB_R16_PACK_DMA 0
#endif
  static constexpr bool kPackDma = FB_R16_PACK_DMA != 0 && kPackInLds;
  static FB_DEV void stage_pack_dma(const double* src, lds_ptr Lp) {
    // (the area's base: the view minus the lane's place in the wavefront - the same in every lane)
    FB_LDS char* const area = reinterpret_cast<FB_LDS char*>(Lp - 2 * (threadIdx.x & 63));
    sfor<0, kPackLdsSlots / 2>([&](auto I) {
      constexpr int pr = decltype(I)::value;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pr * 2 * LPQ),
                                       (FB_LDS void*)(area + pr * kPackPair * 8), 16, 0, 0);
    });
  }
  // ... and the wait for it: every lane of the wavefront calls this at the top of a stage; `issued` says
  // whether this lane's QP had a copy on its way
  static FB_DEV void stage_pack_dma_wait(bool issued) {
    if (__ballot(issued) != 0ull) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }

  // slots [S0, S0 + CNT) of the LDS-resident copy into out[0..CNT)
  template <int S0, int CNT, int NOUT>
  static FB_DEV void ldl(const double* G, double (&out)[NOUT]) {
    ldv<S0, CNT>(G, out);
  }
  template <int S0, int CNT, int NOUT>
  static FB_DEV void ldl(lds_ptr L, double (&out)[NOUT]) {
    static_assert((S0 & 1) == 0 && S0 + CNT <= kPackLdsSlots, "inside the LDS image, on a pair");
    static_assert(S0 >= pABr || S0 + CNT <= pABr, "a range lies on one side of the first [A B] pair");
    static_assert(S0 / 2 >= kLdsFirstPair, "the rows of K are not in this instance's image");
    if constexpr (kTrimAb && S0 >= pABr) L += ab_lane_shift();
    sfor<0, (CNT + 1) / 2>([&](auto P_) {
      constexpr int pr = decltype(P_)::value;
      const dbl2 t = *reinterpret_cast<FB_LDS const dbl2*>(L + pair_at(S0 / 2 + pr));
      out[2 * pr] = t[0];
      if constexpr (2 * pr + 1 < CNT) out[2 * pr + 1] = t[1];
    });
  }

  // (A zz)_k for the constraints k = r + 16 s of this lane, zz given in every
  // lane (zb[c] = element c of the stage vector): C through its LDS transpose.
  // Caller has written Cl and synchronised.
  template <class Out>
  static FB_DEV void rows_of_C_times(lds_ptr Cl, const double (&zb)[NS], int r, Out&& out) {
    sfor<0, KS>([&](auto S_) {
      constexpr int s = decltype(S_)::value;
      const int k = r + LPQ * s;
      const int kk = k < NC ? k : 0;
      double clk[NS];
      sfor<0, NS>([&](auto Cc) { clk[decltype(Cc)::value] = Cl[decltype(Cc)::value * CS + kk]; });
      out(S_, k < NC, dot4<NS>(clk, zb));
    });
  }
  // The same with zz one entry per lane (lane c holds element c): the broadcast rides in the FMA
  // (bc_dot: the same four partial sums in the same order as dot4 over broadcast copies - bitwise equal).
  // A slot that lies wholly inside the NC rows is valid in every lane: said at compile time.
  template <class Out>
  static FB_DEV void rows_of_C_times_lane(lds_ptr Cl, double zz, int r, Out&& out) {
    sfor<0, KS>([&](auto S_) {
      constexpr int s = decltype(S_)::value;
      const int k = r + LPQ * s;
      const int kk = (LPQ * (s + 1) <= NC || k < NC) ? k : 0;
      double clk[NS];
      sfor<0, NS>([&](auto Cc) { clk[decltype(Cc)::value] = Cl[decltype(Cc)::value * CS + kk]; });
      out(S_, LPQ * (s + 1) <= NC || k < NC, bc_dot<0, NS, RQ>(clk, zz));
    });
  }
  static FB_DEV void C_to_lds(const C& c, lds_ptr Cl, const double (&Cc)[NC], int r) {
    c.sync();
    sfor<0, NC>([&](auto Kk) { Cl[r * CS + decltype(Kk)::value] = Cc[decltype(Kk)::value]; });
    c.sync();
  }

  // ---- pointers into the caller's arrays ----------------------------------------
  FB_DEV const double* arr(int a) const { return data->base[a] + q * data->stride[a]; }
  FB_DEV double* xarr(int a) const { return var->base[a] + q * var->stride[a]; }

  // ||(f,h,b)||_2 (mpc_data.h:88-97).
  FB_DEV double forcing_norm(const C& c) const {
    const int N_ = N;
    const double *pq = arr(FBSTAB_MPC_q), *pr = arr(FBSTAB_MPC_r), *pd = arr(FBSTAB_MPC_d),
                 *px0 = arr(FBSTAB_MPC_x0), *pc = arr(FBSTAB_MPC_c);
    double s = 0.0;
    const int nx_ = prob_nx(), nu_ = prob_nu(), nc_ = prob_nc();
    for (int i = c.tid; i < (N_ + 1) * nx_; i += LPQ) s += pq[i] * pq[i];
    for (int i = c.tid; i < (N_ + 1) * nu_; i += LPQ) s += pr[i] * pr[i];
    for (int i = c.tid; i < (N_ + 1) * nc_; i += LPQ) s += pd[i] * pd[i];
    for (int i = c.tid; i < nx_; i += LPQ) s += px0[i] * px0[i];
    for (int i = c.tid; i < N_ * nx_; i += LPQ) s += pc[i] * pc[i];
    return sqrt(qp_reduce<RQ, OpSum16>(s));
  }

  // x <- caller's guess, y = b - A z (impl:334-347, full_variable.cc:47-53), and
  // the one-off copies of the problem data into the records.
  FB_DEV void load_guess(const C& c) {
    FB_WAVE_TIMER(19);
    const int r = c.tid, N_ = N;
    const int nx_ = prob_nx(), nu_ = prob_nu(), nc_ = prob_nc();
    const int ru = r - NX;
    const bool rx = r < nx_;                   // lane holds a real state row
    const bool rin = r >= NX && ru < nu_;      // lane holds a real input row
    const bool rs_ = rx || rin;
    double* const R0 = rec;
    lds_ptr Cl = lds;
    const double *Q = arr(FBSTAB_MPC_Q), *Rm = arr(FBSTAB_MPC_R), *S = arr(FBSTAB_MPC_S),
                 *pq = arr(FBSTAB_MPC_q), *pr = arr(FBSTAB_MPC_r), *A = arr(FBSTAB_MPC_A),
                 *B = arr(FBSTAB_MPC_B), *pc = arr(FBSTAB_MPC_c), *E = arr(FBSTAB_MPC_E),
                 *L = arr(FBSTAB_MPC_L), *pd = arr(FBSTAB_MPC_d), *px0 = arr(FBSTAB_MPC_x0);
    const double *uz = xarr(0), *ul = xarr(1), *uv = xarr(2);
    pend_t = 0.0;
    double* const P0 = pack;
    int* const po = poff;
    const lds_iptr lp = lpo;
    int nzm = 0;  // the most nonzeros seen in one constraint row so far (row-uniform)
    double c2m = 0.0, hm = 0.0;  // this lane's largest sum_k C[k][r]^2 and sum_c |H[r][c]| so far
    int canon = 0;  // offset of the copy the previous stage uses (row-uniform)
    double lastKr[NSP], lastABr[NSP], lastCc[NC], lastABc[NX];  // that copy's values, this lane's share
    sfor<0, NSP>([&](auto Cc_) { lastKr[decltype(Cc_)::value] = lastABr[decltype(Cc_)::value] = 0.0; });
    sfor<0, NC>([&](auto Kk) { lastCc[decltype(Kk)::value] = 0.0; });
    sfor<0, NX>([&](auto J) { lastABc[decltype(J)::value] = 0.0; });
    // The stage's vectors from the caller's arrays (f, h, the guess, b, v): requested a
    // stage AHEAD.  Behind the matrix loads, the comparison and the stores of their own
    // stage - the compiler may not move a load of the caller's memory above a store to
    // the records - they were three more trips to memory per stage, one after the other
    // (f h; z l; b v), in a pass that has nothing to cover a trip with.
    struct Small {
      double f, h, zz, ll, b[KS], vv[KS];
    };
    auto load_small = [&](int i, Small& sm) {
      sm.f = rx ? pq[(long)i * nx_ + r] : (rin ? pr[(long)i * nu_ + ru] : 0.0);
      sm.h = rx ? (i == 0 ? -px0[r] : -pc[(long)(i - 1) * nx_ + r]) : 0.0;
      sm.zz = rx ? uz[(long)i * (nx_ + nu_) + r] : (rin ? uz[(long)i * (nx_ + nu_) + nx_ + ru] : 0.0);
      sm.ll = rx ? ul[(long)i * nx_ + r] : 0.0;
      sfor<0, KS>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        const int k = r + LPQ * sl;
        const bool real = k < NC && k < nc_;
        sm.b[sl] = real ? -pd[(long)i * nc_ + k] : 0.0;
        sm.vv[sl] = real ? uv[(long)i * nc_ + k] : 0.0;
      });
    };
#ifndef FB_R16_LOAD_AHEAD
#define FB_R16_LOAD_AHEAD 0
#endif
    [[maybe_unused]] Small nxt_small;
    if constexpr (FB_R16_LOAD_AHEAD != 0) load_small(0, nxt_small);
    for (int i = 0; i <= N_; i++) {
      double* R = R0 + (long)i * kRec;
      double* PK = P0 + (long)i * kPack;
      const bool has_ab = i < N_;
      Small sm;
      if constexpr (FB_R16_LOAD_AHEAD != 0) {
        sm = nxt_small;
        if (i < N_) load_small(i + 1, nxt_small);
      } else {
        load_small(i, sm);  // (at the top of their own stage, with the matrix loads: 531 k QP/s
                            //  against 519 k with the loads where they were and 480 k with
                            //  them a stage ahead, gpurun_out/r03_ai)
      }
      double Cc[NC];
      bool fresh = true;
      if constexpr (KEEP) fresh = !reuse;
      if (!fresh) {
        const int kept = po[i];
        lp[i] = kept;
        ldv<pC, NC>(P0 + kept, Cc);  // only C is needed here (y = b - A z)
      } else {
        // matrices (the caller's arrays have the problem's own strides nx, nu, nc)
        double Kr[NSP], ABr[NSP], ABc[NX];
        sfor<0, NSP>([&](auto Cc_) {
          constexpr int cc = decltype(Cc_)::value;
          double kv = 0.0, ab = 0.0;
          if constexpr (cc < NX) {
            if (cc < nx_) {  // a real state column
              if (rx) kv = Q[(long)i * nx_ * nx_ + r + cc * nx_];
              else if (rin) kv = S[(long)i * nu_ * nx_ + ru + cc * nu_];
              if (rx && has_ab) ab = A[(long)i * nx_ * nx_ + r + cc * nx_];
            }
          } else if constexpr (cc < NS) {
            if (cc - NX < nu_) {  // a real input column
              if (rx) kv = S[(long)i * nu_ * nx_ + (long)r * nu_ + (cc - NX)];
              else if (rin) kv = Rm[(long)i * nu_ * nu_ + ru + (cc - NX) * nu_];
              if (rx && has_ab) ab = B[(long)i * nx_ * nu_ + r + (cc - NX) * nx_];
            }
          }
          Kr[cc] = kv;
          ABr[cc] = ab;
        });
        {
          const double* src = rx ? E + ((long)i * nx_ + r) * nc_ : L + ((long)i * nu_ + (rin ? ru : 0)) * nc_;
          sfor<0, NC>([&](auto Kk) {
            constexpr int k = decltype(Kk)::value;
            Cc[k] = (rs_ && k < nc_) ? src[k] : 0.0;
          });
        }
        {
          const double* src = rx ? A + (long)i * nx_ * nx_ + (long)r * nx_
                                 : B + (long)i * nx_ * nu_ + (long)(rin ? ru : 0) * nx_;
          sfor<0, NX>([&](auto J) {
            constexpr int j = decltype(J)::value;
            ABc[j] = (rs_ && has_ab && j < nx_) ? src[j] : 0.0;
          });
        }
        {
          const int rowsh = LPQ * ((threadIdx.x & 63) / LPQ);
          double c2 = 0.0, hs = 0.0;
          sfor<0, NC>([&](auto Kk) {
            const unsigned long long m = __ballot(Cc[decltype(Kk)::value] != 0.0);
            nzm = max(nzm, __popc((unsigned)(m >> rowsh) & (unsigned)((1ull << LPQ) - 1ull)));
            c2 = fma(Cc[decltype(Kk)::value], Cc[decltype(Kk)::value], c2);
          });
          sfor<0, NSP>([&](auto Cc_) { hs += fabs(Kr[decltype(Cc_)::value]); });
          c2m = fmax(c2m, c2);
          hm = fmax(hm, hs);
        }
        // A stage whose matrices equal (bitwise) those of the previous stage
        // shares its copy: nothing is written for it.
        bool differs = i == 0;
        sfor<0, NSP>([&](auto Cc_) {
          constexpr int cc = decltype(Cc_)::value;
          differs = differs || !(Kr[cc] == lastKr[cc]) || !(ABr[cc] == lastABr[cc]);
        });
        sfor<0, NC>([&](auto Kk) { differs = differs || !(Cc[decltype(Kk)::value] == lastCc[decltype(Kk)::value]); });
        sfor<0, NX>([&](auto J) { differs = differs || !(ABc[decltype(J)::value] == lastABc[decltype(J)::value]); });
        if (qp_reduce<RQ, OpMax16>(differs ? 1.0 : 0.0) > 0.0) {
          canon = i * kPack;
          stv<pK, NSP>(PK, Kr);
          stv<pABr, NSP>(PK, ABr);
          stv<pC, NC>(PK, Cc);
          stv<pABc, NX>(PK, ABc);
          sfor<0, NSP>([&](auto Cc_) {
            lastKr[decltype(Cc_)::value] = Kr[decltype(Cc_)::value];
            lastABr[decltype(Cc_)::value] = ABr[decltype(Cc_)::value];
          });
          sfor<0, NC>([&](auto Kk) { lastCc[decltype(Kk)::value] = Cc[decltype(Kk)::value]; });
          sfor<0, NX>([&](auto J) { lastABc[decltype(J)::value] = ABc[decltype(J)::value]; });
        }
        po[i] = canon;
        lp[i] = canon;
      }
      lds_off = -1;
      // constants f, h, b (mpc_data.cc:240-289)
      st2(R, sF, sm.f, sm.h);
      // the guess - as the proximal centre, with a zero displacement - and y = b - A z
      const double zz = sm.zz, ll = sm.ll;
      st2(R, sZ, 0.0, 0.0);
      st2(R, sZB, zz, ll);
      st2(R, sL, 0.0, 0.0);
      st2(R, sDZ, 0.0, 0.0);
      st2(R, sDL, 0.0, 0.0);
      double zb[NS];
      bc_all<NS, RQ>(zz, zb);
      C_to_lds(c, Cl, Cc, r);
      rows_of_C_times(Cl, zb, r, [&](auto S_, bool valid, double az) {
        constexpr int s = decltype(S_)::value;
        const int k = r + LPQ * s;
        const bool real = valid && k < nc_;
        const double b = sm.b[s], vv = sm.vv[s];
        st(R, sB + s, b);
        st2(R, sV + 2 * s, vv, real ? b - az : 0.0);
        st2(R, sDV + 2 * s, 0.0, 0.0);
      });
    }
    zero_spare(rec, N_);
    {
      bool fresh_all = true;
      if constexpr (KEEP) fresh_all = !reuse;
      if (fresh_all) {  // (kept with the copies for FBSTAB_HIP_KEEP_MATRICES)
        cmax2 = (float)qp_reduce<RQ, OpMax16>(c2m);
        hmax = (float)qp_reduce<RQ, OpMax16>(hm);
        po[N_ + 1] = nzm;
        po[N_ + 2] = __float_as_int(cmax2);
        po[N_ + 3] = __float_as_int(hmax);
      } else {
        nzm = po[N_ + 1];
        cmax2 = __int_as_float(po[N_ + 2]);
        hmax = __int_as_float(po[N_ + 3]);
      }
      nzmax = nzm;
      bounds = nzm <= 1;
    }
    c.sync();
  }

  // Which form of the costate step this QP's backward sweeps take (see `rowdl`); called
  // once per QP, after load_guess().
#ifndef FB_R16_ROW_COSTATE
#define FB_R16_ROW_COSTATE 1  // 0: every QP takes the reference's form (a) - to tell the two apart in a comparison
#endif
#ifndef FB_R16_ROW_SPARSE
#define FB_R16_ROW_SPARSE 1  // 0: only bound constraints take the row form
#endif
  FB_DEV void choose_costate_form(double sigma) {
    // |C'Gamma C| <= (nonzeros per row) x (column sums of C'C) / sigma: bounds with entries up to 2, or sparse
    // rows (a few entries of order one per row - the bench line's time-varying workload) within the same
    // factor 8 of 1/sigma in all
    const bool small = bounds ? cmax2 <= 4.f : (FB_R16_ROW_SPARSE != 0 && (float)nzmax * cmax2 <= 8.f);
    rowdl = FB_R16_ROW_COSTATE != 0 && small && (double)hmax * sigma <= 1.0;
  }

  // Natural residual blocks at x: rz = Hz + f + G'l + A'v, rl = h - Gz
  // (full_residual.cc:79-91; mpc_data.cc:28-63, :127-152, :171-198, :217-237).
  FB_DEV void residual(const C& c) {
    const int r = c.tid, N_ = N;
    const bool rx = r < NX;
    double* const R0 = rec;
    const double* const P0 = pack;
    const lds_iptr po = lpo;
    pk_ptr Lp = pack_view(c);  // this lane's view of the matrix copy in use
    for (int i = 0; i <= N_; i++) {
      double* R = R0 + (long)i * kRec;
      const int pofs = po[i];
      stage_pack(c, Lp, pofs);
      double Kr[NS], Cc[NC], ABr[NS], ABc[NX];
      if constexpr (kKinLds) ldl<pK, NS>(Lp, Kr);
      else ldv<pK, NS>(P0 + pofs, Kr);  // (K = [Q S'; S R] is not in this instance's image)
      ldl<pC, NC>(Lp, Cc);
      ldl<pABr, NS>(Lp, ABr);
      ldv<pABc, NX>(P0 + pofs, ABc);
      const dbl2 cb = ld2(R, sZB);
      const double zz = cb[0] + ld(R, sZ), ll = cb[1] + ld(R, sL);
      const dbl2 fh = ld2(R, sF);
      double vs[KS];
      sfor<0, KS>([&](auto S_) { vs[decltype(S_)::value] = ld(R, sV + 2 * decltype(S_)::value); });
      double ln = 0.0, zn = 0.0;  // l(i+1), z(i+1)
      double hn = 0.0;
      if (i < N_) {
        const dbl2 cbn = ld2(R + kRec, sZB);
        ln = cbn[1] + ld(R + kRec, sL);
        zn = cbn[0] + ld(R + kRec, sZ);
        hn = ld(R + kRec, sH);
      }
      double zb[NS], lnb[NX];
      bc_all<NS, RQ>(zz, zb);
      bc_all<NX, RQ>(ln, lnb);
      double s = fh[0] + dot4<NS>(Kr, zb);
      s += dot4<NX>(ABc, lnb) - (rx ? ll : 0.0);
      {
        double p[4] = {s, 0.0, 0.0, 0.0};
        bc_cols_dot<NC, RQ>(Cc, vs, p);
        s = (p[0] + p[1]) + (p[2] + p[3]);
      }
      st(R, sRZ, s);  // zero where there is no row: every term is
      // block 0: h0 - (Gz)0 = -x0 + x(0); block i+1: -c(i) - (A x + B u - x(i+1))
      if (i == 0) st(R, sRL, rx ? fh[1] + zz : 0.0);
      if (i < N_) {
        const double abz = dot4<NS>(ABr, zb);
        st(R + kRec, sRL, rx ? hn - (abz - zn) : 0.0);
      }
    }
    c.sync();
  }

  // sqrt(sum rz^2 + rl^2 + pnr(y,v)^2) (full_residual.cc:99-109, :40-42).
  FB_DEV double pnr_norm(const C& c, double alpha) const {
    const int N_ = N;
    const double* const R0 = rec;
    double s = 0.0;
    for (int i = 0; i <= N_; i++) {
      const double* R = R0 + (long)i * kRec;
      const double a = ld(R, sRZ), b = ld(R, sRL);
      s = fma(a, a, s);
      s = fma(b, b, s);
      sfor<0, KS>([&](auto S_) {
        const dbl2 vy = ld2(R, sV + 2 * decltype(S_)::value);
        const double p = pnr(vy[1], vy[0], alpha);  // zero on padding lanes
        s = fma(p, p, s);
      });
    }
    return sqrt(qp_reduce<RQ, OpSum16>(s));
  }

  // (Ei, Eo) at x + t dx for the K step lengths t0 beta^k in one pass
  // (full_residual.cc:49-74 and :99-109).  A pending step is applied first.
  // The light passes below are bound by load latency, not arithmetic: each keeps
  // the loads of stage i+1 in flight while stage i is processed.
  struct TrialIn {
    dbl2 zr, dw, lr, dwl;
    dbl2 vy[KS], da[KS];
    double vb[KS];
  };
  static FB_DEV void load_trial(const double* R, TrialIn& in) {
    in.zr = ld2(R, sZ);
    in.dw = ld2(R, sDZ);
    in.lr = ld2(R, sL);
    in.dwl = ld2(R, sDL);
    sfor<0, KS>([&](auto S_) {
      constexpr int sl = decltype(S_)::value;
      in.vy[sl] = ld2(R, sV + 2 * sl);
      in.da[sl] = ld2(R, sDV + 2 * sl);
    });
    ldv<sVB, KS>(R, in.vb);
  }
  // wl(i) for a pass that walks the stages upwards: dx(0) on the state lanes at
  // stage 0, afterwards the WLN the previous stage held.
  static FB_DEV double wl_of_stage(int i, bool rx, double dz0, double wln_prev) {
    return i == 0 ? (rx ? dz0 : 0.0) : wln_prev;
  }
  template <int K>
  FB_DEV void norms_at_multi(const C& c, double t0, double beta, double sigma, double alpha,
                             double (&Ei)[K], double (&Eo)[K]) {
    FB_WAVE_COUNT(24);
    FB_WAVE_TIMER(23);
    flush(c);
    const int N_ = N;
    const double* const R0 = rec;
    double tt[K], s[2 * K];
    tt[0] = t0;
    sfor<1, K>([&](auto Kk) { tt[decltype(Kk)::value] = tt[decltype(Kk)::value - 1] * beta; });
    sfor<0, 2 * K>([&](auto Kk) { s[decltype(Kk)::value] = 0.0; });
    TrialIn in;
    load_trial(R0, in);
    const bool rx = c.tid < NX;
    double wln = 0.0;
    for (int i = 0; i <= N_; i++) {
      const TrialIn cu = in;
      if (i < N_) load_trial(R0 + (long)(i + 1) * kRec, in);
      const double wl = wl_of_stage(i, rx, cu.dw[0], wln);
      wln = cu.dwl[1];
      sfor<0, K>([&](auto Kk) {
        constexpr int k = decltype(Kk)::value;
        const double rzt = fma(tt[k], cu.dw[1], cu.zr[1]);
        const double rzi = rzt + sigma * fma(tt[k], cu.dw[0], cu.zr[0]);
        const double rlt = fma(tt[k], wl, cu.lr[1]);
        const double rli = rlt + sigma * fma(tt[k], cu.dwl[0], cu.lr[0]);
        s[k] = fma(rzi, rzi, s[k]);
        s[k] = fma(rli, rli, s[k]);
        s[K + k] = fma(rzt, rzt, s[K + k]);
        s[K + k] = fma(rlt, rlt, s[K + k]);
      });
      sfor<0, KS>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        sfor<0, K>([&](auto Kk) {
          constexpr int k = decltype(Kk)::value;
          const double vi = fma(tt[k], cu.da[sl][0], cu.vy[sl][0]);
          const double yi = fma(-tt[k], cu.da[sl][1], cu.vy[sl][1]);
          const double ys = yi + sigma * (vi - cu.vb[sl]);
          const double ph = pfb(ys, vi, alpha);
          const double pn = pnr(yi, vi, alpha);
          s[k] = fma(ph, ph, s[k]);
          s[K + k] = fma(pn, pn, s[K + k]);
        });
      });
    }
    sfor<0, K>([&](auto Kk) {
      constexpr int k = decltype(Kk)::value;
      Ei[k] = sqrt(qp_reduce<RQ, OpSum16>(s[k]));
      Eo[k] = sqrt(qp_reduce<RQ, OpSum16>(s[K + k]));
    });
  }

  // The same pass for ONE QP of the wavefront (the one whose lane `owner` is), run by ALL
  // its rows: row q takes the stages q, q + QW, q + 2 QW, ... - the rows that need a
  // line-search pass at a given moment are usually one or two of the four, and the
  // others' lanes would only wait.  Every lane calls this (wave-uniform control flow);
  // every lane gets the owner's norms.  The sums over the stages are formed per row and
  // then over the rows: same terms as norms_at_multi(), different order of summation.
  // No step is pending here (the Newton step's forward sweep applied it).
  // The pass proper is a real call (trial_pass_coop, not inlined): it takes nothing but
  // scalars, so the policy object stays in registers; inlined into the solver loop the
  // exact <12,4,32> instance came out finishing every QP at its first convergence test
  // (any other change of the surrounding code made it right again; round 3 traced that
  // kind of failure to rows that join these passes with a policy object no fetch has
  // bound yet - bind_idle() - see profiles/r03_q_coop_inline_miscompile_notes.txt;
  // tests/test_gpu_parity.py runs every instance, exact and padded).
#ifndef FB_R16_COOP_TRIALS
#define FB_R16_COOP_TRIALS 1
#endif
  static constexpr bool kCoopTrials = FB_R16_COOP_TRIALS != 0;
  static FB_DEV double lane_value(double x, int lane) {  // x of lane `lane` (wave-uniform index)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane),
                            __builtin_amdgcn_readlane(__double2loint(x), lane));
  }
  // The constraint blocks' share of the trial norms: sum pfb(...)^2 and sum pnr(...)^2 over the stages,
  // K step lengths per pass (the z and l blocks are affine in x: StepOut, trial_zl()).
  template <int K>
  struct TrialNorms {
    double vi[K], vo[K];
  };
  struct TrialInV {
    dbl2 vy[KS], da[KS];
    double vb[KS];
  };
  static FB_DEV void load_trial_v(const double* R, const double* Rd, TrialInV& in) {
    const double* const Rt = tail_of(R, Rd);
    sfor<0, KS>([&](auto S_) {
      constexpr int sl = decltype(S_)::value;
      in.vy[sl] = ld2(sl == KS - 1 ? Rt : R, sV + 2 * sl);
      in.da[sl] = ld2(sl == KS - 1 ? Rt : R, sDV + 2 * sl);
    });
    ldv<sVB, KS>(R, in.vb);
  }
  // R0: the owner's record base with this lane's offset within ITS row (2 * r)
  template <int K>
#ifndef FB_R16_COOP_INLINE
#define FB_R16_COOP_INLINE 0
#endif
#if FB_R16_COOP_INLINE
  static __device__ __forceinline__ TrialNorms<K> trial_pass_coop(const double* R0, int N_, double t0,
                                                                            double beta, double sigma, double alpha) {
#else
  static __device__ __attribute__((noinline)) TrialNorms<K> trial_pass_coop(const double* R0, int N_, double t0,
                                                                            double beta, double sigma, double alpha) {
#endif
    constexpr int QW = kQpPerWave;
    const int lane = threadIdx.x & 63;
    const int q = lane / LPQ;
    double tt[K], s[2 * K];
    tt[0] = t0;
    sfor<1, K>([&](auto Kk) { tt[decltype(Kk)::value] = tt[decltype(Kk)::value - 1] * beta; });
    sfor<0, 2 * K>([&](auto Kk) { s[decltype(Kk)::value] = 0.0; });
    auto stage_ptr = [&](int i) { return R0 + (long)(i < N_ ? i : N_) * kRec; };
    const double* const Rd = spare_of(R0, N_);
    // (The slots of a trip are requested a trip ahead - this pass stores nothing.  At the top
    // of their own trip, FB_R16_TRIAL_AHEAD=0: 485 k against 533 k QP/s, gpurun_out/r03_ap;
    // two trips ahead: -6 %, round 2.  In the passes that also STORE records - open, close,
    // the packing pass - every form of requesting ahead has lost, see open_pass_coop.)
#ifndef FB_R16_TRIAL_AHEAD
#define FB_R16_TRIAL_AHEAD 1
#endif
    // TS stages per trip and row (stages i, i + QW, ...): the pass is bound by the latency of a trip's
    // loads, not by their bytes - with the z and l slots gone a trip of one stage moves five slot pairs,
    // so two stages per trip keep as many loads in flight as the pass had before and halve the trips.
#ifndef FB_R16_TRIAL_STAGES
#define FB_R16_TRIAL_STAGES 1  // (2: one launch at a time 2 % faster, eight in flight 0.7 % slower; 4: +3 % / -7 %; gpurun_out/r05_m)
#endif
    constexpr int TS = FB_R16_TRIAL_STAGES;
    TrialInV in[TS];
    if constexpr (FB_R16_TRIAL_AHEAD != 0)
      sfor<0, TS>([&](auto J) { load_trial_v(stage_ptr(q + QW * decltype(J)::value), Rd, in[decltype(J)::value]); });
    for (int i = q; i - q <= N_; i += TS * QW) {  // (the same trip count in every row)
      FB_PHASE(trip_top);
      if constexpr (FB_R16_TRIAL_AHEAD == 0)
        sfor<0, TS>([&](auto J) { load_trial_v(stage_ptr(i + QW * decltype(J)::value), Rd, in[decltype(J)::value]); });
      TrialInV cu[TS];
      sfor<0, TS>([&](auto J) { cu[decltype(J)::value] = in[decltype(J)::value]; });
      if constexpr (FB_R16_TRIAL_AHEAD != 0)
        sfor<0, TS>([&](auto J) { load_trial_v(stage_ptr(i + QW * (TS + decltype(J)::value)), Rd, in[decltype(J)::value]); });
      sfor<0, TS>([&](auto J) {
        constexpr int j = decltype(J)::value;
        // A row that has run out of stages (the last trip) re-reads the last stage - lines the row that owns
        // that stage has just pulled into the cache - and its terms are selected away.  (Round 6 measured two
        // ways of saving the four selects per step length and slot, both SLOWER although shorter: such rows
        // pointed at an all-zero spare record, no select at all - 25 instructions fewer per trip, 595 k against
        // 627 k QP/s: the spare's lines are cold and every pass ended on a trip to HBM for data nobody uses;
        // and the trip's arithmetic behind a branch on `live` - 46 instructions fewer, 624 k against 637 k.
        // gpurun_out/r06_b, r06_c.)
        const bool live = i + QW * j <= N_;
        sfor<0, KS>([&](auto S_) {
          constexpr int sl = decltype(S_)::value;
          sfor<0, K>([&](auto Kk) {
            constexpr int k = decltype(Kk)::value;
            const double vi = fma(tt[k], cu[j].da[sl][0], cu[j].vy[sl][0]);
            const double yi = fma(-tt[k], cu[j].da[sl][1], cu[j].vy[sl][1]);
            const double ys = yi + sigma * (vi - cu[j].vb[sl]);
            const double ph = pfb(ys, vi, alpha);
            const double pn = pnr(yi, vi, alpha);
            s[k] = live ? fma(ph, ph, s[k]) : s[k];
            s[K + k] = live ? fma(pn, pn, s[K + k]) : s[K + k];
          });
        });
      });
      FB_PHASE(trip_end);
    }
    TrialNorms<K> out;
    sfor<0, 2 * K>([&](auto Kk) {
      constexpr int k = decltype(Kk)::value;
      const double rs = qp_reduce<RQ, OpSum16>(s[k]);  // the row's (row pair's) stages
      double tot = lane_value(rs, 0);
      sfor<1, QW>([&](auto Q_) { tot += lane_value(rs, LPQ * decltype(Q_)::value); });
      if constexpr (k < K) out.vi[k] = tot;
      else out.vo[k - K] = tot;
    });
    return out;
  }
  // ---- the same pass with the LAST constraint slot of four stages evaluated as ONE slot (round 6) --------
  // NC = 20 on 16 lanes: the second slot of a stage holds entries 16..19 on 4 lanes and nothing on the other
  // 12, and the pass above evaluates it - four step lengths, both functions - once per stage all the same: 37 %
  // of the pass's arithmetic on 25 % full slots.  Here a row evaluates the full slots of its stage trip by
  // trip as before, and once per FOUR trips the last slots of the four stages it took in them together: lane
  // l loads entry l % kTail of the stage of trip l / kTail straight from that stage's record.  Same terms;
  // a row's partial sums collect them in a different order (as this pass already does against the
  // sequential one).  12 evaluations of 32 fewer per four trips.
  // MEASURED (gpurun_out/r06_o, same box, 3 interleaved runs): SQ_INSTS_VALU of the 8192-QP launch 3.953 G ->
  // 3.779 G (-4.4 %), every count and the Newton total unchanged, 162 parity tests green - and the headline
  // 630.5 k -> 624.9 k QP/s (-0.9 %), one launch at a time +-0: the pass is a chain of eight trips each
  // waiting for loads requested one trip earlier, and a trip with half the arithmetic covers half the
  // latency.  The third time this round that a shorter trial pass was not a faster one (the other two are
  // in trial_pass_coop).  OFF: the knob is the record of the experiment, not the product.
#ifndef FB_R16_TAIL_PACK
#define FB_R16_TAIL_PACK 0
#endif
  static constexpr int kTailGroup = 4;
  static constexpr bool kTailPack = FB_R16_TAIL_PACK != 0 && KS >= 2 && kTail * kTailGroup <= LPQ && kTailCut &&
                                    FB_R16_TRIAL_STAGES == 1;
  template <int K>
  static __device__ __attribute__((noinline)) TrialNorms<K> trial_pass_coop_packed(const double* R0, int N_, double t0,
                                                                                   double beta, double sigma, double alpha) {
    constexpr int QW = kQpPerWave, G = kTailGroup, KF = KS - 1;
    const int lane = threadIdx.x & 63;
    const int q = lane / LPQ, r = lane & (LPQ - 1);
    double tt[K], s[2 * K];
    tt[0] = t0;
    sfor<1, K>([&](auto Kk) { tt[decltype(Kk)::value] = tt[decltype(Kk)::value - 1] * beta; });
    sfor<0, 2 * K>([&](auto Kk) { s[decltype(Kk)::value] = 0.0; });
    auto stage_ptr = [&](int i) { return R0 + (long)(i < N_ ? i : N_) * kRec; };
    struct Full {
      dbl2 vy[KF], da[KF];
      double vb[KF];
    };
    struct Tail {
      dbl2 vy, da;
      double vb;
    };
    auto load_full = [&](const double* R, Full& in) {
      sfor<0, KF>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        in.vy[sl] = ld2(R, sV + 2 * sl);
        in.da[sl] = ld2(R, sDV + 2 * sl);
        in.vb[sl] = ld(R, sVB + sl);
      });
    };
    // this lane's place in a packed last slot: entry te of the stage this row takes in trip tsub of the group
    const int tsub = r / kTail, te = r - tsub * kTail;
    auto load_tail = [&](int g, Tail& in) {
      const double* Rt = stage_ptr(q + QW * (G * g + tsub)) - 2 * r + 2 * te;  // (R0 carries this lane's 2 r)
      in.vy = ld2(Rt, sV + 2 * KF);
      in.da = ld2(Rt, sDV + 2 * KF);
      in.vb = ld(Rt, sVB + KF);
    };
    auto terms = [&](const dbl2& vy, const dbl2& da, double vb, bool live) {
      sfor<0, K>([&](auto Kk) {
        constexpr int k = decltype(Kk)::value;
        const double vi = fma(tt[k], da[0], vy[0]);
        const double yi = fma(-tt[k], da[1], vy[1]);
        const double ys = yi + sigma * (vi - vb);
        const double ph = pfb(ys, vi, alpha);
        const double pn = pnr(yi, vi, alpha);
        s[k] = live ? fma(ph, ph, s[k]) : s[k];
        s[K + k] = live ? fma(pn, pn, s[K + k]) : s[K + k];
      });
    };
    const int last_trip = N_ / QW;  // (the same trip count in every row: trips t with QW t <= N)
    Full in;
    load_full(stage_ptr(q), in);
    for (int g = 0; G * g <= last_trip; g++) {
      FB_PHASE(trip_top);
      Tail tl;
      load_tail(g, tl);  // (used at the end of the group: three trips of cover)
      sfor<0, G>([&](auto J) {
        const int t = G * g + decltype(J)::value;
        if (t <= last_trip) {  // (wave-uniform)
          const Full cu = in;
          load_full(stage_ptr(q + QW * (t + 1)), in);
          const bool live = q + QW * t <= N_;
          sfor<0, KF>([&](auto S_) {
            constexpr int sl = decltype(S_)::value;
            terms(cu.vy[sl], cu.da[sl], cu.vb[sl], live);
          });
        }
      });
      terms(tl.vy, tl.da, tl.vb, tsub < G && q + QW * (G * g + tsub) <= N_);
      FB_PHASE(trip_end);
    }
    TrialNorms<K> out;
    sfor<0, 2 * K>([&](auto Kk) {
      constexpr int k = decltype(Kk)::value;
      const double rs = qp_reduce<RQ, OpSum16>(s[k]);
      double tot = lane_value(rs, 0);
      sfor<1, QW>([&](auto Q_) { tot += lane_value(rs, LPQ * decltype(Q_)::value); });
      if constexpr (k < K) out.vi[k] = tot;
      else out.vo[k - K] = tot;
    });
    return out;
  }
  // The z and l blocks' share of the squared trial norms at x + t dx, from four sums of the Newton step.
  // Both residuals are affine there: with a = the inner (natural) residual's z, l blocks at x and b its
  // increment along dx (b = W + sigma dx for the inner one, b = W for the natural one),
  //   sum (a + t b)^2 = A (1 - t) + t S + B (t^2 - t),   A = sum a^2, S = sum (a + b)^2, B = sum b^2,
  // so a trial pass has nothing to read of z, l, their steps and residuals - 11 of the 21 slots per stage
  // it used to move.  (Not used at t = 1, where S is the sum itself; for t <= beta the three terms cancel
  // by a factor t / (1 - t) <= 3 at most.)
  // Clamped at zero: where a + t b all but vanishes the three terms cancel to an absolute error of eps A and
  // the sum can come out a hair below zero - the caller takes its root.
  static FB_DEV double trial_zl(double A, double S, double B, double t) {
    return fmax(0.0, fma(B, t * t - t, fma(A, 1.0 - t, t * S)));
  }
  // Vi, Vo: the squared norms' constraint-block shares for t0 beta^k, k < K, of the owner's QP
  template <int K>
  FB_DEV void trial_v_coop(int owner, double t0, double beta, double sigma, double alpha, double (&Vi)[K],
                           double (&Vo)[K]) const {
    FB_WAVE_COUNT(24);
    FB_WAVE_TIMER(23);
    // (the owner's records were written by the owner's lanes: their stores are complete
    // before another row's lanes read them)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // the owner's record base (lane 0 of its row: rec carries 2 * tid) and horizon
    const int own0 = owner & ~(LPQ - 1);
    const unsigned long long rb = (unsigned long long)rec;
    const unsigned long long rbo = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(rb >> 32), own0) << 32) |
                                   (unsigned)__builtin_amdgcn_readlane((int)rb, own0);
    const double* const R0 = reinterpret_cast<const double*>(rbo) + 2 * (threadIdx.x & (LPQ - 1));
    TrialNorms<K> n;
    if constexpr (kTailPack)
      n = trial_pass_coop_packed<K>(R0, __builtin_amdgcn_readlane(N, own0), lane_value(t0, own0), beta, sigma, alpha);
    else
      n = trial_pass_coop<K>(R0, __builtin_amdgcn_readlane(N, own0), lane_value(t0, own0), beta, sigma, alpha);
    sfor<0, K>([&](auto Kk) {
      Vi[decltype(Kk)::value] = n.vi[decltype(Kk)::value];
      Vo[decltype(Kk)::value] = n.vo[decltype(Kk)::value];
    });
  }

  // x <- x + t dx, (rz, rl) <- (rz, rl) + t W for the pending step (impl:298,
  // full_variable.cc:55-65).
  FB_DEV void flush(const C& c) {
    const double t = pend_t;
    pend_t = 0.0;
    if (t == 0.0) return;
    const int N_ = N;
    double* const R0 = rec;
    const bool rx = c.tid < NX;
    double wln = 0.0;
    for (int i = 0; i <= N_; i++) {
      double* R = R0 + (long)i * kRec;
      const dbl2 zr = ld2(R, sZ), dw = ld2(R, sDZ), lr = ld2(R, sL), dwl = ld2(R, sDL);
      const double wl = wl_of_stage(i, rx, dw[0], wln);
      wln = dwl[1];
      st2(R, sZ, fma(t, dw[0], zr[0]), fma(t, dw[1], zr[1]));
      st2(R, sL, fma(t, dwl[0], lr[0]), fma(t, wl, lr[1]));
      sfor<0, KS>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        const dbl2 vy = ld2(R, sV + 2 * sl), da = ld2(R, sDV + 2 * sl);
        st2(R, sV + 2 * sl, fma(t, da[0], vy[0]), fma(-t, da[1], vy[1]));
      });
    }
    c.sync();
  }

  // End of a proximal subproblem, ONE pass over the records: applies the
  // pending step, projects the duals (impl:298-301), forms dx = x - xbar on
  // (z, l, v) with its norm (impl:202-203, full_variable.cc:77-83) and
  // evaluates the infeasibility certificates for it (full_feasibility.cc:25-88).
  // The (z, l) blocks of stage i+1 are needed at stage i (G dz, G'dl): they
  // are fetched one stage ahead and handed on.
  struct ZL {
    double z, rz, l, rl, dz, dl;
  };
  // (z, l) blocks of stage i with the step t applied.  Called for i = 0, 1, 2, ...:
  // wln carries WLN from one stage to the next.
  static FB_DEV ZL stepped_zl(const double* R, double t, int i, bool rx, double& wln) {
    const dbl2 zr = ld2(R, sZ), dw = ld2(R, sDZ), lr = ld2(R, sL), dwl = ld2(R, sDL);
    const double wl = wl_of_stage(i, rx, dw[0], wln);
    wln = dwl[1];
    ZL o;  // (z, l: the displacements ez, el - which ARE dx = x - xbar)
    o.z = fma(t, dw[0], zr[0]);
    o.rz = fma(t, dw[1], zr[1]);
    o.l = fma(t, dwl[0], lr[0]);
    o.rl = fma(t, wl, lr[1]);
    o.dz = o.z;
    o.dl = o.l;
    return o;
  }
  FB_DEV int close_subproblem(const C& c, double tol, bool check, double* dx_norm) {
    FB_WAVE_COUNT(26);
    FB_WAVE_TIMER(21);
    const int r = c.tid, N_ = N;
    const bool rx = r < NX;
    const double t = pend_t;
    pend_t = 0.0;
    double* const R0 = rec;
    const double* const P0 = pack;
    const lds_iptr po = lpo;
    pk_ptr Lp = pack_view(c);  // this lane's view of the matrix copy in use
    lds_ptr Cl = lds;
    double m_adz = -1e300, m_gdz = 0.0, m_hdz = 0.0, m_dz = 0.0, m_atv = 0.0, m_u = 0.0;
    double s_fdz = 0.0, s_p2 = 0.0, s_dx = 0.0;
    struct VIn {
      dbl2 vy[KS], da[KS];
      double vb[KS];
      dbl2 fh;
      double bs[KS], ABc[NX];
    };
    auto load_v = [&](int i, VIn& in) {
      const double* R = R0 + (long)i * kRec;
      sfor<0, KS>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        in.vy[sl] = ld2(R, sV + 2 * sl);
        in.da[sl] = ld2(R, sDV + 2 * sl);
      });
      ldv<sVB, KS>(R, in.vb);
      if (check) {
        in.fh = ld2(R, sF);
        sfor<0, KS>([&](auto S_) { in.bs[decltype(S_)::value] = ld(R, sB + decltype(S_)::value); });
        ldv<pABc, NX>(P0 + po[i], in.ABc);
      }
    };
    double wln = 0.0;
    ZL cur = stepped_zl(R0, t, 0, rx, wln);
    ZL nxt = cur;
    if (N_ > 0) nxt = stepped_zl(R0 + kRec, t, 1, rx, wln);
    VIn vin;
    load_v(0, vin);
    for (int i = 0; i <= N_; i++) {
      double* R = R0 + (long)i * kRec;
      // (z, l) of stage i+2 and the v group of stage i+1, in flight during this stage
      ZL nn;
      nn.z = nn.rz = nn.l = nn.rl = nn.dz = nn.dl = 0.0;
      if (i + 2 <= N_) nn = stepped_zl(R + 2 * kRec, t, i + 2, rx, wln);
      const VIn vc = vin;
      if (i < N_) load_v(i + 1, vin);
      if (i == N_) nxt.z = nxt.rz = nxt.l = nxt.rl = nxt.dz = nxt.dl = 0.0;
      st2(R, sZ, cur.z, cur.rz);
      st2(R, sL, cur.l, cur.rl);
      st2(R, sDZ, cur.dz, 0.0);
      st2(R, sDL, cur.dl, 0.0);
      s_dx = fma(cur.dz, cur.dz, s_dx);
      s_dx = fma(cur.dl, cur.dl, s_dx);
      double dvs[KS];
      sfor<0, KS>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        const double vv = fmax0(fma(t, vc.da[sl][0], vc.vy[sl][0]));
        const double yy = fma(-t, vc.da[sl][1], vc.vy[sl][1]);
        dvs[sl] = vv - vc.vb[sl];
        st2(R, sV + 2 * sl, vv, yy);
        st2(R, sDV + 2 * sl, dvs[sl], 0.0);
        s_dx = fma(dvs[sl], dvs[sl], s_dx);
      });
      if (check) {
        stage_pack(c, Lp, po[i]);
        double Kr[NS], Cc[NC], ABr[NS];
        if constexpr (kKinLds) ldl<pK, NS>(Lp, Kr);
        else ldv<pK, NS>(pack + po[i], Kr);
        ldl<pC, NC>(Lp, Cc);
        ldl<pABr, NS>(Lp, ABr);
        const dbl2 fh = vc.fh;
        const double (&bs)[KS] = vc.bs;
        const double (&ABc)[NX] = vc.ABc;
        double dzb[NS], dlnb[NX];
        bc_all<NS, RQ>(cur.dz, dzb);
        bc_all<NX, RQ>(nxt.dl, dlnb);
        m_hdz = fmax(m_hdz, fabs(dot4<NS>(Kr, dzb)));
        m_dz = fmax(m_dz, fabs(cur.dz));
        {
          double p[4] = {dot4<NX>(ABc, dlnb) - (rx ? cur.dl : 0.0), 0.0, 0.0, 0.0};
          bc_cols_dot<NC, RQ>(Cc, dvs, p);
          m_atv = fmax(m_atv, fabs((p[0] + p[1]) + (p[2] + p[3])));
        }
        s_fdz = fma(fh[0], cur.dz, s_fdz);
        // (G dz): block 0 = -dx(0); block i+1 = A dx + B du - dx(i+1); h'dl
        m_u = fmax(m_u, fabs(cur.dl));
        s_p2 = fma(fh[1], cur.dl, s_p2);
        if (i == 0) m_gdz = fmax(m_gdz, rx ? fabs(cur.dz) : 0.0);
        if (i < N_) {
          const double g = dot4<NS>(ABr, dzb) - nxt.dz;
          m_gdz = fmax(m_gdz, rx ? fabs(g) : 0.0);
        }
        C_to_lds(c, Cl, Cc, r);
        rows_of_C_times(Cl, dzb, r, [&](auto S_, bool valid, double az) {
          constexpr int s = decltype(S_)::value;
          if (valid) m_adz = fmax(m_adz, az);
          m_u = fmax(m_u, fabs(dvs[s]));
          s_p2 = fma(bs[s], dvs[s], s_p2);
        });
      }
      cur = nxt;
      nxt = nn;
    }
    c.sync();
    *dx_norm = sqrt(qp_reduce<RQ, OpSum16>(s_dx));
    if (!check) return kFeasible;
    const double d1 = qp_reduce<RQ, OpMax16>(m_adz), d2 = qp_reduce<RQ, OpMax16>(m_gdz),
                 d3 = qp_reduce<RQ, OpMax16>(m_hdz), w = qp_reduce<RQ, OpMax16>(m_dz),
                 p1 = qp_reduce<RQ, OpMax16>(m_atv), u = qp_reduce<RQ, OpMax16>(m_u);
    const double d4 = qp_reduce<RQ, OpSum16>(s_fdz), p2 = qp_reduce<RQ, OpSum16>(s_p2);
    bool dual_feasible = true, primal_feasible = true;
    if ((d1 <= w * tol) && (d2 <= tol * w) && (d3 <= tol * w) && (d4 < 0) && (w > 1e-14))
      dual_feasible = false;
    if ((p1 <= tol * u) && (p2 < 0)) primal_feasible = false;
    if (primal_feasible && dual_feasible) return kFeasible;
    if (primal_feasible && !dual_feasible) return kDualInfeasible;
    if (!primal_feasible && dual_feasible) return kPrimalInfeasible;
    return kBothInfeasible;
  }

  // Start of a proximal iteration, ONE pass: xbar <- x (impl:212), the natural
  // residual at x (residual() above) and both norms the algorithm needs next:
  //   Ek  = ||(rz, rl, pnr(y, v))||                 (impl:146, :216)
  //   Ei0 = ||(rz, rl, pfb(y, v))||, the inner residual norm at x = xbar, where
  //         the sigma terms vanish identically      (impl:239-243)
  FB_DEV void open_prox(const C& c, double sigma, double alpha, double* Ek, double* Ei0) {
    FB_WAVE_COUNT(25);
    FB_WAVE_TIMER(22);
    (void)sigma;
    const int r = c.tid, N_ = N;
    const bool rx = r < NX;
    double* const R0 = rec;
    const double* const P0 = pack;
    const lds_iptr po = lpo;
    pk_ptr Lp = pack_view(c);  // this lane's view of the matrix copy in use
    double s_nat = 0.0, s_vo = 0.0, s_vi = 0.0;
    struct OIn {
      dbl2 fh, vy[KS];
      double zn, ln, hn;  // z, l, h of the following stage
      double ABc[NX];
    };
    auto load_o = [&](int i, OIn& in) {
      const double* R = R0 + (long)i * kRec;
      in.fh = ld2(R, sF);
      sfor<0, KS>([&](auto S_) { in.vy[decltype(S_)::value] = ld2(R, sV + 2 * decltype(S_)::value); });
      in.zn = in.ln = in.hn = 0.0;
      if (i < N_) {  // (x = xbar + displacement)
        const dbl2 cbn = ld2(R + kRec, sZB);
        in.zn = cbn[0] + ld(R + kRec, sZ);
        in.ln = cbn[1] + ld(R + kRec, sL);
        in.hn = ld(R + kRec, sH);
      }
      ldv<pABc, NX>(P0 + po[i], in.ABc);
    };
    dbl2 zl;  // (z, l) of stage i, handed on
    {
      const dbl2 cb0 = ld2(R0, sZB);
      zl[0] = cb0[0] + ld(R0, sZ);
      zl[1] = cb0[1] + ld(R0, sL);
    }
    OIn oin;
    load_o(0, oin);
    for (int i = 0; i <= N_; i++) {
      double* R = R0 + (long)i * kRec;
      const OIn oc = oin;
      if (i < N_) load_o(i + 1, oin);
      stage_pack(c, Lp, po[i]);
      double Kr[NS], Cc[NC], ABr[NS];
      if constexpr (kKinLds) ldl<pK, NS>(Lp, Kr);
      else ldv<pK, NS>(pack + po[i], Kr);
      ldl<pC, NC>(Lp, Cc);
      ldl<pABr, NS>(Lp, ABr);
      const double (&ABc)[NX] = oc.ABc;
      const dbl2 fh = oc.fh;
      dbl2 vy[KS];
      sfor<0, KS>([&](auto S_) { vy[decltype(S_)::value] = oc.vy[decltype(S_)::value]; });
      const dbl2 zln = {oc.zn, oc.ln};
      const double hn = oc.hn;
      const double zz = zl[0], ll = zl[1];
      st2(R, sZB, zz, ll);  // xbar <- x: the displacement starts again from zero
      st(R, sL, 0.0);
      double zb[NS], lnb[NX];
      bc_all<NS, RQ>(zz, zb);
      bc_all<NX, RQ>(zln[1], lnb);
      double s = fh[0] + dot4<NS>(Kr, zb);
      s += dot4<NX>(ABc, lnb) - (rx ? ll : 0.0);
      {
        double p[4] = {s, 0.0, 0.0, 0.0};
        {
          double vv[KS];
          sfor<0, KS>([&](auto S_) { vv[decltype(S_)::value] = vy[decltype(S_)::value][0]; });
          bc_cols_dot<NC, RQ>(Cc, vv, p);
        }
        s = (p[0] + p[1]) + (p[2] + p[3]);
      }
      st2(R, sZ, 0.0, s);
      s_nat = fma(s, s, s_nat);
      if (i == 0) {
        const double rl0 = rx ? fh[1] + zz : 0.0;
        st(R, sRL, rl0);
        s_nat = fma(rl0, rl0, s_nat);
      }
      if (i < N_) {
        const double abz = dot4<NS>(ABr, zb);
        const double rln = rx ? hn - (abz - zln[0]) : 0.0;
        st(R + kRec, sRL, rln);
        s_nat = fma(rln, rln, s_nat);
      }
      {
        double vbs[KS], ybs[KS];
        sfor<0, KS>([&](auto S_) { vbs[decltype(S_)::value] = vy[decltype(S_)::value][0]; ybs[decltype(S_)::value] = vy[decltype(S_)::value][1]; });
        stv<sVB, KS>(R, vbs);
        stv<sYB, KS>(R, ybs);
      }
      sfor<0, KS>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        const double pn = pnr(vy[sl][1], vy[sl][0], alpha);  // zero on padding lanes
        const double pf = pfb(vy[sl][1], vy[sl][0], alpha);
        s_vo = fma(pn, pn, s_vo);
        s_vi = fma(pf, pf, s_vi);
      });
      zl = zln;
    }
    c.sync();
    const double nat = qp_reduce<RQ, OpSum16>(s_nat);
    *Ek = sqrt(nat + qp_reduce<RQ, OpSum16>(s_vo));
    *Ei0 = sqrt(nat + qp_reduce<RQ, OpSum16>(s_vi));
  }

  // ---- the proximal-level passes run by ALL rows for ONE QP -----------------------------
  // open_prox() and close_subproblem() are passes over the stages without a recurrence:
  // stage i needs nothing but the records of stages i - 1, i, i + 1 and the stage's matrix
  // copy.  Run by the row that owns the QP they keep a quarter of the wavefront busy while
  // the other rows wait (the rows of a wavefront stand at different points of their
  // solves); here row q takes the stages q, q + QW, q + 2 QW, ... of the OWNER's records, as
  // trial_pass_coop() does for the line search.  The matrix rows come straight from the
  // owner's matrix copy in global memory (a few KB that every row of the wavefront reads
  // again and again: L1), NOT through the helper rows' LDS images of their own copies -
  // restaging those after every pass is what made the first attempt at this a loss (round 2).
  // Same terms as the owner's passes; the sums over the stages are formed per row and then
  // over the rows.  Real calls, like the trial pass.
#ifndef FB_R16_COOP_PROX
#define FB_R16_COOP_PROX 1
#endif
  static constexpr bool kCoopProx = FB_R16_COOP_PROX != 0 && kCoopTrials;
  struct OwnerView {
    double* R0;        // the owner's records, this lane's offset within ITS row included
    const double* P0;  // the owner's matrix copies, likewise
    lds_iptr po;       // the owner's table of matrix-copy offsets (LDS)
    int N;
  };
  FB_DEV OwnerView owner_view(int owner) const {
    const int own0 = owner & ~(LPQ - 1);
    auto lane64 = [&](unsigned long long x) {
      return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(x >> 32), own0) << 32) |
             (unsigned)__builtin_amdgcn_readlane((int)x, own0);
    };
    OwnerView o;
    const int r2 = 2 * (threadIdx.x & (LPQ - 1));
    o.R0 = reinterpret_cast<double*>(lane64((unsigned long long)rec)) + r2;
    o.P0 = reinterpret_cast<const double*>(lane64((unsigned long long)pack)) + r2;
    o.po = (lds_iptr)(unsigned long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long)lpo, own0);
    o.N = __builtin_amdgcn_readlane(N, own0);
    return o;
  }
  static FB_DEV double rows_sum(double x) {  // sum over the QP's lanes, then over the rows: every lane gets it
    const double rs = qp_reduce<RQ, OpSum16>(x);
    double tot = lane_value(rs, 0);
    sfor<1, kQpPerWave>([&](auto Q_) { tot += lane_value(rs, LPQ * decltype(Q_)::value); });
    return tot;
  }
  static FB_DEV double rows_max(double x) {
    const double rs = qp_reduce<RQ, OpMax16>(x);
    double tot = lane_value(rs, 0);
    sfor<1, kQpPerWave>([&](auto Q_) { tot = fmax(tot, lane_value(rs, LPQ * decltype(Q_)::value)); });
    return tot;
  }
  static FB_DEV void pass_fence() {  // records written by one row, read by another
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }

  struct OpenSums {
    double nat, vo, vi;
  };
  // open_prox() for the owner's QP (same statements per stage)
  static __device__ __attribute__((noinline)) OpenSums open_pass_coop(double* R0, const double* P0, lds_iptr po,
                                                                      int N_, double alpha) {
    constexpr int QW = kQpPerWave;
    const int lane = threadIdx.x & 63;
    const int q = lane / LPQ, r = lane & (LPQ - 1);
    const bool rx = r < NX;
    double s_nat = 0.0, s_vo = 0.0, s_vi = 0.0;
    // What a trip reads, all of it requested at the top of the trip.  (Measured and dropped,
    // gpurun_out/r03_s, r03_t: the record part requested a trip ahead, 478 k against 518 k
    // QP/s - like every other attempt to put more loads in flight in this kernel -, and the
    // matrix rows a trip ahead as well: the 64 doubles went to scratch memory, 330 k.
    // gpurun_out/r03_ao: the next trip's record slots requested between this trip's
    // arithmetic and its stores - ahead of the stores in the memory queue, held in registers
    // across the stores only - in this pass and in close_pass_coop: 485 k against 533 k.)
    struct In {
      dbl2 fh, vy[KS];
      double zz, ll, zn, ln, hn;
    };
    auto load = [&](int i, In& in) {
      const int ii = i <= N_ ? i : N_;
      const double* R = R0 + (long)ii * kRec;
      const double* Rn = R + (ii < N_ ? kRec : 0);
      in.fh = ld2(R, sF);
      sfor<0, KS>([&](auto S_) { in.vy[decltype(S_)::value] = ld2(R, sV + 2 * decltype(S_)::value); });
      // x = xbar + displacement, of this stage and the next (all of it read before the trip stores
      // anything: the row that takes stage i + 1 in this very trip rewrites what zn, ln are read from)
      const dbl2 cb = ld2(R, sZB), cbn = ld2(Rn, sZB);
      in.zz = cb[0] + ld(R, sZ);
      in.ll = cb[1] + ld(R, sL);
      in.zn = cbn[0] + ld(Rn, sZ);
      in.ln = cbn[1] + ld(Rn, sL);
      in.hn = ld(Rn, sH);
    };
    for (int i = q; i - q <= N_; i += QW) {  // (the same trip count in every row)
      FB_PHASE(trip_top);
      const bool live = i <= N_;
      const int ii = live ? i : N_;
      const bool has_next = ii < N_;
      double* R = R0 + (long)ii * kRec;
      In cu;
      load(i, cu);
      double Kr[NS], Cc[NC], ABr[NS], ABc[NX];
      {
        const double* Pk = P0 + po[ii];
        ldv<pK, NS>(Pk, Kr);
        ldv<pC, NC>(Pk, Cc);
        ldv<pABr, NS>(Pk, ABr);
        ldv<pABc, NX>(Pk, ABc);
      }
      const dbl2 fh = cu.fh;
      dbl2 vy[KS];
      sfor<0, KS>([&](auto S_) { vy[decltype(S_)::value] = cu.vy[decltype(S_)::value]; });
      const double zz = cu.zz, ll = cu.ll;
      double zn = cu.zn, ln = cu.ln, hn = cu.hn;
      if (!has_next) zn = ln = hn = 0.0;
      double zb[NS], lnb[NX];
      bc_all<NS, RQ>(zz, zb);
      bc_all<NX, RQ>(ln, lnb);
      double s = fh[0] + dot4<NS>(Kr, zb);
      s += dot4<NX>(ABc, lnb) - (rx ? ll : 0.0);
      {
        double p[4] = {s, 0.0, 0.0, 0.0};
        {
          double vv[KS];
          sfor<0, KS>([&](auto S_) { vv[decltype(S_)::value] = vy[decltype(S_)::value][0]; });
          bc_cols_dot<NC, RQ>(Cc, vv, p);
        }
        s = (p[0] + p[1]) + (p[2] + p[3]);
      }
      const double rl0 = rx ? fh[1] + zz : 0.0;
      const double abz = dot4<NS>(ABr, zb);
      const double rln = rx ? hn - (abz - zn) : 0.0;
      double vo = 0.0, vi = 0.0;
      sfor<0, KS>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        const double pn = pnr(vy[sl][1], vy[sl][0], alpha);  // zero on padding lanes
        const double pf = pfb(vy[sl][1], vy[sl][0], alpha);
        vo = fma(pn, pn, vo);
        vi = fma(pf, pf, vi);
      });
      if (live) {
        st2(R, sZB, zz, ll);  // xbar <- x: the displacement starts again from zero
        st2(R, sZ, 0.0, s);
        st(R, sL, 0.0);
        double nat = s * s;
        if (i == 0) {
          st(R, sRL, rl0);
          nat = fma(rl0, rl0, nat);
        }
        if (has_next) {
          st(R + kRec, sRL, rln);
          nat = fma(rln, rln, nat);
        }
        double vbs[KS], ybs[KS];
        sfor<0, KS>([&](auto S_) { vbs[decltype(S_)::value] = vy[decltype(S_)::value][0]; ybs[decltype(S_)::value] = vy[decltype(S_)::value][1]; });
        stv<sVB, KS>(R, vbs);
        stv<sYB, KS>(R, ybs);
        s_nat += nat;
        s_vo += vo;
        s_vi += vi;
      }
      FB_PHASE(trip_end);
    }
    OpenSums o;
    o.nat = rows_sum(s_nat);
    o.vo = rows_sum(s_vo);
    o.vi = rows_sum(s_vi);
    return o;
  }
  // every lane of the wavefront calls this; every lane gets the owner's norms
  // nat2: the squared norm of the natural residual's z and l blocks at x = xbar (where it is the inner
  // residual's too): the line search's A of trial_zl() at the start of a subproblem
  FB_DEV void open_prox_coop(int owner, double alpha, double* Ek, double* Ei0, double* nat2) const {
    FB_WAVE_COUNT(25);
    FB_WAVE_TIMER(22);
    pass_fence();
    const OwnerView ov = owner_view(owner);
    const OpenSums o = open_pass_coop(ov.R0, ov.P0, ov.po, ov.N, alpha);
    pass_fence();
    *Ek = sqrt(o.nat + o.vo);
    *Ei0 = sqrt(o.nat + o.vi);
    *nat2 = o.nat;
  }

  struct CloseSums {
    double dx2, m_adz, m_gdz, m_hdz, m_dz, m_atv, m_u, s_fdz, s_p2;
  };
  // close_subproblem() for the owner's QP: t is the owner's pending step; Cl is THIS row's
  // transpose buffer (not the resident matrix copy)
  static __device__ __attribute__((noinline)) CloseSums close_pass_coop(double* R0, const double* P0, lds_iptr po,
                                                                        lds_ptr Cl, int N_, double t, bool check) {
    constexpr int QW = kQpPerWave;
    const int lane = threadIdx.x & 63;
    const int q = lane / LPQ, r = lane & (LPQ - 1);
    const bool rx = r < NX;
    double m_adz = -1e300, m_gdz = 0.0, m_hdz = 0.0, m_dz = 0.0, m_atv = 0.0, m_u = 0.0;
    double s_fdz = 0.0, s_p2 = 0.0, s_dx = 0.0;
    C cc_;
    cc_.tid = r;
    // What a trip reads, all of it requested at the top of the trip, before the trip stores
    // anything: the row that takes stage i + 1 in this very trip rewrites what `nxt` is read
    // from.  (WLN, from which the stage above takes its wl a trip later, is left as it is.)
    struct In {
      dbl2 zr, dw, lr, dwl;       // stage i
      dbl2 nzr, ndw, nlr, ndwl;   // stage i + 1
      double wlp;                     // WLN(i - 1)
      dbl2 vy[KS], da[KS], fh;
      double vb[KS], bs[KS];
    };
    auto load = [&](int i, In& in) {
      const int ii = i <= N_ ? i : N_;
      const double* R = R0 + (long)ii * kRec;
      const double* Rn = R + (ii < N_ ? kRec : 0);
      in.zr = ld2(R, sZ); in.dw = ld2(R, sDZ); in.lr = ld2(R, sL); in.dwl = ld2(R, sDL);
      in.nzr = ld2(Rn, sZ); in.ndw = ld2(Rn, sDZ); in.nlr = ld2(Rn, sL); in.ndwl = ld2(Rn, sDL);
      in.wlp = ld(R - (ii > 0 ? kRec : 0), sWLN);
      sfor<0, KS>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        in.vy[sl] = ld2(R, sV + 2 * sl);
        in.da[sl] = ld2(R, sDV + 2 * sl);
      });
      ldv<sVB, KS>(R, in.vb);
      if (check) {
        in.fh = ld2(R, sF);
        sfor<0, KS>([&](auto S_) { in.bs[decltype(S_)::value] = ld(R, sB + decltype(S_)::value); });
      }
    };
    for (int i = q; i - q <= N_; i += QW) {
      FB_PHASE(trip_top);
      const bool live = i <= N_;
      const int ii = live ? i : N_;
      const bool has_next = ii < N_;
      double* R = R0 + (long)ii * kRec;
      In cu;
      load(i, cu);
      double Kr[NS], Cc[NC], ABr[NS], ABc[NX];
      if (check) {
        const double* Pk = P0 + po[ii];
        ldv<pK, NS>(Pk, Kr);
        ldv<pC, NC>(Pk, Cc);
        ldv<pABr, NS>(Pk, ABr);
        ldv<pABc, NX>(Pk, ABc);
      }
      ZL cur;
      {
        const double wl = wl_of_stage(ii, rx, cu.dw[0], cu.wlp);
        cur.z = fma(t, cu.dw[0], cu.zr[0]);
        cur.rz = fma(t, cu.dw[1], cu.zr[1]);
        cur.l = fma(t, cu.dwl[0], cu.lr[0]);
        cur.rl = fma(t, wl, cu.lr[1]);
        cur.dz = cur.z;  // (the displacements ez, el are x - xbar already)
        cur.dl = cur.l;
      }
      ZL nxt;
      nxt.dz = has_next ? fma(t, cu.ndw[0], cu.nzr[0]) : 0.0;
      nxt.dl = has_next ? fma(t, cu.ndwl[0], cu.nlr[0]) : 0.0;
      dbl2 vy[KS], da[KS];
      double vb[KS];
      sfor<0, KS>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        vy[sl] = cu.vy[sl];
        da[sl] = cu.da[sl];
        vb[sl] = cu.vb[sl];
      });
      dbl2 fh = {0.0, 0.0};
      double bs[KS];
      sfor<0, KS>([&](auto S_) { bs[decltype(S_)::value] = 0.0; });
      if (check) {
        fh = cu.fh;
        sfor<0, KS>([&](auto S_) { bs[decltype(S_)::value] = cu.bs[decltype(S_)::value]; });
      }
      double dx2 = fma(cur.dz, cur.dz, cur.dl * cur.dl);
      double dvs[KS], vvs[KS], yys[KS];
      sfor<0, KS>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        vvs[sl] = fmax0(fma(t, da[sl][0], vy[sl][0]));
        yys[sl] = fma(-t, da[sl][1], vy[sl][1]);
        dvs[sl] = vvs[sl] - vb[sl];
        dx2 = fma(dvs[sl], dvs[sl], dx2);
      });
      double l_adz = -1e300, l_gdz = 0.0, l_hdz = 0.0, l_dz = 0.0, l_atv = 0.0, l_u = 0.0, l_fdz = 0.0, l_p2 = 0.0;
      if (check) {
        double dzb[NS], dlnb[NX];
        bc_all<NS, RQ>(cur.dz, dzb);
        bc_all<NX, RQ>(nxt.dl, dlnb);
        l_hdz = fabs(dot4<NS>(Kr, dzb));
        l_dz = fabs(cur.dz);
        {
          double p[4] = {dot4<NX>(ABc, dlnb) - (rx ? cur.dl : 0.0), 0.0, 0.0, 0.0};
          bc_cols_dot<NC, RQ>(Cc, dvs, p);
          l_atv = fabs((p[0] + p[1]) + (p[2] + p[3]));
        }
        l_fdz = fh[0] * cur.dz;
        l_u = fabs(cur.dl);
        l_p2 = fh[1] * cur.dl;
        if (ii == 0) l_gdz = rx ? fabs(cur.dz) : 0.0;
        if (has_next) {
          const double g = dot4<NS>(ABr, dzb) - nxt.dz;
          l_gdz = fmax(l_gdz, rx ? fabs(g) : 0.0);
        }
        cc_.sync();
        sfor<0, NC>([&](auto Kk) { Cl[r * CS + decltype(Kk)::value] = Cc[decltype(Kk)::value]; });
        cc_.sync();
        sfor<0, KS>([&](auto S_) {
          constexpr int sl = decltype(S_)::value;
          const int k = r + LPQ * sl;
          const int kk = k < NC ? k : 0;
          double clk[NS];
          sfor<0, NS>([&](auto Cc2) { clk[decltype(Cc2)::value] = Cl[decltype(Cc2)::value * CS + kk]; });
          const double az = dot4<NS>(clk, dzb);
          if (k < NC) l_adz = fmax(l_adz, az);
          l_u = fmax(l_u, fabs(dvs[sl]));
          l_p2 = fma(bs[sl], dvs[sl], l_p2);
        });
      }
      if (live) {
        st2(R, sZ, cur.z, cur.rz);
        st2(R, sL, cur.l, cur.rl);
        st2(R, sDZ, cur.dz, 0.0);
        st2(R, sDL, cur.dl, cu.dwl[1]);  // (no step is pending from here on: WLN is dead until the next sweep writes it)
        sfor<0, KS>([&](auto S_) {
          constexpr int sl = decltype(S_)::value;
          st2(R, sV + 2 * sl, vvs[sl], yys[sl]);
          st2(R, sDV + 2 * sl, dvs[sl], 0.0);
        });
        s_dx += dx2;
        m_adz = fmax(m_adz, l_adz);
        m_gdz = fmax(m_gdz, l_gdz);
        m_hdz = fmax(m_hdz, l_hdz);
        m_dz = fmax(m_dz, l_dz);
        m_atv = fmax(m_atv, l_atv);
        m_u = fmax(m_u, l_u);
        s_fdz += l_fdz;
        s_p2 += l_p2;
      }
      FB_PHASE(trip_end);
    }
    CloseSums o;
    o.dx2 = rows_sum(s_dx);
    o.m_adz = rows_max(m_adz);
    o.m_gdz = rows_max(m_gdz);
    o.m_hdz = rows_max(m_hdz);
    o.m_dz = rows_max(m_dz);
    o.m_atv = rows_max(m_atv);
    o.m_u = rows_max(m_u);
    o.s_fdz = rows_sum(s_fdz);
    o.s_p2 = rows_sum(s_p2);
    return o;
  }
  // every lane calls this; every lane gets the owner's verdict and ||dx||.  The owner's
  // pending step is consumed (its row clears pend_t).
  FB_DEV int close_subproblem_coop(int owner, double tol, bool check, double* dx_norm) {
    FB_WAVE_COUNT(26);
    FB_WAVE_TIMER(21);
    pass_fence();
    const OwnerView ov = owner_view(owner);
    const int own0 = owner & ~(LPQ - 1);
    const double t = lane_value(pend_t, own0);
    if (((threadIdx.x ^ owner) & 63 & ~(LPQ - 1)) == 0) pend_t = 0.0;
    const CloseSums o = close_pass_coop(ov.R0, ov.P0, ov.po, lds, ov.N, t, check);
    pass_fence();
    *dx_norm = sqrt(o.dx2);
    if (!check) return kFeasible;
    bool dual_feasible = true, primal_feasible = true;
    if ((o.m_adz <= o.m_dz * tol) && (o.m_gdz <= tol * o.m_dz) && (o.m_hdz <= tol * o.m_dz) && (o.s_fdz < 0) &&
        (o.m_dz > 1e-14))
      dual_feasible = false;
    if ((o.m_atv <= tol * o.m_u) && (o.s_p2 < 0)) primal_feasible = false;
    if (primal_feasible && dual_feasible) return kFeasible;
    if (primal_feasible && !dual_feasible) return kDualInfeasible;
    if (!primal_feasible && dual_feasible) return kPrimalInfeasible;
    return kBothInfeasible;
  }

  // ---- load_guess() for the owner's QP, run by all rows (FB_R16_COOP_LOAD) ---------------
  // Row q packs stages q, q + QW, ...: the stage's matrices and vectors from the caller's
  // arrays, its records, y = b - A z.  Whether a stage shares the matrix copy of the stage
  // before it is decided by reading that stage's matrices as well (the sequential pass
  // carries them from stage to stage); the offsets of the copies in use - "this stage's own"
  // or "whatever the stage before uses" - follow from one scan over the stages afterwards.
#ifndef FB_R16_COOP_LOAD
#define FB_R16_COOP_LOAD 1
#endif
  static constexpr bool kCoopLoad = FB_R16_COOP_LOAD != 0 && kCoopProx && !KEEP;
  struct LoadSums {
    double c2m, hm;
    int nzm;
  };
  static __device__ __attribute__((noinline)) LoadSums load_pass_coop(double* R0, double* P0, int* pog, lds_iptr lp,
                                                                      lds_ptr Cl, const MpcBatchPtrs* data,
                                                                      const VarBatchPtrs* var, long qp, int N_,
                                                                      int nx_, int nu_, int nc_) {
    constexpr int QW = kQpPerWave;
    const int lane = threadIdx.x & 63;
    const int qr = lane / LPQ, r = lane & (LPQ - 1);
    const int ru = r - NX;
    const bool rx = r < nx_;
    const bool rin = r >= NX && ru < nu_;
    const bool rs_ = rx || rin;
    auto arr_ = [&](int a) { return data->base[a] + qp * data->stride[a]; };
    const double *Q = arr_(FBSTAB_MPC_Q), *Rm = arr_(FBSTAB_MPC_R), *S = arr_(FBSTAB_MPC_S), *pq = arr_(FBSTAB_MPC_q),
                 *pr = arr_(FBSTAB_MPC_r), *A = arr_(FBSTAB_MPC_A), *B = arr_(FBSTAB_MPC_B), *pc = arr_(FBSTAB_MPC_c),
                 *E = arr_(FBSTAB_MPC_E), *L = arr_(FBSTAB_MPC_L), *pd = arr_(FBSTAB_MPC_d), *px0 = arr_(FBSTAB_MPC_x0);
    const double *uz = var->base[0] + qp * var->stride[0], *ul = var->base[1] + qp * var->stride[1],
                 *uv = var->base[2] + qp * var->stride[2];
    C cc_;
    cc_.tid = r;
    int nzm = 0;
    double c2m = 0.0, hm = 0.0;
    struct Mats {
      double Kr[NSP], ABr[NSP], Cc[NC], ABc[NX];
    };
    auto load_mats = [&](int i, Mats& m) {
      const bool has_ab = i < N_;
      sfor<0, NSP>([&](auto Cc_) {
        constexpr int cc = decltype(Cc_)::value;
        double kv = 0.0, ab = 0.0;
        if constexpr (cc < NX) {
          if (cc < nx_) {  // a real state column
            if (rx) kv = Q[(long)i * nx_ * nx_ + r + cc * nx_];
            else if (rin) kv = S[(long)i * nu_ * nx_ + ru + cc * nu_];
            if (rx && has_ab) ab = A[(long)i * nx_ * nx_ + r + cc * nx_];
          }
        } else if constexpr (cc < NS) {
          if (cc - NX < nu_) {  // a real input column
            if (rx) kv = S[(long)i * nu_ * nx_ + (long)r * nu_ + (cc - NX)];
            else if (rin) kv = Rm[(long)i * nu_ * nu_ + ru + (cc - NX) * nu_];
            if (rx && has_ab) ab = B[(long)i * nx_ * nu_ + r + (cc - NX) * nx_];
          }
        }
        m.Kr[cc] = kv;
        m.ABr[cc] = ab;
      });
      {
        const double* src = rx ? E + ((long)i * nx_ + r) * nc_ : L + ((long)i * nu_ + (rin ? ru : 0)) * nc_;
        sfor<0, NC>([&](auto Kk) {
          constexpr int k = decltype(Kk)::value;
          m.Cc[k] = (rs_ && k < nc_) ? src[k] : 0.0;
        });
      }
      {
        const double* src = rx ? A + (long)i * nx_ * nx_ + (long)r * nx_
                               : B + (long)i * nx_ * nu_ + (long)(rin ? ru : 0) * nx_;
        sfor<0, NX>([&](auto J) {
          constexpr int j = decltype(J)::value;
          m.ABc[j] = (rs_ && has_ab && j < nx_) ? src[j] : 0.0;
        });
      }
    };
    for (int i = qr; i - qr <= N_; i += QW) {  // (the same trip count in every row)
      FB_PHASE(trip_top);
      const bool live = i <= N_;
      const int ii = live ? i : N_;
      double* R = R0 + (long)ii * kRec;
      double* PK = P0 + (long)ii * kPack;
      // the stage's vectors, with the matrix loads (see load_guess)
      const double f = rx ? pq[(long)ii * nx_ + r] : (rin ? pr[(long)ii * nu_ + ru] : 0.0);
      const double h = rx ? (ii == 0 ? -px0[r] : -pc[(long)(ii - 1) * nx_ + r]) : 0.0;
      const double zz = rx ? uz[(long)ii * (nx_ + nu_) + r] : (rin ? uz[(long)ii * (nx_ + nu_) + nx_ + ru] : 0.0);
      const double ll = rx ? ul[(long)ii * nx_ + r] : 0.0;
      double bs[KS], vs[KS];
      sfor<0, KS>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        const int k = r + LPQ * sl;
        const bool real = k < NC && k < nc_;
        bs[sl] = real ? -pd[(long)ii * nc_ + k] : 0.0;
        vs[sl] = real ? uv[(long)ii * nc_ + k] : 0.0;
      });
      Mats m, mp;
      load_mats(ii, m);
      load_mats(ii > 0 ? ii - 1 : 0, mp);
      {
        const int rowsh = LPQ * qr;
        double c2 = 0.0, hs = 0.0;
        int nz = 0;
        sfor<0, NC>([&](auto Kk) {
          const unsigned long long mk = __ballot(m.Cc[decltype(Kk)::value] != 0.0);
          nz = max(nz, __popc((unsigned)(mk >> rowsh) & (unsigned)((1ull << LPQ) - 1ull)));
          c2 = fma(m.Cc[decltype(Kk)::value], m.Cc[decltype(Kk)::value], c2);
        });
        sfor<0, NSP>([&](auto Cc_) { hs += fabs(m.Kr[decltype(Cc_)::value]); });
        if (live) {
          nzm = max(nzm, nz);
          c2m = fmax(c2m, c2);
          hm = fmax(hm, hs);
        }
      }
      bool differs = ii == 0;
      sfor<0, NSP>([&](auto Cc_) {
        constexpr int cc = decltype(Cc_)::value;
        differs = differs || !(m.Kr[cc] == mp.Kr[cc]) || !(m.ABr[cc] == mp.ABr[cc]);
      });
      sfor<0, NC>([&](auto Kk) { differs = differs || !(m.Cc[decltype(Kk)::value] == mp.Cc[decltype(Kk)::value]); });
      sfor<0, NX>([&](auto J) { differs = differs || !(m.ABc[decltype(J)::value] == mp.ABc[decltype(J)::value]); });
      const bool own_copy = qp_reduce<RQ, OpMax16>(differs ? 1.0 : 0.0) > 0.0;
      double zb[NS];
      bc_all<NS, RQ>(zz, zb);
      cc_.sync();
      sfor<0, NC>([&](auto Kk) { Cl[r * CS + decltype(Kk)::value] = m.Cc[decltype(Kk)::value]; });
      cc_.sync();
      double az[KS];
      sfor<0, KS>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        const int k = r + LPQ * sl;
        const int kk = k < NC ? k : 0;
        double clk[NS];
        sfor<0, NS>([&](auto Cc2) { clk[decltype(Cc2)::value] = Cl[decltype(Cc2)::value * CS + kk]; });
        az[sl] = dot4<NS>(clk, zb);
      });
      if (live) {
        if (own_copy) {
          stv<pK, NSP>(PK, m.Kr);
          stv<pABr, NSP>(PK, m.ABr);
          stv<pC, NC>(PK, m.Cc);
          stv<pABc, NX>(PK, m.ABc);
        }
        if (r == 0) lp[ii] = own_copy ? ii * kPack : -1;
        st2(R, sF, f, h);
        st2(R, sZ, 0.0, 0.0);
        st2(R, sZB, zz, ll);
        st2(R, sL, 0.0, 0.0);
        st2(R, sDZ, 0.0, 0.0);
        st2(R, sDL, 0.0, 0.0);
        sfor<0, KS>([&](auto S_) {
          constexpr int sl = decltype(S_)::value;
          const int k = r + LPQ * sl;
          const bool real = k < NC && k < nc_;
          st(R, sB + sl, bs[sl]);
          st2(R, sV + 2 * sl, vs[sl], real ? bs[sl] - az[sl] : 0.0);
          st2(R, sDV + 2 * sl, 0.0, 0.0);
        });
      }
      FB_PHASE(trip_end);
    }
    zero_spare(R0, N_);  // (every row writes the same zeros)
    LoadSums o;
    o.c2m = rows_max(c2m);
    o.hm = rows_max(hm);
    o.nzm = (int)rows_max((double)nzm);
    // the copy each stage reads: its own, or the one the stage before it reads
    cc_.sync();
    if (lane == 0) {
      int prev = 0;
      for (int i = 0; i <= N_; i++) {
        int v = lp[i];
        if (v < 0) v = prev;
        lp[i] = v;
        pog[i] = v;
        prev = v;
      }
      pog[N_ + 1] = o.nzm;
      pog[N_ + 2] = __float_as_int((float)o.c2m);
      pog[N_ + 3] = __float_as_int((float)o.hm);
    }
    cc_.sync();
    return o;
  }
  // every lane of the wavefront calls this; the owner's row is left as load_guess() leaves it
  FB_DEV void load_guess_coop(int owner) {
    FB_WAVE_TIMER(19);
    pass_fence();
    const OwnerView ov = owner_view(owner);
    const int own0 = owner & ~(LPQ - 1);
    auto lane64 = [&](unsigned long long x) {
      return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(x >> 32), own0) << 32) |
             (unsigned)__builtin_amdgcn_readlane((int)x, own0);
    };
    const int r2 = 2 * (threadIdx.x & (LPQ - 1));
    double* P0w = reinterpret_cast<double*>(lane64((unsigned long long)pack)) + r2;
    int* pog = reinterpret_cast<int*>(lane64((unsigned long long)poff));
    const MpcBatchPtrs* d = reinterpret_cast<const MpcBatchPtrs*>(lane64((unsigned long long)data));
    const VarBatchPtrs* x = reinterpret_cast<const VarBatchPtrs*>(lane64((unsigned long long)var));
    const long qp = (long)lane64((unsigned long long)q);
    const int nx_ = EXACT ? NX : d->nx, nu_ = EXACT ? NU : d->nu, nc_ = EXACT ? NC : d->nc;
    const LoadSums o = load_pass_coop(ov.R0, P0w, pog, ov.po, lds, d, x, qp, ov.N, nx_, nu_, nc_);
    pass_fence();
    if (((threadIdx.x ^ owner) & 63 & ~(LPQ - 1)) == 0) {
      pend_t = 0.0;
      lds_off = -1;
      cmax2 = (float)o.c2m;
      hmax = (float)o.hm;
      nzmax = o.nzm;
      bounds = o.nzm <= 1;
    }
  }

  // ---- results ---------------------------------------------------------------------
  // which: 0 = x, 1 = xbar, 2 = certificate dx with dx.y = y - ybar + b
  // (impl:202-210, full_variable.cc:55-65)
  template <int WHICH>
  FB_DEV void write_out(const C& c) const {
    const int r = c.tid, N_ = N;
    const int nx_ = prob_nx(), nu_ = prob_nu(), nc_ = prob_nc();
    const double* const R0 = rec;
    double *uz = xarr(0), *ul = xarr(1), *uv = xarr(2), *uy = xarr(3);
    // (Measured and dropped, gpurun_out/r03_am: all record slots of a stage - and of two
    // stages - requested before the first store to the caller's arrays, as in load_guess:
    // 518 k against 520 k QP/s.  The stores here are fire-and-forget and the next stage's
    // loads do not wait for them.)
    for (int i = 0; i <= N_; i++) {
      FB_PHASE(write_top);
      const double* R = R0 + (long)i * kRec;
      double zz = ld(R, WHICH == 0 ? sZ : (WHICH == 1 ? sZB : sDZ));
      double ll = ld(R, WHICH == 0 ? sL : (WHICH == 1 ? sLB : sDL));
      if constexpr (WHICH == 0) {  // x = xbar + displacement
        const dbl2 cb = ld2(R, sZB);
        zz = cb[0] + zz;
        ll = cb[1] + ll;
      }
      if (r < nx_) uz[(long)i * (nx_ + nu_) + r] = zz;
      else if (r >= NX && r - NX < nu_) uz[(long)i * (nx_ + nu_) + nx_ + (r - NX)] = zz;
      if (r < nx_) ul[(long)i * nx_ + r] = ll;
      sfor<0, KS>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        const int k = r + LPQ * sl;
        const dbl2 vy = ld2(R, sV + 2 * sl);
        double vv = vy[0], yy = vy[1];
        if (WHICH == 1) {
          vv = ld(R, sVB + sl);
          yy = ld(R, sYB + sl);
        } else if (WHICH == 2) {
          vv = ld(R, sDV + 2 * sl);
          yy = (vy[1] - ld(R, sYB + sl)) + ld(R, sB + sl);
        }
        if (k < nc_) {
          uv[(long)i * nc_ + k] = vv;
          uy[(long)i * nc_ + k] = yy;
        }
      });
    }
  }
  FB_DEV void write_x(const C& c) const { write_out<0>(c); }
  FB_DEV void write_xbar(const C& c) const { write_out<1>(c); }
  FB_DEV void write_certificate(const C& c) const { write_out<2>(c); }

  // ---- diagnostics (tests): xbar in, one Newton step's vectors out ------------------
  // index of this lane's z element of stage i in the caller's z, or -1
  FB_DEV long z_index(int i, int r) const {
    const int nx = prob_nx(), nu = prob_nu();
    if (r < nx) return (long)i * (nx + nu) + r;
    if (r >= NX && r - NX < nu) return (long)i * (nx + nu) + nx + (r - NX);
    return -1;
  }
  FB_DEV void probe_set_xbar(const C& c, const double* dbg) const {
    const int r = c.tid, N_ = N;
    const int nx_ = prob_nx(), nu_ = prob_nu(), nc_ = prob_nc();
    double* const R0 = rec;
    const long nz = (long)(N_ + 1) * (nx_ + nu_), nl = (long)(N_ + 1) * nx_;
    for (int i = 0; i <= N_; i++) {
      double* R = R0 + (long)i * kRec;
      const long zi = z_index(i, r);
      // (load_guess left the guess as the centre, with a zero displacement: x stays, xbar moves)
      const dbl2 x0 = ld2(R, sZB);
      const double zbn = zi >= 0 ? dbg[zi] : 0.0, lbn = r < nx_ ? dbg[nz + (long)i * nx_ + r] : 0.0;
      st2(R, sZB, zbn, lbn);
      st(R, sZ, x0[0] - zbn);
      st(R, sL, x0[1] - lbn);
      sfor<0, KS>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        const int k = r + LPQ * sl;
        st(R, sVB + sl, k < nc_ ? dbg[nz + nl + (long)i * nc_ + k] : 0.0);
      });
    }
    c.sync();
  }
  FB_DEV void probe_dump(const C& c, double* o, bool ok) const {
    const int r = c.tid, N_ = N;
    const int nx_ = prob_nx(), nu_ = prob_nu(), nc_ = prob_nc();
    const double* const R0 = rec;
    const long nz = (long)(N_ + 1) * (nx_ + nu_), nl = (long)(N_ + 1) * nx_, nv = (long)(N_ + 1) * nc_;
    for (int i = 0; i <= N_; i++) {
      const double* R = R0 + (long)i * kRec;
      const long zi = z_index(i, r);
      if (zi >= 0) {
        o[zi] = ld(R, sDZ);
        o[nz + nl + 2 * nv + zi] = ld(R, sWZ);
        o[2 * nz + 2 * nl + 2 * nv + zi] = ld(R, sRZ);
      }
      if (r < nx_) {
        o[nz + (long)i * nx_ + r] = ld(R, sDL);
        o[2 * nz + nl + 2 * nv + (long)i * nx_ + r] = i == 0 ? ld(R, sDZ) : ld(R - kRec, sWLN);
        o[3 * nz + 2 * nl + 2 * nv + (long)i * nx_ + r] = ld(R, sRL);
      }
      sfor<0, KS>([&](auto S_) {
        constexpr int sl = decltype(S_)::value;
        const int k = r + LPQ * sl;
        if (k < nc_) {
          o[nz + nl + (long)i * nc_ + k] = ld(R, sDV + 2 * sl);
          o[nz + nl + nv + (long)i * nc_ + k] = ld(R, sDV + 2 * sl + 1);
        }
      });
    }
    if (r == 0) o[3 * nz + 3 * nl + 2 * nv] = ok ? 1.0 : 0.0;
  }

  // ---- the Newton step ----------------------------------------------------------------
  // Forward sweep: applies the pending step, factors, forward substitution.
  // Backward sweep: back substitution fused with dv, A dz, the residual
  // increment W = (H dz + G'dl + A'dv, -G dz) and the squared norms of the inner
  // and penalised natural residuals at x + dx (the first line-search trial,
  // fbstab_algorithm-impl.h:283-290).  Returns false on a non-positive pivot
  // (riccati_linear_solver.cc:131-136).
  struct FwdIn {
    dbl2 zr, dw, lr, dwl;
    dbl2 vy[KS], da[KS];
    double vb[KS];
  };
  static FB_DEV void load_fwd(const double* R, const double* Rd, FwdIn& in) {
    in.zr = ld2(R, sZ);
    in.dw = ld2(R, sDZ);
    in.lr = ld2(R, sL);
    in.dwl = ld2(R, sDL);
    const double* const Rt = tail_of(R, Rd);
    sfor<0, KS>([&](auto S_) {
      constexpr int s = decltype(S_)::value;
      in.vy[s] = ld2(s == KS - 1 ? Rt : R, sV + 2 * s);
      in.da[s] = ld2(s == KS - 1 ? Rt : R, sDV + 2 * s);
    });
    ldv<sVB, KS>(R, in.vb);
  }
  // What a backward stage reads of its own iterate vectors (fetched a stage ahead).
  struct BwdIn {
    dbl2 zr, lr;
    dbl2 vy[KS], gr[KS];
    double vb[KS];
    dbl2 dw, dwl, da[KS];  // (refinement sweep only) the step in the record: (dz wz) (dl wl+) (dv adz)
  };
  template <bool REFINE = false>
  static FB_DEV void load_bwd(const double* R, const double* Rd, BwdIn& in) {
    in.zr = ld2(R, sZ);
    in.lr = ld2(R, sL);
    const double* const Rt = tail_of(R, Rd);
    sfor<0, KS>([&](auto S_) {
      constexpr int s = decltype(S_)::value;
      in.vy[s] = ld2(s == KS - 1 ? Rt : R, sV + 2 * s);
      if constexpr (kStoreGamma) in.gr[s] = ld2(s == KS - 1 ? Rt : R, sGAM + 2 * s);
      if constexpr (REFINE) in.da[s] = ld2(s == KS - 1 ? Rt : R, sDV + 2 * s);
    });
    ldv<sVB, KS>(R, in.vb);
    if constexpr (REFINE) {
      in.dw = ld2(R, sDZ);
      in.dwl = ld2(R, sDL);
    }
  }
  // (gamma, rv / mu) of constraint k at (v, y) for the subproblem centred at vbar
  // (riccati_linear_solver.cc:91-99)
  static FB_DEV dbl2 barrier_terms(double vk, double yk, double vb, double sigma, double alpha, bool real) {
    const double ys = yk + sigma * (vk - vb);
    double ph, g0, g1;
    pfb_all(ys, vk, alpha, &ph, &g0, &g1);
    const double imu = rcp_fast(g1 + sigma * g0);
    dbl2 o = {real ? g0 * imu : 0.0, real ? -ph * imu : 0.0};
    return o;
  }

  // What a Newton step hands back: the squared norms of the inner and the penalised natural residual at
  // x + dx (the first line-search trial) and `lin2`, the part of the first that belongs to the z and l
  // blocks - there the residual is affine in x, so that part IS the squared norm of the Newton system's
  // own residual r - V dx in those block rows (the third block row is solved exactly, dv = D^-1(rv + C A dz)):
  // what the linear solve left over.  It decides whether the step is refined (refine_step()).
  struct StepOut {
    double in2, out2, lin2;
    double zo2, bi2, bo2;  // z, l share of out2; sum b^2 of the inner and of the natural residual (trial_zl)
    int loff;  // matrix copy left resident in LDS
    bool ok;
  };
  FB_DEV bool newton_step(const C& c, double sigma, double alpha, double* trial_inner2,
                          double* trial_outer2, double* lin2 = nullptr, StepOut* sums = nullptr) {
    FB_WAVE_COUNT(27);  // Newton steps executed by the wavefront (any row active)
    FB_WAVE_TIMER(20);
    // (row-uniform; rows of a wavefront that disagree run one form after the other)
    const double tp = pend_t;  // pending step length
    pend_t = 0.0;
    StepOut o;
    if (rowdl) o = newton_core<true, false>(c, rec, pack, lpo, lds, pack_view(c), N, bounds, tp, lds_off, sigma, alpha);
    else o = newton_core<false, false>(c, rec, pack, lpo, lds, pack_view(c), N, bounds, tp, lds_off, sigma, alpha);
    lds_off = o.loff;
    *trial_inner2 = o.in2;
    *trial_outer2 = o.out2;
    if (lin2) *lin2 = o.lin2;
    if (sums) *sums = o;
    return o.ok;
  }
  // ---- one refinement of the step in the record ---------------------------------
  // An OPTION, off by default (fbstab_options_t::reserved).  Where it can still matter: the ONE-ROW instances
  // (<12,4,20>, <12,4,32>) in the ROW form of the costate step (bounds, sparse rows) multiply with an explicitly
  // inverted inv(Lc) where the reference substitutes (riccati_linear_solver.cc:234-325) - forward stable, not
  // backward stable: on stages whose Pi_i keeps eigenvalues of order sigma (nx > N nu) a step can leave a
  // residual r - V dx orders above eps |V| |dx|.  The REFERENCE form of those instances did the same until round
  // 6, and there it showed: a warm-started one-step solve of a (3, 12, 1, 17) QP left 1.4e-6 where the oracle
  // leaves 2e-8 and took a second proximal iteration (the option closed it: one step refined, the oracle's
  // counts); that form now substitutes (newton_core: FB_R16_SUBST_REF_FORM) and takes the oracle's counts as it
  // is.  The ROW-PAIR instances and the flat-vector kernel substitute since round 5 (kSubst, fb_row16.h
  // subst_rows): for all of those the option only ever makes a solve more accurate than the reference's.
  // refine_step() solves
  // V ddx = r - V dx with the SAME factors (the forward sweep is run again on the residual, which it forms
  // from the step in the record; the factorisation is recomputed - identical values - rather than kept:
  // W and, in the row form, inv(Pi) are not in the record) and adds ddx to the step; dv follows the third
  // block row exactly as before.  One such sweep leaves eps |V| |dx| in every block
  // (tools/riccati_forms_study.py).  The caller decides when: Solver::solve_stream asks for it when
  // sqrt(lin2) exceeds a fraction of the tolerance its next tests compare against - typical QPs never do
  // (BASELINE workload: 1e-12 against 1e-6) and run bitwise as before.
  // Real calls (no inlining): a second copy of the sweeps inside the solver loop would cost the common
  // path its register allocation.  They take scalars only, like the cooperative passes.
  static __device__ __attribute__((noinline)) StepOut refine_row_form(double* R0, const double* P0, lds_iptr po,
                                                                      lds_ptr lds_row, pk_ptr Lp, int N_, bool bnd, int loff,
                                                                      double sigma, double alpha) {
    C c;
    c.tid = threadIdx.x & (LPQ - 1);
    return newton_core<true, true>(c, R0, P0, po, lds_row, Lp, N_, bnd, 0.0, loff, sigma, alpha);
  }
  static __device__ __attribute__((noinline)) StepOut refine_ref_form(double* R0, const double* P0, lds_iptr po,
                                                                      lds_ptr lds_row, pk_ptr Lp, int N_, bool bnd, int loff,
                                                                      double sigma, double alpha) {
    C c;
    c.tid = threadIdx.x & (LPQ - 1);
    return newton_core<false, true>(c, R0, P0, po, lds_row, Lp, N_, bnd, 0.0, loff, sigma, alpha);
  }
  // No step may be pending (the Newton step that wrote the record has consumed it).
  FB_DEV bool refine_step(const C& c, double sigma, double alpha, double* trial_inner2, double* trial_outer2,
                          double* lin2, StepOut* sums = nullptr) {
    StepOut o;
    if (rowdl) o = refine_row_form(rec, pack, lpo, lds, pack_view(c), N, bounds, lds_off, sigma, alpha);
    else o = refine_ref_form(rec, pack, lpo, lds, pack_view(c), N, bounds, lds_off, sigma, alpha);
    lds_off = o.loff;
    if (!o.ok) return false;  // (cannot happen: the same factorisation has just succeeded; the step stands)
    *trial_inner2 = o.in2;
    *trial_outer2 = o.out2;
    *lin2 = o.lin2;
    if (sums) *sums = o;
    return true;
  }

  // ROW: the costate step from the Newton system's row, form (b) above.
  // REFINE: the right-hand side is the Newton system's residual at the step in the record, and the
  // backward sweep ADDS its solution to that step (refine_step()).
  template <bool ROW, bool REFINE>
  static FB_DEV StepOut newton_core(const C& c, double* const R0, const double* const P0, const lds_iptr po,
                                    lds_ptr lds_row, pk_ptr Lp, const int N_, const bool bnd, const double tp,
                                    int loff, double sigma, double alpha) {
    // Locals only below: the lambdas capture no object.
    // (round 6) The REFERENCE form of the costate step substitutes with the factor on the one-row instances as
    // well: that form serves the QPs whose constraint rows are dense or large - the reference's servo motor, random
    // dense rows - and it is where the explicit inverse showed (a warm-started one-step solve of a (3, 12, 1, 17)
    // QP, nx > N nu: 1.4e-6 left of the Newton system where the oracle leaves 2e-8, one proximal iteration more
    // than the oracle - LABNOTES R6.5).  The ROW form - bounds, sparse rows: the headline - keeps the explicit
    // inverse and its independent streams.  A per-template choice: the two forms are two copies of this function.
#ifndef FB_R16_SUBST_REF_FORM
#define FB_R16_SUBST_REF_FORM 1
#endif
    constexpr bool kSubst = MpcR16::kSubst || (FB_R16_SUBST_REF_FORM != 0 && !ROW);
    constexpr bool kAsmFwdX = MpcR16::kAsmImages && !kSubst;  // the forward stage's hand-written blocks move COLUMNS of the inverse
    const int r = c.tid;
    lds_ptr Tr = lds_row;
    lds_ptr Cl = lds_row;
    const bool rx = r < NX;
    StepOut ret;
    ret.in2 = ret.out2 = ret.lin2 = ret.zo2 = ret.bi2 = ret.bo2 = 0.0;
    ret.ok = false;

    double Pinv[NX];  // row r of inv(Pi_i); Pi_0 = sigma I (riccati_linear_solver.cc:127)
    sfor<0, NX>([&](auto Cc) { Pinv[decltype(Cc)::value] = (rx && r == decltype(Cc)::value) ? 1.0 / sigma : 0.0; });
    double thp = 0.0;
    bool ok = true;

    FB_STAMP_DECL;
    // The loads of stage i+1 are issued at the top of stage i (hand software
    // pipelining: with one wavefront per SIMD nothing else covers their latency).
    FwdIn cur;
    double wln = 0.0;  // WLN of the previous stage = wl of this one
    // offsets of the matrix copies of stages i and i+1 (fetched a stage ahead)
    int pcur = po[0], pnxt = po[N_ > 0 ? 1 : 0];
    [[maybe_unused]] bool dma_out = false;  // (kPackDma) this QP's next matrix copy is on its way into the image
    double* const Rd = spare_of(R0, N_);  // (kTailCut: the padding lanes' place for the last constraint slot)
    load_fwd(R0, Rd, cur);  // (loff: the copy resident in LDS)
    // ===================== forward sweep ===================================
    for (int i = 0; i <= N_; i++) {
      FB_PHASE(fwd_top);
      double* R = R0 + (long)i * kRec;
      const int pnn = po[i + 2 <= N_ ? i + 2 : N_];
      if constexpr (kPackDma) stage_pack_dma_wait(dma_out);
      stage_pack_s(c, P0, Lp, loff, pcur);
      // Lane id made opaque per iteration: (ro == j) selects are then recomputed
      // where used instead of being hoisted out of the loop as 16+ live masks.
      int ro = r;
      asm volatile("" : "+v"(ro));
      double Cc_[NC], K[NS];
      ldl<pC, NC>(Lp, Cc_);
      if constexpr (kKinLds) ldl<pK, NS>(Lp, K);
      else ldv<pK, NS>(P0 + loff, K);  // (the copy resident in LDS, in memory)
      // ---- pending step (tp = 0: no-op), PFB gradient (riccati_linear_solver.cc:91-99)
      double Gam[KS], Rvm[KS];
      sfor<0, KS>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
        const int k = r + LPQ * s;
        const double vk = fma(tp, cur.da[s][0], cur.vy[s][0]);
        const double yk = fma(-tp, cur.da[s][1], cur.vy[s][1]);
        const dbl2 bt = barrier_terms(vk, yk, cur.vb[s], sigma, alpha, LPQ * (s + 1) <= NC || k < NC);
        Gam[s] = bt[0];
        Rvm[s] = bt[1];
        if constexpr (!REFINE) {  // (the refinement sweep finds all of this in the record: tp = 0)
          double* const Rs = s == KS - 1 ? tail_of(R, Rd) : R;
          st2(Rs, sV + 2 * s, vk, yk);
          if constexpr (kStoreGamma) st2(Rs, sGAM + 2 * s, Gam[s], Rvm[s]);
        }
      });
      // pending step on (z, rz), (l, rl); eliminated right-hand side (:222-225)
      const double zz = fma(tp, cur.dw[0], cur.zr[0]);
      const double rzz = fma(tp, cur.dw[1], cur.zr[1]);
      const double ll = fma(tp, cur.dwl[0], cur.lr[0]);
      const double wlcur = wl_of_stage(i, rx, cur.dw[0], wln);  // wl(i) of the step in the record
      const double rll = fma(tp, wlcur, cur.lr[1]);
      wln = cur.dwl[1];
      double r1, r2;
      if constexpr (!REFINE) {
        st2(R, sZ, zz, rzz);
        st2(R, sL, ll, rll);
        r1 = -(rzz + sigma * zz);  // zero where there is no row (zz, ll: the displacements z - zbar, l - lbar)
        r2 = rll + sigma * ll;
      } else {
        // r - V dx in the z and l block rows = minus the inner residual's affine blocks at x + dx, formed
        // the way the backward sweep's trial norms form them (the v block row holds exactly: no rv term)
        r1 = -((rzz + cur.dw[1]) + sigma * (zz + cur.dw[0]));
        r2 = (rll + wlcur) + sigma * (ll + cur.dwl[0]);
      }
      FB_SB();
      // next stage's inputs, a whole stage ahead (the last stage fetches itself once
      // more: no branch here)
      load_fwd(i < N_ ? R + kRec : R, Rd, cur);
      FB_SB();
      FB_STAMP_LAP(0);
      // K row: H + sigma I (at pivot time) + inv(Pi) block + C' Gamma C (:101-123, :142-145)
      sfor<0, NX>([&](auto Cc) { K[decltype(Cc)::value] += Pinv[decltype(Cc)::value]; });
      if constexpr (!REFINE) {
        double p[4] = {r1, 0.0, 0.0, 0.0};
        bc_cols_dot<NC, RQ, true>(Cc_, Rvm, p);
        r1 = (p[0] + p[1]) + (p[2] + p[3]);
      }
      FB_PHASE(krhs_end);
      if (bnd) {
        FB_PHASE(k_bounds);
        // one nonzero per constraint row: only the diagonal entry K[r][r] changes.  (Round 5 measured the
        // term kept beside the row and joined to the pivot inside chol_rows - no select of the diagonal out
        // of and back into the row's registers, 80 instructions fewer: +0.8 % pipelined, -3.5 % one launch
        // at a time, and the spacecraft problem's counts part from the oracle's: K_rr + barrier - sum L^2
        // is not (K_rr - sum L^2) + barrier in the last bit.  Dropped; gpurun_out/r05_r.)
        double s = 0.0;
        sfor<0, NS>([&](auto Cc) { s = (ro == decltype(Cc)::value) ? K[decltype(Cc)::value] : s; });
        sfor<0, NC>([&](auto Kk) {
          constexpr int k = decltype(Kk)::value;
          const double gc = bcr<RQ, (k % LPQ)>(Gam[k / LPQ]) * Cc_[k];  // Gamma_k C[k][r]
          s = fma(gc, Cc_[k], s);
        });
        sfor<0, NS>([&](auto Cc) { K[decltype(Cc)::value] = (ro == decltype(Cc)::value) ? s : K[decltype(Cc)::value]; });
      } else {
        FB_PHASE(k_general);
        sfor<0, NC>([&](auto Kk) {
          constexpr int k = decltype(Kk)::value;
          const double gc = bcr<RQ, (k % LPQ)>(Gam[k / LPQ]) * Cc_[k];  // Gamma_k C[k][r]
          if constexpr (kFmacDpp<RQ>) {
            const Spread<RQ> cks = spread<RQ>(Cc_[k]);
            sfor<0, NS>([&](auto I) { fmac_bcs<RQ, decltype(I)::value, 0>(K[decltype(I)::value], cks, gc); });
          } else {
          const Spread<RQ> cks = spread<RQ>(Cc_[k]);
          bc_pipeline<NS>([&](auto I) { return bcs<RQ, decltype(I)::value>(cks); },
                          [&](auto I, double t) { K[decltype(I)::value] = fma(gc, t, K[decltype(I)::value]); });
          }
        });
      }
      FB_SB();
      FB_STAMP_LAP(1);
      // theta(i), h(i) = inv(Pi) theta - rx, g = [-h; ru] (:231-236, :252-261)
      const double th = thp + r2;
      double gv = r1;
      {
        const double hsum = bc_dot<0, NX, RQ>(Pinv, th);
        if (rx) gv = r1 - hsum;
      }
      // inv(Pi_i) joins the record as its lower triangle: rows to the linear image
      // in LDS, 16 consecutive elements per slot back
      const int tri_r = (ro * (ro + 1)) >> 1;
      if constexpr (!ROW) {
        c.sync();  // the previous stage has read its images
        if constexpr (kAsmImages) {
          img_write_pinv(Tr + kPl + tri_r, ro, Pinv);
        } else {
          sfor<0, NX>([&](auto Cc) {
            constexpr int cc = decltype(Cc)::value;
            Tr[(rx && cc <= ro) ? kPl + tri_r + cc : kDump] = Pinv[cc];
          });
        }
        c.sync();
        double Pp[nPs + 1];
        sfor<0, nPs>([&](auto S_) { Pp[decltype(S_)::value] = Tr[kPl + LPQ * decltype(S_)::value + r]; });
        if (LPQ * (nPs - 1) + r >= kPTri) Pp[nPs - 1] = 0.0;
        Pp[nPs] = th;
        stv<fP, nPs + 1>(R, Pp);
      }
      FB_SB();
      FB_STAMP_LAP(2);
      // ---- Lc = chol(K); columns of inv(Lc) and W = [A B] inv(Lc)' (AM and -P of
      // :149-175) from one pass over Lc
      double W[NS];
      double XC[NS];
      // (one-row instances, round 6: factorisation, inverse and W solve as ONE pass over the pivots -
      // fb_row16.h, chol_inv_cols_solve; bitwise the two-pass results)
      constexpr bool kFusedChol = FB_CHOL_FUSED != 0 && !kSubst && kFmacDpp<RQ> && FB_FMAC_DPP_SOLVE != 0;
      // (row-pair instances: factorisation and W solve as one pass likewise, the factor left as chol_rows leaves it)
      constexpr bool kFusedCholSubst = FB_CHOL_FUSED != 0 && kSubst && kFmacDpp<RQ> && !kPackDma;
      if constexpr (kFusedChol) {
        ldl<pABr, NS>(Lp, W);  // [A B] row r, the right-hand side of the W solve
        ok = chol_inv_cols_solve<NS, RQ>(K, XC, W, ro, sigma) && ok;
      } else if constexpr (kFusedCholSubst) {
        ldl<pABr, NS>(Lp, W);
        ok = chol_solve_right<NS, RQ>(K, W, ro, sigma) && ok;
      } else {
        ok = chol_rows<NS, RQ>(K, ro, sigma) && ok;
      }
      if (!ok) { ret.loff = loff; return ret; }
      FB_STAMP_LAP(3);
      FB_SB();
      if constexpr (!kFusedChol && !kFusedCholSubst) ldl<pABr, NS>(Lp, W);  // [A B] row r, the right-hand side of the W solve
      if constexpr (kPackDma) {
        // the image has been read for the last time in this stage: the next stage's copy on its way
        dma_out = i < N_ && pnxt != loff;
        if (dma_out) {
          c.sync();
          stage_pack_dma(P0 + pnxt, Lp);
          loff = pnxt;
        }
      }
      if constexpr (kSubst) {
        if constexpr (!kFusedCholSubst) tri_solve_right<NS, RQ>(K, W, ro);
      } else if constexpr (!kFusedChol) {
        tri_inv_cols_solve<NS, RQ>(K, XC, W, ro);
      }
      FB_STAMP_LAP(4);
      FB_SB();
      // columns of inv(Lc) to the linear image of its lower triangle; rows (for t) and
      // the record's slots come back from it
      double XR[NS];
      double Xp[nXs + 1];
      c.sync();
      if constexpr (kAsmFwdX) {
        img_write_x(Tr + kXl + ro, ro, XC);
        c.sync();
        sfor<0, NS>([&](auto J) { XR[decltype(J)::value] = 0.0; });
        img_read_xrow_slots(Tr + kXl + tri_r, Tr + kXl + ro, ro, XR, Xp);
      } else if constexpr (kSubst) {
        // rows of Lc itself (K[c] = L[r][c] for c < r, K[r] = 1 / L[r][r]) to the image of the triangle
        sfor<0, NS>([&](auto Cc) {
          constexpr int cc = decltype(Cc)::value;
          Tr[(cc <= ro && ro < NS) ? kXl + tri_r + cc : kDump] = K[cc];
        });
        c.sync();
        sfor<0, nXs>([&](auto S_) { Xp[decltype(S_)::value] = Tr[kXl + LPQ * decltype(S_)::value + r]; });
      } else {
      sfor<0, NS>([&](auto J) {
        constexpr int j = decltype(J)::value;
        Tr[j >= ro ? kXl + tri(j) + r : kDump] = XC[j];
      });
      c.sync();
      sfor<0, NS>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const double v = Tr[kXl + tri_r + j];
        XR[j] = j <= ro ? v : 0.0;
      });
      sfor<0, nXs>([&](auto S_) { Xp[decltype(S_)::value] = Tr[kXl + LPQ * decltype(S_)::value + r]; });
      }
      if (LPQ * (nXs - 1) + r >= kXTri) Xp[nXs - 1] = 0.0;
      FB_SB();
      // t = inv(Lc) g
      double tvec;
      if constexpr (kSubst) tvec = subst_rows<NS, RQ>(K, gv, ro);
      else tvec = bc_dot<0, NS, RQ>(XR, gv);
      Xp[nXs] = tvec;
      stv<fX, nXs + 1>(R, Xp);
      FB_STAMP_LAP(5);
      FB_SB();
      // theta(i+1) partial = -W t.  (Outside the branch below on purpose: with W used
      // only inside it, the optimiser sinks the W half of the fused solve into the
      // branch and keeps all 120 broadcasts alive for it - 240 registers.)
      thp = -bc_dot<0, NS, RQ>(W, tvec);
      if (i < N_) {
        FB_SB();
        FB_PHASE(wwt);
        // ---- Pi(i+1) = sigma I + W W' ; L = chol ; inv(Pi) = T'T, T = inv(L).
        // (Measured and dropped: the two symmetric products with the other lanes'
        // rows read as 16-byte LDS broadcasts instead of DPP moves - 96 + 42
        // ds_read_b128 replace 270 v_mov_b64_dpp, and the kernel is 11-16 % slower.)
        double Pn[NX];
        sfor<0, NX>([&](auto Cc) { Pn[decltype(Cc)::value] = 0.0; });
        sfor<0, NS>([&](auto Kk) {
          constexpr int k = decltype(Kk)::value;
          if constexpr (kFmacDpp<RQ>) {
            const Spread<RQ> wks = spread<RQ>(W[k]);
            sfor<0, NX>([&](auto I) { fmac_bcs<RQ, decltype(I)::value, 0>(Pn[decltype(I)::value], wks, W[k]); });
          } else {
          const Spread<RQ> wks = spread<RQ>(W[k]);
          bc_pipeline<NX>([&](auto I) { return bcs<RQ, decltype(I)::value>(wks); },
                          [&](auto I, double t) { Pn[decltype(I)::value] = fma(W[k], t, Pn[decltype(I)::value]); });
          }
        });
        // (rows r >= NX of Pn are zero as they come: those rows of [A B], hence of W, are zero in the
        // matrix copy - no select; the same holds for T and the rows of inv(Pi) below)
        FB_SB();
        FB_STAMP_LAP(7);
        double T[NX];
        if constexpr (FB_CHOL_FUSED != 0 && kFmacDpp<RQ>) {
          ok = chol_inv_cols<NX, RQ>(Pn, T, ro, sigma) && ok;
          if (!ok) { ret.loff = loff; return ret; }
          FB_SB();
          FB_PHASE(tinv12);
        } else {
          ok = chol_rows<NX, RQ>(Pn, ro, sigma) && ok;
          if (!ok) { ret.loff = loff; return ret; }
          FB_SB();
          FB_PHASE(tinv12);
          tri_inv_cols<NX, RQ>(Pn, T, ro);
        }
        FB_SB();
        FB_PHASE(ttt);
        // inv(Pi)[r][cc] = sum_k T[k][r] T[k][cc], T[k][cc] = lane cc's T[k] (zero for k < cc)
        sfor<0, NX>([&](auto Cc) { Pinv[decltype(Cc)::value] = 0.0; });
        sfor<0, NX>([&](auto Kk) {
          constexpr int k = decltype(Kk)::value;
          if constexpr (kFmacDpp<RQ>) {
            const Spread<RQ> tks = spread<RQ>(T[k]);
            sfor<0, k + 1>([&](auto I) { fmac_bcs<RQ, decltype(I)::value, 0>(Pinv[decltype(I)::value], tks, T[k]); });
          } else {
          const Spread<RQ> tks = spread<RQ>(T[k]);
          bc_pipeline<k + 1>([&](auto I) { return bcs<RQ, decltype(I)::value>(tks); },
                             [&](auto I, double t) { Pinv[decltype(I)::value] = fma(T[k], t, Pinv[decltype(I)::value]); });
          }
        });
        FB_STAMP_LAP(8);
      }
      pcur = pnxt;
      pnxt = pnn;
      FB_PHASE(fwd_end);
    }

    // ============ backward sweep (:267-341), fused with dv, A dz, W and the
    // residual norms of the first line-search trial ==========================
    // One wavefront per SIMD: nothing but this wavefront's own instructions covers
    // a load.  The factor record of stage i - 1 is therefore requested while stage i
    // runs, each part into the registers its stage-i counterpart has just left
    // (inv(Lc) after the two substitutions, inv(Pi) after dl, the [A B] columns after
    // u), and the stage's own iterate vectors come a stage ahead as one bundle.
    double lp = 0.0;    // dl(i+1), lanes < NX
    double dzn = 0.0;   // dx(i+1), lanes < NX
    [[maybe_unused]] double lpt = 0.0;  // (REFINE) dl(i+1) of the refined step; lp and dzn are then the correction's
    double s_in = 0.0, s_out = 0.0;
    // The z and l blocks on their own: their share of s_in and s_out (sum (a + b)^2) and sum b^2 of both
    // residuals - what the line search needs of them for every step length but the first (trial_zl()).
    // (Measured when the first of them was added as a single-precision sum for the refinement option: a
    // double, a float and a sum kept in LDS all put the headline 2 % under a build without it, and so did
    // a build in which the optimiser removed the sum again - not the sum, the register allocation of a
    // kernel at 491 of 512 registers moves with any change of its source; gpurun_out/r05_c..f.)
    double s_zi = 0.0, s_zo = 0.0, s_bi = 0.0, s_bo = 0.0;
    auto zl_acc = [&](double ri, double ro_, double bo, double dstep) {
      const double bi = fma(sigma, dstep, bo);  // increment of the inner residual along the step
      s_zi = fma(ri, ri, s_zi);
      s_zo = fma(ro_, ro_, s_zo);
      s_bi = fma(bi, bi, s_bi);
      s_bo = fma(bo, bo, s_bo);
    };
    dbl2 lrn = {0.0, 0.0};  // (el, rl) of block i+1, handed down by stage i+1
    double Ac[NX];
    double Xp[nXs + 1], Pp[nPs + 1];  // the packed factor record of the stage: triangle slots, then t / theta
    auto load_fac = [&](const double* R) {
      ldv<fX, nXs + 1>(R, Xp);
      if constexpr (!ROW) ldv<fP, nPs + 1>(R, Pp);
    };
    BwdIn bin;
    dma_out = false;
    pcur = po[N_];
    {
      const double* R = R0 + (long)N_ * kRec;
      load_fac(R);
      if constexpr (!kAbcFromLds) ldv<pABc, NX>(P0 + pcur, Ac);
      load_bwd<REFINE>(R, Rd, bin);
    }
    // (kAbcFromLds) where this lane finds column r of [A B] in the staged image: slot pABr + r, entry j at + 2 j
    // (lanes without a row or column - NS < LPQ: the <18,5,10> instance - have no such slot: they read lane 0's
    // and get the zero the column slots of the matrix copy hold for them)
    const int rcol = (NS < LPQ && r >= NS) ? 0 : r;
    [[maybe_unused]] const lds_ptr abcol = (kTrimAb || !kKinLds) ? Lp - 2 * r + pair_at(pABr / 2) + (rcol >> 1) * kAbPair + (rcol & 1)
                                                                 : Lp - 2 * r + ((pABr + rcol) >> 1) * kPackPair + ((pABr + rcol) & 1);
    for (int i = N_; i >= 0; i--) {
      FB_PHASE(bwd_top);
      double* R = R0 + (long)i * kRec;
      const double* Rp = i > 0 ? R - kRec : R;  // the stage fetched next (stage 0 once more at the end)
      if constexpr (kPackDma) stage_pack_dma_wait(dma_out);
      stage_pack_s(c, P0, Lp, loff, pcur);
      if constexpr (kAbcFromLds) {
        // the columns of this stage's [A B] out of the rows just staged (entries j < NX of lanes j: rows of A, B)
        sfor<0, NX>([&](auto J) {
          const double e = abcol[2 * decltype(J)::value];
          Ac[decltype(J)::value] = (NS < LPQ && r >= NS) ? 0.0 : e;
        });
      }
      pcur = po[i > 0 ? i - 1 : 0];
      int ro = r;
      asm volatile("" : "+v"(ro));
      BwdIn cu = bin;
      load_bwd<REFINE>(Rp, Rd, bin);
      if constexpr (!kStoreGamma) {
        sfor<0, KS>([&](auto S_) {
          constexpr int sl = decltype(S_)::value;
          cu.gr[sl] = barrier_terms(cu.vy[sl][0], cu.vy[sl][1], cu.vb[sl], sigma, alpha,
                                    LPQ * (sl + 1) <= NC || r + LPQ * sl < NC);
        });
      }
      double Cc_[NC], Hr[NS], AB[NS];
      ldl<pC, NC>(Lp, Cc_);
      // the stage's triangles to their linear images in LDS; the registers they
      // leave take the record of the stage below at once
      const int tri_r = (ro * (ro + 1)) >> 1;
      c.sync();
      sfor<0, nXs>([&](auto S_) { Tr[kXl + LPQ * decltype(S_)::value + r] = Xp[decltype(S_)::value]; });
      if constexpr (!ROW)
        sfor<0, nPs>([&](auto S_) { Tr[kPl + LPQ * decltype(S_)::value + r] = Pp[decltype(S_)::value]; });
      const double tcur = Xp[nXs];
      [[maybe_unused]] double thcur = 0.0;
      if constexpr (!ROW) thcur = Pp[nPs];
      FB_SB();
      load_fac(Rp);
      FB_SB();
      // u = [A B]' dl(i+1) (zero at the terminal stage: lp = 0)
#ifndef FB_R16_BWD_FUSED_BC
#define FB_R16_BWD_FUSED_BC 1  // 0: the round-5 form (a copy of the vector in every lane, plain dot products) for A/B runs
#endif
      constexpr bool kBwdFusedBc = FB_R16_BWD_FUSED_BC != 0;
      double u;
      if constexpr (kBwdFusedBc) {
        u = bc_dot<0, NX, RQ>(Ac, lp);  // (round 6: was bc_all + dot4 - the same sums, 12 moves fewer)
      } else {
        double lpb[NX];
        bc_all<NX, RQ>(lp, lpb);
        u = dot4<NX>(Ac, lpb);
      }
      if constexpr (!kAbcFromLds) ldv<pABc, NX>(P0 + pcur, Ac);
      c.sync();
      // column r and row r of inv(Lc), row r of inv(Pi)
      double XC[NS], XR[NS];
      if constexpr (!kAsmImages) {
      sfor<0, NS>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const double vc = Tr[kXl + tri(j) + r], vr = Tr[kXl + tri_r + j];
        XC[j] = j >= ro ? vc : 0.0;
        XR[j] = j <= ro ? vr : 0.0;
      });
      if constexpr (!ROW)
        sfor<0, NX>([&](auto Cc) {
          constexpr int cc = decltype(Cc)::value;
          const double v = Tr[kPl + (cc <= ro ? tri_r + cc : tri(cc) + r)];
          Pinv[cc] = rx ? v : 0.0;
        });
      }
      FB_SB();
      // s = t - W' dl(i+1) = t - inv(Lc) u ;  [dx; du] = inv(Lc)' s
      if constexpr (kAsmImages) {
        sfor<0, NS>([&](auto J) { XR[decltype(J)::value] = 0.0; });
        img_read_xrow(Tr + kXl + tri_r, ro, XR);
      }
      double s;  // (kSubst: XR, XC are row r and column r of Lc itself, the diagonal as its reciprocal)
      if constexpr (kSubst) s = tcur - subst_rows<NS, RQ>(XR, u, ro);
      else s = tcur - bc_dot<0, NS, RQ>(XR, u);
      if constexpr (kAsmImages) {
        sfor<0, NS>([&](auto J) { XC[decltype(J)::value] = 0.0; });
        img_read_xcol(Tr + kXl + ro, ro, XC);
      }
      double dzu;
      if constexpr (kSubst) {
        dzu = subst_cols_t<NS, RQ>(XC, s, ro);
        if (NS < LPQ && ro >= NS) dzu = 0.0;  // (lanes without a row: their image reads are not theirs)
      } else {
        dzu = bc_dot<0, NS, RQ>(XC, s);
      }
      // dl = -inv(Pi)(theta + dx), form (a); form (b) follows A'dv below
      double dli = 0.0;
      if constexpr (!ROW) {
        const double tx = thcur + dzu;
        if constexpr (kAsmImages) {
          sfor<0, NX>([&](auto Cc) { Pinv[decltype(Cc)::value] = 0.0; });
          img_read_pinv(Tr + kPl + tri_r, Tr + kPl + ro, ro, Pinv);
        }
        dli = -bc_dot<0, NX, RQ>(Pinv, tx);
        if (!rx) dli = 0.0;
      }
      FB_SB();
      // ([dx; du](i) reaches the four products below through the fused broadcast-FMA: round 6, was a copy of
      // it in every lane - bc_all, 16 moves - and plain dot products; bitwise the same sums)
      [[maybe_unused]] double dzb[NS];
      if constexpr (!kBwdFusedBc) bc_all<NS, RQ>(dzu, dzb);
      if constexpr (kKinLds) ldl<pK, NS>(Lp, Hr);
      else ldv<pK, NS>(P0 + loff, Hr);
      ldl<pABr, NS>(Lp, AB);
      if constexpr (kPackDma) {
        // (pcur is the stage below's already) its copy on its way while this stage finishes
        dma_out = i > 0 && pcur != loff;
        if (dma_out) {
          c.sync();
          stage_pack_dma(P0 + pcur, Lp);
          loff = pcur;
        }
      }
      FB_STAMP_LAP(9);
      // ---- A dz and dv (:329-341) through the LDS copy of C
      C_to_lds(c, Cl, Cc_, r);
      double dvs[KS];
      auto c_rows = [&](auto&& f) {
        if constexpr (kBwdFusedBc) rows_of_C_times_lane(Cl, dzu, r, f);
        else rows_of_C_times(Cl, dzb, r, f);
      };
      c_rows([&](auto S_, bool valid, double a) {
        constexpr int sl = decltype(S_)::value;
        double d = 0.0;
        double dt = 0.0, at = valid ? a : 0.0;  // what the record gets: the step, or (REFINE) the step + correction
        if constexpr (REFINE) {
          dt = cu.da[sl][0];
          at = cu.da[sl][1] + at;
        }
        if (valid) {
          if constexpr (REFINE) {
            d = cu.gr[sl][0] * a;  // (no rv / mu: the third block row has no residual)
            dt += d;
          } else {
            d = cu.gr[sl][1] + cu.gr[sl][0] * a;
            dt = d;
          }
          // first line-search trial, v block (full_residual.cc:68-71, :99-106)
          const double vi = cu.vy[sl][0] + dt;
          const double yi = cu.vy[sl][1] - at;
          const double ys = yi + sigma * (vi - cu.vb[sl]);
          const double ph = pfb(ys, vi, alpha);
          const double pn = pnr(yi, vi, alpha);
          s_in = fma(ph, ph, s_in);
          s_out = fma(pn, pn, s_out);
        }
        st2(sl == KS - 1 ? tail_of(R, Rd) : R, sDV + 2 * sl, dt, at);
        dvs[sl] = d;
      });
      // ---- wz = H dz + G'dl + A'dv; (G'dl)_x = [A B]'dl(i+1) - dl(i)
      double w;
      {
        double hdz;
        if constexpr (kBwdFusedBc) hdz = bc_dot<0, NS, RQ>(Hr, dzu);
        else hdz = dot4<NS>(Hr, dzb);
        double p[4] = {hdz + (ROW ? u : u - dli), 0.0, 0.0, 0.0};
        bc_cols_dot<NC, RQ>(Cc_, dvs, p);
        w = (p[0] + p[1]) + (p[2] + p[3]);
        if constexpr (ROW) {
          // form (b): the state rows of (H + sigma I) dz + G'dl + A'dv = -(rz + sigma (z - zbar))
          // inner residual, z block - (REFINE) at x + dx: what is left of the row
          double rin = cu.zr[1] + sigma * cu.zr[0];
          if constexpr (REFINE) rin = (cu.zr[1] + cu.dw[1]) + sigma * (cu.zr[0] + cu.dw[0]);
          dli = rx ? (w + sigma * dzu) + rin : 0.0;
          w -= dli;
        }
      }
      // (REFINE) from here on the STEP: what the record held plus this sweep's correction
      double dzt = dzu, wt = w, dlt = dli;
      if constexpr (REFINE) {
        dzt += cu.dw[0];
        wt += cu.dw[1];
        dlt += cu.dwl[0];
      }
      double wlv = 0.0;  // wl(i + 1)
      if (i < N_) {
        // l block i+1: wl = -(A dx + B du - dx(i+1)); trial norms (full_residual.cc:60-66)
        double abz;
        if constexpr (kBwdFusedBc) abz = bc_dot<0, NS, RQ>(AB, dzu);
        else abz = dot4<NS>(AB, dzb);
        wlv = rx ? -(abz - dzn) : 0.0;
        if constexpr (REFINE) wlv += cu.dwl[1];  // (dzn is the correction's: wl(i + 1) of the step + its increment)
        const double lr = lrn[1] + wlv;
        const double li = lrn[0] + (REFINE ? lpt : lp);
        const double ri = lr + sigma * li;
        s_in = fma(ri, ri, s_in);
        s_out = fma(lr, lr, s_out);
        zl_acc(ri, lr, wlv, REFINE ? lpt : lp);
      }
      {
        st2(R, sDZ, dzt, wt);
        const double zrr = cu.zr[1] + wt;
        const double zi = cu.zr[0] + dzt;
        const double ri = zrr + sigma * zi;
        s_in = fma(ri, ri, s_in);
        s_out = fma(zrr, zrr, s_out);
        zl_acc(ri, zrr, wt, dzt);
      }
      st2(R, sDL, dlt, wlv);
      if (i == 0) {
        // l block 0: -(G dz)_0 = dx(0)
        const double wl0 = rx ? dzt : 0.0;
        const double lr = cu.lr[1] + wl0;
        const double li = cu.lr[0] + dlt;
        const double ri = lr + sigma * li;
        s_in = fma(ri, ri, s_in);
        s_out = fma(lr, lr, s_out);
        zl_acc(ri, lr, wl0, dlt);
      }
      lp = dli;
      if constexpr (REFINE) lpt = dlt;
      dzn = rx ? dzu : 0.0;
      lrn = cu.lr;
      FB_STAMP_LAP(10);
      FB_PHASE(bwd_end);
    }
    ret.loff = loff;
    ret.in2 = qp_reduce<RQ, OpSum16>(s_in);
    ret.out2 = qp_reduce<RQ, OpSum16>(s_out);
    ret.lin2 = qp_reduce<RQ, OpSum16>(s_zi);
    ret.zo2 = qp_reduce<RQ, OpSum16>(s_zo);
    ret.bi2 = qp_reduce<RQ, OpSum16>(s_bi);
    ret.bo2 = qp_reduce<RQ, OpSum16>(s_bo);
    ret.ok = true;
    return ret;
  }
};


}  // namespace fbk
