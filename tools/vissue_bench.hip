// What does a wave's MFMA issue cost in phase V of the lean kernel (rollout_fwd_lean.hip) when NOTHING is loaded?  Round 6: with its loads compiled
// out (RLX_NOLOAD) the real kernel still spends 10.5 k cycles per step in phase V on its heavy waves -- 19 register buffers x 12
// v_mfma_f64_4x4x4_4b_f64 = 45 cycles per MFMA and wave, where tools/mfma4x4_probe.hip measures 16.3 for independent accumulators.  This
// model keeps the kernel's per-buffer shape (6 operand tiles of 2 doubles per lane, 3 k operands of 2 doubles from LDS, 4 or 6 accumulators by
// the wave's row-tile count) and varies HOW the 12 MFMAs of a buffer are written:
//   form 0  the kernel's: builtin MFMAs, the wiring (2 / 3 row tiles) a wave-uniform branch inside every buffer
//   form 1  wiring a compile-time property of the loop a wave runs, builtin MFMAs
//   form 2  the same with in-place inline-asm MFMAs (no accumulator renaming, no copies where branches meet)
//   form 3  form 2 with the k operands of the whole stream read BEFORE the loop (no LDS instruction between the MFMAs)
//   form 4  form 2, two buffers (24 MFMAs) per loop body half
// 8 waves per workgroup: waves 2, 5, 7 carry 19 buffers (3 row tiles), the others 13 (2 row tiles), as at N = 300.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/vissue_bench tools/vissue_bench.hip && tools/bin/vissue_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v2d __attribute__((ext_vector_type(2)));
#define NL 6
#define P 4

__device__ __forceinline__ void readk(v2d (&K)[3], const double* ka, int g) {
  const double* kp = ka + g * (8 * P);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    K[q].x = kp[q * 8 * P];
    K[q].y = kp[q * 8 * P + 4 * P];
  }
}
__device__ __forceinline__ void mfma_b(double& acc, double a, double b) { acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0); }
__device__ __forceinline__ void mfma_a(double& acc, double a, double b) { asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); }

// form 0: the kernel's kt_use
__device__ __forceinline__ void use0(const v2d (&A)[NL], const v2d (&K)[3], int nrt, double (&acc3)[2][3], double (&acc2)[2][2]) {
  asm volatile("" ::"v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(A[4]), "v"(A[5]), "v"(K[0]), "v"(K[1]), "v"(K[2]));
  if (nrt == 3) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int r = 0; r < 3; ++r) mfma_b(acc3[0][r], A[3 * q + r].x, K[q].x);
#pragma unroll
      for (int r = 0; r < 3; ++r) mfma_b(acc3[1][r], A[3 * q + r].y, K[q].y);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
      for (int r = 0; r < 2; ++r) mfma_b(acc2[0][r], A[2 * q + r].x, K[q].x);
#pragma unroll
      for (int r = 0; r < 2; ++r) mfma_b(acc2[1][r], A[2 * q + r].y, K[q].y);
    }
  }
}
template <int NRT, bool ASM>
__device__ __forceinline__ void usef(const v2d (&A)[NL], const v2d (&K)[3], double (&acc)[2][NRT]) {
  asm volatile("" ::"v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(A[4]), "v"(A[5]), "v"(K[0]), "v"(K[1]), "v"(K[2]));
  constexpr int NQ = NL / NRT;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
#pragma unroll
    for (int r = 0; r < NRT; ++r) {
      if (ASM) mfma_a(acc[0][r], A[NRT * q + r].x, K[q].x); else mfma_b(acc[0][r], A[NRT * q + r].x, K[q].x);
    }
#pragma unroll
    for (int r = 0; r < NRT; ++r) {
      if (ASM) mfma_a(acc[1][r], A[NRT * q + r].y, K[q].y); else mfma_b(acc[1][r], A[NRT * q + r].y, K[q].y);
    }
  }
}


// ---- forms 7 / 8: WITH the Kinv stream (6 global_load_dwordx4 per buffer from an L2-resident matrix, double buffered) ------------------------
// form 8: as the kernel writes it (compiler-scheduled: builtin MFMAs, wiring branch per buffer, kt_load / kt_readk in C++)
// form 7: the whole streaming loop of a wave as ONE asm block -- buffers and k operands in fixed registers (v[184:255]), the waits written by
//         hand (vmcnt(6) lgkmcnt(3): the other buffer's loads and operand reads stay in flight), nothing between the MFMAs of a buffer
typedef const v2d __attribute__((address_space(1))) * gptr2_t;
__device__ __forceinline__ void ld6(v2d (&A)[NL], gptr2_t p, int b) {
  const gptr2_t pb = p + (size_t)b * (NL * 64);
#pragma unroll
  for (int s = 0; s < NL; ++s) A[s] = pb[s * 64];
}
#define VB_A "184"
// register map of the asm block: buffer A v[184:207], buffer B v[208:231], kA v[232:243], kB v[244:255]
#define VB_CLOB "v184","v185","v186","v187","v188","v189","v190","v191","v192","v193","v194","v195","v196","v197","v198","v199","v200","v201","v202","v203","v204","v205","v206","v207", \
                "v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223","v224","v225","v226","v227","v228","v229","v230","v231", \
                "v232","v233","v234","v235","v236","v237","v238","v239","v240","v241","v242","v243","v244","v245","v246","v247","v248","v249","v250","v251","v252","v253","v254","v255"
// six tiles of a buffer through a pointer centred on its fourth tile: offsets -3072 .. 2048
#define VB_LOAD(R0, PTR)                                                  \
  "global_load_dwordx4 v[" #R0 "+0:" #R0 "+3], " PTR ", off offset:-3072\n"   \
  "global_load_dwordx4 v[" #R0 "+4:" #R0 "+7], " PTR ", off offset:-2048\n"   \
  "global_load_dwordx4 v[" #R0 "+8:" #R0 "+11], " PTR ", off offset:-1024\n"  \
  "global_load_dwordx4 v[" #R0 "+12:" #R0 "+15], " PTR ", off\n"              \
  "global_load_dwordx4 v[" #R0 "+16:" #R0 "+19], " PTR ", off offset:1024\n"  \
  "global_load_dwordx4 v[" #R0 "+20:" #R0 "+23], " PTR ", off offset:2048\n"
// MFMA of tile T (0..5) half H (0: x, 2: y) of the buffer at R0 against operand register pair K
#define VB_MF(ACC, R0, T, H, K) "v_mfma_f64_4x4x4_4b_f64 " ACC ", v[" #R0 "+" #T "*4+" #H ":" #R0 "+" #T "*4+" #H "+1], v[" #K ":" #K "+1], " ACC "\n"

// the streamed part of a wave's phase V: n (even, >= 2) buffers from `pa` (this lane's pointer to the FOURTH tile of the first one), k operands from
// the LDS byte address `ka` (group 0 of the first buffer; a group = 256 bytes at P = 4), accumulated into acc
#define VB_S(x) #x
#define VB_X(x) VB_S(x)
#define RA 184
#define RB 208
#define KA_ 232
#define KB_ 244
#define VB_RD(K0, ADDR, G)                                                                                   \
  "ds_read2_b64 v[" VB_X(K0) "+0:" VB_X(K0) "+3], " ADDR " offset0:" VB_X(G) "*32+0 offset1:" VB_X(G) "*32+16\n"     \
  "ds_read2_b64 v[" VB_X(K0) "+4:" VB_X(K0) "+7], " ADDR " offset0:" VB_X(G) "*32+32 offset1:" VB_X(G) "*32+48\n"
#define VB_RD3(K0, ADDR, G) VB_RD(K0, ADDR, G) "ds_read2_b64 v[" VB_X(K0) "+8:" VB_X(K0) "+11], " ADDR " offset0:" VB_X(G) "*32+64 offset1:" VB_X(G) "*32+80\n"
#define VB_M(ACC, R0, T, H, K0, Q) "v_mfma_f64_4x4x4_4b_f64 " ACC ", v[" VB_X(R0) "+" #T "*4+" #H ":" VB_X(R0) "+" #T "*4+" #H "+1], v[" VB_X(K0) "+" #Q "*4+" #H ":" VB_X(K0) "+" #Q "*4+" #H "+1], " ACC "\n"
// two row tiles: column group q holds tiles 2q, 2q+1; accumulators %0 %1 (x halves of row tile 0, 1), %2 %3 (y halves)
#define VB_USE2(R0, K0)                                                                                                  \
  VB_M("%0", R0, 0, 0, K0, 0) VB_M("%1", R0, 1, 0, K0, 0) VB_M("%2", R0, 0, 2, K0, 0) VB_M("%3", R0, 1, 2, K0, 0) "s_nop 0\n" \
  VB_M("%0", R0, 2, 0, K0, 1) VB_M("%1", R0, 3, 0, K0, 1) VB_M("%2", R0, 2, 2, K0, 1) VB_M("%3", R0, 3, 2, K0, 1) "s_nop 0\n" \
  VB_M("%0", R0, 4, 0, K0, 2) VB_M("%1", R0, 5, 0, K0, 2) VB_M("%2", R0, 4, 2, K0, 2) VB_M("%3", R0, 5, 2, K0, 2)
// three row tiles: column group q holds tiles 3q .. 3q+2; accumulators %0 %1 %2 (x halves), %3 %4 %5 (y halves)
#define VB_USE3(R0, K0)                                                                                                  \
  VB_M("%0", R0, 0, 0, K0, 0) VB_M("%1", R0, 1, 0, K0, 0) VB_M("%2", R0, 2, 0, K0, 0) VB_M("%3", R0, 0, 2, K0, 0) VB_M("%4", R0, 1, 2, K0, 0) VB_M("%5", R0, 2, 2, K0, 0) \
  VB_M("%0", R0, 3, 0, K0, 1) VB_M("%1", R0, 4, 0, K0, 1) VB_M("%2", R0, 5, 0, K0, 1) VB_M("%3", R0, 3, 2, K0, 1) VB_M("%4", R0, 4, 2, K0, 1) VB_M("%5", R0, 5, 2, K0, 1)
__device__ __forceinline__ void stream_asm2(double (&acc)[2][2], gptr2_t pa, unsigned ka, int n) {
  unsigned long long inc = 12288;  // two buffers
  gptr2_t pb = pa + 6 * 64;
  int it = n / 2 - 1;
  asm volatile(VB_LOAD(184, "%4") VB_LOAD(208, "%5") VB_RD3(KA_, "%6", 0) VB_RD3(KB_, "%6", 3)
               "s_cmp_lt_i32 %7, 1\n"
               "s_cbranch_scc1 2f\n"
               "1:\n"
               "s_waitcnt vmcnt(6) lgkmcnt(3)\n" VB_USE2(RA, KA_)
               "v_lshl_add_u64 %4, %4, 0, %8\n"
               "v_add_u32 %6, 0x600, %6\n" VB_LOAD(184, "%4") VB_RD3(KA_, "%6", 0)
               "s_waitcnt vmcnt(6) lgkmcnt(3)\n" VB_USE2(RB, KB_)
               "v_lshl_add_u64 %5, %5, 0, %8\n" VB_LOAD(208, "%5") VB_RD3(KB_, "%6", 3)
               "s_sub_i32 %7, %7, 1\n"
               "s_cmp_lt_i32 %7, 1\n"
               "s_cbranch_scc0 1b\n"
               "2:\n"
               "s_waitcnt vmcnt(6) lgkmcnt(3)\n" VB_USE2(RA, KA_)
               "s_waitcnt vmcnt(0) lgkmcnt(0)\n" VB_USE2(RB, KB_)
               "s_nop 15\n"
               : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(pa), "+v"(pb), "+v"(ka), "+s"(it)
               : "s"(inc)
               : VB_CLOB, "memory", "scc");
}
__device__ __forceinline__ void stream_asm3(double (&acc)[2][3], gptr2_t pa, unsigned ka, int n) {
  unsigned long long inc = 12288;
  gptr2_t pb = pa + 6 * 64;
  int it = n / 2 - 1;
  asm volatile(VB_LOAD(184, "%6") VB_LOAD(208, "%7") VB_RD(KA_, "%8", 0) VB_RD(KB_, "%8", 2)
               "s_cmp_lt_i32 %9, 1\n"
               "s_cbranch_scc1 2f\n"
               "1:\n"
               "s_waitcnt vmcnt(6) lgkmcnt(2)\n" VB_USE3(RA, KA_)
               "v_lshl_add_u64 %6, %6, 0, %10\n"
               "v_add_u32 %8, 0x400, %8\n" VB_LOAD(184, "%6") VB_RD(KA_, "%8", 0)
               "s_waitcnt vmcnt(6) lgkmcnt(2)\n" VB_USE3(RB, KB_)
               "v_lshl_add_u64 %7, %7, 0, %10\n" VB_LOAD(208, "%7") VB_RD(KB_, "%8", 2)
               "s_sub_i32 %9, %9, 1\n"
               "s_cmp_lt_i32 %9, 1\n"
               "s_cbranch_scc0 1b\n"
               "2:\n"
               "s_waitcnt vmcnt(6) lgkmcnt(2)\n" VB_USE3(RA, KA_)
               "s_waitcnt vmcnt(0) lgkmcnt(0)\n" VB_USE3(RB, KB_)
               "s_nop 15\n"
               : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(pa), "+v"(pb), "+v"(ka), "+s"(it)
               : "s"(inc)
               : VB_CLOB, "memory", "scc");
}

// form 9: the reload of a pair of tiles follows the four MFMAs that read it (not the whole buffer's twelve): every wait is vmcnt(10) lgkmcnt(5)
#define VB_LD1(R0, T, PTR, OFF) "global_load_dwordx4 v[" VB_X(R0) "+" #T "*4:" VB_X(R0) "+" #T "*4+3], " PTR ", off offset:" #OFF "\n"
#define VB_RD1(K0, Q, ADDR, G) "ds_read2_b64 v[" VB_X(K0) "+" #Q "*4:" VB_X(K0) "+" #Q "*4+3], " ADDR " offset0:(" VB_X(G) "+" #Q ")*32 offset1:(" VB_X(G) "+" #Q ")*32+16\n"
#define VB_G2(R0, K0, Q, T0, T1) VB_M("%0", R0, T0, 0, K0, Q) VB_M("%1", R0, T1, 0, K0, Q) VB_M("%2", R0, T0, 2, K0, Q) VB_M("%3", R0, T1, 2, K0, Q)
__device__ __forceinline__ void stream_asm2f(double (&acc)[2][2], gptr2_t pa, unsigned ka, int n) {
  unsigned long long inc = 12288;
  gptr2_t pb = pa + 6 * 64;
  int it = n / 2 - 1;
  asm volatile(VB_LOAD(184, "%4") VB_LOAD(208, "%5") VB_RD3(KA_, "%6", 0) VB_RD3(KB_, "%6", 3)
               "s_cmp_lt_i32 %7, 1\n"
               "s_cbranch_scc1 2f\n"
               "1:\n"
               "v_lshl_add_u64 %4, %4, 0, %8\n"
               "v_add_u32 %6, 0x600, %6\n"
               "s_waitcnt vmcnt(10) lgkmcnt(5)\n" VB_G2(RA, KA_, 0, 0, 1) VB_LD1(RA, 0, "%4", -3072) VB_LD1(RA, 1, "%4", -2048) VB_RD1(KA_, 0, "%6", 0)
               "s_waitcnt vmcnt(10) lgkmcnt(5)\n" VB_G2(RA, KA_, 1, 2, 3) VB_LD1(RA, 2, "%4", -1024) VB_LD1(RA, 3, "%4", 0) VB_RD1(KA_, 1, "%6", 0)
               "s_waitcnt vmcnt(10) lgkmcnt(5)\n" VB_G2(RA, KA_, 2, 4, 5) VB_LD1(RA, 4, "%4", 1024) VB_LD1(RA, 5, "%4", 2048) VB_RD1(KA_, 2, "%6", 0)
               "v_lshl_add_u64 %5, %5, 0, %8\n"
               "s_waitcnt vmcnt(10) lgkmcnt(5)\n" VB_G2(RB, KB_, 0, 0, 1) VB_LD1(RB, 0, "%5", -3072) VB_LD1(RB, 1, "%5", -2048) VB_RD1(KB_, 0, "%6", 3)
               "s_waitcnt vmcnt(10) lgkmcnt(5)\n" VB_G2(RB, KB_, 1, 2, 3) VB_LD1(RB, 2, "%5", -1024) VB_LD1(RB, 3, "%5", 0) VB_RD1(KB_, 1, "%6", 3)
               "s_waitcnt vmcnt(10) lgkmcnt(5)\n" VB_G2(RB, KB_, 2, 4, 5) VB_LD1(RB, 4, "%5", 1024) VB_LD1(RB, 5, "%5", 2048) VB_RD1(KB_, 2, "%6", 3)
               "s_sub_i32 %7, %7, 1\n"
               "s_cmp_lt_i32 %7, 1\n"
               "s_cbranch_scc0 1b\n"
               "2:\n"
               "s_waitcnt vmcnt(6) lgkmcnt(3)\n" VB_USE2(RA, KA_)
               "s_waitcnt vmcnt(0) lgkmcnt(0)\n" VB_USE2(RB, KB_)
               "s_nop 15\n"
               : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(pa), "+v"(pb), "+v"(ka), "+s"(it)
               : "s"(inc)
               : VB_CLOB, "memory", "scc");
}
#define VB_G3(R0, K0, Q, T0, T1, T2) VB_M("%0", R0, T0, 0, K0, Q) VB_M("%1", R0, T1, 0, K0, Q) VB_M("%2", R0, T2, 0, K0, Q) VB_M("%3", R0, T0, 2, K0, Q) VB_M("%4", R0, T1, 2, K0, Q) VB_M("%5", R0, T2, 2, K0, Q)
__device__ __forceinline__ void stream_asm3f(double (&acc)[2][3], gptr2_t pa, unsigned ka, int n) {
  unsigned long long inc = 12288;
  gptr2_t pb = pa + 6 * 64;
  int it = n / 2 - 1;
  asm volatile(VB_LOAD(184, "%6") VB_LOAD(208, "%7") VB_RD(KA_, "%8", 0) VB_RD(KB_, "%8", 2)
               "s_cmp_lt_i32 %9, 1\n"
               "s_cbranch_scc1 2f\n"
               "1:\n"
               "v_lshl_add_u64 %6, %6, 0, %10\n"
               "v_add_u32 %8, 0x400, %8\n"
               "s_waitcnt vmcnt(9) lgkmcnt(3)\n" VB_G3(RA, KA_, 0, 0, 1, 2) VB_LD1(RA, 0, "%6", -3072) VB_LD1(RA, 1, "%6", -2048) VB_LD1(RA, 2, "%6", -1024) VB_RD1(KA_, 0, "%8", 0)
               "s_waitcnt vmcnt(9) lgkmcnt(3)\n" VB_G3(RA, KA_, 1, 3, 4, 5) VB_LD1(RA, 3, "%6", 0) VB_LD1(RA, 4, "%6", 1024) VB_LD1(RA, 5, "%6", 2048) VB_RD1(KA_, 1, "%8", 0)
               "v_lshl_add_u64 %7, %7, 0, %10\n"
               "s_waitcnt vmcnt(9) lgkmcnt(3)\n" VB_G3(RB, KB_, 0, 0, 1, 2) VB_LD1(RB, 0, "%7", -3072) VB_LD1(RB, 1, "%7", -2048) VB_LD1(RB, 2, "%7", -1024) VB_RD1(KB_, 0, "%8", 2)
               "s_waitcnt vmcnt(9) lgkmcnt(3)\n" VB_G3(RB, KB_, 1, 3, 4, 5) VB_LD1(RB, 3, "%7", 0) VB_LD1(RB, 4, "%7", 1024) VB_LD1(RB, 5, "%7", 2048) VB_RD1(KB_, 1, "%8", 2)
               "s_sub_i32 %9, %9, 1\n"
               "s_cmp_lt_i32 %9, 1\n"
               "s_cbranch_scc0 1b\n"
               "2:\n"
               "s_waitcnt vmcnt(6) lgkmcnt(2)\n" VB_USE3(RA, KA_)
               "s_waitcnt vmcnt(0) lgkmcnt(0)\n" VB_USE3(RB, KB_)
               "s_nop 15\n"
               : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(pa), "+v"(pb), "+v"(ka), "+s"(it)
               : "s"(inc)
               : VB_CLOB, "memory", "scc");
}

template <int NRT, int FORM>
__device__ __forceinline__ double stream_fixed(const v2d (&bufA)[NL], const v2d (&bufB)[NL], const double* ka, int nb) {
  constexpr int gpb = NL / NRT;
  double acc[2][NRT];
#pragma unroll
  for (int r = 0; r < NRT; ++r) acc[0][r] = acc[1][r] = 0.0;
  constexpr bool ASM = FORM >= 2;
  if (FORM == 3) {  // (numerically meaningless: every buffer meets the same three operands -- what the MFMAs cost with NO LDS instruction between them)
    v2d kq[3];
    readk(kq, ka, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kq[0]), "+v"(kq[1]), "+v"(kq[2]));
    for (int b = 0; b + 1 < nb; b += 2) {
      usef<NRT, true>(bufA, kq, acc);
      usef<NRT, true>(bufB, kq, acc);
    }
    if (nb & 1) usef<NRT, true>(bufA, kq, acc);
  } else if (FORM == 5) {  // the six doubles of a buffer's operands contiguous per lane: ds_read_b128 + ds_read_b64 (a packed copy of k: [buffer][lane slot 16][6])
    const double* kp6 = ka;  // (the model reads a lane-private run of the same table)
    auto readk6 = [&](v2d (&K)[3], int b) {
      const double* q = kp6 + ((b * 16) % 300);
      const v2d t0 = *reinterpret_cast<const v2d*>(__builtin_assume_aligned(q, 16));
      const v2d t1 = *reinterpret_cast<const v2d*>(__builtin_assume_aligned(q + 2, 16));
      const v2d t2 = *reinterpret_cast<const v2d*>(__builtin_assume_aligned(q + 4, 16));
      K[0] = t0;
      K[1] = t1;
      K[2] = t2;
    };
    v2d kA[3], kB[3];
    readk6(kA, 0);
    readk6(kB, 1);
    int b = 0;
    for (; b + 2 < nb; b += 2) {
      usef<NRT, true>(bufA, kA, acc);
      readk6(kA, b + 2);
      usef<NRT, true>(bufB, kB, acc);
      readk6(kB, b + 3);
    }
    usef<NRT, true>(bufA, kA, acc);
    if (b + 1 < nb) usef<NRT, true>(bufB, kB, acc);
  } else if (FORM == 6) {  // the operand reads of TWO buffers issued together, once per 24 MFMAs
    v2d kA[3], kB[3], kC[3], kD[3];
    readk(kA, ka, 0);
    readk(kB, ka, gpb);
    int b = 0;
    for (; b + 3 < nb; b += 4) {
      readk(kC, ka, (b + 2) * gpb);
      readk(kD, ka, (b + 3) * gpb);
      __builtin_amdgcn_sched_barrier(0);
      usef<NRT, true>(bufA, kA, acc);
      usef<NRT, true>(bufB, kB, acc);
      __builtin_amdgcn_sched_barrier(0);
      readk(kA, ka, (b + 4) * gpb);
      readk(kB, ka, (b + 5) * gpb);
      __builtin_amdgcn_sched_barrier(0);
      usef<NRT, true>(bufA, kC, acc);
      usef<NRT, true>(bufB, kD, acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (; b < nb; ++b) {
      readk(kA, ka, b * gpb);
      usef<NRT, true>(bufA, kA, acc);
    }
  } else if (FORM == 4) {
    v2d kA[3], kB[3], kC[3], kD[3];
    readk(kA, ka, 0);
    readk(kB, ka, gpb);
    int b = 0;
    for (; b + 3 < nb; b += 4) {
      readk(kC, ka, (b + 2) * gpb);
      readk(kD, ka, (b + 3) * gpb);
      usef<NRT, true>(bufA, kA, acc);
      usef<NRT, true>(bufB, kB, acc);
      readk(kA, ka, (b + 4) * gpb);
      readk(kB, ka, (b + 5) * gpb);
      usef<NRT, true>(bufA, kC, acc);
      usef<NRT, true>(bufB, kD, acc);
    }
    for (; b < nb; ++b) {
      readk(kA, ka, b * gpb);
      usef<NRT, true>(bufA, kA, acc);
    }
  } else {
    v2d kA[3], kB[3];
    readk(kA, ka, 0);
    readk(kB, ka, gpb);
    int b = 0;
    for (; b + 2 < nb; b += 2) {
      usef<NRT, ASM>(bufA, kA, acc);
      readk(kA, ka, (b + 2) * gpb);
      usef<NRT, ASM>(bufB, kB, acc);
      readk(kB, ka, (b + 3) * gpb);
    }
    usef<NRT, ASM>(bufA, kA, acc);
    if (b + 1 < nb) usef<NRT, ASM>(bufB, kB, acc);
  }
  if (ASM) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < NRT; ++r) s += acc[0][r] + acc[1][r];
  return s;
}

template <int FORM>
__global__ __launch_bounds__(512) void vbench(int nstep, double* out, unsigned long long* cyc, const double* tiles) {
  __shared__ __attribute__((aligned(16))) double kb[(304 + 64) * P];
  __shared__ double part[8][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < (304 + 64) * P; i += 512) kb[i] = 1e-3 * (i % 17) + 1e-4 * blockIdx.x;
  const int nrt = (wv == 2 || wv == 5 || wv == 7) ? 3 : 2;
  const int nb = nrt == 3 ? 19 : 13;
  v2d bufA[NL], bufB[NL];
#pragma unroll
  for (int s = 0; s < NL; ++s) {
    bufA[s] = (v2d){1.0 + 1e-3 * lane + s, 0.5 - 1e-3 * lane};
    bufB[s] = (v2d){0.25 + 1e-3 * lane - s, 1.5 + 1e-3 * lane};
  }
  const double* ka = kb + (lane >> 4) * P + (lane & 3);
  __syncthreads();
  double tot = 0.0;
  unsigned long long own = 0;
  const unsigned long long t0 = clock64();
  for (int t = 0; t < nstep; ++t) {
    asm volatile("" : "+v"(bufA[0]), "+v"(bufA[1]), "+v"(bufA[2]), "+v"(bufA[3]), "+v"(bufA[4]), "+v"(bufA[5]));
    asm volatile("" : "+v"(bufB[0]), "+v"(bufB[1]), "+v"(bufB[2]), "+v"(bufB[3]), "+v"(bufB[4]), "+v"(bufB[5]));
    const unsigned long long w0 = clock64();
    double s;
    if (FORM == 7 || FORM == 8 || FORM == 9) {
      // this wave's stream: row tiles [rt0, rt0 + nrt) of a 304 x 304 matrix = nrt * 38 tiles of 1 KB; 3 buffers resident (not modelled: their 36
      // MFMAs run on bufA), nb - 3 streamed
      const int rt0 = wv == 0 ? 0 : (wv == 1 ? 2 : (wv == 2 ? 4 : (wv == 3 ? 7 : (wv == 4 ? 9 : (wv == 5 ? 11 : (wv == 6 ? 14 : 16))))));
      const gptr2_t p = (gptr2_t)(tiles + (size_t)rt0 * 38 * 128) + lane;
      const int nstream = nb - 3;
      double acc3[2][3], acc2[2][2];
#pragma unroll
      for (int r = 0; r < 3; ++r) acc3[0][r] = acc3[1][r] = 0.0;
#pragma unroll
      for (int r = 0; r < 2; ++r) acc2[0][r] = acc2[1][r] = 0.0;
      const int gpb = nrt == 3 ? 2 : 3;
      {  // the resident buffers (both forms alike, compiler-scheduled)
        v2d kR[3];
        for (int r = 0; r < 3; ++r) {
          readk(kR, ka, r * gpb);
          use0(bufA, kR, nrt, acc3, acc2);
        }
      }
      if (FORM == 8) {
        v2d A[NL], B[NL], kA[3], kB[3];
        int b = 3;
        ld6(A, p, b);
        readk(kA, ka, b * gpb);
        ld6(B, p, b + 1);
        readk(kB, ka, (b + 1) * gpb);
        for (; b + 2 < nb; b += 2) {
          use0(A, kA, nrt, acc3, acc2);
          ld6(A, p, b + 2);
          readk(kA, ka, (b + 2) * gpb);
          use0(B, kB, nrt, acc3, acc2);
          ld6(B, p, b + 3);
          readk(kB, ka, (b + 3) * gpb);
        }
        use0(A, kA, nrt, acc3, acc2);
        use0(B, kB, nrt, acc3, acc2);
      } else {
        const unsigned kaddr = (unsigned)(size_t)(const __attribute__((address_space(3))) double*)ka + 3u * gpb * 256u;
        const gptr2_t pa = p + (size_t)3 * (NL * 64) + 3 * 64;  // (centred on the fourth tile of the first streamed buffer)
        if (FORM == 9) {
          if (nrt == 3) stream_asm3f(acc3, pa, kaddr, nstream); else stream_asm2f(acc2, pa, kaddr, nstream);
        } else {
          if (nrt == 3) stream_asm3(acc3, pa, kaddr, nstream); else stream_asm2(acc2, pa, kaddr, nstream);
        }
      }
      s = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r) s += acc3[0][r] + acc3[1][r];
#pragma unroll
      for (int r = 0; r < 2; ++r) s += acc2[0][r] + acc2[1][r];
    } else if (FORM == 0) {
      double acc3[2][3], acc2[2][2];
#pragma unroll
      for (int r = 0; r < 3; ++r) acc3[0][r] = acc3[1][r] = 0.0;
#pragma unroll
      for (int r = 0; r < 2; ++r) acc2[0][r] = acc2[1][r] = 0.0;
      const int gpb = nrt == 3 ? 2 : 3;
      v2d kA[3], kB[3];
      readk(kA, ka, 0);
      readk(kB, ka, gpb);
      int b = 0;
      for (; b + 2 < nb; b += 2) {
        use0(bufA, kA, nrt, acc3, acc2);
        readk(kA, ka, (b + 2) * gpb);
        use0(bufB, kB, nrt, acc3, acc2);
        readk(kB, ka, (b + 3) * gpb);
      }
      use0(bufA, kA, nrt, acc3, acc2);
      if (b + 1 < nb) use0(bufB, kB, nrt, acc3, acc2);
      s = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r) s += acc3[0][r] + acc3[1][r];
#pragma unroll
      for (int r = 0; r < 2; ++r) s += acc2[0][r] + acc2[1][r];
    } else {
      s = nrt == 3 ? stream_fixed<3, FORM>(bufA, bufB, ka, nb) : stream_fixed<2, FORM>(bufA, bufB, ka, nb);
    }
    own += clock64() - w0;
    part[wv][lane] = s;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid < 64) {  // a little feedback so that nothing is hoisted: k depends on the previous step's sums
      double q = part[0][tid] + part[3][tid] + part[7][tid];
      kb[tid] = 1e-3 + 1e-12 * q;
      tot += q;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  const unsigned long long t1 = clock64();
  out[blockIdx.x * 512 + tid] = tot;
  if (blockIdx.x == 0 && lane == 0) cyc[1 + wv] = own;
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  const int nwg = 200, nstep = 150;
  double* out;
  unsigned long long* cyc;
  hipMalloc(&out, nwg * 512 * 8);
  hipMalloc(&cyc, 16 * 8);
  double* tiles;
  hipMalloc(&tiles, (size_t)(304 * 304 + 8 * 128 * 6) * 8);
  hipMemset(tiles, 0, (size_t)(304 * 304 + 8 * 128 * 6) * 8);
  const char* names[10] = {"form 0: kernel's (builtin, wiring branch per buffer)", "form 1: compile-time wiring, builtin", "form 2: compile-time wiring, in-place asm",
                          "form 3: form 2, NO LDS read in the loop (fixed operands)", "form 4: form 2, 24 MFMAs per run",
                          "form 5: form 2, a buffer's 6 operand doubles contiguous (b128 reads)", "form 6: form 2, operand reads of two buffers issued together",
                          "form 7: WITH the Kinv stream, the streaming loop one asm block", "form 8: WITH the Kinv stream, as the kernel writes it (compiler)",
                          "form 9: form 7 with every pair of tiles reloaded right behind its own MFMAs"};
  for (int rep = 0; rep < 2; ++rep)
    for (int form = 0; form < 10; ++form) {
      if (form >= 1 && form <= 6) continue;
      hipMemset(cyc, 0, 16 * 8);
      if (form == 0) hipLaunchKernelGGL(vbench<0>, dim3(nwg), dim3(512), 0, 0, nstep, out, cyc, tiles);
      if (form == 1) hipLaunchKernelGGL(vbench<1>, dim3(nwg), dim3(512), 0, 0, nstep, out, cyc, tiles);
      if (form == 2) hipLaunchKernelGGL(vbench<2>, dim3(nwg), dim3(512), 0, 0, nstep, out, cyc, tiles);
      if (form == 3) hipLaunchKernelGGL(vbench<3>, dim3(nwg), dim3(512), 0, 0, nstep, out, cyc, tiles);
      if (form == 4) hipLaunchKernelGGL(vbench<4>, dim3(nwg), dim3(512), 0, 0, nstep, out, cyc, tiles);
      if (form == 5) hipLaunchKernelGGL(vbench<5>, dim3(nwg), dim3(512), 0, 0, nstep, out, cyc, tiles);
      if (form == 6) hipLaunchKernelGGL(vbench<6>, dim3(nwg), dim3(512), 0, 0, nstep, out, cyc, tiles);
      if (form == 7) hipLaunchKernelGGL(vbench<7>, dim3(nwg), dim3(512), 0, 0, nstep, out, cyc, tiles);
      if (form == 8) hipLaunchKernelGGL(vbench<8>, dim3(nwg), dim3(512), 0, 0, nstep, out, cyc, tiles);
      if (form == 9) hipLaunchKernelGGL(vbench<9>, dim3(nwg), dim3(512), 0, 0, nstep, out, cyc, tiles);
      hipDeviceSynchronize();
      unsigned long long hc[16];
      hipMemcpy(hc, cyc, 16 * 8, hipMemcpyDeviceToHost);
      if (rep) {
        printf("%-58s step %6.0f cycles | per wave (own):", names[form], (double)hc[0] / nstep);
        for (int w = 0; w < 8; ++w) printf(" %5.0f", (double)hc[1 + w] / nstep);
        printf(" | heavy wave: %.1f cycles per MFMA\n", (double)hc[1 + 7] / nstep / (19 * 12));
      }
    }
  return 0;
}
