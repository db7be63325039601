// (The last column: the same run on the wall clock.  All 256 CUs issuing fp64 MFMAs back to back run at about half the clock the rollout kernels see --
//  64 flop per cycle and SIMD, 78 TFLOP/s = the datasheet figure on the wall clock.)
// Does the register BANK of its operands decide the issue rate of v_mfma_f64_4x4x4_4b_f64 on gfx950?  (round 6; tools/mfma4x4_probe.hip: 8.2 cycles per
// SIMD with four operand registers used over and over, 17.8 with 12 + 6 distinct ones -- the lean kernel's phase V streams fresh operands.)
// Eight accumulators in rotation, 12 A operands, 6 B operands, every register pair placed by hand: pair at v[base + 4 i + off], off = 0 -> banks 0, 1,
// off = 2 -> banks 2, 3 (a VGPR's bank = its number mod 4).  Variants = (A off, B off, C off); plus: A and B in ONE bank pair but C in the other,
// and the reuse pattern of mode 0 (4 operand registers) written the same way.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_bank_probe tools/mfma_bank_probe.hip && /tmp/mfma_bank_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define S_(x) #x
#define X_(x) S_(x)
#define MF(c, a, b, AO, BO, CO) "v_mfma_f64_4x4x4_4b_f64 v[200+" #c "*4+" X_(CO) ":200+" #c "*4+" X_(CO) "+1], v[100+" #a "*4+" X_(AO) ":100+" #a "*4+" X_(AO) "+1], v[160+" #b "*4+" X_(BO) ":160+" #b "*4+" X_(BO) "+1], v[200+" #c "*4+" X_(CO) ":200+" #c "*4+" X_(CO) "+1]\n"
#define BODY(AO, BO, CO) \
  MF(0, 0, 0, AO, BO, CO) \
  MF(1, 1, 0, AO, BO, CO) \
  MF(2, 2, 1, AO, BO, CO) \
  MF(3, 3, 1, AO, BO, CO) \
  MF(4, 4, 2, AO, BO, CO) \
  MF(5, 5, 2, AO, BO, CO) \
  MF(6, 6, 3, AO, BO, CO) \
  MF(7, 7, 3, AO, BO, CO) \
  MF(0, 8, 4, AO, BO, CO) \
  MF(1, 9, 4, AO, BO, CO) \
  MF(2, 10, 5, AO, BO, CO) \
  MF(3, 11, 5, AO, BO, CO) \
  MF(4, 0, 0, AO, BO, CO) \
  MF(5, 1, 0, AO, BO, CO) \
  MF(6, 2, 1, AO, BO, CO) \
  MF(7, 3, 1, AO, BO, CO) \
  MF(0, 4, 2, AO, BO, CO) \
  MF(1, 5, 2, AO, BO, CO) \
  MF(2, 6, 3, AO, BO, CO) \
  MF(3, 7, 3, AO, BO, CO) \
  MF(4, 8, 4, AO, BO, CO) \
  MF(5, 9, 4, AO, BO, CO) \
  MF(6, 10, 5, AO, BO, CO) \
  MF(7, 11, 5, AO, BO, CO) \
  MF(0, 0, 0, AO, BO, CO) \
  MF(1, 1, 0, AO, BO, CO) \
  MF(2, 2, 1, AO, BO, CO) \
  MF(3, 3, 1, AO, BO, CO) \
  MF(4, 4, 2, AO, BO, CO) \
  MF(5, 5, 2, AO, BO, CO) \
  MF(6, 6, 3, AO, BO, CO) \
  MF(7, 7, 3, AO, BO, CO)
// the reuse pattern: A from 4 registers, B from 2
#define BODYR(AO, BO, CO) \
  MF(0, 0, 0, AO, BO, CO) \
  MF(1, 1, 0, AO, BO, CO) \
  MF(2, 2, 1, AO, BO, CO) \
  MF(3, 3, 1, AO, BO, CO) \
  MF(4, 0, 0, AO, BO, CO) \
  MF(5, 1, 0, AO, BO, CO) \
  MF(6, 2, 1, AO, BO, CO) \
  MF(7, 3, 1, AO, BO, CO) \
  MF(0, 0, 0, AO, BO, CO) \
  MF(1, 1, 0, AO, BO, CO) \
  MF(2, 2, 1, AO, BO, CO) \
  MF(3, 3, 1, AO, BO, CO) \
  MF(4, 0, 0, AO, BO, CO) \
  MF(5, 1, 0, AO, BO, CO) \
  MF(6, 2, 1, AO, BO, CO) \
  MF(7, 3, 1, AO, BO, CO) \
  MF(0, 0, 0, AO, BO, CO) \
  MF(1, 1, 0, AO, BO, CO) \
  MF(2, 2, 1, AO, BO, CO) \
  MF(3, 3, 1, AO, BO, CO) \
  MF(4, 0, 0, AO, BO, CO) \
  MF(5, 1, 0, AO, BO, CO) \
  MF(6, 2, 1, AO, BO, CO) \
  MF(7, 3, 1, AO, BO, CO) \
  MF(0, 0, 0, AO, BO, CO) \
  MF(1, 1, 0, AO, BO, CO) \
  MF(2, 2, 1, AO, BO, CO) \
  MF(3, 3, 1, AO, BO, CO) \
  MF(4, 0, 0, AO, BO, CO) \
  MF(5, 1, 0, AO, BO, CO) \
  MF(6, 2, 1, AO, BO, CO) \
  MF(7, 3, 1, AO, BO, CO)
// banks alternating from one instruction to the next: (A, B, C) = (0, 2, 0), (2, 0, 2), ...
#define BODYALT(AO, BO, CO) \
  MF(0, 0, 0, 0, 2, 0) \
  MF(1, 1, 0, 2, 0, 2) \
  MF(2, 2, 1, 0, 2, 0) \
  MF(3, 3, 1, 2, 0, 2) \
  MF(4, 4, 2, 0, 2, 0) \
  MF(5, 5, 2, 2, 0, 2) \
  MF(6, 6, 3, 0, 2, 0) \
  MF(7, 7, 3, 2, 0, 2) \
  MF(0, 8, 4, 0, 2, 0) \
  MF(1, 9, 4, 2, 0, 2) \
  MF(2, 10, 5, 0, 2, 0) \
  MF(3, 11, 5, 2, 0, 2) \
  MF(4, 0, 0, 0, 2, 0) \
  MF(5, 1, 0, 2, 0, 2) \
  MF(6, 2, 1, 0, 2, 0) \
  MF(7, 3, 1, 2, 0, 2) \
  MF(0, 4, 2, 0, 2, 0) \
  MF(1, 5, 2, 2, 0, 2) \
  MF(2, 6, 3, 0, 2, 0) \
  MF(3, 7, 3, 2, 0, 2) \
  MF(4, 8, 4, 0, 2, 0) \
  MF(5, 9, 4, 2, 0, 2) \
  MF(6, 10, 5, 0, 2, 0) \
  MF(7, 11, 5, 2, 0, 2) \
  MF(0, 0, 0, 0, 2, 0) \
  MF(1, 1, 0, 2, 0, 2) \
  MF(2, 2, 1, 0, 2, 0) \
  MF(3, 3, 1, 2, 0, 2) \
  MF(4, 4, 2, 0, 2, 0) \
  MF(5, 5, 2, 2, 0, 2) \
  MF(6, 6, 3, 0, 2, 0) \
  MF(7, 7, 3, 2, 0, 2)
#define BODYALT2(AO, BO, CO) \
  MF(0, 0, 0, 0, 2, 0) \
  MF(1, 1, 0, 2, 0, 0) \
  MF(2, 2, 1, 0, 2, 2) \
  MF(3, 3, 1, 2, 0, 2) \
  MF(4, 4, 2, 0, 2, 0) \
  MF(5, 5, 2, 2, 0, 0) \
  MF(6, 6, 3, 0, 2, 2) \
  MF(7, 7, 3, 2, 0, 2) \
  MF(0, 8, 4, 0, 2, 0) \
  MF(1, 9, 4, 2, 0, 0) \
  MF(2, 10, 5, 0, 2, 2) \
  MF(3, 11, 5, 2, 0, 2) \
  MF(4, 0, 0, 0, 2, 0) \
  MF(5, 1, 0, 2, 0, 0) \
  MF(6, 2, 1, 0, 2, 2) \
  MF(7, 3, 1, 2, 0, 2) \
  MF(0, 4, 2, 0, 2, 0) \
  MF(1, 5, 2, 2, 0, 0) \
  MF(2, 6, 3, 0, 2, 2) \
  MF(3, 7, 3, 2, 0, 2) \
  MF(4, 8, 4, 0, 2, 0) \
  MF(5, 9, 4, 2, 0, 0) \
  MF(6, 10, 5, 0, 2, 2) \
  MF(7, 11, 5, 2, 0, 2) \
  MF(0, 0, 0, 0, 2, 0) \
  MF(1, 1, 0, 2, 0, 0) \
  MF(2, 2, 1, 0, 2, 2) \
  MF(3, 3, 1, 2, 0, 2) \
  MF(4, 4, 2, 0, 2, 0) \
  MF(5, 5, 2, 2, 0, 0) \
  MF(6, 6, 3, 0, 2, 2) \
  MF(7, 7, 3, 2, 0, 2)
// v_mfma_f64_16x16x4_f64, four accumulators of eight registers in rotation, the same fresh operands
#define BODY16(AO, BO, CO) \
  "v_mfma_f64_16x16x4_f64 v[200+0*8:200+0*8+7], v[100+0*4:100+0*4+1], v[160+0*4:160+0*4+1], v[200+0*8:200+0*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+1*8:200+1*8+7], v[100+1*4:100+1*4+1], v[160+0*4:160+0*4+1], v[200+1*8:200+1*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+2*8:200+2*8+7], v[100+2*4:100+2*4+1], v[160+1*4:160+1*4+1], v[200+2*8:200+2*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+3*8:200+3*8+7], v[100+3*4:100+3*4+1], v[160+1*4:160+1*4+1], v[200+3*8:200+3*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+0*8:200+0*8+7], v[100+4*4:100+4*4+1], v[160+2*4:160+2*4+1], v[200+0*8:200+0*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+1*8:200+1*8+7], v[100+5*4:100+5*4+1], v[160+2*4:160+2*4+1], v[200+1*8:200+1*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+2*8:200+2*8+7], v[100+6*4:100+6*4+1], v[160+3*4:160+3*4+1], v[200+2*8:200+2*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+3*8:200+3*8+7], v[100+7*4:100+7*4+1], v[160+3*4:160+3*4+1], v[200+3*8:200+3*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+0*8:200+0*8+7], v[100+8*4:100+8*4+1], v[160+4*4:160+4*4+1], v[200+0*8:200+0*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+1*8:200+1*8+7], v[100+9*4:100+9*4+1], v[160+4*4:160+4*4+1], v[200+1*8:200+1*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+2*8:200+2*8+7], v[100+10*4:100+10*4+1], v[160+5*4:160+5*4+1], v[200+2*8:200+2*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+3*8:200+3*8+7], v[100+11*4:100+11*4+1], v[160+5*4:160+5*4+1], v[200+3*8:200+3*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+0*8:200+0*8+7], v[100+0*4:100+0*4+1], v[160+0*4:160+0*4+1], v[200+0*8:200+0*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+1*8:200+1*8+7], v[100+1*4:100+1*4+1], v[160+0*4:160+0*4+1], v[200+1*8:200+1*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+2*8:200+2*8+7], v[100+2*4:100+2*4+1], v[160+1*4:160+1*4+1], v[200+2*8:200+2*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+3*8:200+3*8+7], v[100+3*4:100+3*4+1], v[160+1*4:160+1*4+1], v[200+3*8:200+3*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+0*8:200+0*8+7], v[100+4*4:100+4*4+1], v[160+2*4:160+2*4+1], v[200+0*8:200+0*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+1*8:200+1*8+7], v[100+5*4:100+5*4+1], v[160+2*4:160+2*4+1], v[200+1*8:200+1*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+2*8:200+2*8+7], v[100+6*4:100+6*4+1], v[160+3*4:160+3*4+1], v[200+2*8:200+2*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+3*8:200+3*8+7], v[100+7*4:100+7*4+1], v[160+3*4:160+3*4+1], v[200+3*8:200+3*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+0*8:200+0*8+7], v[100+8*4:100+8*4+1], v[160+4*4:160+4*4+1], v[200+0*8:200+0*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+1*8:200+1*8+7], v[100+9*4:100+9*4+1], v[160+4*4:160+4*4+1], v[200+1*8:200+1*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+2*8:200+2*8+7], v[100+10*4:100+10*4+1], v[160+5*4:160+5*4+1], v[200+2*8:200+2*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+3*8:200+3*8+7], v[100+11*4:100+11*4+1], v[160+5*4:160+5*4+1], v[200+3*8:200+3*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+0*8:200+0*8+7], v[100+0*4:100+0*4+1], v[160+0*4:160+0*4+1], v[200+0*8:200+0*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+1*8:200+1*8+7], v[100+1*4:100+1*4+1], v[160+0*4:160+0*4+1], v[200+1*8:200+1*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+2*8:200+2*8+7], v[100+2*4:100+2*4+1], v[160+1*4:160+1*4+1], v[200+2*8:200+2*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+3*8:200+3*8+7], v[100+3*4:100+3*4+1], v[160+1*4:160+1*4+1], v[200+3*8:200+3*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+0*8:200+0*8+7], v[100+4*4:100+4*4+1], v[160+2*4:160+2*4+1], v[200+0*8:200+0*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+1*8:200+1*8+7], v[100+5*4:100+5*4+1], v[160+2*4:160+2*4+1], v[200+1*8:200+1*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+2*8:200+2*8+7], v[100+6*4:100+6*4+1], v[160+3*4:160+3*4+1], v[200+2*8:200+2*8+7]\n" \
  "v_mfma_f64_16x16x4_f64 v[200+3*8:200+3*8+7], v[100+7*4:100+7*4+1], v[160+3*4:160+3*4+1], v[200+3*8:200+3*8+7]\n"
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231"
#define VARIANT(NAME, BODYM, AO, BO, CO)                                              \
  __device__ __forceinline__ void NAME(int nit) {                                      \
    for (int it = 0; it < nit; ++it) asm volatile(BODYM(AO, BO, CO)::: CLOB);           \
  }
VARIANT(v_a0b2c0, BODY, 0, 2, 0)
VARIANT(v_a0b0c0, BODY, 0, 0, 0)
VARIANT(v_a0b2c2, BODY, 0, 2, 2)
VARIANT(v_a0b0c2, BODY, 0, 0, 2)
VARIANT(r_a0b2c0, BODYR, 0, 2, 0)
VARIANT(r_a0b0c0, BODYR, 0, 0, 0)
VARIANT(v_alt, BODYALT, 0, 0, 0)
VARIANT(v_16, BODY16, 0, 0, 0)
VARIANT(v_alt2, BODYALT2, 0, 0, 0)
__global__ __launch_bounds__(512) void rate(int mode, int nit, unsigned long long* cyc) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // every register of the three ranges = 1.0 (hi word) / 0 (lo word)
  asm volatile("v_mov_b32 v100, 0\n"
               "v_mov_b32 v101, 0x3ff00000\n"
               "v_mov_b32 v102, 0\n"
               "v_mov_b32 v103, 0x3ff00000\n"
               "v_mov_b32 v104, 0\n"
               "v_mov_b32 v105, 0x3ff00000\n"
               "v_mov_b32 v106, 0\n"
               "v_mov_b32 v107, 0x3ff00000\n"
               "v_mov_b32 v108, 0\n"
               "v_mov_b32 v109, 0x3ff00000\n"
               "v_mov_b32 v110, 0\n"
               "v_mov_b32 v111, 0x3ff00000\n"
               "v_mov_b32 v112, 0\n"
               "v_mov_b32 v113, 0x3ff00000\n"
               "v_mov_b32 v114, 0\n"
               "v_mov_b32 v115, 0x3ff00000\n"
               "v_mov_b32 v116, 0\n"
               "v_mov_b32 v117, 0x3ff00000\n"
               "v_mov_b32 v118, 0\n"
               "v_mov_b32 v119, 0x3ff00000\n"
               "v_mov_b32 v120, 0\n"
               "v_mov_b32 v121, 0x3ff00000\n"
               "v_mov_b32 v122, 0\n"
               "v_mov_b32 v123, 0x3ff00000\n"
               "v_mov_b32 v124, 0\n"
               "v_mov_b32 v125, 0x3ff00000\n"
               "v_mov_b32 v126, 0\n"
               "v_mov_b32 v127, 0x3ff00000\n"
               "v_mov_b32 v128, 0\n"
               "v_mov_b32 v129, 0x3ff00000\n"
               "v_mov_b32 v130, 0\n"
               "v_mov_b32 v131, 0x3ff00000\n"
               "v_mov_b32 v132, 0\n"
               "v_mov_b32 v133, 0x3ff00000\n"
               "v_mov_b32 v134, 0\n"
               "v_mov_b32 v135, 0x3ff00000\n"
               "v_mov_b32 v136, 0\n"
               "v_mov_b32 v137, 0x3ff00000\n"
               "v_mov_b32 v138, 0\n"
               "v_mov_b32 v139, 0x3ff00000\n"
               "v_mov_b32 v140, 0\n"
               "v_mov_b32 v141, 0x3ff00000\n"
               "v_mov_b32 v142, 0\n"
               "v_mov_b32 v143, 0x3ff00000\n"
               "v_mov_b32 v144, 0\n"
               "v_mov_b32 v145, 0x3ff00000\n"
               "v_mov_b32 v146, 0\n"
               "v_mov_b32 v147, 0x3ff00000\n"
               "v_mov_b32 v160, 0\n"
               "v_mov_b32 v161, 0x3ff00000\n"
               "v_mov_b32 v162, 0\n"
               "v_mov_b32 v163, 0x3ff00000\n"
               "v_mov_b32 v164, 0\n"
               "v_mov_b32 v165, 0x3ff00000\n"
               "v_mov_b32 v166, 0\n"
               "v_mov_b32 v167, 0x3ff00000\n"
               "v_mov_b32 v168, 0\n"
               "v_mov_b32 v169, 0x3ff00000\n"
               "v_mov_b32 v170, 0\n"
               "v_mov_b32 v171, 0x3ff00000\n"
               "v_mov_b32 v172, 0\n"
               "v_mov_b32 v173, 0x3ff00000\n"
               "v_mov_b32 v174, 0\n"
               "v_mov_b32 v175, 0x3ff00000\n"
               "v_mov_b32 v176, 0\n"
               "v_mov_b32 v177, 0x3ff00000\n"
               "v_mov_b32 v178, 0\n"
               "v_mov_b32 v179, 0x3ff00000\n"
               "v_mov_b32 v180, 0\n"
               "v_mov_b32 v181, 0x3ff00000\n"
               "v_mov_b32 v182, 0\n"
               "v_mov_b32 v183, 0x3ff00000\n"
               "v_mov_b32 v200, 0\n"
               "v_mov_b32 v201, 0x3ff00000\n"
               "v_mov_b32 v202, 0\n"
               "v_mov_b32 v203, 0x3ff00000\n"
               "v_mov_b32 v204, 0\n"
               "v_mov_b32 v205, 0x3ff00000\n"
               "v_mov_b32 v206, 0\n"
               "v_mov_b32 v207, 0x3ff00000\n"
               "v_mov_b32 v208, 0\n"
               "v_mov_b32 v209, 0x3ff00000\n"
               "v_mov_b32 v210, 0\n"
               "v_mov_b32 v211, 0x3ff00000\n"
               "v_mov_b32 v212, 0\n"
               "v_mov_b32 v213, 0x3ff00000\n"
               "v_mov_b32 v214, 0\n"
               "v_mov_b32 v215, 0x3ff00000\n"
               "v_mov_b32 v216, 0\n"
               "v_mov_b32 v217, 0x3ff00000\n"
               "v_mov_b32 v218, 0\n"
               "v_mov_b32 v219, 0x3ff00000\n"
               "v_mov_b32 v220, 0\n"
               "v_mov_b32 v221, 0x3ff00000\n"
               "v_mov_b32 v222, 0\n"
               "v_mov_b32 v223, 0x3ff00000\n"
               "v_mov_b32 v224, 0\n"
               "v_mov_b32 v225, 0x3ff00000\n"
               "v_mov_b32 v226, 0\n"
               "v_mov_b32 v227, 0x3ff00000\n"
               "v_mov_b32 v228, 0\n"
               "v_mov_b32 v229, 0x3ff00000\n"
               "v_mov_b32 v230, 0\n"
               "v_mov_b32 v231, 0x3ff00000\n" ::: CLOB);
  __syncthreads();
  const unsigned long long t0 = clock64();
  if (mode == 0) v_a0b2c0(nit);
  else if (mode == 1) v_a0b0c0(nit);
  else if (mode == 2) v_a0b2c2(nit);
  else if (mode == 3) v_a0b0c2(nit);
  else if (mode == 4) r_a0b2c0(nit);
  else if (mode == 5) r_a0b0c0(nit);
  else if (mode == 6) v_alt(nit);
  else if (mode == 7) v_alt2(nit);
  else v_16(nit);
  asm volatile("s_nop 15\ns_nop 15" ::: "memory");
  const unsigned long long t1 = clock64();
  if (lane == 0 && blockIdx.x == 0) cyc[wv] = t1 - t0;
}
int main() {
  unsigned long long* cyc;
  (void)hipMalloc(&cyc, 64);
  const int nit = 200;
  const char* names[9] = {"fresh operands: A banks 0,1  B banks 2,3  C banks 0,1", "fresh operands: A, B, C all banks 0,1", "fresh operands: A banks 0,1  B, C banks 2,3",
                          "fresh operands: A, B banks 0,1  C banks 2,3", "4 + 2 operand registers: A 0,1  B 2,3  C 0,1", "4 + 2 operand registers: all banks 0,1", "fresh operands, (A, B, C) banks (01, 23, 01) / (23, 01, 23) in turn", "fresh operands, A / B alternate, C alternates every second", "v_mfma_f64_16x16x4_f64, 4 accumulators, fresh operands"};
  for (int mode = 0; mode < 9; ++mode) {
    hipLaunchKernelGGL(rate, dim3(256), dim3(512), 0, 0, mode, nit, cyc);
    hipLaunchKernelGGL(rate, dim3(256), dim3(512), 0, 0, mode, nit, cyc);
    (void)hipDeviceSynchronize();
    unsigned long long h[8];
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const double per_wave = (double)h[0] / (nit * 32.0);
    // the same on the wall clock (the cycle counter need not tick at the clock the matrix pipe runs at): 20 x the iterations between two events
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate, dim3(256), dim3(512), 0, 0, mode, 20 * nit, cyc);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (mode == 8 ? 2048.0 : 512.0) * 256.0 * 8.0 * 20.0 * nit * 32.0;
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);  // (wave 0's cycle count of the long run: cycles / wall time = the clock the run had)
    printf("%-58s %6.1f cycles per instruction and wave, %5.1f per SIMD | wall clock: %.3f ms, %.1f TFLOP/s on 256 CUs, %.2f GHz\n", names[mode], per_wave, per_wave / 2, ms,
           flop / (ms * 1e-3) * 1e-12, (double)h[0] / (ms * 1e-3) * 1e-9);
  }
  return 0;
}
