// Latency-lean GP-sharded forward rollout for small swarms of narrow models (BASELINE.json's headline: cart-pole, M = 400), split
// from rollout_fwd.hip (its own translation unit: the instantiations compile in parallel with the general kernel's).
// Contract, cluster geometry and hand-off protocol: rollout_fwd.hip (rollout_fwd_kernel<P, true, ., true>).
//
// Replaces MC_PILCO.apply_policy (policy_learning/MC_PILCO.py:615-674) for these shapes: per time step
//   Sum_of_gaussians*.forward                (policy_learning/Policy.py:242-265, 323-335)
//   Model_learning.get_next_state            (model_learning/Model_learning.py:210-242, 265-336, 685-718)
//     -> GP_prior.get_estimate_from_alpha    (gpr_lib/GP_prior/GP_prior.py:137-155), one GP per workgroup
#include "rollout_fwd_shared.h"

using namespace mcp;

// =======================================================================================
// Latency-lean GP-sharded kernel for narrow SE-only models (BASELINE.json's headline: cart-pole, M = 400)
// =======================================================================================
// Same contract and the same cluster / hand-off protocol as rollout_fwd_kernel<P, true, 0, true>, rebuilt around what the stamps of
// that kernel show: at M = 400 a step is a chain of barrier-separated phases whose cost is their INSTRUCTION COUNT (a wave issues
// about one instruction per 5 cycles whatever it is) plus the N^2 product v = Kinv k, which in v_fmac_f64_dpp form costs 10 k cycles
// of vector issue beside a 9-10 k cycle L2 -> CU stream.
//   * FOUR workgroup barriers per step instead of eight.  Policy: phi_b w_kb summed over aligned groups of 16 basis functions
//     inside one DPP row (the same tree whatever P is), group sums to LDS, ONE barrier, then every thread adds the group sums in
//     a fixed order and squashes for itself -- there is no phase U.  The state-only part of the GP distances (D - U of the D input
//     dimensions are known once the state is) is formed beside the policy features, so after u only one fused multiply-add per
//     input and the exp remain (no phase K of its own).
//   * everything a thread reads repeatedly is pre-scaled and laid out at launch: X^T / l and the policy centres / l (zero padded to
//     fixed dimension counts: no run-time bounds inside the unrolled loops, padded terms add exact zeros), the published features
//     are already divided by their lengthscales; per-thread roles of the serial section live in an LDS table and the thread id
//     goes through an opaque move once per step (what is derived from it is recomputed, not held in registers across phase V).
//   * phase V on the matrix cores: Kinv as 16 x 8 operand tiles of v_mfma_f64_4x4x4_4b_f64 (kt_pack_kernel), every wave owns whole
//     row tiles, so v is complete inside a wave: no cross-wave partial sums, no phase "vsum", and the wave goes straight on to its
//     share of phase J (4x4x4 MFMA as well, the rows it has just produced) without a workgroup barrier in between.
//   * phase F, the hand-off, the integrator and the next step's phase S run back to back in wave 0 (wave-level ordering only, the
//     8 partial tiles of phase J summed by 64 lanes at once); the other waves meanwhile draw the next step's dropout decisions and
//     process noise (never wave 4, which shares wave 0's SIMD).
// Results do not depend on P (1, 2, 4): equal shards reproduce each other bit for bit, as before.  Covers SE-only models with
// D <= 8 (<= 6 state-derived + <= 2 inputs), <= 6 policy features, 32 <= Npad <= 384; everything else runs the general kernel
// (rollout_fwd.hip).
//
// Round 4: the template parameters MAXDEG and PMS extend the kernel to the reference's other small-swarm launch scripts:
//   * MAXDEG = 1, 2 -- SE + Volterra polynomial kernels (gpr_lib/GP_prior/Sparse_GP.py:559-737; test_mcpilco_cartpole.py:67-101), all GPs
//     of the model of the same degree.  With p1_j = w1_D + sum_d w1_d z_d X_jd, A_j = sum_d w20_d z_d X_jd, B_j = sum_d w21_d z_d X_jd:
//     k_j = kse_j + p1_j + A_j B_j.  The three bilinear forms ride in the policy + K(state) pass (one product z_d X_jd per dimension
//     feeds all of them), phase V is unchanged (its operand is the total k), and phase J takes NW = 4 / 6 weight columns per particle
//     instead of 2, one 16-byte-aligned record per (training point, particle), W[j][p NW + s]:  [kse alpha | kse v | v | k v]  (degree 1),
//     [kse alpha | kse v | v B | v A | v | k v]  (degree 2): phase K fills the slots that do not need v with vector stores, the tail of
//     phase V multiplies by v in place and appends the rest, each with at most three LDS instructions per side --
//     the alpha-weighted polynomial sums need no column at all: they are launch constants contracted with z
//     (sum_j alpha_j, sum_j alpha_j X_jd, sum_j alpha_j X_jc X_je), formed by an otherwise idle wave while the others finish phase K.
//   * PMS -- the measurement model of MC_PILCO4PMS.apply_policy (policy_learning/MC_PILCO.py:808-906, mcp_meas): phase S also produces
//     what the policy sees (noisy positions, backward-difference velocities through the first-order filter: three carried values per
//     lane); the position noise of step t is drawn two steps ahead by the lanes of wave 1 that draw the process noise.
#define RL_DSM 6  // state-derived GP-input dimensions (D - U), zero padded
#define RL_UM 2   // inputs, zero padded
#define RL_ZD (RL_DSM + RL_UM)
#define RL_PFM 6  // policy features, zero padded
#ifndef RL_NRES
#define RL_NRES 2  // register buffers kept resident: RL_NRES or RL_NRES + 1, whichever leaves an even number to stream
#endif
#ifndef RL_NRES_CUT2
#define RL_NRES_CUT2 0  // resident buffers the degree-2 / 4-particle instantiation gives up (experiment switch)
#endif
#ifndef RL_PRE
#define RL_PRE 0  // register buffers of the Kinv stream issued ahead of the barrier that ends phase K (0, 1, 2): measured equal
#endif           // within 1 % (what the stream gains the phase before it loses waiting at the full memory queue)

// LDS plan: the regions whose size is bounded by compile-time limits come first, at compile-time offsets (no scalar register per
// pointer); the ones sized by N and B follow
#define RL_MAXD RL_MAXD_  // D <= RL_DSM + RL_UM
struct LatFixed {
  static constexpr int pol = 0;                                // policy inverse lengthscales | u_max | bias | 1 / u_max
  static constexpr int xs = pol + RL_PFM + 3 * MCP_MAX_INPUT;  // [2][64] state, double buffered by step parity
  static constexpr int z = xs + 2 * 64;                        // [P][D] raw GP input (phase F: Jacobians)
  static constexpr int zs = z + 4 * RL_MAXD;                   // [P][RL_ZD] GP input / lengthscale, zero padded
  static constexpr int sf = zs + 4 * RL_ZD;                    // [P][RL_PFM] policy features / lengthscale, zero padded
  static constexpr int dl = sf + 4 * RL_PFM;                   // [P][G] delta_g | abort word
  static constexpr int eps = dl + 4 * MCP_MAX_GP + 2;          // [2][P] process noise of this workgroup's GP, by step parity
  static constexpr int red = eps + 2 * 4;                      // [RF_NW][NCG <= 3][8][8] phase-J partial tiles
  static constexpr int rt = red + RF_NW * 3 * 64;              // [NCG][8][8] their sum (phase F)
  static constexpr int pc = rt + 3 * 64;                       // polynomial constants: kc1 | kcA | kcB [8 each, by xq row] | w1_D | sum alpha | AXX [8][8] | lengthscale by dimension [8]
  static constexpr int fz = pc + 24 + 4 + 64 + 8;              // per step, z-only polynomial terms: [P][4] mpoly, kzz, Sa, Sb | [P][8][2] qB_c, qA_c
  static constexpr int pn = fz + 4 * 4 + 4 * 8 * 2;            // [2][32] position measurement noise by step parity (PMS)
  static constexpr int gpl = pn + 2 * 32;
  static constexpr int kpar = gpl + GPL_DOUBLES;
  static constexpr int role = ((kpar + 5 * RL_MAXD + 2) + 1) & ~1;          // [64][16] ints: roles of the threads of wave 0 in the serial section
  static constexpr int sro = role + 64 * 8;                    // [64][8]: phase S of thread (p, s): 6 int LDS addresses (in doubles) | 4 scale factors
  static constexpr int dump = sro + 64 * 8;                    // where phase S writes what a state component does not feed
  static constexpr int stl = dump + 2;                         // [16 + 8] u64 phase-cycle totals (diagnostic)
  static constexpr int end = stl + 24;
};
static_assert(LatFixed::pc % 2 == 0 && LatFixed::fz % 2 == 0 && LatFixed::gpl % 2 == 0, "16-byte alignment");
static_assert(LatFixed::red % 2 == 0 && LatFixed::rt % 2 == 0 && LatFixed::zs % 2 == 0 && LatFixed::role % 2 == 0 && LatFixed::sro % 2 == 0 && LatFixed::end % 2 == 0, "16-byte alignment of the v2d regions");
struct LatLayout {
  int gs, kb, vb, xq, al, cen, wgt, mk, total;  // offsets in doubles
};
__host__ __device__ inline int lat_ng(int B) { return (B + 15) >> 4; }                 // groups of 16 basis functions
__host__ __device__ inline int lat_ngp(int B) { return ((lat_ng(B) + 7) >> 3) << 3; }  // padded to whole chunks of 8 (zeros)
// phase-J weight columns per particle, column groups of 8 (one 4x4x4 B operand each)
__host__ __device__ constexpr int lat_nw(int deg) { return deg == 0 ? 2 : (deg == 1 ? 4 : 6); }
__host__ __device__ constexpr int lat_ncg(int P, int deg) { return (P * lat_nw(deg) + 7) / 8; }
__host__ __device__ inline LatLayout lat_layout(int P, int B, int Npad, int maxdeg) {
  LatLayout L;
  int o = LatFixed::end;
  auto take = [&](int n) {
    int r = o;
    o += (n + 1) & ~1;
    return r;
  };
  const int Bp = lat_ng(B) * 16;
  L.gs = take(RL_UM * P * lat_ngp(B));
  L.kb = take((Npad + 32) * P);   // (+ KT_ZROWS zero rows)
  L.vb = take(Npad * P * lat_nw(maxdeg) + 8);  // phase-J weights: W[j][2p + a] (SE only), W[j][s P + p] (polynomial); + the last column group's overhang
  L.xq = take(RL_ZD * Npad);      // X^T / l by row (RL_DSM state-derived dimensions, RL_UM inputs, zero padded) with ONES in the first padded row:
                                  // phase K's operand and phase J's A operand [X^T / l; 1] (round 4 kept a second, unscaled copy: 8 Npad doubles)
  L.al = take(Npad);
  L.cen = take(RL_PFM * Bp);
  L.wgt = take(RL_UM * Bp);
  L.mk = take((P * ((B + 3) / 4) + 1) / 2);
  L.total = o;
  return L;
}
// rows of xq: GP-input dimension d sits in row d (state-derived, d < DS) or RL_DSM + (d - DS) (inputs); the first padded row carries ones
// (there is one whenever D <= 7: lean_applies) -- phase J's unweighted sums; phase K sees z = 1 there: a zero distance term
__host__ __device__ inline int lat_row_of(int d, int DS) { return d < DS ? d : RL_DSM + (d - DS); }
__host__ __device__ inline int lat_ones_row(int DS, int U) { return DS < RL_DSM ? DS : RL_DSM + U; }
#define LAT_PC_LS 92  // pc[LAT_PC_LS + d]: lengthscale of dimension d
// role table entries (ints) of a thread of wave 0
enum { RO_OP, RO_OS, RO_ZPLAIN, RO_ZANG, RO_PPLAIN, RO_PANG, RO_GVEL, RO_GPOS, RO_VELOFPOS, RO_PMPOS, RO_PMVEL, RO_PMPAIR, RO_FP, RO_FC, RO_OM, RO_FROW };

// ---- Kinv as MFMA operand tiles (lean kernel, phase V) ----------------------------------------------------------------
// v = Kinv k for P <= 4 particles is [N x N] x [N x 4]: v_mfma_f64_4x4x4_4b_f64 -- four independent 4x4x4 products per
// instruction -- takes a [16 rows x 4 columns] block of Kinv against k[4 columns][4 particles] with every lane busy, where
// the 16x16x4 form would run 3/4 empty.  (Measured, tools/v4_bench.hip: the v_fmac_f64_dpp form of the general kernel costs
// 10.2 k cycles of vector issue per step beside a 9.4 k stream, 12.7 k together; the MFMA form overlaps with the stream.)
// Operand layout of the instruction (tools/mfma4x4_probe.hip): lane l = 16 k + 4 blk + e;  A: A_blk[i = e][k],  B: B_blk[k][j = e],
// D: lane 16 i + 4 blk + j.  A tile = 16 rows x 8 columns of Kinv in that lane order, two doubles per lane (columns J0 + k and
// J0 + 4 + k): one global_load_dwordx4 = 1 KB contiguous per wave feeds two MFMAs.  Wave w owns the row tiles
// [rt0(w), rt0(w + 1)) for ALL columns -- every v_i is complete inside one wave: no cross-wave partial sums -- and its tiles are
// stored in the order it streams them:  [w][column group jg][row tile r][lane][2].  Built once per rollout in the caller's
// workspace (kt_pack_kernel; 2 x 0.7 MB for the cart-pole model).
// The row tiles of a GP are dealt to the waves in parts of 2 or 3 (all 8 waves from 16 row tiles on; 2 <= row tiles <= 24,
// i.e. 32 <= Npad <= 384): a register buffer is then always 6 CONSECUTIVE tiles of the wave's stream -- two column groups x 3 row
// tiles or three groups x 2 -- so the loads, the double buffering and the resident buffers are one code path; only the wiring of
// the 12 MFMAs of a buffer (which accumulator, which k operand) differs.
// Round 5: beyond 24 row tiles (Npad > 384, up to 48 = Npad 768) a wave streams TWO segments of 2 or 3 row tiles one after the other through
// the same code, each with fresh accumulators, its own tail and its own share of phase J.  The phase is bound per SIMD (waves w and w + 4 share
// one's issue port and matrix pipe), so the tiles are dealt per SIMD first -- nrtt / 4 each, the remainder to the last SIMDs (SIMD 0 hosts the
// serial section's wave) --, a SIMD's share goes to its two waves (the older one gets the smaller half), and a wave's 4 / 5 / 6 tiles become
// segments of (2, 2) / (3, 2) / (3, 3); 2 or 3 tiles: one segment.  (Dealing nrtt / 14 .. 16 "virtual waves" round the real ones left the SIMDs
// with 6 / 6 / 9 / 8 tiles at N = 450: phase V of the slowest wave 33.5 k cycles against 18.2 k of the fastest.)
// Segment (w, seg) = virtual wave seg * 8 + w; tiles are stored virtual wave by virtual wave.
#define KT_NL 6
#define KT_MAX_RT 48
__host__ __device__ inline int kt_waves(int nrtt) { return nrtt >= 2 * RF_NW ? RF_NW : (nrtt >= 2 ? nrtt / 2 : 1); }
// row tiles of virtual wave vw (0 .. 15): count
__host__ __device__ inline int kt_vcount(int nrtt, int vw) {
  const int w = vw & (RF_NW - 1), seg = vw >> 3;
  if (nrtt <= 3 * RF_NW) {  // one segment per wave, floor distribution
    if (seg) return 0;
    const int nw = kt_waves(nrtt);
    return w >= nw ? 0 : (nrtt * (w + 1)) / nw - (nrtt * w) / nw;
  }
  const int s = w & 3, q = nrtt / 4 + (s >= 4 - (nrtt & 3) ? 1 : 0);  // this SIMD's share
  const int c = w < 4 ? q / 2 : q - q / 2;                            // this wave's
  if (c <= 3) return seg ? 0 : c;
  return seg ? c / 2 : c - c / 2;
}
__host__ __device__ inline int kt_vrt0(int nrtt, int vw) {
  int r = 0;
  for (int v = 0; v < vw; ++v) r += kt_vcount(nrtt, v);
  return r;
}
__global__ void kt_pack_kernel(mcp_model md, double* __restrict__ kt, int stride) {
  const mcp_gp& gp = md.gp[blockIdx.y];
  const int Npad = gp.Npad, nrtt = Npad >> 4, njg = Npad >> 3;
  double* out = kt + (size_t)blockIdx.y * stride;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < Npad * Npad; idx += gridDim.x * blockDim.x) {
    // idx = ((rt * njg + jg) * 64 + l) * 2 + h  in "row tile major" numbering; the destination re-orders the tiles per wave
    const int h = idx & 1, l = (idx >> 1) & 63, tile = idx >> 7;
    const int rt = tile / njg, jg = tile - rt * njg;
    int w = 0, r0 = 0;
    while (w + 1 < 2 * RF_NW && r0 + kt_vcount(nrtt, w) <= rt) {  // (virtual wave that owns row tile rt)
      r0 += kt_vcount(nrtt, w);
      ++w;
    }
    const int nrt = kt_vcount(nrtt, w);
    const int row = 16 * rt + 4 * ((l >> 2) & 3) + (l & 3), col = 8 * jg + 4 * h + (l >> 4);
    out[((size_t)r0 * njg + (size_t)jg * nrt + (rt - r0)) * 128 + 2 * l + h] = gp.Kinv[(size_t)row * Npad + col];
  }
  // a wave's last register buffer may run up to 5 tiles past its stream (those tiles meet k = 0): what lies behind the last stream of a
  // GP must be finite -- the next GP's tiles are, the gap before them (a GP with fewer rows than the largest) and the slack behind
  // the last GP are zeroed here
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < KT_NL * 128; idx += gridDim.x * blockDim.x) {
    const size_t pos = (size_t)Npad * Npad + idx;
    if (pos < (size_t)stride || blockIdx.y + 1 == gridDim.y) out[pos] = 0.0;
  }
}

// One instruction = four independent 4x4x4 products.  Through the BUILTIN, not inline asm: the compiler then knows the
// instruction and inserts the wait states its results need before a vector instruction may read them (it likes to copy
// accumulators with v_mov_b64 where branches meet; behind an asm statement those copies read registers the matrix core has not
// written yet -- seen as wrong trajectories the moment the copies landed right behind an MFMA).  Measured equal in speed
// (tools/v4_bench.hip, both forms).
__device__ __forceinline__ void mfma4(double& acc, double a, double b) {
#ifdef RLX_NOFMA  // experiment: the stream alone
  asm volatile("" : "+v"(acc) : "v"(a), "v"(b));
#else
  acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
#endif
}

// buffer b of this wave's stream = its tiles [6 b, 6 b + 6) (p: the lane's slot of tile 0).  A buffer may run past the end of the
// stream: those tiles are the next wave's (the workspace ends with a buffer of slack) and meet k = 0.
__device__ __forceinline__ void kt_load(v2d (&A)[KT_NL], gptr2_t p, int b) {
  const gptr2_t pb = p + (size_t)b * (KT_NL * 64);
#ifdef RLX_NOLOAD  // experiment: the MFMAs alone
#pragma unroll
  for (int s = 0; s < KT_NL; ++s) asm volatile("" : "=v"(A[s]) : "v"(pb));
  return;
#endif
#pragma unroll
  for (int s = 0; s < KT_NL; ++s) A[s] = pb[s * 64];
}
// the B operands of a buffer, one per column group (2 groups of 3 tiles, or 3 groups of 2): lane l -> k[8 jg + (l >> 4)][l & 3] and
// k[8 jg + 4 + (l >> 4)][l & 3]   (kb is [j][P] with KT_ZROWS zero rows behind row Npad: the groups past the end of a stream read
// zeros, nothing is clamped or selected).  `ka` = this lane's address of group 0 (lanes of absent particles borrow particle 0:
// their output columns are never stored); the three reads are one base register + immediate offsets.
#define KT_ZROWS 32
template <int P>
__device__ __forceinline__ void kt_readk(v2d (&K)[3], const double* ka, int g) {
  const double* kp = ka + g * (8 * P);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    K[q].x = kp[q * 8 * P];
    K[q].y = kp[q * 8 * P + 4 * P];
  }
}
// 12 MFMAs; two accumulator sets (first / second half of a column group): the same accumulator comes up again 2 nrt MFMAs later.
// The two wirings keep their own accumulators (a wave only ever takes one of them): nothing to copy where the branches meet.
// One wait for the whole buffer (the pin), not one per tile: the wave's instruction issue is what bounds the phase.
__device__ __forceinline__ void kt_use(const v2d (&A)[KT_NL], const v2d (&K)[3], int nrt, double (&acc3)[2][3], double (&acc2)[2][2]) {
  asm volatile("" ::"v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(A[4]), "v"(A[5]), "v"(K[0]), "v"(K[1]), "v"(K[2]));
  if (nrt == 3) {  // wave-uniform
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int r = 0; r < 3; ++r) mfma4(acc3[0][r], A[3 * q + r].x, K[q].x);
#pragma unroll
      for (int r = 0; r < 3; ++r) mfma4(acc3[1][r], A[3 * q + r].y, K[q].y);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
      for (int r = 0; r < 2; ++r) mfma4(acc2[0][r], A[2 * q + r].x, K[q].x);
#pragma unroll
      for (int r = 0; r < 2; ++r) mfma4(acc2[1][r], A[2 * q + r].y, K[q].y);
    }
  }
}
// this wave's whole stream, double buffered; bufA / bufB hold buffers nres and nres + 1 already when npre says so (issued ahead of
// the barrier).  The steady-state loop issues its reloads UNCONDITIONALLY (nb - nres is even and >= 2, kt_resident_count): with a
// test around the reloads the compiler must assume at every use that no younger loads are outstanding and waits for vmcnt(5..0) --
// for the OTHER buffer's loads as well, i.e. no double buffering at all (seen in the ISA).
template <int P, int NRES>
__device__ __forceinline__ void kt_stream(gptr2_t p, int nrt, int njg, const double* kb, int lane, double (&acc3)[2][3], double (&acc2)[2][2],
                                          const v2d (&res)[NRES + 1][KT_NL], int rlo, int nres, v2d (&bufA)[KT_NL], v2d (&bufB)[KT_NL],
                                          int npre) {
  // (the stream's first nres buffers are resident in res[rlo .. rlo + nres))
  const int nb = (nrt * njg + KT_NL - 1) / KT_NL;
  const int gpb = nrt == 3 ? 2 : 3;  // column groups per buffer
  const double* ka = kb + (lane >> 4) * P + ((lane & 3) < P ? (lane & 3) : 0);
  int b = nres;
  v2d kA[3], kB[3];
  if (npre < 1) kt_load(bufA, p, b);
  kt_readk<P>(kA, ka, b * gpb);
  if (npre < 2) kt_load(bufB, p, b + 1);
  kt_readk<P>(kB, ka, (b + 1) * gpb);
#pragma unroll
  for (int r = 0; r < NRES + 1; ++r) {
    if (r >= rlo && r < rlo + nres) {  // wave-uniform
      v2d kR[3];
      kt_readk<P>(kR, ka, (r - rlo) * gpb);
      kt_use(res[r], kR, nrt, acc3, acc2);
    }
  }
  for (; b + 2 < nb; b += 2) {
    kt_use(bufA, kA, nrt, acc3, acc2);
    kt_load(bufA, p, b + 2);
    kt_readk<P>(kA, ka, (b + 2) * gpb);
    kt_use(bufB, kB, nrt, acc3, acc2);
    kt_load(bufB, p, b + 3);
    kt_readk<P>(kB, ka, (b + 3) * gpb);
  }
  kt_use(bufA, kA, nrt, acc3, acc2);
  kt_use(bufB, kB, nrt, acc3, acc2);
}
// how many of a wave's nb register buffers stay resident: at most RL_NRES, leaving an even number >= 2 to stream (the double-buffered
// loop then needs no test around its reloads)
__device__ __forceinline__ int kt_resident_count(int nb, int maxres) {
  int nres = imin(maxres, nb - 2);
  if (nres < 0) nres = 0;
  if ((nb - nres) & 1) nres += (nb - nres >= 3) ? 1 : -1;  // (the register array has RL_NRES + 1 slots; nb >= 2 always)
  return nres;
}
// the tail of phase V: v is complete in this wave, so it forms the two phase-J weights of its rows on the spot,
//   W[j][2p] = kse_j alpha_j,  W[j][2p+1] = kse_j v_j     (D lane = 16 i + 4 blk + p: row 16 rt + 4 blk + i, particle p)
// Polynomial kernels (MAXDEG > 0): phase K left the record  W[j][p NW + .] = [kse alpha (final) | kse | B | A]  (degree 1: the first two) and the
// total k in kb; the tail turns it into  [kse alpha | kse v | v B | v A | v | k v]  (degree 1: [kse alpha | kse v | v | k v]).
template <int P, int MAXDEG>
__device__ __forceinline__ void kt_tail(double (&acc3)[2][3], double (&acc2)[2][2], int rt0, int nrt, const double* kb, const double* al_l,
                                        double* vb, int lane) {
  const int p = lane & 3;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    if (r < nrt && p < P) {
      const int row = 16 * (rt0 + r) + 4 * ((lane >> 2) & 3) + (lane >> 4);
      double v = acc3[0][r] + acc3[1][r];
      if (r < 2) v += acc2[0][r] + acc2[1][r];  // (the wiring this wave did not take left its accumulators at zero)
      if (MAXDEG == 0) {
        const double kse = kb[row * P + p], alj = al_l[row];
        v2d w;
        w.x = kse * alj;
        w.y = kse * v;
        *reinterpret_cast<v2d*>(__builtin_assume_aligned(vb + 2 * (row * P + p), 16)) = w;
      } else {
        double* w = vb + (row * P + p) * lat_nw(MAXDEG);  // (records of 4 or 6 doubles: 16-byte aligned)
        const double kt = kb[row * P + p], kse = w[1];
        v2d vk;
        vk.x = v;
        vk.y = kt * v;
        if (MAXDEG >= 2) {
          v2d ba = *reinterpret_cast<const v2d*>(__builtin_assume_aligned(w + 2, 16));
          ba.x *= v;
          ba.y *= v;
          w[1] = kse * v;
          *reinterpret_cast<v2d*>(__builtin_assume_aligned(w + 2, 16)) = ba;
          *reinterpret_cast<v2d*>(__builtin_assume_aligned(w + 4, 16)) = vk;
        } else {
          w[1] = kse * v;
          *reinterpret_cast<v2d*>(__builtin_assume_aligned(w + 2, 16)) = vk;
        }
      }
    }
  }
}

// sum over the 16 lanes of a DPP row; the total lands in lane 15 of the row (shifted-in lanes read zero)
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_take<0x111, 0xf>(v);  // row_shr:1
  v += dpp_take<0x112, 0xf>(v);  // row_shr:2
  v += dpp_take<0x114, 0xf>(v);  // row_shr:4
  v += dpp_take<0x118, 0xf>(v);  // row_shr:8
  return v;
}

template <int P>
struct LatLog2 {
  static constexpr int v = P == 4 ? 2 : (P == 2 ? 1 : 0);
};

// KR = phase-K items per thread: Npad * P <= KR * RF_NT
// phase stamps of the lean kernel: accumulated in LDS (no global round trip inside the step), written out once at the end
#define RL_STAMP(k)                           \
  do {                                        \
    if (stamping && tid == 0) {               \
      unsigned long long now_ = clock64();    \
      stl[k] += now_ - last_stamp;            \
      last_stamp = now_;                      \
    }                                         \
  } while (0)
#define RL_SUB(k)                             \
  do {                                        \
    if (stamping && lane == 0) {              \
      unsigned long long now_ = clock64();    \
      stl[k] += now_ - sub_stamp;             \
      sub_stamp = now_;                       \
    }                                         \
  } while (0)
// phase J of the lean kernel on the 4x4x4 MFMA:  R[c][n] = sum_j Xe[c][j] W[j][n],  Xe = [X^T; 1; 0] (8 rows),  W (N x 2P, 8 columns
// at most).  One instruction = blocks (row half, column half) x 4 training points.  A wave sums over the rows it has just
// produced in phase V (its own row tiles: 32 or 48 training points) -- no workgroup barrier between the two phases -- and stores
// its partial tile to red[wave][8][8], ONE unconditional store per lane.
template <int P>
__device__ __forceinline__ void lean_j(const double* xe, const double* vb, int Npad, int j0, int nu, int lane, double& acc0, double& acc1) {
  const int kq = lane >> 4, blk = (lane >> 2) & 3, e = lane & 3;
  const double* ap = xe + (4 * (blk >> 1) + e) * Npad + j0 + kq;
  const double* bp = vb + (j0 + kq) * (2 * P) + 4 * (blk & 1) + e;  // (P < 4: columns >= 2P read a neighbour's weights; those output columns are never used)
  for (int u0 = 0; u0 < nu; u0 += 4) {  // nu = 8 or 12
    double av[4], bw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      av[u] = ap[4 * (u0 + u)];
      bw[u] = bp[4 * (u0 + u) * (2 * P)];
    }
    asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(bw[0]), "+v"(bw[1]), "+v"(bw[2]), "+v"(bw[3]));
    mfma4(acc0, av[0], bw[0]);
    mfma4(acc1, av[1], bw[1]);
    mfma4(acc0, av[2], bw[2]);
    mfma4(acc1, av[3], bw[3]);
  }
}
// D lane = 16 i + 4 blk + j: row 4 (blk >> 1) + i, column 4 (blk & 1) + j
__device__ __forceinline__ int lean_j_slot(int lane) { return (4 * ((lane >> 3) & 1) + (lane >> 4)) * 8 + 4 * ((lane >> 2) & 1) + (lane & 3); }

// The same with NCG > 1 column groups of 8 (polynomial kernels: NW P = 16 or 24 columns at P = 4): the A operand of a step is shared by
// the groups, each group keeps its own accumulator pair; partial tiles to red[wave][group][8][8].  W is [j][NCOL] (columns p NW + s);
// the last group may reach up to 7 columns past NCOL -- into the next row, finite values whose output columns nobody reads.
template <int NCOL, int NCG>
__device__ __forceinline__ void lean_j_groups(const double* xe, const double* vb, int Npad, int j0, int nu, int lane, double (&acc)[NCG][2]) {
  const int kq = lane >> 4, blk = (lane >> 2) & 3, e = lane & 3;
  const double* ap = xe + (4 * (blk >> 1) + e) * Npad + j0 + kq;
  const double* bp = vb + (j0 + kq) * NCOL + 4 * (blk & 1) + e;
  for (int u0 = 0; u0 < nu; u0 += 4) {  // nu = 8 or 12
    double av[4], bw[NCG][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      av[u] = ap[4 * (u0 + u)];
#pragma unroll
      for (int g = 0; g < NCG; ++g) bw[g][u] = bp[4 * (u0 + u) * NCOL + 8 * g];
    }
    asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]));
#pragma unroll
    for (int g = 0; g < NCG; ++g) asm volatile("" : "+v"(bw[g][0]), "+v"(bw[g][1]), "+v"(bw[g][2]), "+v"(bw[g][3]));
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int g = 0; g < NCG; ++g) mfma4(acc[g][u & 1], av[u], bw[g][u]);
  }
}

// The z-only part of a polynomial kernel's posterior (MAXDEG > 0), per particle and step, by ONE otherwise idle wave between u and
// the end of phase K -- off wave 0's serial section.  With zw = w (*) z:
//   qB_c = sum_e w21_e z_e AXX[c][e] (= sum_j alpha_j B_j X_jc),  qA_c likewise with w20          -> fz[P*4 + (p*8 + c)*2 + {0, 1}]
//   mpoly = w1_D sum alpha + sum_d w1_d z_d aX_d + sum_d w20_d z_d qB_d   (= sum_j alpha_j (p1_j + A_j B_j))
//   kzz   = lambda + w1_D + sum_d w1_d z_d^2 + Sa Sb,   Sa = sum_d w20_d z_d^2,  Sb = sum_d w21_d z_d^2        -> fz[p*4 + {0,1,2,3}]
// Lane (p, c), c < 8 (P * 8 <= 32 lanes); the sums over c by one DPP reduction inside the particle's group of 8 lanes.
template <int P, int MAXDEG>
__device__ __forceinline__ void lean_prefz(const double* z, const double* kpar, const double* pc, double* fz, double lambda, int D, int lane) {
  const int p = (lane >> 3) & 3, c = lane & 7;
  const bool act = lane < P * 8 && c < D;
  const int cc = c < D ? c : 0, pp = p < P ? p : 0;
  double qB = 0.0, qA = 0.0;
  const double zc = z[pp * D + cc];
  const double w1c = kpar[KP_W1(D) + cc], w20c = MAXDEG >= 2 ? kpar[KP_W20(D) + cc] : 0.0, w21c = MAXDEG >= 2 ? kpar[KP_W21(D) + cc] : 0.0;
  if (MAXDEG >= 2) {
    double ze[RL_MAXD], a20[RL_MAXD], a21[RL_MAXD], ax[RL_MAXD];
#pragma unroll
    for (int e = 0; e < RL_MAXD; ++e) {
      const int ee = e < D ? e : 0;
      ze[e] = z[pp * D + ee];
      a20[e] = kpar[KP_W20(D) + ee];
      a21[e] = kpar[KP_W21(D) + ee];
      ax[e] = pc[28 + cc * 8 + e];  // AXX[c][e] (zero for e >= D)
    }
#pragma unroll
    for (int e = 0; e < RL_MAXD; ++e) {
      qB = fma(a21[e] * ze[e], ax[e], qB);
      qA = fma(a20[e] * ze[e], ax[e], qA);
    }
  }
  // per-(p, c) terms, then the sums over c of one particle: lanes p*8 .. p*8+7 (one aligned group of 8 lanes inside a DPP row)
  double t[5];
  t[0] = act ? fma(w20c * zc, qB, (w1c * zc) * kpar[KP_AX(D) + cc]) : 0.0;  // -> mpoly
  t[1] = act ? (w1c * zc) * zc : 0.0;                                        // -> p1(z, z)
  t[2] = act ? (w20c * zc) * zc : 0.0;                                       // -> Sa
  t[3] = act ? (w21c * zc) * zc : 0.0;                                       // -> Sb
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // sum over the group's 8 lanes: lane ^ 1, lane ^ 2 (quad_perm), then lane + 4 (row_shl:4): the lanes c < 4 hold the total
    t[i] += dpp_take<0xB1, 0xf>(t[i]);
    t[i] += dpp_take<0x4E, 0xf>(t[i]);
    t[i] += dpp_take<0x104, 0xf>(t[i]);
  }
  if (act && MAXDEG >= 2) {
    fz[P * 4 + (p * 8 + c) * 2] = qB;
    fz[P * 4 + (p * 8 + c) * 2 + 1] = qA;
  }
  if (lane < P * 8 && c == 0) {
    const double w1D = pc[24], sal = pc[25];
    fz[p * 4 + 0] = fma(w1D, sal, t[0]);
    fz[p * 4 + 1] = (lambda + (w1D + t[1])) + t[2] * t[3];
    fz[p * 4 + 2] = t[2];
    fz[p * 4 + 3] = t[3];
  }
}

template <int P, int KR, int MAXDEG, bool PMS>
__global__ __launch_bounds__(RF_NT) void rollout_fwd_lat_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int LP = LatLog2<P>::v;
  // two segments of Kinv per wave only beyond 24 row tiles (Npad > 384), i.e. in the instantiations with more phase-K items per thread than
  // the base ones (lean_items_per_thread): the base instantiations carry no second stream pointer, no segment loop
  constexpr bool SEG2 = KR > (P == 4 ? 3 : (P == 2 ? 2 : 1));
  constexpr int NWC = lat_nw(MAXDEG), NCOL = P * NWC, NCG = lat_ncg(P, MAXDEG);  // phase-J weight columns: per particle, in all, groups of 8
  const mcp_model& md = a.model;
  const mcp_policy& pl = a.pol;
  const mcp_noise nzl = noise_of_launch(a.nz);
  const int tid0 = threadIdx.x;
  const int S = md.S, U = md.U, G = md.G, D = md.D, B = pl.B, PF = pl.P, M = a.M, T = a.T;
  const int DS = D - U;
  const int ROW1 = lat_ones_row(DS, U);  // the row of xq that carries ones (phase J's unweighted sums)
  const int Npad = a.NpadMax;
  const int NG = lat_ng(B), NGP = lat_ngp(B), Bp = NG * 16, BQ = (B + 3) >> 2;
  const LatLayout L = lat_layout(P, B, Npad, MAXDEG);
  double* pol = smem + LatFixed::pol;
  double* umax_l = pol + RL_PFM;
  double* bias_l = umax_l + MCP_MAX_INPUT;
  double* iumax_l = bias_l + MCP_MAX_INPUT;
  double* xs = smem + LatFixed::xs;  // [2][P][S] double-buffered
  double* z = smem + LatFixed::z;
  double* zs = smem + LatFixed::zs;
  double* sf = smem + LatFixed::sf;
  double* dl = smem + LatFixed::dl;
  double* epsb = smem + LatFixed::eps;
  double* red = smem + LatFixed::red;
  double* rtot = smem + LatFixed::rt;
  double* pc = smem + LatFixed::pc;    // polynomial constants (MAXDEG > 0)
  double* fz = smem + LatFixed::fz;    // z-only polynomial terms of the step (MAXDEG > 0)
  double* pnz = smem + LatFixed::pn;   // position measurement noise (PMS)
  GpL* gpl = reinterpret_cast<GpL*>(smem + LatFixed::gpl);
  double* kpar = smem + LatFixed::kpar;
  int* role = reinterpret_cast<int*>(smem + LatFixed::role);
  unsigned long long* stl = reinterpret_cast<unsigned long long*>(smem + LatFixed::stl);  // phase-cycle totals (diagnostic)
  const bool stamping = a.stamps && blockIdx.x == a.stamp_block;
  double* gs = smem + L.gs;
  double* kb = smem + L.kb;
  double* vb = smem + L.vb;
  double* xq = smem + L.xq;
  double* al_l = smem + L.al;
  double* cen = smem + L.cen;
  double* wgt = smem + L.wgt;
  int* mk = reinterpret_cast<int*>(smem + L.mk);
  int cluster, myg;
  {
    const int b = blockIdx.x, grp = b / (8 * G), r = b - grp * 8 * G;
    cluster = grp * 8 + (r & 7);
    myg = r >> 3;
    if (cluster >= a.nclusters) return;  // padding blocks of the last group of 8 clusters
  }
  const int gcluster = a.m_off / P + cluster;  // cluster index in the whole swarm (hand-off slots)
  const bool writer = myg == 0;                // states / inputs are identical in the workgroups of a cluster: one of them stores
  int* abortw = reinterpret_cast<int*>(dl + P * G);
  if (tid0 == 0) *abortw = 0;
  if (tid0 < 24) stl[tid0] = 0;
  const int m0 = a.m_off + cluster * P;
  const int Mend = a.m_off + a.m_cnt;
  uint32_t bad = 0;
  const bool drop = pl.p_drop > 0.0;
  const double keep_scale = 1.0 / (1.0 - pl.p_drop);
  const uint32_t drop_thr = drop_threshold(pl.p_drop);
  const int nna = md.n_not_angle, na = md.n_angle;
  const int pol_nna = pl.n_non_angle, pol_na = pl.n_angle;
  const double Ts = md.Ts;

  // ---- one-time staging ------------------------------------------------------------------
  for (int it = tid0; it < PF; it += RF_NT) pol[it] = exp(-pl.log_ls[it]);
  if (tid0 < U) {
    const double um = pl.u_max[tid0];
    umax_l[tid0] = um;
    iumax_l[tid0] = 1.0 / um;
    bias_l[tid0] = pl.bias ? pl.bias[tid0] : 0.0;
  }
  const mcp_gp* gps_l = md.gp + myg;
  stage_gp_tables(gps_l, md.var_scale + myg, 1, D, gpl, kpar, tid0);
  {
    const mcp_gp& gp = gps_l[0];
    for (int it = tid0; it < Npad; it += RF_NT) al_l[it] = it < gp.Npad ? gp.alpha[it] : 0.0;
    for (int it = tid0; it < Npad * NCOL + 8; it += RF_NT) vb[it] = 0.0;
  }
  for (int it = tid0; it < RL_UM * Bp; it += RF_NT) {
    const int k = it / Bp, b = it - k * Bp;
    wgt[it] = (k < U && b < B) ? pl.weight[(size_t)k * B + b] : 0.0;
  }
  for (int it = tid0; it < RL_UM * P * NGP; it += RF_NT) gs[it] = 0.0;  // (the padding groups stay zero)
  for (int it = tid0; it < KT_ZROWS * P; it += RF_NT) kb[gps_l[0].Npad * P + it] = 0.0;  // (zero rows behind THIS GP's k: phase V's out-of-range operands)
  for (int it = tid0; it < P * RL_ZD; it += RF_NT) zs[it] = (it % RL_ZD) == ROW1 ? 1.0 : 0.0;  // (the ones row meets z = 1: no distance term)
  for (int it = tid0; it < P * RL_PFM; it += RF_NT) sf[it] = 0.0;
  // (before the barrier below: wave 1 writes step 1's position noise into this table in draw_step(0), behind it -- zeroed any later, a slow
  //  wave 0 could wipe what wave 1 had just drawn)
  if (PMS)
    for (int it = tid0; it < 2 * 32; it += RF_NT) pnz[it] = 0.0;
  // roles of the threads of wave 0 in the serial section (read back from LDS every step: values derived from the thread id would
  // otherwise be hoisted out of the time loop and held in registers across every phase).  Thread (p, s) = p * S + s owns state
  // component s of particle p; thread (p, c) = p * (D + 1) + c evaluates column c of phase F for particle p.
  double xn = 0.0;  // the owned state component (threads < P * S)
  if (tid0 < 64) {
    const bool own = tid0 < P * S;
    const int op = own ? tid0 / S : 0, os = own ? tid0 - op * S : 0;
    int zi_plain = -1, zi_ang = -1, pi_plain = -1, pi_ang = -1, g_vel = -1, g_pos = -1, vel_of_pos = 0;
    if (own) {
      for (int i = 0; i < nna; ++i)
        if (md.not_angle[i] == os) zi_plain = i;
      for (int i = 0; i < na; ++i)
        if (md.angle[i] == os) zi_ang = i;
      if (pl.kind == MCP_POLICY_ANGLES) {
        for (int i = 0; i < pol_nna; ++i)
          if (pl.non_angle[i] == os) pi_plain = i;
        for (int i = 0; i < pol_na; ++i)
          if (pl.angle[i] == os) pi_ang = i;
      }
      for (int g = 0; g < G; ++g) {
        if (md.vel[g] == os) g_vel = g;
        if (md.not_vel[g] == os) {
          g_pos = g;
          vel_of_pos = md.vel[g];
        }
      }
      xn = a.x0[(size_t)imin(m0 + op, Mend - 1) * S + os];
    }
    int* ro = role + tid0 * 16;
    ro[RO_OP] = op;
    ro[RO_OS] = os;
    ro[RO_ZPLAIN] = zi_plain;
    ro[RO_ZANG] = zi_ang;
    ro[RO_PPLAIN] = pi_plain;
    ro[RO_PANG] = pi_ang;
    ro[RO_GVEL] = g_vel;
    ro[RO_GPOS] = g_pos;
    ro[RO_VELOFPOS] = vel_of_pos;
    // measurement model (PMS): thread (p, s) produces the measurement of its own component; a velocity thread rebuilds the noisy position
    // of its pair from the pair's lane (same noise value) -- rollout_fwd.hip, phase S
    int pm_pos = -1, pm_vel = -1, pm_pair = tid0;
    if (PMS && own) {
      for (int i = 0; i < pl.meas.n; ++i) {
        if (pl.meas.pos[i] == os) pm_pos = i;
        if (pl.meas.vel[i] == os) {
          pm_vel = i;
          pm_pair = op * S + pl.meas.pos[i];
        }
      }
    }
    ro[RO_PMPOS] = pm_pos;
    ro[RO_PMVEL] = pm_vel;
    ro[RO_PMPAIR] = pm_pair;
    ro[RO_FP] = tid0 / (D + 1);
    ro[RO_FC] = tid0 % (D + 1);
    ro[RO_OM] = imin(m0 + op, Mend - 1);
    {  // row of xq that belongs to column c of phase F (c == D: the ones row)
      const int c = tid0 % (D + 1);
      ro[RO_FROW] = c < D ? lat_row_of(c, DS) : ROW1;
    }
  }
  lds_barrier();
  // phase S as a table: thread (p, s) writes x, sin x or cos x to at most two GP-input slots (raw and divided by the lengthscale) and two
  // policy-feature slots; the slot addresses and scale factors are read in ONE batch at the top of the phase and every write is
  // unconditional (unused ones go to a dump word) -- as tests on the role indices each slot was its own LDS read -> multiply -> write
  // round trip behind a branch: 7 dependent round trips per step on the wave that bounds the step
  if (tid0 < P * S) {
    const int* ro = role + tid0 * 16;
    const int op = ro[RO_OP], os = ro[RO_OS], zi_plain = ro[RO_ZPLAIN], zi_ang = ro[RO_ZANG], pi_plain = ro[RO_PPLAIN], pi_ang = ro[RO_PANG];
    int* si = reinterpret_cast<int*>(smem + LatFixed::sro + tid0 * 8);
    double* sd = smem + LatFixed::sro + tid0 * 8 + 4;
    const double* il = kpar + KP_INVLS(D);
    const int zA = zi_ang >= 0 ? nna + zi_ang : zi_plain, zB = zi_ang >= 0 ? nna + na + zi_ang : -1;
    int pA, pB = -1;
    if (pl.kind == MCP_POLICY_ANGLES) {
      pA = pi_ang >= 0 ? pol_nna + pi_ang : pi_plain;
      pB = pi_ang >= 0 ? pol_nna + pol_na + pi_ang : -1;
    } else {
      pA = os;
    }
    si[0] = zA >= 0 ? LatFixed::z + op * D + zA : LatFixed::dump;
    si[1] = zB >= 0 ? LatFixed::z + op * D + zB : LatFixed::dump;
    si[2] = zA >= 0 ? LatFixed::zs + op * RL_ZD + zA : LatFixed::dump;
    si[3] = zB >= 0 ? LatFixed::zs + op * RL_ZD + zB : LatFixed::dump;
    si[4] = pA >= 0 ? LatFixed::sf + op * RL_PFM + pA : LatFixed::dump;
    si[5] = pB >= 0 ? LatFixed::sf + op * RL_PFM + pB : LatFixed::dump;
    si[6] = si[7] = 0;
    if (PMS) {  // (the spare double of the row: std of this component's pair's position noise)
      const int pi = ro[RO_PMPOS] >= 0 ? ro[RO_PMPOS] : ro[RO_PMVEL];
      double std = 0.0;
      for (int i = 0; i < pl.meas.n; ++i)
        if (i == pi) std = pl.meas.std_pos[i];  // (uniform index into the by-value argument)
      sd[-1] = std;
    }
    sd[0] = zA >= 0 ? il[zA] : 0.0;
    sd[1] = zB >= 0 ? il[zB] : 0.0;
    sd[2] = pA >= 0 ? pol[pA] : 0.0;
    sd[3] = pB >= 0 ? pol[pB] : 0.0;
  }
  for (int it = tid0; it < RL_PFM * Bp; it += RF_NT) {
    const int q = it / Bp, b = it - q * Bp;
    cen[it] = (q < PF && b < B) ? pl.centers[(size_t)b * PF + q] * pol[q] : 0.0;
  }
  {
    const mcp_gp& gp = gps_l[0];
    for (int it = tid0; it < RL_ZD * Npad; it += RF_NT) {
      const int r = it / Npad, j = it - r * Npad;
      const int d = r < RL_DSM ? (r < DS ? r : -1) : (r - RL_DSM < U ? DS + r - RL_DSM : -1);
      xq[it] = (d >= 0) ? (j < gp.Npad ? gp.Xt[(size_t)d * gp.Npad + j] * kpar[KP_INVLS(D) + d] : 0.0) : (r == ROW1 ? 1.0 : 0.0);
    }
  }
  if (tid0 < RL_MAXD) pc[LAT_PC_LS + tid0] = tid0 < D ? 1.0 / kpar[KP_INVLS(D) + tid0] : 0.0;
  if (MAXDEG > 0) {
    lds_barrier();  // (xq and the lengthscales, read below)
    // launch constants of the polynomial terms.  The bilinear forms are evaluated on SCALED operands (z_d / l_d)(X_jd / l_d), the tables
    // the SE distance already reads: their weights carry l_d^2.  Rows of xq: r < RL_DSM a state-derived dimension, RL_DSM + k an input.
    if (tid0 < RL_ZD) {
      const int r = tid0, d = r < RL_DSM ? (r < DS ? r : -1) : (r - RL_DSM < U ? DS + r - RL_DSM : -1);
      const int dd = d >= 0 ? d : 0;
      const double il = kpar[KP_INVLS(D) + dd], l2 = 1.0 / (il * il);
      pc[r] = d >= 0 ? kpar[KP_W1(D) + dd] * l2 : 0.0;
      pc[8 + r] = (d >= 0 && MAXDEG >= 2) ? kpar[KP_W20(D) + dd] * l2 : 0.0;
      pc[16 + r] = (d >= 0 && MAXDEG >= 2) ? kpar[KP_W21(D) + dd] * l2 : 0.0;
    }
    if (tid0 == 8) {
      pc[24] = kpar[KP_W1(D) + D];
      pc[26] = pc[27] = 0.0;
    }
    if (tid0 >= 64 && tid0 < 128) {  // sum_j alpha_j (wave 1)
      double sa = 0.0;
      for (int j = tid0 - 64; j < Npad; j += 64) sa += al_l[j];
      sa = wave_sum(sa);
      if (tid0 == 64) pc[25] = sa;
    }
    if (tid0 >= 128 && tid0 < 192) {  // AXX[c][e] = sum_j alpha_j X_jc X_je (rows / columns >= D: zero)
      const int c = (tid0 - 128) >> 3, e = tid0 & 7;
      double sx = 0.0;
      if (MAXDEG >= 2 && c < D && e < D) {
        const double* xc_ = xq + lat_row_of(c, DS) * Npad;
        const double* xe_ = xq + lat_row_of(e, DS) * Npad;
        for (int j = 0; j < Npad; ++j) sx = fma(al_l[j] * xc_[j], xe_[j], sx);
        sx *= pc[LAT_PC_LS + c] * pc[LAT_PC_LS + e];  // (xq holds X / l)
      }
      pc[28 + c * 8 + e] = sx;
    }
    for (int it = tid0; it < 4 * 4 + 4 * 8 * 2; it += RF_NT) fz[it] = 0.0;
  }
  int cur = 0;

  // the random numbers of step `ts`: process noise of this workgroup's GP by wave 1, dropout decisions (one Philox block per 4 basis
  // functions, bit-packed) by waves 2, 3, 5, 6, 7 -- never waves 0 and 4 (wave 0 runs the serial section meanwhile, wave 4 shares its SIMD)
  auto draw_step = [&](int ts, int wv, int lane) {
    if (wv == 1) {
      if (lane < P && ts < T - 1) {
        double ev = 0.0;
        if (a.particle_pred) {
          const int mm = imin(m0 + lane, Mend - 1);
          ev = nzl.eps ? nzl.eps[((size_t)ts * M + mm) * G + myg] : philox_normal(nzl, mm, ts, myg);
        }
        epsb[(ts & 1) * P + lane] = ev;
      }
      if (PMS) {
        // position measurement noise of step ts + 1 (phase S of that step reads it BEFORE the step's first barrier, so it is drawn one
        // step earlier than the process noise): lane P + (p, pair); same draw as the general kernel: (particle, step, pair, stream POS)
        const int e = lane - P, np = pl.meas.n;
        if (e >= 0 && e < P * np && ts + 1 < T) {
          const int pp = e / np, pi = e - pp * np, mm = imin(m0 + pp, Mend - 1);
          pnz[((ts + 1) & 1) * 32 + e] = pl.meas.pos_noise ? pl.meas.pos_noise[((size_t)ts * M + mm) * np + pi]
                                                          : philox_normal(nzl, mm, ts + 1, pi, MCP_STREAM_POS);
        }
      }
    } else if (wv >= 2 && wv != 4 && drop && !nzl.masks) {
      const int ti = (wv - 2 - (wv > 4 ? 1 : 0)) * 64 + lane;
      for (int it = ti; it < P * BQ; it += 5 * 64) {
        const int p = it / BQ, q = it - p * BQ;
        const u32x4 r = philox_draw(nzl, imin(m0 + p, Mend - 1), ts, MCP_STREAM_MASK, (uint32_t)q);
        mk[it] = (int)(r.x >= drop_thr) | ((int)(r.y >= drop_thr) << 1) | ((int)(r.z >= drop_thr) << 2) | ((int)(r.w >= drop_thr) << 3);
      }
    }
  };
  draw_step(0, __builtin_amdgcn_readfirstlane(tid0 >> 6), tid0 & 63);
  lds_barrier();  // tables (incl. the chunk table written by thread 0) visible to every wave
  // this wave's share of Kinv: the row tiles [vrt0, vrt0 + vnrt) of its GP, as MFMA operand tiles in streaming order; the first
  // RL_NRES register buffers of the stream stay in registers for the whole rollout
  const int vnpad = __builtin_amdgcn_readfirstlane(gpl[0].Npad), vnjg = vnpad >> 3;
  int vrt0, vnrt, vrt0b, vnrtb;  // first / second segment: row tiles [vrt0, vrt0 + vnrt), [vrt0b, vrt0b + vnrtb) (vnrtb = 0: none)
  {
    const int w = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    vrt0 = kt_vrt0(vnpad >> 4, w);
    vnrt = kt_vcount(vnpad >> 4, w);
    vrt0b = SEG2 ? kt_vrt0(vnpad >> 4, RF_NW + w) : 0;
    vnrtb = SEG2 ? kt_vcount(vnpad >> 4, RF_NW + w) : 0;
  }
  const gptr2_t vp = (gptr2_t)(a.kt + (size_t)myg * a.kt_stride + (size_t)vrt0 * vnjg * 128) + (tid0 & 63);
  const gptr2_t vpb = (gptr2_t)(a.kt + (size_t)myg * a.kt_stride + (size_t)vrt0b * vnjg * 128) + (tid0 & 63);
  const int vnt = vnrt * vnjg;  // tiles in this wave's (first) stream
  // (degree-2 polynomial kernels at 4 particles: one resident buffer fewer -- the two halves of phase K carry six values per item
  //  across the barrier in between, and with three resident buffers the allocator spilled one of them to scratch: its reload in
  //  phase V waits with vmcnt(0), i.e. for the whole stream in flight: +1-2 k cycles per wave and step)
  constexpr int NRES = (MAXDEG >= 2 && P == 4) ? (RL_NRES > 1 ? RL_NRES - RL_NRES_CUT2 : RL_NRES) : RL_NRES;
  v2d vres[NRES + 1][KT_NL];  // (+ 1: a wave whose buffer count has the other parity keeps one more or one fewer)
  // the second segment streams an even number of buffers as well: with an odd count its first buffer takes the register array's last slot
  // (the first segment then keeps at most NRES)
  int nres = 0;
  const int nresb = (SEG2 && vnrtb > 0 && (((vnrtb * vnjg + KT_NL - 1) / KT_NL) & 1)) ? 1 : 0;
  if (vnrt > 0) {
    nres = kt_resident_count((vnt + KT_NL - 1) / KT_NL, NRES);
    if (nresb && nres > NRES) nres -= 2;  // (the parity rule had taken the spare slot)
#pragma unroll
    for (int r = 0; r < NRES + 1; ++r) {
      if (r < nres) kt_load(vres[r], vp, r);
      if (r == NRES && nresb) kt_load(vres[r], vpb, 0);
    }
  }
  unsigned long long last_stamp = clock64(), sub_stamp = last_stamp;
  double pm_prev_np = 0.0, pm_prev_nv = 0.0, pm_prev_mv = 0.0;  // PMS: previous noisy position / noisy velocity / filtered velocity of this lane's pair

  for (int t = 0; t < T; ++t) {
    // the thread id goes through an opaque move once per step: what is derived from it is recomputed where it is used (a handful
    // of integer instructions) instead of being hoisted out of the time loop and kept in registers across phase V
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- phase S (wave 0): publish x_t and everything derived from a single state component ------
    if (wv == 0) {
      const int* ro = role + lane * 16;
      const int4 r0 = *reinterpret_cast<const int4*>(ro), r1 = *reinterpret_cast<const int4*>(ro + 4);
      const int op = r0.x, os = r0.y, zi_ang = r0.w, pi_ang = r1.y;
      const bool own = lane < P * S;
      const bool ovalid = own && (m0 + op < Mend);
      // what the policy sees of this component: the state itself, or (PMS, MC_PILCO.py:873-903) its simulated measurement
      double xm = xn;
      if (PMS) {
        const int4 r2 = *reinterpret_cast<const int4*>(ro + 8);
        const int pm_pos = r2.y, pm_vel = r2.z;
        const double xpair = __shfl(xn, r2.w);  // (all of wave 0 takes part; lanes without a pair read themselves)
        if (own) {
          const int pi = pm_pos >= 0 ? pm_pos : pm_vel;
          double npos = pm_pos >= 0 ? xn : xpair;
          if (pi >= 0 && t > 0) npos = fma(smem[LatFixed::sro + lane * 8 + 3], pnz[(t & 1) * 32 + op * pl.meas.n + pi], npos);
          if (pm_pos >= 0) xm = npos;
          if (pm_vel >= 0) {
            if (t == 0) {  // (at t = 0 the measurement is the true state, :856)
              pm_prev_nv = xn;
              pm_prev_mv = xn;
            } else {
              const double nv = (npos - pm_prev_np) / Ts;
              xm = (pl.meas.b0 * nv + pl.meas.b1 * pm_prev_nv - pl.meas.a1 * pm_prev_mv) / pl.meas.a0;
              pm_prev_nv = nv;
              pm_prev_mv = xm;
            }
            pm_prev_np = npos;
          }
        }
      }
      if (own) {
        const double* srow = smem + LatFixed::sro + lane * 8;
        const int4 sa = *reinterpret_cast<const int4*>(srow);
        const int2 sb = *reinterpret_cast<const int2*>(srow + 2);
        const v2d sc0 = *reinterpret_cast<const v2d*>(srow + 4), sc1 = *reinterpret_cast<const v2d*>(srow + 6);
        double* xc = xs + cur * P * S;
        xc[op * S + os] = xn;
        if (ovalid) {
          if (writer) {
            a.states[((size_t)t * M + m0 + op) * S + os] = xn;
            if (PMS) pl.meas.meas[((size_t)t * M + m0 + op) * S + os] = xm;
          }
          if (is_bad(xn) || (PMS && is_bad(xm))) bad |= MCP_STATUS_NAN;
        }
        double sn = 0.0, cs = 0.0;
        if (zi_ang >= 0 || pi_ang >= 0) sincos_fast(xn, &sn, &cs);
        double snm = sn, csm = cs;  // trig of the measured value (policy features)
        if (PMS && pi_ang >= 0 && xm != xn) sincos_fast(xm, &snm, &csm);
        // GP input z = [x[not_angle], sin x[angle], cos x[angle], u]   (Model_learning.py:670-683), raw and divided by its lengthscale;
        // policy features (Policy.py:326-333: [x_nonangle, COS, SIN]; plain policy: x) divided by theirs: slots and factors from the table
        const double vz = zi_ang >= 0 ? sn : xn, vp = pi_ang >= 0 ? csm : xm;
        smem[sa.x] = vz;
        smem[sa.y] = cs;
        smem[sa.z] = vz * sc0.x;
        smem[sa.w] = cs * sc0.y;
        smem[sb.x] = vp * sc1.x;
        smem[sb.y] = snm * sc1.y;
      }
    }
    if (wv == 0) RL_SUB(15);
    lds_barrier();  // B0
    if (*abortw) {  // uniform: a partner never arrived (set by wave 0 in the previous step's hand-off)
      bad |= MCP_STATUS_SYNC;
      break;
    }
    RL_STAMP(0);
    // ---- policy + state-only part of the GP distances --------------------------------------------
    const int pK = tid & (P - 1);  // particle of this thread's phase-K items (RF_NT % P == 0)
    double ds[KR], xin[KR][RL_UM];
    double p1s[MAXDEG >= 1 ? KR : 1], pAs[MAXDEG >= 2 ? KR : 1], pBs[MAXDEG >= 2 ? KR : 1];  // polynomial kernels: the three bilinear forms of an item
    {
      const int row = tid >> 4, e16 = tid & 15;
      const int pP = row & (P - 1);  // particle of this DPP row in the policy phase (32 rows per round, 32 % P == 0)
      double sfr[RL_PFM], zr[RL_DSM];
#pragma unroll
      for (int q = 0; q < RL_PFM; ++q) sfr[q] = sf[pP * RL_PFM + q];
#pragma unroll
      for (int d = 0; d < RL_DSM; ++d) zr[d] = zs[pK * RL_ZD + d];
      auto k_state = [&]() {
        if (t < T - 1) {
#pragma unroll
          for (int r = 0; r < KR; ++r) {
            // (the last round holds N P - (KR - 1) 512 items: at N = 300, P = 4 three of the eight waves; the others skip it -- wave-uniform)
            if (r == KR - 1 && r > 0 && ((wv * 64 + r * RF_NT) >> LP) >= Npad) {
              ds[r] = 0.0;
              if (MAXDEG < 2) {
#pragma unroll
                for (int k = 0; k < RL_UM; ++k) xin[r][k] = 0.0;
              }
              if (MAXDEG >= 1) p1s[r] = 0.0;
              if (MAXDEG >= 2) pAs[r] = pBs[r] = 0.0;
              continue;
            }
            const int j = imin((tid + r * RF_NT) >> LP, Npad - 1);
            double xv[RL_DSM];
#pragma unroll
            for (int d = 0; d < RL_DSM; ++d) xv[d] = xq[d * Npad + j];
            if (MAXDEG < 2) {
#pragma unroll
              for (int k = 0; k < RL_UM; ++k) xin[r][k] = xq[(RL_DSM + k) * Npad + j];
            }
            double acc = 0.0;
#pragma unroll
            for (int d = 0; d < RL_DSM; ++d) {
              const double rr = zr[d] - xv[d];
              acc = fma(rr, rr, acc);
            }
            ds[r] = acc;
            if (MAXDEG >= 1) {
              // p1 = w1_D + sum_d w1_d z_d X_jd,  A = sum_d w20_d z_d X_jd,  B = sum_d w21_d z_d X_jd  on the scaled operands: one product
              // per dimension feeds the three forms (weights w l^2 from the constants table; padded dimensions carry weight zero)
              // (a padded dimension costs four instructions here, not two as in the distance: the last two state dimensions and the
              //  second input sit behind a uniform test)
              double q1 = pc[24], qa = 0.0, qb = 0.0;
#pragma unroll
              for (int d = 0; d < RL_DSM; ++d) {
                if (d < RL_DSM - 2 || d < DS) {
                  const double zx = zr[d] * xv[d];
                  q1 = fma(pc[d], zx, q1);
                  if (MAXDEG >= 2) {
                    qa = fma(pc[8 + d], zx, qa);
                    qb = fma(pc[16 + d], zx, qb);
                  }
                }
              }
              p1s[r] = q1;
              if (MAXDEG >= 2) {
                pAs[r] = qa;
                pBs[r] = qb;
              }
            }
          }
        }
      };
      auto policy_pass = [&]() {
        const int NPR = NG << LP;  // (basis group, particle) pairs, particle fastest
        for (int pr0 = 0; pr0 < NPR; pr0 += 32) {
          const int pr = pr0 + row;
          const int g = pr >> LP;
          const int b = imin(g, NG - 1) * 16 + e16;
          double cv[RL_PFM], wv_[RL_UM];
#pragma unroll
          for (int q = 0; q < RL_PFM; ++q) cv[q] = cen[q * Bp + b];
#pragma unroll
          for (int k = 0; k < RL_UM; ++k) wv_[k] = wgt[k * Bp + b];
          int kbits = 0;
          if (drop) {
            if (nzl.masks)
              kbits = (b < B && nzl.masks[((size_t)t * M + imin(m0 + pP, Mend - 1)) * B + b] != 0) ? 1 : 0;
            else
              kbits = (mk[pP * BQ + imin(b >> 2, BQ - 1)] >> (b & 3)) & 1;  // (b >= B: weight 0, whatever the bit)
          }
          double dist = 0.0;
#pragma unroll
          for (int q = 0; q < RL_PFM; ++q) {
            const double rr = sfr[q] - cv[q];
            dist = fma(rr, rr, dist);
          }
          double phi = exp(-dist);
          if (drop) phi = kbits ? phi * keep_scale : 0.0;
#pragma unroll
          for (int k = 0; k < RL_UM; ++k) {
            if (k < U) {  // uniform
              const double sgrp = row16_sum(wv_[k] * phi);
              if (e16 == 15 && pr < NPR) gs[(k * P + pP) * NGP + g] = sgrp;
            }
          }
        }
      };
      // (polynomial kernels: the policy pass first -- the six values per item that phase K carries to its second half are then not live
      //  beside the policy pass's operands; in the other order the degree-2 instantiation spilled a resident Kinv buffer to scratch)
      if (MAXDEG >= 2) {
        policy_pass();
        k_state();
      } else {
        k_state();
        policy_pass();
      }
    }
    lds_barrier();  // B1
    RL_STAMP(1);
    // ---- u = u_max tanh((W phi + b) / u_max): every thread adds the group sums of its own particle in the same fixed order ----
    double ur[RL_UM];
#pragma unroll
    for (int k = 0; k < RL_UM; ++k) {
      ur[k] = (RL_DSM + k == ROW1) ? 1.0 : 0.0;  // (a padded input row that carries the ones: z = 1 there)
      if (k < U) {  // uniform
        const double* gk = gs + (k * P + pK) * NGP;
        double s = 0.0;
#pragma unroll 1
        for (int g0 = 0; g0 < NGP; g0 += 8) {
          const v2d a0 = *reinterpret_cast<const v2d*>(__builtin_assume_aligned(gk + g0, 16));
          const v2d a1 = *reinterpret_cast<const v2d*>(__builtin_assume_aligned(gk + g0 + 2, 16));
          const v2d a2 = *reinterpret_cast<const v2d*>(__builtin_assume_aligned(gk + g0 + 4, 16));
          const v2d a3 = *reinterpret_cast<const v2d*>(__builtin_assume_aligned(gk + g0 + 6, 16));
          s += ((a0.x + a0.y) + (a1.x + a1.y)) + ((a2.x + a2.y) + (a3.x + a3.y));
        }
        s += bias_l[k];
        const double um = umax_l[k];
        const double u = pl.squash ? um * fast_tanh(s * iumax_l[k]) : s;
        ur[k] = u * kpar[KP_INVLS(D) + DS + k];
        if (tid < P) {  // thread p publishes the raw input of particle p (phase F's Jacobians, the inputs array)
          z[tid * D + DS + k] = u;
          if (m0 + tid < Mend) {
            if (t == T - 1 && writer) a.inputs[((size_t)t * M + m0 + tid) * U + k] = u;
            if (is_bad(u)) bad |= MCP_STATUS_NAN;
          }
        }
      }
    }
    if (t == T - 1) break;
    // ---- phase V, first half: issue the head of this wave's Kinv stream (independent of k) ----
    v2d bufA[KT_NL], bufB[KT_NL];
    if (RL_PRE > 0 && vnrt > 0) {
      kt_load(bufA, vp, nres);
      if (RL_PRE > 1) kt_load(bufB, vp, nres + 1);
    }
    // ---- phase K, second half: the input dimensions and the exp ----
    {
      const int Nown = gpl[0].N;
      const double lambda = gpl[0].lambda;
#pragma unroll
      for (int r = 0; r < KR; ++r) {
        if (r == KR - 1 && r > 0 && ((wv * 64 + r * RF_NT) >> LP) >= Npad) continue;  // (no item of this wave in the last round)
        const int it = tid + r * RF_NT;
        const int j = it >> LP;
        double dd = ds[r];
        if (MAXDEG >= 2) {  // (re-read, not carried across the barrier: registers -- see the resident buffers of phase V)
#pragma unroll
          for (int k = 0; k < RL_UM; ++k) xin[r][k] = xq[(RL_DSM + k) * Npad + imin(j, Npad - 1)];
        }
#pragma unroll
        for (int k = 0; k < RL_UM; ++k) {
          const double rr = ur[k] - xin[r][k];
          dd = fma(rr, rr, dd);
        }
        const double kv = j < Nown ? lambda * exp(-dd) : 0.0;
        if (MAXDEG == 0) {
          if (j < Npad) kb[it] = kv;
        } else {
          // k = kse + p1 + A B (Sparse_GP.py:625-646, GP_prior.py:314-335); the slots of (j, p)'s phase-J record that do not need v: kse alpha
          // (final), kse, B, A (multiplied by v in the tail of phase V).  Rows j >= N carry zeros in every slot.
          double q1 = p1s[r], qa = 0.0, qb = 0.0;
          if (MAXDEG >= 2) {
            qa = pAs[r];
            qb = pBs[r];
          }
#pragma unroll
          for (int k = 0; k < RL_UM; ++k) {
            if (k == 0 || k < U) {
              const double zx = ur[k] * xin[r][k];
              q1 = fma(pc[RL_DSM + k], zx, q1);
              if (MAXDEG >= 2) {
                qa = fma(pc[8 + RL_DSM + k], zx, qa);
                qb = fma(pc[16 + RL_DSM + k], zx, qb);
              }
            }
          }
          // (rows N <= j < Npad need no masking beyond kse: their alpha and their v are zero -- Kinv and alpha are zero padded -- so every
          //  phase-J weight of such a row vanishes whatever its k, B, A; and k_j itself only meets zero columns of Kinv in phase V)
          double kt = kv + q1;
          if (MAXDEG >= 2) kt = fma(qa, qb, kt);
          if (j < Npad) {
            kb[it] = kt;
            double* w = vb + it * NWC;  // the record of (j, p): it = j P + p
#ifndef RLX_NOWREC  // (experiment switch: timing without the phase-J records of phase K)
            v2d w01;
            w01.x = kv * al_l[j];
            w01.y = kv;
            *reinterpret_cast<v2d*>(__builtin_assume_aligned(w, 16)) = w01;
            if (MAXDEG >= 2) {
              v2d w23;
              w23.x = qb;
              w23.y = qa;
              *reinterpret_cast<v2d*>(__builtin_assume_aligned(w + 2, 16)) = w23;
            }
#endif
          }
        }
      }
    }
    lds_barrier();  // B2
    RL_STAMP(3);
    // ---- phase V, second half: v = Kinv k on the 4x4x4 MFMA, then the phase-J weights of this wave's rows ----
    const unsigned long long tv0_ = stamping ? clock64() : 0;
#ifdef RLX_PRIO  // experiment: issue priority for the waves that end the phase (1: the younger wave of every SIMD, 2: the waves with three row tiles)
    if (RLX_PRIO == 1 ? wv >= 4 : vnrt >= 3) __builtin_amdgcn_s_setprio(2);
#endif
    if (vnrt > 0) {
      double jacc[NCG][2];  // this wave's partial tile of phase J (both segments)
#pragma unroll
      for (int g = 0; g < NCG; ++g) jacc[g][0] = jacc[g][1] = 0.0;
      const int nseg = (SEG2 && vnrtb > 0) ? 2 : 1;  // wave-uniform
      for (int sg = 0; sg < nseg; ++sg) {
        const int rt0 = sg ? vrt0b : vrt0, nrt = sg ? vnrtb : vnrt;
        double acc3[2][3], acc2[2][2];
#pragma unroll
        for (int r = 0; r < 3; ++r) acc3[0][r] = acc3[1][r] = 0.0;
#pragma unroll
        for (int r = 0; r < 2; ++r) acc2[0][r] = acc2[1][r] = 0.0;
        kt_stream<P, NRES>(sg ? vpb : vp, nrt, vnjg, kb, lane, acc3, acc2, vres, sg ? NRES : 0, sg ? nresb : nres, bufA, bufB, sg ? 0 : RL_PRE);
        kt_tail<P, MAXDEG>(acc3, acc2, rt0, nrt, kb, al_l, vb, lane);
        if (stamping && lane == 0 && sg == nseg - 1) stl[16 + wv] += clock64() - tv0_;  // this wave's own phase V
        // ---- phase J over the rows this wave has just finished (wave-level ordering only) ----
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (MAXDEG == 0)
          lean_j<P>(xq, vb, Npad, 16 * rt0, 4 * nrt, lane, jacc[0][0], jacc[0][1]);
        else
          lean_j_groups<NCOL, NCG>(xq, vb, Npad, 16 * rt0, 4 * nrt, lane, jacc);
      }
#pragma unroll
      for (int g = 0; g < NCG; ++g) red[(wv * NCG + g) * 64 + lean_j_slot(lane)] = jacc[g][0] + jacc[g][1];
      // the z-only polynomial terms of this step (2.3 k cycles of one wave: LDS round trips in series), by wave 0 behind its own phase J --
      // it owns the fewest rows of Kinv and would wait ~4 k cycles at the barrier below; phase F (wave 0 itself) reads the result.
      // (At the end of phase K on a wave without an item in the last round it lengthened that phase by 0.9 k: profiles/r04_lean_variants.txt.)
      if (MAXDEG > 0 && wv == 0) lean_prefz<P, MAXDEG>(z, kpar, pc, fz, gpl[0].lambda, D, lane);
    } else {
#pragma unroll
      for (int g = 0; g < NCG; ++g) red[(wv * NCG + g) * 64 + lane] = 0.0;  // (a wave without rows)
    }
#ifdef RLX_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    lds_barrier();  // B4
    RL_STAMP(6);
    if (wv == 0) {
      sub_stamp = stamping ? clock64() : 0;
      // ---- phase F: sample delta and fold the sampling into d delta/dz; hand-off; integrate ------------
      const int* ro = role + lane * 16;
      const int4 r0 = *reinterpret_cast<const int4*>(ro), r1 = *reinterpret_cast<const int4*>(ro + 4), r2 = *reinterpret_cast<const int4*>(ro + 8),
                 r3 = *reinterpret_cast<const int4*>(ro + 12);
#pragma unroll
      for (int g = 0; g < NCG; ++g) {  // the 8 waves' partial tiles, added in wave order: lane l -> element (c = l >> 3, n = 8 g + (l & 7))
        double rv[RF_NW];
#pragma unroll
        for (int w = 0; w < RF_NW; ++w) rv[w] = red[(w * NCG + g) * 64 + lane];
        double sr = rv[0];
#pragma unroll
        for (int w = 1; w < RF_NW; ++w) sr += rv[w];
        rtot[g * 64 + lane] = sr;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (lane < P * (D + 1)) {
        const int p = r3.x, c = r3.y, crow = r3.w;  // (crow: the row of R that belongs to column c -- the sums come out in xq's row order)
        const GpL& gp = gpl[0];
        const double vscale = gp.var_scale;
        // SE only: R[D][2p] = sum_j k_j alpha_j,  R[D][2p+1] = k^T Kinv k;  R[c][.] the same sums weighted by X_jc.
        // Polynomial kernels: column p NW + s of R, slots s = 0 kse alpha, 1 kse v, then [v, k v] (degree 1) or [v B, v A, v, k v] (degree 2)
        // (GP_prior.py:137-155 with the kernel of GP_prior.py:314-335; the alpha-weighted polynomial sums and k(z, z): lean_prefz).
        const int cc = imin(c, D - 1);
        const double lc = pc[LAT_PC_LS + cc];  // (phase J summed against X / l: R[c][.] = l_c x the row of its result)
        constexpr int SV = MAXDEG >= 2 ? 4 : 2, SKV = SV + 1;  // slots of v and k v
        auto Rp = [&](int row, int slot) {
          const int col = p * NWC + slot;
          return rtot[(col >> 3) * 64 + row * 8 + (col & 7)];
        };
        v2d RD, RC;
        double mu, var;
        if (MAXDEG == 0) {
          RD = *reinterpret_cast<const v2d*>(__builtin_assume_aligned(rtot + ROW1 * 8 + 2 * p, 16));
          RC = *reinterpret_cast<const v2d*>(__builtin_assume_aligned(rtot + crow * 8 + 2 * p, 16));
          RC.x *= lc;
          RC.y *= lc;
          mu = gp.mean + RD.x;
          var = (gp.lambda - RD.y) * vscale;  // k(z,z) = lambda: Stationary_GP.py:172-181
        } else {
          RD.x = Rp(ROW1, 0);
          RD.y = Rp(ROW1, 1);
          RC.x = Rp(crow, 0) * lc;
          RC.y = Rp(crow, 1) * lc;
          mu = gp.mean + (RD.x + fz[p * 4 + 0]);
          var = (fz[p * 4 + 1] - Rp(ROW1, SKV)) * vscale;
        }
        double eps = 0.0, wj = 0.0, sd = 0.0;
        if (a.particle_pred) {
          eps = epsb[(t & 1) * P + p];
          sd = sqrt(var);
          wj = eps / (2.0 * sd);
        }
        if (c == D) {
          const double dv = a.particle_pred ? fma(sd, eps, mu) : mu;
          dl[p * G + myg] = dv;
          const unsigned long long bits = (unsigned long long)__double_as_longlong(dv);
          gu64_t slot = (gu64_t)a.xch + xch_slot(gcluster, t, G, myg, P) + 2 * p;
          store_granule(slot, (unsigned)t + 1u, (unsigned)bits);
          store_granule(slot + 1, (unsigned)t + 1u, (unsigned)(bits >> 32));
          if (m0 + p < Mend) {
            if (a.particle_pred && var <= 0.0) bad |= MCP_STATUS_NONPOS_VAR;  // (finite and not positive: a NaN variance is MCP_STATUS_NAN, the retry case)
            if (is_bad(mu) || is_bad(var)) bad |= MCP_STATUS_NAN;
          }
        } else if (a.jac && m0 + p < Mend) {
          // centred sums  sum_j w_j (z_c - X_jc) = z_c R[D][.] - R[c][.]
          const double il = kpar[KP_INVLS(D) + c], il2 = il * il, zc = z[p * D + c];
          double Jmu = -2.0 * il2 * fma(zc, RD.x, -RC.x);
          double Jvar = 4.0 * il2 * fma(zc, RD.y, -RC.y);
          if (MAXDEG >= 1) {
            const double w1c = kpar[KP_W1(D) + c];
            Jmu = fma(w1c, kpar[KP_AX(D) + c], Jmu);
            Jvar += 2.0 * w1c * (zc - Rp(crow, SV) * lc);
            if (MAXDEG >= 2) {
              const double a_ = kpar[KP_W20(D) + c], b_ = kpar[KP_W21(D) + c];
              const double qB = fz[P * 4 + (p * 8 + c) * 2], qA = fz[P * 4 + (p * 8 + c) * 2 + 1];
              Jmu += a_ * qB + b_ * qA;
              Jvar += 2.0 * zc * (a_ * fz[p * 4 + 3] + b_ * fz[p * 4 + 2]) - 2.0 * lc * (a_ * Rp(crow, 2) + b_ * Rp(crow, 3));
            }
          }
          a.jac[(((size_t)t * M + m0 + p) * G + myg) * D + c] = a.particle_pred ? fma(wj, Jvar * vscale, Jmu) : Jmu;
        }
      }
      RL_SUB(9);
      {
        // collect the other GPs' increments: lane -> (other GP, particle, half); every pass re-reads every granule
        const unsigned long long tx0_ = stamping ? clock64() : 0;
        const int ngr = (G - 1) * P * 2;
        const bool act = lane < ngr;
        const int go = act ? lane / (2 * P) : 0, r = act ? lane - go * 2 * P : 0;
        const int gq = go < myg ? go : go + 1;
        gu64_t slot = (gu64_t)a.xch + xch_slot(gcluster, t, G, gq, P) + r;
        unsigned val = 0;
        bool done = false;
        for (unsigned spins = 0; spins < RF_SPIN_LIMIT; ++spins) {
          bool ok = true;
          if (act) {
            const unsigned long long x = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            val = (unsigned)x;
            ok = (unsigned)(x >> 32) == (unsigned)t + 1u;
          }
          if (__all(ok)) {
            done = true;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        if (act) reinterpret_cast<unsigned*>(dl)[2 * ((r >> 1) * G + gq) + (r & 1)] = val;
        if (!done && lane == 0) *abortw = 1;
        if (stamping && lane == 0) stl[8] += clock64() - tx0_;
      }
      RL_SUB(10);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // ---- integrate:  v' = v + delta ;  q' = q + Ts v + Ts/2 delta   (Model_learning.py:711-716) ----
      if (lane < P * S) {
        const int op = r0.x, os = r0.y, g_vel = r1.z, g_pos = r1.w, vel_of_pos = r2.x;
        const double* xc = xs + cur * P * S + op * S;
        double nx = 0.0;
        if (g_vel >= 0) nx = xc[os] + dl[op * G + g_vel];
        if (g_pos >= 0) nx = xc[os] + Ts * xc[vel_of_pos] + 0.5 * Ts * dl[op * G + g_pos];
        xn = nx;
      }
      if (writer && lane < P * U) {  // the inputs of this step (off the critical path here); U <= 2
        const int ip = U == 1 ? lane : lane >> 1, ik = U == 1 ? 0 : lane & 1;
        if (m0 + ip < Mend) a.inputs[((size_t)t * M + m0 + ip) * U + ik] = z[ip * D + DS + ik];
      }
    } else {
      draw_step(t + 1, wv, lane);
    }
    cur ^= 1;  // x_{t+1} goes to the other buffer
    RL_STAMP(7);
    if (wv == 0) sub_stamp = stamping ? clock64() : 0;
  }
  if (stamping && tid0 < 24) a.stamps[tid0] += stl[tid0];  // (the stamp buffer of this kernel has 24 slots: tools/phase_stamps.py)
  if (bad) atomicOr(a.status, bad);
}


// the latency-lean GP-sharded kernel: narrow SE-only models (cart-pole class); KR = phase-K items per thread
static int gsh_grid(int nclusters, int G) { return ((nclusters + 7) / 8) * 8 * G; }  // (whole groups of 8 clusters: rollout_fwd.hip)
template <int P, int KR, int MAXDEG, bool PMS>
static int launch_fwd_lean_i(const FwdArgs& a, size_t lds, hipStream_t st) {
  MCP_ENSURE_MAX_LDS(rollout_fwd_lat_kernel<P, KR, MAXDEG, PMS>);
  hipLaunchKernelGGL((rollout_fwd_lat_kernel<P, KR, MAXDEG, PMS>), dim3(gsh_grid(a.nclusters, a.model.G)), dim3(RF_NT), lds, st, a);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}
// MAXDEGS: the polynomial degrees this (P, KR) pair is instantiated for (0 .. MAXDEGS)
template <int P, int KR, int MAXDEGS>
static int launch_fwd_lean_kr(const FwdArgs& a, size_t lds, hipStream_t st) {
  const bool pms = a.pol.meas.n > 0;
  if (a.maxdeg == 0) return pms ? launch_fwd_lean_i<P, KR, 0, true>(a, lds, st) : launch_fwd_lean_i<P, KR, 0, false>(a, lds, st);
  if constexpr (MAXDEGS >= 1)
    if (a.maxdeg == 1) return pms ? launch_fwd_lean_i<P, KR, 1, true>(a, lds, st) : launch_fwd_lean_i<P, KR, 1, false>(a, lds, st);
  if constexpr (MAXDEGS >= 2)
    if (a.maxdeg == 2) return pms ? launch_fwd_lean_i<P, KR, 2, true>(a, lds, st) : launch_fwd_lean_i<P, KR, 2, false>(a, lds, st);
  return MCP_ERR_LIMIT;
}
// phase-K items per thread (Npad * P <= KR * RF_NT) among the instantiations of P; 0: the shape has none.  Round 5: Npad up to 640
// (KR = 4, 5 at four particles) -- the training sets the launch scripts grow to without a subset (test_mcpilco4pms_cartpole.py: N = 450).
static int lean_items_per_thread(int P, int NpadMax, int maxdeg) {
  if (NpadMax < 32 || NpadMax > 640 || (NpadMax >> 4) > KT_MAX_RT) return 0;  // phase V deals 2 .. 48 row tiles of 16 in parts of 2 or 3
  if (NpadMax <= 384) return P == 4 ? 3 : (P == 2 ? 2 : 1);  // the base instantiations: one segment of Kinv per wave, every degree
  if (maxdeg > 1) return 0;                                  // beyond: two segments; SE and SE + polynomial(1) (the LDS takes no more)
  const int need = (NpadMax * P + RF_NT - 1) / RF_NT;
  if (P == 4) return need <= 4 ? 4 : (maxdeg == 0 ? 5 : 0);
  return P == 2 ? 3 : 2;
}
static bool lean_applies(const mcp_model* m, const mcp_policy* p, int P, int NpadMax, int maxdeg) {
  if (maxdeg < 0 || maxdeg > 2 || m->G < 2) return false;
  if (p->kind == MCP_POLICY_TRAJ) return false;  // (trajectory policies: the general kernel)
  for (int g = 0; g < m->G; ++g)
    if (m->gp[g].kern.poly_deg != maxdeg) return false;  // (one kernel structure for the whole model, as every launch script builds it)
  if (p->meas.n > 0) {  // measurement model: wave 1 draws P * n position noises beside the P process noises; one pair per state component
    if (P + P * p->meas.n > 64 || P * p->meas.n > 32) return false;
    for (int i = 0; i < p->meas.n; ++i)
      for (int j = 0; j < p->meas.n; ++j)
        if ((i != j && (p->meas.pos[i] == p->meas.pos[j] || p->meas.vel[i] == p->meas.vel[j])) || p->meas.pos[i] == p->meas.vel[j]) return false;
  }
  for (int i = 0; i < m->n_angle; ++i)  // (phase S writes a state component to the plain OR the sin / cos slots)
    for (int j = 0; j < m->n_not_angle; ++j)
      if (m->angle[i] == m->not_angle[j]) return false;
  if (p->kind == MCP_POLICY_ANGLES)
    for (int i = 0; i < p->n_angle; ++i)
      for (int j = 0; j < p->n_non_angle; ++j)
        if (p->angle[i] == p->non_angle[j]) return false;
  for (int g = 0; g < m->G; ++g)
    if (m->gp[g].Npad < 32) return false;  // (every GP needs two row tiles at least)
  if (m->D - m->U > RL_DSM || m->U > RL_UM || p->P > RL_PFM) return false;
  if (m->D >= RL_ZD) return false;  // (the ones row of phase J's operand takes a padded row of X^T / l)
  if (P * m->S > 64 || P * (m->D + 1) > 64 || (m->G - 1) * P * 2 > 64 || m->D > RL_MAXD) return false;  // wave 0 carries the serial section
  return lean_items_per_thread(P, NpadMax, maxdeg) > 0;
}
namespace mcp {
int launch_fwd_lean(const FwdArgs& a, int P, size_t lds, hipStream_t st) {
  const int kr = lean_items_per_thread(P, a.NpadMax, a.maxdeg);
  if (P == 4) return kr == 3 ? launch_fwd_lean_kr<4, 3, 2>(a, lds, st) : (kr == 4 ? launch_fwd_lean_kr<4, 4, 1>(a, lds, st) : launch_fwd_lean_kr<4, 5, 0>(a, lds, st));
  if (P == 2) return kr == 2 ? launch_fwd_lean_kr<2, 2, 2>(a, lds, st) : launch_fwd_lean_kr<2, 3, 1>(a, lds, st);
  return kr == 1 ? launch_fwd_lean_kr<1, 1, 2>(a, lds, st) : launch_fwd_lean_kr<1, 2, 1>(a, lds, st);
}
// dynamic LDS of the lean kernel for this shape at P particles per workgroup; 0: the kernel does not take the shape
size_t fwd_lean_lds_bytes(const mcp_model* m, const mcp_policy* p, int P, int NpadMax, int maxdeg) {
  if (!lean_applies(m, p, P, NpadMax, maxdeg)) return 0;
  return sizeof(double) * (size_t)lat_layout(P, p->B, NpadMax, maxdeg).total;
}
// Kinv of every GP as MFMA operand tiles, in each wave's streaming order (into a.kt: the caller's workspace)
int launch_fwd_lean_pack(const FwdArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(kt_pack_kernel, dim3(64, a.model.G), dim3(256), 0, st, a.model, (double*)a.kt, a.kt_stride);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}
}  // namespace mcp
