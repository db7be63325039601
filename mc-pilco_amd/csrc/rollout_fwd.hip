// Fused Monte-Carlo particle rollout, forward pass, for gfx950 (MI355X).
//
// Replaces MC_PILCO.apply_policy (policy_learning/MC_PILCO.py:615-674), i.e. per time step
//   Sum_of_gaussians*.forward                (policy_learning/Policy.py:242-265, 323-335, 389-403)
//   Model_learning.get_next_state            (model_learning/Model_learning.py:210-242, 265-336)
//     -> GP_prior.get_estimate_from_alpha    (gpr_lib/GP_prior/GP_prior.py:137-155), one per GP
//     -> get_next_state_from_gp_output       (Model_learning.py:685-718)
// and also produces what autograd's backward (MC_PILCO.py:522) needs from the GP.
//
// Parallel axis: particles.  They never interact inside the rollout, so a 512-thread workgroup
// owns P particles for all T steps and keeps their state in LDS.  Small swarms are launched
// GP-sharded (template GSH): the G workgroups of a cluster of P particles each evaluate one GP and
// hand each other the sampled increment once per step (tagged 8-byte granules, see below); larger
// ones one workgroup per P particles with no inter-workgroup traffic at all.  Per step and GP:  k = k(z,X) [N],  v = Kinv k [N]  (the N^2 term),  mu = m + k.a,
// var = k(z,z) - k.v,  and d mu/dz, d var/dz -- formed HERE from v (d var/dz = dk(z,z)/dz -
// 2 sum_j v_j dk_j/dz).  Only d delta_g/dz (G x D doubles per particle-step, sampling folded in)
// is stored, so the backward sweep never touches the GP again (rollout_bwd.hip).
//
// Memory plan: Kinv (N x N fp64 = 720 KB per GP at N=300) cannot live in the 160 KiB
// LDS; it stays L2-resident and is streamed once per step per workgroup -- symmetric, so
// "column i" is read as 128-double row segments (one 16-byte load per lane), double-buffered in
// registers, the stream cut into equal contiguous shares for the 8 waves (phase V).  Everything
// small and re-read every step (X^T, alpha, policy centres/weights) is copied to LDS once per
// launch when it fits (template XLDS).  The moment / Jacobian sums over the training index are one
// skinny MFMA product per GP (phase J), the other reductions wave64 DPP sums; partial results meet in
// LDS; 7 LDS-only workgroup barriers per time step.  Swarms above 1024 particles are handed to the
// 16-particle tile kernel (rollout_fwd_tile.hip).  With a measurement model (mcp_meas: partially
// measurable systems, MC_PILCO.py:808-906) phase S also produces what the policy sees: noisy
// positions, backward-difference velocities, first-order filter, three carried values per pair.
#include "rollout_fwd_shared.h"
#include "../../include/mcpilco_hip_debug.h"

using namespace mcp;

#define RF_MAX_NA 7  // accumulators per Jacobian item: 2 (SE), 3 (SE+P1), 7 (SE+P2)
#define RF_MAX_CHUNKS (MCP_MAX_GP * (MCP_MAX_TRAIN / 128))
#define RF_GS 8  // rows of Kinv per register buffer (two buffers in flight per wave)
#ifndef RF_NRES
#define RF_NRES 1  // register groups of Kinv a wave keeps resident for the whole rollout (GP-sharded 4-particle launch)
#endif

struct FwdLayout {
  int mk;  // dropout keep bits of the step, one int per (particle, 4 basis functions): drawn in phase S by idle waves
  int invl, xs, us, z, sf, dl, kb, ks, pa, pb, vb, part, red, xt, al, cen, wgt, tab, gpl, kpar, total;  // offsets in doubles
};

// integer tables (in the `tab` region): cstart[NC+1], cg[NC], cbase[NC], cR[NC], gcb[GB], wc0[NW], slo[NC], shi[NC]
#define TAB_CSTART 0
#define TAB_CG (RF_MAX_CHUNKS + 1)
#define TAB_CBASE (TAB_CG + RF_MAX_CHUNKS)
#define TAB_CR (TAB_CBASE + RF_MAX_CHUNKS)
#define TAB_GCB (TAB_CR + RF_MAX_CHUNKS)
#define TAB_WC0 (TAB_GCB + MCP_MAX_GP)
#define TAB_SLO (TAB_WC0 + RF_NW)            // first / last partial-sum slot of a chunk (phase vsum)
#define TAB_SHI (TAB_SLO + RF_MAX_CHUNKS)
#define TAB_INTS (TAB_SHI + RF_MAX_CHUNKS)
#define RF_CW 128  // rows of v per column chunk: 64 lanes x 2 rows (one 16-byte load per lane)

// GX = number of GPs whose operands are staged in LDS (G, or 1 in a GP-sharded launch)
__host__ __device__ inline FwdLayout fwd_layout(int P, int S, int U, int D, int G, int PF, int B, int NpadMax, int maxdeg, int GB,
                                                int NCmax, bool xlds, int GX = -1) {
  if (GX < 0) GX = G;
  FwdLayout L;
  int o = 0;
  auto take = [&](int n) {
    int r = o;
    o += (n + 1) & ~1;  // keep 16-byte alignment
    return r;
  };
  L.invl = take(PF + 2 * MCP_MAX_INPUT);  // policy inverse lengthscales | u_max | bias (staged once: a global load on phase U's critical path otherwise)
  L.xs = take(2 * P * S);
  L.us = take(P * U);
  L.z = take(P * D);
  L.sf = take(P * PF);
  L.dl = take(2 * P * G + 2);  // delta_g | process noise of the step (drawn in phase S by otherwise idle threads) | abort word (GSH)
  L.kb = take(GB * NpadMax * P);
  L.ks = maxdeg > 0 ? take(GB * NpadMax * P) : L.kb;
  L.pa = maxdeg > 1 ? take(GB * NpadMax * P) : L.kb;
  L.pb = maxdeg > 1 ? take(GB * NpadMax * P) : L.kb;
  L.vb = take(GB * NpadMax * P * (maxdeg == 0 ? 2 : 1));  // SE-only models store the phase-J weight matrix W[j][2p+a] instead of v
  L.part = take(imax((NCmax + RF_NW) * 128 * P, P * B));
  L.red = take(GB * RF_NW * (D + 1) * P * (maxdeg == 0 ? 2 : 9));
  L.xt = xlds ? take(GX * D * NpadMax) : 0;
  L.al = xlds ? take(GX * NpadMax) : 0;
  L.cen = xlds ? take(B * PF) : 0;
  L.wgt = xlds ? take(U * B) : 0;
  L.tab = take((TAB_INTS + 1) / 2);
  L.mk = take((P * ((B + 3) / 4) + 1) / 2);
  L.gpl = take(G * GPL_DOUBLES);
  L.kpar = take(G * (5 * D + 1));
  L.total = o;
  return L;
}

__device__ __forceinline__ int gp_num_acc(int deg) { return deg == 0 ? 2 : (deg == 1 ? 3 : RF_MAX_NA); }

// ---------------------------------------------------------------------------------------
// chunk table of one pass over GPs [g0, g0+gn): the Kinv stream is the list of 64-row-wide
// column chunks (g, ic), each N_g row segments ("units") long
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int chunks_of(int Npad) { return (Npad + RF_CW - 1) / RF_CW; }

// the Kinv stream of one pass over GPs [g0, g0+gn): the list of column chunks, each N_g (or N_g/2) units long
__device__ __forceinline__ int build_chunk_table(const GpL* gpl, int g0, int gn, int* tab, int tid) {
  int NC = 0;
  for (int g = 0; g < gn; ++g) NC += chunks_of(gpl[g0 + g].Npad);
  if (tid == 0) {
    int c = 0, acc = 0;
    for (int g = 0; g < gn; ++g) {
      tab[TAB_GCB + g] = c;
      const int Npad = gpl[g0 + g].Npad, N = gpl[g0 + g].N;
      const int nic = chunks_of(Npad);
      for (int ic = 0; ic < nic; ++ic) {
        const int width = imin(RF_CW, Npad - ic * RF_CW);
        const int R = width <= 64 ? 2 : 1;  // a narrow (tail) chunk takes two rows j per wave-load
        tab[TAB_CSTART + c] = acc;
        tab[TAB_CG + c] = g;
        tab[TAB_CBASE + c] = ic * RF_CW;
        tab[TAB_CR + c] = R;
        acc += (N + R - 1) / R;
        ++c;
      }
    }
    tab[TAB_CSTART + c] = acc;
    const int L = (acc + RF_NW - 1) / RF_NW;
    for (int w = 0; w < RF_NW; ++w) {
      int u0 = w * L, cc = 0;
      while (cc + 1 < c && tab[TAB_CSTART + cc + 1] <= u0) ++cc;
      tab[TAB_WC0 + w] = cc;
    }
    for (int cc = 0; cc < c; ++cc) {
      tab[TAB_SLO + cc] = cc + tab[TAB_CSTART + cc] / L;
      tab[TAB_SHI + cc] = cc + (tab[TAB_CSTART + cc + 1] - 1) / L;
    }
  }
  return NC;
}

// ---------------------------------------------------------------------------------------
// Phase K: covariance vectors of P test points against the training points of gn GPs
// -> LDS [gl][j][p]   (rows j >= N are never read)
// ---------------------------------------------------------------------------------------
template <int P, bool XLDS, int MAXDEG>
__device__ __forceinline__ void phase_k(const GpL* gpl, const double* kpar, int g0, int gn, int D, int NpadMax, const double* z,
                                        const double* xt_l, double* kb, double* ks, double* pa, double* pb, int tid) {
  // item = (training point j, GP gl, particle p), particle fastest: the decode is shifts for a single GP (integer division by
  // a run-time value costs more VALU time here than the kernel evaluation itself)
  const int Q = gn * P;
  for (int it = tid; it < NpadMax * Q; it += RF_NT) {
    int j, gl, p;
    if (gn == 1) {
      j = it / P;
      gl = 0;
      p = it - j * P;
    } else {
      j = it / Q;
      const int q = it - j * Q;
      gl = q / P;
      p = q - gl * P;
    }
    const GpL& gp = gpl[g0 + gl];
    const int N = gp.N, Npad = gp.Npad;
    if (j >= N) {
      if (j < Npad) kb[(gl * NpadMax + j) * P + p] = 0.0;  // phase V may touch one padded row
      continue;
    }
    const double* kp = kpar + (g0 + gl) * KP_STRIDE(D);
    const int deg = MAXDEG == 0 ? 0 : gp.deg;  // MAXDEG == 0: the polynomial code is compiled out
    const double* zp = z + p * D;
    const double* xc = XLDS ? xt_l + (g0 + gl) * D * NpadMax + j : gp.Xt + j;
    const int xs_ = XLDS ? NpadMax : Npad;
    // six dimensions at a time, all 18 operand reads issued before the first use (pinned: whether the compiler batches them
    // or emits read -> wait -> use per operand flips with unrelated changes elsewhere in the kernel: 4.5 k vs 5.6-7 k cycles)
    double dist = 0.0;
    for (int d0 = 0; d0 < D; d0 += 6) {
      double xv[6], zv[6], lv[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int d = imin(d0 + i, D - 1);
        xv[i] = xc[d * xs_];
        zv[i] = zp[d];
        lv[i] = kp[KP_INVLS(D) + d];
      }
      asm volatile("" ::"v"(xv[0]), "v"(xv[1]), "v"(xv[2]), "v"(xv[3]), "v"(xv[4]), "v"(xv[5]), "v"(zv[0]), "v"(zv[1]), "v"(zv[2]), "v"(zv[3]),
                   "v"(zv[4]), "v"(zv[5]), "v"(lv[0]), "v"(lv[1]), "v"(lv[2]), "v"(lv[3]), "v"(lv[4]), "v"(lv[5]));
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const double rr = (zv[i] - xv[i]) * (d0 + i < D ? lv[i] : 0.0);
        dist = fma(rr, rr, dist);
      }
    }
    double kse = gp.lambda * exp(-dist);
    double kt = kse;
    const int o = (gl * NpadMax + j) * P + p;
    if (deg >= 1) {
      double p1 = kp[KP_W1(D) + D];
      for (int d = 0; d < D; ++d) p1 = fma(kp[KP_W1(D) + d] * zp[d], xc[d * xs_], p1);
      kt += p1;
      if (deg >= 2) {
        double A = 0.0, Bv = 0.0;
        for (int d = 0; d < D; ++d) {
          double zx = zp[d] * xc[d * xs_];
          A = fma(kp[KP_W20(D) + d], zx, A);
          Bv = fma(kp[KP_W21(D) + d], zx, Bv);
        }
        kt = fma(A, Bv, kt);
        pa[o] = A;
        pb[o] = Bv;
      }
    }
    if (MAXDEG >= 1) ks[o] = kse;  // (with MAXDEG == 0 the layout aliases ks to kb)
    kb[o] = kt;
  }
}

// ---------------------------------------------------------------------------------------
// Phase V: v = Kinv k.  Measured on MI355X (tools/l2_stream_bench.hip): an L2-resident stream costs
// ~16 cycles per wave-load per CU whatever its width -- 8 B/lane loads cap at 30 B/clk/CU, 16 B/lane
// loads reach 60 B/clk/CU.  So: one 16-byte load per lane (global_load_dwordx4), a lane owns two
// adjacent rows (2l, 2l+1) of a 128-row column chunk and walks the summation index j; Kinv is
// symmetric, so element (i,j) is read from row j (contiguous).  A narrow tail chunk (<= 64 rows)
// packs two rows j, j+1 into one wave-load (lanes 32..63 take j+1) and folds the two halves at the
// end.  RF_GS loads per register buffer, two buffers in flight.
// The stream (all chunks of all GPs of the pass) is cut into RF_NW equal contiguous shares.
// ---------------------------------------------------------------------------------------
// k_j is the same for every lane, but a broadcast LDS read per row is not free: LDS returns and vector-memory returns share
// the path into the VGPRs, and phase V's time was (stream time) + (LDS cycles of the k reads): 130 / 147 / 190 cycles per row
// and wave round at 1 / 2 / 4 particles (halving the FMAs changed nothing).  So a group's whole k block (RF_GS rows x P
// particles) is fetched by ONE read, spread over the 16 lanes of every DPP row, and the FMAs take their k operand through
// DPP row_newbcast (v_fmac_f64_dpp: the lane select rides on the FMA, no extra instruction).
template <int N>
__device__ __forceinline__ void fmac_bcast(double& acc, double k, double a) {  // acc += k[lane N of this 16-lane row] * a
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(k), "v"(a), "n"(N));
}
template <int P>
struct KBlock {  // rows u = 0..RF_GS-1 of k, element e = u*P + p held by lane e/2 (component e&1) for even P, by lane e for P = 1
  double x, y;
};
template <int P>
__device__ __forceinline__ KBlock<P> read_k_block(const double* __restrict__ kk, int kstride, int n, int lane) {
  KBlock<P> kb;
  const int t = lane & 15;
  if (P % 2 == 0) {
    constexpr int H = P / 2;              // 16-byte pieces per row
    const int u = imin(t / H, n - 1), h = t % H;
    const v2d v = *reinterpret_cast<const v2d*>(__builtin_assume_aligned(kk + u * kstride + 2 * h, 16));
    kb.x = v.x;
    kb.y = v.y;
  } else {
    kb.x = kk[imin(t, n - 1) * kstride];
    kb.y = 0.0;
  }
  return kb;
}
template <int P, int U, int PP>
__device__ __forceinline__ void fma_unit_p(const v2d& a, const KBlock<P>& kb, double (&acc)[2][P]) {
  constexpr int e = U * P + PP;
  if (P % 2 == 0) {
    const double kv = (e & 1) ? kb.y : kb.x;
    fmac_bcast<(e >> 1)>(acc[0][PP], kv, a.x);
    fmac_bcast<(e >> 1)>(acc[1][PP], kv, a.y);
  } else {
    fmac_bcast<e>(acc[0][PP], kb.x, a.x);
    fmac_bcast<e>(acc[1][PP], kb.x, a.y);
  }
}
template <int P, int U>
__device__ __forceinline__ void fma_unit(const v2d& a, const KBlock<P>& kb, double (&acc)[2][P]) {
  fma_unit_p<P, U, 0>(a, kb, acc);
  if (P >= 2) fma_unit_p<P, U, (P >= 2 ? 1 : 0)>(a, kb, acc);
  if (P >= 4) {
    fma_unit_p<P, U, (P >= 4 ? 2 : 0)>(a, kb, acc);
    fma_unit_p<P, U, (P >= 4 ? 3 : 0)>(a, kb, acc);
  }
}
// rows 0..n-1 of the register buffer (n wave-uniform; n == RF_GS: no branches)
template <int P>
__device__ __forceinline__ void consume_rows(const v2d (&A)[RF_GS], KBlock<P> kb, int n, double (&acc)[2][P]) {
  static_assert(RF_GS == 8 && RF_GS * P <= 32, "k block: one 16-byte piece per lane of a 16-lane row");
  // a VALU write of the block followed at once by a DPP read of it needs two wait states: take them here, once
  asm volatile("s_nop 1" : "+v"(kb.x), "+v"(kb.y));
  if (n == RF_GS) {
    fma_unit<P, 0>(A[0], kb, acc);
    fma_unit<P, 1>(A[1], kb, acc);
    fma_unit<P, 2>(A[2], kb, acc);
    fma_unit<P, 3>(A[3], kb, acc);
    fma_unit<P, 4>(A[4], kb, acc);
    fma_unit<P, 5>(A[5], kb, acc);
    fma_unit<P, 6>(A[6], kb, acc);
    fma_unit<P, 7>(A[7], kb, acc);
  } else {
    if (0 < n) fma_unit<P, 0>(A[0], kb, acc);
    if (1 < n) fma_unit<P, 1>(A[1], kb, acc);
    if (2 < n) fma_unit<P, 2>(A[2], kb, acc);
    if (3 < n) fma_unit<P, 3>(A[3], kb, acc);
    if (4 < n) fma_unit<P, 4>(A[4], kb, acc);
    if (5 < n) fma_unit<P, 5>(A[5], kb, acc);
    if (6 < n) fma_unit<P, 6>(A[6], kb, acc);
  }
}

__device__ __forceinline__ void load_rows(v2d (&A)[RF_GS], gptr_t p, size_t rstride) {
#pragma unroll
  for (int u = 0; u < RF_GS; ++u) A[u] = *(gptr2_t)(p + (size_t)u * rstride);
}

// units [ua, ub) of one chunk; `base` already points at this lane's two columns of its first row,
// `kk` at its k row; consecutive units are rstride / kstride apart.
// RESIDENT GROUPS (NRES > 0): the unit stream of a launch is static, so a wave reads the same rows of Kinv every time step.
// The first `nres` (<= NRES) register groups of the wave's first segment are loaded ONCE, before the time loop, and stay in
// VGPRs for the whole rollout (res[][]): every step they cost FMAs only, and the L2 -> CU stream -- the bound of this phase at
// 16-21 cycles per wave-load whatever else the CU does -- carries that much less.  They are consumed after the first streamed
// buffer has been issued (their FMAs hide its latency) and in the same order as before: results are bit-identical.
template <int P, int NRES>
__device__ __forceinline__ void matvec_rows(gptr_t base, size_t rstride, const double* __restrict__ kk, int kstride,
                                            int ua, int ub, int lane, double (&acc)[2][P], const v2d (&res)[NRES > 0 ? NRES : 1][RF_GS],
                                            int nres) {
  const size_t gstep = (size_t)RF_GS * rstride;
  int u0 = ua + nres * RF_GS;  // first streamed unit
  gptr_t p = base + (size_t)u0 * rstride;
  const int nfull = (ub - u0) / RF_GS;
  v2d A[RF_GS], Bf[RF_GS];
  KBlock<P> kA, kB;  // the k block of a register buffer is read when the buffer's loads are issued
  if (nfull > 0) {
    load_rows(A, p, rstride);
    kA = read_k_block<P>(kk + u0 * kstride, kstride, RF_GS, lane);
    p += gstep;
  }
  if (NRES > 0) {
#pragma unroll
    for (int r = 0; r < NRES; ++r) {
      if (r < nres) {  // wave-uniform
        const KBlock<P> kR = read_k_block<P>(kk + (ua + r * RF_GS) * kstride, kstride, RF_GS, lane);
        consume_rows<P>(res[NRES > 0 ? r : 0], kR, RF_GS, acc);
      }
    }
  }
  if (nfull > 0) {
    for (int g = 0; g < nfull; g += 2) {
      const bool hasB = g + 1 < nfull;
      if (hasB) {
        load_rows(Bf, p, rstride);
        kB = read_k_block<P>(kk + (u0 + RF_GS) * kstride, kstride, RF_GS, lane);
        p += gstep;
      }
      consume_rows<P>(A, kA, RF_GS, acc);
      u0 += RF_GS;
      if (hasB) {
        if (g + 2 < nfull) {
          load_rows(A, p, rstride);
          kA = read_k_block<P>(kk + (u0 + RF_GS) * kstride, kstride, RF_GS, lane);
          p += gstep;
        }
        consume_rows<P>(Bf, kB, RF_GS, acc);
        u0 += RF_GS;
      }
    }
  }
  const int rem = ub - u0;  // 0 .. RF_GS-1, wave-uniform
  if (rem > 0) {
#pragma unroll
    for (int u = 0; u < RF_GS - 1; ++u)
      if (u < rem) A[u] = *(gptr2_t)(p + (size_t)u * rstride);
    kA = read_k_block<P>(kk + u0 * kstride, kstride, rem, lane);
    consume_rows<P>(A, kA, rem, acc);
  }
}

// the first segment of wave wv's share of the unit stream: chunk, unit range and this lane's base pointer
struct VSeg {
  int c, ua, ub, gl, rb, R, Npad, sub, li;
  gptr_t base;
};
__device__ __forceinline__ VSeg v_segment(const GpL* gpl, int g0, const int* tab, int c, int u, int u1, int lane) {
  VSeg q;
  q.c = c;
  const int cs = tab[TAB_CSTART + c], ce = tab[TAB_CSTART + c + 1];
  q.gl = tab[TAB_CG + c];
  q.rb = tab[TAB_CBASE + c];
  q.R = tab[TAB_CR + c];
  q.Npad = __builtin_amdgcn_readfirstlane(gpl[g0 + q.gl].Npad);
  q.ua = u - cs;
  q.ub = imin(ce, u1) - cs;
  q.sub = (q.R == 2) ? (lane >> 5) : 0;   // which of the unit's R rows this lane reads
  q.li = (q.R == 2) ? (lane & 31) : lane;  // lane's column pair inside the chunk
  const int i = q.rb + 2 * q.li;
  q.base = (gptr_t)gpl[g0 + q.gl].Kinv + (size_t)q.sub * q.Npad + (i < q.Npad ? i : q.rb);
  return q;
}

// loads the resident groups of this wave (once per launch); returns how many there are
template <int NRES>
__device__ __forceinline__ int load_resident(const GpL* gpl, int g0, const int* tab, int NC, int wv, int lane, v2d (&res)[NRES > 0 ? NRES : 1][RF_GS]) {
  if (NRES == 0) return 0;
  const int total = tab[TAB_CSTART + NC];
  const int L = (total + RF_NW - 1) / RF_NW;
  const int u = wv * L, u1 = imin(total, u + L);
  int nres = 0;
  if (u < u1) {
    const VSeg q = v_segment(gpl, g0, tab, tab[TAB_WC0 + wv], u, u1, lane);
    nres = imin(NRES, (q.ub - q.ua) / RF_GS);
    const size_t rstride = (size_t)q.R * q.Npad;
#pragma unroll
    for (int r = 0; r < NRES; ++r)
      if (r < nres) load_rows(res[NRES > 0 ? r : 0], q.base + (size_t)(q.ua + r * RF_GS) * rstride, rstride);
  }
  return nres;
}

template <int P, int NRES = 0>
__device__ __forceinline__ void phase_v(const GpL* gpl, int g0, const int* tab, int NC, int NpadMax, const double* kb, double* part,
                                        int wv, int lane, const v2d (&res)[NRES > 0 ? NRES : 1][RF_GS], int nres) {
  const int total = tab[TAB_CSTART + NC];
  const int L = (total + RF_NW - 1) / RF_NW;
  int u = wv * L;
  const int u1 = imin(total, u + L);
  if (u >= u1) return;
  int c = tab[TAB_WC0 + wv];
  bool first = true;
  while (u < u1) {
    const VSeg q = v_segment(gpl, g0, tab, c, u, u1, lane);
    double acc[2][P];
#pragma unroll
    for (int p = 0; p < P; ++p) acc[0][p] = acc[1][p] = 0.0;
    matvec_rows<P, NRES>(q.base, (size_t)q.R * q.Npad, kb + (q.gl * NpadMax + q.sub) * P, q.R * P, q.ua, q.ub, lane, acc, res, first ? nres : 0);
    first = false;
    if (q.R == 2) {
      // fold rows j+1 (lanes 32..63) into rows j (lanes 0..31)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        acc[0][p] = sum_xor32(acc[0][p]);
        acc[1][p] = sum_xor32(acc[1][p]);
      }
    }
    double* slot = part + (c + wv) * 128 * P;  // slot id = chunk + wave: unique, contiguous per chunk
    if (lane == q.li) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        slot[(2 * q.li) * P + p] = acc[0][p];
        slot[(2 * q.li + 1) * P + p] = acc[1][p];
      }
    }
    u += q.ub - q.ua;
    ++c;
  }
}

// v[gl][i][p] = sum of the chunk's partial slots, in a fixed order, for the rows [j0, j1) of one GP -- the rows whose phase-J
// sums the calling wave forms next, so the wave needs no workgroup barrier between the two (only its own LDS order).  For
// SE-only models (MAXDEG == 0) the lane goes one step further and stores the two phase-J weights of its (j, p):
// W[j][2p] = kse_j alpha_j,  W[j][2p+1] = kse_j v_j,  so that phase J needs a single LDS read per MFMA operand.
template <int P, bool XLDS, int MAXDEG>
__device__ __forceinline__ void vsum_range(const GpL& gp, int ggl, int gl, const int* tab, int NpadMax, const double* part, const double* kb,
                                           const double* al_l, double* vb, int j0, int j1, int lane) {
  const int cb = tab[TAB_GCB + gl];
  for (int itq = lane; itq < (j1 - j0) * P; itq += 64) {
    const int i = j0 + itq / P, p = itq % P;
    const int it = (gl * NpadMax + i) * P + p;
    const int c = cb + i / RF_CW;
    const int s_lo = tab[TAB_SLO + c], s_hi = tab[TAB_SHI + c];
    double s = 0.0;
    for (int sid = s_lo; sid <= s_hi; ++sid) s += part[(sid * 128 + (i % RF_CW)) * P + p];
    if (MAXDEG == 0) {
      const double kse = kb[it];
      const double alj = XLDS ? al_l[ggl * NpadMax + i] : ((gptr_t)gp.alpha)[i];
      vb[2 * it] = kse * alj;
      vb[2 * it + 1] = kse * s;
    } else {
      vb[it] = s;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------
// Phase J: moment / Jacobian sums, one wave per (gp, particle, column) item, lanes over j, DPP sum
//   column c <  D : a0 = sum kse_j alpha_j (z_c - X_jc)      a1 = sum kse_j v_j (z_c - X_jc)
//                   a2 = sum v_j X_jc            (deg>=1)
//                   a3 = sum alpha_j B_j X_jc, a4 = sum alpha_j A_j X_jc, a5 = sum v_j B_j X_jc, a6 = sum v_j A_j X_jc (deg 2)
//   column c == D : a0 = sum k_j alpha_j  (= mu - m)          a1 = sum k_j v_j  (= k^T Kinv k)
// ---------------------------------------------------------------------------------------
// Phase J on the matrix cores.  Per GP the moment / Jacobian sums are one skinny contraction over the training index
//     R[c][n] = sum_j Xe[c][j] * W[j][n],      Xe = [X^T ; 1]  ((D+1) x N),   W (N x P*NAX) = per-particle weight vectors
// with, per particle, the columns  0: kse*alpha   1: kse*v   [deg>=1] 2: v   [deg 2] 3: alpha*B  4: alpha*A  5: v*B  6: v*A
// and, when deg>=1, the two totals columns  NAX-2: k*alpha   NAX-1: k*v   (for deg 0  k == kse, so columns 0/1 serve).
// v_mfma_f64_16x16x4_f64 sums 4 values of j per instruction inside the matrix core, so no cross-lane reduction is needed:
// A operand  lane l -> Xe[c = l&15][j = jb + (l>>4)]   (an LDS read),  B operand  lane l -> W[j = jb + (l>>4)][n = l&15]
// (three LDS reads and a multiply), accumulator  D[row = (l>>4) + 4r][col = l&15].  The 8 waves split N; their partial tiles
// meet in LDS (redw) and the finalize phase adds the 8 partials.  The centred sums the Jacobians need follow from
//     sum_j w_j (z_c - X_jc) = z_c * R[D][n] - R[c][n].
typedef double v4d __attribute__((ext_vector_type(4)));
#define RF_NAX(deg) ((deg) == 0 ? 2 : ((deg) == 1 ? 5 : 9))

template <int P, bool XLDS, int DEG, bool WPRE>
__device__ __forceinline__ void phase_j_gp(const GpL& gp, int ggl, int gl, int D, int NpadMax, int ncolmax, const double* xt_l,
                                           const double* al_l, const double* kb, const double* ks, const double* pa, const double* pb,
                                           double* vb, const int* tab, const double* part, double* redw, int wv, int lane,
                                           unsigned long long* jst = nullptr) {
  unsigned long long t0_ = jst ? clock64() : 0;
  constexpr int NAX = RF_NAX(DEG);
  constexpr int NCOLS = P * NAX;
  constexpr int CT = (NCOLS + 15) / 16;
  const int RT = (D + 1 + 15) >> 4;  // <= 3
  const int N = __builtin_amdgcn_readfirstlane(gp.N);
  const int Npad = __builtin_amdgcn_readfirstlane(gp.Npad);
  const int per = ((N + RF_NW * 4 - 1) / (RF_NW * 4)) * 4;  // this wave's share of j, a multiple of 4
  const int j0 = wv * per, j1 = imin(N, j0 + per);
  const int kq = lane >> 4, li = lane & 15;
  vsum_range<P, XLDS, (WPRE ? 0 : 2)>(gp, ggl, gl, tab, NpadMax, part, kb, al_l, vb, j0, j1, lane);
  v4d acc[3][CT];
#pragma unroll
  for (int rt = 0; rt < 3; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (v4d){0.0, 0.0, 0.0, 0.0};
  if (WPRE && CT == 1 && RT == 1) {
    // fast path (SE-only model, one 16x16 tile): batch the operand reads of RF_JU steps, then issue the MFMAs back to back
    constexpr int RF_JU = 10;  // N = 300 over 8 waves is 10 steps of 4: one batch of operand reads, then the MFMAs back to back
    const int c = li, cc = imin(c, D - 1);
    const double* xrow = XLDS ? xt_l + (ggl * D + cc) * NpadMax : nullptr;
    const double* wrow = vb + gl * NpadMax * NCOLS + imin(li, NCOLS - 1);
    const bool nok = li < NCOLS;
    v4d a0 = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int jb = j0; jb < j1; jb += 4 * RF_JU) {
      // all 2*RF_JU operand reads are issued before anything consumes them (pinned by the asm statement): left alone, the
      // compiler sinks every read to its use and each MFMA step pays an LDS latency
      double av[RF_JU], bw[RF_JU];
#pragma unroll
      for (int u = 0; u < RF_JU; ++u) {
        const int j = jb + 4 * u + kq;
        const int jc = j < j1 ? j : j0;
        av[u] = XLDS ? xrow[jc] : ((gptr_t)gp.Xt)[(size_t)cc * Npad + jc];
        bw[u] = wrow[jc * NCOLS];
      }
      static_assert(RF_JU == 10, "operand list below");
      asm volatile("" : "+v"(av[0]) : "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(av[9]),
                   "v"(bw[0]), "v"(bw[1]), "v"(bw[2]), "v"(bw[3]), "v"(bw[4]), "v"(bw[5]), "v"(bw[6]), "v"(bw[7]), "v"(bw[8]), "v"(bw[9]));
#pragma unroll
      for (int u = 0; u < RF_JU; ++u) {
        const bool jok = jb + 4 * u + kq < j1;
        av[u] = !jok ? 0.0 : (c < D ? av[u] : (c == D ? 1.0 : 0.0));
        bw[u] = (jok && nok) ? bw[u] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < RF_JU; ++u) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bw[u], a0, 0, 0, 0);
    }
    acc[0][0] = a0;
  } else
  for (int jb = j0; jb < j1; jb += 4) {
    const int j = jb + kq;
    const bool jok = j < j1;
    const int jc = jok ? j : j0;  // clamp: out-of-range lanes read a valid address and contribute zero
    double alj = 0.0;
    if (!WPRE) alj = XLDS ? al_l[ggl * NpadMax + jc] : ((gptr_t)gp.alpha)[jc];
    double bv[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int n = ct * 16 + li;
      if (WPRE) {  // SE-only: the weights were stored by phase vsum, W[j][n], n < 2P
        const double w = vb[(gl * NpadMax + jc) * NCOLS + imin(n, NCOLS - 1)];
        bv[ct] = (jok && n < NCOLS) ? w : 0.0;
        continue;
      }
      const int p = imin(n / NAX, P - 1), aidx = n % NAX;
      const int o = (gl * NpadMax + jc) * P + p;
      const double kse = ks[o], v = vb[o];
      double w = (aidx == 0) ? kse * alj : kse * v;
      if (DEG >= 1) {
        const double kt = kb[o];
        if (aidx == 2) w = v;
        if (aidx == NAX - 2) w = kt * alj;
        if (aidx == NAX - 1) w = kt * v;
        if (DEG >= 2) {
          const double A = pa[o], Bv = pb[o];
          if (aidx == 3) w = alj * Bv;
          if (aidx == 4) w = alj * A;
          if (aidx == 5) w = v * Bv;
          if (aidx == 6) w = v * A;
        }
      }
      bv[ct] = (jok && n < NCOLS) ? w : 0.0;
    }
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
      if (rt < RT) {
        const int c = rt * 16 + li;
        const int cc = imin(c, D - 1);
        const double x = XLDS ? xt_l[(ggl * D + cc) * NpadMax + jc] : ((gptr_t)gp.Xt)[(size_t)cc * Npad + jc];
        const double av = !jok ? 0.0 : (c < D ? x : (c == D ? 1.0 : 0.0));
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[ct], acc[rt][ct], 0, 0, 0);
      }
    }
  }
  unsigned long long t1_ = jst ? clock64() : 0;
  if (jst && lane == 0) jst[12] += t1_ - t0_;
  double* out = redw + (gl * RF_NW + wv) * (D + 1) * ncolmax;
#pragma unroll
  for (int rt = 0; rt < 3; ++rt) {
    if (rt < RT) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = rt * 16 + kq + 4 * r, n = ct * 16 + li;
          if (c <= D && n < NCOLS) out[c * ncolmax + n] = acc[rt][ct][r];
        }
      }
    }
  }
  if (jst && lane == 0) jst[13] += clock64() - t1_;
}

template <int P, bool XLDS, int MAXDEG>
__device__ __forceinline__ void phase_j(const GpL* gpl, int g0, int gn, int D, int NpadMax, const double* xt_l, const double* al_l,
                                        const double* kb, const double* ks, const double* pa, const double* pb, double* vb,
                                        const int* tab, const double* part, double* redw, int wv, int lane, unsigned long long* jst = nullptr) {
  constexpr int NCOLMAX = P * RF_NAX(MAXDEG);
  for (int gl = 0; gl < gn; ++gl) {
    const GpL& gp = gpl[g0 + gl];
    const int deg = MAXDEG == 0 ? 0 : __builtin_amdgcn_readfirstlane(gp.deg);
    if (MAXDEG == 0)
      phase_j_gp<P, XLDS, 0, true>(gp, g0 + gl, gl, D, NpadMax, NCOLMAX, xt_l, al_l, kb, ks, pa, pb, vb, tab, part, redw, wv, lane, jst);
    else if (deg == 0)
      phase_j_gp<P, XLDS, 0, false>(gp, g0 + gl, gl, D, NpadMax, NCOLMAX, xt_l, al_l, kb, ks, pa, pb, vb, tab, part, redw, wv, lane);
    else if (deg == 1)
      phase_j_gp<P, XLDS, 1, false>(gp, g0 + gl, gl, D, NpadMax, NCOLMAX, xt_l, al_l, kb, ks, pa, pb, vb, tab, part, redw, wv, lane);
    else
      phase_j_gp<P, XLDS, 2, false>(gp, g0 + gl, gl, D, NpadMax, NCOLMAX, xt_l, al_l, kb, ks, pa, pb, vb, tab, part, redw, wv, lane);
  }
}

// R[c][col] summed over the 8 waves' partial tiles (fixed order); Rg = redw + gl*RF_NW*(D+1)*ncolmax
__device__ __forceinline__ double j_sum(const double* Rg, int D, int ncolmax, int c, int col) {
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < RF_NW; ++w) s += Rg[(w * (D + 1) + c) * ncolmax + col];
  return s;
}

// posterior mean / variance and their z-Jacobians from the contraction results of this (gp, particle); col0 = p * NAX
template <int MAXDEG>
__device__ __forceinline__ void gp_point(const GpL& gp, const double* kp, int D, const double* zp, const double* Rg, int ncolmax, int p,
                                         double& mu, double& var) {
  const int deg = MAXDEG == 0 ? 0 : gp.deg;
  const int nax = RF_NAX(deg), col0 = p * nax;
  mu = gp.mean + j_sum(Rg, D, ncolmax, D, col0 + (deg == 0 ? 0 : nax - 2));
  double kzz = gp.lambda;  // k(z,z): Stationary_GP.py:172-181, Sparse_GP.py:443-453,658-668
  if (MAXDEG >= 1 && deg >= 1) {
    double p1 = kp[KP_W1(D) + D];
    for (int d = 0; d < D; ++d) p1 = fma(kp[KP_W1(D) + d] * zp[d], zp[d], p1);
    kzz += p1;
    if (deg >= 2) {
      double sa = 0.0, sb = 0.0;
      for (int d = 0; d < D; ++d) {
        double zz = zp[d] * zp[d];
        sa = fma(kp[KP_W20(D) + d], zz, sa);
        sb = fma(kp[KP_W21(D) + d], zz, sb);
      }
      kzz = fma(sa, sb, kzz);
    }
  }
  var = kzz - j_sum(Rg, D, ncolmax, D, col0 + (deg == 0 ? 1 : nax - 1));
}
template <int MAXDEG>
__device__ __forceinline__ void gp_jac(const GpL& gp, const double* kp, int D, const double* zp, const double* Rg, int ncolmax, int p, int d,
                                       double& Jmu, double& Jvar) {
  const int deg = MAXDEG == 0 ? 0 : gp.deg;
  const int col0 = p * RF_NAX(deg);
  double il = kp[KP_INVLS(D) + d];
  double il2 = il * il;
  // centred sums  sum_j w_j (z_d - X_jd) = z_d * R[D][.] - R[d][.]
  double r0 = fma(zp[d], j_sum(Rg, D, ncolmax, D, col0 + 0), -j_sum(Rg, D, ncolmax, d, col0 + 0));
  double r1 = fma(zp[d], j_sum(Rg, D, ncolmax, D, col0 + 1), -j_sum(Rg, D, ncolmax, d, col0 + 1));
  Jmu = -2.0 * il2 * r0;
  Jvar = 4.0 * il2 * r1;
  if (MAXDEG >= 1 && deg >= 1) {
    double w1d = kp[KP_W1(D) + d];
    Jmu = fma(w1d, kp[KP_AX(D) + d], Jmu);
    Jvar += 2.0 * w1d * (zp[d] - j_sum(Rg, D, ncolmax, d, col0 + 2));
    if (deg >= 2) {
      double Sa = 0.0, Sb = 0.0;
      for (int e = 0; e < D; ++e) {
        double zz = zp[e] * zp[e];
        Sa = fma(kp[KP_W20(D) + e], zz, Sa);
        Sb = fma(kp[KP_W21(D) + e], zz, Sb);
      }
      double a_ = kp[KP_W20(D) + d], b_ = kp[KP_W21(D) + d];
      Jmu += a_ * j_sum(Rg, D, ncolmax, d, col0 + 3) + b_ * j_sum(Rg, D, ncolmax, d, col0 + 4);
      Jvar += 2.0 * zp[d] * (a_ * Sb + b_ * Sa) - 2.0 * (a_ * j_sum(Rg, D, ncolmax, d, col0 + 5) + b_ * j_sum(Rg, D, ncolmax, d, col0 + 6));
    }
  }
}

// ---------------------------------------------------------------------------------------
// GP-sharded launch (GSH): hand-off of the sampled increments between the G workgroups of a particle cluster.
// The data is the flag: every double travels as two naturally aligned 8-byte {tag = t + 1, 32-bit half} granules, each
// written by ONE agent-scope (write-through) store and re-read by agent-scope loads until its tag matches; no fence, no
// ordering between granules.  Slots alternate with the parity of t: a workgroup can only be one hand-off ahead of its
// partners, so slot (t & 1) is not rewritten before every partner has consumed step t.  The buffer is zeroed by the
// launch function on the stream (tags start at 1).  Spins are bounded: a partner that never shows up (a grid larger
// than the device can hold would be the only reason) ends the rollout with MCP_STATUS_SYNC instead of hanging.
// ---------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// forward rollout.  GSH = false: one workgroup per P particles, all GPs.  GSH = true: G workgroups per P particles, each
// evaluates ONE GP (streams one Kinv) and the policy; they meet once per step in the hand-off above.  Blocks b and b + 8
// are dealt to the same XCD, so the members of a cluster sit 8 apart (a speed matter only).
// ---------------------------------------------------------------------------------------
template <int P, bool XLDS, int MAXDEG, bool GSH>
__global__ __launch_bounds__(RF_NT) void rollout_fwd_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const mcp_model& md = a.model;
  const mcp_policy& pl = a.pol;
  const mcp_noise nzl = noise_of_launch(a.nz);
  const mcp_gp* gps = md.gp;
  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int wv0 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int S = md.S, U = md.U, G = md.G, D = md.D, B = pl.B, PF = pl.P, M = a.M, T = a.T;
  const int NpadMax = a.NpadMax, GB = a.GB;
  const FwdLayout L = fwd_layout(P, S, U, D, G, PF, B, NpadMax, a.maxdeg, GB, a.NCmax, XLDS, GSH ? 1 : G);
  double* invl = smem + L.invl;
  double* xs = smem + L.xs;  // [2][P][S] double-buffered
  double* us = smem + L.us;
  double* z = smem + L.z;
  double* sf = smem + L.sf;
  double* dl = smem + L.dl;
  double* epsb = dl + P * G;
  double* kb = smem + L.kb;
  double* ks = smem + L.ks;
  double* pa = smem + L.pa;
  double* pb = smem + L.pb;
  double* vb = smem + L.vb;
  double* part = smem + L.part;
  double* red = smem + L.red;
  double* xt_l = smem + L.xt;
  double* al_l = smem + L.al;
  double* cen_l = smem + L.cen;
  double* wgt_l = smem + L.wgt;
  int* tab = reinterpret_cast<int*>(smem + L.tab);
  int* mk = reinterpret_cast<int*>(smem + L.mk);
  GpL* gpl = reinterpret_cast<GpL*>(smem + L.gpl);
  double* kpar = smem + L.kpar;
  int cluster = blockIdx.x, myg = 0;
  if (GSH) {
    const int b = blockIdx.x, grp = b / (8 * G), r = b - grp * 8 * G;
    cluster = grp * 8 + (r & 7);
    myg = r >> 3;
    if (cluster >= a.nclusters) return;  // padding blocks of the last group of 8 clusters
  }
  const int gcluster = GSH ? a.m_off / P + cluster : cluster;  // cluster index in the whole swarm (hand-off slots)
  const bool writer = !GSH || myg == 0;  // states / inputs are identical in the workgroups of a cluster: one of them stores
  int* abortw = reinterpret_cast<int*>(dl + 2 * P * G);
  if (GSH && tid0 == 0) *abortw = 0;
  const int m0 = (GSH ? a.m_off : 0) + cluster * P;
  const int Mend = GSH ? a.m_off + a.m_cnt : M;  // one past the last particle of this launch (M itself: strides of the [T][M][.] arrays)
  uint32_t bad = 0;
  const bool drop = pl.p_drop > 0.0;
  const double keep_scale = 1.0 / (1.0 - pl.p_drop);
  const uint32_t drop_thr = drop_threshold(pl.p_drop);
  const int nna = md.n_not_angle, na = md.n_angle;

  // ---- one-time staging ------------------------------------------------------------------
  for (int it = tid0; it < PF; it += RF_NT) invl[it] = exp(-pl.log_ls[it]);
  double* umax_l = invl + PF;
  double* bias_l = umax_l + MCP_MAX_INPUT;  // f_linear.bias (0 without flg_bias)
  if (tid0 < U) {
    umax_l[tid0] = pl.u_max[tid0];
    bias_l[tid0] = pl.bias ? pl.bias[tid0] : 0.0;
  }
  // the GPs this workgroup evaluates: all of them, or (GSH) its own one, which then lives in slot 0 of every LDS table
  const int GL = GSH ? 1 : G;
  const mcp_gp* gps_l = gps + myg;
  stage_gp_tables(gps_l, md.var_scale + myg, GL, D, gpl, kpar, tid0);
  if (XLDS) {
    for (int g = 0; g < GL; ++g) {
      const mcp_gp& gp = gps_l[g];
      for (int it = tid0; it < D * gp.Npad; it += RF_NT) {
        int d = it / gp.Npad, j = it - d * gp.Npad;
        xt_l[(g * D + d) * NpadMax + j] = gp.Xt[it];
      }
      for (int it = tid0; it < gp.Npad; it += RF_NT) al_l[g * NpadMax + it] = gp.alpha[it];
    }
    for (int it = tid0; it < B * PF; it += RF_NT) cen_l[it] = pl.centers[it];
    for (int it = tid0; it < U * B; it += RF_NT) wgt_l[it] = pl.weight[it];
  }
  const double* cen = XLDS ? cen_l : pl.centers;
  const double* wgt = XLDS ? wgt_l : pl.weight;
  lds_barrier();
  int NC = 0;
  if (GSH)
    NC = build_chunk_table(gpl, 0, 1, tab, tid0);
  else if (GB >= G)
    NC = build_chunk_table(gpl, 0, G, tab, tid0);

  // thread (p, s) owns state component s of particle p
  const bool own = tid0 < P * S;
  const int op = own ? tid0 / S : 0, os = own ? tid0 - op * S : 0;
  const int om = imin(m0 + op, Mend - 1);
  const bool ovalid = own && (m0 + op < Mend);
  double xn = own ? a.x0[(size_t)om * S + os] : 0.0;
  int cur = 0;
  // what this state component feeds (fixed for the whole rollout): GP-feature slots, policy-feature slots, integrator role
  int zi_plain = -1, zi_ang = -1, pi_plain = -1, pi_ang = -1, g_vel = -1, g_pos = -1;
  if (own) {
    for (int i = 0; i < nna; ++i)
      if (md.not_angle[i] == os) zi_plain = i;
    for (int i = 0; i < na; ++i)
      if (md.angle[i] == os) zi_ang = i;
    if (pl.kind == MCP_POLICY_ANGLES) {
      for (int i = 0; i < pl.n_non_angle; ++i)
        if (pl.non_angle[i] == os) pi_plain = i;
      for (int i = 0; i < pl.n_angle; ++i)
        if (pl.angle[i] == os) pi_ang = i;
    }
    for (int g = 0; g < G; ++g) {
      if (md.vel[g] == os) g_vel = g;
      if (md.not_vel[g] == os) g_pos = g;
    }
  }
  const int pol_nna = pl.n_non_angle, pol_na = pl.n_angle;
  // partially measurable system (MC_PILCO4PMS.apply_policy, MC_PILCO.py:808-906): the policy sees a measured state.  Thread (p, s)
  // produces the measurement of its own component; a velocity thread rebuilds the noisy position of its pair (same draw) and
  // carries the previous noisy position, noisy velocity and filtered velocity.
  const mcp_meas& ms = pl.meas;
  const bool pms = ms.n > 0;
  int pm_pos = -1, pm_vel = -1, pm_pairlane = lane0;
  double pm_std = 0.0;  // (looked up here with a uniform index: per-lane0 indexing of the by-value argument would spill it)
  if (pms && own) {
    for (int i = 0; i < ms.n; ++i) {
      if (ms.pos[i] == os) {
        pm_pos = i;
        pm_std = ms.std_pos[i];
      }
      if (ms.vel[i] == os) {
        pm_vel = i;
        pm_pairlane = op * S + ms.pos[i];
        pm_std = ms.std_pos[i];
      }
    }
  }
  double pm_prev_np = 0.0, pm_prev_nv = 0.0, pm_prev_mv = 0.0;
  int vel_of_pos = 0;
  for (int g = 0; g < G; ++g)
    if (own && md.not_vel[g] == os) vel_of_pos = md.vel[g];
  const double Ts = md.Ts;
  // resident groups of phase V (see matvec_rows): the GP-sharded launch of the headline shape keeps RF_NRES register groups of
  // its wave's share of Kinv for the whole rollout
  constexpr int NRES = (GSH && MAXDEG == 0 && P == 4) ? RF_NRES : 0;
  v2d vres[NRES > 0 ? NRES : 1][RF_GS];
  if (NRES > 0) lds_barrier();  // the chunk table (written by thread 0) is read by every wave
  const int nres = load_resident<NRES>(gpl, 0, tab, NC, wv0, lane0, vres);
  unsigned long long last_stamp = clock64();

  for (int t = 0; t < T; ++t) {
    // (Laundering the ids per step, as the 16-particle kernel and the backward sweep do to stay out of scratch, was measured
    //  here and dropped: 219 -> 189 VGPRs, but phases K / S / PHI pay for the recomputed decodes: +0.95 k cycles per step.)
    const int tid = tid0, wv = wv0, lane = lane0;
    // ---- phase S: publish x_t and everything derived from a single state component --------------
    double xm = xn;  // what the policy sees of this component
    if (pms && wv == 0) {
      const double xpair = __shfl(xn, pm_pairlane);  // all of wave 0 takes part (the state threads live in it)
      if (own) {
        const int pi = pm_pos >= 0 ? pm_pos : pm_vel;
        double npos = pm_pos >= 0 ? xn : xpair;
        if (pi >= 0 && t > 0) {
          const double nn = ms.pos_noise ? ms.pos_noise[((size_t)(t - 1) * M + om) * ms.n + pi] : philox_normal(nzl, om, t, pi, MCP_STREAM_POS);
          npos = fma(pm_std, nn, npos);
        }
        if (pm_pos >= 0) xm = npos;
        if (pm_vel >= 0) {
          if (t == 0) {
            pm_prev_nv = xn;
            pm_prev_mv = xn;
          } else {
            const double nv = (npos - pm_prev_np) / Ts;
            xm = (ms.b0 * nv + ms.b1 * pm_prev_nv - ms.a1 * pm_prev_mv) / ms.a0;
            pm_prev_nv = nv;
            pm_prev_mv = xm;
          }
          pm_prev_np = npos;
        }
      }
    }
    if (own) {
      double* xc = xs + cur * P * S;
      xc[op * S + os] = xn;
      if (ovalid) {
        if (writer) {
          a.states[((size_t)t * M + m0 + op) * S + os] = xn;
          if (pms) ms.meas[((size_t)t * M + m0 + op) * S + os] = xm;
        }
        if (is_bad(xn) || is_bad(xm)) bad |= MCP_STATUS_NAN;
      }
      double sn = 0.0, cs = 0.0;
      if (zi_ang >= 0 || pi_ang >= 0) sincos_fast(xn, &sn, &cs);
      double snm = sn, csm = cs;  // trig of the measured value (policy features)
      if (pms && pi_ang >= 0 && xm != xn) sincos_fast(xm, &snm, &csm);
      // GP input z = [x[not_angle], sin x[angle], cos x[angle], u]   (Model_learning.py:670-683)
      if (zi_plain >= 0) z[op * D + zi_plain] = xn;
      if (zi_ang >= 0) {
        z[op * D + nna + zi_ang] = sn;
        z[op * D + nna + na + zi_ang] = cs;
      }
      // policy features (Policy.py:326-333: [x_nonangle, COS, SIN];  :397-399: [x, x*_t - x])
      if (pl.kind == MCP_POLICY_ANGLES) {
        if (pi_plain >= 0) sf[op * PF + pi_plain] = xm;
        if (pi_ang >= 0) {
          sf[op * PF + pol_nna + pi_ang] = csm;
          sf[op * PF + pol_nna + pol_na + pi_ang] = snm;
        }
      } else if (pl.kind == MCP_POLICY_TRAJ) {
        sf[op * PF + os] = xm;
        sf[op * PF + S + os] = pl.target_traj[(size_t)t * S + os] - xm;
      } else {
        sf[op * PF + os] = xm;
      }
    }
    else if (tid >= 64 && tid < 64 + P * G && t < T - 1) {
      // wave 1 is idle here: draw the process noise of this step now instead of on phase F's critical path
      const int e = tid - 64, p = e / G, g = e - p * G;
      double ev = 0.0;
      if (a.particle_pred) {
        const int mm = imin(m0 + p, Mend - 1);
        ev = nzl.eps ? nzl.eps[((size_t)t * M + mm) * G + g] : philox_normal(nzl, mm, t, g);
      }
      epsb[e] = ev;
    }
    else if (tid >= 128 && drop && !nzl.masks) {
      // waves 2.. are idle here too: the step's dropout decisions, one Philox block per 4 basis functions (the integer
      // multiplies of a block cost more than the exp of the feature it gates; drawn per basis function in phase PHI they
      // were most of that phase)
      const int BQ = (B + 3) >> 2;
      for (int it = tid - 128; it < P * BQ; it += RF_NT - 128) {
        const int p = it / BQ, q = it - p * BQ;
        const u32x4 r = philox_draw(nzl, imin(m0 + p, Mend - 1), t, MCP_STREAM_MASK, (uint32_t)q);
        mk[it] = (int)(r.x >= drop_thr) | ((int)(r.y >= drop_thr) << 1) | ((int)(r.z >= drop_thr) << 2) | ((int)(r.w >= drop_thr) << 3);
      }
    }
    lds_barrier();
    RF_STAMP(0);
    // ---- phase PHI: phi_b = exp(-sum_q ((s_q - c_bq)/l_q)^2) * keep/(1-p) -------------------------
    double* ph = part;
    for (int itq = tid; itq < P * B; itq += RF_NT) {
      const int b = itq / P, p = itq - b * P;  // particle fastest: no division by a run-time value
      const int it = p * B + b;
      const double* cb = cen + b * PF;
      double dist = 0.0;
#pragma unroll 5
      for (int q = 0; q < PF; ++q) {
        double r = (sf[p * PF + q] - cb[q]) * invl[q];
        dist = fma(r, r, dist);
      }
      double phi = exp(-dist);
      if (drop) {
        int mm = imin(m0 + p, Mend - 1);
        bool keep = nzl.masks ? (nzl.masks[((size_t)t * M + mm) * B + b] != 0) : (((mk[p * ((B + 3) >> 2) + (b >> 2)] >> (b & 3)) & 1) != 0);
        phi = keep ? phi * keep_scale : 0.0;
      }
      ph[it] = phi;
    }
    lds_barrier();
    RF_STAMP(1);
    // ---- phase U: u = u_max tanh((W phi)/u_max), one wave per (particle, input) -------------------
    for (int task = wv; task < P * U; task += RF_NW) {
      int p = task / U, k = task - p * U;
      const double* wk = wgt + k * B;
      double s = 0.0;
#pragma unroll 4
      for (int b = lane; b < B; b += 64) s = fma(wk[b], ph[p * B + b], s);
      s = wave_sum(s);
      if (lane == 0) {
        s += bias_l[k];
        double um = umax_l[k];
        double u = pl.squash ? um * fast_tanh(s / um) : s;
        us[p * U + k] = u;
        z[p * D + nna + 2 * na + k] = u;
        if (m0 + p < Mend) {
          if (writer) a.inputs[((size_t)t * M + m0 + p) * U + k] = u;
          if (is_bad(u)) bad |= MCP_STATUS_NAN;
        }
      }
    }
    lds_barrier();
    RF_STAMP(2);
    if (t == T - 1) break;
    // ---- GPs, GB at a time ---------------------------------------------------------------------
    for (int g0 = 0; g0 < GL; g0 += GB) {
      const int gn = imin(GB, GL - g0);
      if (!GSH && GB < G) {
        lds_barrier();
        NC = build_chunk_table(gpl, g0, gn, tab, tid);
        lds_barrier();
      }
      phase_k<P, XLDS, MAXDEG>(gpl, kpar, g0, gn, D, NpadMax, z, xt_l, kb, ks, pa, pb, tid);
      lds_barrier();
      RF_STAMP(3);
      phase_v<P, NRES>(gpl, g0, tab, NC, NpadMax, kb, part, wv, lane, vres, nres);
      lds_barrier();
      RF_STAMP(4);
      RF_STAMP(5);
      phase_j<P, XLDS, MAXDEG>(gpl, g0, gn, D, NpadMax, xt_l, al_l, kb, ks, pa, pb, vb, tab, part, red, wv, lane,
                               (a.stamps && blockIdx.x == a.stamp_block && wv == 0) ? a.stamps : nullptr);
      lds_barrier();
      RF_STAMP(6);
      // ---- phase F: sample delta_g and fold the sampling into d delta/dz ------------------------
      for (int it = tid; it < gn * P * (D + 1); it += RF_NT) {
        int gl = it / (P * (D + 1));
        int r = it - gl * P * (D + 1);
        int p = r / (D + 1), c = r - p * (D + 1);
        const int g = GSH ? myg : g0 + gl;  // global GP index (noise, increments, Jacobian rows)
        const GpL& gp = gpl[g0 + gl];
        const double* kp = kpar + (g0 + gl) * KP_STRIDE(D);
        const double vscale = gp.var_scale;
        constexpr int NCOLMAX = P * RF_NAX(MAXDEG);
        const double* Rg = red + gl * RF_NW * (D + 1) * NCOLMAX;
        const double* zp = z + p * D;
        double mu, var;
        gp_point<MAXDEG>(gp, kp, D, zp, Rg, NCOLMAX, p, mu, var);
        var *= vscale;
        double eps = 0.0, wj = 0.0, sd = 0.0;
        if (a.particle_pred) {
          eps = epsb[p * G + g];
          sd = sqrt(var);
          wj = eps / (2.0 * sd);
        }
        if (c == D) {
          const double dv = a.particle_pred ? fma(sd, eps, mu) : mu;
          dl[p * G + g] = dv;
          if (GSH) {  // publish straight from the register: two granules
            const unsigned long long bits = (unsigned long long)__double_as_longlong(dv);
            gu64_t slot = (gu64_t)a.xch + xch_slot(gcluster, t, G, g, P) + 2 * p;
            store_granule(slot, (unsigned)t + 1u, (unsigned)bits);
            store_granule(slot + 1, (unsigned)t + 1u, (unsigned)(bits >> 32));
          }
          if (m0 + p < Mend) {
            if (a.particle_pred && var <= 0.0) bad |= MCP_STATUS_NONPOS_VAR;  // (finite and not positive: a NaN variance is MCP_STATUS_NAN, the retry case)
            if (is_bad(mu) || is_bad(var)) bad |= MCP_STATUS_NAN;
          }
        } else if (a.jac && m0 + p < Mend) {
          double Jmu, Jvar;
          gp_jac<MAXDEG>(gp, kp, D, zp, Rg, NCOLMAX, p, c, Jmu, Jvar);
          a.jac[(((size_t)t * M + m0 + p) * G + g) * D + c] = a.particle_pred ? fma(wj, Jvar * vscale, Jmu) : Jmu;
        }
      }
    }
    if (GSH && wv == 0) {
      // collect the other GPs' increments: lane -> (other GP, particle, half); every pass re-reads every granule
      const unsigned long long tx0_ = (a.stamps && blockIdx.x == a.stamp_block) ? clock64() : 0;
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
      if (a.stamps && blockIdx.x == a.stamp_block && lane == 0) a.stamps[8] += clock64() - tx0_;
    }
    lds_barrier();
    if (GSH && *abortw) {  // uniform (the barrier's memory clobber forces the re-read): a partner never arrived
      bad |= MCP_STATUS_SYNC;
      break;
    }
    RF_STAMP(7);
    // ---- integrate:  v' = v + delta ;  q' = q + Ts v + Ts/2 delta   (Model_learning.py:711-716) ----
    if (own) {
      const double* xc = xs + cur * P * S + op * S;
      double nx = 0.0;
      if (g_vel >= 0) nx = xc[os] + dl[op * G + g_vel];
      if (g_pos >= 0) nx = xc[os] + Ts * xc[vel_of_pos] + 0.5 * Ts * dl[op * G + g_pos];
      xn = nx;
    }
    cur ^= 1;  // x_{t+1} goes to the other buffer: no barrier between this read and the next write
  }
  if (bad) atomicOr(a.status, bad);
}

// ---------------------------------------------------------------------------------------
// single-step posterior (GP_prior.get_estimate_from_alpha) through the same phases
// ---------------------------------------------------------------------------------------
struct PostArgs {
  mcp_gp gp;
  int M, NCmax;
  const double* Z;
  double* mu;
  double* var;
  double* Jmu;
  double* Jvar;
  uint32_t* status;
};

template <int P>
__global__ __launch_bounds__(RF_NT) void posterior_fwd_kernel(PostArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const mcp_gp* gps = &a.gp;
  const mcp_gp& gp = a.gp;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = gp.kern.D, NpadMax = gp.Npad;
  const FwdLayout L = fwd_layout(P, 1, 1, D, 1, 1, 1, NpadMax, 2, 1, a.NCmax, false);
  double* z = smem + L.z;
  double* kb = smem + L.kb;
  double* ks = smem + L.ks;
  double* pa = smem + L.pa;
  double* pb = smem + L.pb;
  double* vb = smem + L.vb;
  double* part = smem + L.part;
  double* red = smem + L.red;
  int* tab = reinterpret_cast<int*>(smem + L.tab);
  GpL* gpl = reinterpret_cast<GpL*>(smem + L.gpl);
  double* kpar = smem + L.kpar;
  const int m0 = blockIdx.x * P;
  uint32_t bad = 0;
  for (int it = tid; it < P * D; it += RF_NT) {
    int p = it / D, d = it - p * D;
    z[it] = a.Z[(size_t)imin(m0 + p, a.M - 1) * D + d];
  }
  stage_gp_tables(gps, nullptr, 1, D, gpl, kpar, tid);
  __syncthreads();
  const int NC = build_chunk_table(gpl, 0, 1, tab, tid);
  __syncthreads();
  phase_k<P, false, 2>(gpl, kpar, 0, 1, D, NpadMax, z, nullptr, kb, ks, pa, pb, tid);
  __syncthreads();
  {
    v2d nores[1][RF_GS];
    phase_v<P, 0>(gpl, 0, tab, NC, NpadMax, kb, part, wv, lane, nores, 0);
  }
  __syncthreads();
  phase_j<P, false, 2>(gpl, 0, 1, D, NpadMax, nullptr, nullptr, kb, ks, pa, pb, vb, tab, part, red, wv, lane);
  __syncthreads();
  for (int it = tid; it < P * (D + 1); it += RF_NT) {
    int p = it / (D + 1), c = it - p * (D + 1);
    if (m0 + p >= a.M) continue;
    constexpr int NCOLMAX = P * RF_NAX(2);
    const double* zp = z + p * D;
    if (c == D) {
      double mu, var;
      gp_point<2>(gpl[0], kpar, D, zp, red, NCOLMAX, p, mu, var);
      a.mu[m0 + p] = mu;
      a.var[m0 + p] = var;
      if (is_bad(mu) || is_bad(var)) bad |= MCP_STATUS_NAN;
      if (var <= 0.0) bad |= MCP_STATUS_NONPOS_VAR;
    } else if (a.Jmu) {
      double Jm, Jv;
      gp_jac<2>(gpl[0], kpar, D, zp, red, NCOLMAX, p, c, Jm, Jv);
      a.Jmu[(size_t)(m0 + p) * D + c] = Jm;
      a.Jvar[(size_t)(m0 + p) * D + c] = Jv;
    }
  }
  if (bad && a.status) atomicOr(a.status, bad);
}

__global__ void posterior_bwd_kernel(int M, int D, const double* __restrict__ gmu, const double* __restrict__ gvar,
                                     const double* __restrict__ Jmu, const double* __restrict__ Jvar, double* __restrict__ gZ) {
  size_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * D) return;
  size_t m = i / D;
  gZ[i] = fma(gmu[m], Jmu[i], gvar[m] * Jvar[i]);
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
static int pick_particles_per_wg(int M) {
  // small swarms: spread over as many CUs as possible (every workgroup re-streams Kinv, so the
  // per-CU L2->L1 rate is the bound); large swarms: amortise the Kinv stream over more particles
  if (M <= 256) return 1;
  if (M <= 1024) return 2;
  return 16;  // falls back to 4 when the model does not fit the tile kernel
}

// What a call may be asked to do differently from the automatic dispatch, and what it reports back: carried by the call itself
// (mcp_dispatch, include/mcpilco_hip_debug.h) -- the library holds no dispatch state.  The values below are the ones the dispatch code reads.
// the automatic row split of the GP-sharded 16-particle kernel: parts and deal (experiment builds may change them: -DMCP_ROW_PARTS_DEFAULT=2 ...)
#ifndef MCP_ROW_PARTS_DEFAULT
#define MCP_ROW_PARTS_DEFAULT 3
#endif
#ifndef MCP_ROW_PART_MAJOR_DEFAULT
#define MCP_ROW_PART_MAJOR_DEFAULT 1
#endif
struct FwdHooks {
  int force_ppw = 0;     // particles per workgroup (0 = automatic)
  int force_xlds = -1;   // -1 automatic, 0 never stage small operands in LDS
  int force_gb = 0;      // GPs per pass (0 = as many as fit)
  int gp_sharding = -1;  // -1 automatic, 0 never, 1 whenever the grid fits the device
  int policy_split = -1; // -1 automatic, 0 every member evaluates the whole policy, 1 the split whenever the shape allows it
  int fwd_lean = -1;     // -1 / 1 the latency-lean kernel wherever it applies, 0 never
  int row_split = -1;    // -1 automatic, 0 one workgroup per (tile, GP range), 2 / 3 that many (row parts of Kinv) whenever the shape allows it
  int cluster_map = -1;  // -1 automatic, 0 the members of a tile on one XCD, 1 row part major (FwdArgs.gsh_map)
  unsigned long long* stamps = nullptr;
  unsigned stamp_block = 0;
  int last_ppw = 0, last_sharded = 0, last_lean = 0, last_row_split = 0;  // report
};
static FwdHooks fwd_hooks(const mcp_dispatch* d) {
  FwdHooks h;
  if (d) {
    h.force_ppw = d->fwd_particles;
    h.force_xlds = d->fwd_no_xlds ? 0 : -1;
    h.force_gb = d->fwd_gb;
    h.gp_sharding = d->gp_sharding == 1 ? 0 : (d->gp_sharding == 2 ? 1 : -1);
    h.policy_split = d->policy_split == 1 ? 0 : (d->policy_split == 2 ? 1 : -1);
    h.fwd_lean = d->fwd_lean == 1 ? 0 : -1;
    h.row_split = d->row_split == 1 ? 0 : ((d->row_split == 2 || d->row_split == 3) ? d->row_split : -1);
    h.cluster_map = d->cluster_map == 1 ? 0 : (d->cluster_map == 2 ? 1 : -1);
    h.stamps = (unsigned long long*)d->fwd_stamps;
    h.stamp_block = d->stamp_block;
  }
  return h;
}
static const int g_gp_max_launches = 2;  // a swarm goes out GP-sharded when it fits this many resident grids (cart-pole shape, forward ms,
                                         // tools/sweep_fwd_swarm.py: M=1024 two launches 4.9 vs 6.5 unsharded; M=1280 three launches 7.3 vs 6.9 on the tile kernel)
static int gsh_grid(int nclusters, int G) { return ((nclusters + 7) / 8) * 8 * G; }
// every workgroup of a GP-sharded grid waits for its partners, so the whole grid must be resident: one 512-thread
// workgroup per CU (the LDS footprint allows no more)
static int device_cu_count() {
  static int cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cus[dev] = n;
  }
  return cus[dev];
}

// Workgroups per tile with which the GP-sharded 16-particle kernel can take the whole swarm in one resident grid (0 = it cannot):
// the largest divisor of G that fits, i.e. the fewest GPs per workgroup (rollout_fwd_tile.hip: cart-pole and UR5 register classes)
static int tile_sharded_cluster(const mcp_model* m, const mcp_policy* p, int NpadMax, int M, int T) {
  if (m->G < 2 || T <= 1 || NpadMax > 512) return 0;
  if (!(m->D <= 24 && p->P <= 24 && m->U <= 6) || !fwd_tile_fits(m, p)) return 0;
  const int cus = device_cu_count(), ncl = (M + 15) / 16;
  for (int cs = m->G; cs >= 2; --cs)
    if (m->G % cs == 0 && ((ncl + 7) / 8) * 8 * cs <= cus) return cs;
  return 0;
}

static int chunks_in_pass(const mcp_model* m, int GB) {
  int best = 0;
  for (int g0 = 0; g0 < m->G; g0 += GB) {
    int nc = 0;
    for (int g = g0; g < m->G && g < g0 + GB; ++g) nc += (m->gp[g].Npad + RF_CW - 1) / RF_CW;
    best = imax(best, nc);
  }
  return best;
}

template <int P, bool XLDS, int MAXDEG, bool GSH>
static int launch_fwd_deg(const FwdArgs& a, size_t lds, hipStream_t st) {
  MCP_ENSURE_MAX_LDS(rollout_fwd_kernel<P, XLDS, MAXDEG, GSH>);
  const int grid = GSH ? gsh_grid(a.nclusters, a.model.G) : (a.M + P - 1) / P;
  hipLaunchKernelGGL((rollout_fwd_kernel<P, XLDS, MAXDEG, GSH>), dim3(grid), dim3(RF_NT), lds, st, a);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}
// the SE-only instantiation carries no polynomial code: a markedly smaller kernel (instruction-cache footprint)
template <int P, bool XLDS>
static int launch_fwd(const FwdArgs& a, size_t lds, hipStream_t st) {
  return a.maxdeg == 0 ? launch_fwd_deg<P, XLDS, 0, false>(a, lds, st) : launch_fwd_deg<P, XLDS, 2, false>(a, lds, st);
}
template <int P>
static int launch_fwd_sharded(const FwdArgs& a, size_t lds, hipStream_t st) {
  return a.maxdeg == 0 ? launch_fwd_deg<P, true, 0, true>(a, lds, st) : launch_fwd_deg<P, true, 2, true>(a, lds, st);
}

static int rollout_fwd_impl(const mcp_model* model, const mcp_policy* policy, const mcp_noise* noise, int M, int T, int particle_pred,
                            const double* x0, double* states, double* inputs, double* jac, uint32_t* status, void* workspace,
                            size_t workspace_bytes, void* stream, FwdHooks& hk) {
  if (!noise || !x0 || !states || !inputs || !status || !policy || M <= 0 || T <= 0) return MCP_ERR_ARG;
  const bool no_gp_sharding = (particle_pred & MCP_FWD_NO_GP_SHARDING) != 0;  // (the recovery path after MCP_STATUS_SYNC keeps its workspace)
  const int operands_packed = ((particle_pred & MCP_FWD_KT_PACKED) ? 1 : 0) | ((particle_pred & MCP_FWD_XJ_PACKED) ? 2 : 0);
  particle_pred &= 1;
  mcp_model stub;
  if (!model) {
    if (T != 1) return MCP_ERR_ARG;  // without a dynamics model only the policy can be evaluated
    stub = policy_only_model(policy);
    model = &stub;
  } else if (!model_ok(model)) {
    return MCP_ERR_ARG;
  }
  if (!policy_ok(policy, model->S, model->U, T)) return MCP_ERR_ARG;
  FwdArgs a;
  a.model = *model;
  a.pol = *policy;
  a.nz = *noise;
  a.M = M;
  a.T = T;
  a.particle_pred = particle_pred;
  a.operands_packed = operands_packed;
  a.NpadMax = 0;
  a.maxdeg = 0;
  for (int g = 0; g < model->G; ++g) {
    a.NpadMax = imax(a.NpadMax, model->gp[g].Npad);
    a.maxdeg = imax(a.maxdeg, model->gp[g].kern.poly_deg);
  }
  a.x0 = x0;
  a.states = states;
  a.inputs = inputs;
  a.jac = jac;
  a.status = status;
  a.stamps = hk.stamps;
  a.stamp_block = hk.stamp_block;
  a.xch = nullptr;
  a.xj = nullptr;
  a.xj_stride = 0;
  {
    const size_t xoff = rollout_xch_bytes(M, model->G), xb = rollout_xj_bytes(model);
    if (xb && workspace && workspace_bytes >= xoff + xb) {
      a.xj = (double*)((char*)workspace + xoff);
      a.xj_stride = (int)(xb / sizeof(double) / (size_t)model->G);
    }
  }
  a.kt = nullptr;
  a.kt_stride = 0;
  {
    const size_t koff = rollout_xch_bytes(M, model->G) + rollout_xj_bytes(model), kb_ = rollout_kt_bytes(model);
    if (kb_ && workspace && workspace_bytes >= koff + kb_) {
      a.kt = (const double*)((char*)workspace + koff);
      a.kt_stride = a.NpadMax * a.NpadMax;
    }
  }
  a.nclusters = 0;
  a.m_off = 0;
  a.m_cnt = M;
  a.gsh_cs = 0;
  a.uxch = nullptr;
  a.gsh_rs = 1;
  a.gsh_map = 0;
  a.rxch = nullptr;
  hipStream_t st = (hipStream_t)stream;
  // configuration search: most particles per workgroup first, operands in LDS if they fit, all GPs per pass if they fit
  int P0 = hk.force_ppw ? hk.force_ppw : pick_particles_per_wg(M);
  if (P0 != 1 && P0 != 2 && P0 != 4 && P0 != 16) return MCP_ERR_ARG;
  if (policy->meas.n > 0 && !policy->meas.meas) return MCP_ERR_ARG;
  // small swarms: shard the GPs of a particle cluster over G workgroups (each streams one Kinv) when the whole grid is
  // resident at one workgroup per CU; smallest cluster size first (most CUs busy)
  hk.last_sharded = 0;
  hk.last_lean = 0;
  hk.last_row_split = 0;
  if (hk.gp_sharding != 0 && !no_gp_sharding && model->G >= 2 && T > 1 && workspace && workspace_bytes >= rollout_xch_bytes(M, model->G) &&
      (hk.force_ppw == 0 || (hk.gp_sharding == 1 && hk.force_ppw != 16))) {
    const int cus = device_cu_count();
    int NC1 = 0;
    for (int g = 0; g < model->G; ++g) NC1 = imax(NC1, (model->gp[g].Npad + RF_CW - 1) / RF_CW);
    const bool forced = hk.force_ppw == 1 || hk.force_ppw == 2 || hk.force_ppw == 4;
    const bool tile_sh = tile_sharded_cluster(model, policy, a.NpadMax, M, T) > 0;
    a.xch = (unsigned long long*)workspace;
    a.GB = 1;
    a.NCmax = NC1;
    for (int P = forced ? hk.force_ppw : 1; P <= (forced ? hk.force_ppw : 4) && NC1 <= RF_MAX_CHUNKS; P <<= 1) {
      // particles one resident grid takes at this cluster size (whole groups of 8 clusters, G workgroups each)
      const int cap = (cus / (8 * model->G)) * 8 * P;
      if (cap <= 0) break;
      // up to g_gp_max_launches launches back to back (each resident on its own; they may overlap where one drains and the next
      // starts, which only delays a partner)
      const int nchunk = (M + cap - 1) / cap;
      if (nchunk > 1 && (P < 4 || nchunk > g_gp_max_launches || (!forced && tile_sh))) continue;  // (the sharded 16-particle kernel is the faster form then)
      FwdLayout L = fwd_layout(P, model->S, model->U, model->D, model->G, policy->P, policy->B, a.NpadMax, a.maxdeg, 1, NC1, true, 1);
      size_t lds = sizeof(double) * (size_t)L.total;
      bool lean = false;
      if (hk.fwd_lean != 0 && a.kt) {
        const size_t ll = fwd_lean_lds_bytes(model, policy, P, a.NpadMax, a.maxdeg);  // 0: the lean kernel does not take this shape
        if (ll > 0 && ll <= MCP_LDS_LIMIT) {
          lean = true;
          lds = ll;
        }
      }
      if (lds > MCP_LDS_LIMIT) break;
      if (hipMemsetAsync(workspace, 0, rollout_xch_bytes(M, model->G), st) != hipSuccess) return MCP_ERR_LAUNCH;
      if (lean && !(a.operands_packed & 1)) {  // Kinv of every GP as MFMA operand tiles, in each wave's streaming order (unless an earlier call left them)
        const int rcp = launch_fwd_lean_pack(a, st);
        if (rcp != MCP_OK) return rcp;
      }
      hk.last_ppw = P;
      hk.last_sharded = nchunk;
      hk.last_lean = lean ? 1 : 0;
      const int per = (((M + nchunk - 1) / nchunk + P - 1) / P) * P;  // particles per launch, whole clusters
      for (int off = 0; off < M; off += per) {
        a.m_off = off;
        a.m_cnt = imin(per, M - off);
        a.nclusters = (a.m_cnt + P - 1) / P;
        const int rc = lean ? launch_fwd_lean(a, P, lds, st)
                            : (P == 4 ? launch_fwd_sharded<4>(a, lds, st) : (P == 2 ? launch_fwd_sharded<2>(a, lds, st) : launch_fwd_sharded<1>(a, lds, st)));
        if (rc != MCP_OK) return rc;
      }
      return MCP_OK;
    }
  }
  if ((P0 == 16 || hk.force_ppw == 0) && hk.gp_sharding != 0 && !no_gp_sharding && workspace && workspace_bytes >= rollout_xch_bytes(M, model->G) &&
      tile_sharded_cluster(model, policy, a.NpadMax, M, T) > 0) {
    // swarms beyond one resident grid of the small-tile kernel, up to 2048 particles at two GPs: the 16-particle kernel GP-sharded --
    // twice the workgroups, each with one GP's contractions (tools/sweep_fwd_swarm.py, cart-pole shape, forward ms: M=1024 3.8 vs 4.9
    // for two small-tile launches vs 6.4 unsharded; M=2048 3.9 vs 6.8 for the unsharded 16-particle kernel)
    const int ncl = (M + 15) / 16;
    {
      a.xch = (unsigned long long*)workspace;
      a.nclusters = ncl;
      a.gsh_cs = tile_sharded_cluster(model, policy, a.NpadMax, M, T);
      if (hipMemsetAsync(workspace, 0, rollout_xch_bytes(M, model->G), st) != hipSuccess) return MCP_ERR_LAUNCH;
      {  // the policy split over the members of a cluster (every member needs a tile of 16 basis functions)
        const size_t uoff = rollout_xch_bytes(M, model->G) + rollout_xj_bytes(model) + rollout_kt_bytes(model);
        const size_t ub = (size_t)ncl * 2 * a.gsh_cs * 16 * model->U * 2 * sizeof(unsigned long long);
        // automatic: clusters of three or more on small swarms (the UR5 launch script's M = 200: six members, the policy 1/6 of the step;
        // ur5_script 11.2 -> 10.5 ms).  Not with two members -- the exchange costs what half a cart-pole policy does -- and not on large
        // swarms, whose halves run another cluster size: they would no longer reproduce the whole bit for bit (tests: *_full_size_properties)
        const bool want = hk.policy_split == 1 || (hk.policy_split < 0 && a.gsh_cs >= 3 && M <= 512);
        if (want && (policy->B + 15) / 16 >= a.gsh_cs && workspace_bytes >= uoff + rollout_uxch_bytes(M, model->G, model->U)) {
          a.uxch = (unsigned long long*)((char*)workspace + uoff);
          if (hipMemsetAsync(a.uxch, 0, ub, st) != hipSuccess) return MCP_ERR_LAUNCH;
        }
      }
      {  // Two workgroups per (tile, GP range), one per half of the rows of Kinv, when the grid is still resident at twice the size: the UR5
         // launch script's M = 200 is 13 tiles x 6 GPs = 78 workgroups on 256 CUs, each bound by ONE GP's phase V on ONE CU
         // (profiles/r04_ur5_script_stamps.txt: V 50 k of 112 k cycles per step).  Wide classes with the per-tile phase J only (degree <= 1).
        const size_t roff = rollout_xch_bytes(M, model->G) + rollout_xj_bytes(model) + rollout_kt_bytes(model) + rollout_uxch_bytes(M, model->G, model->U);
        const size_t rb = rollout_rxch_bytes(model, M);
        // Round 6: THREE row parts where three times the grid is resident (13 x 6 x 3 = 234), and the workgroups dealt row part major
        // (FwdArgs.gsh_map): one workgroup per CU, at most 32 per XCD -- the grid is rounded up to a multiple of 8, the XCDs take its blocks in turn.
        const bool can = a.xj && a.maxdeg <= 1 && a.NpadMax >= 128 && rb > 0 && workspace_bytes >= roff + rb;
        const int cus = device_cu_count();
        const int map = hk.cluster_map < 0 ? MCP_ROW_PART_MAJOR_DEFAULT : hk.cluster_map;
        auto resident = [&](int rs) { return map == 1 ? ((ncl * a.gsh_cs * rs + 7) / 8) * 8 <= cus : ((ncl + 7) / 8) * 8 * a.gsh_cs * rs <= cus && ((ncl + 7) / 8) * a.gsh_cs * rs <= cus / 8; };
        if (can && hk.row_split != 0) {
          const int want = hk.row_split < 0 ? MCP_ROW_PARTS_DEFAULT : hk.row_split;
          const int rs = (want >= 3 && resident(3)) ? 3 : (resident(2) ? 2 : 1);
          if (rs > 1) {
            a.gsh_rs = rs;
            a.gsh_map = map;
            a.rxch = (unsigned long long*)((char*)workspace + roff);
            if (hipMemsetAsync(a.rxch, 0, rb, st) != hipSuccess) return MCP_ERR_LAUNCH;
          }
        }
      }
      const int rc = launch_fwd_tile_sharded(a, st);
      if (rc == MCP_OK) {
        hk.last_ppw = 16;
        hk.last_sharded = 1;
        hk.last_row_split = a.gsh_rs > 1 ? a.gsh_rs : 0;
        return MCP_OK;
      }
      if (rc != MCP_ERR_LIMIT) return rc;
      a.xch = nullptr;
      a.uxch = nullptr;
      a.gsh_rs = 1;
      a.gsh_map = 0;
      a.rxch = nullptr;
      a.nclusters = 0;
    }
  }
  if (P0 == 16) {
    // large swarms: 16-particle tiles on the matrix cores (rollout_fwd_tile.hip) when the problem fits that kernel
    if (model->G >= 1 && T > 1 && fwd_tile_fits(model, policy)) {
      hk.last_ppw = 16;
      return launch_fwd_tile(a, st);
    }
    P0 = 4;
  }
  for (int P = P0; P >= 1; P >>= 1) {
    for (int xl = (hk.force_xlds == 0 ? 0 : 1); xl >= 0; --xl) {
      for (int GB = imax(1, (hk.force_gb > 0 ? imin(hk.force_gb, model->G) : model->G)); GB >= 1; --GB) {
        int NCmax = chunks_in_pass(model, GB);
        if (NCmax > RF_MAX_CHUNKS) continue;
        FwdLayout L = fwd_layout(P, model->S, model->U, model->D, model->G, policy->P, policy->B, a.NpadMax, a.maxdeg, GB, NCmax, xl != 0);
        size_t lds = sizeof(double) * (size_t)L.total;
        if (lds > MCP_LDS_LIMIT) continue;
        a.GB = GB;
        a.NCmax = NCmax;
        hk.last_ppw = P;
        if (P == 4) return xl ? launch_fwd<4, true>(a, lds, st) : launch_fwd<4, false>(a, lds, st);
        if (P == 2) return xl ? launch_fwd<2, true>(a, lds, st) : launch_fwd<2, false>(a, lds, st);
        return xl ? launch_fwd<1, true>(a, lds, st) : launch_fwd<1, false>(a, lds, st);
      }
    }
  }
  return MCP_ERR_LIMIT;
}

template <int P>
static int launch_post(const PostArgs& a, size_t lds, hipStream_t st) {
  MCP_ENSURE_MAX_LDS(posterior_fwd_kernel<P>);
  hipLaunchKernelGGL(posterior_fwd_kernel<P>, dim3((a.M + P - 1) / P), dim3(RF_NT), lds, st, a);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

extern "C" int mcp_rollout_fwd_ex(const mcp_model* model, const mcp_policy* policy, const mcp_noise* noise, int M, int T, int particle_pred,
                                  const double* x0, double* states, double* inputs, double* jac, uint32_t* status, void* workspace,
                                  size_t workspace_bytes, void* stream, mcp_dispatch* d) {
  FwdHooks hk = fwd_hooks(d);
  const int rc = rollout_fwd_impl(model, policy, noise, M, T, particle_pred, x0, states, inputs, jac, status, workspace, workspace_bytes, stream, hk);
  if (d) {
    d->ran_particles = hk.last_ppw;
    d->ran_gp_sharded = hk.last_sharded;
    d->ran_fwd_lean = hk.last_lean;
    d->ran_row_split = hk.last_row_split;
  }
  return rc;
}
extern "C" int mcp_rollout_fwd(const mcp_model* model, const mcp_policy* policy, const mcp_noise* noise, int M, int T, int particle_pred,
                               const double* x0, double* states, double* inputs, double* jac, uint32_t* status, void* workspace,
                               size_t workspace_bytes, void* stream) {
  return mcp_rollout_fwd_ex(model, policy, noise, M, T, particle_pred, x0, states, inputs, jac, status, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int mcp_posterior_fwd_ex(const mcp_gp* gp, int M, const double* Z, double* mu, double* var, double* Jmu, double* Jvar,
                                    uint32_t* status, void* stream, mcp_dispatch* d) {
  FwdHooks hk = fwd_hooks(d);
  if (!gp || !Z || !mu || !var || M <= 0) return MCP_ERR_ARG;
  if ((Jmu == nullptr) != (Jvar == nullptr)) return MCP_ERR_ARG;
  if (gp->kern.D <= 0 || gp->kern.D > MCP_MAX_GPDIM || gp->N <= 0 || gp->Npad < gp->N || (gp->Npad % 16) != 0) return MCP_ERR_ARG;
  if (gp->N > MCP_MAX_TRAIN) return MCP_ERR_LIMIT;
  if (!gp->Xt || !gp->X || !gp->alpha || !gp->Kinv || !gp->kern.inv_ls) return MCP_ERR_ARG;
  if (gp->kern.poly_deg < 0 || gp->kern.poly_deg > 2) return MCP_ERR_ARG;
  if (gp->kern.poly_deg >= 1 && (!gp->kern.w1 || !gp->aX)) return MCP_ERR_ARG;
  if (gp->kern.poly_deg >= 2 && (!gp->kern.w20 || !gp->kern.w21)) return MCP_ERR_ARG;
  PostArgs a;
  a.gp = *gp;
  a.M = M;
  a.NCmax = (gp->Npad + RF_CW - 1) / RF_CW;
  a.Z = Z;
  a.mu = mu;
  a.var = var;
  a.Jmu = Jmu;
  a.Jvar = Jvar;
  a.status = status;
  int P0 = hk.force_ppw ? hk.force_ppw : pick_particles_per_wg(M);
  if (P0 == 16) P0 = 4;  // (the single-step operator has no 16-particle form: more than 1024 test points run 4 per workgroup)
  if (P0 != 1 && P0 != 2 && P0 != 4) return MCP_ERR_ARG;
  for (int P = P0; P >= 1; P >>= 1) {
    FwdLayout L = fwd_layout(P, 1, 1, gp->kern.D, 1, 1, 1, gp->Npad, 2, 1, a.NCmax, false);
    size_t lds = sizeof(double) * (size_t)L.total;
    if (lds > MCP_LDS_LIMIT) continue;
    hipStream_t st = (hipStream_t)stream;
    if (d) d->ran_particles = P;
    if (P == 4) return launch_post<4>(a, lds, st);
    if (P == 2) return launch_post<2>(a, lds, st);
    return launch_post<1>(a, lds, st);
  }
  return MCP_ERR_LIMIT;
}
extern "C" int mcp_posterior_fwd(const mcp_gp* gp, int M, const double* Z, double* mu, double* var, double* Jmu, double* Jvar,
                                 uint32_t* status, void* stream) {
  return mcp_posterior_fwd_ex(gp, M, Z, mu, var, Jmu, Jvar, status, stream, nullptr);
}

extern "C" int mcp_posterior_bwd(int M, int D, const double* gmu, const double* gvar, const double* Jmu, const double* Jvar, double* gZ,
                                 void* stream) {
  if (!gmu || !gvar || !Jmu || !Jvar || !gZ || M <= 0 || D <= 0) return MCP_ERR_ARG;
  size_t n = (size_t)M * D;
  hipLaunchKernelGGL(posterior_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, D, gmu, gvar, Jmu, Jvar,
                     gZ);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}
