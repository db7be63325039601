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
          const double nn = ms.pos_noise ? ms.pos_noise[((size_t)(t - 1) * M + om) * ms.n + pi] : philox_normal(a.nz, om, t, pi, MCP_STREAM_POS);
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
        ev = a.nz.eps ? a.nz.eps[((size_t)t * M + mm) * G + g] : philox_normal(a.nz, mm, t, g);
      }
      epsb[e] = ev;
    }
    else if (tid >= 128 && drop && !a.nz.masks) {
      // waves 2.. are idle here too: the step's dropout decisions, one Philox block per 4 basis functions (the integer
      // multiplies of a block cost more than the exp of the feature it gates; drawn per basis function in phase PHI they
      // were most of that phase)
      const int BQ = (B + 3) >> 2;
      for (int it = tid - 128; it < P * BQ; it += RF_NT - 128) {
        const int p = it / BQ, q = it - p * BQ;
        const u32x4 r = philox_draw(a.nz, imin(m0 + p, Mend - 1), t, MCP_STREAM_MASK, (uint32_t)q);
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
        bool keep = a.nz.masks ? (a.nz.masks[((size_t)t * M + mm) * B + b] != 0) : (((mk[p * ((B + 3) >> 2) + (b >> 2)] >> (b & 3)) & 1) != 0);
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
// D <= 8 (<= 6 state-derived + <= 2 inputs), <= 6 policy features, 32 <= Npad <= 384, no measurement model; everything else runs the
// general kernel above.
#define RL_DSM 6  // state-derived GP-input dimensions (D - U), zero padded
#define RL_UM 2   // inputs, zero padded
#define RL_ZD (RL_DSM + RL_UM)
#define RL_PFM 6  // policy features, zero padded
#ifndef RL_NRES
#define RL_NRES 2  // register buffers kept resident: RL_NRES or RL_NRES + 1, whichever leaves an even number to stream
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
  static constexpr int red = eps + 2 * 4;                      // [RF_NW][8][8] phase-J partial tiles
  static constexpr int rt = red + RF_NW * 64;                  // [8][8] their sum (phase F)
  static constexpr int gpl = rt + 64;
  static constexpr int kpar = gpl + GPL_DOUBLES;
  static constexpr int role = ((kpar + 5 * RL_MAXD + 2) + 1) & ~1;          // [64][16] ints: roles of the threads of wave 0 in the serial section
  static constexpr int sro = role + 64 * 8;                    // [64][8]: phase S of thread (p, s): 6 int LDS addresses (in doubles) | 4 scale factors
  static constexpr int dump = sro + 64 * 8;                    // where phase S writes what a state component does not feed
  static constexpr int stl = dump + 2;                         // [16 + 8] u64 phase-cycle totals (diagnostic)
  static constexpr int end = stl + 24;
};
static_assert(LatFixed::red % 2 == 0 && LatFixed::rt % 2 == 0 && LatFixed::zs % 2 == 0 && LatFixed::role % 2 == 0 && LatFixed::sro % 2 == 0 && LatFixed::end % 2 == 0, "16-byte alignment of the v2d regions");
struct LatLayout {
  int gs, kb, vb, xe, xq, al, cen, wgt, mk, total;  // offsets in doubles
};
__host__ __device__ inline int lat_ng(int B) { return (B + 15) >> 4; }                 // groups of 16 basis functions
__host__ __device__ inline int lat_ngp(int B) { return ((lat_ng(B) + 7) >> 3) << 3; }  // padded to whole chunks of 8 (zeros)
__host__ __device__ inline LatLayout lat_layout(int P, int B, int Npad) {
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
  L.vb = take(Npad * P * 2);      // phase-J weights W[j][2p + a]
  L.xe = take(8 * Npad);          // [X^T; 1; 0] (phase J's A operand)
  L.xq = take(RL_ZD * Npad);
  L.al = take(Npad);
  L.cen = take(RL_PFM * Bp);
  L.wgt = take(RL_UM * Bp);
  L.mk = take((P * ((B + 3) / 4) + 1) / 2);
  L.total = o;
  return L;
}
// role table entries (ints) of a thread of wave 0
enum { RO_OP, RO_OS, RO_ZPLAIN, RO_ZANG, RO_PPLAIN, RO_PANG, RO_GVEL, RO_GPOS, RO_VELOFPOS, RO_PMPOS, RO_PMVEL, RO_PMPAIR, RO_FP, RO_FC, RO_OM, RO_N };

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
#define KT_NL 6
__host__ __device__ inline int kt_waves(int nrtt) { return nrtt >= 2 * RF_NW ? RF_NW : (nrtt >= 2 ? nrtt / 2 : 1); }
__host__ __device__ inline int kt_rt0(int nrtt, int w) {
  const int nw = kt_waves(nrtt);
  return w >= nw ? nrtt : (nrtt * w) / nw;
}
__global__ void kt_pack_kernel(mcp_model md, double* __restrict__ kt, int stride) {
  const mcp_gp& gp = md.gp[blockIdx.y];
  const int Npad = gp.Npad, nrtt = Npad >> 4, njg = Npad >> 3;
  double* out = kt + (size_t)blockIdx.y * stride;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < Npad * Npad; idx += gridDim.x * blockDim.x) {
    // idx = ((rt * njg + jg) * 64 + l) * 2 + h  in "row tile major" numbering; the destination re-orders the tiles per wave
    const int h = idx & 1, l = (idx >> 1) & 63, tile = idx >> 7;
    const int rt = tile / njg, jg = tile - rt * njg;
    int w = 0;
    while (w + 1 < RF_NW && kt_rt0(nrtt, w + 1) <= rt) ++w;
    const int r0 = kt_rt0(nrtt, w), nrt = kt_rt0(nrtt, w + 1) - r0;
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
template <int P>
__device__ __forceinline__ void kt_stream(gptr2_t p, int nrt, int njg, const double* kb, int lane, double (&acc3)[2][3], double (&acc2)[2][2],
                                          const v2d (&res)[RL_NRES + 1][KT_NL], int nres, v2d (&bufA)[KT_NL], v2d (&bufB)[KT_NL],
                                          int npre) {
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
  for (int r = 0; r < RL_NRES + 1; ++r) {
    if (r < nres) {  // wave-uniform
      v2d kR[3];
      kt_readk<P>(kR, ka, r * gpb);
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
__device__ __forceinline__ int kt_resident_count(int nb) {
  int nres = imin(RL_NRES, nb - 2);
  if (nres < 0) nres = 0;
  if ((nb - nres) & 1) nres += (nb - nres >= 3) ? 1 : -1;  // (the register array has RL_NRES + 1 slots; nb >= 2 always)
  return nres;
}
// the tail of phase V: v is complete in this wave, so it forms the two phase-J weights of its rows on the spot,
//   W[j][2p] = kse_j alpha_j,  W[j][2p+1] = kse_j v_j     (D lane = 16 i + 4 blk + p: row 16 rt + 4 blk + i, particle p)
template <int P>
__device__ __forceinline__ void kt_tail(double (&acc3)[2][3], double (&acc2)[2][2], int rt0, int nrt, const double* kb, const double* al_l,
                                        double* vb, int lane) {
  const int p = lane & 3;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    if (r < nrt && p < P) {
      const int row = 16 * (rt0 + r) + 4 * ((lane >> 2) & 3) + (lane >> 4);
      double v = acc3[0][r] + acc3[1][r];
      if (r < 2) v += acc2[0][r] + acc2[1][r];  // (the wiring this wave did not take left its accumulators at zero)
      const double kse = kb[row * P + p], alj = al_l[row];
      v2d w;
      w.x = kse * alj;
      w.y = kse * v;
      *reinterpret_cast<v2d*>(__builtin_assume_aligned(vb + 2 * (row * P + p), 16)) = w;
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
__device__ __forceinline__ void lean_j(const double* xe, const double* vb, double* red, int Npad, int j0, int nu, int wv, int lane) {
  const int kq = lane >> 4, blk = (lane >> 2) & 3, e = lane & 3;
  const double* ap = xe + (4 * (blk >> 1) + e) * Npad + j0 + kq;
  const double* bp = vb + (j0 + kq) * (2 * P) + 4 * (blk & 1) + e;  // (P < 4: columns >= 2P read a neighbour's weights; those output columns are never used)
  double acc0 = 0.0, acc1 = 0.0;
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
  // D lane = 16 i + 4 blk + j: row 4 (blk >> 1) + i, column 4 (blk & 1) + j
  red[wv * 64 + (4 * (blk >> 1) + kq) * 8 + 4 * (blk & 1) + e] = acc0 + acc1;
}

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
// at most).  One instruction = blocks (row half, column half) x 4 training points; the 8 waves split N (whole shares of `jper`
// points: the tables are zero beyond N), their partial tiles go to red[wave][8][8], ONE unconditional store per lane.
template <int P>
__device__ __forceinline__ void lean_j(const double* xe, const double* vb, double* red, int Np8, int jper, int wv, int lane) {
  const int kq = lane >> 4, blk = (lane >> 2) & 3, e = lane & 3;
  const double* ap = xe + (4 * (blk >> 1) + e) * Np8 + wv * jper + kq;
  const double* bp = vb + (wv * jper + kq) * (2 * P) + 4 * (blk & 1) + e;  // (P < 4: columns >= 2P read a neighbour's weights; those output columns are never used)
  const int nu = jper >> 2;
  double acc0 = 0.0, acc1 = 0.0;
  for (int u0 = 0; u0 < nu; u0 += 10) {
    double av[10], bw[10];
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int uc = imin(u0 + u, nu - 1);
      av[u] = ap[4 * uc];
      bw[u] = bp[4 * uc * (2 * P)];
    }
    asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5]), "+v"(av[6]), "+v"(av[7]), "+v"(av[8]), "+v"(av[9]),
                 "+v"(bw[0]), "+v"(bw[1]), "+v"(bw[2]), "+v"(bw[3]), "+v"(bw[4]), "+v"(bw[5]), "+v"(bw[6]), "+v"(bw[7]), "+v"(bw[8]), "+v"(bw[9]));
#pragma unroll
    for (int u = 0; u < 10; u += 2) {
      if (u0 + u < nu) mfma4(acc0, av[u], bw[u]);          // (uniform tests)
      if (u0 + u + 1 < nu) mfma4(acc1, av[u + 1], bw[u + 1]);
    }
  }
  // D lane = 16 i + 4 blk + j: row 4 (blk >> 1) + i, column 4 (blk & 1) + j
  red[wv * 64 + (4 * (blk >> 1) + kq) * 8 + 4 * (blk & 1) + e] = acc0 + acc1;
}

template <int P, int KR>
__global__ __launch_bounds__(RF_NT) void rollout_fwd_lat_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int LP = LatLog2<P>::v;
  const mcp_model& md = a.model;
  const mcp_policy& pl = a.pol;
  const int tid0 = threadIdx.x;
  const int S = md.S, U = md.U, G = md.G, D = md.D, B = pl.B, PF = pl.P, M = a.M, T = a.T;
  const int DS = D - U;
  const int Npad = a.NpadMax;
  const int NG = lat_ng(B), NGP = lat_ngp(B), Bp = NG * 16, BQ = (B + 3) >> 2;
  const LatLayout L = lat_layout(P, B, Npad);
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
  GpL* gpl = reinterpret_cast<GpL*>(smem + LatFixed::gpl);
  double* kpar = smem + LatFixed::kpar;
  int* role = reinterpret_cast<int*>(smem + LatFixed::role);
  unsigned long long* stl = reinterpret_cast<unsigned long long*>(smem + LatFixed::stl);  // phase-cycle totals (diagnostic)
  const bool stamping = a.stamps && blockIdx.x == a.stamp_block;
  double* gs = smem + L.gs;
  double* kb = smem + L.kb;
  double* vb = smem + L.vb;
  double* xe = smem + L.xe;
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
    for (int it = tid0; it < 8 * Npad; it += RF_NT) {  // [X^T; 1; 0]
      const int d = it / Npad, j = it - d * Npad;
      xe[it] = d < D ? (j < gp.Npad ? gp.Xt[(size_t)d * gp.Npad + j] : 0.0) : (d == D ? 1.0 : 0.0);
    }
    for (int it = tid0; it < Npad; it += RF_NT) al_l[it] = it < gp.Npad ? gp.alpha[it] : 0.0;
    for (int it = tid0; it < Npad * P * 2; it += RF_NT) vb[it] = 0.0;
  }
  for (int it = tid0; it < RL_UM * Bp; it += RF_NT) {
    const int k = it / Bp, b = it - k * Bp;
    wgt[it] = (k < U && b < B) ? pl.weight[(size_t)k * B + b] : 0.0;
  }
  for (int it = tid0; it < RL_UM * P * NGP; it += RF_NT) gs[it] = 0.0;  // (the padding groups stay zero)
  for (int it = tid0; it < KT_ZROWS * P; it += RF_NT) kb[gps_l[0].Npad * P + it] = 0.0;  // (zero rows behind THIS GP's k: phase V's out-of-range operands)
  for (int it = tid0; it < P * RL_ZD; it += RF_NT) zs[it] = 0.0;
  for (int it = tid0; it < P * RL_PFM; it += RF_NT) sf[it] = 0.0;
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
    ro[RO_PMPOS] = ro[RO_PMVEL] = ro[RO_PMPAIR] = 0;
    ro[RO_FP] = tid0 / (D + 1);
    ro[RO_FC] = tid0 % (D + 1);
    ro[RO_OM] = imin(m0 + op, Mend - 1);
    ro[RO_N] = 0;
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
    sd[0] = zA >= 0 ? il[zA] : 0.0;
    sd[1] = zB >= 0 ? il[zB] : 0.0;
    sd[2] = pA >= 0 ? pol[pA] : 0.0;
    sd[3] = pB >= 0 ? pol[pB] : 0.0;
  }
  for (int it = tid0; it < RL_PFM * Bp; it += RF_NT) {
    const int q = it / Bp, b = it - q * Bp;
    cen[it] = (q < PF && b < B) ? pl.centers[(size_t)b * PF + q] * pol[q] : 0.0;
  }
  for (int it = tid0; it < RL_ZD * Npad; it += RF_NT) {
    const int r = it / Npad, j = it - r * Npad;
    const int d = r < RL_DSM ? (r < DS ? r : -1) : (r - RL_DSM < U ? DS + r - RL_DSM : -1);
    xq[it] = (d >= 0) ? xe[d * Npad + j] * kpar[KP_INVLS(D) + d] : 0.0;
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
          ev = a.nz.eps ? a.nz.eps[((size_t)ts * M + mm) * G + myg] : philox_normal(a.nz, mm, ts, myg);
        }
        epsb[(ts & 1) * P + lane] = ev;
      }
    } else if (wv >= 2 && wv != 4 && drop && !a.nz.masks) {
      const int ti = (wv - 2 - (wv > 4 ? 1 : 0)) * 64 + lane;
      for (int it = ti; it < P * BQ; it += 5 * 64) {
        const int p = it / BQ, q = it - p * BQ;
        const u32x4 r = philox_draw(a.nz, imin(m0 + p, Mend - 1), ts, MCP_STREAM_MASK, (uint32_t)q);
        mk[it] = (int)(r.x >= drop_thr) | ((int)(r.y >= drop_thr) << 1) | ((int)(r.z >= drop_thr) << 2) | ((int)(r.w >= drop_thr) << 3);
      }
    }
  };
  draw_step(0, __builtin_amdgcn_readfirstlane(tid0 >> 6), tid0 & 63);
  lds_barrier();  // tables (incl. the chunk table written by thread 0) visible to every wave
  // this wave's share of Kinv: the row tiles [vrt0, vrt0 + vnrt) of its GP, as MFMA operand tiles in streaming order; the first
  // RL_NRES register buffers of the stream stay in registers for the whole rollout
  const int vnpad = __builtin_amdgcn_readfirstlane(gpl[0].Npad), vnjg = vnpad >> 3;
  int vrt0, vnrt;
  {
    const int w = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    vrt0 = kt_rt0(vnpad >> 4, w);
    vnrt = kt_rt0(vnpad >> 4, w + 1) - vrt0;
  }
  const gptr2_t vp = (gptr2_t)(a.kt + (size_t)myg * a.kt_stride + (size_t)vrt0 * vnjg * 128) + (tid0 & 63);
  const int vnt = vnrt * vnjg;  // tiles in this wave's stream
  v2d vres[RL_NRES + 1][KT_NL];  // (+ 1: a wave whose buffer count has the other parity keeps one more or one fewer)
  int nres = 0;
  if (vnrt > 0) {
    nres = kt_resident_count((vnt + KT_NL - 1) / KT_NL);
#pragma unroll
    for (int r = 0; r < RL_NRES + 1; ++r)
      if (r < nres) kt_load(vres[r], vp, r);
  }
  unsigned long long last_stamp = clock64(), sub_stamp = last_stamp;

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
      if (own) {
        const double* srow = smem + LatFixed::sro + lane * 8;
        const int4 sa = *reinterpret_cast<const int4*>(srow);
        const int2 sb = *reinterpret_cast<const int2*>(srow + 2);
        const v2d sc0 = *reinterpret_cast<const v2d*>(srow + 4), sc1 = *reinterpret_cast<const v2d*>(srow + 6);
        double* xc = xs + cur * P * S;
        xc[op * S + os] = xn;
        if (ovalid) {
          if (writer) a.states[((size_t)t * M + m0 + op) * S + os] = xn;
          if (is_bad(xn)) bad |= MCP_STATUS_NAN;
        }
        double sn = 0.0, cs = 0.0;
        if (zi_ang >= 0 || pi_ang >= 0) sincos_fast(xn, &sn, &cs);
        // GP input z = [x[not_angle], sin x[angle], cos x[angle], u]   (Model_learning.py:670-683), raw and divided by its lengthscale;
        // policy features (Policy.py:326-333: [x_nonangle, COS, SIN]; plain policy: x) divided by theirs: slots and factors from the table
        const double vz = zi_ang >= 0 ? sn : xn, vp = pi_ang >= 0 ? cs : xn;
        smem[sa.x] = vz;
        smem[sa.y] = cs;
        smem[sa.z] = vz * sc0.x;
        smem[sa.w] = cs * sc0.y;
        smem[sb.x] = vp * sc1.x;
        smem[sb.y] = sn * sc1.y;
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
    {
      const int row = tid >> 4, e16 = tid & 15;
      const int pP = row & (P - 1);  // particle of this DPP row in the policy phase (32 rows per round, 32 % P == 0)
      double sfr[RL_PFM], zr[RL_DSM];
#pragma unroll
      for (int q = 0; q < RL_PFM; ++q) sfr[q] = sf[pP * RL_PFM + q];
#pragma unroll
      for (int d = 0; d < RL_DSM; ++d) zr[d] = zs[pK * RL_ZD + d];
      if (t < T - 1) {
#pragma unroll
        for (int r = 0; r < KR; ++r) {
          // (the last round holds N P - (KR - 1) 512 items: at N = 300, P = 4 three of the eight waves; the others skip it -- wave-uniform)
          if (r == KR - 1 && r > 0 && ((wv * 64 + r * RF_NT) >> LP) >= Npad) {
            ds[r] = 0.0;
#pragma unroll
            for (int k = 0; k < RL_UM; ++k) xin[r][k] = 0.0;
            continue;
          }
          const int j = imin((tid + r * RF_NT) >> LP, Npad - 1);
          double xv[RL_DSM];
#pragma unroll
          for (int d = 0; d < RL_DSM; ++d) xv[d] = xq[d * Npad + j];
#pragma unroll
          for (int k = 0; k < RL_UM; ++k) xin[r][k] = xq[(RL_DSM + k) * Npad + j];
          double acc = 0.0;
#pragma unroll
          for (int d = 0; d < RL_DSM; ++d) {
            const double rr = zr[d] - xv[d];
            acc = fma(rr, rr, acc);
          }
          ds[r] = acc;
        }
      }
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
          if (a.nz.masks)
            kbits = (b < B && a.nz.masks[((size_t)t * M + imin(m0 + pP, Mend - 1)) * B + b] != 0) ? 1 : 0;
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
    }
    lds_barrier();  // B1
    RL_STAMP(1);
    // ---- u = u_max tanh((W phi + b) / u_max): every thread adds the group sums of its own particle in the same fixed order ----
    double ur[RL_UM];
#pragma unroll
    for (int k = 0; k < RL_UM; ++k) {
      ur[k] = 0.0;
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
#pragma unroll
        for (int k = 0; k < RL_UM; ++k) {
          const double rr = ur[k] - xin[r][k];
          dd = fma(rr, rr, dd);
        }
        const double kv = j < Nown ? lambda * exp(-dd) : 0.0;
        if (j < Npad) kb[it] = kv;
      }
    }
    lds_barrier();  // B2
    RL_STAMP(3);
    // ---- phase V, second half: v = Kinv k on the 4x4x4 MFMA, then the phase-J weights of this wave's rows ----
    const unsigned long long tv0_ = stamping ? clock64() : 0;
    if (vnrt > 0) {
      double acc3[2][3], acc2[2][2];
#pragma unroll
      for (int r = 0; r < 3; ++r) acc3[0][r] = acc3[1][r] = 0.0;
#pragma unroll
      for (int r = 0; r < 2; ++r) acc2[0][r] = acc2[1][r] = 0.0;
      kt_stream<P>(vp, vnrt, vnjg, kb, lane, acc3, acc2, vres, nres, bufA, bufB, RL_PRE);
      kt_tail<P>(acc3, acc2, vrt0, vnrt, kb, al_l, vb, lane);
      if (stamping && lane == 0) stl[16 + wv] += clock64() - tv0_;  // this wave's own phase V
      // ---- phase J over the rows this wave has just finished (wave-level ordering only) ----
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const unsigned long long tj0_ = (stamping && wv == 0) ? clock64() : 0;
      lean_j<P>(xe, vb, red, Npad, 16 * vrt0, 4 * vnrt, wv, lane);
      if (stamping && tid == 0) stl[12] += clock64() - tj0_;
    } else {
      red[wv * 64 + lane] = 0.0;  // (a wave without rows)
    }
    lds_barrier();  // B4
    RL_STAMP(6);
    if (wv == 0) {
      sub_stamp = stamping ? clock64() : 0;
      // ---- phase F: sample delta and fold the sampling into d delta/dz; hand-off; integrate ------------
      const int* ro = role + lane * 16;
      const int4 r0 = *reinterpret_cast<const int4*>(ro), r1 = *reinterpret_cast<const int4*>(ro + 4), r2 = *reinterpret_cast<const int4*>(ro + 8),
                 r3 = *reinterpret_cast<const int4*>(ro + 12);
      {  // the 8 waves' partial tiles, added in wave order: lane l -> element (c = l >> 3, n = l & 7)
        double rv[RF_NW];
#pragma unroll
        for (int w = 0; w < RF_NW; ++w) rv[w] = red[w * 64 + lane];
        double sr = rv[0];
#pragma unroll
        for (int w = 1; w < RF_NW; ++w) sr += rv[w];
        rtot[lane] = sr;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (lane < P * (D + 1)) {
        const int p = r3.x, c = r3.y;
        const GpL& gp = gpl[0];
        const double vscale = gp.var_scale;
        // R[D][2p] = sum_j k_j alpha_j,  R[D][2p+1] = k^T Kinv k;  R[c][.] the same sums weighted by X_jc
        const v2d RD = *reinterpret_cast<const v2d*>(__builtin_assume_aligned(rtot + D * 8 + 2 * p, 16));
        const v2d RC = *reinterpret_cast<const v2d*>(__builtin_assume_aligned(rtot + imin(c, D - 1) * 8 + 2 * p, 16));
        const double mu = gp.mean + RD.x;
        const double var = (gp.lambda - RD.y) * vscale;  // k(z,z) = lambda: Stationary_GP.py:172-181
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
          const double Jmu = -2.0 * il2 * fma(zc, RD.x, -RC.x);
          const double Jvar = 4.0 * il2 * fma(zc, RD.y, -RC.y);
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

static int g_force_ppw = 0;   // test hook: force particles per workgroup (0 = automatic)
static int g_force_xlds = -1; // test hook: -1 automatic, 0 never stage small operands in LDS, 1 = automatic
static int g_force_gb = 0;    // test hook: GPs per pass (0 = as many as fit)
static int g_last_ppw = 0;    // test hook: particles per workgroup of the last forward launch (16 = tile kernel)
extern "C" void mcp_debug_set_particles_per_wg(int p) { g_force_ppw = p; }
extern "C" int mcp_debug_last_particles_per_wg(void) { return g_last_ppw; }
extern "C" void mcp_debug_set_fwd_mode(int xlds, int gb) {
  g_force_xlds = xlds;
  g_force_gb = gb;
}
static unsigned long long* g_stamps = nullptr;  // diagnostic hook: device buffer of 16 u64 phase-cycle totals
extern "C" void mcp_debug_set_stamp_buffer(void* p) { g_stamps = (unsigned long long*)p; }
static unsigned g_stamp_block = 0;
extern "C" void mcp_debug_set_stamp_block(int b) { g_stamp_block = b > 0 ? (unsigned)b : 0u; }

// ---- GP-sharded launch: G workgroups per particle cluster ----------------------------------------------------------
static int g_gp_sharding = -1;  // test hook: -1 automatic, 0 never, 1 whenever the grid fits the device
static int g_last_sharded = 0;  // test hook: number of GP-sharded launches the last forward call made (0 = not sharded)
static int g_gp_max_launches = 2;  // a swarm goes out GP-sharded when it fits this many resident grids (cart-pole shape, forward ms,
                                   // tools/sweep_fwd_swarm.py: M=1024 two launches 4.9 vs 6.5 unsharded; M=1280 three launches 7.3 vs 6.9 on the tile kernel)
extern "C" void mcp_debug_set_gp_sharding(int mode) { g_gp_sharding = mode; }
extern "C" int mcp_debug_last_gp_sharded(void) { return g_last_sharded; }
static int g_fwd_lean = -1;  // test hook: -1 / 1 the latency-lean GP-sharded kernel wherever it applies, 0 never (the general one)
static int g_last_lean = 0;  // test hook: 1 when the last forward call ran the latency-lean kernel
extern "C" void mcp_debug_set_fwd_lean(int mode) { g_fwd_lean = mode; }
extern "C" int mcp_debug_last_fwd_lean(void) { return g_last_lean; }
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

// the latency-lean GP-sharded kernel: narrow SE-only models (cart-pole class); KR = phase-K items per thread
template <int P, int KR>
static int launch_fwd_lean_kr(const FwdArgs& a, size_t lds, hipStream_t st) {
  MCP_ENSURE_MAX_LDS(rollout_fwd_lat_kernel<P, KR>);
  hipLaunchKernelGGL((rollout_fwd_lat_kernel<P, KR>), dim3(gsh_grid(a.nclusters, a.model.G)), dim3(RF_NT), lds, st, a);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}
static int lean_items_per_thread(int P, int NpadMax) {  // 0: the shape has no instantiation
  if (NpadMax < 32 || NpadMax > 384) return 0;  // phase V deals 2 .. 24 row tiles of 16 to the waves in parts of 2 or 3
  return P == 4 ? 3 : (P == 2 ? 2 : 1);         // Npad * P <= KR * RF_NT
}
static bool lean_applies(const mcp_model* m, const mcp_policy* p, int P, int NpadMax, int maxdeg) {
  if (maxdeg != 0 || m->G < 2) return false;
  if (p->meas.n > 0 || p->kind == MCP_POLICY_TRAJ) return false;  // (measurement models and trajectory policies: the general kernel)
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
  if (P * m->S > 64 || P * (m->D + 1) > 64 || (m->G - 1) * P * 2 > 64 || m->D > RL_MAXD) return false;  // wave 0 carries the serial section
  return lean_items_per_thread(P, NpadMax) > 0;
}
static int launch_fwd_lean(const FwdArgs& a, int P, size_t lds, hipStream_t st) {
  if (P == 4) return launch_fwd_lean_kr<4, 3>(a, lds, st);
  if (P == 2) return launch_fwd_lean_kr<2, 2>(a, lds, st);
  return launch_fwd_lean_kr<1, 1>(a, lds, st);
}

extern "C" int mcp_rollout_fwd(const mcp_model* model, const mcp_policy* policy, const mcp_noise* noise, int M, int T, int particle_pred,
                               const double* x0, double* states, double* inputs, double* jac, uint32_t* status, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (!noise || !x0 || !states || !inputs || !status || !policy || M <= 0 || T <= 0) return MCP_ERR_ARG;
  const bool no_gp_sharding = (particle_pred & MCP_FWD_NO_GP_SHARDING) != 0;  // (the recovery path after MCP_STATUS_SYNC keeps its workspace)
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
  a.stamps = g_stamps;
  a.stamp_block = g_stamp_block;
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
  hipStream_t st = (hipStream_t)stream;
  // configuration search: most particles per workgroup first, operands in LDS if they fit, all GPs per pass if they fit
  int P0 = g_force_ppw ? g_force_ppw : pick_particles_per_wg(M);
  if (P0 != 1 && P0 != 2 && P0 != 4 && P0 != 16) return MCP_ERR_ARG;
  if (policy->meas.n > 0 && !policy->meas.meas) return MCP_ERR_ARG;
  // small swarms: shard the GPs of a particle cluster over G workgroups (each streams one Kinv) when the whole grid is
  // resident at one workgroup per CU; smallest cluster size first (most CUs busy)
  g_last_sharded = 0;
  g_last_lean = 0;
  if (g_gp_sharding != 0 && !no_gp_sharding && model->G >= 2 && T > 1 && workspace && workspace_bytes >= rollout_xch_bytes(M, model->G) &&
      (g_force_ppw == 0 || (g_gp_sharding == 1 && g_force_ppw != 16))) {
    const int cus = device_cu_count();
    int NC1 = 0;
    for (int g = 0; g < model->G; ++g) NC1 = imax(NC1, (model->gp[g].Npad + RF_CW - 1) / RF_CW);
    const bool forced = g_force_ppw == 1 || g_force_ppw == 2 || g_force_ppw == 4;
    const bool tile_sh = tile_sharded_cluster(model, policy, a.NpadMax, M, T) > 0;
    a.xch = (unsigned long long*)workspace;
    a.GB = 1;
    a.NCmax = NC1;
    for (int P = forced ? g_force_ppw : 1; P <= (forced ? g_force_ppw : 4) && NC1 <= RF_MAX_CHUNKS; P <<= 1) {
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
      if (g_fwd_lean != 0 && a.kt && lean_applies(model, policy, P, a.NpadMax, a.maxdeg)) {
        const LatLayout LL = lat_layout(P, policy->B, a.NpadMax);
        if (sizeof(double) * (size_t)LL.total <= MCP_LDS_LIMIT) {
          lean = true;
          lds = sizeof(double) * (size_t)LL.total;
        }
      }
      if (lds > MCP_LDS_LIMIT) break;
      if (hipMemsetAsync(workspace, 0, rollout_xch_bytes(M, model->G), st) != hipSuccess) return MCP_ERR_LAUNCH;
      if (lean) {  // Kinv of every GP as MFMA operand tiles, in each wave's streaming order
        hipLaunchKernelGGL(kt_pack_kernel, dim3(64, model->G), dim3(256), 0, st, *model, (double*)a.kt, a.kt_stride);
        MCP_LAUNCH_CHECK();
      }
      g_last_ppw = P;
      g_last_sharded = nchunk;
      g_last_lean = lean ? 1 : 0;
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
  if ((P0 == 16 || g_force_ppw == 0) && g_gp_sharding != 0 && !no_gp_sharding && workspace && workspace_bytes >= rollout_xch_bytes(M, model->G) &&
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
      const int rc = launch_fwd_tile_sharded(a, st);
      if (rc == MCP_OK) {
        g_last_ppw = 16;
        g_last_sharded = 1;
        return MCP_OK;
      }
      if (rc != MCP_ERR_LIMIT) return rc;
      a.xch = nullptr;
      a.nclusters = 0;
    }
  }
  if (P0 == 16) {
    // large swarms: 16-particle tiles on the matrix cores (rollout_fwd_tile.hip) when the problem fits that kernel
    if (model->G >= 1 && T > 1 && fwd_tile_fits(model, policy)) {
      g_last_ppw = 16;
      return launch_fwd_tile(a, st);
    }
    P0 = 4;
  }
  for (int P = P0; P >= 1; P >>= 1) {
    for (int xl = (g_force_xlds == 0 ? 0 : 1); xl >= 0; --xl) {
      for (int GB = imax(1, (g_force_gb > 0 ? imin(g_force_gb, model->G) : model->G)); GB >= 1; --GB) {
        int NCmax = chunks_in_pass(model, GB);
        if (NCmax > RF_MAX_CHUNKS) continue;
        FwdLayout L = fwd_layout(P, model->S, model->U, model->D, model->G, policy->P, policy->B, a.NpadMax, a.maxdeg, GB, NCmax, xl != 0);
        size_t lds = sizeof(double) * (size_t)L.total;
        if (lds > MCP_LDS_LIMIT) continue;
        a.GB = GB;
        a.NCmax = NCmax;
        g_last_ppw = P;
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

extern "C" int mcp_posterior_fwd(const mcp_gp* gp, int M, const double* Z, double* mu, double* var, double* Jmu, double* Jvar,
                                 uint32_t* status, void* stream) {
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
  int P0 = g_force_ppw ? g_force_ppw : pick_particles_per_wg(M);
  if (P0 == 16) P0 = 4;  // (the single-step operator has no 16-particle form: more than 1024 test points run 4 per workgroup)
  if (P0 != 1 && P0 != 2 && P0 != 4) return MCP_ERR_ARG;
  for (int P = P0; P >= 1; P >>= 1) {
    FwdLayout L = fwd_layout(P, 1, 1, gp->kern.D, 1, 1, 1, gp->Npad, 2, 1, a.NCmax, false);
    size_t lds = sizeof(double) * (size_t)L.total;
    if (lds > MCP_LDS_LIMIT) continue;
    hipStream_t st = (hipStream_t)stream;
    if (P == 4) return launch_post<4>(a, lds, st);
    if (P == 2) return launch_post<2>(a, lds, st);
    return launch_post<1>(a, lds, st);
  }
  return MCP_ERR_LIMIT;
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
