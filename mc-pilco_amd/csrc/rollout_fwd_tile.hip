// Fused Monte-Carlo particle rollout, forward pass, large-swarm variant for gfx950 (MI355X):
// 16 particles per workgroup, the two N-long contractions of every GP on the matrix cores.
//
// Same contract as rollout_fwd.hip (MC_PILCO.apply_policy, policy_learning/MC_PILCO.py:615-674; per step
// Policy.py:242-265 / 323-335 / 389-403, Model_learning.py:210-242 / 265-336 / 685-718, GP_prior.py:137-155) and the
// same outputs (states, inputs, d delta/dz, status).  What changes is the mapping: with >= ~1000 particles per GPU
// the small-tile kernel re-streams Kinv (720 KB per GP at N=300) for every 4 particles and every per-particle
// phase runs at a fraction of a wave.  Here a 512-thread workgroup owns a 16-particle tile:
//
//   pol u = u_max tanh(W (phi o mask)/u_max)    squared distances in the reference's expanded form: the cross term is a
//                                              [16 x PF] x [PF x 16] MFMA product per 16 basis functions; W phi is folded in
//   K   k[j][p] = k(z_p, X_j)                  same product against 16 training points per tile ([16 x D] x [D x 16]), also
//                                              for the bilinear forms of the polynomial kernel; 4 exp() per lane and tile
//   V   v = Kinv k  ([N x N] x [N x 16])       v_mfma_f64_16x16x4_f64: A = 32x4 panel of Kinv (one 16-byte load per
//                                              lane feeds two MFMAs: even / odd rows), B = k[j..j+3][0..15] from LDS;
//                                              32-row blocks are dealt to the 8 waves, accumulators stay in registers
//                                              and overwrite k in LDS once every wave is done reading it
//   J   R = [X^T;1] W  ((D+1) x N x 16*ncol)   the moment / Jacobian sums as one skinny matrix product per GP, the 8
//                                              waves split N, their partial tiles meet in LDS slots and are added once
//   F   mu, var, d mu/dz, d var/dz from R; sample; fold the sampling into d delta/dz; integrate
//
// GPs are processed one after the other (k and v panels of one GP: 2 x N x 18 doubles of LDS).  Polynomial kernel
// terms whose sum over the training set does not depend on v are contracted once per launch (sum_j alpha_j,
// sum_j alpha_j X_jc X_je), so the per-step product carries 2 / 3 / 5 weight columns per particle for
// SE / SE+P1 / SE+P2.  Sums are formed in a different order than in the small-tile kernel: results agree to
// rounding, not bit for bit.
#include "rollout_fwd_shared.h"

using namespace mcp;

typedef double v4d __attribute__((ext_vector_type(4)));

#define TL_PT 16       // particles per workgroup
#define TL_KR 18       // row pitch (doubles) of the k / v panels: 16 particles + 2 pad (bank spread of the phase-K stores)
#define TL_MAXTASK 4   // 32-row blocks of Kinv per wave (N <= 1024)
#define TL_VU 4        // 4-row steps per register batch in phase V (two batches in flight)
#define TL_JU 4        // 4-row steps per operand batch in phase J
#ifndef TL_JNB
#define TL_JNB 4       // 16-row operand batches in flight in the per-tile form of phase J (wide classes)
#endif
#define TL_NCOL(deg) ((deg) == 0 ? 2 : ((deg) == 1 ? 3 : 5))

struct TileLayout {
  int invl, xs, us, z, sf, dl, eps, ks, kv, qa, mup, gpl, kpar, scr, xt, al, total;
  int xl;      // training inputs X^T and alpha of every GP staged in LDS (xt, al; row pitch NpadMax)
  int ptile, upart;  // policy phase: per-wave phi tiles (alias the k / v panels, idle then, when those are large enough) and partial sums  // offsets in doubles
  int nslot;   // phase-J partial-tile slots in scr
  int vslots;  // phase-V partial (32x16) slots in scr (7 let all 8 waves share the remainder blocks)
};

__host__ __device__ inline TileLayout tile_layout(int S, int U, int D, int G, int PF, int NpadMax, int maxdeg, bool want_xl = false) {
  TileLayout L;
  int o = 0;
  auto take = [&](int n) {
    int r = o;
    o += (n + 1) & ~1;
    return r;
  };
  L.invl = take(PF);
  L.xs = take(2 * TL_PT * S);
  L.us = take(TL_PT * U);
  L.z = take(TL_PT * D);
  L.sf = take(TL_PT * PF);
  L.dl = take(TL_PT * G + 2);  // sampled increments | abort word of the GP-sharded launch
  L.eps = take(2 * TL_PT * G);  // process noise of steps t (read by phase F) and t+1 (drawn by idle threads of phase F)
  L.ks = take(NpadMax * TL_KR);
  L.kv = take(NpadMax * TL_KR);  // directly after ks: the phi buffer of the policy phase aliases both
  const int RT = (D + 1 + 15) / 16, CT = TL_NCOL(maxdeg);
  L.qa = take(maxdeg >= 2 ? G * D * D : 0);
  L.mup = take(RF_NW * TL_PT);  // per-wave partial sums of  sum_j alpha_j k_j  (the posterior mean, accumulated in phase K)
  L.gpl = take(G * GPL_DOUBLES);
  L.kpar = take(G * KP_STRIDE(D));
  const int slot = RT * CT * 256;
  const int xneed = G * (D + 1) * NpadMax + 4;  // X^T [G][D][NpadMax] | alpha [G][NpadMax]
  const bool panels_ok = RF_NW * slot <= 2 * NpadMax * TL_KR;  // phase J can park its 8 partial tiles in the dead k / v panels
  const int ptiles = RF_NW * 16 * 17;
  const bool pt_in_scr = 2 * NpadMax * TL_KR < ptiles;
  const int polneed = RF_NW * 16 + (pt_in_scr ? ptiles : 0) + RF_NW * TL_PT * U;  // exchange slots, [phi tiles,] partial sums
  auto scr_size = [&](int nslot, int avail) {
    int scr = nslot * slot;
    if (scr < 7 * 512 && 7 * 512 <= avail) scr = 7 * 512;
    if (scr < polneed) scr = polneed;  // per-wave exchange slots + policy partial sums
    return scr;
  };
  // with the panels as parking space ONE phase-J slot (the result) is enough: the LDS that frees takes the small operands of phases K and J
  const bool xl = want_xl && panels_ok && o + scr_size(1, MCP_LDS_LIMIT / 8 - o - xneed) + xneed <= MCP_LDS_LIMIT / 8;
  const int avail = MCP_LDS_LIMIT / 8 - o - (xl ? xneed : 0);
  int nslot = xl ? 1 : 8;
  while (nslot > 1 && nslot * slot > avail) nslot >>= 1;
  const int scr = scr_size(nslot, avail);
  L.nslot = nslot;
  L.vslots = scr / 512;
  L.scr = take(scr);
  L.ptile = pt_in_scr ? L.scr + RF_NW * 16 : L.ks;
  L.upart = L.scr + RF_NW * 16 + (pt_in_scr ? ptiles : 0);
  L.xl = xl ? 1 : 0;
  L.xt = xl ? take(G * D * NpadMax) : 0;
  L.al = xl ? take(G * NpadMax) : 0;
  L.total = o;
  return L;
}

#define TL_STAMP(k) RF_STAMP(k)

// ---------------------------------------------------------------------------------------
// phase K: k[j][p] = k(z_p, X_j) for the 16 particles and all training points of one GP
// ---------------------------------------------------------------------------------------
// The weighted squared distance in the reference's own expanded form (Stationary_GP.py:65-109)
//     dist[p][j] = sum_d (z_pd/l_d)^2 + sum_d (X_jd/l_d)^2 - 2 sum_d (z_pd/l_d^2) X_jd
// makes the cross term -- and the bilinear forms of the polynomial kernel, sum_d w_d z_pd X_jd -- a [16 x D] x [D x 16]
// product per 16 training points: v_mfma_f64_16x16x4_f64 with
//   A operand  lane (m = l&15, kk = l>>4) : weight_d * z_md,  d = 4 i + kk   (one register per group i of 4 features)
//   B operand  lane (kk = l>>4, n = l&15) : X[j0 + n][d = 4 i + kk]          (raw training inputs: shared by all products)
//   result     lane (kq = l>>4, n = l&15), register r : particle kq + 4 r, training point j0 + n
// so a lane finishes 4 particles of one training point: 4 exp() per MFMA group instead of one per 2*D multiply-adds.
// |z_p|^2 and |X_j|^2 are partial sums over the lane's own features, folded across the 4 feature lanes by two
// lane exchanges.  Tiles of 16 training points are dealt round-robin to the 8 waves, the B operands of the next tile
// are loaded (unconditionally) before the current tile is consumed.
template <int NDQ, typename PT>
__device__ __forceinline__ void tile_k_load(double (&bx)[NDQ], double& alj, PT Xt, PT al, int Npad, int D, int tile, int kk, int n) {
#pragma unroll
  for (int i = 0; i < NDQ; ++i) bx[i] = Xt[(size_t)imin(4 * i + kk, D - 1) * Npad + 16 * tile + n];
  alj = al[16 * tile + n];  // zero on the padding rows (unused, and optimised away, for SE-only models)
}
template <int MAXDEG, int NDQ>
struct TileKConst {
  double a_se[NDQ], ilq[NDQ];
  double a_p1[MAXDEG >= 1 ? NDQ : 1], a_A[MAXDEG >= 2 ? NDQ : 1], a_B[MAXDEG >= 2 ? NDQ : 1];
  double zz4[4], lam, w1D;
  int deg, N;
};
// EXACT: the GP's polynomial degree IS MAXDEG (the usual case: one kernel family for all GPs of a model), so the degree tests below are
// compile-time; with the run-time tests every operand step and every result row had a scalar branch, and the joins of those branches
// made the loop's counter waits conservative (the next tile's loads were waited for before the current tile's MFMAs).
template <int MAXDEG, int NDQ, bool EXACT>
__device__ __forceinline__ void tile_k_consume(const double (&bx)[NDQ], double alj, const TileKConst<MAXDEG, NDQ>& c, int tile, int kk, int n,
                                               double* ks, double* kv, double (&macc)[4]) {
  const int deg = EXACT ? MAXDEG : c.deg;
  double sxx = 0.0;
#pragma unroll
  for (int i = 0; i < NDQ; ++i) {
    const double t = c.ilq[i] * bx[i];  // ilq is 0 for the padding features
    sxx = fma(t, t, sxx);
  }
  const double xx = fold_kk(sxx);
  v4d Cse = (v4d){0.0, 0.0, 0.0, 0.0}, Cp1 = Cse, CA = Cse, CB = Cse;
#pragma unroll
  for (int i = 0; i < NDQ; ++i) {
    Cse = __builtin_amdgcn_mfma_f64_16x16x4f64(c.a_se[i], bx[i], Cse, 0, 0, 0);
    if (MAXDEG >= 1 && deg >= 1) Cp1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c.a_p1[i], bx[i], Cp1, 0, 0, 0);
    if (MAXDEG >= 2 && deg >= 2) {
      CA = __builtin_amdgcn_mfma_f64_16x16x4f64(c.a_A[i], bx[i], CA, 0, 0, 0);
      CB = __builtin_amdgcn_mfma_f64_16x16x4f64(c.a_B[i], bx[i], CB, 0, 0, 0);
    }
  }
  const int j = 16 * tile + n;
  const bool live = j < c.N;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double dist = (c.zz4[r] + xx) + Cse[r];
    double kse = c.lam * exp(-dist);
    double kt = kse;
    if (MAXDEG >= 1 && deg >= 1) {
      kt += Cp1[r] + c.w1D;
      if (MAXDEG >= 2 && deg >= 2) kt = fma(CA[r], CB[r], kt);
    }
    if (!live) kse = kt = 0.0;
    if (MAXDEG >= 1) macc[r] = fma(alj, kt, macc[r]);  // posterior mean  sum_j alpha_j k_j  (GP_prior.py:145), all kernel terms at once
    const int o = j * TL_KR + kk + 4 * r;
    ks[o] = kse;
    kv[o] = kt;
  }
}
// PT: where X^T and alpha come from -- global memory (gptr_t; row pitch = the GP's Npad) or their LDS copies (const double*; pitch NpadMax)
template <int MAXDEG, int NDQ, typename PT, bool EXACT = false>
__device__ __forceinline__ void tile_phase_k(const GpL& gp, PT Xt, PT al, int xpitch, const double* kp, int D, const double* z, double* ks, double* kv,
                                             double* wslot, double* mup, int wv, int lane, unsigned long long* dbg = nullptr) {
  unsigned long long tq0 = dbg ? clock64() : 0;
  const int Npad = __builtin_amdgcn_readfirstlane(gp.Npad);
  const int kk = lane >> 4, n = lane & 15;
  TileKConst<MAXDEG, NDQ> c;
  c.N = __builtin_amdgcn_readfirstlane(gp.N);
  c.deg = MAXDEG == 0 ? 0 : __builtin_amdgcn_readfirstlane(gp.deg);
  c.lam = gp.lambda;
  c.w1D = (MAXDEG >= 1 && c.deg >= 1) ? kp[KP_W1(D) + D] : 0.0;
  double szz = 0.0;
#pragma unroll
  for (int i = 0; i < NDQ; ++i) {
    const int d = 4 * i + kk;
    const bool dv = d < D;
    const double zv = dv ? z[n * D + d] : 0.0;
    const double il = dv ? kp[KP_INVLS(D) + d] : 0.0;
    const double il2z = il * il * zv;
    c.ilq[i] = il;
    c.a_se[i] = -2.0 * il2z;
    szz = fma(il2z, zv, szz);
    if (MAXDEG >= 1) c.a_p1[i] = (dv && c.deg >= 1) ? kp[KP_W1(D) + d] * zv : 0.0;
    if (MAXDEG >= 2) {
      c.a_A[i] = (dv && c.deg >= 2) ? kp[KP_W20(D) + d] * zv : 0.0;
      c.a_B[i] = (dv && c.deg >= 2) ? kp[KP_W21(D) + d] * zv : 0.0;
    }
  }
  szz = fold_kk(szz);
  // |z_p|^2 goes from the operand layout (lane = particle) to the result layout (4 particles per lane) through 16 doubles of
  // this wave's own LDS slot (same wave writes and reads: program order, no barrier)
  if (kk == 0) wslot[n] = szz;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int r = 0; r < 4; ++r) c.zz4[r] = wslot[kk + 4 * r];
  const int ntile = Npad >> 4;
  if (dbg && lane == 0) { unsigned long long now = clock64(); dbg[12] += now - tq0; tq0 = now; }
  double macc[4] = {0.0, 0.0, 0.0, 0.0};
  if (wv < ntile) {
    const int nt = (ntile - wv + RF_NW - 1) / RF_NW;
    double b0[NDQ], b1[NDQ], al0, al1;
    tile_k_load<NDQ>(b0, al0, Xt, al, xpitch, D, wv, kk, n);
    for (int sI = 0; sI + 1 < nt; sI += 2) {
      tile_k_load<NDQ>(b1, al1, Xt, al, xpitch, D, wv + RF_NW * (sI + 1), kk, n);
      tile_k_consume<MAXDEG, NDQ, EXACT>(b0, al0, c, wv + RF_NW * sI, kk, n, ks, kv, macc);
      tile_k_load<NDQ>(b0, al0, Xt, al, xpitch, D, wv + RF_NW * imin(sI + 2, nt - 1), kk, n);
      tile_k_consume<MAXDEG, NDQ, EXACT>(b1, al1, c, wv + RF_NW * (sI + 1), kk, n, ks, kv, macc);
    }
    if (nt & 1) tile_k_consume<MAXDEG, NDQ, EXACT>(b0, al0, c, wv + RF_NW * (nt - 1), kk, n, ks, kv, macc);
  }
  // this wave's share of the mean: sum over its 16 training-point lanes, one partial per particle.  (SE-only models read the
  // mean off the ones-row of phase J's product instead, which costs nothing there.)
#pragma unroll
  for (int r = 0; r < (MAXDEG >= 1 ? 4 : 0); ++r) {
    double v = macc[r];
    v += dpp_take<0x111, 0xf>(v);
    v += dpp_take<0x112, 0xf>(v);
    v += dpp_take<0x114, 0xf>(v);
    v += dpp_take<0x118, 0xf>(v);
    if (n == 15) mup[wv * TL_PT + kk + 4 * r] = v;
  }
  if (dbg && lane == 0) dbg[13] += clock64() - tq0;
}

// ---------------------------------------------------------------------------------------
// phase V: one 32-row block of v = Kinv k over the summation range [js, je)
//   A operand  lane (m = l&15, kk = l>>4) : Kinv[j0+kk][I0 + 2m], Kinv[j0+kk][I0 + 2m + 1]   (Kinv symmetric: row j0+kk)
//   B operand  lane (kk = l>>4, n = l&15) : k[j0+kk][n]
//   acc_e[r] / acc_o[r] : v[I0 + 2((l>>4)+4r) (+1)][n = l&15]
// ---------------------------------------------------------------------------------------
// one batch = TL_VU steps of 4 rows (16 rows of Kinv); every summation range is a multiple of 16 rows, so batches are
// never partial.  All loads are unconditional: with no branch (and no select) between issue and use the compiler keeps
// the next batch in flight behind the MFMAs of the current one.  In the last block of a GP whose Npad is not a
// multiple of 32 the lanes of the 8 missing row pairs read column 0 instead: MFMA rows are independent, those rows of
// the result are simply never stored.
__device__ __forceinline__ void tile_v_load(v2d (&A)[TL_VU], double (&B)[TL_VU], gptr2_t ap, size_t astep, const double* bp) {
#pragma unroll
  for (int u = 0; u < TL_VU; ++u) {
    A[u] = ap[(size_t)u * astep];
    B[u] = bp[u * 4 * TL_KR];
  }
}
__device__ __forceinline__ void tile_v_mfma(const v2d (&A)[TL_VU], const double (&B)[TL_VU], v4d& acc_e, v4d& acc_o) {
#pragma unroll
  for (int u = 0; u < TL_VU; ++u) {
    acc_e = __builtin_amdgcn_mfma_f64_16x16x4f64(A[u].x, B[u], acc_e, 0, 0, 0);
    acc_o = __builtin_amdgcn_mfma_f64_16x16x4f64(A[u].y, B[u], acc_o, 0, 0, 0);
  }
}

// DEPTH = register batches in flight: 2 (one lands while one is consumed) or 3 (TLX_VDEPTH3, cart-pole class: the registers are there) -- two
// waves share a SIMD's matrix pipe, so a batch's loads have the time of ONE batch of the wave's own MFMAs to land when the partner stalls too
template <int DEPTH>
__device__ __forceinline__ void tile_v_block(const double* Kinv, int Npad, int I0, int js, int je, const double* kv, int lane, v4d& acc_e, v4d& acc_o) {
  const int m = lane & 15, kk = lane >> 4;
  const int col = I0 + 2 * m;
  gptr2_t a0 = (gptr2_t)((gptr_t)Kinv + (size_t)kk * Npad + (col < Npad ? col : 0));  // row kk, this lane's column pair
  const size_t astep = (size_t)4 * Npad / 2;                                            // 4 rows, in v2d units (Npad is even)
  const size_t abatch = (size_t)TL_VU * astep;
  const int bbatch = TL_VU * 4 * TL_KR;
  const double* b0 = kv + kk * TL_KR + m;
  const int nb = (je - js) >> 4;  // batches
  const int bs = js >> 4;         // first batch
  v2d A0[TL_VU], A1[TL_VU];
  double B0[TL_VU], B1[TL_VU];
  if constexpr (DEPTH == 3) {
    v2d A2[TL_VU];
    double B2[TL_VU];
    const int bl = bs + nb - 1;  // past the end: reload the last batch (harmless) rather than branch
    tile_v_load(A0, B0, a0 + (size_t)bs * abatch, astep, b0 + bs * bbatch);
    const int b1s = imin(bs + 1, bl);
    tile_v_load(A1, B1, a0 + (size_t)b1s * abatch, astep, b0 + b1s * bbatch);
    int b = 0;
    for (; b + 2 < nb; b += 3) {
      const int c2 = bs + b + 2, c3 = imin(bs + b + 3, bl), c4 = imin(bs + b + 4, bl);
      tile_v_load(A2, B2, a0 + (size_t)c2 * abatch, astep, b0 + c2 * bbatch);
      tile_v_mfma(A0, B0, acc_e, acc_o);
      tile_v_load(A0, B0, a0 + (size_t)c3 * abatch, astep, b0 + c3 * bbatch);
      tile_v_mfma(A1, B1, acc_e, acc_o);
      tile_v_load(A1, B1, a0 + (size_t)c4 * abatch, astep, b0 + c4 * bbatch);
      tile_v_mfma(A2, B2, acc_e, acc_o);
    }
    if (b < nb) tile_v_mfma(A0, B0, acc_e, acc_o);
    if (b + 1 < nb) tile_v_mfma(A1, B1, acc_e, acc_o);
    return;
  }
  tile_v_load(A0, B0, a0 + (size_t)bs * abatch, astep, b0 + bs * bbatch);
  for (int b = 0; b + 1 < nb; b += 2) {
    const int b1 = bs + b + 1;
    tile_v_load(A1, B1, a0 + (size_t)b1 * abatch, astep, b0 + b1 * bbatch);
    tile_v_mfma(A0, B0, acc_e, acc_o);
    const int b2 = bs + imin(b + 2, nb - 1);  // past the end: reload the last batch (harmless) rather than branch
    tile_v_load(A0, B0, a0 + (size_t)b2 * abatch, astep, b0 + b2 * bbatch);
    tile_v_mfma(A1, B1, acc_e, acc_o);
  }
  if (nb & 1) tile_v_mfma(A0, B0, acc_e, acc_o);
}

// the block schedule of one GP: wave w takes blocks w, w+8, ... in full; the last (nblk mod 8) blocks are cut into
// `s` summation ranges so that all waves stay busy, part 0 of each collects the partial tiles of the others
struct VSched {
  int nfull, rem, s, Jp;
};
__device__ __forceinline__ VSched tile_v_sched(int Npad, int vslots) {
  VSched q;
  const int nblk = (Npad + 31) >> 5;
  q.nfull = nblk >> 3;
  q.rem = nblk & 7;
  q.s = q.rem ? 8 / q.rem : 1;
  if (q.rem) q.s = imin(q.s, 1 + vslots / q.rem);
  q.Jp = (((Npad + q.s - 1) / q.s) + 15) & ~15;  // ranges in whole 16-row batches (Npad is a multiple of 16)
  return q;
}

// ---------------------------------------------------------------------------------------
// phase J: R[c][kind][p] = sum_j Xe[c][j] W_kind[j][p] over this wave's share of j
//   kinds: 0 kse*alpha | 1 kse*v | (deg>=1) 2 v | (deg 2) 3 v*B  4 v*A      (A_j = sum_d w20_d z_d X_jd, B_j likewise with w21)
// ---------------------------------------------------------------------------------------
// Operands that come from global memory (rows of [X^T;1], alpha, and for degree 2 the X panel of the polynomial
// mini-product) are loaded one 16-row batch ahead, unconditionally (indices clamped; rows outside the wave's range
// get zero weights).  Degree 2:  A_j = sum_d w20_d z_pd X_jd  and  B_j  are themselves a [16 x D] x [D x 16] product per
// batch and come out of the matrix core in exactly the lane layout the weights are needed in
// (accumulator register r of lane (kk, p) = row j0 + 4r + kk).
template <int DEG, int NDQ>
struct TileJBatch {
  double a0[4], a1[4], al[4], xq[DEG >= 2 ? NDQ : 1];
};
// ONE_RT: the class has a single row tile of [X^T; 1] (D + 1 <= 16, the cart-pole class), so the second tile's loads and MFMAs are not
// compiled at all (as a run-time test they were a scalar branch per 4-row step).
template <int DEG, int NDQ, typename PT, bool ONE_RT>
__device__ __forceinline__ void tile_j_load(TileJBatch<DEG, NDQ>& b, PT Xt, PT al, int xpitch, int Npad, int D, int RT, int cc0, int cc1, int jb,
                                            int kk, int n) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = imin(jb + 4 * u + kk, Npad - 1);
    b.a0[u] = Xt[(size_t)cc0 * xpitch + j];
    b.a1[u] = (ONE_RT || RT > 1) ? Xt[(size_t)cc1 * xpitch + j] : 0.0;  // (ONE_RT: rows 4 .. 7 of the 4-row operand form, see tile_j_consume)
    b.al[u] = al[j];
  }
  if (DEG >= 2) {
    const int jr = imin(jb + n, Npad - 1);
#pragma unroll
    for (int i = 0; i < NDQ; ++i) b.xq[i] = Xt[(size_t)imin(4 * i + kk, D - 1) * xpitch + jr];
  }
}
template <int DEG, int NDQ, bool ONE_RT>
__device__ __forceinline__ void tile_j_consume(const TileJBatch<DEG, NDQ>& b, const double (&zwa)[NDQ], const double (&zwb)[NDQ], int D, int RT, int Npad,
                                               int jb, int j1, int kk, int n, const double* ks, const double* kv, v4d (&acc)[2][TL_NCOL(DEG)]) {
  constexpr int CT = TL_NCOL(DEG);
  double bv[4][CT], vv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const bool ok = jb + 4 * u < j1;  // wave-uniform
    const int j = imin(jb + 4 * u + kk, Npad - 1);
    const double kse = ks[j * TL_KR + n], v = kv[j * TL_KR + n];
    vv[u] = ok ? v : 0.0;
    bv[u][0] = ok ? kse * b.al[u] : 0.0;
    bv[u][1] = ok ? kse * v : 0.0;
    if (DEG >= 1) bv[u][2] = vv[u];
  }
  if (DEG >= 2) {
    v4d Aq = (v4d){0.0, 0.0, 0.0, 0.0}, Bq = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NDQ; ++i) {
      Aq = __builtin_amdgcn_mfma_f64_16x16x4f64(b.xq[i], zwa[i], Aq, 0, 0, 0);
      Bq = __builtin_amdgcn_mfma_f64_16x16x4f64(b.xq[i], zwb[i], Bq, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bv[u][3] = vv[u] * Bq[u];
      bv[u][4] = vv[u] * Aq[u];
    }
  }
  if constexpr (ONE_RT) {
    // Cart-pole class (D + 1 <= 8 rows of [X^T; 1]): the 16x16x4 product would compute 16 rows for the 7 that exist.  v_mfma_f64_4x4x4_4b
    // (four independent 4x4x4 products, one per group of 4 particles; lane l = 16 k + 4 blk + e: A_blk[i = e][k], B_blk[k][j = e], D at
    // lane 16 i + 4 blk + j) takes rows c = 0..3 and 4..7 as two A operands against the SAME B operand the wide form uses (lane (k, p) ->
    // W[j0 + k][p]), and its result lands exactly where registers 0 and 1 of the wide form's accumulator would be (lane (i, p) -> rows i
    // and 4 + i): 2 small instructions instead of one wide one per weight kind, at a quarter of the pipe time each -- and nothing
    // downstream (parking, the 8-way add, phase F) changes.
    const int ci = n & 3;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (jb + 4 * u >= j1) continue;  // (wave-uniform)
      const double av0 = ci < D ? b.a0[u] : (ci == D ? 1.0 : 0.0);
      const double av1 = 4 + ci < D ? b.a1[u] : (4 + ci == D ? 1.0 : 0.0);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        acc[0][ct].x = __builtin_amdgcn_mfma_f64_4x4x4f64(av0, bv[u][ct], acc[0][ct].x, 0, 0, 0);
        acc[0][ct].y = __builtin_amdgcn_mfma_f64_4x4x4f64(av1, bv[u][ct], acc[0][ct].y, 0, 0, 0);
      }
    }
    return;
  }
  const int c0 = n, c1 = 16 + n;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (jb + 4 * u >= j1) continue;  // (wave-uniform) rows past this wave's share carry zero weights: their MFMAs are not issued at all
    const double av0 = c0 < D ? b.a0[u] : (c0 == D ? 1.0 : 0.0);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[0][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av0, bv[u][ct], acc[0][ct], 0, 0, 0);
    if (!ONE_RT && RT > 1) {
      const double av1 = c1 < D ? b.a1[u] : (c1 == D ? 1.0 : 0.0);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[1][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av1, bv[u][ct], acc[1][ct], 0, 0, 0);
    }
  }
}
template <int DEG, int NDQ, typename PT, bool ONE_RT = false>
__device__ __forceinline__ void tile_phase_j(const GpL& gp, PT Xt, PT al, int xpitch, const double* kp, int D, const double* z, const double* ks,
                                             const double* kv, v4d (&acc)[2][TL_NCOL(DEG)], int RT, int wv, int lane,
                                             unsigned long long* dbg = nullptr) {
  constexpr int CT = TL_NCOL(DEG);
  unsigned long long tq0 = dbg ? clock64() : 0;
  const int Npad = __builtin_amdgcn_readfirstlane(gp.Npad);
  const int per = ((Npad + RF_NW * 4 - 1) / (RF_NW * 4)) * 4;
  const int j0 = imin(wv * per, Npad), j1 = imin(Npad, j0 + per);
  const int kk = lane >> 4, n = lane & 15;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (v4d){0.0, 0.0, 0.0, 0.0};
  double zwa[NDQ], zwb[NDQ];
#pragma unroll
  for (int i = 0; i < NDQ; ++i) {
    const int d = 4 * i + kk;
    const bool dv = DEG >= 2 && d < D;
    const double zz = dv ? z[n * D + d] : 0.0;
    zwa[i] = dv ? kp[KP_W20(D) + d] * zz : 0.0;
    zwb[i] = dv ? kp[KP_W21(D) + d] * zz : 0.0;
  }
  const int cc0 = ONE_RT ? imin(n & 3, D - 1) : imin(n, D - 1), cc1 = ONE_RT ? imin(4 + (n & 3), D - 1) : imin(16 + n, D - 1);
  const int nbat = (j1 - j0 + 15) >> 4;
  if (dbg && lane == 0) { unsigned long long now = clock64(); dbg[14] += now - tq0; tq0 = now; }
  if constexpr (DEG >= 2 && NDQ > 2) {
    // wide classes with the degree-2 kernel: one operand batch at a time.  Two in flight (72 VGPRs beside 10 accumulator tiles and the 12
    // factor operands) put 38-95 registers of these instantiations into scratch, reloaded inside this loop.
    TileJBatch<DEG, NDQ> b0;
    for (int b = 0; b < nbat; ++b) {
      tile_j_load<DEG, NDQ, PT, ONE_RT>(b0, Xt, al, xpitch, Npad, D, RT, cc0, cc1, j0 + 16 * b, kk, n);
      tile_j_consume<DEG, NDQ, ONE_RT>(b0, zwa, zwb, D, RT, Npad, j0 + 16 * b, j1, kk, n, ks, kv, acc);
    }
    if (dbg && lane == 0) dbg[15] += clock64() - tq0;
    return;
  }
  TileJBatch<DEG, NDQ> b0, b1;
  tile_j_load<DEG, NDQ, PT, ONE_RT>(b0, Xt, al, xpitch, Npad, D, RT, cc0, cc1, j0, kk, n);
  for (int b = 0; b + 1 < nbat; b += 2) {
    const int ja = j0 + 16 * b;
    tile_j_load<DEG, NDQ, PT, ONE_RT>(b1, Xt, al, xpitch, Npad, D, RT, cc0, cc1, ja + 16, kk, n);
    tile_j_consume<DEG, NDQ, ONE_RT>(b0, zwa, zwb, D, RT, Npad, ja, j1, kk, n, ks, kv, acc);
    tile_j_load<DEG, NDQ, PT, ONE_RT>(b0, Xt, al, xpitch, Npad, D, RT, cc0, cc1, j0 + 16 * imin(b + 2, nbat - 1), kk, n);
    tile_j_consume<DEG, NDQ, ONE_RT>(b1, zwa, zwb, D, RT, Npad, ja + 16, j1, kk, n, ks, kv, acc);
  }
  if (nbat & 1) tile_j_consume<DEG, NDQ, ONE_RT>(b0, zwa, zwb, D, RT, Npad, j0 + 16 * (nbat - 1), j1, kk, n, ks, kv, acc);
  if (dbg && lane == 0) dbg[15] += clock64() - tq0;
}

// Wide classes (D + 1 > 16: two row tiles), degree <= 1: ONE output tile (row tile, weight column) per wave over ALL of j, so there are no
// partial tiles to bring together -- no parking, no add, no second barrier -- and the only operands a wave streams are its own 16 rows of
// [X^T; 1] (global, four 16-row batches in flight) and the panel values of its weight kind.  RT x CT = 4 or 6 tiles keep 4 or 6 of the 8 waves
// busy for Npad / 4 MFMAs each (two accumulators, even / odd steps); at the UR5 shape the split-j form spent 17 k cycles per GP in its
// batch loop (58 % of its MFMA time) plus 5.5 k in the finish, this one ~13 k in all.
// (the weight kind KIND is a template parameter of the loop: as a run-time switch per operand it became a tree of scalar branches whose
//  joins made every counter wait conservative -- 188 s_waitcnt in the loop, a memory round trip per step)
// The A operand comes from a packed copy of [X^T; 1] that the launch function builds in the caller's workspace (tile_xj_pack_kernel):
// per GP and row tile, for every PAIR of 4-row steps the 64 lanes' two operand values side by side, so one dwordx4 per lane = 1 KB
// contiguous per wave feeds two MFMAs.  Read from X^T (16 pieces of 32 B per step) or from row-major X (4 runs of 128 B) the loop was
// bound by the number of cache lines its loads touch (23 k cycles per GP either way, deeper prefetch made it worse).
template <int KIND>
__device__ __forceinline__ v4d tile_j_bytile_run(gptr2_t xa0, int bfirst, int nbat, unsigned okv, int lane, const double* ks0, const double* kv0) {
  constexpr int NB = TL_JNB;
  const gptr2_t xa = xa0 + (size_t)bfirst * 128;  // batches [bfirst, bfirst + nbat) of the GP
  const double* ks = ks0 + bfirst * 16 * TL_KR;
  const double* kv = kv0 + bfirst * 16 * TL_KR;
  v2d A[NB][2];
#pragma unroll
  for (int s = 0; s < NB; ++s) {
    const int b = imin(s, nbat - 1);
    A[s][0] = xa[(2 * b) * 64 + lane];
    A[s][1] = xa[(2 * b + 1) * 64 + lane];
  }
  v4d acc0 = (v4d){0.0, 0.0, 0.0, 0.0}, acc1 = (v4d){0.0, 0.0, 0.0, 0.0};
  auto consume = [&](const v2d (&Au)[2], int b) {
    const double* kb_ = ks + b * 16 * TL_KR;
    const double* vb_ = kv + b * 16 * TL_KR;
    double w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (KIND == 0) w[u] = kb_[okv + 4 * u * TL_KR];  // (alpha_j sits in the operand copy this kind reads)
      if (KIND == 1) w[u] = kb_[okv + 4 * u * TL_KR] * vb_[okv + 4 * u * TL_KR];
      if (KIND == 2) w[u] = vb_[okv + 4 * u * TL_KR];
    }
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Au[0].x, w[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Au[0].y, w[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Au[1].x, w[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Au[1].y, w[3], acc1, 0, 0, 0);
  };
  // whole groups of NB batches without a branch inside; the last nbat mod NB batches afterwards
  const int nfull = nbat / NB;
  for (int g4 = 0; g4 < nfull; ++g4) {
#pragma unroll
    for (int s = 0; s < NB; ++s) {
      const int b = g4 * NB + s;
      consume(A[s], b);
      const int bn = imin(b + NB, nbat - 1);
      A[s][0] = xa[(2 * bn) * 64 + lane];
      A[s][1] = xa[(2 * bn + 1) * 64 + lane];
    }
  }
#pragma unroll
  for (int s = 0; s < NB; ++s)
    if (nfull * NB + s < nbat) consume(A[s], nfull * NB + s);
  return acc0 + acc1;
}
template <int DEG>
__device__ __forceinline__ void tile_phase_j_bytile(const GpL& gp, const double* xj_g, int npb, const double* ks, const double* kv, double* scr, int RT,
                                                    int nh, int wv, int lane, unsigned long long* dbg = nullptr, int jb0 = 0, int jb1 = 1 << 30) {
  static_assert(DEG <= 1, "degree 2 keeps the split-j form (its mini-product would be repeated per weight kind)");
  constexpr int CT = TL_NCOL(DEG);
  unsigned long long tq0 = dbg ? clock64() : 0;
  const int Npad = __builtin_amdgcn_readfirstlane(gp.Npad);
  const int kk = lane >> 4, n = lane & 15;
  const int nbat = Npad >> 4;  // (Npad is a multiple of 16; padded rows carry alpha = v = 0)
  const unsigned okv = (unsigned)(kk * TL_KR + n);
  // work items = (tile, j-half): RT x CT = 6 tiles are 12 items, wave w takes items w and w + 8, so that every SIMD (waves s, s + 4) gets three
  // half tiles; the two partial tiles of a tile go to scratch slots 0 / 1 and phase F adds them as it reads (nh = 1: whole tiles, one slot)
  const int ntile = RT * CT, slot = ntile * 256;
  // (row-split cluster: the batches [jb0, jb1) of my half of the rows only)
  const int bbeg = imin(jb0, nbat), bend = imin(jb1, nbat);
  const int bh = (bend - bbeg + nh - 1) / nh;
  for (int item = wv; item < nh * ntile; item += RF_NW) {
    const int hh = item / ntile, tile = item - hh * ntile;
    const int rt = tile / CT, ct = tile - rt * CT;  // wave-uniform
    const int bfirst = bbeg + hh * bh, nb = imin(bend, bfirst + bh) - bfirst;
    // variant 0: [X^T; 1]   variant 1: its columns scaled by alpha_j (the weight kind kse alpha then needs no alpha of its own)
    const gptr2_t xa = (gptr2_t)((gptr_t)xj_g + (size_t)((ct == 0 ? 2 : 0) + rt) * npb * 128);
    v4d r = (v4d){0.0, 0.0, 0.0, 0.0};
    if (nb <= 0) {
    } else if (ct == 0)
      r = tile_j_bytile_run<0>(xa, bfirst, nb, okv, lane, ks, kv);
    else if (ct == 1)
      r = tile_j_bytile_run<1>(xa, bfirst, nb, okv, lane, ks, kv);
    else
      r = tile_j_bytile_run<2>(xa, bfirst, nb, okv, lane, ks, kv);
#pragma unroll
    for (int i = 0; i < 4; ++i) scr[hh * slot + tile * 256 + i * 64 + lane] = r[i];
  }
  if (dbg && lane == 0) dbg[15] += clock64() - tq0;
}
// [X^T; 1] of every GP in the operand order of tile_j_bytile_run: element ((var * 2 + rt) * npb + q) * 128 + lane * 2 + h  holds row
// c = 16 rt + (lane & 15) of step 2 q + h, i.e. training point j = 4 (2 q + h) + (lane >> 4):  X[j][c] for c < D, 1 for c == D, 0 beyond (and
// for j >= Npad); variant 1 holds the same times alpha_j.
__global__ __launch_bounds__(256) void tile_xj_pack_kernel(mcp_model model, double* xj, int xj_stride, int npb) {
  const int g = blockIdx.y, rt = blockIdx.z & 1, var = blockIdx.z >> 1;
  const mcp_gp& gp = model.gp[g];
  const int D = model.D;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < npb * 128; e += gridDim.x * 256) {
    const int h = e & 1, lane = (e >> 1) & 63, q = e >> 7;
    const int c = 16 * rt + (lane & 15), j = 4 * (2 * q + h) + (lane >> 4);
    double v = 0.0;
    if (j < gp.Npad) {
      v = c < D ? gp.X[(size_t)j * D + c] : (c == D ? 1.0 : 0.0);
      if (var) v *= gp.alpha[j];
    }
    xj[(size_t)g * xj_stride + (size_t)(var * 2 + rt) * npb * 128 + e] = v;
  }
}

// The 8 waves' partial tiles meet in `nslot` (1, 2, 4 or 8, whatever fits the LDS) slots: while more waves than slots hold a
// partial, the upper half hands its tiles to the lower half through the slots (fixed pairing, fixed order); the
// remaining min(8, nslot) partials stay in the slots and phase F adds them as it reads.  With 8 slots: no exchange at all.
// Only the accumulator registers that hold rows c <= D are parked and added (row = (lane >> 4) + 4 r: with D + 1 = 7 rows of
// [X^T; 1] registers 2 and 3 of every tile are never read); the slot layout itself is unchanged.
__device__ __forceinline__ int tile_j_rmax(int D, int rt) { return (imin(D - 16 * rt, 15)) >> 2; }
template <int CT>
__device__ __forceinline__ void tile_j_store(const v4d (&acc)[2][CT], int RT, int D, double* s, int lane) {
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
    if (rt < RT) {
      const int rmax = tile_j_rmax(D, rt);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r <= rmax) s[(rt * CT + ct) * 256 + r * 64 + lane] = acc[rt][ct][r];
    }
}
template <int CT>
__device__ __forceinline__ int tile_j_reduce(v4d (&acc)[2][CT], int RT, int D, double* scr, int nslot, int wv, int lane) {
  const int slot = RT * CT * 256;
  int active = RF_NW;
  for (; active > nslot; active >>= 1) {
    const int half = active >> 1;
    for (int base = 0; base < half; base += nslot) {
      // writers: waves half+base .. half+base+nslot-1 ; readers: waves base .. base+nslot-1
      const int wi = wv - half - base, ri = wv - base;
      if (wi >= 0 && wi < nslot && wv < active) tile_j_store<CT>(acc, RT, D, scr + wi * slot, lane);
      lds_barrier();
      if (ri >= 0 && ri < nslot && ri + base < half) {
        const double* s = scr + ri * slot;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          if (rt < RT) {
            const int rmax = tile_j_rmax(D, rt);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (r <= rmax) acc[rt][ct][r] += s[(rt * CT + ct) * 256 + r * 64 + lane];
          }
      }
      lds_barrier();
    }
  }
  if (wv < active) tile_j_store<CT>(acc, RT, D, scr + wv * slot, lane);
  return active;  // partials left in the slots
}

// Brings the 8 waves' partial tiles of phase J together: result (fixed summation order) in slot 0 of `scr`, where phase F reads
// R.  With room for 8 slots in `scr` every wave simply parks its tiles there.  The wide classes have room for 1-2 slots only
// (UR5: 12 KB per slot beside 115 KB of k / v panels): their waves park the tiles in the k / v panels instead -- dead once
// every wave has left phase J, one barrier -- which replaces the pairwise hand-down through the few slots (three rounds of
// store / barrier / add / barrier).  The final add walks the live (tile, register) pairs only and issues all partial reads of an
// element before the first add (a loop over a run-time count made it one LDS round trip per partial).
template <int CT>
__device__ __forceinline__ void tile_j_finish(v4d (&acc)[2][CT], int RT, int D, double* scr, int nslot, double* panels, int panel_doubles, int wv,
                                              int lane, int tid, unsigned long long* dbg = nullptr) {
  const int slot = RT * CT * 256;
  const bool in_panels = nslot < RF_NW && RF_NW * slot <= panel_doubles;
  double* base = in_panels ? panels : scr;
  unsigned long long tq0 = dbg ? clock64() : 0;
  if (in_panels) lds_barrier();  // every wave is done reading k and v
  if (dbg && lane == 0) { unsigned long long now = clock64(); dbg[9] += now - tq0; tq0 = now; }
  const int nfin = tile_j_reduce<CT>(acc, RT, D, base, in_panels ? RF_NW : nslot, wv, lane);
  lds_barrier();
  if (dbg && lane == 0) { unsigned long long now = clock64(); dbg[10] += now - tq0; tq0 = now; }
  if (nfin > 1 || in_panels) {  // add the partials with all threads (fixed order), result in slot 0 of scr
    const int n0 = tile_j_rmax(D, 0) + 1, n1 = RT > 1 ? tile_j_rmax(D, 1) + 1 : 0;  // live registers per tile, row tile 0 / 1
    const int nlive = CT * (n0 + n1);
    for (int q = wv; q < nlive; q += RF_NW) {  // one (tile, register) pair = 64 elements per wave and pass
      const int q1 = q - CT * n0;
      const int n1s = imax(n1, 1);
      const int tt = q1 < 0 ? q / n0 : CT + q1 / n1s, r = q1 < 0 ? q - (q / n0) * n0 : q1 - (q1 / n1s) * n1s;
      const int e = tt * 256 + r * 64 + lane;
      double sacc;
      if (nfin == RF_NW) {
        double x[RF_NW];
#pragma unroll
        for (int w = 0; w < RF_NW; ++w) x[w] = base[w * slot + e];
        sacc = x[0];
#pragma unroll
        for (int w = 1; w < RF_NW; ++w) sacc += x[w];
      } else {
        sacc = base[e];
        for (int w = 1; w < nfin; ++w) sacc += base[w * slot + e];
      }
      scr[e] = sacc;
    }
    lds_barrier();
  }
  if (dbg && lane == 0) dbg[11] += clock64() - tq0;
}

// R[c][kind][p] = sum of the nfin partials; tile (c>>4, kind), element (row c&15, col p) in the accumulator layout
// row = (lane>>4) + 4 r, col = lane & 15
struct TileR {
  const double* base;
  int CT, nfin, slot;
};
__device__ __forceinline__ double tile_r(const TileR& R, int c, int kind, int p) {
  const double* q = R.base + ((c >> 4) * R.CT + kind) * 256 + ((c & 15) >> 2) * 64 + ((c & 3) << 4) + p;
  // nfin is 1 or 2: both reads always (the second at offset 0 again when there is one partial), a select instead of a loop -- phase F
  // calls this dozens of times per thread and a run-time loop made each call a branch with its own LDS wait
  const double s0 = q[0], s1 = q[R.nfin > 1 ? R.slot : 0];
  return R.nfin > 1 ? s0 + s1 : s0;
}
// the cart-pole class always leaves ONE finished tile set (split-j form of phase J)
__device__ __forceinline__ double tile_r1(const TileR& R, int c, int kind, int p) {
  return R.base[((c >> 4) * R.CT + kind) * 256 + ((c & 15) >> 2) * 64 + ((c & 3) << 4) + p];
}

// ---------------------------------------------------------------------------------------
// policy phase: u = u_max tanh((W (phi o mask)) / u_max),  phi_b = exp(-|| (s - c_b)/l ||^2)   (Policy.py:242-265)
// Same expanded-distance product as phase K, the 16 particles against 16 basis functions per tile:
//   A operand  lane (m, kk) : -2 s_mq / l_q^2,  q = 4 i + kk      B operand  lane (kk, n) : c[b0 + n][q]
//   result     lane (kq, n), register r : particle kq + 4 r, basis b0 + n
// A lane therefore holds phi of 4 particles for one basis function and folds it straight into its partial sums of
// W phi (U inputs x 4 particles): phi never goes to memory.  The dropout keep bits of (particle, basis) come from one
// Philox draw per (particle, 4 consecutive bases) as in philox_keep(); a lane draws for particle kq + 4 (n & 3) and the
// four lanes of a quad exchange words so that each ends up with the bit of its own basis for its 4 particles.
// Partial sums meet through a 16-lane DPP row reduction and an 8-wave LDS reduction.
// ---------------------------------------------------------------------------------------
// Narrow shapes (at most 2 inputs): W phi accumulated in registers (4 particles x U per lane) and reduced over the 16 basis
// lanes by DPP at the end -- cheaper than the round trip through LDS of the two-product form below.
template <int NQ, int UM>
__device__ __forceinline__ void tile_polr_load(double (&cb)[NQ], double (&wk)[UM], gptr_t cen, gptr_t wgt, int B, int PF, int U, int tile, int kk, int n) {
  const int b = imin(16 * tile + n, B - 1);
#pragma unroll
  for (int i = 0; i < NQ; ++i) cb[i] = cen[(size_t)b * PF + imin(4 * i + kk, PF - 1)];
#pragma unroll
  for (int k = 0; k < UM; ++k) wk[k] = wgt[(size_t)imin(k, U - 1) * B + b];
}
template <int NQ>
struct TilePolRConst {
  double a_s[NQ], ilq[NQ], ss4[4];
};
// DM: dropout mode -- 0 none, 1 Philox keep bits, 2 the caller's mask buffer (a template parameter: as run-time tests inside the tile
// loop the modes were scalar branches between the loads and the MFMAs, with conservative counter waits at their joins)
template <int NQ, int UM, int DM>
__device__ __forceinline__ void tile_polr_consume(const double (&cb)[NQ], const double (&wk)[UM], const TilePolRConst<NQ>& c, const FwdArgs& a,
                                                 const mcp_noise& nzl, int B, int U, int t, int m0, int tile, int kk, int n, int lane, bool drop, double keep_scale,
                                                 uint32_t drop_thr, double (&uacc)[4][UM]) {
  double scc = 0.0;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const double v = c.ilq[i] * cb[i];
    scc = fma(v, v, scc);
  }
  const double cc = fold_kk(scc);
  v4d C = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < NQ; ++i) C = __builtin_amdgcn_mfma_f64_16x16x4f64(c.a_s[i], cb[i], C, 0, 0, 0);
  const int b = 16 * tile + n;
  const bool bvalid = b < B;
  const int bc = imin(b, B - 1);
  bool keep[4] = {true, true, true, true};
  if (DM != 0) {
    if (DM == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = imin(m0 + kk + 4 * r, a.M - 1);
        keep[r] = nzl.masks[((size_t)t * a.M + mm) * B + bc] != 0;
      }
    } else {
      const int cq = n & 3;
      const int mm = imin(m0 + kk + 4 * cq, a.M - 1);
      const u32x4 rnd = philox_draw(nzl, mm, t, MCP_STREAM_MASK, (uint32_t)(bc >> 2));
      uint32_t wr[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const int ws = (cq + k4) & 3;  // the word lane (cq - k4) needs from me: its basis index is ... see below
        // round k4: lane cq sends word[(cq + k4) & 3] to the lane of the quad whose basis offset is (cq + k4) & 3;
        // equivalently lane c' receives, from lane r = (c' - k4) & 3 (the drawer for particle kq + 4 r), word[c']
        const uint32_t snd = ws == 0 ? rnd.x : ws == 1 ? rnd.y : ws == 2 ? rnd.z : rnd.w;
        const int src = (cq - k4) & 3;
        const uint32_t rcv = quad_from_back(snd, k4);
#pragma unroll
        for (int r = 0; r < 4; ++r) wr[r] = (src == r) ? rcv : wr[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) keep[r] = wr[r] >= drop_thr;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double dist = (c.ss4[r] + cc) + C[r];
    double phi = exp(-dist);
    if (DM != 0) phi = keep[r] ? phi * keep_scale : 0.0;
    if (!bvalid) phi = 0.0;
#pragma unroll
    for (int k = 0; k < UM; ++k)
      if (k < U) uacc[r][k] = fma(wk[k], phi, uacc[r][k]);
  }
}
template <int NQ, int UM, int DM>
__device__ __forceinline__ void tile_policy_reg(const FwdArgs& a, const mcp_noise& nzl, const double* invl, const double* sf, gptr_t cen, gptr_t wgt, double* wslot, double* upart,
                                            int B, int PF, int U, int t, int m0, int wv, int lane, bool drop, double keep_scale, uint32_t drop_thr, int tb, int te) {
  const int kk = lane >> 4, n = lane & 15;
  TilePolRConst<NQ> c;
  double sss = 0.0;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int q = 4 * i + kk;
    const bool qv = q < PF;
    const double sv = qv ? sf[n * PF + q] : 0.0;
    const double il = qv ? invl[q] : 0.0;
    const double il2s = il * il * sv;
    c.ilq[i] = il;
    c.a_s[i] = -2.0 * il2s;
    sss = fma(il2s, sv, sss);
  }
  sss = fold_kk(sss);
  if (kk == 0) wslot[n] = sss;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int r = 0; r < 4; ++r) c.ss4[r] = wslot[kk + 4 * r];
  double uacc[4][UM];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < UM; ++k) uacc[r][k] = 0.0;
  const int ntile = te - tb;  // this workgroup's tiles of 16 basis functions: [tb, te)
  if (wv < ntile) {
    const int nt = (ntile - wv + RF_NW - 1) / RF_NW;
    double c0[NQ], c1[NQ], w0[UM], w1[UM];
    tile_polr_load<NQ, UM>(c0, w0, cen, wgt, B, PF, U, tb + wv, kk, n);
    for (int sI = 0; sI + 1 < nt; sI += 2) {
      tile_polr_load<NQ, UM>(c1, w1, cen, wgt, B, PF, U, tb + wv + RF_NW * (sI + 1), kk, n);
      tile_polr_consume<NQ, UM, DM>(c0, w0, c, a, nzl, B, U, t, m0, tb + wv + RF_NW * sI, kk, n, lane, drop, keep_scale, drop_thr, uacc);
      tile_polr_load<NQ, UM>(c0, w0, cen, wgt, B, PF, U, tb + wv + RF_NW * imin(sI + 2, nt - 1), kk, n);
      tile_polr_consume<NQ, UM, DM>(c1, w1, c, a, nzl, B, U, t, m0, tb + wv + RF_NW * (sI + 1), kk, n, lane, drop, keep_scale, drop_thr, uacc);
    }
    if (nt & 1) tile_polr_consume<NQ, UM, DM>(c0, w0, c, a, nzl, B, U, t, m0, tb + wv + RF_NW * (nt - 1), kk, n, lane, drop, keep_scale, drop_thr, uacc);
  }
  // sum over the 16 basis lanes of each row; lane 15 of row kq holds the partial of particles kq + 4 r
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int k = 0; k < UM; ++k) {
      if (k < U) {
        double v = uacc[r][k];
        v += dpp_take<0x111, 0xf>(v);
        v += dpp_take<0x112, 0xf>(v);
        v += dpp_take<0x114, 0xf>(v);
        v += dpp_take<0x118, 0xf>(v);
        if (n == 15) upart[(wv * TL_PT + kk + 4 * r) * U + k] = v;
      }
    }
  }
}

// Wide shapes: W phi as a second matrix product per tile (the accumulator is one 16x16 tile instead of 4 x U registers).
#define TL_PHP 17  // row pitch (doubles) of the per-wave 16 x 16 phi tile
template <int NQ>
__device__ __forceinline__ void tile_pol_load(double (&cb)[NQ], double (&wb)[4], gptr_t cen, gptr_t wgt, int B, int PF, int U, int tile, int kk, int n) {
  const int b = imin(16 * tile + n, B - 1);
#pragma unroll
  for (int i = 0; i < NQ; ++i) cb[i] = cen[(size_t)b * PF + imin(4 * i + kk, PF - 1)];
  // B operand of the second product: W[k = n][16 tile + 4 s + kk]
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) wb[s4] = wgt[(size_t)imin(n, U - 1) * B + imin(16 * tile + 4 * s4 + kk, B - 1)];
}
template <int NQ>
struct TilePolConst {
  double a_s[NQ], ilq[NQ], ss4[4];
};
template <int NQ, int DM>
__device__ __forceinline__ void tile_pol_consume(const double (&cb)[NQ], const double (&wb)[4], const TilePolConst<NQ>& c, const FwdArgs& a,
                                                 const mcp_noise& nzl, int B,
                                                 int U, int t, int m0, int tile, int kk, int n, int lane, bool drop, double keep_scale,
                                                 uint32_t drop_thr, double* ptile, v4d& uacc) {
  double scc = 0.0;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const double v = c.ilq[i] * cb[i];
    scc = fma(v, v, scc);
  }
  const double cc = fold_kk(scc);
  v4d C = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < NQ; ++i) C = __builtin_amdgcn_mfma_f64_16x16x4f64(c.a_s[i], cb[i], C, 0, 0, 0);
  const int b = 16 * tile + n;
  const bool bvalid = b < B;
  const int bc = imin(b, B - 1);
  bool keep[4] = {true, true, true, true};
  if (DM != 0) {
    if (DM == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = imin(m0 + kk + 4 * r, a.M - 1);
        keep[r] = nzl.masks[((size_t)t * a.M + mm) * B + bc] != 0;
      }
    } else {
      const int cq = n & 3;
      const int mm = imin(m0 + kk + 4 * cq, a.M - 1);
      const u32x4 rnd = philox_draw(nzl, mm, t, MCP_STREAM_MASK, (uint32_t)(bc >> 2));
      uint32_t wr[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        // round k4: lane cq sends word[(cq + k4) & 3]; lane c' receives, from lane r = (c' - k4) & 3 (the drawer for particle
        // kq + 4 r), word[c'] -- the keep bit of its own basis for that particle
        const int ws = (cq + k4) & 3;
        const uint32_t snd = ws == 0 ? rnd.x : ws == 1 ? rnd.y : ws == 2 ? rnd.z : rnd.w;
        const int src = (cq - k4) & 3;
        const uint32_t rcv = quad_from_back(snd, k4);
#pragma unroll
        for (int r = 0; r < 4; ++r) wr[r] = (src == r) ? rcv : wr[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) keep[r] = wr[r] >= drop_thr;
    }
  }
  // phi tile (particle x basis) -> this wave's LDS scratch, then W phi as a second product: A = phi[p][b], B = W[k][b]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double dist = (c.ss4[r] + cc) + C[r];
    double phi = exp(-dist);
    if (DM != 0) phi = keep[r] ? phi * keep_scale : 0.0;
    if (!bvalid) phi = 0.0;
    ptile[(kk + 4 * r) * TL_PHP + n] = phi;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave writes and reads: program order is enough
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const double av = ptile[n * TL_PHP + 4 * s4 + kk];
    const double bw = n < U ? wb[s4] : 0.0;
    uacc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bw, uacc, 0, 0, 0);
  }
}
template <int NQ, int DM>
__device__ __forceinline__ void tile_policy(const FwdArgs& a, const mcp_noise& nzl, const double* invl, const double* sf, gptr_t cen, gptr_t wgt, double* wslot, double* ptile,
                                            double* upart, int B, int PF, int U, int t, int m0, int wv, int lane, bool drop, double keep_scale,
                                            uint32_t drop_thr, int tb, int te) {
  const int kk = lane >> 4, n = lane & 15;
  TilePolConst<NQ> c;
  double sss = 0.0;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int q = 4 * i + kk;
    const bool qv = q < PF;
    const double sv = qv ? sf[n * PF + q] : 0.0;
    const double il = qv ? invl[q] : 0.0;
    const double il2s = il * il * sv;
    c.ilq[i] = il;
    c.a_s[i] = -2.0 * il2s;
    sss = fma(il2s, sv, sss);
  }
  sss = fold_kk(sss);
  if (kk == 0) wslot[n] = sss;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int r = 0; r < 4; ++r) c.ss4[r] = wslot[kk + 4 * r];
  v4d uacc = (v4d){0.0, 0.0, 0.0, 0.0};  // rows: particles kq + 4 r, column n: input k = n
  const int ntile = te - tb;  // this workgroup's tiles of 16 basis functions: [tb, te)
  if (wv < ntile) {
    const int nt = (ntile - wv + RF_NW - 1) / RF_NW;
    double c0[NQ], c1[NQ], w0[4], w1[4];
    tile_pol_load<NQ>(c0, w0, cen, wgt, B, PF, U, tb + wv, kk, n);
    for (int sI = 0; sI + 1 < nt; sI += 2) {
      tile_pol_load<NQ>(c1, w1, cen, wgt, B, PF, U, tb + wv + RF_NW * (sI + 1), kk, n);
      tile_pol_consume<NQ, DM>(c0, w0, c, a, nzl, B, U, t, m0, tb + wv + RF_NW * sI, kk, n, lane, drop, keep_scale, drop_thr, ptile, uacc);
      tile_pol_load<NQ>(c0, w0, cen, wgt, B, PF, U, tb + wv + RF_NW * imin(sI + 2, nt - 1), kk, n);
      tile_pol_consume<NQ, DM>(c1, w1, c, a, nzl, B, U, t, m0, tb + wv + RF_NW * (sI + 1), kk, n, lane, drop, keep_scale, drop_thr, ptile, uacc);
    }
    if (nt & 1) tile_pol_consume<NQ, DM>(c0, w0, c, a, nzl, B, U, t, m0, tb + wv + RF_NW * (nt - 1), kk, n, lane, drop, keep_scale, drop_thr, ptile, uacc);
  }
  if (n < U) {
#pragma unroll
    for (int r = 0; r < 4; ++r) upart[(wv * TL_PT + kk + 4 * r) * U + n] = uacc[r];
  }
}

// ---------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------
// BIG selects the operand-group counts: small problems (D <= 8, policy features <= 8, inputs <= 2: cart-pole class) keep two
// feature groups per product in registers, everything else eight.  One kernel per class: compiling both paths into one
// function made the register allocator spill the small path's long-lived values for the benefit of the big one.
// CLS: 0 = cart-pole class (D <= 7, policy features <= 8, inputs <= 2, N <= 512), 1 = UR5 class (<= 24, <= 24, <= 6, N <= 512), 2 = any
// PMS: the policy is evaluated on a simulated measurement (mcp_meas, MC_PILCO4PMS.apply_policy) instead of the true state
// GSH: a.gsh_cs workgroups per 16-particle tile, each evaluates G / gsh_cs consecutive GPs and (round 4, a.uxch) its share of the policy's
// basis functions -- rounds 2-3: the whole policy, redundantly; they hand
// each other the sampled increments once per step exactly as the small-tile kernel's GP-sharded launch does (rollout_fwd.hip).
// XL: X^T and alpha of every GP are staged in LDS once per launch (cart-pole class, when the layout has the room): phases K and J
// then take their small operands with LDS latency instead of an L2 round trip per tile / batch
template <int MAXDEG, int CLS, bool PMS, bool GSH = false, bool XL = false>
__global__ __launch_bounds__(RF_NT) void rollout_fwd_tile_kernel(FwdArgs a) {
  constexpr int NG = CLS == 0 ? 2 : (CLS == 1 ? 6 : 8);              // feature groups of 4 (GP inputs, policy features)
  constexpr int MAXTASK = CLS == 2 ? TL_MAXTASK : 2;                  // 32-row blocks of Kinv per wave
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const mcp_model& md = a.model;
  const mcp_policy& pl = a.pol;
  const mcp_noise nzl = noise_of_launch(a.nz);
  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int wv0 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int S = md.S, U = md.U, G = md.G, D = md.D, B = pl.B, PF = pl.P, M = a.M, T = a.T;
  constexpr int P = TL_PT;
  const TileLayout L = tile_layout(S, U, D, G, PF, a.NpadMax, a.maxdeg, XL);
  double* invl = smem + L.invl;
  double* xs = smem + L.xs;
  double* us = smem + L.us;
  double* z = smem + L.z;
  double* sf = smem + L.sf;
  double* dl = smem + L.dl;
  double* epsb = smem + L.eps;
  double* ks = smem + L.ks;
  double* kv = smem + L.kv;
  double* qa = smem + L.qa;
  double* mup = smem + L.mup;
  double* scr = smem + L.scr;
  GpL* gpl = reinterpret_cast<GpL*>(smem + L.gpl);
  double* kpar = smem + L.kpar;
  int cluster = blockIdx.x, gbeg = 0, gend = G;  // the GPs [gbeg, gend) this workgroup evaluates
  int myc = 0;                                   // ... as member myc of its cluster of CS workgroups
  const int CS = GSH ? a.gsh_cs : 1;
  // round 5: RS = 2 workgroups per (tile, GP range), one per half of the rows of Kinv (phases V and J; see FwdArgs.gsh_rs) -- only the instantiations
  // that take the per-tile form of phase J carry the code
  constexpr bool RSP = GSH && CLS >= 1 && MAXDEG <= 1;
  const int RS = RSP ? imax(1, a.gsh_rs) : 1;
  int hrow = 0;  // which half (the last one finishes the GP: moments, sample, hand-off -- it has the larger share of the blocks, so its partner's sums are there when it arrives)
  if (GSH) {  // blocks b and b + 8 share an XCD: the members of a cluster sit 8 apart (speed only)
    const int CW = CS * RS;  // workgroups per cluster
    const int b = blockIdx.x, grp = b / (8 * CW), r = b - grp * 8 * CW;
    if (RSP && a.gsh_map == 1) {
      // row part major (round 6): XCD x = the blocks b = x (mod 8) hosts the items [x per, (x + 1) per) of the order ((member, row part), tile)
      const int per = (int)gridDim.x >> 3, j = (b & 7) * per + (b >> 3);
      if (j >= CW * a.nclusters) return;
      const int combo = j / a.nclusters;
      cluster = j - combo * a.nclusters;
      myc = combo / RS;
      hrow = combo - myc * RS;
    } else if (RS > 1) {
      // the two halves of a GP range are neighbours in the member order: with one cluster per XCD they share its L2, as the CS members did
      const int mt = r >> 3;
      cluster = grp * 8 + (r & 7);
      myc = mt / RS;
      hrow = mt - myc * RS;
    } else if (8 % CS == 0) {
      // Member c of every cluster goes to the XCDs [c * 8/CS, (c+1) * 8/CS) (blocks are dealt round-robin over the 8 XCDs), so
      // an XCD's L2 only ever holds the Kinv of ONE GP range: the UR5 shape needs 3.8 MB per range against 4 MB of L2 -- with both
      // members of a cluster on one XCD (round 1) its six Kinv thrashed the L2 (FETCH_SIZE 10.8 GB per launch, profiles/r02_c5_*).
      const int W = 8 / CS, x = r & 7;  // x: the block's XCD under round-robin placement
      myc = x / W;
      cluster = grp * 8 + (r >> 3) * W + (x % W);
    } else {
      cluster = grp * 8 + (r & 7);
      myc = r >> 3;
    }
    if (cluster >= a.nclusters) return;
    gbeg = (myc * G) / CS;
    gend = ((myc + 1) * G) / CS;
  }
  const bool fin = hrow == RS - 1;              // this workgroup finishes its GPs
  const bool writer = !GSH || (gbeg == 0 && fin);  // states / inputs are identical in the workgroups of a cluster: one of them stores
  // the policy: member myc evaluates the tiles [ptb, pte) of 16 basis functions and the members add their partial sums (a.uxch), or everyone all of it
  const bool psplit = GSH && a.uxch != nullptr;
  const int pnt = (B + 15) >> 4;
  const int ptb = psplit ? (myc * pnt) / CS : 0, pte = psplit ? ((myc + 1) * pnt) / CS : pnt;
  int* abortw = reinterpret_cast<int*>(dl + P * G);
  if (GSH && tid0 == 0) *abortw = 0;
  const int m0 = cluster * P;
  uint32_t bad = 0;
  const bool drop = pl.p_drop > 0.0;
  const double keep_scale = 1.0 / (1.0 - pl.p_drop);
  const uint32_t drop_thr = drop_threshold(pl.p_drop);
  const int nna = md.n_not_angle, na = md.n_angle;
  const int RT = CLS == 0 ? 1 : ((D + 1 + 15) >> 4);  // row tiles of [X^T;1]: a compile-time 1 for the small class

  // ---- one-time staging ------------------------------------------------------------------
  for (int it = tid0; it < PF; it += RF_NT) invl[it] = exp(-pl.log_ls[it]);
  stage_gp_tables(md.gp, md.var_scale, G, D, gpl, kpar, tid0);
  if (XL) {
    double* xt_w = smem + L.xt;
    double* al_w = smem + L.al;
    for (int g = 0; g < G; ++g) {
      const mcp_gp& gq = md.gp[g];
      for (int it = tid0; it < D * a.NpadMax; it += RF_NT) {
        const int d = it / a.NpadMax, j = it - d * a.NpadMax;
        xt_w[(g * D + d) * a.NpadMax + j] = j < gq.Npad ? gq.Xt[(size_t)d * gq.Npad + j] : 0.0;
      }
      for (int it = tid0; it < a.NpadMax; it += RF_NT) al_w[g * a.NpadMax + it] = it < gq.Npad ? gq.alpha[it] : 0.0;
    }
  }
  lds_barrier();
  // launch constants of the degree-2 polynomial term: sum_j alpha_j X_jc X_je (for d mu/dz)
  if (MAXDEG >= 2 && a.maxdeg >= 2) {  // (qa has no storage when maxdeg < 2)
    for (int it = tid0; it < G * D * D; it += RF_NT) {
      const int g = it / (D * D), r = it - g * D * D, c = r / D, e = r - c * D;
      const GpL& gp = gpl[g];
      double s = 0.0;
      if (gp.deg >= 2)
        for (int j = 0; j < gp.N; ++j) s = fma(gp.alpha[j] * gp.X[(size_t)j * D + c], gp.X[(size_t)j * D + e], s);
      qa[it] = s;
    }
  }
  gptr_t cen = (gptr_t)pl.centers;
  gptr_t wgt = (gptr_t)pl.weight;

  // thread (p, s) owns state component s of particle p; threads 256.. draw the process noise of the step
  const bool own = tid0 < P * S;
  const int op = own ? tid0 / S : 0, os = own ? tid0 - op * S : 0;
  const int om = imin(m0 + op, M - 1);
  const bool ovalid = own && (m0 + op < M);
  double xn = own ? a.x0[(size_t)om * S + os] : 0.0;
  int cur = 0;
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
  int vel_of_pos = 0;
  for (int g = 0; g < G; ++g)
    if (own && md.not_vel[g] == os) vel_of_pos = md.vel[g];
  const double Ts = md.Ts;
  // measurement model (see rollout_fwd.hip): thread (p, s) produces the measured value of its own component; a velocity thread
  // rebuilds the noisy position of its pair (same draw) and carries the filter's three values
  const mcp_meas& ms = pl.meas;
  int pm_pos = -1, pm_vel = -1, pm_pair = 0;
  double pm_std = 0.0, pm_prev_np = 0.0, pm_prev_nv = 0.0, pm_prev_mv = 0.0;
  if (PMS && own) {
    for (int i = 0; i < ms.n; ++i) {
      if (ms.pos[i] == os) {
        pm_pos = i;
        pm_std = ms.std_pos[i];
      }
      if (ms.vel[i] == os) {
        pm_vel = i;
        pm_pair = ms.pos[i];
        pm_std = ms.std_pos[i];
      }
    }
  }
  // the last P*G threads draw the process noise of step t+1 while phase F of the first GP keeps only a few waves busy
  const int et = tid0 - (RF_NT - P * G);
  const bool edraw = et >= 0;
  const int ep = edraw ? et / G : 0, eg = edraw ? et - ep * G : 0;
  auto draw_eps = [&](int tt) {
    double e = 0.0;
    if (a.particle_pred) {
      const int mm = imin(m0 + ep, M - 1);
      e = nzl.eps ? nzl.eps[((size_t)tt * M + mm) * G + eg] : philox_normal(nzl, mm, tt, eg);
    }
    epsb[(tt & 1) * P * G + et] = e;
  };
  if (edraw && T > 1) draw_eps(0);
  unsigned long long last_stamp = clock64();
  lds_barrier();

  for (int t = 0; t < T; ++t) {
    // Lane and wave ids are laundered once per time step: everything the phases derive from them (operand addresses, feature
    // and particle indices, predicates -- hundreds of values over the unrolled operand groups of the wide classes) is then
    // recomputed where it is used instead of being hoisted out of the time loop, kept live across every phase and spilled to
    // scratch (UR5 class: 158 -> see tools/kernel_resources.py).  A handful of integer instructions per phase against
    // scratch reloads with global-memory latency inside the loop.
    int lane = lane0, wv = wv0, tid = tid0;
    asm volatile("" : "+v"(lane));
    asm volatile("" : "+v"(tid));
    asm volatile("" : "+s"(wv));
    // ---- phase S: publish x_t, the GP / policy features of each state component ------------------
    double xm = xn;  // what the policy sees of this component
    if (own) {
      double* xc = xs + cur * P * S;
      xc[op * S + os] = xn;
      if (ovalid) {
        if (writer) store_through(&a.states[((size_t)t * M + m0 + op) * S + os], xn);
        if (is_bad(xn)) bad |= MCP_STATUS_NAN;
      }
    }
    if (PMS) {
      lds_barrier();  // a velocity thread reads its pair's position (the state threads span several waves)
      if (own) {
        const int pi = pm_pos >= 0 ? pm_pos : pm_vel;
        double npos = pm_pos >= 0 ? xn : xs[cur * P * S + op * S + pm_pair];
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
        if (ovalid) {
          if (writer) store_through(&ms.meas[((size_t)t * M + m0 + op) * S + os], xm);
          if (is_bad(xm)) bad |= MCP_STATUS_NAN;
        }
      }
    }
    if (own) {
      double sn = 0.0, cs = 0.0;
      if (zi_ang >= 0 || pi_ang >= 0) sincos_fast(xn, &sn, &cs);
      double snm = sn, csm = cs;  // trig of the measured value (policy features)
      if (PMS && pi_ang >= 0 && xm != xn) sincos_fast(xm, &snm, &csm);
      if (zi_plain >= 0) z[op * D + zi_plain] = xn;
      if (zi_ang >= 0) {
        z[op * D + nna + zi_ang] = sn;
        z[op * D + nna + na + zi_ang] = cs;
      }
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
    lds_barrier();
    TL_STAMP(0);
    // ---- policy: phi and W phi on the matrix cores, partial sums per wave -> LDS ----------------------
    {
      const int dm = !drop ? 0 : (nzl.masks ? 2 : 1);  // wave-uniform
      double* pt_w = smem + L.ptile + wv * 16 * TL_PHP;
      if (CLS == 0) {
        if (dm == 1)
          tile_policy_reg<NG, 2, 1>(a, nzl, invl, sf, cen, wgt, scr + wv * 16, smem + L.upart, B, PF, U, t, m0, wv, lane, drop, keep_scale, drop_thr, ptb, pte);
        else if (dm == 2)
          tile_policy_reg<NG, 2, 2>(a, nzl, invl, sf, cen, wgt, scr + wv * 16, smem + L.upart, B, PF, U, t, m0, wv, lane, drop, keep_scale, drop_thr, ptb, pte);
        else
          tile_policy_reg<NG, 2, 0>(a, nzl, invl, sf, cen, wgt, scr + wv * 16, smem + L.upart, B, PF, U, t, m0, wv, lane, drop, keep_scale, drop_thr, ptb, pte);
      } else {
        if (dm == 1)
          tile_policy<NG, 1>(a, nzl, invl, sf, cen, wgt, scr + wv * 16, pt_w, smem + L.upart, B, PF, U, t, m0, wv, lane, drop, keep_scale, drop_thr, ptb, pte);
        else if (dm == 2)
          tile_policy<NG, 2>(a, nzl, invl, sf, cen, wgt, scr + wv * 16, pt_w, smem + L.upart, B, PF, U, t, m0, wv, lane, drop, keep_scale, drop_thr, ptb, pte);
        else
          tile_policy<NG, 0>(a, nzl, invl, sf, cen, wgt, scr + wv * 16, pt_w, smem + L.upart, B, PF, U, t, m0, wv, lane, drop, keep_scale, drop_thr, ptb, pte);
      }
    }
    lds_barrier();
    TL_STAMP(1);
    // ---- squash, publish u ----------------------------------------------------------------------------
    if (tid < P * U) {
      const int p = tid / U, k = tid - p * U;
      const double* up = smem + L.upart;
      double sacc = 0.0;
#pragma unroll
      for (int w = 0; w < RF_NW; ++w) sacc += up[(w * P + p) * U + k];
      if (psplit) {
        // this member's partial sum goes out as two granules; the members' sums are added in member order, so every workgroup of the
        // cluster ends up with the same bits
        const unsigned long long bits = (unsigned long long)__double_as_longlong(sacc);
        gu64_t ub = (gu64_t)a.uxch + ((size_t)(cluster * 2 + (t & 1)) * CS) * (size_t)(P * U * 2);
        gu64_t mine = ub + (size_t)myc * (P * U * 2) + 2 * tid;
        if (fin) {  // (the other half of my GP range holds the same bits)
          store_granule(mine, (unsigned)t + 1u, (unsigned)bits);
          store_granule(mine + 1, (unsigned)t + 1u, (unsigned)(bits >> 32));
        }
        // all partners' granules are requested together and re-read until every tag matches: one L2 round trip, not one per partner
        unsigned lo[MCP_MAX_GP], hi[MCP_MAX_GP];
        bool ok = false;
        for (unsigned spins = 0; spins < RF_SPIN_LIMIT && !ok; ++spins) {
          ok = true;
#pragma unroll
          for (int r = 0; r < MCP_MAX_GP; ++r) {
            if (r < CS && r != myc) {
              gu64_t src = ub + (size_t)r * (P * U * 2) + 2 * tid;
              const unsigned long long x0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const unsigned long long x1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ok = ok && (unsigned)(x0 >> 32) == (unsigned)t + 1u && (unsigned)(x1 >> 32) == (unsigned)t + 1u;
              lo[r] = (unsigned)x0;
              hi[r] = (unsigned)x1;
            }
          }
          if (!ok) __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) *abortw = 1;
        double tot = 0.0;
#pragma unroll
        for (int r = 0; r < MCP_MAX_GP; ++r) {
          if (r < CS) tot += r == myc ? sacc : __longlong_as_double((long long)(((unsigned long long)hi[r] << 32) | lo[r]));
        }
        sacc = tot;
      }
      if (pl.bias) sacc += pl.bias[k];
      const double um = pl.u_max[k];
      const double u = pl.squash ? um * fast_tanh(sacc / um) : sacc;
      us[p * U + k] = u;
      z[p * D + nna + 2 * na + k] = u;
      if (m0 + p < M) {
        if (writer) store_through(&a.inputs[((size_t)t * M + m0 + p) * U + k], u);
        if (is_bad(u)) bad |= MCP_STATUS_NAN;
      }
    }
    lds_barrier();
    TL_STAMP(2);
    if (t == T - 1) break;

    for (int gi = gbeg; gi < gend; ++gi) {
      // The GPs of a step are independent: odd steps walk them backwards.  A workgroup streams every Kinv of its range once per step,
      // and the XCD's L2 (4 MB) sees the same cyclic sweep from all its workgroups: walked in the same order every step, a range that
      // nearly fills the cache (UR5 shape: 3 x 1.28 MB) is evicted just before it comes round again -- measured: every byte re-fetched
      // from the fabric every step.  Back and forth, the last GPs of one step are the first of the next while they are still resident.
      const int g = (t & 1) ? gbeg + gend - 1 - gi : gi;
      asm volatile("" : "+v"(lane));  // (again per GP: nothing derived from the ids stays live across the GP loop)
      asm volatile("" : "+v"(tid));
      asm volatile("" : "+s"(wv));
      const GpL& gp = gpl[g];
      const double* kp = kpar + g * KP_STRIDE(D);
      const int Npad = __builtin_amdgcn_readfirstlane(gp.Npad);
      const int deg = MAXDEG == 0 ? 0 : __builtin_amdgcn_readfirstlane(gp.deg);
      const double* xt_g = smem + L.xt + g * D * a.NpadMax;  // (XL only)
      const double* al_g = smem + L.al + g * a.NpadMax;
      const bool wstamp = a.stamps && blockIdx.x == a.stamp_block;  // diagnostic: every wave's own time in phases K and V, slots 16.. / 24..
      unsigned long long wt0 = wstamp ? clock64() : 0;
      {
        unsigned long long* kdbg = (a.stamps && blockIdx.x == a.stamp_block && wv == 0) ? a.stamps : nullptr;
        const bool exact = MAXDEG >= 1 && deg == MAXDEG;  // (MAXDEG == 0 has no degree tests to remove)
        if (XL) {
          if (exact)
            tile_phase_k<MAXDEG, NG, const double*, MAXDEG >= 1>(gp, xt_g, al_g, a.NpadMax, kp, D, z, ks, kv, scr + wv * 16, mup, wv, lane, kdbg);
          else
            tile_phase_k<MAXDEG, NG, const double*, false>(gp, xt_g, al_g, a.NpadMax, kp, D, z, ks, kv, scr + wv * 16, mup, wv, lane, kdbg);
        } else {
          if (exact)
            tile_phase_k<MAXDEG, NG, gptr_t, MAXDEG >= 1>(gp, (gptr_t)gp.Xt, (gptr_t)gp.alpha, Npad, kp, D, z, ks, kv, scr + wv * 16, mup, wv, lane, kdbg);
          else
            tile_phase_k<MAXDEG, NG, gptr_t, false>(gp, (gptr_t)gp.Xt, (gptr_t)gp.alpha, Npad, kp, D, z, ks, kv, scr + wv * 16, mup, wv, lane, kdbg);
        }
      }
      if (wstamp && lane == 0) a.stamps[16 + wv] += clock64() - wt0;
      lds_barrier();
      TL_STAMP(3);
      wt0 = wstamp ? clock64() : 0;
      // ---- phase V ---------------------------------------------------------------------------
      // Work = 32-row blocks of v x 16-row batches of the summation index.  Wave w takes blocks w, w + 8, ... in full; the
      // last (nblk mod 8) blocks are dealt as ONE contiguous run of batches cut into 8 equal shares, so a share is at most two
      // pieces (the tail of one block, the head of the next) and every wave carries the same load whatever nblk mod 8 is
      // (Npad = 400: 41 batches per wave instead of 50 for five of them and 25 for the rest).  The piece that starts a block
      // collects the partial tiles of the others through the scratch slots, in wave order (fixed summation order).
      {
        constexpr int NACC = MAXTASK + 1;
        const int nblk_all = (Npad + 31) >> 5, nb = Npad >> 4;  // blocks; 16-row batches per block
        const int vb0 = RS > 1 ? (hrow * nblk_all) / RS : 0;      // my blocks [vb0, vb0 + nblk) (row split: one half of them)
        const int nblk = (RS > 1 ? ((hrow + 1) * nblk_all) / RS : nblk_all) - vb0;
        const int nfull = nblk >> 3, rem = nblk & 7;
        const int nshare = imin(RF_NW, L.vslots + 1);  // waves that share the remainder run (one scratch slot per non-collecting piece)
        // batches per share; with too few scratch slots for equal shares, whole blocks (one per wave, no partial tiles at all)
        const int tot = rem * nb, per = imin(nb, (tot + nshare - 1) / nshare);
        const int s0 = imin(wv * per, tot), s1 = imin(tot, s0 + per);
        v4d acc[NACC][2];
        int blk[NACC];     // block of accumulator r (-1: unused)
        bool coll[NACC];   // this accumulator's piece starts its block: it collects and stores
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
          acc[r][0] = (v4d){0.0, 0.0, 0.0, 0.0};
          acc[r][1] = (v4d){0.0, 0.0, 0.0, 0.0};
          blk[r] = -1;
          coll[r] = false;
        }
        const int ba = rem ? s0 / nb : 0;                    // remainder block of piece A (index among the remainder blocks)
        const int a0 = s0 - ba * nb, a1 = imin(nb, s1 - ba * nb);  // its batches [a0, a1)
        const int b1 = s1 - (ba + 1) * nb;                    // piece B: batches [0, b1) of remainder block ba + 1 (if b1 > 0)
        // ONE inlined copy of the block routine: the pieces run through a loop and park their tile in the accumulator set of their
        // index (uniform branches).  (Measured and dropped: letting waves 4-7 do their share with DPP-broadcast vector FMAs so
        // that each SIMD's matrix AND vector pipe work on phase V.  fp64 MFMA and fp64 vector FMA do not run side by side on
        // gfx950 -- tools/mfma_valu_coexec.hip, profiles/r02_mfma_valu_coexec.txt: a SIMD delivers 32 flop/tick whichever pipe or
        // mix of pipes its two waves use -- so the split only added the vector path's overheads: V 148 k -> 176 k cycles at C5.)
        for (int r = 0; r < NACC; ++r) {
          int pb = -1, pjs = 0, pje = 0;
          bool pc = true;
          if (r < nfull) {
            pb = r * RF_NW + wv;
            pje = Npad;
          } else if (r == nfull && s0 < s1) {
            pb = nfull * RF_NW + ba;
            pc = a0 == 0;
            pjs = 16 * a0;
            pje = 16 * a1;
          } else if (r == nfull + 1 && s0 < s1 && b1 > 0) {
            pb = nfull * RF_NW + ba + 1;
            pje = 16 * b1;
          }
          if (pb < 0) continue;
          v4d te = (v4d){0.0, 0.0, 0.0, 0.0}, to = (v4d){0.0, 0.0, 0.0, 0.0};
          #ifdef TLX_VDEPTH3
          tile_v_block<(CLS == 0 || (TLX_VDEPTH3 + 0) > 1) ? 3 : 2>(gp.Kinv, Npad, (vb0 + pb) * 32, pjs, pje, kv, lane, te, to);  // (-DTLX_VDEPTH3=2: every class)
#else
          tile_v_block<2>(gp.Kinv, Npad, (vb0 + pb) * 32, pjs, pje, kv, lane, te, to);
#endif
#pragma unroll
          for (int q = 0; q < NACC; ++q) {
            if (q == r) {
              acc[q][0] = te;
              acc[q][1] = to;
              blk[q] = pb;
              coll[q] = pc;
            }
          }
        }
        // a piece that does not start its block (only piece A can; never wave 0) -> scratch slot wv - 1
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
          if (r == nfull && blk[r] >= 0 && !coll[r]) {
            double* sl = scr + (wv - 1) * 512;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              sl[i * 64 + lane] = acc[r][0][i];
              sl[256 + i * 64 + lane] = acc[r][1][i];
            }
          }
        }
        if (wstamp && lane == 0) a.stamps[24 + wv] += clock64() - wt0;
        lds_barrier();  // every wave is done reading k: v may overwrite it
        TL_STAMP(4);
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
          if (blk[r] < 0 || !coll[r]) continue;
          if (r >= nfull) {
            // the other pieces of this block: the waves after mine whose share starts strictly inside it, in wave order
            const int beta = blk[r] - nfull * RF_NW;
            const int lim = imin((beta + 1) * nb, tot);
            for (int w2 = wv + 1; w2 < RF_NW && w2 * per < lim; ++w2) {
              const double* sl = scr + (w2 - 1) * 512;
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                acc[r][0][i] += sl[i * 64 + lane];
                acc[r][1][i] += sl[256 + i * 64 + lane];
              }
            }
          }
          const int n = lane & 15, kq = lane >> 4;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = (vb0 + blk[r]) * 32 + 2 * (kq + 4 * i);
            if (row < Npad) {
              kv[row * TL_KR + n] = acc[r][0][i];
              kv[(row + 1) * TL_KR + n] = acc[r][1][i];
            }
          }
        }
      }
      lds_barrier();
      TL_STAMP(5);
      // ---- phase J -----------------------------------------------------------------------------
      int CTg;
      const int jnh = (CLS >= 1 && a.xj && L.nslot >= 2 && (MAXDEG <= 1 || deg <= 1)) ? 2 : 1;  // partial tiles per tile left by the per-tile form of phase J
      const int panel_doubles = 2 * a.NpadMax * TL_KR;  // ks and kv are adjacent in the layout
      // row-split cluster: phase J over the rows whose v this workgroup has (32-row blocks = two batches each)
      const int nblk_j = (Npad + 31) >> 5;
      const int jrb0 = RS > 1 ? 2 * ((hrow * nblk_j) / RS) : 0, jrb1 = RS > 1 ? 2 * (((hrow + 1) * nblk_j) / RS) : (1 << 30);
      if (CLS >= 1 && a.xj && (MAXDEG == 0 || deg == 0)) {
        tile_phase_j_bytile<0>(gp, a.xj + (size_t)g * a.xj_stride, a.xj_stride / 512, ks, kv, scr, RT, jnh, wv, lane, (a.stamps && blockIdx.x == a.stamp_block && wv == 0) ? a.stamps : nullptr, jrb0, jrb1);
        lds_barrier();
        CTg = TL_NCOL(0);
      } else if (CLS >= 1 && a.xj && (MAXDEG == 1 || deg == 1)) {
        tile_phase_j_bytile<1>(gp, a.xj + (size_t)g * a.xj_stride, a.xj_stride / 512, ks, kv, scr, RT, jnh, wv, lane, (a.stamps && blockIdx.x == a.stamp_block && wv == 0) ? a.stamps : nullptr, jrb0, jrb1);
        lds_barrier();
        CTg = TL_NCOL(1);
      } else if (MAXDEG == 0 || deg == 0) {
        v4d acc[2][TL_NCOL(0)];
        if (XL)
          tile_phase_j<0, 1, const double*, CLS == 0>(gp, xt_g, al_g, a.NpadMax, kp, D, z, ks, kv, acc, RT, wv, lane, (a.stamps && blockIdx.x == a.stamp_block && wv == 0) ? a.stamps : nullptr);
        else
          tile_phase_j<0, 1, gptr_t, CLS == 0>(gp, (gptr_t)gp.Xt, (gptr_t)gp.alpha, Npad, kp, D, z, ks, kv, acc, RT, wv, lane,
                               (a.stamps && blockIdx.x == a.stamp_block && wv == 0) ? a.stamps : nullptr);
        tile_j_finish<TL_NCOL(0)>(acc, RT, D, scr, L.nslot, ks, panel_doubles, wv, lane, tid, (a.stamps && blockIdx.x == a.stamp_block && wv == 0) ? a.stamps : nullptr);
        CTg = TL_NCOL(0);
      } else if (MAXDEG == 1 || deg == 1) {
        v4d acc[2][TL_NCOL(1)];
        if (XL)
          tile_phase_j<1, 1, const double*, CLS == 0>(gp, xt_g, al_g, a.NpadMax, kp, D, z, ks, kv, acc, RT, wv, lane, (a.stamps && blockIdx.x == a.stamp_block && wv == 0) ? a.stamps : nullptr);
        else
          tile_phase_j<1, 1, gptr_t, CLS == 0>(gp, (gptr_t)gp.Xt, (gptr_t)gp.alpha, Npad, kp, D, z, ks, kv, acc, RT, wv, lane,
                               (a.stamps && blockIdx.x == a.stamp_block && wv == 0) ? a.stamps : nullptr);
        tile_j_finish<TL_NCOL(1)>(acc, RT, D, scr, L.nslot, ks, panel_doubles, wv, lane, tid, (a.stamps && blockIdx.x == a.stamp_block && wv == 0) ? a.stamps : nullptr);
        CTg = TL_NCOL(1);
      } else {
        v4d acc[2][TL_NCOL(2)];
        if (XL)
          tile_phase_j<2, NG, const double*, CLS == 0>(gp, xt_g, al_g, a.NpadMax, kp, D, z, ks, kv, acc, RT, wv, lane, (a.stamps && blockIdx.x == a.stamp_block && wv == 0) ? a.stamps : nullptr);
        else
          tile_phase_j<2, NG, gptr_t, CLS == 0>(gp, (gptr_t)gp.Xt, (gptr_t)gp.alpha, Npad, kp, D, z, ks, kv, acc, RT, wv, lane,
                               (a.stamps && blockIdx.x == a.stamp_block && wv == 0) ? a.stamps : nullptr);
        tile_j_finish<TL_NCOL(2)>(acc, RT, D, scr, L.nslot, ks, panel_doubles, wv, lane, tid, (a.stamps && blockIdx.x == a.stamp_block && wv == 0) ? a.stamps : nullptr);
        CTg = TL_NCOL(2);
      }
      TL_STAMP(6);
      // ---- phase F: moments, sample, d delta/dz -------------------------------------------------
      const TileR Rr = {scr, CTg, jnh, RT * CTg * 256};
#define TL_R(c, k, p) (CLS == 0 ? tile_r1(Rr, c, k, p) : tile_r(Rr, c, k, p))
      if (RSP && RS > 1) {
        // Row-split cluster: R holds the sums over MY half of the rows.  Everything phase F takes from R is linear in it, so each item (p, c)
        // boils its share down to two values -- c < D: the mean-Jacobian difference and the variance-Jacobian sum; c == D: k^T Kinv k and (degree 0)
        // the mean -- the half that does NOT finish the GP (half 0: `!fin`) sends them (4 granules), the finishing half (half 1 = RS - 1: `fin`, the
        // one with the larger share of the blocks) adds them to its own (own + partner: fixed order) and finishes as the one-workgroup form does.
        gu64_t rxb = (gu64_t)a.rxch + ((size_t)((cluster * 2 + (t & 1)) * G + g)) * (size_t)(2 * P * (D + 1) * 4);  // [sender < 2][4][nit]
        const bool rs_dbg = a.stamps && blockIdx.x == a.stamp_block && tid == 0;  // diagnostic: slots 9 (my sums), 10 (partner poll), 11 (finish), 14 (end of J to the end of the hand-off)
        unsigned long long rs_t0 = rs_dbg ? clock64() : 0;
        // One pass: P (D + 1) <= 16 x 25 items on 512 threads (this class has D <= 24).  k^T Kinv k and k(z, z) of a particle are sums over the
        // D + 1 columns (degree 1): every item forms the term of its own column, the terms meet in the k panel (dead since phase J) and each item
        // adds the D + 1 of its particle in column order -- instead of every item walking all columns through R by itself (5.6 k -> see NOTES).
        const int nit = P * (D + 1);
        const bool act = tid < nit;
        const int it = act ? tid : 0;
        const int p = it / (D + 1), c = it - p * (D + 1);
        const double* zp = z + p * D;
        const double vscale = gp.var_scale;
        double v0 = 0.0, v1 = 0.0, il2 = 0.0, w1c = 0.0;
        const bool lin = MAXDEG >= 1 && deg >= 1;  // (uniform)
        if (lin) {
          const double w = kp[KP_W1(D) + c];
          double tk, tz;
          if (c < D) {
            const double wz = w * zp[c];
            tk = wz * TL_R(c, 2, p);
            tz = wz * zp[c];
          } else {
            tk = fma(w, TL_R(D, 2, p), TL_R(D, 1, p));
            tz = w;
          }
          if (act) {
            ks[it] = tk;
            ks[nit + it] = tz;
          }
        }
        if (c < D) {
          const double il = kp[KP_INVLS(D) + c];
          il2 = il * il;
          v0 = fma(zp[c], TL_R(D, 0, p), -TL_R(c, 0, p));
          v1 = 4.0 * il2 * fma(zp[c], TL_R(D, 1, p), -TL_R(c, 1, p));
          if (lin) {
            w1c = kp[KP_W1(D) + c];
            v1 = fma(-2.0 * w1c, TL_R(c, 2, p), v1);
          }
        } else if (MAXDEG == 0) {
          v1 = TL_R(D, 0, p);
        }
        // granule (value q, half h) of item `it` sits at [2 q + h][it]: every store / load instruction of a wave covers 64 consecutive granules.
        // The finishing half asks for its partner's values -- the two of this item and its k^T Kinv k of this particle -- BEFORE it adds up its own:
        // the partner is half a block ahead, its granules are usually there and the round trip runs under the sums below.
        // (three row parts: the senders 0 and 1 own a block of 4 nit granules each)
        gu64_t sl = rxb + (fin ? 0 : hrow * 4 * nit) + it;
        gu64_t sk = rxb + (p * (D + 1) + D);
        unsigned long long x[2][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
        auto ask = [&]() {
#pragma unroll
          for (int sd = 0; sd < 2; ++sd) {
            if (sd < RS - 1) {
#pragma unroll
              for (int q = 0; q < 4; ++q) x[sd][q] = __hip_atomic_load(sl + (sd * 4 + q) * nit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              x[sd][4] = __hip_atomic_load(sk + sd * 4 * nit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              x[sd][5] = __hip_atomic_load(sk + (sd * 4 + 1) * nit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        };
        if (fin && act) ask();
        double kzz = gp.lambda;
        double ktv = TL_R(D, 1, p);
        if (lin) {
          lds_barrier();
          if (fin || c == D) {  // (the half that sends needs the sum in its c == D items only)
            const double* fk = ks + p * (D + 1);
            double sk_ = 0.0, sz_ = 0.0;
#pragma unroll 5
            for (int d = 0; d <= D; ++d) {
              sk_ += fk[d];
              sz_ += fk[nit + d];
            }
            ktv = sk_;
            kzz += sz_;
          }
        }
        if (c == D) v0 = ktv;
        if (act) {
          if (rs_dbg) { const unsigned long long now = clock64(); a.stamps[9] += now - rs_t0; rs_t0 = now; }
          if (!fin) {
            const unsigned long long b0 = (unsigned long long)__double_as_longlong(v0), b1 = (unsigned long long)__double_as_longlong(v1);
            store_granule(sl, (unsigned)t + 1u, (unsigned)b0);
            store_granule(sl + nit, (unsigned)t + 1u, (unsigned)(b0 >> 32));
            store_granule(sl + 2 * nit, (unsigned)t + 1u, (unsigned)b1);
            store_granule(sl + 3 * nit, (unsigned)t + 1u, (unsigned)(b1 >> 32));
          } else {
          // re-read until every tag matches
          bool ok = false;
          for (unsigned spins = 0; spins < RF_SPIN_LIMIT; ++spins) {
            ok = true;
#pragma unroll
            for (int sd = 0; sd < 2; ++sd)
#pragma unroll
              for (int q = 0; q < 6; ++q) ok = ok && (sd >= RS - 1 || (unsigned)(x[sd][q] >> 32) == (unsigned)t + 1u);
            if (ok) break;
            __builtin_amdgcn_s_sleep(2);
            ask();
          }
          if (!ok) *abortw = 1;
          if (rs_dbg) { const unsigned long long now = clock64(); a.stamps[10] += now - rs_t0; rs_t0 = now; }
          // own + sender 0 (+ sender 1), in that order: the senders' values are folded into the own ones here, the formulas below stay the two-part ones
          double q0 = __longlong_as_double((long long)((x[0][1] << 32) | (x[0][0] & 0xffffffffull)));
          double q1 = __longlong_as_double((long long)((x[0][3] << 32) | (x[0][2] & 0xffffffffull)));
          double qk = __longlong_as_double((long long)((x[0][5] << 32) | (x[0][4] & 0xffffffffull)));
          if (RS > 2) {  // (uniform)
            v0 += q0;
            v1 += q1;
            ktv += qk;
            q0 = __longlong_as_double((long long)((x[1][1] << 32) | (x[1][0] & 0xffffffffull)));
            q1 = __longlong_as_double((long long)((x[1][3] << 32) | (x[1][2] & 0xffffffffull)));
            qk = __longlong_as_double((long long)((x[1][5] << 32) | (x[1][4] & 0xffffffffull)));
          }
          const double var = (kzz - (ktv + qk)) * vscale;
          double eps = 0.0, wj = 0.0, sd = 0.0;
          if (a.particle_pred) {
            eps = epsb[(t & 1) * P * G + p * G + g];
            sd = sqrt(var);
            wj = eps / (2.0 * sd);
          }
          if (c == D) {
            double mu = gp.mean;
            if (MAXDEG >= 1) {
#pragma unroll
              for (int w = 0; w < RF_NW; ++w) mu += mup[w * P + p];
            } else {
              mu += v1 + q1;
            }
            const double dv = a.particle_pred ? fma(sd, eps, mu) : mu;
            dl[p * G + g] = dv;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(dv);
            gu64_t slot = (gu64_t)a.xch + xch_slot(cluster, t, G, g, P) + 2 * p;
            store_granule(slot, (unsigned)t + 1u, (unsigned)bits);
            store_granule(slot + 1, (unsigned)t + 1u, (unsigned)(bits >> 32));
            if (m0 + p < M) {
              if (a.particle_pred && var <= 0.0) bad |= MCP_STATUS_NONPOS_VAR;
              if (is_bad(mu) || is_bad(var)) bad |= MCP_STATUS_NAN;
            }
          } else if (a.jac && m0 + p < M) {
            double Jmu = -2.0 * il2 * (v0 + q0);
            double Jvar = v1 + q1;
            if (MAXDEG >= 1 && deg >= 1) {
              Jmu = fma(w1c, kp[KP_AX(D) + c], Jmu);
              Jvar = fma(2.0 * w1c, zp[c], Jvar);
            }
            store_through(&a.jac[(((size_t)t * M + m0 + p) * G + g) * D + c], a.particle_pred ? fma(wj, Jvar * vscale, Jmu) : Jmu);
          }
          if (rs_dbg) { const unsigned long long now = clock64(); a.stamps[11] += now - rs_t0; rs_t0 = now; }
          }
        }
      } else
      for (int it = tid; it < P * (D + 1); it += RF_NT) {
        const int p = it / (D + 1), c = it - p * (D + 1);
        const double* zp = z + p * D;
        const double vscale = gp.var_scale;
        // k(z,z) and the v-weighted sum  k^T Kinv k
        double kzz = gp.lambda;
        double ktv = TL_R(D, 1, p);
        double Sa = 0.0, Sb = 0.0;
        if (MAXDEG >= 1 && deg >= 1) {
          double p1 = kp[KP_W1(D) + D];
          double pv = kp[KP_W1(D) + D] * TL_R(D, 2, p);
#pragma unroll 4
          for (int d = 0; d < D; ++d) {
            const double wz = kp[KP_W1(D) + d] * zp[d];
            p1 = fma(wz, zp[d], p1);
            pv = fma(wz, TL_R(d, 2, p), pv);
          }
          kzz += p1;
          ktv += pv;
          if (MAXDEG >= 2 && deg >= 2) {
            double qv = 0.0;
#pragma unroll 4
            for (int d = 0; d < D; ++d) {
              const double zz = zp[d] * zp[d];
              Sa = fma(kp[KP_W20(D) + d], zz, Sa);
              Sb = fma(kp[KP_W21(D) + d], zz, Sb);
              qv = fma(kp[KP_W20(D) + d] * zp[d], TL_R(d, 3, p), qv);
            }
            kzz = fma(Sa, Sb, kzz);
            ktv += qv;
          }
        }
        const double var = (kzz - ktv) * vscale;
        double eps = 0.0, wj = 0.0, sd = 0.0;
        if (a.particle_pred) {
          eps = epsb[(t & 1) * P * G + p * G + g];
          sd = sqrt(var);
          wj = eps / (2.0 * sd);
        }
        if (c == D) {
          double mu = gp.mean;
          if (MAXDEG >= 1) {
#pragma unroll
            for (int w = 0; w < RF_NW; ++w) mu += mup[w * P + p];
          } else {
            mu += TL_R(D, 0, p);
          }
          const double dv = a.particle_pred ? fma(sd, eps, mu) : mu;
          dl[p * G + g] = dv;
          if (GSH) {  // publish straight from the register: two granules
            const unsigned long long bits = (unsigned long long)__double_as_longlong(dv);
            gu64_t slot = (gu64_t)a.xch + xch_slot(cluster, t, G, g, P) + 2 * p;
            store_granule(slot, (unsigned)t + 1u, (unsigned)bits);
            store_granule(slot + 1, (unsigned)t + 1u, (unsigned)(bits >> 32));
          }
          if (m0 + p < M) {
            if (a.particle_pred && var <= 0.0) bad |= MCP_STATUS_NONPOS_VAR;  // (finite and not positive: a NaN variance is MCP_STATUS_NAN, the retry case)
            if (is_bad(mu) || is_bad(var)) bad |= MCP_STATUS_NAN;
          }
        } else if (a.jac && m0 + p < M) {
          const double il = kp[KP_INVLS(D) + c];
          const double il2 = il * il;
          const double r0 = fma(zp[c], TL_R(D, 0, p), -TL_R(c, 0, p));
          const double r1 = fma(zp[c], TL_R(D, 1, p), -TL_R(c, 1, p));
          double Jmu = -2.0 * il2 * r0;
          double Jvar = 4.0 * il2 * r1;
          if (MAXDEG >= 1 && deg >= 1) {
            const double w1c = kp[KP_W1(D) + c];
            Jmu = fma(w1c, kp[KP_AX(D) + c], Jmu);
            Jvar += 2.0 * w1c * (zp[c] - TL_R(c, 2, p));
            if (MAXDEG >= 2 && deg >= 2) {
              const double* Q = qa + g * D * D;
              const double a_ = kp[KP_W20(D) + c], b_ = kp[KP_W21(D) + c];
              double qa_ = 0.0, qb_ = 0.0;  // sum_e w21_e z_e Q[c][e],  sum_e w20_e z_e Q[c][e]
#pragma unroll 4
              for (int e = 0; e < D; ++e) {
                qa_ = fma(kp[KP_W21(D) + e] * zp[e], Q[c * D + e], qa_);
                qb_ = fma(kp[KP_W20(D) + e] * zp[e], Q[c * D + e], qb_);
              }
              Jmu += a_ * qa_ + b_ * qb_;
              Jvar += 2.0 * zp[c] * (a_ * Sb + b_ * Sa) - 2.0 * (a_ * TL_R(c, 3, p) + b_ * TL_R(c, 4, p));
            }
          }
          store_through(&a.jac[(((size_t)t * M + m0 + p) * G + g) * D + c], a.particle_pred ? fma(wj, Jvar * vscale, Jmu) : Jmu);
        }
      }
#undef TL_R
      if (gi == gbeg && edraw && t + 1 < T - 1) draw_eps(t + 1);
      if (GSH && wv == 0 && gi == gend - 1) {
        // collect the other workgroups' increments (rollout_fwd.hip): lane -> (other GP, particle, half), 64 granules per pass,
        // each pass re-read until every tag matches
        const int nown = fin ? gend - gbeg : 0, ngr = (G - nown) * P * 2;  // (the half that does not finish its GPs collects them too)
        bool done = true;
        for (int base = 0; base < ngr && done; base += 64) {
          const int idx = base + lane;
          const bool act = idx < ngr;
          const int go = act ? idx / (2 * P) : 0, r = act ? idx - go * 2 * P : 0;
          const int gq = go < gbeg ? go : go + nown;
          gu64_t slot = (gu64_t)a.xch + xch_slot(cluster, t, G, gq, P) + r;
          unsigned val = 0;
          done = false;
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
        }
        if (!done && lane == 0) *abortw = 1;
        if (RSP && RS > 1 && a.stamps && blockIdx.x == a.stamp_block && tid == 0) a.stamps[14] += clock64() - last_stamp;  // (since the end of phase J)
      }
      lds_barrier();  // R, k/v panels and the scratch are reused by the next GP
      TL_STAMP(7);
    }
    if (GSH && *abortw) {  // uniform: a partner never arrived
      bad |= MCP_STATUS_SYNC;
      break;
    }
    // ---- integrate:  v' = v + delta ;  q' = q + Ts v + Ts/2 delta   (Model_learning.py:711-716) ----
    if (own) {
      const double* xc = xs + cur * P * S + op * S;
      double nx = 0.0;
      if (g_vel >= 0) nx = xc[os] + dl[op * G + g_vel];
      if (g_pos >= 0) nx = xc[os] + Ts * xc[vel_of_pos] + 0.5 * Ts * dl[op * G + g_pos];
      xn = nx;
    }
    cur ^= 1;
  }
  if (bad) atomicOr(a.status, bad);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
namespace mcp {

bool fwd_tile_fits(const mcp_model* model, const mcp_policy* policy) {
  if (!model || !policy || model->G < 1) return false;
  if (model->D + 1 > 32 || TL_PT * model->S > 256 || TL_PT * model->G > 256) return false;
  int NpadMax = 0, maxdeg = 0;
  for (int g = 0; g < model->G; ++g) {
    NpadMax = imax(NpadMax, model->gp[g].Npad);
    maxdeg = imax(maxdeg, model->gp[g].kern.poly_deg);
    if (((model->gp[g].Npad + 31) / 32 + RF_NW - 1) / RF_NW > TL_MAXTASK) return false;
  }
  TileLayout L = tile_layout(model->S, model->U, model->D, model->G, policy->P, NpadMax, maxdeg);
  const int RT = (model->D + 1 + 15) / 16;
  if (L.nslot * RT * TL_NCOL(maxdeg) * 256 > (L.total - L.scr)) return false;
  if (policy->P > 32) return false;
  return sizeof(double) * (size_t)L.total <= MCP_LDS_LIMIT;
}

template <int MAXDEG, int CLS, bool PMS, bool XL>
static int launch_tile_pms(const FwdArgs& a, size_t lds, hipStream_t st) {
  MCP_ENSURE_MAX_LDS(rollout_fwd_tile_kernel<MAXDEG, CLS, PMS, false, XL>);
  const int grid = (a.M + TL_PT - 1) / TL_PT;
  hipLaunchKernelGGL((rollout_fwd_tile_kernel<MAXDEG, CLS, PMS, false, XL>), dim3(grid), dim3(RF_NT), lds, st, a);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

template <int MAXDEG, int CLS, bool PMS, bool XL>
static int launch_tile_gsh(const FwdArgs& a, size_t lds, hipStream_t st) {
  MCP_ENSURE_MAX_LDS(rollout_fwd_tile_kernel<MAXDEG, CLS, PMS, true, XL>);
  const int rs = a.gsh_rs > 1 ? a.gsh_rs : 1;
  const int grid = a.gsh_map == 1 ? ((a.nclusters * a.gsh_cs * rs + 7) / 8) * 8 : ((a.nclusters + 7) / 8) * 8 * a.gsh_cs * rs;
  hipLaunchKernelGGL((rollout_fwd_tile_kernel<MAXDEG, CLS, PMS, true, XL>), dim3(grid), dim3(RF_NT), lds, st, a);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}
// the cart-pole class stages X^T / alpha in LDS when the layout has the room for it (tile_layout decides; the wider classes never have)
template <int MAXDEG, int CLS>
static int launch_tile_gsh_deg(const FwdArgs& a, hipStream_t st) {
  const TileLayout L = tile_layout(a.model.S, a.model.U, a.model.D, a.model.G, a.pol.P, a.NpadMax, a.maxdeg, CLS == 0);
  const size_t lds = sizeof(double) * (size_t)L.total;
  const bool pms = a.pol.meas.n > 0;
  if (CLS == 0 && L.xl) return pms ? launch_tile_gsh<MAXDEG, CLS, true, CLS == 0>(a, lds, st) : launch_tile_gsh<MAXDEG, CLS, false, CLS == 0>(a, lds, st);
  return pms ? launch_tile_gsh<MAXDEG, CLS, true, false>(a, lds, st) : launch_tile_gsh<MAXDEG, CLS, false, false>(a, lds, st);
}

template <int MAXDEG, int CLS>
static int launch_tile_deg(const FwdArgs& a, hipStream_t st) {
  const TileLayout L = tile_layout(a.model.S, a.model.U, a.model.D, a.model.G, a.pol.P, a.NpadMax, a.maxdeg, CLS == 0);
  const size_t lds = sizeof(double) * (size_t)L.total;
  const bool pms = a.pol.meas.n > 0;
  if (CLS == 0 && L.xl) return pms ? launch_tile_pms<MAXDEG, CLS, true, CLS == 0>(a, lds, st) : launch_tile_pms<MAXDEG, CLS, false, CLS == 0>(a, lds, st);
  return pms ? launch_tile_pms<MAXDEG, CLS, true, false>(a, lds, st) : launch_tile_pms<MAXDEG, CLS, false, false>(a, lds, st);
}

// builds the packed phase-J operand in the caller's workspace (wide classes with a workspace; a few microseconds per rollout)
static int tile_xj_pack(const FwdArgs& a, hipStream_t st) {
  if (!a.xj || (a.operands_packed & 2)) return MCP_OK;
  const int npb = a.xj_stride / 512;
  hipLaunchKernelGGL(tile_xj_pack_kernel, dim3((npb * 128 + 255) / 256, a.model.G, 4), dim3(256), 0, st, a.model, a.xj, a.xj_stride, npb);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}
int launch_fwd_tile(const FwdArgs& a, hipStream_t st) {
  if (!fwd_tile_fits(&a.model, &a.pol)) return MCP_ERR_LIMIT;
  if (tile_xj_pack(a, st) != MCP_OK) return MCP_ERR_LAUNCH;
  const int D = a.model.D, PF = a.pol.P, U = a.model.U;
  const int cls = a.NpadMax > 512 ? 2 : ((D <= 7 && PF <= 8 && U <= 2) ? 0 : ((D <= 24 && PF <= 24 && U <= 6) ? 1 : 2));  // (class 0: D + 1 <= 8 rows of [X^T; 1], phase J)
  // one instantiation per (highest polynomial degree, class): no code or registers for kernel terms the model does not have
  switch (cls * 3 + a.maxdeg) {
    case 0: return launch_tile_deg<0, 0>(a, st);
    case 1: return launch_tile_deg<1, 0>(a, st);
    case 2: return launch_tile_deg<2, 0>(a, st);
    case 3: return launch_tile_deg<0, 1>(a, st);
    case 4: return launch_tile_deg<1, 1>(a, st);
    case 5: return launch_tile_deg<2, 1>(a, st);
    case 6: return launch_tile_deg<0, 2>(a, st);
    case 7: return launch_tile_deg<1, 2>(a, st);
    default: return launch_tile_deg<2, 2>(a, st);
  }
}

// GP-sharded launch of the 16-particle kernel (a.gsh_cs workgroups per tile, a.xch, a.nclusters set by the caller): the cart-pole
// and the UR5 register classes
int launch_fwd_tile_sharded(const FwdArgs& a, hipStream_t st) {
  if (!fwd_tile_fits(&a.model, &a.pol) || a.gsh_cs < 2 || a.gsh_cs > a.model.G || !a.xch) return MCP_ERR_LIMIT;
  const int D = a.model.D, PF = a.pol.P, U = a.model.U;
  if (a.NpadMax > 512) return MCP_ERR_LIMIT;
  const int cls = (D <= 7 && PF <= 8 && U <= 2) ? 0 : ((D <= 24 && PF <= 24 && U <= 6) ? 1 : 2);
  if (cls == 2) return MCP_ERR_LIMIT;
  if (a.gsh_map == 1 && (a.gsh_rs < 2 || cls == 0 || a.maxdeg > 1)) return MCP_ERR_ARG;  // (the row-part-major deal exists where the row split does)
  if (a.gsh_rs > 1 && (a.gsh_rs > 3 || cls == 0 || a.maxdeg > 1 || !a.xj || !a.rxch || TL_PT * (D + 1) > RF_NT)) return MCP_ERR_ARG;  // (the row split exists in the per-tile form of phase J only)
  if (tile_xj_pack(a, st) != MCP_OK) return MCP_ERR_LAUNCH;
  switch (cls * 3 + a.maxdeg) {
    case 0: return launch_tile_gsh_deg<0, 0>(a, st);
    case 1: return launch_tile_gsh_deg<1, 0>(a, st);
    case 2: return launch_tile_gsh_deg<2, 0>(a, st);
    case 3: return launch_tile_gsh_deg<0, 1>(a, st);
    case 4: return launch_tile_gsh_deg<1, 1>(a, st);
    default: return launch_tile_gsh_deg<2, 1>(a, st);
  }
}

}  // namespace mcp
