// Fused Monte-Carlo particle rollout, forward pass, large-swarm variant for gfx950 (MI355X):
// 16 particles per workgroup, the two N-long contractions of every GP on the matrix cores.
//
// Same contract as rollout_fwd.hip (MC_PILCO.apply_policy, policy_learning/MC_PILCO.py:615-674; per step
// Policy.py:242-265 / 323-335 / 389-403, Model_learning.py:210-242 / 265-336 / 685-718, GP_prior.py:137-155) and the
// same outputs (states, inputs, d delta/dz, status).  What changes is the mapping: with >= ~1000 particles per GPU
// the small-tile kernel re-streams Kinv (720 KB per GP at N=300) for every 4 particles and every per-particle
// phase runs at a fraction of a wave.  Here a 512-thread workgroup owns a 16-particle tile:
//
//   K   k[j][p] = k(z_p, X_j)                  wave w <-> particles (2w, 2w+1), lanes <-> j, X_j read once per lane
//   V   v = Kinv k  ([N x N] x [N x 16])       v_mfma_f64_16x16x4_f64: A = 32x4 panel of Kinv (one 16-byte load per
//                                              lane feeds two MFMAs: even / odd rows), B = k[j..j+3][0..15] from LDS;
//                                              32-row blocks are dealt to the 8 waves, accumulators stay in registers
//                                              and overwrite k in LDS once every wave is done reading it
//   J   R = [X^T;1] W  ((D+1) x N x 16*ncol)   the moment / Jacobian sums as one skinny matrix product per GP, the 8
//                                              waves split N and add their partial tiles through LDS in a fixed order
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
#define TL_NCOL(deg) ((deg) == 0 ? 2 : ((deg) == 1 ? 3 : 5))

struct TileLayout {
  int invl, xs, us, z, sf, dl, eps, ks, kv, R, qa, sal, gpl, kpar, scr, total;  // offsets in doubles
  int nslot;   // phase-J partial-tile slots in scr
  int vslots;  // phase-V partial (32x16) slots in scr
};

__host__ __device__ inline TileLayout tile_layout(int S, int U, int D, int G, int PF, int NpadMax, int maxdeg) {
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
  L.dl = take(TL_PT * G);
  L.eps = take(2 * TL_PT * G);  // process noise of steps t (read by phase F) and t+1 (drawn by idle threads of phase F)
  L.ks = take(NpadMax * TL_KR);
  L.kv = take(NpadMax * TL_KR);  // directly after ks: the phi buffer of the policy phase aliases both
  const int RT = (D + 1 + 15) / 16, CT = TL_NCOL(maxdeg);
  L.R = take(RT * CT * 256);
  L.qa = take(maxdeg >= 2 ? G * D * D : 0);
  L.sal = take(G);
  L.gpl = take(G * GPL_DOUBLES);
  L.kpar = take(G * KP_STRIDE(D));
  const int slot = RT * CT * 256;
  const int avail = MCP_LDS_LIMIT / 8 - o;
  int nslot = 4;
  while (nslot > 1 && nslot * slot > avail) nslot >>= 1;
  int scr = nslot * slot;
  if (scr < 7 * 512 && 7 * 512 <= avail) scr = 7 * 512;
  L.nslot = nslot;
  L.vslots = scr / 512;
  L.scr = take(scr);
  L.total = o;
  return L;
}

#define TL_STAMP(k) RF_STAMP(k)

// ---------------------------------------------------------------------------------------
// phase K: wave w computes k(z_p, X_j) for p in {2w, 2w+1} and all j (lanes over j)
// ---------------------------------------------------------------------------------------
// The training inputs come from global memory (L1/L2 hits, one coalesced row segment per feature); the loads of a
// step (64 training points x 8 features) are issued unconditionally, one step ahead of their use.
#define TL_KD 8  // features per step
struct TileKAcc {
  double d0, d1, p10, p11, A0, A1, B0, B1;
};
__device__ __forceinline__ void tile_k_load(double (&xv)[TL_KD], gptr_t Xt, int Npad, int D, int nbd, int step, int lane) {
  const int chunk = step / nbd, db = (step - chunk * nbd) * TL_KD;
  const int jl = imin(chunk * 64 + lane, Npad - 1);
#pragma unroll
  for (int i = 0; i < TL_KD; ++i) xv[i] = Xt[(size_t)imin(db + i, D - 1) * Npad + jl];
}
template <int MAXDEG>
__device__ __forceinline__ void tile_k_consume(const double (&xv)[TL_KD], TileKAcc& q, const GpL& gp, const double* kp, int deg, int N, int Npad, int D,
                                               int nbd, int step, const double* za, const double* zb, double* ks, double* kv, int wv, int lane) {
  const int chunk = step / nbd, db = (step - chunk * nbd) * TL_KD;
  const int j = chunk * 64 + lane;
  if (db == 0) {
    q.d0 = q.d1 = q.A0 = q.A1 = q.B0 = q.B1 = 0.0;
    q.p10 = q.p11 = (MAXDEG >= 1 && deg >= 1) ? kp[KP_W1(D) + D] : 0.0;
  }
#pragma unroll
  for (int i = 0; i < TL_KD; ++i) {
    const int d = db + i;
    if (d < D) {  // wave-uniform
      const double x = xv[i];
      const double il = kp[KP_INVLS(D) + d];
      const double z0 = za[d], z1 = zb[d];
      const double r0 = (z0 - x) * il, r1 = (z1 - x) * il;
      q.d0 = fma(r0, r0, q.d0);
      q.d1 = fma(r1, r1, q.d1);
      if (MAXDEG >= 1 && deg >= 1) {
        const double w1 = kp[KP_W1(D) + d];
        q.p10 = fma(w1 * z0, x, q.p10);
        q.p11 = fma(w1 * z1, x, q.p11);
        if (deg >= 2) {
          const double zx0 = z0 * x, zx1 = z1 * x;
          const double wa = kp[KP_W20(D) + d], wb = kp[KP_W21(D) + d];
          q.A0 = fma(wa, zx0, q.A0);
          q.B0 = fma(wb, zx0, q.B0);
          q.A1 = fma(wa, zx1, q.A1);
          q.B1 = fma(wb, zx1, q.B1);
        }
      }
    }
  }
  if (db + TL_KD >= D && j < Npad) {
    v2d s2, t2;
    s2.x = s2.y = t2.x = t2.y = 0.0;
    if (j < N) {
      const double lam = gp.lambda;
      s2.x = lam * exp(-q.d0);
      s2.y = lam * exp(-q.d1);
      t2 = s2;
      if (MAXDEG >= 1 && deg >= 1) {
        t2.x += q.p10;
        t2.y += q.p11;
        if (deg >= 2) {
          t2.x = fma(q.A0, q.B0, t2.x);
          t2.y = fma(q.A1, q.B1, t2.y);
        }
      }
    }
    *reinterpret_cast<v2d*>(ks + j * TL_KR + 2 * wv) = s2;
    *reinterpret_cast<v2d*>(kv + j * TL_KR + 2 * wv) = t2;
  }
}
template <int MAXDEG>
__device__ __forceinline__ void tile_phase_k(const GpL& gp, const double* kp, int D, const double* z, double* ks, double* kv, int wv, int lane) {
  const int N = __builtin_amdgcn_readfirstlane(gp.N), Npad = __builtin_amdgcn_readfirstlane(gp.Npad);
  const int deg = MAXDEG == 0 ? 0 : __builtin_amdgcn_readfirstlane(gp.deg);
  gptr_t Xt = (gptr_t)gp.Xt;
  const double* za = z + (2 * wv) * D;
  const double* zb = za + D;
  const int nbd = (D + TL_KD - 1) / TL_KD;
  const int nsteps = ((Npad + 63) >> 6) * nbd;
  double xa[TL_KD], xb[TL_KD];
  TileKAcc q;
  q.d0 = q.d1 = q.p10 = q.p11 = q.A0 = q.A1 = q.B0 = q.B1 = 0.0;
  tile_k_load(xa, Xt, Npad, D, nbd, 0, lane);
  for (int st = 0; st + 1 < nsteps; st += 2) {
    tile_k_load(xb, Xt, Npad, D, nbd, st + 1, lane);
    tile_k_consume<MAXDEG>(xa, q, gp, kp, deg, N, Npad, D, nbd, st, za, zb, ks, kv, wv, lane);
    tile_k_load(xa, Xt, Npad, D, nbd, imin(st + 2, nsteps - 1), lane);
    tile_k_consume<MAXDEG>(xb, q, gp, kp, deg, N, Npad, D, nbd, st + 1, za, zb, ks, kv, wv, lane);
  }
  if (nsteps & 1) tile_k_consume<MAXDEG>(xa, q, gp, kp, deg, N, Npad, D, nbd, nsteps - 1, za, zb, ks, kv, wv, lane);
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
template <int DEG, int NDQ>
__device__ __forceinline__ void tile_j_load(TileJBatch<DEG, NDQ>& b, gptr_t Xt, gptr_t al, int Npad, int D, int RT, int cc0, int cc1, int jb, int kk,
                                            int n) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = imin(jb + 4 * u + kk, Npad - 1);
    b.a0[u] = Xt[(size_t)cc0 * Npad + j];
    b.a1[u] = RT > 1 ? Xt[(size_t)cc1 * Npad + j] : 0.0;
    b.al[u] = al[j];
  }
  if (DEG >= 2) {
    const int jr = imin(jb + n, Npad - 1);
#pragma unroll
    for (int i = 0; i < NDQ; ++i) b.xq[i] = Xt[(size_t)imin(4 * i + kk, D - 1) * Npad + jr];
  }
}
template <int DEG, int NDQ>
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
  const int c0 = n, c1 = 16 + n;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const double av0 = c0 < D ? b.a0[u] : (c0 == D ? 1.0 : 0.0);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[0][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av0, bv[u][ct], acc[0][ct], 0, 0, 0);
    if (RT > 1) {
      const double av1 = c1 < D ? b.a1[u] : (c1 == D ? 1.0 : 0.0);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[1][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av1, bv[u][ct], acc[1][ct], 0, 0, 0);
    }
  }
}
template <int DEG, int NDQ>
__device__ __forceinline__ void tile_phase_j(const GpL& gp, const double* kp, int D, const double* z, const double* ks, const double* kv,
                                             v4d (&acc)[2][TL_NCOL(DEG)], int wv, int lane) {
  constexpr int CT = TL_NCOL(DEG);
  const int RT = (D + 1 + 15) >> 4;
  const int Npad = __builtin_amdgcn_readfirstlane(gp.Npad);
  const int per = ((Npad + RF_NW * 4 - 1) / (RF_NW * 4)) * 4;
  const int j0 = imin(wv * per, Npad), j1 = imin(Npad, j0 + per);
  const int kk = lane >> 4, n = lane & 15;
  gptr_t Xt = (gptr_t)gp.Xt;
  gptr_t al = (gptr_t)gp.alpha;
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
  const int cc0 = imin(n, D - 1), cc1 = imin(16 + n, D - 1);
  const int nbat = (j1 - j0 + 15) >> 4;
  TileJBatch<DEG, NDQ> b0, b1;
  tile_j_load<DEG, NDQ>(b0, Xt, al, Npad, D, RT, cc0, cc1, j0, kk, n);
  for (int b = 0; b + 1 < nbat; b += 2) {
    const int ja = j0 + 16 * b;
    tile_j_load<DEG, NDQ>(b1, Xt, al, Npad, D, RT, cc0, cc1, ja + 16, kk, n);
    tile_j_consume<DEG, NDQ>(b0, zwa, zwb, D, RT, Npad, ja, j1, kk, n, ks, kv, acc);
    tile_j_load<DEG, NDQ>(b0, Xt, al, Npad, D, RT, cc0, cc1, j0 + 16 * imin(b + 2, nbat - 1), kk, n);
    tile_j_consume<DEG, NDQ>(b1, zwa, zwb, D, RT, Npad, ja + 16, j1, kk, n, ks, kv, acc);
  }
  if (nbat & 1) tile_j_consume<DEG, NDQ>(b0, zwa, zwb, D, RT, Npad, j0 + 16 * (nbat - 1), j1, kk, n, ks, kv, acc);
}

// add the 8 waves' partial tiles in a fixed order (pairwise tree through `nslot` LDS slots), result -> R
template <int CT>
__device__ __forceinline__ void tile_j_reduce(v4d (&acc)[2][CT], int RT, double* scr, int nslot, double* R, int wv, int lane) {
  const int ntile = RT * CT;
  const int slot = ntile * 256;
  for (int active = RF_NW; active > 1; active >>= 1) {
    const int half = active >> 1;
    for (int base = 0; base < half; base += nslot) {
      // writers: waves half+base .. half+base+nslot-1 ; readers: waves base .. base+nslot-1
      const int wi = wv - half - base, ri = wv - base;
      if (wi >= 0 && wi < nslot && wv < active) {
        double* s = scr + wi * slot;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          if (rt < RT)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
              for (int r = 0; r < 4; ++r) s[(rt * CT + ct) * 256 + r * 64 + lane] = acc[rt][ct][r];
      }
      lds_barrier();
      if (ri >= 0 && ri < nslot && ri + base < half) {
        const double* s = scr + ri * slot;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          if (rt < RT)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[rt][ct][r] += s[(rt * CT + ct) * 256 + r * 64 + lane];
      }
      lds_barrier();
    }
  }
  if (wv == 0) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
      if (rt < RT)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) R[(rt * CT + ct) * 256 + r * 64 + lane] = acc[rt][ct][r];
  }
}

// R[c][kind][p]: tile (c>>4, kind), element (row c&15, col p) in the accumulator layout  row = (lane>>4) + 4 r, col = lane & 15
__device__ __forceinline__ double tile_r(const double* R, int CT, int c, int kind, int p) {
  return R[((c >> 4) * CT + kind) * 256 + ((c & 15) >> 2) * 64 + ((c & 3) << 4) + p];
}

// ---------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------
template <int MAXDEG>
__global__ __launch_bounds__(RF_NT) void rollout_fwd_tile_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const mcp_model& md = a.model;
  const mcp_policy& pl = a.pol;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = md.S, U = md.U, G = md.G, D = md.D, B = pl.B, PF = pl.P, M = a.M, T = a.T;
  constexpr int P = TL_PT;
  const TileLayout L = tile_layout(S, U, D, G, PF, a.NpadMax, a.maxdeg);
  double* invl = smem + L.invl;
  double* xs = smem + L.xs;
  double* us = smem + L.us;
  double* z = smem + L.z;
  double* sf = smem + L.sf;
  double* dl = smem + L.dl;
  double* epsb = smem + L.eps;
  double* ks = smem + L.ks;
  double* kv = smem + L.kv;
  double* ph = ks;  // policy phase only
  double* R = smem + L.R;
  double* qa = smem + L.qa;
  double* sal = smem + L.sal;
  double* scr = smem + L.scr;
  GpL* gpl = reinterpret_cast<GpL*>(smem + L.gpl);
  double* kpar = smem + L.kpar;
  const int m0 = blockIdx.x * P;
  uint32_t bad = 0;
  const bool drop = pl.p_drop > 0.0;
  const double keep_scale = 1.0 / (1.0 - pl.p_drop);
  const uint32_t drop_thr = drop_threshold(pl.p_drop);
  const int nna = md.n_not_angle, na = md.n_angle;
  const int RT = (D + 1 + 15) >> 4;

  // ---- one-time staging ------------------------------------------------------------------
  for (int it = tid; it < PF; it += RF_NT) invl[it] = exp(-pl.log_ls[it]);
  stage_gp_tables(md.gp, md.var_scale, G, D, gpl, kpar, tid);
  lds_barrier();
  // launch constants of the polynomial terms: sum_j alpha_j and (degree 2) sum_j alpha_j X_jc X_je
  if (MAXDEG >= 1) {
    for (int it = tid; it < G; it += RF_NT) {
      const GpL& gp = gpl[it];
      double s = 0.0;
      for (int j = 0; j < gp.N; ++j) s += gp.alpha[j];
      sal[it] = s;
    }
    if (MAXDEG >= 2 && a.maxdeg >= 2) {  // (the template is instantiated for 0 and 2 only; qa has no storage when maxdeg == 1)
      for (int it = tid; it < G * D * D; it += RF_NT) {
        const int g = it / (D * D), r = it - g * D * D, c = r / D, e = r - c * D;
        const GpL& gp = gpl[g];
        double s = 0.0;
        if (gp.deg >= 2)
          for (int j = 0; j < gp.N; ++j) s = fma(gp.alpha[j] * gp.X[(size_t)j * D + c], gp.X[(size_t)j * D + e], s);
        qa[it] = s;
      }
    }
  }
  gptr_t cen = (gptr_t)pl.centers;
  gptr_t wgt = (gptr_t)pl.weight;

  // thread (p, s) owns state component s of particle p; threads 256.. draw the process noise of the step
  const bool own = tid < P * S;
  const int op = own ? tid / S : 0, os = own ? tid - op * S : 0;
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
  // the last P*G threads draw the process noise of step t+1 while phase F of the first GP keeps only a few waves busy
  const int et = tid - (RF_NT - P * G);
  const bool edraw = et >= 0;
  const int ep = edraw ? et / G : 0, eg = edraw ? et - ep * G : 0;
  auto draw_eps = [&](int tt) {
    double e = 0.0;
    if (a.particle_pred) {
      const int mm = imin(m0 + ep, M - 1);
      e = a.nz.eps ? a.nz.eps[((size_t)tt * M + mm) * G + eg] : philox_normal(a.nz, mm, tt, eg);
    }
    epsb[(tt & 1) * P * G + et] = e;
  };
  if (edraw && T > 1) draw_eps(0);
  const int B4 = (B + 3) >> 2;
  unsigned long long last_stamp = clock64();
  lds_barrier();

  for (int t = 0; t < T; ++t) {
    // ---- phase S: publish x_t, the GP / policy features of each state component; draw eps_t ------
    if (own) {
      double* xc = xs + cur * P * S;
      xc[op * S + os] = xn;
      if (ovalid) {
        a.states[((size_t)t * M + m0 + op) * S + os] = xn;
        if (is_bad(xn)) bad |= MCP_STATUS_NAN;
      }
      double sn = 0.0, cs = 0.0;
      if (zi_ang >= 0 || pi_ang >= 0) sincos(xn, &sn, &cs);
      if (zi_plain >= 0) z[op * D + zi_plain] = xn;
      if (zi_ang >= 0) {
        z[op * D + nna + zi_ang] = sn;
        z[op * D + nna + na + zi_ang] = cs;
      }
      if (pl.kind == MCP_POLICY_ANGLES) {
        if (pi_plain >= 0) sf[op * PF + pi_plain] = xn;
        if (pi_ang >= 0) {
          sf[op * PF + pol_nna + pi_ang] = cs;
          sf[op * PF + pol_nna + pol_na + pi_ang] = sn;
        }
      } else if (pl.kind == MCP_POLICY_TRAJ) {
        sf[op * PF + os] = xn;
        sf[op * PF + S + os] = pl.target_traj[(size_t)t * S + os] - xn;
      } else {
        sf[op * PF + os] = xn;
      }
    }
    lds_barrier();
    TL_STAMP(0);
    // ---- phase PHI: four basis functions per thread share one Philox draw -------------------------
    for (int it = tid; it < P * B4; it += RF_NT) {
      const int p = it / B4, bq = it - p * B4;
      const int mm = imin(m0 + p, M - 1);
      u32x4 rnd = {0, 0, 0, 0};
      if (drop && !a.nz.masks) rnd = philox_draw(a.nz, mm, t, MCP_STREAM_MASK, (uint32_t)bq);
      double dist[4] = {0.0, 0.0, 0.0, 0.0};
      for (int qb = 0; qb < PF; qb += 8) {
        double cv[4][8];  // centres of the 4 basis functions, 8 features: 32 loads in flight
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const size_t row = (size_t)imin(4 * bq + i, B - 1) * PF;
#pragma unroll
          for (int q = 0; q < 8; ++q) cv[i][q] = cen[row + imin(qb + q, PF - 1)];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (qb + q < PF) {
            const double sv = sf[p * PF + qb + q], il = invl[qb + q];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const double r = (sv - cv[i][q]) * il;
              dist[i] = fma(r, r, dist[i]);
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int b = 4 * bq + i;
        if (b < B) {
          double phi = exp(-dist[i]);
          if (drop) {
            const uint32_t word = i == 0 ? rnd.x : i == 1 ? rnd.y : i == 2 ? rnd.z : rnd.w;
            const bool keep = a.nz.masks ? (a.nz.masks[((size_t)t * M + mm) * B + b] != 0) : (word >= drop_thr);
            phi = keep ? phi * keep_scale : 0.0;
          }
          ph[p * B + b] = phi;
        }
      }
    }
    lds_barrier();
    TL_STAMP(1);
    // ---- phase U: one wave per (particle, input) -----------------------------------------------
    for (int task = wv; task < P * U; task += RF_NW) {
      const int p = task / U, k = task - p * U;
      gptr_t wk = wgt + (size_t)k * B;
      double s = 0.0;
#pragma unroll 4
      for (int b = lane; b < B; b += 64) s = fma(wk[b], ph[p * B + b], s);
      s = wave_sum(s);
      if (lane == 0) {
        const double um = pl.u_max[k];
        const double u = pl.squash ? um * tanh(s / um) : s;
        us[p * U + k] = u;
        z[p * D + nna + 2 * na + k] = u;
        if (m0 + p < M) {
          a.inputs[((size_t)t * M + m0 + p) * U + k] = u;
          if (is_bad(u)) bad |= MCP_STATUS_NAN;
        }
      }
    }
    lds_barrier();
    TL_STAMP(2);
    if (t == T - 1) break;

    for (int g = 0; g < G; ++g) {
      const GpL& gp = gpl[g];
      const double* kp = kpar + g * KP_STRIDE(D);
      const int Npad = __builtin_amdgcn_readfirstlane(gp.Npad);
      const int deg = MAXDEG == 0 ? 0 : __builtin_amdgcn_readfirstlane(gp.deg);
      tile_phase_k<MAXDEG>(gp, kp, D, z, ks, kv, wv, lane);
      lds_barrier();
      TL_STAMP(3);
      // ---- phase V ---------------------------------------------------------------------------
      {
        const VSched q = tile_v_sched(Npad, L.vslots);
        v4d acc[TL_MAXTASK][2];
        int blk[TL_MAXTASK];
#pragma unroll
        for (int r = 0; r < TL_MAXTASK; ++r) {
          acc[r][0] = (v4d){0.0, 0.0, 0.0, 0.0};
          acc[r][1] = (v4d){0.0, 0.0, 0.0, 0.0};
          blk[r] = -1;
        }
        const bool in_rem = q.rem > 0 && wv < q.rem * q.s;
        const int part = in_rem ? wv % q.s : 0;
#pragma unroll
        for (int r = 0; r < TL_MAXTASK; ++r) {
          if (r < q.nfull) {
            blk[r] = r * RF_NW + wv;
            tile_v_block(gp.Kinv, Npad, blk[r] * 32, 0, Npad, kv, lane, acc[r][0], acc[r][1]);
          } else if (r == q.nfull && in_rem) {
            blk[r] = q.nfull * RF_NW + wv / q.s;
            const int js = part * q.Jp, je = imin(Npad, js + q.Jp);
            if (js < je) tile_v_block(gp.Kinv, Npad, blk[r] * 32, js, je, kv, lane, acc[r][0], acc[r][1]);
          }
        }
        // partial tiles of the split blocks -> scratch (slot = block-in-remainder * (s-1) + part-1)
#pragma unroll
        for (int r = 0; r < TL_MAXTASK; ++r) {
          if (r == q.nfull && in_rem && part > 0) {
            double* s = scr + ((wv / q.s) * (q.s - 1) + part - 1) * 512;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              s[i * 64 + lane] = acc[r][0][i];
              s[256 + i * 64 + lane] = acc[r][1][i];
            }
          }
        }
        lds_barrier();  // every wave is done reading k: v may overwrite it
        TL_STAMP(4);
#pragma unroll
        for (int r = 0; r < TL_MAXTASK; ++r) {
          if (blk[r] < 0) continue;
          if (r == q.nfull && in_rem) {
            if (part > 0) continue;
            for (int o = 1; o < q.s; ++o) {
              const double* s = scr + ((wv / q.s) * (q.s - 1) + o - 1) * 512;
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                acc[r][0][i] += s[i * 64 + lane];
                acc[r][1][i] += s[256 + i * 64 + lane];
              }
            }
          }
          const int n = lane & 15, kq = lane >> 4;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = blk[r] * 32 + 2 * (kq + 4 * i);
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
      if (MAXDEG == 0 || deg == 0) {
        v4d acc[2][TL_NCOL(0)];
        tile_phase_j<0, 1>(gp, kp, D, z, ks, kv, acc, wv, lane);
        tile_j_reduce<TL_NCOL(0)>(acc, RT, scr, L.nslot, R, wv, lane);
        CTg = TL_NCOL(0);
      } else if (deg == 1) {
        v4d acc[2][TL_NCOL(1)];
        tile_phase_j<1, 1>(gp, kp, D, z, ks, kv, acc, wv, lane);
        tile_j_reduce<TL_NCOL(1)>(acc, RT, scr, L.nslot, R, wv, lane);
        CTg = TL_NCOL(1);
      } else {
        v4d acc[2][TL_NCOL(2)];
        if (D <= 8)
          tile_phase_j<2, 2>(gp, kp, D, z, ks, kv, acc, wv, lane);
        else
          tile_phase_j<2, 8>(gp, kp, D, z, ks, kv, acc, wv, lane);
        tile_j_reduce<TL_NCOL(2)>(acc, RT, scr, L.nslot, R, wv, lane);
        CTg = TL_NCOL(2);
      }
      lds_barrier();
      TL_STAMP(6);
      // ---- phase F: moments, sample, d delta/dz -------------------------------------------------
      for (int it = tid; it < P * (D + 1); it += RF_NT) {
        const int p = it / (D + 1), c = it - p * (D + 1);
        const double* zp = z + p * D;
        const double vscale = gp.var_scale;
        // k(z,z) and the v-weighted sum  k^T Kinv k
        double kzz = gp.lambda;
        double ktv = tile_r(R, CTg, D, 1, p);
        double Sa = 0.0, Sb = 0.0;
        if (MAXDEG >= 1 && deg >= 1) {
          double p1 = kp[KP_W1(D) + D];
          double pv = kp[KP_W1(D) + D] * tile_r(R, CTg, D, 2, p);
          for (int d = 0; d < D; ++d) {
            const double wz = kp[KP_W1(D) + d] * zp[d];
            p1 = fma(wz, zp[d], p1);
            pv = fma(wz, tile_r(R, CTg, d, 2, p), pv);
          }
          kzz += p1;
          ktv += pv;
          if (deg >= 2) {
            double qv = 0.0;
            for (int d = 0; d < D; ++d) {
              const double zz = zp[d] * zp[d];
              Sa = fma(kp[KP_W20(D) + d], zz, Sa);
              Sb = fma(kp[KP_W21(D) + d], zz, Sb);
              qv = fma(kp[KP_W20(D) + d] * zp[d], tile_r(R, CTg, d, 3, p), qv);
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
          double mu = gp.mean + tile_r(R, CTg, D, 0, p);
          if (MAXDEG >= 1 && deg >= 1) {
            double pm = kp[KP_W1(D) + D] * sal[g];
            for (int d = 0; d < D; ++d) pm = fma(kp[KP_W1(D) + d] * zp[d], kp[KP_AX(D) + d], pm);
            mu += pm;
            if (deg >= 2) {
              const double* Q = qa + g * D * D;
              double qm = 0.0;
              for (int d = 0; d < D; ++d) {
                double row = 0.0;
                for (int e = 0; e < D; ++e) row = fma(kp[KP_W21(D) + e] * zp[e], Q[d * D + e], row);
                qm = fma(kp[KP_W20(D) + d] * zp[d], row, qm);
              }
              mu += qm;
            }
          }
          dl[p * G + g] = a.particle_pred ? fma(sd, eps, mu) : mu;
          if (m0 + p < M) {
            if (a.particle_pred && !(var > 0.0)) bad |= MCP_STATUS_NONPOS_VAR;
            if (is_bad(mu) || is_bad(var)) bad |= MCP_STATUS_NAN;
          }
        } else if (a.jac && m0 + p < M) {
          const double il = kp[KP_INVLS(D) + c];
          const double il2 = il * il;
          const double r0 = fma(zp[c], tile_r(R, CTg, D, 0, p), -tile_r(R, CTg, c, 0, p));
          const double r1 = fma(zp[c], tile_r(R, CTg, D, 1, p), -tile_r(R, CTg, c, 1, p));
          double Jmu = -2.0 * il2 * r0;
          double Jvar = 4.0 * il2 * r1;
          if (MAXDEG >= 1 && deg >= 1) {
            const double w1c = kp[KP_W1(D) + c];
            Jmu = fma(w1c, kp[KP_AX(D) + c], Jmu);
            Jvar += 2.0 * w1c * (zp[c] - tile_r(R, CTg, c, 2, p));
            if (deg >= 2) {
              const double* Q = qa + g * D * D;
              const double a_ = kp[KP_W20(D) + c], b_ = kp[KP_W21(D) + c];
              double qa_ = 0.0, qb_ = 0.0;  // sum_e w21_e z_e Q[c][e],  sum_e w20_e z_e Q[c][e]
              for (int e = 0; e < D; ++e) {
                qa_ = fma(kp[KP_W21(D) + e] * zp[e], Q[c * D + e], qa_);
                qb_ = fma(kp[KP_W20(D) + e] * zp[e], Q[c * D + e], qb_);
              }
              Jmu += a_ * qa_ + b_ * qb_;
              Jvar += 2.0 * zp[c] * (a_ * Sb + b_ * Sa) - 2.0 * (a_ * tile_r(R, CTg, c, 3, p) + b_ * tile_r(R, CTg, c, 4, p));
            }
          }
          a.jac[(((size_t)t * M + m0 + p) * G + g) * D + c] = a.particle_pred ? fma(wj, Jvar * vscale, Jmu) : Jmu;
        }
      }
      if (g == 0 && edraw && t + 1 < T - 1) draw_eps(t + 1);
      lds_barrier();  // R, k/v panels and the scratch are reused by the next GP
    }
    TL_STAMP(7);
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
  if (TL_PT * policy->B > 2 * NpadMax * TL_KR) return false;  // the phi buffer aliases the k / v panels
  TileLayout L = tile_layout(model->S, model->U, model->D, model->G, policy->P, NpadMax, maxdeg);
  const int RT = (model->D + 1 + 15) / 16;
  if (L.nslot * RT * TL_NCOL(maxdeg) * 256 > (L.total - L.scr)) return false;
  return sizeof(double) * (size_t)L.total <= MCP_LDS_LIMIT;
}

template <int MAXDEG>
static int launch_tile_deg(const FwdArgs& a, size_t lds, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rollout_fwd_tile_kernel<MAXDEG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              MCP_LDS_LIMIT);
    attr_set = true;
  }
  const int grid = (a.M + TL_PT - 1) / TL_PT;
  hipLaunchKernelGGL((rollout_fwd_tile_kernel<MAXDEG>), dim3(grid), dim3(RF_NT), lds, st, a);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

int launch_fwd_tile(const FwdArgs& a, hipStream_t st) {
  if (!fwd_tile_fits(&a.model, &a.pol)) return MCP_ERR_LIMIT;
  TileLayout L = tile_layout(a.model.S, a.model.U, a.model.D, a.model.G, a.pol.P, a.NpadMax, a.maxdeg);
  const size_t lds = sizeof(double) * (size_t)L.total;
  return a.maxdeg == 0 ? launch_tile_deg<0>(a, lds, st) : launch_tile_deg<2>(a, lds, st);
}

}  // namespace mcp
