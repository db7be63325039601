// Reverse-time adjoint of the fused particle rollout for gfx950 (MI355X).
//
// Replaces autograd's backward through MC_PILCO.apply_policy (policy_learning/MC_PILCO.py:522):
// given dJ/dstates and dJ/dinputs it sweeps t = T-1 .. 0 through the integrator
// (model_learning/Model_learning.py:711-716), the stored GP Jacobians d delta_g/dz (written by
// rollout_fwd.hip; the GP itself is never re-evaluated), the GP and policy feature maps
// (Model_learning.py:670-683, policy_learning/Policy.py:326-333, 397-399) and the RBF network with
// dropout and tanh squashing (Policy.py:242-265, 52-60).
//
// One thread per basis function keeps its rows of dJ/dcenters, dJ/dweight in registers across all
// time steps and all particles the workgroup visits; per-step cross-basis sums (the adjoint of the
// policy features) are wave64 DPP sums meeting in LDS.  Workgroup partial gradients go to a slab
// that grad_reduce_kernel sums in a fixed order (deterministic, no atomics).
#include "rollout_common.h"

using namespace mcp;

// ---------------------------------------------------------------------------------------
// backward rollout: reverse-time adjoint, one thread per basis function
// ---------------------------------------------------------------------------------------
struct BwdArgs {
  mcp_model model;
  mcp_policy pol;
  mcp_noise nz;
  int M, T;
  const double* states;
  const double* inputs;
  const double* jac;
  const double* g_states;
  const double* g_inputs;
  double* slab;  // [gridDim.x][nparam]
  double* g_x0;
  unsigned long long* stamps;  // diagnostic only: per-stage cycle totals of workgroup 0 (slots 8..11)
};

#define BW_STAMP(k)                                 \
  do {                                              \
    if (a.stamps && tid == 0 && blockIdx.x == 0) {  \
      unsigned long long now_ = clock64();          \
      a.stamps[k] += now_ - last_stamp;             \
      last_stamp = now_;                            \
    }                                               \
  } while (0)

struct BwdLayout {
  int invl, rec, xn, xb, zb, db, ab, sf, sb, sn, cs, snm, csm, cmv, cnv, nvx, red, gla, cen, itab, total;
  int pstride;  // doubles between the per-particle copies of xn..cs
};
__host__ __device__ inline int bwd_rec_len(int S, int U, int D, int G, bool pms = false) { return 2 * S + 2 * U + G * D + (pms ? S : 0); }
__host__ __device__ inline BwdLayout bwd_layout(int S, int U, int D, int G, int PF, int NW, int PB, bool pms = false, bool cen_lds = false) {
  BwdLayout L;
  int o = 0;
  auto take = [&](int n) {
    int r = o;
    o += (n + 1) & ~1;
    return r;
  };
  L.invl = take(PF);
  L.rec = take(2 * PB * bwd_rec_len(S, U, D, G, pms));
  // per-particle working set of the serial section (particle p at + p * pstride)
  const int o0 = o;
  L.xn = take(S);
  L.xb = take(S);
  L.zb = take(D);
  L.db = take(G);
  L.ab = take(U);
  L.sf = take(PF);
  L.sb = take(PF);
  L.sn = take(S);
  L.cs = take(S);
  L.snm = take(pms ? S : 0);  // trig of the measured angles (policy features of a partially measurable system)
  L.csm = take(pms ? S : 0);
  L.cmv = take(pms ? S : 0);  // adjoints carried backward through the velocity filter: filtered / finite-difference velocity
  L.cnv = take(pms ? S : 0);
  L.nvx = take(pms ? S : 0);  // hand-over of d/d(noisy velocity) from a velocity lane to its position lane
  L.pstride = o - o0;
  o = o0 + PB * L.pstride;
  L.red = take(PB * NW * PF);
  L.gla = take(PB * (PF + MCP_MAX_INPUT));    // the serial waves' shares of dJ/dlog_lengthscales and dJ/dbias at the end of the sweep
  L.cen = take(cen_lds ? PF * NW * 64 : 0);  // RBF centres, transposed [q][thread] (wide policy classes)
  L.itab = take((2 * MCP_MAX_GP + 2 * MCP_MAX_STATE + MCP_MAX_INPUT + 1) / 2 + 1);
  L.total = o;
  return L;
}

#ifndef BW_LAUNDER
#define BW_LAUNDER(PFM, PB) ((PFM) > 8 || (PB) > 1)  // (narrow class, one particle per workgroup -- the latency-bound small-swarm sweep: measured 7 % slower with it)
#endif
#ifndef BW_MASK_FROM
#define BW_MASK_FROM 8  // classes with PFM beyond this run the RBF stage with chunk-level tests and masked values
#endif
#ifndef BW_WPE_A
#define BW_WPE_A 2
#endif
#define BW_RPT 5  // record elements a thread prefetches at most (PB * record length <= BW_RPT * threads)

// Per particle and time step the record is  [x_t (S) | u_t (U) | dJ/dx_t (S) | dJ/du_t (U) | d delta/dz (G*D)].
// It is prefetched into registers one step ahead (global/L2 latency hidden behind the current step) and
// parked in a double-buffered LDS copy.  A workgroup sweeps PB particles in lockstep: the short dependent chain of
// tiny stages (integrator adjoint, GP-Jacobian product, feature-map adjoints, squashing) of particle p runs in wave p
// alone with wave-level ordering (latency bound, so PB chains on PB waves cost the time of one); the RBF network stage
// uses the whole workgroup, thread b looping over the PB particles (its gradient accumulators are shared by all
// particles anyway): two workgroup barriers per time step for PB particle-steps.
template <int PFM, int UM, int MAXNT, int WPE, int PB>
__global__ __launch_bounds__(MAXNT, WPE) void rollout_bwd_kernel(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const mcp_model& md = a.model;
  const mcp_policy& pl = a.pol;
  int tid = threadIdx.x, lane = tid & 63;  // (not const: laundered once per time step, see the sweep loop)
  const int NT = blockDim.x, NW = NT >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = md.S, U = md.U, G = md.G, D = md.D, B = pl.B, PF = pl.P, M = a.M, T = a.T;
  const mcp_meas& ms = pl.meas;
  const bool pms = ms.n > 0;
  constexpr bool CENREG = PFM <= 8;  // narrow policies keep their centres in registers; wider ones in LDS (transposed: conflict-free)
  const BwdLayout L = bwd_layout(S, U, D, G, PF, NW, PB, pms, !CENREG);
  const int NR = bwd_rec_len(S, U, D, G, pms);
  const int PS = L.pstride;
  const int sp = wv < PB ? wv : 0;  // the particle slot whose serial chain this wave runs
  // Every LDS array is addressed through an explicit address-space-3 pointer.  Through generic pointers the serial section's
  // volatile accesses compile to FLAT loads / stores (57 + 46 in the cart-pole instantiation): slower than ds_read / ds_write on
  // the one dependent chain that bounds the sweep, each one waited for with vmcnt(0) AND lgkmcnt(0); and hipcc 7.2 hoists the
  // LDS -> flat address casts out of the sweep and then fails its own machine verifier on some of them.
  typedef volatile double __attribute__((address_space(3))) * vlds_t;
  typedef const double __attribute__((address_space(3))) * clds_t;
  typedef double __attribute__((address_space(3))) * lds_t;
  lds_t invl_w = (lds_t)(smem + L.invl);
  clds_t invl = (clds_t)(smem + L.invl);
  vlds_t rec = (vlds_t)(smem + L.rec);  // [2][PB][NR]
  vlds_t xn = (vlds_t)(smem + L.xn + sp * PS);  // adjoint of x_{t+1}
  vlds_t xb = (vlds_t)(smem + L.xb + sp * PS);  // adjoint of x_t without the policy path
  vlds_t zb = (vlds_t)(smem + L.zb + sp * PS);
  vlds_t db = (vlds_t)(smem + L.db + sp * PS);
  vlds_t ab = (vlds_t)(smem + L.ab + sp * PS);
  vlds_t sf = (vlds_t)(smem + L.sf + sp * PS);
  vlds_t sb = (vlds_t)(smem + L.sb + sp * PS);
  vlds_t sn = (vlds_t)(smem + L.sn + sp * PS);
  vlds_t cs = (vlds_t)(smem + L.cs + sp * PS);
  vlds_t snm = pms ? (vlds_t)(smem + L.snm + sp * PS) : sn;  // measured angles' trig (== sn/cs without a measurement model)
  vlds_t csm = pms ? (vlds_t)(smem + L.csm + sp * PS) : cs;
  vlds_t cmv = (vlds_t)(smem + L.cmv + sp * PS);
  vlds_t cnv = (vlds_t)(smem + L.cnv + sp * PS);
  vlds_t nvx = (vlds_t)(smem + L.nvx + sp * PS);
  clds_t sf_all = (clds_t)(smem + L.sf);  // particle p at + p * PS
  clds_t ab_all = (clds_t)(smem + L.ab);
  vlds_t red = (vlds_t)(smem + L.red);  // [PB][NW][PF]
  // integer / per-input tables in LDS: indexing the by-value kernel argument with a per-lane index would
  // make the compiler spill it to scratch
  int* t_vel = reinterpret_cast<int*>(smem + L.itab);
  int* t_pos = t_vel + MCP_MAX_GP;
  int* t_pna = t_pos + MCP_MAX_GP;      // policy non_angle[]
  int* t_pan = t_pna + MCP_MAX_STATE;   // policy angle[]
  const int b = tid;
  const bool act = b < B;
  const bool drop = pl.p_drop > 0.0;
  const double keep_scale = 1.0 / (1.0 - pl.p_drop);
  const uint32_t drop_thr = drop_threshold(pl.p_drop);
  const int nna_g = md.n_not_angle, na_g = md.n_angle;
  const int oX = 0, oU = S, oGX = S + U, oGU = 2 * S + U, oJ = 2 * S + 2 * U;
  const int oM = pms ? oJ + G * D : oX;  // measured state (what the policy was evaluated on)

  for (int it = tid; it < PF; it += NT) invl_w[it] = exp(-pl.log_ls[it]);
  if (tid == 0) {
    for (int g = 0; g < G; ++g) {
      t_vel[g] = md.vel[g];
      t_pos[g] = md.not_vel[g];
    }
    for (int i = 0; i < pl.n_non_angle; ++i) t_pna[i] = pl.non_angle[i];
    for (int i = 0; i < pl.n_angle; ++i) t_pan[i] = pl.angle[i];
  }
  // Per-thread state across the whole sweep: the thread's rows of dJ/dcentres and dJ/dweight (and, narrow class only, its
  // centres).  dJ/dlog_lengthscales has NO per-thread accumulator: with rr = (s_q - c_bq)/l_q and t2 = 2 dd rr,
  //   dJ/dlog l_q = sum_b -t2 rr = sum_b (dJ/dc_bq contribution) (s_q - c_bq) = - s_q sb_q - sum_b c_bq (dJ/dc_bq contribution),
  // where sb_q = sum_b t2/l_q is the feature adjoint the serial wave forms anyway: it accumulates -s_q sb_q per step, and the
  // centre term is one product with the finished dJ/dcentres at the end.  (3 PFM -> 2 PFM [1 PFM] doubles of live state per
  // thread: the UR5 class no longer updates spilled accumulators through scratch every step.)
  lds_t cen_l = (lds_t)(smem + L.cen);
  double cen[CENREG ? PFM : 1], gc[PFM], wgt[UM], gw[UM];
#pragma unroll
  for (int q = 0; q < PFM; ++q) {
    if (CENREG) cen[q] = (act && q < PF) ? pl.centers[(size_t)b * PF + q] : 0.0;
    gc[q] = 0.0;
  }
  if (!CENREG)
    for (int q = 0; q < PF; ++q) cen_l[q * NT + tid] = act ? pl.centers[(size_t)b * PF + q] : 0.0;
#define BW_CEN(q) (CENREG ? cen[CENREG ? (q) : 0] : cen_l[(q) * NT + tid])
#pragma unroll
  for (int k = 0; k < UM; ++k) {
    wgt[k] = (act && k < U) ? pl.weight[(size_t)k * B + b] : 0.0;
    gw[k] = 0.0;
  }
  // lane-private index tables of the serial section (lane = state / feature index)
  const bool serial = wv < PB;
  int zi_plain = -1, zi_ang = -1, pi_plain = -1, pi_ang = -1, g_vel = -1, g_pos = -1;
  if (serial && lane < S) {
    for (int i = 0; i < nna_g; ++i)
      if (md.not_angle[i] == lane) zi_plain = i;
    for (int i = 0; i < na_g; ++i)
      if (md.angle[i] == lane) zi_ang = i;
    if (pl.kind == MCP_POLICY_ANGLES) {
      for (int i = 0; i < pl.n_non_angle; ++i)
        if (pl.non_angle[i] == lane) pi_plain = i;
      for (int i = 0; i < pl.n_angle; ++i)
        if (pl.angle[i] == lane) pi_ang = i;
    }
    for (int g = 0; g < G; ++g) {
      if (md.vel[g] == lane) g_vel = g;
      if (md.not_vel[g] == lane) g_pos = g;
    }
  }
  int pm_pos = -1, pm_vel = -1;  // measurement model: index of this lane's state in pos_indeces / vel_indeces
  if (pms && serial && lane < S) {
    for (int i = 0; i < ms.n; ++i) {
      if (ms.pos[i] == lane) pm_pos = i;
      if (ms.vel[i] == lane) pm_vel = i;
    }
  }
  const double pm_a = pms ? -ms.a1 / ms.a0 : 0.0, pm_b0 = pms ? ms.b0 / ms.a0 : 0.0, pm_b1 = pms ? ms.b1 / ms.a0 : 0.0;
  int pos_of_vel = 0;  // the position state integrated from this lane's velocity state
  for (int g = 0; g < G; ++g)
    if (md.vel[g] == lane) pos_of_vel = md.not_vel[g];
  const double umax_lane = (serial && lane < U) ? pl.u_max[lane] : 1.0;
  const bool need_trig = (zi_ang >= 0) || (pi_ang >= 0);
  lds_barrier();

  const int NRP = PB * NR;
  // particles of this sweep: m_p = mbase + p (clamped for the loads; slots past M contribute nothing)
  // the record traffic (address arithmetic for BW_RPT elements, ~1.3 k cycles per step) is kept off the serial waves when the
  // workgroup has others: measured on wave 0's critical path before
  const bool pf_split = NW > PB;
  int pf_tid = pf_split ? tid - 64 * PB : tid;  // (recomputed from the laundered id every step)
  const int pf_nt = pf_split ? NT - 64 * PB : NT;
  auto prefetch = [&](double (&pre)[BW_RPT], int t, int mbase) {
#pragma unroll
    for (int k = 0; k < BW_RPT; ++k) {
      const int e = pf_tid + k * pf_nt;
      double v = 0.0;
      if (pf_tid >= 0 && e < NRP) {
        const int p = e / NR, i = e - p * NR;
        const size_t tm = (size_t)t * M + imin(mbase + p, M - 1);
        if (i < oU)
          v = a.states[tm * S + i];
        else if (i < oGX)
          v = a.inputs[tm * U + (i - oU)];
        else if (i < oGU)
          v = a.g_states ? a.g_states[tm * S + (i - oGX)] : 0.0;
        else if (i < oJ)
          v = a.g_inputs ? a.g_inputs[tm * U + (i - oGU)] : 0.0;
        else if (i < oJ + G * D)
          v = (t < T - 1) ? a.jac[tm * G * D + (i - oJ)] : 0.0;
        else
          v = ms.meas[tm * S + (i - oJ - G * D)];
      }
      pre[k] = v;
    }
  };
  auto park = [&](const double (&pre)[BW_RPT], int buf) {
#pragma unroll
    for (int k = 0; k < BW_RPT; ++k) {
      const int e = pf_tid + k * pf_nt;
      if (pf_tid >= 0 && e < NRP) rec[buf * NRP + e] = pre[k];
    }
  };

  // Adjoint of the measurement model (MC_PILCO.py:881-899) at step tt, lane = state index.  In: s = dJ/d(measured state);
  // out: the part of dJ/dx_tt that flows through the measurement.  With mv_t = (b0 nv_t + b1 nv_{t-1} - a1 mv_{t-1})/a0 and
  // nv_t = (np_t - np_{t-1})/Ts, np_t = x_t[pos] + noise:   mvb_t = s_vel - a1/a0 mvb_{t+1};   nvb_t = b0/a0 mvb_t + b1/a0 mvb_{t+1};
  // x_t[pos] gets s_pos + (nvb_t - nvb_{t+1})/Ts, x_t[vel] nothing -- except at t = 0, where the measurement is the true
  // state: x_0[vel] gets mvb_0 + b1/a0 mvb_1 and x_0[pos] gets s_pos - nvb_1/Ts.  cmv / cnv carry mvb_{t+1}, nvb_{t+1}.
  auto meas_adjoint = [&](double s_in, int tt) -> double {
    double sx = s_in, mvb = 0.0, nvb = 0.0;
    if (pm_vel >= 0) {
      const double c1 = cmv[pm_vel];
      mvb = fma(pm_a, c1, s_in);
      nvb = tt >= 1 ? fma(pm_b0, mvb, pm_b1 * c1) : 0.0;
      sx = tt == 0 ? fma(pm_b1, c1, mvb) : 0.0;
      nvx[pm_vel] = nvb;
    }
    __builtin_amdgcn_wave_barrier();
    if (pm_pos >= 0) sx = s_in + (nvx[pm_pos] - cnv[pm_pos]) / md.Ts;
    __builtin_amdgcn_wave_barrier();
    if (pm_vel >= 0) {
      cmv[pm_vel] = mvb;
      cnv[pm_vel] = nvb;
    }
    return sx;
  };

  double glacc = 0.0;  // serial waves, lane q < PF: - sum over this wave's particle-steps of s_q sb_q
  double gbacc = 0.0;  // serial waves, lane k < U: dJ/dbias_k = sum over this wave's particle-steps of the pre-squash adjoint
  double fprev = 0.0;  // the policy feature this lane formed in the previous iteration of the sweep (= of step t+1)
  unsigned long long last_stamp = clock64();
  for (int mbase = blockIdx.x * PB; mbase < M; mbase += gridDim.x * PB) {
    const int msp = imin(mbase + sp, M - 1);  // particle of this wave's serial chain
    const bool spvalid = mbase + sp < M;
    double pre[BW_RPT];
    int cur = 0;
    if (pms && serial && lane < S) {
      cmv[lane] = 0.0;
      cnv[lane] = 0.0;
    }
    prefetch(pre, T - 1, mbase);
    park(pre, cur);
    lds_barrier();
    for (int t = T - 1; t >= 0; --t) {
      // thread / lane ids are laundered per step: what the unrolled feature loops derive from them (LDS addresses, predicates) is
      // recomputed where it is used instead of being hoisted out of the sweep, kept live next to the accumulators and spilled
      if (BW_LAUNDER(PFM, PB)) {
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        pf_tid = pf_split ? tid - 64 * PB : tid;
      }
      const int b = tid;
      BW_STAMP(11);
      if (t > 0) prefetch(pre, t - 1, mbase);
      // ---- serial section: wave p for particle slot p -----------------------------------------------
      if (serial) {
        // the record of this step was parked before the last workgroup barrier and is not written again until the next one:
        // plain loads (the compiler may batch them), unlike the section's own scratch arrays, which need program order
        clds_t r = (clds_t)(smem + L.rec) + cur * NRP + sp * NR;
        const vlds_t redp = red + sp * NW * PF;
        const bool last = (t == T - 1);
        if (!last) {
          // finish step t+1: adjoint of the policy features -> adjoint of x_{t+1}
          if (lane < PF) {
            double s = 0.0;
            for (int w = 0; w < NW; ++w) s += redp[w * PF + lane];
            sb[lane] = s;
            glacc = fma(-fprev, s, glacc);  // fprev: this lane's policy feature of step t+1
          }
          __builtin_amdgcn_wave_barrier();
          if (lane < S) {
            double s;
            if (pl.kind == MCP_POLICY_ANGLES) {
              s = (pi_plain >= 0) ? sb[pi_plain] : 0.0;
              if (pi_ang >= 0) s += -sb[pl.n_non_angle + pi_ang] * snm[lane] + sb[pl.n_non_angle + pl.n_angle + pi_ang] * csm[lane];
            } else if (pl.kind == MCP_POLICY_TRAJ) {
              s = sb[lane] - sb[S + lane];
            } else {
              s = sb[lane];
            }
            if (pms) s = meas_adjoint(s, t + 1);
            xn[lane] = xb[lane] + s;
          }
          __builtin_amdgcn_wave_barrier();
          // through the integrator:  x_{t+1}[vel] = x[vel] + delta ; x_{t+1}[pos] = x[pos] + Ts x[vel] + Ts/2 delta
          if (lane < G) db[lane] = xn[t_vel[lane]] + 0.5 * md.Ts * xn[t_pos[lane]];
        }
        // trig of this step's angles (used now by the GP feature map, next iteration by the policy's)
        if (lane < S && need_trig) {
          double sv, cv;
          sincos(r[oX + lane], &sv, &cv);
          sn[lane] = sv;
          cs[lane] = cv;
        }
        if (pms && lane < S && pi_ang >= 0) {
          double sv, cv;
          sincos(r[oM + lane], &sv, &cv);
          snm[lane] = sv;
          csm[lane] = cv;
        }
        __builtin_amdgcn_wave_barrier();
        // through the GP Jacobian and the integrator's direct paths
        if (lane < D) {
          double s = 0.0;
          if (!last)
            for (int g = 0; g < G; ++g) s = fma(db[g], r[oJ + g * D + lane], s);
          zb[lane] = s;
        }
        double xbv = 0.0;
        if (lane < S) {
          xbv = r[oGX + lane];
          if (!last) {
            if (g_vel >= 0) xbv += xn[lane] + md.Ts * xn[pos_of_vel];
            if (g_pos >= 0) xbv += xn[lane];
          }
        }
        // policy features of x_t  (Policy.py:326-333: [x_nonangle, COS, SIN];  :397-399: [x, x*_t - x])
        if (lane < PF) {
          double f;
          if (pl.kind == MCP_POLICY_ANGLES) {
            const int pn = pl.n_non_angle, pa_ = pl.n_angle;
            if (lane < pn)
              f = r[oM + t_pna[lane]];
            else if (lane < pn + pa_)
              f = csm[t_pan[lane - pn]];
            else
              f = snm[t_pan[lane - pn - pa_]];
          } else if (pl.kind == MCP_POLICY_TRAJ) {
            f = (lane < S) ? r[oM + lane] : pl.target_traj[(size_t)t * S + (lane - S)] - r[oM + lane - S];
          } else {
            f = r[oM + lane];
          }
          sf[lane] = f;
          fprev = f;
        }
        __builtin_amdgcn_wave_barrier();
        // through the GP feature map z=[x_na, sin, cos, u]; adjoint of the pre-squash activation
        if (lane < S) {
          if (zi_plain >= 0) xbv += zb[zi_plain];
          if (zi_ang >= 0) xbv += zb[nna_g + zi_ang] * cs[lane] - zb[nna_g + na_g + zi_ang] * sn[lane];
          xb[lane] = xbv;
        }
        if (lane < U) {
          double ubar = r[oGU + lane] + zb[nna_g + 2 * na_g + lane];
          double th = r[oU + lane] / umax_lane;  // = tanh(a/u_max)
          const double abv = pl.squash ? ubar * (1.0 - th * th) : ubar;
          ab[lane] = abv;
          if (spvalid) gbacc += abv;
        }
      }
      BW_STAMP(8);
      lds_barrier();
      BW_STAMP(9);
      // ---- RBF network, thread b owns basis b, loops over the particle slots -----------------------------
      // dropout keep bits: one Philox draw serves 4 consecutive bases of one particle (philox_keep); the lanes of a quad
      // draw for particle slots (b & 3) % PB and pass each other the word of the receiver's basis
      uint32_t kw[PB];
#pragma unroll
      for (int p = 0; p < PB; ++p) kw[p] = 0xFFFFFFFFu;
      if (drop && !a.nz.masks) {
        const int cq = b & 3;
        const int bq = imin(b, B - 1) >> 2;
        const u32x4 rnd = philox_draw(a.nz, imin(mbase + (cq % PB), M - 1), t, MCP_STREAM_MASK, (uint32_t)bq);
        if (PB == 1) {
          kw[0] = cq == 0 ? rnd.x : cq == 1 ? rnd.y : cq == 2 ? rnd.z : rnd.w;
        } else {
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const int ws = (cq + k4) & 3;
            const uint32_t snd = ws == 0 ? rnd.x : ws == 1 ? rnd.y : ws == 2 ? rnd.z : rnd.w;
            const int src = (cq - k4) & 3;  // lane of the quad that drew for slot src % PB; it sends word[cq]
            const uint32_t rcv = (uint32_t)__shfl((int)snd, (lane & ~3) | src);
#pragma unroll
            for (int p = 0; p < PB; ++p)
              if ((src % PB) == p) kw[p] = rcv;
          }
        }
      }
#pragma unroll
      for (int p = 0; p < PB; ++p) {
        // (plain reads: the serial waves' writes are ordered by the workgroup barrier above, and volatile would force a
        // separate LDS round trip for every use)
        clds_t sfp = sf_all + p * PS;
        clds_t abp = ab_all + p * PS;
        const bool pv = mbase + p < M;
        double dd = 0.0;  // adjoint of dist_b (0 for idle threads and empty slots, so they add nothing below)
        if (act && pv) {
          double dist = 0.0;
          if constexpr (PFM > BW_MASK_FROM) {
            // wide classes: one uniform test per 8 features, reads at clamped indices and masked values inside (a test per feature put
            // every LDS read into its own basic block: 616 s_waitcnt for 600 ds_read in the 24 / 6 instantiation, a round trip per operand;
            // masked terms add exact zeros, so the sums are unchanged)
#pragma unroll
            for (int q0 = 0; q0 < PFM; q0 += 8) {
              if (q0 < PF) {
                double sv[8], cv[8], iv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                  const int q = imin(q0 + i, PF - 1);
                  sv[i] = sfp[q];
                  cv[i] = CENREG ? cen[CENREG ? imin(q0 + i, PFM - 1) : 0] : cen_l[q * NT + tid];  // (register centres: 0 beyond PF)
                  iv[i] = invl[q];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                  double rr = (sv[i] - cv[i]) * iv[i];
                  rr = q0 + i < PF ? rr : 0.0;
                  dist = fma(rr, rr, dist);
                }
              }
            }
          } else {
#pragma unroll
            for (int q = 0; q < PFM; ++q) {
              if (q < PF) {
                double rr = (sfp[q] - BW_CEN(q)) * invl[q];
                dist = fma(rr, rr, dist);
              }
            }
          }
          double phi = exp(-dist);
          double mk = 1.0;
          if (drop) {
            bool keep = a.nz.masks ? (a.nz.masks[((size_t)t * M + mbase + p) * B + b] != 0) : (kw[p] >= drop_thr);
            mk = keep ? keep_scale : 0.0;
          }
          double phibar = 0.0;
#pragma unroll
          for (int k = 0; k < UM; ++k) {
            if constexpr (PFM > BW_MASK_FROM) {  // (wide classes: masked instead of tested, as above)
              const double abk = k < U ? abp[imin(k, U - 1)] : 0.0;
              gw[k] = fma(abk, phi * mk, gw[k]);
              phibar = fma(k < U ? wgt[k] : 0.0, abk, phibar);
            } else if (k < U) {
              double abk = abp[k];
              gw[k] = fma(abk, phi * mk, gw[k]);
              phibar = fma(wgt[k], abk, phibar);
            }
          }
          dd = -phi * mk * phibar;
        }
        // 8 features at a time: their wave sums interleave (ILP) without keeping all PFM partial products live
#pragma unroll
        for (int q0 = 0; q0 < PFM; q0 += 8) {
          if (q0 < PF) {
            double t2v[8];  // 2 dd rr / l: the adjoint of the policy feature, before the sum over bases
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int q = q0 + i;
              double v = 0.0;
              if constexpr (PFM > BW_MASK_FROM) {
                if (q < PFM) {
                  const int qc = imin(q, PF - 1);
                  const double il = invl[qc];
                  const double rr = (sfp[qc] - (CENREG ? cen[CENREG ? q : 0] : cen_l[qc * NT + tid])) * il;
                  const double t2 = q < PF ? 2.0 * dd * rr : 0.0;
                  gc[q < PFM ? q : 0] = fma(-t2, il, gc[q < PFM ? q : 0]);
                  v = t2 * il;
                }
              } else if (q < PFM && q < PF) {
                double rr = (sfp[q] - BW_CEN(q)) * invl[q];
                double t2 = 2.0 * dd * rr;
                gc[q] = fma(-t2, invl[q], gc[q]);
                v = t2 * invl[q];
              }
              t2v[i] = v;
            }
            const double tot = wave_sum_pack8(t2v, lane);  // lane l: the wave's sum of feature q0 + (l & 7)
            if (lane < 8 && q0 + lane < PF) red[(p * NW + wv) * PF + q0 + lane] = tot;
          }
        }
      }
      BW_STAMP(10);
      if (t > 0) park(pre, cur ^ 1);
      cur ^= 1;
      lds_barrier();
    }
    // finish step 0: adjoint of x_0
    if (serial) {
      const vlds_t redp = red + sp * NW * PF;
      if (lane < PF) {
        double s = 0.0;
        for (int w = 0; w < NW; ++w) s += redp[w * PF + lane];
        sb[lane] = s;
        glacc = fma(-fprev, s, glacc);
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < S) {
        double s;
        if (pl.kind == MCP_POLICY_ANGLES) {
          s = (pi_plain >= 0) ? sb[pi_plain] : 0.0;
          if (pi_ang >= 0) s += -sb[pl.n_non_angle + pi_ang] * snm[lane] + sb[pl.n_non_angle + pl.n_angle + pi_ang] * csm[lane];
        } else if (pl.kind == MCP_POLICY_TRAJ) {
          s = sb[lane] - sb[S + lane];
        } else {
          s = sb[lane];
        }
        if (pms) s = meas_adjoint(s, 0);
        if (a.g_x0 && spvalid) a.g_x0[(size_t)msp * S + lane] = xb[lane] + s;
      }
    }
    lds_barrier();
  }

  // ---- write this workgroup's partial parameter gradients ------------------------------------
  const int nparam = PF + B * PF + U * B;  // (+ U when the policy has a bias: the slab stride)
  double* out = a.slab + (size_t)blockIdx.x * (nparam + (pl.bias ? U : 0));
  if (act) {
#pragma unroll
    for (int q = 0; q < PFM; ++q)
      if (q < PF) out[PF + (size_t)b * PF + q] = gc[q];
#pragma unroll
    for (int k = 0; k < UM; ++k)
      if (k < U) out[PF + (size_t)B * PF + (size_t)k * B + b] = gw[k];
  }
  lds_barrier();
  lds_t gla = (lds_t)(smem + L.gla);
#pragma unroll
  for (int q = 0; q < PFM; ++q) {
    if (q < PF) {
      double sm = wave_sum(act ? -BW_CEN(q) * gc[q] : 0.0);  // - sum_b c_bq dJ/dc_bq
      if (lane == 0) red[wv * PF + q] = sm;
    }
  }
  if (serial && lane < PF) gla[sp * PF + lane] = glacc;
  if (serial && lane < U) gla[PB * PF + sp * U + lane] = gbacc;
  lds_barrier();
  for (int it = tid; it < PF; it += NT) {
    double sm = 0.0;
    for (int w = 0; w < NW; ++w) sm += red[w * PF + it];
    for (int p = 0; p < PB; ++p) sm += gla[p * PF + it];
    out[it] = sm;
  }
  if (pl.bias && tid < U) {  // dJ/dbias: behind the three gradients in the slab
    double sm = 0.0;
    for (int p = 0; p < PB; ++p) sm += gla[PB * PF + p * U + tid];
    out[nparam + tid] = sm;
  }
}

// sum the per-workgroup slabs in a fixed order (deterministic, no atomics): 64 parameters per workgroup, its 4 waves take a
// quarter of the slabs each (4 independent partial sums per thread keep loads in flight), partials meet in LDS.  (One thread per
// parameter walking all slabs left 5 workgroups on the device: 34 us for 3.9 MB at the headline shape.)
__global__ __launch_bounds__(256) void grad_reduce_kernel(int nblk, int nparam, int PF, int BPF, int UB, const double* __restrict__ slab,
                                                          double* __restrict__ g_log_ls, double* __restrict__ g_centers,
                                                          double* __restrict__ g_weight, double* __restrict__ g_bias) {
  __shared__ double part[4][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const int per = (nblk + 3) / 4, k0 = q * per, k1 = imin(nblk, k0 + per);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (i < nparam) {
    int k = k0;
    for (; k + 3 < k1; k += 4) {
      s0 += slab[(size_t)k * nparam + i];
      s1 += slab[(size_t)(k + 1) * nparam + i];
      s2 += slab[(size_t)(k + 2) * nparam + i];
      s3 += slab[(size_t)(k + 3) * nparam + i];
    }
    for (; k < k1; ++k) s0 += slab[(size_t)k * nparam + i];
  }
  part[q][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q != 0 || i >= nparam) return;
  const double s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  if (i < PF)
    g_log_ls[i] = s;
  else if (i < PF + BPF)
    g_centers[i - PF] = s;
  else if (i < PF + BPF + UB)
    g_weight[i - PF - BPF] = s;
  else if (g_bias)
    g_bias[i - PF - BPF - UB] = s;
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
static unsigned long long* g_bwd_stamps = nullptr;  // diagnostic hook
extern "C" void mcp_debug_set_bwd_stamp_buffer(void* p) { g_bwd_stamps = (unsigned long long*)p; }
static int g_force_bwd_pb = 0;  // test hook: particles per workgroup of the backward sweep (0 = automatic)
extern "C" void mcp_debug_set_bwd_particles(int pb) { g_force_bwd_pb = pb; }
static int bwd_threads(int B) { return imax(64, ((B + 63) / 64) * 64); }
static int bwd_blocks(int M) { return imin(M, 1024); }

extern "C" size_t mcp_rollout_workspace_bytes(const mcp_model* model, const mcp_policy* policy, int M, int T) {
  if (!policy || M <= 0 || T <= 0) return 0;
  size_t nparam = (size_t)policy->P + (size_t)policy->B * policy->P + (size_t)policy->U * policy->B + (size_t)policy->U;  // (+ U: dJ/dbias)
  const size_t bwd = sizeof(double) * nparam * (size_t)bwd_blocks(M);   // mcp_rollout_bwd: per-workgroup gradient slabs
  const size_t fwd = model ? rollout_xch_bytes(M, model->G) + rollout_xj_bytes(model) + rollout_kt_bytes(model) : 0;  // mcp_rollout_fwd: hand-off granules (GP-sharded launch) + packed phase-J operand (wide classes)
  return bwd > fwd ? bwd : fwd;
}

template <int PFM, int UM, int MAXNT, int WPE, int PB>
static int launch_bwd(const BwdArgs& a, int NT, hipStream_t st) {
  if (NT > MAXNT || NT < 64 * PB) return MCP_ERR_LIMIT;
  const mcp_model& md = a.model;
  const bool pms = a.pol.meas.n > 0;
  if (PB * bwd_rec_len(md.S, md.U, md.D, md.G, pms) > BW_RPT * (NT / 64 > PB ? NT - 64 * PB : NT)) return MCP_ERR_LIMIT;
  const int grid = imin((a.M + PB - 1) / PB, 1024);
  BwdLayout L = bwd_layout(md.S, md.U, md.D, md.G, a.pol.P, NT / 64, PB, pms, PFM > 8);
  const size_t lds = sizeof(double) * (size_t)L.total;
  if (lds > MCP_LDS_LIMIT) return MCP_ERR_LIMIT;
  MCP_ENSURE_MAX_LDS(rollout_bwd_kernel<PFM, UM, MAXNT, WPE, PB>);
  hipLaunchKernelGGL((rollout_bwd_kernel<PFM, UM, MAXNT, WPE, PB>), dim3(grid), dim3(NT), lds, st, a);
  MCP_LAUNCH_CHECK();
  return grid;
}

extern "C" int mcp_rollout_bwd(const mcp_model* model, const mcp_policy* policy, const mcp_noise* noise, int M, int T,
                               const double* states, const double* inputs, const double* jac, const double* g_states,
                               const double* g_inputs, double* g_log_ls, double* g_centers, double* g_weight, double* g_x0,
                               void* workspace, size_t workspace_bytes, void* stream) {
  if (!noise || !states || !inputs || !g_log_ls || !g_centers || !g_weight || !workspace || !policy || M <= 0 || T <= 0) return MCP_ERR_ARG;
  if (T > 1 && !jac) return MCP_ERR_ARG;
  if (policy->meas.n > 0 && !policy->meas.meas) return MCP_ERR_ARG;
  mcp_model stub;
  if (!model) {
    if (T != 1) return MCP_ERR_ARG;
    stub = policy_only_model(policy);
    model = &stub;
  } else if (!model_ok(model)) {
    return MCP_ERR_ARG;
  }
  if (!policy_ok(policy, model->S, model->U, T)) return MCP_ERR_ARG;
  if (workspace_bytes < mcp_rollout_workspace_bytes(model, policy, M, T)) return MCP_ERR_WORKSPACE;
  BwdArgs a;
  a.model = *model;
  a.pol = *policy;
  a.nz = *noise;
  a.M = M;
  a.T = T;
  a.states = states;
  a.inputs = inputs;
  a.jac = jac;
  a.g_states = g_states;
  a.g_inputs = g_inputs;
  a.slab = (double*)workspace;
  a.g_x0 = g_x0;
  a.stamps = g_bwd_stamps;
  hipStream_t st = (hipStream_t)stream;
  const int PF = policy->P, U = policy->U;
  // particles per workgroup: large swarms are latency bound per workgroup, so several particles share one sweep; small
  // swarms keep one particle per workgroup to spread over the CUs
  // (two 256-thread workgroups per CU are resident: one particle per workgroup while M of them fit in one round, then 2, then 4;
  //  measured, tools/sweep_bwd_particles.py: M=800 1.74 / 1.34 / 1.92 ms, M=2000 3.24 / 2.51 / 2.07 ms for 1 / 2 / 4)
  int PB = g_force_bwd_pb ? g_force_bwd_pb : (M > 1024 ? 4 : (M > 512 ? 2 : 1));
  if (!g_force_bwd_pb && (PF > 16 || U > 4)) PB = M > 1024 ? 4 : 1;  // wide policies: four particles per sweep on large swarms where the
                                                                      // instantiation exists (> 256 basis functions), else two
                                                                      // (tools/time_bwd.py, UR5 shape, M = 2000, T = 300: 18.1 / 15.5 / 14.5 ms for 1 / 2 / 4)
  if (PB != 1 && PB != 2 && PB != 4) return MCP_ERR_ARG;
  int NT = imax(bwd_threads(policy->B), 64 * PB);
  int rc = MCP_ERR_LIMIT;
  // register budget: 3*PFM + 2*UM doubles of per-thread accumulators plus the prefetched record; the launch bound is the
  // tightest that fits the thread count, capped so that two 256-thread workgroups share a CU
  for (; PB >= 1 && rc == MCP_ERR_LIMIT; PB >>= 1) {
    if (PF <= 8 && U <= 2) {
      if (NT <= 256)
        rc = PB == 4 ? launch_bwd<8, 2, 256, BW_WPE_A, 4>(a, NT, st) : PB == 2 ? launch_bwd<8, 2, 256, BW_WPE_A, 2>(a, NT, st) : launch_bwd<8, 2, 256, BW_WPE_A, 1>(a, NT, st);
      else
        rc = PB == 1 ? launch_bwd<8, 2, 1024, 4, 1>(a, NT, st) : MCP_ERR_LIMIT;
    } else if (PF <= 16 && U <= 4) {
      if (NT <= 256)
        rc = PB == 2 ? launch_bwd<16, 4, 256, BW_WPE_A, 2>(a, NT, st) : PB == 1 ? launch_bwd<16, 4, 256, BW_WPE_A, 1>(a, NT, st) : MCP_ERR_LIMIT;
      else
        rc = PB == 1 ? launch_bwd<16, 4, 1024, 4, 1>(a, NT, st) : MCP_ERR_LIMIT;
    } else if (PF <= 24 && U <= 6) {  // UR5 class
      if (NT <= 256)
        rc = PB == 2 ? launch_bwd<24, 6, 256, BW_WPE_A, 2>(a, NT, st) : PB == 1 ? launch_bwd<24, 6, 256, BW_WPE_A, 1>(a, NT, st) : MCP_ERR_LIMIT;
      else
        rc = PB == 4 ? launch_bwd<24, 6, 512, 2, 4>(a, NT, st) : PB == 2 ? launch_bwd<24, 6, 512, 2, 2>(a, NT, st) : PB == 1 ? launch_bwd<24, 6, 512, 2, 1>(a, NT, st) : MCP_ERR_LIMIT;
    } else {
      if (NT <= 256)
        rc = PB == 2 ? launch_bwd<MCP_MAX_PFEAT, MCP_MAX_INPUT, 256, BW_WPE_A, 2>(a, NT, st)
                     : PB == 1 ? launch_bwd<MCP_MAX_PFEAT, MCP_MAX_INPUT, 256, BW_WPE_A, 1>(a, NT, st) : MCP_ERR_LIMIT;
      else
        rc = PB == 2 ? launch_bwd<MCP_MAX_PFEAT, MCP_MAX_INPUT, 512, 2, 2>(a, NT, st)
                     : PB == 1 ? launch_bwd<MCP_MAX_PFEAT, MCP_MAX_INPUT, 512, 2, 1>(a, NT, st) : MCP_ERR_LIMIT;
    }
    if (rc == MCP_ERR_LIMIT && PB > 1) NT = imax(bwd_threads(policy->B), 64 * (PB >> 1));
  }
  if (rc < 0) return rc;
  const int grid = rc;
  const int nparam = PF + policy->B * PF + U * policy->B + (policy->bias ? U : 0);
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((nparam + 63) / 64), dim3(256), 0, st, grid, nparam, PF, policy->B * PF, U * policy->B, a.slab,
                     g_log_ls, g_centers, g_weight, policy->bias ? policy->g_bias : nullptr);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

