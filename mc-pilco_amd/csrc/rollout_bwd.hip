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
//
// Two kernels:  rollout_bwd_kernel<PFM, UM, MAXNT, WPE, PB, PMS>  -- the general sweep (any class, 1 / 2 / 4 (wide class: 8) particles per workgroup, measurement
// models), and  rollout_bwd_lat_kernel<GM>  -- narrow plain / angle policies on swarms up to 3072 particles: one chain wave per particle working
// from registers beside RBF waves that prepare their step ahead of the barrier.
#include "rollout_common.h"
#include "../../include/mcpilco_hip_debug.h"
#include <type_traits>

using namespace mcp;
typedef double bw_v2d __attribute__((ext_vector_type(2)));

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
  int m_base, slab_accum;      // rollout_bwd_lat_kernel: first particle of this launch; its slabs add to what an earlier launch left
  int pipe;                    // rollout_bwd_kernel, one particle per workgroup, register chain: wave 0 runs the chain only and the launch has a wave more (see PIPEC)
};

#define BW_STAMP(k)                                 \
  do {                                              \
    if (a.stamps && tid == 0 && blockIdx.x == 0) {  \
      unsigned long long now_ = clock64();          \
      a.stamps[k] += now_ - last_stamp;             \
      last_stamp = now_;                            \
    }                                               \
  } while (0)

#define BW_RT 12    // doubles per lane in the role table of the register chain
#define BW_FL0 16   // its first policy-feature lane (states on lanes 0-15, features on 16-47, inputs on 48-55)
#define BW_UL0 48
struct BwdLayout {
  int invl, rec, xn, xb, zb, db, ab, sf, sb, sn, cs, snm, csm, cmv, cnv, nvx, red, gla, cen, itab, rtab, total;
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
  L.rtab = take(pms ? 0 : 64 * BW_RT);  // lane roles of the register chain (serial section without a measurement model)
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
// PMS: the policy sees a measurement model (mcp_meas); a template parameter so that the two forms of the serial section -- the register
// chain, and the LDS-staged chain with the measurement adjoint -- do not hold each other's lane-role registers live.
template <int PFM, int UM, int MAXNT, int WPE, int PB, bool PMS>
__global__ __launch_bounds__(MAXNT, WPE) void rollout_bwd_kernel(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const mcp_model& md = a.model;
  const mcp_policy& pl = a.pol;
  const mcp_noise nzl = noise_of_launch(a.nz);
  int tid = threadIdx.x, lane = tid & 63;  // (not const: laundered once per time step, see the sweep loop)
  const int NT = blockDim.x, NW = NT >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = md.S, U = md.U, G = md.G, D = md.D, B = pl.B, PF = pl.P, M = a.M, T = a.T;
  const mcp_meas& ms = pl.meas;
  constexpr bool pms = PMS;
  // the register chain serves the wide classes (C5 backward 9.2 -> 8.4 ms).  On the narrow class at 2-4 particles per workgroup the sweep is issue
  // bound with four 256-thread workgroups per CU and must stay within 128 registers for that: the chain's batch of operands then spills (24-40
  // VGPRs) and the gain measured 1.5 % (C3 backward 1.64 -> 1.62 ms) -- not kept; small swarms of the narrow class run rollout_bwd_lat_kernel.
  constexpr bool FASTCHAIN = !PMS && PFM > 8 && MAXNT <= 512;  // (the 1024-thread forms have 128 registers: the LDS-staged chain)
  // Round 6, one particle per workgroup on the wide classes (the UR5 launch script's M = 200): the sweep was chain (wave 0, 4.4 k cycles) -> barrier ->
  // RBF stage (5.2 k) -> park -> barrier per step.  With a wave more than the basis functions need (a.pipe) wave 0 owns NO basis function and the step
  // becomes  [chain of step t on wave 0  |  the RBF waves: exp / Philox / distances of step t, park the record of t - 1]  -> barrier ->
  // [the RBF waves: the adjoint half of step t  |  wave 0: sincos + policy features of step t - 1 from the parked record]  -> barrier:
  // what does not depend on the adjoint runs beside the chain, and the chain starts from features prepared a step earlier.
  constexpr bool PIPEC = PB == 1 && !PMS && PFM > 8 && MAXNT == 512;
  const bool pipe = PIPEC && a.pipe != 0;  // (uniform)
  constexpr bool CENREG = PFM <= 8 && MAXNT <= 512;  // narrow policies keep their centres in registers; wider ones -- and the 1024-thread forms with their 128 registers -- in LDS (transposed: conflict-free)
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
  const int b = (PIPEC && pipe) ? tid - 64 : tid;
  const bool act = b >= 0 && b < B;
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
  // The RBF stage works on SCALED quantities: centres and policy features already divided by the lengthscales (rr = s_q/l_q - c_bq/l_q: one
  // subtraction per feature and basis, no factor to fetch), centre gradients accumulated without their 1/l_q (applied once at the end) and the
  // feature adjoints summed over the bases before THEIR 1/l_q (applied once per feature by the serial wave): per (particle, basis, feature)
  // 4 vector instructions and 2 LDS operands where the unscaled form had 7 and 3 -- the stage is issue bound on large swarms.
  lds_t cen_l = (lds_t)(smem + L.cen);
  double cen[CENREG ? PFM : 1], gc[PFM], wgt[UM], gw[UM];
#pragma unroll
  for (int q = 0; q < PFM; ++q) {
    if (CENREG) cen[q] = (act && q < PF) ? pl.centers[(size_t)b * PF + q] * exp(-pl.log_ls[q < PF ? q : 0]) : 0.0;
    gc[q] = 0.0;
  }
  if (!CENREG)
    for (int q = 0; q < PF; ++q) cen_l[q * NT + tid] = act ? pl.centers[(size_t)b * PF + q] * exp(-pl.log_ls[q]) : 0.0;
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
  // ---- role table of the register chain (no measurement model) --------------------------------------------------------------
  // The serial section used to hand its intermediate vectors from one 10-lane stage to the next through LDS (six round trips per
  // step on the one wave everything waits for).  Without a measurement model it now runs from registers: lane roles -- lanes 0..S-1 a
  // state each, BW_FL0.. a policy feature each, BW_UL0.. an input each --; a lane reads the record entries and the partial feature
  // adjoints ITS role needs in one batch, the 2 G entries the integrator needs come by v_readlane, and the adjoints of x_t and of the
  // pre-squash activation are 2 G + 2 fused multiply-adds against the lane's own columns of d delta/dz (rollout_bwd_lat_kernel has the
  // derivation).  What a lane loads and gathers is fixed for the rollout: one 96-byte table row per lane, read back every step.
  double* rtab = smem + L.rtab;
  const bool any_trig = md.n_angle > 0 || (pl.kind == MCP_POLICY_ANGLES && pl.n_angle > 0);
  if (FASTCHAIN && wv == 0) {
    const bool st_l = lane < S, ft_l = lane >= BW_FL0 && lane < BW_FL0 + PF, in_l = lane >= BW_UL0 && lane < BW_UL0 + U;
    const int fq = lane - BW_FL0, uk = lane - BW_UL0;
    int src = 0, ftype = 0, i0 = 0, i1 = 0, i2 = 0, ia = -1, ib = -1, gvel = -1, flags = 0;
    double k0 = 0.0, k1c = 0.0, k2c = 0.0, ilf = 0.0, own = 0.0, rum = 1.0;
    const int pn = pl.n_non_angle, pa = pl.n_angle;
    if (st_l) {
      src = lane;
      flags |= 1;
      if (pl.kind == MCP_POLICY_ANGLES) {
        for (int i = 0; i < pn; ++i)
          if (pl.non_angle[i] == lane) { i0 = i; k0 = 1.0; }
        for (int i = 0; i < pa; ++i)
          if (pl.angle[i] == lane) { i1 = pn + i; i2 = pn + pa + i; k1c = 1.0; k2c = 1.0; flags |= 16; }
      } else if (pl.kind == MCP_POLICY_TRAJ) {
        i0 = lane;
        k0 = 1.0;
        i1 = S + lane;
        k1c = -1.0;
      } else {
        i0 = lane;
        k0 = 1.0;
      }
      for (int i = 0; i < nna_g; ++i)
        if (md.not_angle[i] == lane) ia = i;
      for (int i = 0; i < na_g; ++i)
        if (md.angle[i] == lane) { ia = nna_g + i; ib = nna_g + na_g + i; flags |= 8; }
      bool isv = false, isp = false;
      for (int g = 0; g < G; ++g) {
        if (md.vel[g] == lane) { isv = true; gvel = g; }
        if (md.not_vel[g] == lane) isp = true;
      }
      own = (isv ? 1.0 : 0.0) + (isp ? 1.0 : 0.0);
    } else if (ft_l) {
      flags |= 2;
      i0 = fq;
      k0 = 1.0;
      src = fq;
      if (pl.kind == MCP_POLICY_ANGLES) {
        for (int i = 0; i < pn; ++i)
          if (i == fq) src = pl.non_angle[i];
        for (int i = 0; i < pa; ++i) {
          if (pn + i == fq) { src = pl.angle[i]; ftype = 1; }
          if (pn + pa + i == fq) { src = pl.angle[i]; ftype = 2; }
        }
      } else if (pl.kind == MCP_POLICY_TRAJ && fq >= S) {
        src = fq - S;
        ftype = 3;
      }
    } else if (in_l) {
      flags |= 4;
      ia = nna_g + 2 * na_g + uk;
      for (int k = 0; k < U; ++k)
        if (k == uk) rum = 1.0 / pl.u_max[k];
    }
    if (ia >= 0) flags |= 32;
    // 1 / l_q of the feature columns the lane gathers (the RBF stage sums l_q x adjoint) and of the feature it publishes
    auto ilq = [&](int q) { return exp(-pl.log_ls[imin(q, PF - 1)]); };
    k0 *= ilq(i0);
    k1c *= ilq(i1);
    k2c *= ilq(i2);
    if (ft_l) ilf = ilq(fq);
    int* ri = reinterpret_cast<int*>(rtab + lane * BW_RT);
    ri[0] = src; ri[1] = imax(ia, 0); ri[2] = imax(ib, 0); ri[3] = i0;
    ri[4] = i1; ri[5] = i2; ri[6] = flags | (ftype << 8); ri[7] = gvel;
    double* rd = rtab + lane * BW_RT + 4;
    rd[0] = k0; rd[1] = k1c; rd[2] = k2c; rd[3] = ilf; rd[4] = own; rd[5] = rum;
  }
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
  double tgt_c = 0.0;                       // register chain, trajectory policies: the target entry of the step to come
  double xbr = 0.0, kp1 = 0.0, kp2 = 0.0;  // register chain: adjoint of x_t without the policy path; feature-map coefficients of step t+1
  unsigned long long last_stamp = clock64();
#ifdef BWX_WSTAMPS  // experiment build: every wave's own intervals of a step (workgroup 0): chain + prefetch | barrier 1 | RBF stage | park | barrier 2
  unsigned long long ws_[5] = {0, 0, 0, 0, 0}, wt_ = clock64();
#define BW_WS(k)                                              \
  do {                                                        \
    if (a.stamps && blockIdx.x == 0) {                        \
      const unsigned long long n_ = clock64();                \
      ws_[k] += n_ - wt_;                                     \
      wt_ = n_;                                               \
    }                                                         \
  } while (0)
#else
#define BW_WS(k)
#endif
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
    xbr = 0.0;
    kp1 = kp2 = 0.0;
    if (FASTCHAIN && serial && pl.kind == MCP_POLICY_TRAJ) {
      const int* ri = reinterpret_cast<const int*>(rtab + lane * BW_RT);
      tgt_c = (ri[6] >> 8) == 3 ? pl.target_traj[(size_t)(T - 1) * S + ri[0]] : 0.0;
    }
    lds_barrier();
    // pipelined form: what the chain of a step takes from the step's own record alone -- prepared here for T - 1, then a step ahead
    double p_sv = 0.0, p_cv = 1.0, p_fn = 0.0;
    auto chain_prep = [&](int tt, int buf) {
      const double* rrow = rtab + lane * BW_RT;
      const int4 ra = *reinterpret_cast<const int4*>(rrow), rb = *reinterpret_cast<const int4*>(rrow + 2);
      const bw_v2d rk2f = *reinterpret_cast<const bw_v2d*>(rrow + 6);
      const int fl = rb.z & 255, ftype = rb.z >> 8;
      const bool ft_l = fl & 2;
      clds_t r = (clds_t)(smem + L.rec) + buf * NRP + sp * NR;
      const double xs = r[oX + ra.x];
      const double tgt = tgt_c;  // (trajectory policies: loaded a step ahead)
      if (pl.kind == MCP_POLICY_TRAJ && ftype == 3 && tt > 0) tgt_c = pl.target_traj[(size_t)(tt - 1) * S + ra.x];
      double sv = 0.0, cv = 1.0;
      if (any_trig) sincos_fast(xs, &sv, &cv);
      const double fn = ftype == 0 ? xs : (ftype == 1 ? cv : (ftype == 2 ? sv : tgt - xs));
      if (ft_l) sf[lane - BW_FL0] = fn * rk2f.y;
      p_sv = sv;
      p_cv = cv;
      p_fn = fn;
    };
    if (PIPEC && pipe) {
      if (serial) {
        chain_prep(T - 1, cur);
        for (int q = lane; q < PF; q += 64) red[(sp * NW + wv) * PF + q] = 0.0;  // (wave 0 has no basis function: its row of partial feature adjoints stays zero)
      }
      lds_barrier();
    }
    for (int t = T - 1; t >= 0; --t) {
      // thread / lane ids are laundered per step: what the unrolled feature loops derive from them (LDS addresses, predicates) is
      // recomputed where it is used instead of being hoisted out of the sweep, kept live next to the accumulators and spilled
      if (BW_LAUNDER(PFM, PB) || MAXNT > 512) {  // (the 1024-thread forms: 128 registers)
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        pf_tid = pf_split ? tid - 64 * PB : tid;
      }
      const int b = (PIPEC && pipe) ? tid - 64 : tid;
      BW_STAMP(11);
      // the next step's record: issued at the top by the waves that have no chain to run, BEHIND the chain by the serial waves (the loads
      // have the whole RBF stage to land; their registers and ~50 address instructions stay off the chain)
      if (t > 0 && !(FASTCHAIN && serial)) prefetch(pre, t - 1, mbase);
      // ---- serial section: wave p for particle slot p -----------------------------------------------
      if (serial) {
        if (PIPEC && pipe) {
          // ---- the register chain, pipelined form: sincos / features of this step are in p_sv, p_cv, p_fn (chain_prep, a step earlier) ----
          const double* rrow = rtab + lane * BW_RT;
          const int4 ra = *reinterpret_cast<const int4*>(rrow), rb = *reinterpret_cast<const int4*>(rrow + 2);
          const bw_v2d rk01 = *reinterpret_cast<const bw_v2d*>(rrow + 4), rk2f = *reinterpret_cast<const bw_v2d*>(rrow + 6),
                       row_ = *reinterpret_cast<const bw_v2d*>(rrow + 8);
          const int fl = rb.z & 255;
          const bool st_l = fl & 1, in_l = fl & 4, zang = fl & 8, pang = fl & 16, has_j = fl & 32;
          const int uk = in_l ? lane - BW_UL0 : 0;
          clds_t r = (clds_t)(smem + L.rec) + cur * NRP + sp * NR;
          const bool last = (t == T - 1);
          const double gb = r[st_l ? oGX + lane : oGU + uk];
          const double uu = r[oU + uk];
          constexpr int GMX = MCP_MAX_GP;
          double Ja[GMX], Jb[GMX];
#pragma unroll
          for (int g = 0; g < GMX; ++g) {
            if (g < G) {
              Ja[g] = r[oJ + g * D + ra.y];
              Jb[g] = r[oJ + g * D + ra.z];
            } else {
              Ja[g] = Jb[g] = 0.0;
            }
          }
          double c0 = 0.0, c1 = 0.0, c2 = 0.0;
          if (!last) {
            const vlds_t redp = red + sp * NW * PF;
            for (int w0 = 0; w0 < NW; w0 += 2) {  // (two waves' partials per pass: six reads in flight)
              const int wa = w0, wb = imin(w0 + 1, NW - 1);
              const double a0 = ((clds_t)redp)[wa * PF + ra.w], a1 = ((clds_t)redp)[wa * PF + rb.x], a2 = ((clds_t)redp)[wa * PF + rb.y];
              const double b0 = ((clds_t)redp)[wb * PF + ra.w], b1 = ((clds_t)redp)[wb * PF + rb.x], b2 = ((clds_t)redp)[wb * PF + rb.y];
              const bool okb = w0 + 1 < NW;
              c0 += a0;
              c1 += a1;
              c2 += a2;
              c0 += okb ? b0 : 0.0;
              c1 += okb ? b1 : 0.0;
              c2 += okb ? b2 : 0.0;
            }
          }
          const double sv = p_sv, cv = p_cv, fn = p_fn;
          const double s = last ? 0.0 : fma(kp2, c2, fma(kp1, c1, rk01.x * c0));
          if (!last) glacc = fma(-fprev, s, glacc);  // feature lanes: - f_q(t+1) * (adjoint of f_q(t+1))
          const double xnr = xbr + s;                 // state lanes: adjoint of x_{t+1}
          double val = fma(row_.x, xnr, gb);
#pragma unroll
          for (int g = 0; g < GMX; ++g) {
            if (g < G) {
              const int lv = md.vel[g], lp = md.not_vel[g];
              const double xnv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xnr), lv), __builtin_amdgcn_readlane(__double2loint(xnr), lv));
              const double xnp = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xnr), lp), __builtin_amdgcn_readlane(__double2loint(xnr), lp));
              const double dbg = fma(0.5 * md.Ts, xnp, xnv);
              const double jc = zang ? fma(Ja[g], cv, -(Jb[g] * sv)) : Ja[g];
              val = fma(g == rb.w ? md.Ts : 0.0, xnp, val);
              val = fma(dbg, has_j ? jc : 0.0, val);
            }
          }
          xbr = val;
          const double th = uu * row_.y;
          const double abv = (in_l && pl.squash) ? val * (1.0 - th * th) : val;
          if (in_l) {
            ab[uk] = abv;
            if (spvalid) gbacc += abv;
          }
          kp1 = pang ? -sv * rk01.y : rk01.y;
          kp2 = pang ? cv * rk2f.x : 0.0;
          fprev = fn;
        } else if constexpr (FASTCHAIN) {
          // ---- the register chain (see the role table above) ----
          const double* rrow = rtab + lane * BW_RT;
          const int4 ra = *reinterpret_cast<const int4*>(rrow), rb = *reinterpret_cast<const int4*>(rrow + 2);
          const bw_v2d rk01 = *reinterpret_cast<const bw_v2d*>(rrow + 4), rk2f = *reinterpret_cast<const bw_v2d*>(rrow + 6),
                       row_ = *reinterpret_cast<const bw_v2d*>(rrow + 8);
          const int fl = rb.z & 255, ftype = rb.z >> 8;
          const bool st_l = fl & 1, ft_l = fl & 2, in_l = fl & 4, zang = fl & 8, pang = fl & 16, has_j = fl & 32;
          const int uk = in_l ? lane - BW_UL0 : 0;
          clds_t r = (clds_t)(smem + L.rec) + cur * NRP + sp * NR;
          const bool last = (t == T - 1);
          // one batch of LDS reads: the record entries of this lane's role, its columns of d delta/dz, the partial feature adjoints
          const double xs = r[oX + ra.x];
          const double gb = r[st_l ? oGX + lane : oGU + uk];
          const double uu = r[oU + uk];
          constexpr int GMX = MCP_MAX_GP;
          double Ja[GMX], Jb[GMX];
#pragma unroll
          for (int g = 0; g < GMX; ++g) {
            if (g < G) {
              Ja[g] = r[oJ + g * D + ra.y];
              Jb[g] = r[oJ + g * D + ra.z];
            } else {
              Ja[g] = Jb[g] = 0.0;
            }
          }
          double c0 = 0.0, c1 = 0.0, c2 = 0.0;
          if (!last) {
            const vlds_t redp = red + sp * NW * PF;
            for (int w0 = 0; w0 < NW; w0 += 2) {  // (two waves' partials per pass: six reads in flight)
              const int wa = w0, wb = imin(w0 + 1, NW - 1);
              const double a0 = ((clds_t)redp)[wa * PF + ra.w], a1 = ((clds_t)redp)[wa * PF + rb.x], a2 = ((clds_t)redp)[wa * PF + rb.y];
              const double b0 = ((clds_t)redp)[wb * PF + ra.w], b1 = ((clds_t)redp)[wb * PF + rb.x], b2 = ((clds_t)redp)[wb * PF + rb.y];
              const bool okb = w0 + 1 < NW;
              c0 += a0;
              c1 += a1;
              c2 += a2;
              c0 += okb ? b0 : 0.0;
              c1 += okb ? b1 : 0.0;
              c2 += okb ? b2 : 0.0;
            }
          }
          const double tgt = tgt_c;  // (trajectory policies: loaded a step ahead)
          if (pl.kind == MCP_POLICY_TRAJ && ftype == 3 && t > 0) tgt_c = pl.target_traj[(size_t)(t - 1) * S + ra.x];
          double sv = 0.0, cv = 1.0;
          if (any_trig) sincos_fast(xs, &sv, &cv);
          const double fn = ftype == 0 ? xs : (ftype == 1 ? cv : (ftype == 2 ? sv : tgt - xs));
          if (ft_l) sf[lane - BW_FL0] = fn * rk2f.y;  // (scaled: what the RBF stage subtracts the scaled centres from)
          const double s = last ? 0.0 : fma(kp2, c2, fma(kp1, c1, rk01.x * c0));
          if (!last) glacc = fma(-fprev, s, glacc);  // feature lanes: - f_q(t+1) * (adjoint of f_q(t+1))
          const double xnr = xbr + s;                 // state lanes: adjoint of x_{t+1}
          double val = fma(row_.x, xnr, gb);
#pragma unroll
          for (int g = 0; g < GMX; ++g) {
            if (g < G) {
              const int lv = md.vel[g], lp = md.not_vel[g];
              const double xnv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xnr), lv), __builtin_amdgcn_readlane(__double2loint(xnr), lv));
              const double xnp = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xnr), lp), __builtin_amdgcn_readlane(__double2loint(xnr), lp));
              const double dbg = fma(0.5 * md.Ts, xnp, xnv);
              const double jc = zang ? fma(Ja[g], cv, -(Jb[g] * sv)) : Ja[g];
              val = fma(g == rb.w ? md.Ts : 0.0, xnp, val);
              val = fma(dbg, has_j ? jc : 0.0, val);
            }
          }
          xbr = val;
          const double th = uu * row_.y;
          const double abv = (in_l && pl.squash) ? val * (1.0 - th * th) : val;
          if (in_l) {
            ab[uk] = abv;
            if (spvalid) gbacc += abv;
          }
          kp1 = pang ? -sv * rk01.y : rk01.y;
          kp2 = pang ? cv * rk2f.x : 0.0;
          fprev = fn;
        } else {
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
              s *= invl[lane];  // (the RBF stage sums the feature adjoints without their 1 / l_q)
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
            sincos_fast(r[oX + lane], &sv, &cv);
            sn[lane] = sv;
            cs[lane] = cv;
          }
          if (pms && lane < S && pi_ang >= 0) {
            double sv, cv;
            sincos_fast(r[oM + lane], &sv, &cv);
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
            sf[lane] = f * invl[lane];  // (scaled: what the RBF stage subtracts the scaled centres from)
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
      }
      if (FASTCHAIN && t > 0 && serial) prefetch(pre, t - 1, mbase);
      if (!(PIPEC && pipe)) {  // (pipelined form: the barrier comes behind the RBF stage's first half, below)
        BW_STAMP(8);
        BW_WS(0);
        lds_barrier();
        BW_STAMP(9);
        BW_WS(1);
      }
      // ---- RBF network, thread b owns basis b, loops over the particle slots -----------------------------
      // dropout keep bits: one Philox draw serves 4 consecutive bases of one particle (philox_keep); the lanes of a quad
      // draw for particle slots (b & 3) % PB and pass each other the word of the receiver's basis
      uint32_t kw[PB];
#pragma unroll
      for (int p = 0; p < PB; ++p) kw[p] = 0xFFFFFFFFu;
      if (drop && !nzl.masks && !(PIPEC && pipe && serial)) {  // (pipelined form: wave 0 has no basis function to draw for)
        const int cq = b & 3;
        const int bq = imin(b, B - 1) >> 2;
        // (PB = 8: two rounds, slots cq and cq + 4)
        constexpr int PQ = PB < 4 ? PB : 4;  // slots served by one round of the quad
#pragma unroll
        for (int rd = 0; rd < (PB + 3) / 4; ++rd) {
          const u32x4 rnd = philox_draw(nzl, imin(mbase + 4 * rd + (cq % PQ), M - 1), t, MCP_STREAM_MASK, (uint32_t)bq);
          if (PB == 1) {
            kw[0] = cq == 0 ? rnd.x : cq == 1 ? rnd.y : cq == 2 ? rnd.z : rnd.w;
          } else {
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
              const int ws = (cq + k4) & 3;
              const uint32_t snd = ws == 0 ? rnd.x : ws == 1 ? rnd.y : ws == 2 ? rnd.z : rnd.w;
              const int src = (cq - k4) & 3;  // lane of the quad that drew for slot 4 rd + src % PQ; it sends word[cq]
              const uint32_t rcv = quad_from_back(snd, k4);
#pragma unroll
              for (int p = 0; p < PQ; ++p)
                if ((src % PQ) == p) kw[4 * rd + p] = rcv;
            }
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
        constexpr bool KEEPRR = PFM > BW_MASK_FROM && MAXNT <= 512;  // (the 1024-thread forms have 128 registers: they read twice)
        static_assert(!PIPEC || KEEPRR, "the pipelined form rewrites the features while the adjoint half runs: it needs the kept differences");
        double rrk[KEEPRR ? PFM : 1];  // wide classes: the scaled differences, kept for the adjoint pass (no second LDS pass)
        if constexpr (KEEPRR) {
#pragma unroll
          for (int q = 0; q < PFM; ++q) rrk[q] = 0.0;
        }
        double phi = 0.0, mk = 1.0;
        if (act && pv) {
          double dist = 0.0;
          if constexpr (PFM > BW_MASK_FROM) {
            // wide classes: one uniform test per 8 features, reads at clamped indices and masked values inside (a test per feature put
            // every LDS read into its own basic block: 616 s_waitcnt for 600 ds_read in the 24 / 6 instantiation, a round trip per operand;
            // masked terms add exact zeros, so the sums are unchanged)
#pragma unroll
            for (int q0 = 0; q0 < PFM; q0 += 8) {
              if (q0 < PF) {
                double sv[8], cv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                  const int q = imin(q0 + i, PF - 1);
                  sv[i] = sfp[q];
                  cv[i] = CENREG ? cen[CENREG ? imin(q0 + i, PFM - 1) : 0] : cen_l[q * NT + tid];  // (register centres: 0 beyond PF)
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                  double rr = sv[i] - cv[i];
                  rr = q0 + i < PF ? rr : 0.0;
                  if constexpr (KEEPRR) rrk[KEEPRR ? q0 + i : 0] = rr;
                  dist = fma(rr, rr, dist);
                }
              }
            }
          } else {
#pragma unroll
            for (int q = 0; q < PFM; ++q) {
              if (q < PF) {
                double rr = sfp[q] - BW_CEN(q);
                dist = fma(rr, rr, dist);
              }
            }
          }
          phi = exp(-dist);
          if (drop) {
            bool keep = nzl.masks ? (nzl.masks[((size_t)t * M + mbase + p) * B + b] != 0) : (kw[p] >= drop_thr);
            mk = keep ? keep_scale : 0.0;
          }
        }
        if (PIPEC && pipe) {
          // ---- pipelined form: everything above ran beside the chain; the record of step t - 1 has had that time to land ----
          if (t > 0) park(pre, cur ^ 1);
          BW_STAMP(8);
          BW_WS(0);
          lds_barrier();
          BW_STAMP(9);
          BW_WS(1);
        }
        if (act && pv) {
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
        const double dd2 = 2.0 * dd;
        // 8 features at a time: their wave sums interleave (ILP) without keeping all PFM partial products live
#pragma unroll
        for (int q0 = 0; q0 < PFM; q0 += 8) {
          if (q0 < PF && !(PIPEC && pipe && serial)) {  // (pipelined form: wave 0 has no basis function -- its row of `red` was zeroed once)
            double t2v[8];  // 2 dd rr / l: the adjoint of the policy feature, before the sum over bases
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int q = q0 + i;
              double v = 0.0;
              if constexpr (PFM > BW_MASK_FROM) {
                if (q < PFM) {
                  double t2;
                  if constexpr (KEEPRR) {
                    t2 = dd2 * rrk[KEEPRR ? q : 0];  // (rrk is zero beyond PF and for idle threads)
                  } else {
                    const int qc = imin(q, PF - 1);
                    const double rr = sfp[qc] - (CENREG ? cen[CENREG ? q : 0] : cen_l[qc * NT + tid]);
                    t2 = q < PF ? dd2 * rr : 0.0;
                  }
                  gc[q < PFM ? q : 0] -= t2;
                  v = t2;
                }
              } else if (q < PFM && q < PF) {
                double rr = sfp[q] - BW_CEN(q);
                double t2 = dd2 * rr;
                gc[q] -= t2;
                v = t2;
              }
              t2v[i] = v;
            }
            const double tot = wave_sum_pack8(t2v, lane);  // lane l: the wave's sum of feature q0 + (l & 7)
            if (lane < 8 && q0 + lane < PF) red[(p * NW + wv) * PF + q0 + lane] = tot;
          }
        }
      }
      BW_STAMP(10);
      BW_WS(2);
      if (PIPEC && pipe) {
        if (serial && t > 0) chain_prep(t - 1, cur ^ 1);  // (beside the RBF waves' adjoint half: the record was parked before the barrier)
      } else if (t > 0) {
        park(pre, cur ^ 1);
      }
      cur ^= 1;
      BW_WS(3);
      lds_barrier();
      BW_WS(4);
    }
    // finish step 0: adjoint of x_0
    if constexpr (FASTCHAIN) {
     if (serial) {
      const double* rrow = rtab + lane * BW_RT;
      const int4 ra = *reinterpret_cast<const int4*>(rrow), rb = *reinterpret_cast<const int4*>(rrow + 2);
      const double k0 = rrow[4];
      clds_t redp = (clds_t)(red + sp * NW * PF);
      double c0 = 0.0, c1 = 0.0, c2 = 0.0;
      for (int w = 0; w < NW; ++w) {
        c0 += redp[w * PF + ra.w];
        c1 += redp[w * PF + rb.x];
        c2 += redp[w * PF + rb.y];
      }
      const double s = fma(kp2, c2, fma(kp1, c1, k0 * c0));
      glacc = fma(-fprev, s, glacc);
      if ((rb.z & 1) && a.g_x0 && spvalid) a.g_x0[(size_t)msp * S + lane] = xbr + s;
     }
    } else if (serial) {
      const vlds_t redp = red + sp * NW * PF;
      if (lane < PF) {
        double s = 0.0;
        for (int w = 0; w < NW; ++w) s += redp[w * PF + lane];
        s *= invl[lane];
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

#ifdef BWX_WSTAMPS
  if (a.stamps && blockIdx.x == 0 && lane == 0)
    for (int k = 0; k < 5; ++k) a.stamps[16 + wv * 5 + k] = ws_[k];
#endif
  // ---- write this workgroup's partial parameter gradients ------------------------------------
  const int nparam = PF + B * PF + U * B;  // (+ U when the policy has a bias: the slab stride)
  double* out = a.slab + (size_t)blockIdx.x * (nparam + (pl.bias ? U : 0));
  if (act) {
#pragma unroll
    for (int q = 0; q < PFM; ++q)
      if (q < PF) out[PF + (size_t)b * PF + q] = gc[q] * invl[q];  // (accumulated without the 1 / l_q)
#pragma unroll
    for (int k = 0; k < UM; ++k)
      if (k < U) out[PF + (size_t)B * PF + (size_t)k * B + b] = gw[k];
  }
  lds_barrier();
  lds_t gla = (lds_t)(smem + L.gla);
#pragma unroll
  for (int q = 0; q < PFM; ++q) {
    if (q < PF) {
      double sm = wave_sum(act ? -BW_CEN(q) * gc[q] : 0.0);  // - sum_b c_bq dJ/dc_bq  (scaled centre x unscaled gradient: the 1 / l_q cancel)
      if (lane == 0) red[wv * PF + q] = sm;
    }
  }
  if (FASTCHAIN) {  // (register chain: feature q sits on lane BW_FL0 + q, input k on lane BW_UL0 + k)
    if (serial && lane >= BW_FL0 && lane < BW_FL0 + PF) gla[sp * PF + lane - BW_FL0] = glacc;
    if (serial && lane >= BW_UL0 && lane < BW_UL0 + U) gla[PB * PF + sp * U + lane - BW_UL0] = gbacc;
  } else {
    if (serial && lane < PF) gla[sp * PF + lane] = glacc;
    if (serial && lane < U) gla[PB * PF + sp * U + lane] = gbacc;
  }
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

// ---------------------------------------------------------------------------------------
// Latency-lean backward sweep: small swarms (one particle per workgroup), narrow class
// ---------------------------------------------------------------------------------------
// At M <= 512 every workgroup of the sweep above is resident and walks its particle's T steps alone: the launch takes T times the
// latency of ONE step, and a step of the general kernel is a chain of ~6 LDS round trips between 10-lane stages (4.7 k cycles) followed
// by the whole RBF stage (exp, Philox, distances: 3.2 k) behind a barrier.  What actually depends on the adjoint of step t + 1 is
// small; this kernel takes everything else off the chain:
//   * wave 0 owns the chain and nothing else (the basis functions live in waves 1 .. NWB).  Lane roles: lanes 0..S-1 a state each,
//     lanes 8..8+PF-1 a policy feature each, lanes 16..16+U-1 an input each.  Per step it reads the NWB partial feature adjoints of the
//     columns ITS role needs straight from the RBF waves' LDS rows (one batch of reads, one latency), forms the adjoint of x_{t+1} in
//     registers, fetches the 2 G entries the integrator needs with v_readlane (uniform indices), and finishes x_t's adjoint and the
//     pre-squash adjoint with 2 G + 2 fused multiply-adds against operands it prepared a step earlier:
//       xb_s  = gx_s + own_s xn_s + sum_g [s = vel_g] Ts xn_{pos_g} + sum_g db_g Jx[g][s],   Jx = dz/dx_s folded into d delta_g/dz
//       ab_k  = (gu_k + sum_g db_g J[g][u_k]) (1 - tanh^2)
//     (the general kernel forms zb = J^T db first and maps it through the feature map afterwards: three more LDS round trips).
//   * between the two barriers of a step (while the other waves run the RBF stage) wave 0 prepares step t - 1: sincos of the angles,
//     Jx, the policy features of x_{t-1} for the RBF waves, tanh'; its record elements come straight from global memory into the lanes
//     that use them, two steps ahead (two register stages, the time loop unrolled by two).
//   * the RBF waves compute exp(-dist), the dropout bit and the scaled distances of step t BEFORE the barrier that publishes ab_t
//     (all of it depends on x_t only), so that behind it only the multiply-adds and the 8-value wave sum remain.
// Same gradients as the general kernel up to summation order (tests/test_gpu_parity.py compares them on every policy kind it covers).
#define BL_GM 4    // GPs
#ifndef BL_MAX_M
#define BL_MAX_M 3072  // largest swarm the lean sweep takes (tools/sweep_bwd_particles.py on a 256-CU device: it wins up to ~3000 particles)
#endif
#define BL_FL0 8   // first feature lane of wave 0
#define BL_UL0 16  // first input lane of wave 0
// (scalar members only: with arrays inside, part of the struct stayed a stack object -- two of its prefetched values went through
//  scratch with the wait for the load right behind it)
template <int GM>
struct BlStage {
  double xs, xm, gb, uu, ja0, jb0, ja1, jb1, ja2, jb2, ja3, jb3;  // (xm: the measured state, loaded only with a measurement model)
};
template <int g, int GM>
__device__ __forceinline__ double& bl_ja(BlStage<GM>& s) {
  if constexpr (g == 0) return s.ja0;
  else if constexpr (g == 1) return s.ja1;
  else if constexpr (g == 2) return s.ja2;
  else return s.ja3;
}
template <int g, int GM>
__device__ __forceinline__ double& bl_jb(BlStage<GM>& s) {
  if constexpr (g == 0) return s.jb0;
  else if constexpr (g == 1) return s.jb1;
  else if constexpr (g == 2) return s.jb2;
  else return s.jb3;
}
template <int g, int GM, typename F>
__device__ __forceinline__ void bl_for_g(F&& f) {
  if constexpr (g < GM) {
    f(std::integral_constant<int, g>());
    bl_for_g<g + 1, GM>(f);
  }
}
#ifndef BL_SW
#define BL_SW 1  // the RBF wave whose intervals are stamped (diagnostic builds: build.py --variant)
#endif
#define BL_STAMP(k)                                          \
  do {                                                       \
    if (a.stamps && lane == 0 && blockIdx.x == 0 && slot == 0) { \
      unsigned long long now_ = clock64();                   \
      s_stamp[k] += now_ - last_stamp; /* (LDS: a global read-modify-write would wait for the prefetch in flight) */ \
      last_stamp = now_;                                     \
    }                                                        \
  } while (0)

// PMS (round 4): the measurement model of MC_PILCO4PMS.apply_policy (mcp_meas).  The policy features of a step are functions of the MEASURED
// state, so what the chain gathers is dJ/d(measured x_{t+1}); the state lanes map it to dJ/dx_{t+1} through the adjoint of the measurement
// (general kernel: meas_adjoint) in registers: a velocity lane carries mvb (adjoint of the filtered velocity), a position lane nvb of its pair
// (adjoint of the finite-difference velocity), the pair's value of the step crosses lanes by v_readlane (uniform pair indices).
template <int GM, bool PMS>
__global__ __launch_bounds__(640) __attribute__((amdgpu_waves_per_eu(3, 3))) void rollout_bwd_lat_kernel(BwdArgs a) {
  constexpr int PFM = 8, UM = 2;
  __shared__ double s_red_[2][4][8];  // [slot][RBF wave][feature]: the wave's sum over its basis functions of the feature adjoint
  __shared__ double s_sf_[2][PFM];    // policy features of the step the RBF waves prepare
  __shared__ double s_ab_[2][UM];     // adjoint of the pre-squash activation
  __shared__ double s_invl[PFM];
  __shared__ double s_fin_[2][PFM + UM];
  __shared__ double s_cen[PFM][256];  // RBF centres, [feature][basis]: read before the barrier only, so they need not sit in registers
  __shared__ int s_role[64][8];       // wave 0: what a lane loads (read in the prefetch, off the chain)
  __shared__ unsigned long long s_stamp[16];
  const mcp_model& md = a.model;
  const mcp_policy& pl = a.pol;
  const mcp_noise nzl = noise_of_launch(a.nz);
  // two particle slots per workgroup (each: one chain wave + the RBF waves behind it), so that M <= 512 particles are ONE resident round:
  // a 5-wave workgroup needs 3 waves on one SIMD, two of them do not share a CU, and 400 single-slot workgroups ran as two rounds
  const int WPS = (int)(blockDim.x >> 7);  // waves per slot
  const int slot = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) >= WPS ? 1 : 0);
  const int tid = threadIdx.x - slot * WPS * 64, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  double (*s_red)[8] = s_red_[slot];
  double* s_sf = s_sf_[slot];
  double* s_ab = s_ab_[slot];
  double* s_fin = s_fin_[slot];
  const int S = md.S, U = md.U, G = md.G, D = md.D, B = pl.B, PF = pl.P, M = a.M, T = a.T;
  const int nna = md.n_not_angle, na = md.n_angle, pn = pl.n_non_angle, pa = pl.n_angle;
  const bool drop = pl.p_drop > 0.0;
  const double keep_scale = 1.0 / (1.0 - pl.p_drop);
  const uint32_t drop_thr = drop_threshold(pl.p_drop);
  unsigned long long last_stamp = clock64();
  const unsigned long long t_begin = last_stamp, w_begin = a.stamps ? wall_clock64() : 0;

  if (tid < PFM) {
    if (slot == 0) s_invl[tid] = tid < PF ? exp(-pl.log_ls[tid]) : 0.0;
    s_sf[tid] = 0.0;
  }
  if (tid < 32) (&s_red[0][0])[tid] = 0.0;  // rows of absent waves and columns beyond PF stay zero: the chain reads them unconditionally
  if (tid < UM) s_ab[tid] = 0.0;
  if (threadIdx.x < 16) s_stamp[threadIdx.x] = 0;
  __syncthreads();

  // ---- RBF threads: basis b = tid - 64 --------------------------------------------------------------------------------------
  const int b = tid - 64;
  // (swarms beyond 512 particles: one launch per 512, a.m_base = its first particle -- a loop over the rounds inside the kernel cost the
  //  one-round form its register allocation: the chain's LDS reads serialised, 0.19 -> 0.21 ms at M = 400)
  const int mg = a.m_base + (int)blockIdx.x * 2 + slot;
  const int m = imin(mg, M - 1);
  const bool valid = mg < M;  // (an odd swarm's last slot idles through the barriers)
  const bool act = valid && wv > 0 && b < B;
  const int bs = wv > 0 ? b : 0;  // column of s_cen
  double gc[PFM], wgt[UM], gw[UM];
#pragma unroll
  for (int q = 0; q < PFM; ++q) {
    if (wv > 0 && slot == 0) s_cen[q][bs] = (b < B && q < PF) ? pl.centers[(size_t)b * PF + q] * exp(-pl.log_ls[q]) : 0.0;  // (scaled, like the features)
    gc[q] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < UM; ++k) {
    wgt[k] = (act && k < U) ? pl.weight[(size_t)k * B + b] : 0.0;
    gw[k] = 0.0;
  }

  // ---- wave 0: lane roles (uniform reads of the descriptors, compared with the lane: no per-lane indexing of kernel arguments) ----
  const bool st_lane = lane < S, ft_lane = lane >= BL_FL0 && lane < BL_FL0 + PF, in_lane = lane >= BL_UL0 && lane < BL_UL0 + U;
  const int fq = lane - BL_FL0, uk = lane - BL_UL0;
  int src = 0, ftype = 0;            // state whose value the lane loads; what it makes of it: 0 x, 1 cos x, 2 sin x
  int i0 = 0, i1 = 0, i2 = 0;        // feature columns whose adjoints the lane gathers
  double k0 = 0.0;
  bool pang = false, zang = false;
  int ia = -1, ib = -1;              // columns of d delta_g/dz the lane needs
  double own = 0.0, umax = 1.0;
  int gvel = -1;
  if (wv == 0) {
    if (st_lane) {
      src = lane;
      if (pl.kind == MCP_POLICY_ANGLES) {
        for (int i = 0; i < pn; ++i)
          if (pl.non_angle[i] == lane) { i0 = i; k0 = 1.0; }
        for (int i = 0; i < pa; ++i)
          if (pl.angle[i] == lane) { i1 = pn + i; i2 = pn + pa + i; pang = true; }
      } else {
        i0 = lane;
        k0 = 1.0;
      }
      for (int i = 0; i < nna; ++i)
        if (md.not_angle[i] == lane) ia = i;
      for (int i = 0; i < na; ++i)
        if (md.angle[i] == lane) { ia = nna + i; ib = nna + na + i; zang = true; }
      bool isv = false, isp = false;
      for (int g = 0; g < G; ++g) {
        if (md.vel[g] == lane) { isv = true; gvel = g; }
        if (md.not_vel[g] == lane) isp = true;
      }
      own = (isv ? 1.0 : 0.0) + (isp ? 1.0 : 0.0);
    } else if (ft_lane) {
      i0 = fq;
      k0 = 1.0;
      src = fq;
      if (pl.kind == MCP_POLICY_ANGLES) {
        for (int i = 0; i < pn; ++i)
          if (i == fq) src = pl.non_angle[i];
        for (int i = 0; i < pa; ++i) {
          if (pn + i == fq) { src = pl.angle[i]; ftype = 1; }
          if (pn + pa + i == fq) { src = pl.angle[i]; ftype = 2; }
        }
      }
    } else if (in_lane) {
      ia = nna + 2 * na + uk;
      for (int k = 0; k < U; ++k)
        if (k == uk) umax = pl.u_max[k];
    }
  }
  const bool need_trig = na > 0 || (pl.kind == MCP_POLICY_ANGLES && pa > 0);
  const bool has_j = ia >= 0;
  bool pm_vel = false, pm_pos = false;  // this state lane is the velocity / position of a measurement pair
  if (PMS && wv == 0 && st_lane)
    for (int i = 0; i < pl.meas.n; ++i) {
      if (pl.meas.vel[i] == lane) pm_vel = true;
      if (pl.meas.pos[i] == lane) pm_pos = true;
    }
  const double pm_a = PMS ? -pl.meas.a1 / pl.meas.a0 : 0.0, pm_b0 = PMS ? pl.meas.b0 / pl.meas.a0 : 0.0, pm_b1 = PMS ? pl.meas.b1 / pl.meas.a0 : 0.0;
  const double pm_its = PMS ? 1.0 / md.Ts : 0.0;
  double pm_cm = 0.0, pm_cn = 0.0;  // carried: mvb_{t+1} (velocity lanes), nvb_{t+1} of the pair (position lanes)
  // 1 / l_q of the feature columns this lane gathers (the RBF waves sum l_q x adjoint) and of the feature it publishes
  const double il1 = s_invl[i1], il2 = s_invl[i2], ilf = ft_lane ? s_invl[fq] : 0.0;
  k0 *= s_invl[i0];
  const double rumax = 1.0 / umax;
  if (wv == 0 && slot == 0) {
    s_role[lane][0] = src;
    s_role[lane][1] = ia;
    s_role[lane][2] = ib;
    s_role[lane][3] = ftype;
  }
  __syncthreads();

  double glacc = 0.0, gbacc = 0.0;
  {
    if (wv == 0) {
      // ================= the chain =================
      // (no selects on the loaded values here: a select is a use, and the wait it needs would turn the prefetch into a round trip --
      //  every lane loads from a valid address, what a lane has no use for is masked where it is consumed, in prep)
      // Loads run tt = T-1, T-2, ... one call after the other: uniform running element offsets (scalar registers, one subtraction per array
      // and call) plus a constant 32-bit offset per lane -- the address arithmetic of the first version (64-bit products per load) was
      // half of this wave's work between the barriers.
      const int4 ro = *reinterpret_cast<const int4*>(&s_role[lane][0]);  // src, ia, ib, ftype
      const unsigned lx = (unsigned)ro.x, lja = (unsigned)imax(ro.y, 0), ljb = (unsigned)imax(ro.z, 0);
      size_t oS = ((size_t)(T - 1) * M + m) * S, oU = ((size_t)(T - 1) * M + m) * U, oJ = ((size_t)(T - 1) * M + m) * G * D;
      const size_t dS = (size_t)M * S, dU = (size_t)M * U, dJ = (size_t)M * G * D;
      int tl = T - 1;  // the step the next call loads
      auto load = [&](BlStage<GM>& st) {
        st.xs = a.states[oS + lx];
        if (PMS) st.xm = pl.meas.meas[oS + lx];
        double gv = 0.0, uv = 0.0;
        if (a.g_states && st_lane) gv = a.g_states[oS + (unsigned)lane];
        if (in_lane) {
          if (a.g_inputs) gv = a.g_inputs[oU + (unsigned)uk];
          uv = a.inputs[oU + (unsigned)uk];
        }
        st.gb = gv;
        st.uu = uv;
        if (tl < T - 1) {
          bl_for_g<0, GM>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const unsigned gg = (unsigned)(imin(g, G - 1) * D);
            bl_ja<g>(st) = a.jac[oJ + (gg + lja)];
            bl_jb<g>(st) = a.jac[oJ + (gg + ljb)];
          });
        } else {
          bl_for_g<0, GM>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            bl_ja<g>(st) = 0.0;
            bl_jb<g>(st) = 0.0;
          });
        }
        --tl;
        oS -= dS;
        oU -= dU;
        oJ -= dJ;
      };
      double Jc[GM], gbase = 0.0, sq = 1.0, k1n = 0.0, k2n = 0.0, fn = 0.0;  // operands of the step the chain runs next
      double k1p = 0.0, k2p = 0.0, fprev = 0.0;                               // feature map / feature of the step after it
      auto prep = [&](BlStage<GM>& st) {
        double sv = 0.0, cv = 1.0;
        if (need_trig) sincos_fast(st.xs, &sv, &cv);
        // the policy's side of the step (feature values, feature-map coefficients): on the measured state
        double svm = sv, cvm = cv;
        const double xpol = PMS ? st.xm : st.xs;
        if (PMS && need_trig) sincos_fast(st.xm, &svm, &cvm);
        bl_for_g<0, GM>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
          const double ja = bl_ja<g>(st), jb = bl_jb<g>(st);
          const double jc = zang ? fma(ja, cv, -(jb * sv)) : ja;
          Jc[g] = (g < G && has_j) ? jc : 0.0;
        });
        gbase = st.gb;
        const double th = st.uu * rumax;
        sq = (in_lane && pl.squash) ? 1.0 - th * th : 1.0;
        k1n = pang ? -svm * il1 : 0.0;  // (with the 1 / l of the feature columns they weigh)
        k2n = pang ? cvm * il2 : 0.0;
        fn = ro.w == 0 ? xpol : (ro.w == 1 ? cvm : svm);
        if (ft_lane) s_sf[fq] = fn * ilf;
      };
      double xb = 0.0;
      auto gather = [&]() -> double {  // the feature adjoints of the step just finished by the RBF waves, mapped to this lane's role
        // all twelve reads in flight before the first add (the pin: with fewer registers at hand the allocator reused one destination
        // for all of them and the chain waited for six LDS round trips one after the other -- seen in the ISA, 0.7 k -> 1.15 k cycles)
        double r0[4], r1[4], r2[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          r0[w] = s_red[w][i0];
          r1[w] = s_red[w][i1];
          r2[w] = s_red[w][i2];
        }
        asm volatile("" ::"v"(r0[0]), "v"(r0[1]), "v"(r0[2]), "v"(r0[3]), "v"(r1[0]), "v"(r1[1]), "v"(r1[2]), "v"(r1[3]), "v"(r2[0]), "v"(r2[1]),
                     "v"(r2[2]), "v"(r2[3]));
        const double c0 = ((r0[0] + r0[1]) + r0[2]) + r0[3];
        const double c1 = ((r1[0] + r1[1]) + r1[2]) + r1[3];
        const double c2 = ((r2[0] + r2[1]) + r2[2]) + r2[3];
        return fma(k2p, c2, fma(k1p, c1, k0 * c0));
      };
      // adjoint of the measurement model at step tt (tt = 0: `first`) -- MC_PILCO.py:881-899, see meas_adjoint of the general kernel:
      //   mvb_tt = s_vel - a1/a0 mvb_{tt+1};  nvb_tt = b0/a0 mvb_tt + b1/a0 mvb_{tt+1} (0 at tt = 0);  x_tt[pos] gets s_pos + (nvb_tt - nvb_{tt+1})/Ts,
      //   x_tt[vel] nothing -- except at tt = 0, where the measurement is the true state: x_0[vel] gets mvb_0 + b1/a0 mvb_1
      auto meas_adj = [&](double s_in, bool first) -> double {
        const double mvb = pm_vel ? fma(pm_a, pm_cm, s_in) : 0.0;
        const double nvb = (pm_vel && !first) ? fma(pm_b0, mvb, pm_b1 * pm_cm) : 0.0;
        double sx = pm_vel ? (first ? fma(pm_b1, pm_cm, mvb) : 0.0) : s_in;
        double nvp = 0.0;  // nvb_tt of this position lane's pair
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < pl.meas.n) {  // uniform
            const int lv = pl.meas.vel[i];
            const double nvi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(nvb), lv), __builtin_amdgcn_readlane(__double2loint(nvb), lv));
            if (lane == pl.meas.pos[i]) nvp = nvi;
          }
        }
        if (pm_pos) {
          sx = fma(nvp - pm_cn, pm_its, s_in);
          pm_cn = nvp;
        }
        if (pm_vel) pm_cm = mvb;
        return sx;
      };
      auto chain = [&](bool last) {
        double s = 0.0;
        if (!last) {
          s = gather();
          glacc = fma(-fprev, s, glacc);  // feature lanes: - f_q(t+1) * (adjoint of f_q(t+1))
        }
        if (PMS && !last) s = meas_adj(s, false);  // dJ/d(measured x_{t+1}) -> its part of dJ/dx_{t+1}
        const double xn = xb + s;  // state lanes: adjoint of x_{t+1}
        double val = fma(own, xn, gbase);
#pragma unroll
        for (int g = 0; g < GM; ++g) {
          if (g < G) {
            const int lv = md.vel[g], lp = md.not_vel[g];
            const double xnv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xn), lv), __builtin_amdgcn_readlane(__double2loint(xn), lv));
            const double xnp = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xn), lp), __builtin_amdgcn_readlane(__double2loint(xn), lp));
            const double dbg = fma(0.5 * md.Ts, xnp, xnv);
            val = fma(g == gvel ? md.Ts : 0.0, xnp, val);
            val = fma(dbg, Jc[g], val);
          }
        }
        xb = val;
        const double abv = val * sq;
        if (in_lane) {
          s_ab[uk] = abv;
          if (valid) gbacc += abv;
        }
      };
      BlStage<GM> stA, stB;
      load(stA);  // T - 1
      prep(stA);
      if (T >= 2) load(stA);  // T - 2
      if (T >= 3) load(stB);  // T - 3
      lds_barrier();
      BL_STAMP(11);
      auto step = [&](int t, BlStage<GM>& sx) {  // sx holds the record of step t - 1
        chain(t == T - 1);
        BL_STAMP(8);
        lds_barrier();
        BL_STAMP(9);
        k1p = k1n;
        k2p = k2n;
        fprev = fn;
        if (t > 0) {
          prep(sx);
          if (t >= 3) load(sx);  // t - 3
        }
        BL_STAMP(10);
        lds_barrier();
        BL_STAMP(11);
      };
      int t = T - 1;
      for (; t >= 1; t -= 2) {
        step(t, stA);
        step(t - 1, stB);
      }
      if (t == 0) step(0, stA);
      // adjoint of x_0
      {
        double s = gather();
        glacc = fma(-fprev, s, glacc);
        if (PMS) s = meas_adj(s, true);
        if (a.g_x0 && st_lane && valid) a.g_x0[(size_t)m * S + lane] = xb + s;
      }
    } else {
      // ================= the RBF waves =================
      lds_barrier();
      uint32_t kwo[4] = {0u, 0u, 0u, 0u};
      for (int t = T - 1; t >= 0; --t) {
        // before the barrier: everything of step t that depends on x_t only
        // dropout keep word of (t, basis b): word b & 3 of the Philox block of (t, b >> 2).  The four lanes of a quad would all compute the
        // same block: instead, every fourth step lane j of the quad draws the block of step t - j and the lanes pass each other the word
        // of the receiver's basis (quad_perm rotations) -- a quarter of the Philox work per step.
        uint32_t kwd = 0xFFFFFFFFu;
        uint8_t mk8 = 1;
        if (drop) {
          if (nzl.masks) {
            mk8 = nzl.masks[((size_t)t * M + m) * B + imin(b, B - 1)];
          } else {
            const int ph = (T - 1 - t) & 3;
            if (ph == 0) {
              const int cq = b & 3;
              const u32x4 rnd = philox_draw(nzl, m, imax(t - cq, 0), MCP_STREAM_MASK, (uint32_t)(imin(b, B - 1) >> 2));
              uint32_t rc[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) {  // rotation r: lane i receives from lane (i + r) & 3 that lane's word i
                const int wi = (cq - r) & 3;
                const uint32_t snd = wi == 0 ? rnd.x : wi == 1 ? rnd.y : wi == 2 ? rnd.z : rnd.w;
                rc[r] = r == 0 ? snd
                               : (uint32_t)(r == 1 ? __builtin_amdgcn_update_dpp(0, (int)snd, 0x39, 0xf, 0xf, false)
                                                   : r == 2 ? __builtin_amdgcn_update_dpp(0, (int)snd, 0x4E, 0xf, 0xf, false)
                                                            : __builtin_amdgcn_update_dpp(0, (int)snd, 0x93, 0xf, 0xf, false));
              }
#pragma unroll
              for (int o = 0; o < 4; ++o) {  // the word of step t - o came with rotation (o - cq) & 3
                const int r = (o - cq) & 3;
                kwo[o] = r == 0 ? rc[0] : r == 1 ? rc[1] : r == 2 ? rc[2] : rc[3];
              }
            }
            kwd = ph == 0 ? kwo[0] : ph == 1 ? kwo[1] : ph == 2 ? kwo[2] : kwo[3];
          }
        }
        double e[PFM], dist = 0.0;  // e_q = rr_q = s_q / l_q - c_bq / l_q (features and centres are stored scaled)
#pragma unroll
        for (int q = 0; q < PFM; ++q) {
          e[q] = s_sf[q] - s_cen[q][bs];
          dist = fma(e[q], e[q], dist);
        }
        const double phi = exp(-dist);
        double mk = 1.0;
        if (drop) mk = (nzl.masks ? (mk8 != 0) : (kwd >= drop_thr)) ? keep_scale : 0.0;
        const double pm = act ? phi * mk : 0.0;
        if (wv == BL_SW) BL_STAMP(12);
        lds_barrier();
        if (wv == BL_SW) BL_STAMP(13);
        // behind it: what depends on the adjoint
        double phibar = 0.0;
#pragma unroll
        for (int k = 0; k < UM; ++k) {
          const double abk = s_ab[k];
          gw[k] = fma(abk, pm, gw[k]);
          phibar = fma(wgt[k], abk, phibar);
        }
        const double dd = -2.0 * pm * phibar;
        double t2v[8];
#pragma unroll
        for (int q = 0; q < PFM; ++q) {
          t2v[q] = dd * e[q];  // l_q x the adjoint of feature q through basis b (the 1 / l_q: once per feature, in the chain's gather)
          gc[q] -= t2v[q];
        }
        const double tot = wave_sum_pack8(t2v, lane);
        if (lane < 8) s_red[wv - 1][lane] = tot;
        if (wv == BL_SW) BL_STAMP(14);
        lds_barrier();
        if (wv == BL_SW) BL_STAMP(15);
      }
    }
  }

  // ---- this workgroup's partial parameter gradients (the slab layout of the general kernel) -----------------------------------
  const int nparam = PF + B * PF + U * B;
  double* out = a.slab + (size_t)(mg & 1023) * (nparam + (pl.bias ? U : 0));  // one slab per particle (modulo the 1024 the workspace holds)
  const bool accum = a.slab_accum != 0;  // (this launch's particles share their slabs with an earlier launch's: add)
  if (act) {
#pragma unroll
    for (int q = 0; q < PFM; ++q)
      if (q < PF) {
        double* o = out + PF + (size_t)b * PF + q;
        const double v = gc[q] * s_invl[q];  // (accumulated without the 1 / l_q)
        *o = accum ? *o + v : v;
      }
#pragma unroll
    for (int k = 0; k < UM; ++k)
      if (k < U) {
        double* o = out + PF + (size_t)B * PF + (size_t)k * B + b;
        *o = accum ? *o + gw[k] : gw[k];
      }
  }
  __syncthreads();  // (the chain's last reads of s_red are done)
  if (wv > 0) {
    double cg[8];
#pragma unroll
    for (int q = 0; q < PFM; ++q) cg[q] = act ? -s_cen[q][bs] * gc[q] : 0.0;  // - sum_b c_bq dJ/dc_bq
    const double tot = wave_sum_pack8(cg, lane);
    if (lane < 8) s_red[wv - 1][lane] = tot;
  } else {
    if (ft_lane) s_fin[fq] = glacc;
    if (in_lane) s_fin[PFM + uk] = gbacc;
  }
  __syncthreads();
  if (valid && tid < PF) {
    const double v = (((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid]) + s_fin[tid];
    out[tid] = accum ? out[tid] + v : v;
  }
  if (valid && pl.bias && tid < U) out[nparam + tid] = accum ? out[nparam + tid] + s_fin[PFM + tid] : s_fin[PFM + tid];
  if (a.stamps && blockIdx.x == 0 && threadIdx.x >= 8 && threadIdx.x < 16) a.stamps[threadIdx.x] = s_stamp[threadIdx.x];
  if (a.stamps && tid == 0) {  // diagnostic: spread of the workgroups' run times (core cycles) and start / end times (100 MHz wall clock)
    atomicMax(&a.stamps[0], clock64() - t_begin);
    atomicMin(&a.stamps[1], clock64() - t_begin);
    atomicMin(&a.stamps[2], w_begin);
    atomicMax(&a.stamps[3], w_begin);
    atomicMax(&a.stamps[4], wall_clock64());
  }
}

// sum the per-workgroup slabs in a fixed order (deterministic, no atomics): 64 parameters per workgroup, its 4 waves take a
// quarter of the slabs each (4 independent partial sums per thread keep loads in flight), partials meet in LDS.  (One thread per
// parameter walking all slabs left 5 workgroups on the device: 34 us for 3.9 MB at the headline shape.)
// (round 6: sixteen waves per 64 parameters instead of four -- 200 slabs are 13 loads in a row per thread, not 50: 10.3 -> 4 us at the headline shape)
#define GR_NW 16
__global__ __launch_bounds__(64 * GR_NW) void grad_reduce_kernel(int nblk, int nparam, int PF, int BPF, int UB, const double* __restrict__ slab,
                                                                 double* __restrict__ g_log_ls, double* __restrict__ g_centers,
                                                                 double* __restrict__ g_weight, double* __restrict__ g_bias) {
  __shared__ double part[GR_NW][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const int per = (nblk + GR_NW - 1) / GR_NW, k0 = imin(nblk, q * per), k1 = imin(nblk, k0 + per);
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
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < GR_NW; w += 4) s += (part[w][lane] + part[w + 1][lane]) + (part[w + 2][lane] + part[w + 3][lane]);  // (fixed order)
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
// what rollout_bwd_lat_kernel covers: the narrow class with its lane roles (states on lanes 0-7, features on 8-15, inputs on 16-17),
// up to 256 basis functions, plain / angle policies on the true or (measurement model) the measured state, disjoint index lists
static bool bwd_lean_applies(const mcp_model* md, const mcp_policy* pl, int T) {
  if (T < 2 || md->G < 1 || md->G > BL_GM || md->S > 8 || pl->P > 8 || pl->U > 2 || pl->B > 256 || md->U != pl->U) return false;
  if (pl->kind != MCP_POLICY_PLAIN && pl->kind != MCP_POLICY_ANGLES) return false;
  if (pl->meas.n > 4) return false;  // (measurement pairs cross lanes by uniform-index v_readlane: up to 4 pairs on 8 state lanes)
  for (int i = 0; i < pl->meas.n; ++i)
    for (int j = 0; j < pl->meas.n; ++j)
      if ((i != j && (pl->meas.pos[i] == pl->meas.pos[j] || pl->meas.vel[i] == pl->meas.vel[j])) || pl->meas.pos[i] == pl->meas.vel[j]) return false;
  if (pl->kind == MCP_POLICY_PLAIN && pl->P != md->S) return false;
  if (md->n_not_angle + 2 * md->n_angle + md->U != md->D) return false;
  for (int i = 0; i < md->n_angle; ++i)
    for (int j = 0; j < md->n_not_angle; ++j)
      if (md->angle[i] == md->not_angle[j]) return false;
  if (pl->kind == MCP_POLICY_ANGLES) {
    if (pl->n_non_angle + 2 * pl->n_angle != pl->P) return false;
    for (int i = 0; i < pl->n_angle; ++i)
      for (int j = 0; j < pl->n_non_angle; ++j)
        if (pl->angle[i] == pl->non_angle[j]) return false;
  }
  for (int g = 0; g < md->G; ++g)
    for (int h = 0; h < g; ++h)
      if (md->vel[g] == md->vel[h]) return false;
  return true;
}
static int bwd_threads(int B) { return imax(64, ((B + 63) / 64) * 64); }
static int bwd_blocks(int M) { return imin(M, 1024); }

extern "C" size_t mcp_rollout_workspace_bytes(const mcp_model* model, const mcp_policy* policy, int M, int T) {
  if (!policy || M <= 0 || T <= 0) return 0;
  size_t nparam = (size_t)policy->P + (size_t)policy->B * policy->P + (size_t)policy->U * policy->B + (size_t)policy->U;  // (+ U: dJ/dbias)
  const size_t bwd = sizeof(double) * nparam * (size_t)bwd_blocks(M);   // mcp_rollout_bwd: per-workgroup gradient slabs
  const size_t fwd = model ? rollout_xch_bytes(M, model->G) + rollout_xj_bytes(model) + rollout_kt_bytes(model) + rollout_uxch_bytes(M, model->G, model->U) + rollout_rxch_bytes(model, M) : 0;  // mcp_rollout_fwd: hand-off granules (GP-sharded launch) + packed phase-J operand (wide classes)
  return bwd > fwd ? bwd : fwd;
}

template <int PFM, int UM, int MAXNT, int WPE, int PB, bool PMS>
static int launch_bwd_pms(const BwdArgs& a, int grid, int NT, size_t lds, hipStream_t st) {
  MCP_ENSURE_MAX_LDS(rollout_bwd_kernel<PFM, UM, MAXNT, WPE, PB, PMS>);
  hipLaunchKernelGGL((rollout_bwd_kernel<PFM, UM, MAXNT, WPE, PB, PMS>), dim3(grid), dim3(NT), lds, st, a);
  MCP_LAUNCH_CHECK();
  return grid;
}
template <int PFM, int UM, int MAXNT, int WPE, int PB>
static int launch_bwd(const BwdArgs& a, int NT, hipStream_t st) {
  if (NT > MAXNT || NT < 64 * PB) return MCP_ERR_LIMIT;
  const mcp_model& md = a.model;
  const bool pms = a.pol.meas.n > 0;
  if (PB * bwd_rec_len(md.S, md.U, md.D, md.G, pms) > BW_RPT * (NT / 64 > PB ? NT - 64 * PB : NT)) return MCP_ERR_LIMIT;
  const int grid = imin((a.M + PB - 1) / PB, 1024);
  BwdLayout L = bwd_layout(md.S, md.U, md.D, md.G, a.pol.P, NT / 64, PB, pms, PFM > 8 || MAXNT > 512);
  const size_t lds = sizeof(double) * (size_t)L.total;
  if (lds > MCP_LDS_LIMIT) return MCP_ERR_LIMIT;
  return pms ? launch_bwd_pms<PFM, UM, MAXNT, WPE, PB, true>(a, grid, NT, lds, st) : launch_bwd_pms<PFM, UM, MAXNT, WPE, PB, false>(a, grid, NT, lds, st);
}

static int rollout_bwd_impl(const mcp_model* model, const mcp_policy* policy, const mcp_noise* noise, int M, int T,
                               const double* states, const double* inputs, const double* jac, const double* g_states,
                               const double* g_inputs, double* g_log_ls, double* g_centers, double* g_weight, double* g_x0,
                               void* workspace, size_t workspace_bytes, void* stream, unsigned long long* g_bwd_stamps, int g_force_bwd_pb, int g_bwd_lean, int& g_last_bwd_lean,
                               int g_bwd_pipe = -1, int* g_last_bwd_pipe = nullptr) {
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
  a.m_base = 0;
  a.slab_accum = 0;
  a.pipe = 0;
  hipStream_t st = (hipStream_t)stream;
  const int PF = policy->P, U = policy->U;
  // particles per workgroup: large swarms are latency bound per workgroup, so several particles share one sweep; small
  // swarms keep one particle per workgroup to spread over the CUs
  // (two 256-thread workgroups per CU are resident: one particle per workgroup while M of them fit in one round, then 2, then 4;
  //  measured, tools/sweep_bwd_particles.py: M=800 1.74 / 1.34 / 1.92 ms, M=2000 3.24 / 2.51 / 2.07 ms for 1 / 2 / 4)
  int PB = g_force_bwd_pb ? g_force_bwd_pb : (M > 2816 ? 4 : (M > 512 ? 2 : 1));  // (round 3, after the RBF stage's diet: 2 particles win up to ~2800, tools/sweep_bwd_particles.py)
  if (!g_force_bwd_pb && (PF > 16 || U > 4)) {
    PB = M > 1024 ? 4 : 1;
    // eight per sweep (round 4) where that saves resident rounds: a 512-thread workgroup of this class has a CU to itself (256 per round), and a
    // step of eight particles costs 1.86 x a step of four (tools/phase_stamps.py c5: 64.7 k vs 34.7 k cycles -- the RBF stage is per particle) --
    // M = 2000: one round instead of two, 8.74 -> 8.1 ms; M = 3072: two instead of three, slower (16.8 vs 13.6 ms)
    const int r4 = (((M + 3) / 4) + 255) / 256, r8 = (((M + 7) / 8) + 255) / 256;
    if (PB == 4 && bwd_threads(policy->B) > 256 && 1.86 * r8 < (double)r4) PB = 8;  // (the 512-thread instantiation: > 256 basis functions)
  }  // wide policies: eight (round 4; four before) particles per sweep on large swarms where the
                                                                      // instantiation exists (> 256 basis functions), else two
                                                                      // (tools/time_bwd.py, UR5 shape, M = 2000, T = 300: 18.1 / 15.5 / 14.5 ms for 1 / 2 / 4)
  if (PB != 1 && PB != 2 && PB != 4 && PB != 8) return MCP_ERR_ARG;
  int NT = imax(bwd_threads(policy->B), 64 * PB);
  int rc = MCP_ERR_LIMIT;
  g_last_bwd_lean = 0;
  if (g_bwd_lean != 0 && !g_force_bwd_pb && M <= BL_MAX_M && model != &stub && bwd_lean_applies(model, policy, T)) {
    // small swarm, narrow class: the latency-lean sweep (wave 0 = the chain, the basis functions in the waves behind it)
    const int nt = 2 * (64 + bwd_threads(policy->B));  // two particle slots per workgroup
    for (int mb = 0; mb < M; mb += 512) {           // one launch per 512 particles = 256 workgroups: a resident round each
      const int grid = (imin(M - mb, 512) + 1) / 2;
      a.m_base = mb;
      a.slab_accum = mb >= 1024;
      if (policy->meas.n > 0) {
        if (model->G <= 2)
          hipLaunchKernelGGL((rollout_bwd_lat_kernel<2, true>), dim3(grid), dim3(nt), 0, st, a);
        else
          hipLaunchKernelGGL((rollout_bwd_lat_kernel<BL_GM, true>), dim3(grid), dim3(nt), 0, st, a);
      } else if (model->G <= 2) {
        hipLaunchKernelGGL((rollout_bwd_lat_kernel<2, false>), dim3(grid), dim3(nt), 0, st, a);
      } else {
        hipLaunchKernelGGL((rollout_bwd_lat_kernel<BL_GM, false>), dim3(grid), dim3(nt), 0, st, a);
      }
      MCP_LAUNCH_CHECK();
    }
    g_last_bwd_lean = 1;
    rc = imin(M, 1024);  // slabs: one per particle, modulo the 1024 the workspace holds
    PB = 0;
  }
  // register budget: 3*PFM + 2*UM doubles of per-thread accumulators plus the prefetched record; the launch bound is the
  // tightest that fits the thread count, capped so that two 256-thread workgroups share a CU
  for (; PB >= 1 && rc == MCP_ERR_LIMIT; PB >>= 1) {
    // one particle per workgroup on the 512-thread wide instantiations, no measurement model, a wave to spare: the pipelined form (PIPEC)
    a.pipe = (PB == 1 && g_bwd_pipe != 0 && !(PF <= 16 && U <= 4) && NT > 256 && NT + 64 <= 512 && policy->meas.n == 0) ? 1 : 0;
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
        rc = PB == 8   ? launch_bwd<24, 6, 512, 2, 8>(a, NT, st)
             : PB == 4 ? launch_bwd<24, 6, 512, 2, 4>(a, NT, st)
             : PB == 2 ? launch_bwd<24, 6, 512, 2, 2>(a, NT, st)
             : PB == 1 ? launch_bwd<24, 6, 512, 2, 1>(a, NT + 64 * a.pipe, st)
                       : MCP_ERR_LIMIT;
    } else {
      if (NT <= 256)
        rc = PB == 2 ? launch_bwd<MCP_MAX_PFEAT, MCP_MAX_INPUT, 256, BW_WPE_A, 2>(a, NT, st)
                     : PB == 1 ? launch_bwd<MCP_MAX_PFEAT, MCP_MAX_INPUT, 256, BW_WPE_A, 1>(a, NT, st) : MCP_ERR_LIMIT;
      else
        rc = PB == 2 ? launch_bwd<MCP_MAX_PFEAT, MCP_MAX_INPUT, 512, 2, 2>(a, NT, st)
                     : PB == 1 ? launch_bwd<MCP_MAX_PFEAT, MCP_MAX_INPUT, 512, 2, 1>(a, NT + 64 * a.pipe, st) : MCP_ERR_LIMIT;
    }
    if (rc == MCP_ERR_LIMIT && PB > 1) NT = imax(bwd_threads(policy->B), 64 * (PB >> 1));
  }
  if (rc < 0) return rc;
  if (g_last_bwd_pipe) *g_last_bwd_pipe = a.pipe;  // (set by the launch that went out; 0 when the lean sweep ran)
  const int grid = rc;
  const int nparam = PF + policy->B * PF + U * policy->B + (policy->bias ? U : 0);
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((nparam + 63) / 64), dim3(64 * GR_NW), 0, st, grid, nparam, PF, policy->B * PF, U * policy->B, a.slab,
                     g_log_ls, g_centers, g_weight, policy->bias ? policy->g_bias : nullptr);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

extern "C" int mcp_rollout_bwd_ex(const mcp_model* model, const mcp_policy* policy, const mcp_noise* noise, int M, int T,
                               const double* states, const double* inputs, const double* jac, const double* g_states,
                               const double* g_inputs, double* g_log_ls, double* g_centers, double* g_weight, double* g_x0,
                               void* workspace, size_t workspace_bytes, void* stream, mcp_dispatch* d) {
  // (the request travels with the call: include/mcpilco_hip_debug.h; d == NULL: automatic)
  int last_lean = 0, last_pipe = 0;
  const int rc = rollout_bwd_impl(model, policy, noise, M, T, states, inputs, jac, g_states, g_inputs, g_log_ls, g_centers, g_weight, g_x0, workspace, workspace_bytes, stream, d ? (unsigned long long*)d->bwd_stamps : nullptr, d ? d->bwd_particles : 0,
                                  (d && d->bwd_lean == 1) ? 0 : -1, last_lean, (d && d->bwd_pipe == 1) ? 0 : -1, &last_pipe);
  if (d) d->ran_bwd_lean = last_lean;
  if (d) d->ran_bwd_pipe = last_pipe;
  return rc;
}
extern "C" int mcp_rollout_bwd(const mcp_model* model, const mcp_policy* policy, const mcp_noise* noise, int M, int T,
                               const double* states, const double* inputs, const double* jac, const double* g_states,
                               const double* g_inputs, double* g_log_ls, double* g_centers, double* g_weight, double* g_x0,
                               void* workspace, size_t workspace_bytes, void* stream) {
  return mcp_rollout_bwd_ex(model, policy, noise, M, T, states, inputs, jac, g_states, g_inputs, g_log_ls, g_centers, g_weight, g_x0, workspace, workspace_bytes, stream, nullptr);
}
