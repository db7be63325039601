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
};

struct BwdLayout {
  int invl, x, u, J, gs, xn, xb, zb, db, ub, ab, sf, sb, red, total;
};
__host__ __device__ inline BwdLayout bwd_layout(int S, int U, int D, int G, int PF, int NW) {
  BwdLayout L;
  int o = 0;
  auto take = [&](int n) {
    int r = o;
    o += (n + 1) & ~1;
    return r;
  };
  L.invl = take(PF);
  L.x = take(S);
  L.u = take(U);
  L.J = take(G * D);
  L.gs = take(S + U);
  L.xn = take(S);
  L.xb = take(S);
  L.zb = take(D);
  L.db = take(G);
  L.ub = take(U);
  L.ab = take(U);
  L.sf = take(PF);
  L.sb = take(PF);
  L.red = take(NW * PF);
  L.total = o;
  return L;
}

template <int PFM, int UM, int MAXNT>
__global__ __launch_bounds__(MAXNT) void rollout_bwd_kernel(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const mcp_model& md = a.model;
  const mcp_policy& pl = a.pol;
  const int tid = threadIdx.x, NT = blockDim.x, NW = NT >> 6, wv = tid >> 6, lane = tid & 63;
  const int S = md.S, U = md.U, G = md.G, D = md.D, B = pl.B, PF = pl.P, M = a.M, T = a.T;
  const BwdLayout L = bwd_layout(S, U, D, G, PF, NW);
  double* invl = smem + L.invl;
  double* xc = smem + L.x;
  double* uc = smem + L.u;
  double* Jr = smem + L.J;
  double* gsu = smem + L.gs;  // [S] upstream dJ/dx_t, then [U] upstream dJ/du_t
  double* xn = smem + L.xn;   // adjoint of x_{t+1}
  double* xb = smem + L.xb;   // adjoint of x_t (being built)
  double* zb = smem + L.zb;
  double* db = smem + L.db;
  double* ub = smem + L.ub;
  double* ab = smem + L.ab;
  double* sf = smem + L.sf;
  double* sb = smem + L.sb;
  double* red = smem + L.red;
  const int b = tid;
  const bool act = b < B;
  const bool drop = pl.p_drop > 0.0;
  const double keep_scale = 1.0 / (1.0 - pl.p_drop);
  const uint32_t drop_thr = drop_threshold(pl.p_drop);
  const int nna_g = md.n_not_angle, na_g = md.n_angle;

  for (int it = tid; it < PF; it += NT) invl[it] = exp(-pl.log_ls[it]);
  double cen[PFM], gc[PFM], gl[PFM], wgt[UM], gw[UM];
#pragma unroll
  for (int q = 0; q < PFM; ++q) {
    cen[q] = (act && q < PF) ? pl.centers[(size_t)b * PF + q] : 0.0;
    gc[q] = 0.0;
    gl[q] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < UM; ++k) {
    wgt[k] = (act && k < U) ? pl.weight[(size_t)k * B + b] : 0.0;
    gw[k] = 0.0;
  }
  __syncthreads();

  for (int m = blockIdx.x; m < M; m += gridDim.x) {
    for (int it = tid; it < S; it += NT) xn[it] = 0.0;
    for (int t = T - 1; t >= 0; --t) {
      __syncthreads();
      // ---- stage A: this step's record ------------------------------------------------------
      const size_t tm = (size_t)t * M + m;
      for (int it = tid; it < S; it += NT) {
        xc[it] = a.states[tm * S + it];
        gsu[it] = a.g_states ? a.g_states[tm * S + it] : 0.0;
      }
      for (int it = tid; it < U; it += NT) {
        uc[it] = a.inputs[tm * U + it];
        gsu[S + it] = a.g_inputs ? a.g_inputs[tm * U + it] : 0.0;
      }
      if (t < T - 1) {
        for (int it = tid; it < G * D; it += NT) Jr[it] = a.jac[tm * G * D + it];
        // adjoint of delta_g:  x_{t+1}[vel] = x[vel] + delta ; x_{t+1}[pos] = x[pos] + Ts x[vel] + Ts/2 delta
        for (int it = tid; it < G; it += NT) db[it] = xn[md.vel[it]] + 0.5 * md.Ts * xn[md.not_vel[it]];
      }
      __syncthreads();
      // ---- stage B: through the integrator and the GP Jacobian --------------------------------
      for (int it = tid; it < D; it += NT) {
        double s = 0.0;
        if (t < T - 1)
          for (int g = 0; g < G; ++g) s = fma(db[g], Jr[g * D + it], s);
        zb[it] = s;
      }
      for (int it = tid; it < S; it += NT) {
        double s = gsu[it];
        if (t < T - 1) {
          for (int g = 0; g < G; ++g) {
            if (md.vel[g] == it) s += xn[it] + md.Ts * xn[md.not_vel[g]];
            if (md.not_vel[g] == it) s += xn[it];
          }
        }
        xb[it] = s;
      }
      for (int it = tid; it < PF; it += NT) sf[it] = policy_feature(pl, xc, it, t);
      __syncthreads();
      // ---- stage C: through the GP feature map; adjoint of the pre-squash activation ------------
      for (int it = tid; it < S; it += NT) {
        double s = 0.0;
        for (int i = 0; i < nna_g; ++i)
          if (md.not_angle[i] == it) s += zb[i];
        for (int i = 0; i < na_g; ++i)
          if (md.angle[i] == it) s += zb[nna_g + i] * cos(xc[it]) - zb[nna_g + na_g + i] * sin(xc[it]);
        xb[it] += s;
      }
      for (int it = tid; it < U; it += NT) {
        double ubar = gsu[S + it] + zb[nna_g + 2 * na_g + it];
        ub[it] = ubar;
        double um = pl.u_max[it];
        double th = uc[it] / um;  // = tanh(a/u_max)
        ab[it] = pl.squash ? ubar * (1.0 - th * th) : ubar;
      }
      __syncthreads();
      // ---- stage D: RBF network, thread b owns basis b --------------------------------------------
      double dd = 0.0;  // adjoint of dist_b (0 for idle threads, so they add nothing below)
      if (act) {
        double dist = 0.0;
#pragma unroll
        for (int q = 0; q < PFM; ++q) {
          if (q < PF) {
            double r = (sf[q] - cen[q]) * invl[q];
            dist = fma(r, r, dist);
          }
        }
        double phi = exp(-dist);
        double mk = 1.0;
        if (drop) {
          bool keep = a.nz.masks ? (a.nz.masks[tm * B + b] != 0) : philox_keep(a.nz, m, t, b, drop_thr);
          mk = keep ? keep_scale : 0.0;
        }
        double phibar = 0.0;
#pragma unroll
        for (int k = 0; k < UM; ++k) {
          if (k < U) {
            gw[k] = fma(ab[k], phi * mk, gw[k]);
            phibar = fma(wgt[k], ab[k], phibar);
          }
        }
        dd = -phi * mk * phibar;
      }
#pragma unroll
      for (int q = 0; q < PFM; ++q) {
        if (q < PF) {
          double r = (sf[q] - cen[q]) * invl[q];
          double t2 = 2.0 * dd * r;
          gc[q] = fma(-t2, invl[q], gc[q]);
          gl[q] = fma(-t2, r, gl[q]);
          double s = wave_sum(t2 * invl[q]);
          if (lane == 0) red[wv * PF + q] = s;
        }
      }
      __syncthreads();
      for (int it = tid; it < PF; it += NT) {
        double s = 0.0;
        for (int w = 0; w < NW; ++w) s += red[w * PF + it];
        sb[it] = s;
      }
      __syncthreads();
      // ---- stage E: through the policy feature map; x_bar complete -> becomes x_{t+1}'s adjoint ----
      for (int it = tid; it < S; it += NT) {
        double s = 0.0;
        if (pl.kind == MCP_POLICY_ANGLES) {
          int nna = pl.n_non_angle, na = pl.n_angle;
          for (int i = 0; i < nna; ++i)
            if (pl.non_angle[i] == it) s += sb[i];
          for (int i = 0; i < na; ++i)
            if (pl.angle[i] == it) s += -sb[nna + i] * sin(xc[it]) + sb[nna + na + i] * cos(xc[it]);
        } else if (pl.kind == MCP_POLICY_TRAJ) {
          s = sb[it] - sb[S + it];
        } else {
          s = sb[it];
        }
        xn[it] = xb[it] + s;
      }
    }
    __syncthreads();
    if (a.g_x0)
      for (int it = tid; it < S; it += NT) a.g_x0[(size_t)m * S + it] = xn[it];
    __syncthreads();
  }

  // ---- write this workgroup's partial parameter gradients ------------------------------------
  const int nparam = PF + B * PF + U * B;
  double* out = a.slab + (size_t)blockIdx.x * nparam;
  if (act) {
#pragma unroll
    for (int q = 0; q < PFM; ++q)
      if (q < PF) out[PF + (size_t)b * PF + q] = gc[q];
#pragma unroll
    for (int k = 0; k < UM; ++k)
      if (k < U) out[PF + (size_t)B * PF + (size_t)k * B + b] = gw[k];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PFM; ++q) {
    if (q < PF) {
      double s = wave_sum(gl[q]);
      if (lane == 0) red[wv * PF + q] = s;
    }
  }
  __syncthreads();
  for (int it = tid; it < PF; it += NT) {
    double s = 0.0;
    for (int w = 0; w < NW; ++w) s += red[w * PF + it];
    out[it] = s;
  }
}

// sum the per-workgroup slabs in a fixed order (deterministic, no atomics)
__global__ void grad_reduce_kernel(int nblk, int nparam, int PF, int BPF, const double* __restrict__ slab, double* __restrict__ g_log_ls,
                                   double* __restrict__ g_centers, double* __restrict__ g_weight) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nparam) return;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int k = 0;
  for (; k + 3 < nblk; k += 4) {
    s0 += slab[(size_t)k * nparam + i];
    s1 += slab[(size_t)(k + 1) * nparam + i];
    s2 += slab[(size_t)(k + 2) * nparam + i];
    s3 += slab[(size_t)(k + 3) * nparam + i];
  }
  for (; k < nblk; ++k) s0 += slab[(size_t)k * nparam + i];
  double s = (s0 + s1) + (s2 + s3);
  if (i < PF)
    g_log_ls[i] = s;
  else if (i < PF + BPF)
    g_centers[i - PF] = s;
  else
    g_weight[i - PF - BPF] = s;
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
static int bwd_threads(int B) { return imax(64, ((B + 63) / 64) * 64); }
static int bwd_blocks(int M) { return imin(M, 1024); }

extern "C" size_t mcp_rollout_workspace_bytes(const mcp_model* model, const mcp_policy* policy, int M, int T) {
  if (!model || !policy || M <= 0 || T <= 0) return 0;
  size_t nparam = (size_t)policy->P + (size_t)policy->B * policy->P + (size_t)policy->U * policy->B;
  return sizeof(double) * nparam * (size_t)bwd_blocks(M);
}

template <int PFM, int UM, int MAXNT>
static int launch_bwd(const BwdArgs& a, int grid, int NT, size_t lds, hipStream_t st) {
  if (NT > MAXNT) return MCP_ERR_LIMIT;
  hipLaunchKernelGGL((rollout_bwd_kernel<PFM, UM, MAXNT>), dim3(grid), dim3(NT), lds, st, a);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

extern "C" int mcp_rollout_bwd(const mcp_model* model, const mcp_policy* policy, const mcp_noise* noise, int M, int T,
                               const double* states, const double* inputs, const double* jac, const double* g_states,
                               const double* g_inputs, double* g_log_ls, double* g_centers, double* g_weight, double* g_x0,
                               void* workspace, size_t workspace_bytes, void* stream) {
  if (!noise || !states || !inputs || !g_log_ls || !g_centers || !g_weight || !workspace || M <= 0 || T <= 0) return MCP_ERR_ARG;
  if (T > 1 && !jac) return MCP_ERR_ARG;
  if (!model_ok(model)) return MCP_ERR_ARG;
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
  const int NT = bwd_threads(policy->B);
  const int grid = bwd_blocks(M);
  BwdLayout L = bwd_layout(model->S, model->U, model->D, model->G, policy->P, NT / 64);
  size_t lds = sizeof(double) * (size_t)L.total;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  const int PF = policy->P, U = policy->U;
  // register budget: 3*PFM + 2*UM doubles of per-thread accumulators -> the widest variant runs
  // with at most 512 threads (B <= 512) so that it keeps 256 VGPRs per lane
  if (PF <= 8 && U <= 2)
    rc = launch_bwd<8, 2, 1024>(a, grid, NT, lds, st);
  else if (PF <= 16 && U <= 4)
    rc = launch_bwd<16, 4, 1024>(a, grid, NT, lds, st);
  else
    rc = launch_bwd<MCP_MAX_PFEAT, MCP_MAX_INPUT, 512>(a, grid, NT, lds, st);
  if (rc != MCP_OK) return rc;
  const int nparam = PF + policy->B * PF + U * policy->B;
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((nparam + 255) / 256), dim3(256), 0, st, grid, nparam, PF, policy->B * PF, a.slab, g_log_ls,
                     g_centers, g_weight);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

