// Fused Monte-Carlo particle rollout for gfx950 (MI355X): forward T-loop and reverse-time adjoint.
//
// Replaces MC_PILCO.apply_policy (policy_learning/MC_PILCO.py:615-674) -- the per-step chain
//   Model_learning.get_next_state  (model_learning/Model_learning.py:210-229, 231-242, 265-336)
//     -> GP_prior.get_estimate_from_alpha (gpr_lib/GP_prior/GP_prior.py:137-155), one per GP
//     -> get_next_state_from_gp_output    (Model_learning.py:685-718)
//   Sum_of_gaussians*.forward            (policy_learning/Policy.py:242-265, 323-335, 389-403)
// and autograd's backward through it (MC_PILCO.py:522).
//
// Parallel axis: particles.  They never interact inside the rollout, so a workgroup owns P
// particles for all T steps (no inter-workgroup synchronisation anywhere) and keeps their state
// in LDS.  Per step and GP the work is   k = k(z, X) [N],  v = Kinv k [N]  (the N^2 term),
// mu = m + k.alpha,  var = k(z,z) - k.v,  and the Jacobians d mu/dz, d var/dz, which are formed
// HERE, in the forward pass, from v (d var/dz = dk(z,z)/dz - 2 sum_j v_j dk_j/dz).  The forward
// stores only d delta_g/dz (G x D doubles per particle-step, sampling folded in), so the backward
// pass never touches the GP again: it is a cheap reverse sweep through integration, feature maps
// and the RBF policy (rollout_bwd_kernel, one thread per basis function).
//
// Kinv (N x N fp64, 720 KB per GP at N=300) does not fit the 160 KiB LDS; it stays L2-resident
// and is streamed once per step per workgroup with coalesced row reads (Kinv is symmetric, so
// column i of the product is read as row-contiguous data).  DESIGN.md has the roofline.
#include "mcp_device.h"

using namespace mcp;

#define RF_MAX_NA 7  // accumulators per Jacobian task: 2 (SE), 3 (SE+P1), 7 (SE+P2)

struct FwdLayout {
  int invl, xs, us, z, sf, dl, kb, ks, pa, pb, vb, part, red, total;  // offsets in doubles
};

__host__ __device__ inline int imax(int a, int b) { return a > b ? a : b; }
__host__ __device__ inline int imin(int a, int b) { return a < b ? a : b; }

__host__ __device__ inline FwdLayout fwd_layout(int P, int S, int U, int D, int G, int PF, int B, int NpadMax, int maxdeg, int NT) {
  FwdLayout L;
  int o = 0;
  auto take = [&](int n) {
    int r = o;
    o += (n + 1) & ~1;  // keep 16-byte alignment
    return r;
  };
  L.invl = take(PF);
  L.xs = take(P * S);
  L.us = take(P * U);
  L.z = take(P * D);
  L.sf = take(P * PF);
  L.dl = take(P * G);
  L.kb = take(NpadMax * P);
  L.ks = maxdeg > 0 ? take(NpadMax * P) : L.kb;
  L.pa = maxdeg > 1 ? take(NpadMax * P) : L.kb;
  L.pb = maxdeg > 1 ? take(NpadMax * P) : L.kb;
  L.vb = take(NpadMax * P);
  int part = imax(imax(NT * P, NpadMax * P), imax(P * B, NT * RF_MAX_NA));
  L.part = take(part);
  L.red = take(P * (D + 1) * RF_MAX_NA);
  L.total = o;
  return L;
}

struct FwdArgs {
  mcp_model model;
  mcp_policy pol;
  mcp_noise nz;
  int M, T, particle_pred;
  int NpadMax, maxdeg;
  const double* x0;
  double* states;
  double* inputs;
  double* jac;
  uint32_t* status;
};

// ---- feature maps -----------------------------------------------------------------------
// GP input z = [x[not_angle], sin x[angle], cos x[angle], u]   (Model_learning.py:670-683)
__device__ __forceinline__ double gp_feature(const mcp_model& md, const double* x, const double* u, int d) {
  int nna = md.n_not_angle, na = md.n_angle;
  if (d < nna) return x[md.not_angle[d]];
  if (d < nna + na) return sin(x[md.angle[d - nna]]);
  if (d < nna + 2 * na) return cos(x[md.angle[d - nna - na]]);
  return u[d - nna - 2 * na];
}

// policy feature s  (Policy.py:326-333: [x_nonangle, COS, SIN];  :397-399: [x, x*_t - x])
__device__ __forceinline__ double policy_feature(const mcp_policy& pl, const double* x, int q, int t) {
  if (pl.kind == MCP_POLICY_ANGLES) {
    int nna = pl.n_non_angle, na = pl.n_angle;
    if (q < nna) return x[pl.non_angle[q]];
    if (q < nna + na) return cos(x[pl.angle[q - nna]]);
    return sin(x[pl.angle[q - nna - na]]);
  }
  if (pl.kind == MCP_POLICY_TRAJ) {
    if (q < pl.S) return x[q];
    return pl.target_traj[(size_t)t * pl.S + (q - pl.S)] - x[q - pl.S];
  }
  return x[q];
}

// ---- per-GP phases (shared by the rollout and by mcp_posterior_fwd) ------------------------
// Phase K: covariance vector(s) of P test points against the N training points -> LDS [Npad][P]
template <int P>
__device__ __forceinline__ void gp_phase_k(const mcp_gp& gp, const double* z, double* kb, double* ks, double* pa, double* pb, int tid,
                                           int NT) {
  const mcp_kernel& kn = gp.kern;
  const int D = kn.D, N = gp.N, Npad = gp.Npad, deg = kn.poly_deg;
  for (int it = tid; it < P * Npad; it += NT) {
    int p = it / Npad, j = it - p * Npad;
    double kse = 0.0, kt = 0.0, A = 0.0, Bv = 0.0;
    if (j < N) {
      const double* zp = z + p * D;
      double dist = 0.0;
      for (int d = 0; d < D; ++d) {
        double r = (zp[d] - gp.Xt[(size_t)d * Npad + j]) * kn.inv_ls[d];
        dist = fma(r, r, dist);
      }
      kse = kn.lambda * exp(-dist);
      kt = kse;
      if (deg >= 1) {
        double p1 = kn.w1[D];
        for (int d = 0; d < D; ++d) p1 = fma(kn.w1[d] * zp[d], gp.Xt[(size_t)d * Npad + j], p1);
        kt += p1;
        if (deg >= 2) {
          for (int d = 0; d < D; ++d) {
            double zx = zp[d] * gp.Xt[(size_t)d * Npad + j];
            A = fma(kn.w20[d], zx, A);
            Bv = fma(kn.w21[d], zx, Bv);
          }
          kt = fma(A, Bv, kt);
        }
      }
    }
    kb[j * P + p] = kt;
    if (deg >= 1) ks[j * P + p] = kse;
    if (deg >= 2) {
      pa[j * P + p] = A;
      pb[j * P + p] = Bv;
    }
  }
}

// Phase V (VALU form): partial products of v = Kinv k.  A wave owns 64 rows i (lane <-> row) and
// a slice of the summation index j; Kinv is symmetric so element (i,j) is read from row j -- 64
// consecutive doubles per wave-load.  k_j[0..P) is an LDS broadcast read.
template <int P>
__device__ __forceinline__ void gp_phase_v(const mcp_gp& gp, const double* kb, double* part, int tid, int NT) {
  const int N = gp.N, Npad = gp.Npad;
  const int NW = NT >> 6, wv = tid >> 6, lane = tid & 63;
  const int NIC = (Npad + 63) >> 6;
  const int NJS = imax(1, NW / NIC);
  const int Jlen = (N + NJS - 1) / NJS;
  for (int task = wv; task < NIC * NJS; task += NW) {
    int ic = task % NIC, js = task / NIC;
    int i = ic * 64 + lane;
    bool act = i < Npad;
    int j0 = js * Jlen, j1 = imin(N, j0 + Jlen);
    double acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p) acc[p] = 0.0;
    const double* col = gp.Kinv + (act ? i : 0);
    int j = j0;
    for (; j + 8 <= j1; j += 8) {
      double a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = col[(size_t)(j + u) * Npad];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p] = fma(a[u], kb[(j + u) * P + p], acc[p]);
      }
    }
    for (; j < j1; ++j) {
      double a0 = col[(size_t)j * Npad];
#pragma unroll
      for (int p = 0; p < P; ++p) acc[p] = fma(a0, kb[j * P + p], acc[p]);
    }
    if (act) {
#pragma unroll
      for (int p = 0; p < P; ++p) part[((size_t)js * Npad + i) * P + p] = acc[p];
    }
  }
}

template <int P>
__device__ __forceinline__ void gp_phase_vsum(const mcp_gp& gp, const double* part, double* vb, int tid, int NT) {
  const int Npad = gp.Npad, NW = NT >> 6;
  const int NIC = (Npad + 63) >> 6;
  const int NJS = imax(1, NW / NIC);
  for (int it = tid; it < Npad * P; it += NT) {
    double s = 0.0;
    for (int js = 0; js < NJS; ++js) s += part[(size_t)js * Npad * P + it];
    vb[it] = s;
  }
}

__device__ __forceinline__ int gp_num_acc(int deg) { return deg == 0 ? 2 : (deg == 1 ? 3 : RF_MAX_NA); }

// Phase J: the moment / Jacobian sums, a skinny GEMM [P x N].[N x (D+1)] per accumulator kind.
//   column c <  D : a0 = sum kse_j alpha_j (z_c - X_jc)      a1 = sum kse_j v_j (z_c - X_jc)
//                   a2 = sum v_j X_jc            (deg>=1)
//                   a3 = sum alpha_j B_j X_jc, a4 = sum alpha_j A_j X_jc, a5 = sum v_j B_j X_jc, a6 = sum v_j A_j X_jc (deg 2)
//   column c == D : a0 = sum k_j alpha_j  (= mu - m)          a1 = sum k_j v_j  (= k^T Kinv k)
template <int P>
__device__ __forceinline__ void gp_phase_j(const mcp_gp& gp, const double* z, const double* kb, const double* ks, const double* pa,
                                           const double* pb, const double* vb, double* part, int tid, int NT) {
  const int D = gp.kern.D, N = gp.N, deg = gp.kern.poly_deg;
  const int NA = gp_num_acc(deg);
  const int NIT = P * (D + 1);
  const int NJS2 = imin(imax(1, NT / NIT), 64);
  if (tid < NJS2 * NIT) {
    int js = tid / NIT, item = tid - js * NIT;
    int p = item / (D + 1), c = item - p * (D + 1);
    double acc[RF_MAX_NA];
#pragma unroll
    for (int a = 0; a < RF_MAX_NA; ++a) acc[a] = 0.0;
    if (c < D) {
      double zc = z[p * D + c];
      for (int j = js; j < N; j += NJS2) {
        double al = gp.alpha[j], v = vb[j * P + p], x = gp.X[(size_t)j * D + c];
        double kse = ks[j * P + p];
        double dz = zc - x;
        acc[0] = fma(kse * al, dz, acc[0]);
        acc[1] = fma(kse * v, dz, acc[1]);
        if (deg >= 1) acc[2] = fma(v, x, acc[2]);
        if (deg >= 2) {
          double A = pa[j * P + p], Bv = pb[j * P + p];
          acc[3] = fma(al * Bv, x, acc[3]);
          acc[4] = fma(al * A, x, acc[4]);
          acc[5] = fma(v * Bv, x, acc[5]);
          acc[6] = fma(v * A, x, acc[6]);
        }
      }
    } else {
      for (int j = js; j < N; j += NJS2) {
        double kt = kb[j * P + p];
        acc[0] = fma(kt, gp.alpha[j], acc[0]);
        acc[1] = fma(kt, vb[j * P + p], acc[1]);
      }
    }
#pragma unroll
    for (int a = 0; a < RF_MAX_NA; ++a)
      if (a < NA) part[((size_t)js * NIT + item) * NA + a] = acc[a];
  }
}

template <int P>
__device__ __forceinline__ void gp_phase_jsum(const mcp_gp& gp, const double* part, double* red, int tid, int NT) {
  const int D = gp.kern.D, NA = gp_num_acc(gp.kern.poly_deg);
  const int NIT = P * (D + 1);
  const int NJS2 = imin(imax(1, NT / NIT), 64);
  for (int it = tid; it < NIT * NA; it += NT) {
    double s = 0.0;
    for (int js = 0; js < NJS2; ++js) s += part[(size_t)js * NIT * NA + it];
    red[it] = s;
  }
}

// posterior mean / variance and their z-Jacobians for one (particle, column) from the reduced sums
struct GpPoint {
  double mu, var;
};
__device__ __forceinline__ GpPoint gp_point(const mcp_gp& gp, const double* zp, const double* R /* [(D+1)][NA] */) {
  const int D = gp.kern.D, NA = gp_num_acc(gp.kern.poly_deg);
  GpPoint o;
  o.mu = gp.kern.mean + R[D * NA + 0];
  o.var = kern_diag(gp.kern, zp, 1) - R[D * NA + 1];
  return o;
}
__device__ __forceinline__ void gp_jac(const mcp_gp& gp, const double* zp, const double* R, int d, double& Jmu, double& Jvar) {
  const mcp_kernel& kn = gp.kern;
  const int D = kn.D, deg = kn.poly_deg, NA = gp_num_acc(deg);
  const double* r = R + d * NA;
  double il2 = kn.inv_ls[d] * kn.inv_ls[d];
  Jmu = -2.0 * il2 * r[0];
  Jvar = 4.0 * il2 * r[1];
  if (deg >= 1) {
    Jmu = fma(kn.w1[d], gp.aX[d], Jmu);
    Jvar += 2.0 * kn.w1[d] * (zp[d] - r[2]);
    if (deg >= 2) {
      double Sa = 0.0, Sb = 0.0;
      for (int e = 0; e < D; ++e) {
        double zz = zp[e] * zp[e];
        Sa = fma(kn.w20[e], zz, Sa);
        Sb = fma(kn.w21[e], zz, Sb);
      }
      Jmu += kn.w20[d] * r[3] + kn.w21[d] * r[4];
      Jvar += 2.0 * zp[d] * (kn.w20[d] * Sb + kn.w21[d] * Sa) - 2.0 * (kn.w20[d] * r[5] + kn.w21[d] * r[6]);
    }
  }
}

// ---------------------------------------------------------------------------------------
// forward rollout
// ---------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(1024) void rollout_fwd_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const mcp_model& md = a.model;
  const mcp_policy& pl = a.pol;
  const int tid = threadIdx.x, NT = blockDim.x, NW = NT >> 6, wv = tid >> 6, lane = tid & 63;
  const int S = md.S, U = md.U, G = md.G, D = md.D, B = pl.B, PF = pl.P, M = a.M, T = a.T;
  const FwdLayout L = fwd_layout(P, S, U, D, G, PF, B, a.NpadMax, a.maxdeg, NT);
  double* invl = smem + L.invl;
  double* xs = smem + L.xs;
  double* us = smem + L.us;
  double* z = smem + L.z;
  double* sf = smem + L.sf;
  double* dl = smem + L.dl;
  double* kb = smem + L.kb;
  double* ks = smem + L.ks;
  double* pa = smem + L.pa;
  double* pb = smem + L.pb;
  double* vb = smem + L.vb;
  double* part = smem + L.part;
  double* red = smem + L.red;
  const int m0 = blockIdx.x * P;
  uint32_t bad = 0;
  const bool drop = pl.p_drop > 0.0;
  const double keep_scale = 1.0 / (1.0 - pl.p_drop);
  const uint32_t drop_thr = drop_threshold(pl.p_drop);

  for (int it = tid; it < PF; it += NT) invl[it] = exp(-pl.log_ls[it]);
  for (int it = tid; it < P * S; it += NT) {
    int p = it / S, s = it - p * S;
    int mm = imin(m0 + p, M - 1);
    xs[it] = a.x0[(size_t)mm * S + s];
  }
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    // ---- record x_t, policy features -------------------------------------------------------
    for (int it = tid; it < P * S; it += NT) {
      int p = it / S, s = it - p * S;
      double v = xs[it];
      if (m0 + p < M) {
        a.states[((size_t)t * M + m0 + p) * S + s] = v;
        if (is_bad(v)) bad |= MCP_STATUS_NAN;
      }
    }
    for (int it = tid; it < P * PF; it += NT) {
      int p = it / PF, q = it - p * PF;
      sf[it] = policy_feature(pl, xs + p * S, q, t);
    }
    __syncthreads();
    // ---- basis functions with dropout:  phi_b = exp(-sum_q ((s_q - c_bq)/l_q)^2) * keep/(1-p) ---
    double* ph = part;
    for (int it = tid; it < P * B; it += NT) {
      int p = it / B, b = it - p * B;
      const double* cb = pl.centers + (size_t)b * PF;
      double dist = 0.0;
      for (int q = 0; q < PF; ++q) {
        double r = (sf[p * PF + q] - cb[q]) * invl[q];
        dist = fma(r, r, dist);
      }
      double phi = exp(-dist);
      if (drop) {
        int mm = imin(m0 + p, M - 1);
        bool keep = a.nz.masks ? (a.nz.masks[((size_t)t * M + mm) * B + b] != 0) : philox_keep(a.nz, mm, t, b, drop_thr);
        phi = keep ? phi * keep_scale : 0.0;
      }
      ph[it] = phi;
    }
    __syncthreads();
    // ---- u = u_max tanh((W phi)/u_max): one wave per (particle, input) ----------------------
    for (int task = wv; task < P * U; task += NW) {
      int p = task / U, k = task - p * U;
      const double* wk = pl.weight + (size_t)k * B;
      double s = 0.0;
      for (int b = lane; b < B; b += 64) s = fma(wk[b], ph[p * B + b], s);
      s = wave_sum(s);
      if (lane == 0) {
        double um = pl.u_max[k];
        double u = pl.squash ? um * tanh(s / um) : s;
        us[p * U + k] = u;
        if (m0 + p < M) {
          a.inputs[((size_t)t * M + m0 + p) * U + k] = u;
          if (is_bad(u)) bad |= MCP_STATUS_NAN;
        }
      }
    }
    __syncthreads();
    if (t == T - 1) break;
    // ---- GP input features ------------------------------------------------------------------
    for (int it = tid; it < P * D; it += NT) {
      int p = it / D, d = it - p * D;
      z[it] = gp_feature(md, xs + p * S, us + p * U, d);
    }
    __syncthreads();
    // ---- one GP after the other -------------------------------------------------------------
    for (int g = 0; g < G; ++g) {
      const mcp_gp& gp = md.gp[g];
      gp_phase_k<P>(gp, z, kb, ks, pa, pb, tid, NT);
      __syncthreads();
      gp_phase_v<P>(gp, kb, part, tid, NT);
      __syncthreads();
      gp_phase_vsum<P>(gp, part, vb, tid, NT);
      __syncthreads();
      gp_phase_j<P>(gp, z, kb, ks, pa, pb, vb, part, tid, NT);
      __syncthreads();
      gp_phase_jsum<P>(gp, part, red, tid, NT);
      __syncthreads();
      // finalize: sample delta_g and fold the sampling into d delta/dz
      const int NA = gp_num_acc(gp.kern.poly_deg);
      for (int it = tid; it < P * (D + 1); it += NT) {
        int p = it / (D + 1), c = it - p * (D + 1);
        const double* R = red + (size_t)p * (D + 1) * NA;
        const double* zp = z + p * D;
        int mm = imin(m0 + p, M - 1);
        GpPoint pt = gp_point(gp, zp, R);
        double var = pt.var * md.var_scale[g];
        double eps = 0.0, wj = 0.0, sd = 0.0;
        if (a.particle_pred) {
          eps = a.nz.eps ? a.nz.eps[((size_t)t * M + mm) * G + g] : philox_normal(a.nz, mm, t, g);
          sd = sqrt(var);
          wj = eps / (2.0 * sd);
        }
        if (c == D) {
          dl[p * G + g] = a.particle_pred ? fma(sd, eps, pt.mu) : pt.mu;
          if (m0 + p < M) {
            if (a.particle_pred && !(var > 0.0)) bad |= MCP_STATUS_NONPOS_VAR;
            if (is_bad(pt.mu) || is_bad(var)) bad |= MCP_STATUS_NAN;
          }
        } else if (a.jac && m0 + p < M) {
          double Jmu, Jvar;
          gp_jac(gp, zp, R, c, Jmu, Jvar);
          a.jac[(((size_t)t * M + m0 + p) * G + g) * D + c] = a.particle_pred ? fma(wj, Jvar * md.var_scale[g], Jmu) : Jmu;
        }
      }
      // no barrier needed here: the next GP's phase K writes kb/ks/pa/pb only, its phase V writes
      // `part`, and both are separated from the reads above by the barriers that follow.
    }
    __syncthreads();
    // ---- integrate:  v' = v + delta ;  q' = q + Ts v + Ts/2 delta   (Model_learning.py:711-716) ---
    double xn = 0.0;
    int my = -1;
    if (tid < P * S) {
      int p = tid / S, s = tid - p * S;
      my = tid;
      for (int g = 0; g < G; ++g) {
        if (md.vel[g] == s) xn = xs[p * S + s] + dl[p * G + g];
        if (md.not_vel[g] == s) xn = xs[p * S + s] + md.Ts * xs[p * S + md.vel[g]] + 0.5 * md.Ts * dl[p * G + g];
      }
    }
    __syncthreads();
    if (my >= 0) xs[my] = xn;
    __syncthreads();
  }
  if (bad) atomicOr(a.status, bad);
}

// ---------------------------------------------------------------------------------------
// single-step posterior (GP_prior.get_estimate_from_alpha) through the same phases
// ---------------------------------------------------------------------------------------
struct PostArgs {
  mcp_gp gp;
  int M;
  const double* Z;
  double* mu;
  double* var;
  double* Jmu;
  double* Jvar;
  uint32_t* status;
};

template <int P>
__global__ __launch_bounds__(1024) void posterior_fwd_kernel(PostArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const mcp_gp& gp = a.gp;
  const int tid = threadIdx.x, NT = blockDim.x;
  const int D = gp.kern.D, deg = gp.kern.poly_deg;
  const FwdLayout L = fwd_layout(P, 1, 1, D, 1, 1, 1, gp.Npad, deg, NT);
  double* z = smem + L.z;
  double* kb = smem + L.kb;
  double* ks = smem + L.ks;
  double* pa = smem + L.pa;
  double* pb = smem + L.pb;
  double* vb = smem + L.vb;
  double* part = smem + L.part;
  double* red = smem + L.red;
  const int m0 = blockIdx.x * P;
  uint32_t bad = 0;
  for (int it = tid; it < P * D; it += NT) {
    int p = it / D, d = it - p * D;
    z[it] = a.Z[(size_t)imin(m0 + p, a.M - 1) * D + d];
  }
  __syncthreads();
  gp_phase_k<P>(gp, z, kb, ks, pa, pb, tid, NT);
  __syncthreads();
  gp_phase_v<P>(gp, kb, part, tid, NT);
  __syncthreads();
  gp_phase_vsum<P>(gp, part, vb, tid, NT);
  __syncthreads();
  gp_phase_j<P>(gp, z, kb, ks, pa, pb, vb, part, tid, NT);
  __syncthreads();
  gp_phase_jsum<P>(gp, part, red, tid, NT);
  __syncthreads();
  const int NA = gp_num_acc(deg);
  for (int it = tid; it < P * (D + 1); it += NT) {
    int p = it / (D + 1), c = it - p * (D + 1);
    if (m0 + p >= a.M) continue;
    const double* R = red + (size_t)p * (D + 1) * NA;
    const double* zp = z + p * D;
    if (c == D) {
      GpPoint pt = gp_point(gp, zp, R);
      a.mu[m0 + p] = pt.mu;
      a.var[m0 + p] = pt.var;
      if (is_bad(pt.mu) || is_bad(pt.var)) bad |= MCP_STATUS_NAN;
      if (!(pt.var > 0.0)) bad |= MCP_STATUS_NONPOS_VAR;
    } else if (a.Jmu) {
      double Jm, Jv;
      gp_jac(gp, zp, R, c, Jm, Jv);
      a.Jmu[(size_t)(m0 + p) * D + c] = Jm;
      a.Jvar[(size_t)(m0 + p) * D + c] = Jv;
    }
  }
  if (bad && a.status) atomicOr(a.status, bad);
}

__global__ void posterior_bwd_kernel(int M, int D, const double* __restrict__ gmu, const double* __restrict__ gvar,
                                     const double* __restrict__ Jmu, const double* __restrict__ Jvar, double* __restrict__ gZ) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * D) return;
  size_t m = i / D;
  gZ[i] = fma(gmu[m], Jmu[i], gvar[m] * Jvar[i]);
}

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
#define RF_NT 1024
#define MCP_LDS_LIMIT (160 * 1024)

static bool model_ok(const mcp_model* m) {
  if (!m) return false;
  if (m->S <= 0 || m->S > MCP_MAX_STATE || m->U <= 0 || m->U > MCP_MAX_INPUT || m->G <= 0 || m->G > MCP_MAX_GP) return false;
  if (m->D <= 0 || m->D > MCP_MAX_GPDIM) return false;
  if (m->n_angle < 0 || m->n_not_angle < 0 || m->n_not_angle + 2 * m->n_angle + m->U != m->D) return false;
  for (int g = 0; g < m->G; ++g) {
    const mcp_gp& gp = m->gp[g];
    if (gp.kern.D != m->D || gp.N <= 0 || gp.Npad < gp.N || (gp.Npad % 16) != 0 || gp.N > MCP_MAX_TRAIN) return false;
    if (!gp.Xt || !gp.X || !gp.alpha || !gp.Kinv || !gp.kern.inv_ls) return false;
    if (gp.kern.poly_deg < 0 || gp.kern.poly_deg > 2) return false;
    if (gp.kern.poly_deg >= 1 && (!gp.kern.w1 || !gp.aX)) return false;
    if (gp.kern.poly_deg >= 2 && (!gp.kern.w20 || !gp.kern.w21)) return false;
    if (m->vel[g] < 0 || m->vel[g] >= m->S || m->not_vel[g] < 0 || m->not_vel[g] >= m->S) return false;
  }
  return true;
}

static bool policy_ok(const mcp_policy* p, int S, int U, int T) {
  if (!p || p->S != S || p->U != U) return false;
  if (p->B <= 0 || p->B > MCP_MAX_BASIS || p->P <= 0 || p->P > MCP_MAX_PFEAT) return false;
  if (!p->log_ls || !p->centers || !p->weight || !p->u_max) return false;
  if (!(p->p_drop >= 0.0 && p->p_drop < 1.0)) return false;
  if (p->kind == MCP_POLICY_PLAIN) return p->P == S;
  if (p->kind == MCP_POLICY_ANGLES) return p->n_non_angle + 2 * p->n_angle == p->P;
  if (p->kind == MCP_POLICY_TRAJ) return p->P == 2 * S && p->target_traj && p->traj_len >= T;
  return false;
}

static int pick_particles_per_wg(int M) {
  // small swarms: spread over as many CUs as possible (each workgroup re-streams Kinv, so the
  // per-CU L2->L1 rate is the bound); large swarms: amortise the Kinv stream over more particles
  if (M <= 256) return 1;
  if (M <= 1024) return 2;
  return 4;
}

static void model_dims(const mcp_model* m, int* NpadMax, int* maxdeg) {
  int np = 0, dg = 0;
  for (int g = 0; g < m->G; ++g) {
    np = imax(np, m->gp[g].Npad);
    dg = imax(dg, m->gp[g].kern.poly_deg);
  }
  *NpadMax = np;
  *maxdeg = dg;
}

static int bwd_threads(int B) { return imax(64, ((B + 63) / 64) * 64); }
static int bwd_blocks(int M) { return imin(M, 1024); }

extern "C" size_t mcp_rollout_workspace_bytes(const mcp_model* model, const mcp_policy* policy, int M, int T) {
  if (!model || !policy || M <= 0 || T <= 0) return 0;
  size_t nparam = (size_t)policy->P + (size_t)policy->B * policy->P + (size_t)policy->U * policy->B;
  return sizeof(double) * nparam * (size_t)bwd_blocks(M);
}

template <int P>
static int launch_fwd(const FwdArgs& a, int NT, size_t lds, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(rollout_fwd_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize, MCP_LDS_LIMIT);
    attr_set = true;
  }
  int grid = (a.M + P - 1) / P;
  hipLaunchKernelGGL(rollout_fwd_kernel<P>, dim3(grid), dim3(NT), lds, st, a);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

static int g_force_ppw = 0;  // test hook: force particles per workgroup (0 = automatic)
extern "C" void mcp_debug_set_particles_per_wg(int p) { g_force_ppw = p; }

extern "C" int mcp_rollout_fwd(const mcp_model* model, const mcp_policy* policy, const mcp_noise* noise, int M, int T, int particle_pred,
                               const double* x0, double* states, double* inputs, double* jac, uint32_t* status, void* workspace,
                               size_t workspace_bytes, void* stream) {
  (void)workspace;
  (void)workspace_bytes;
  if (!noise || !x0 || !states || !inputs || !status || M <= 0 || T <= 0) return MCP_ERR_ARG;
  if (!model_ok(model)) return MCP_ERR_ARG;
  if (!policy_ok(policy, model->S, model->U, T)) return MCP_ERR_ARG;
  FwdArgs a;
  a.model = *model;
  a.pol = *policy;
  a.nz = *noise;
  a.M = M;
  a.T = T;
  a.particle_pred = particle_pred;
  model_dims(model, &a.NpadMax, &a.maxdeg);
  a.x0 = x0;
  a.states = states;
  a.inputs = inputs;
  a.jac = jac;
  a.status = status;
  int P = g_force_ppw ? g_force_ppw : pick_particles_per_wg(M);
  const int NT = RF_NT;
  // shrink P until the LDS layout fits
  for (;;) {
    FwdLayout L = fwd_layout(P, model->S, model->U, model->D, model->G, policy->P, policy->B, a.NpadMax, a.maxdeg, NT);
    size_t lds = sizeof(double) * (size_t)L.total;
    if (lds <= MCP_LDS_LIMIT) {
      hipStream_t st = (hipStream_t)stream;
      switch (P) {
        case 1: return launch_fwd<1>(a, NT, lds, st);
        case 2: return launch_fwd<2>(a, NT, lds, st);
        case 4: return launch_fwd<4>(a, NT, lds, st);
        default: return MCP_ERR_ARG;
      }
    }
    if (P == 1) return MCP_ERR_LIMIT;
    P >>= 1;
  }
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

template <int P>
static int launch_post(const PostArgs& a, size_t lds, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(posterior_fwd_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize, MCP_LDS_LIMIT);
    attr_set = true;
  }
  hipLaunchKernelGGL(posterior_fwd_kernel<P>, dim3((a.M + P - 1) / P), dim3(RF_NT), lds, st, a);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

extern "C" int mcp_posterior_fwd(const mcp_gp* gp, int M, const double* Z, double* mu, double* var, double* Jmu, double* Jvar,
                                 uint32_t* status, void* stream) {
  if (!gp || !Z || !mu || !var || M <= 0) return MCP_ERR_ARG;
  if ((Jmu == nullptr) != (Jvar == nullptr)) return MCP_ERR_ARG;
  if (gp->kern.D <= 0 || gp->kern.D > MCP_MAX_GPDIM || gp->N <= 0 || gp->Npad < gp->N || (gp->Npad % 16) != 0) return MCP_ERR_ARG;
  if (!gp->Xt || !gp->X || !gp->alpha || !gp->Kinv || !gp->kern.inv_ls) return MCP_ERR_ARG;
  if (gp->kern.poly_deg >= 1 && (!gp->kern.w1 || !gp->aX)) return MCP_ERR_ARG;
  if (gp->kern.poly_deg >= 2 && (!gp->kern.w20 || !gp->kern.w21)) return MCP_ERR_ARG;
  PostArgs a;
  a.gp = *gp;
  a.M = M;
  a.Z = Z;
  a.mu = mu;
  a.var = var;
  a.Jmu = Jmu;
  a.Jvar = Jvar;
  a.status = status;
  int P = g_force_ppw ? g_force_ppw : pick_particles_per_wg(M);
  for (;;) {
    FwdLayout L = fwd_layout(P, 1, 1, gp->kern.D, 1, 1, 1, gp->Npad, gp->kern.poly_deg, RF_NT);
    size_t lds = sizeof(double) * (size_t)L.total;
    if (lds <= MCP_LDS_LIMIT) {
      hipStream_t st = (hipStream_t)stream;
      switch (P) {
        case 1: return launch_post<1>(a, lds, st);
        case 2: return launch_post<2>(a, lds, st);
        case 4: return launch_post<4>(a, lds, st);
        default: return MCP_ERR_ARG;
      }
    }
    if (P == 1) return MCP_ERR_LIMIT;
    P >>= 1;
  }
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
