// Expected-cost kernels for gfx950: Expected_cost.forward (policy_learning/Cost_function.py:25-36)
// with cart_pole_cost (:170-182) or saturated_distance_from_trajectory (:124-147), and their
// state gradient.  HBM-bound elementwise + per-time-step reductions over the particle axis.
#include "mcp_device.h"

using namespace mcp;

__device__ __forceinline__ double cost_point(const mcp_cost& c, const double* x, int t, double* dist_out) {
  double dist = 0.0;
  if (c.kind == MCP_COST_CARTPOLE) {
    double a = (fabs(x[c.angle_index]) - c.target_angle) / c.ls_angle;
    double b = (x[c.pos_index] - c.target_pos) / c.ls_pos;
    dist = a * a + b * b;
  } else {
    for (int i = 0; i < c.n_used; ++i) {
      int s = c.used[i];
      double r = (x[s] - c.target_traj[(size_t)t * c.S + s]) / c.lengthscales[i];
      dist = fma(r, r, dist);
    }
  }
  if (dist_out) *dist_out = dist;
  return 1.0 - exp(-dist);
}

// one workgroup per time step: costs[t][:], then mean and centred sum of squares (two passes,
// like torch.mean / torch.std)
__global__ __launch_bounds__(256) void cost_fwd_kernel(mcp_cost c, int T, int M, const double* __restrict__ states,
                                                       double* __restrict__ costs, double* __restrict__ moments,
                                                       uint32_t* __restrict__ status) {
  __shared__ double red[4];
  __shared__ double mean_s;
  const int t = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  double s = 0.0;
  uint32_t bad = 0;
  for (int m = tid; m < M; m += 256) {
    double v = cost_point(c, states + ((size_t)t * M + m) * c.S, t, nullptr);
    costs[(size_t)t * M + m] = v;
    if (is_bad(v)) bad |= MCP_STATUS_NAN;
    s += v;
  }
  s = wave_sum(s);
  if (lane == 0) red[wv] = s;
  __syncthreads();
  if (tid == 0) mean_s = ((red[0] + red[1]) + (red[2] + red[3])) / (double)M;
  __syncthreads();
  const double mean = mean_s;
  double q = 0.0;
  for (int m = tid; m < M; m += 256) {
    double d = costs[(size_t)t * M + m] - mean;
    q = fma(d, d, q);
  }
  q = wave_sum(q);
  __syncthreads();
  if (lane == 0) red[wv] = q;
  __syncthreads();
  if (tid == 0) {
    moments[2 * t] = mean;
    moments[2 * t + 1] = (red[0] + red[1]) + (red[2] + red[3]);
  }
  if (bad) atomicOr(status, bad);
}

// pooled mean / unbiased std over R ranks (Chan et al. parallel-variance combine), then the sums
// over time.  One workgroup.
struct RankCounts {
  int64_t n[64];
};
__global__ __launch_bounds__(256) void cost_finalize_kernel(int T, int R, const double* __restrict__ moments, RankCounts rc,
                                                            double* __restrict__ out) {
  const int64_t* counts = rc.n;
  __shared__ double red[2][4];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  double n_tot = 0.0;
  for (int r = 0; r < R; ++r) n_tot += (double)counts[r];
  double cs = 0.0, ss = 0.0;
  for (int t = tid; t < T; t += 256) {
    double mean = 0.0;
    for (int r = 0; r < R; ++r) mean += (double)counts[r] * moments[((size_t)r * T + t) * 2];
    mean /= n_tot;
    double m2 = 0.0;
    for (int r = 0; r < R; ++r) {
      double d = moments[((size_t)r * T + t) * 2] - mean;
      m2 += moments[((size_t)r * T + t) * 2 + 1] + (double)counts[r] * d * d;
    }
    cs += mean;
    ss += sqrt(m2 / (n_tot - 1.0));
  }
  cs = wave_sum(cs);
  ss = wave_sum(ss);
  if (lane == 0) {
    red[0][wv] = cs;
    red[1][wv] = ss;
  }
  __syncthreads();
  if (tid == 0) {
    out[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    out[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// Summable form of one rank's moments, for the single all-reduce of a particle-sharded step:
//   sums[t] = sum_m (c - shift_t),  sums[T + t] = sum_m (c - shift_t)^2   over this rank's M particles,
// from the two-pass {mean, centred sum of squares} cost_fwd_kernel produced.  shift (same on every rank; NULL = 0) keeps
// the pooled variance free of cancellation: the caller passes the previous step's pooled means.
__global__ void cost_sums_kernel(int T, double n, const double* __restrict__ moments, const double* __restrict__ shift,
                                 double* __restrict__ sums) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const double d = moments[2 * t] - (shift ? shift[t] : 0.0);
  sums[t] = n * d;
  sums[T + t] = fma(n * d, d, moments[2 * t + 1]);
}

// all ranks' sums added up (n_total particles) -> out[0] = sum_t mean, out[1] = sum_t unbiased std; mean_out[t] (optional)
// receives the pooled mean per time step (the next step's shift).  One workgroup.
__global__ __launch_bounds__(256) void cost_finalize_sums_kernel(int T, double n_tot, const double* __restrict__ sums,
                                                                 const double* __restrict__ shift, double* __restrict__ out,
                                                                 double* __restrict__ mean_out) {
  __shared__ double red[2][4];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  double cs = 0.0, ss = 0.0;
  for (int t = tid; t < T; t += 256) {
    const double a = sums[t], b = sums[T + t];
    const double mean = (shift ? shift[t] : 0.0) + a / n_tot;
    const double m2 = b - a * a / n_tot;
    cs += mean;
    ss += sqrt(fmax(m2, 0.0) / (n_tot - 1.0));
    if (mean_out) mean_out[t] = mean;
  }
  cs = wave_sum(cs);
  ss = wave_sum(ss);
  if (lane == 0) {
    red[0][wv] = cs;
    red[1][wv] = ss;
  }
  __syncthreads();
  if (tid == 0) {
    out[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    out[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

__global__ void cost_bwd_kernel(mcp_cost c, int T, int M, const double* __restrict__ states, const double* __restrict__ g_cost,
                                double gscale, double* __restrict__ g_states) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)T * M) return;
  int t = (int)(i / M);
  const double* x = states + i * c.S;
  double* g = g_states + i * c.S;
  double dist;
  cost_point(c, x, t, &dist);
  double e = (g_cost ? *g_cost : 1.0) * gscale * exp(-dist);  // d c / d dist = exp(-dist)
  for (int s = 0; s < c.S; ++s) g[s] = 0.0;
  if (c.kind == MCP_COST_CARTPOLE) {
    double th = x[c.angle_index];
    double a = (fabs(th) - c.target_angle) / c.ls_angle;
    double b = (x[c.pos_index] - c.target_pos) / c.ls_pos;
    double sg = th > 0.0 ? 1.0 : (th < 0.0 ? -1.0 : 0.0);  // d|theta|/dtheta, 0 at 0 like torch.abs
    g[c.angle_index] += e * 2.0 * a * sg / c.ls_angle;
    g[c.pos_index] += e * 2.0 * b / c.ls_pos;
  } else {
    for (int k = 0; k < c.n_used; ++k) {
      int s = c.used[k];
      double r = (x[s] - c.target_traj[(size_t)t * c.S + s]) / c.lengthscales[k];
      g[s] += e * 2.0 * r / c.lengthscales[k];
    }
  }
}

static bool cost_ok(const mcp_cost* c) {
  if (!c || c->S <= 0 || c->S > MCP_MAX_STATE) return false;
  if (c->kind == MCP_COST_CARTPOLE)
    return c->angle_index >= 0 && c->angle_index < c->S && c->pos_index >= 0 && c->pos_index < c->S && c->ls_angle != 0.0 &&
           c->ls_pos != 0.0;
  if (c->kind == MCP_COST_TRAJ) {
    if (c->n_used <= 0 || c->n_used > MCP_MAX_STATE || !c->target_traj || !c->lengthscales) return false;
    for (int i = 0; i < c->n_used; ++i)
      if (c->used[i] < 0 || c->used[i] >= c->S) return false;
    return true;
  }
  return false;
}

extern "C" int mcp_cost_fwd(const mcp_cost* cost, int T, int M, const double* states, double* costs, double* moments, uint32_t* status,
                            void* stream) {
  if (!cost_ok(cost) || !states || !costs || !moments || !status || T <= 0 || M <= 0) return MCP_ERR_ARG;
  hipLaunchKernelGGL(cost_fwd_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, *cost, T, M, states, costs, moments, status);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

extern "C" int mcp_cost_finalize(int T, int R, const double* moments, const int64_t* counts, double* out, void* stream) {
  if (!moments || !counts || !out || T <= 0 || R <= 0) return MCP_ERR_ARG;
  if (R > 64) return MCP_ERR_LIMIT;
  RankCounts rc;
  for (int r = 0; r < 64; ++r) rc.n[r] = r < R ? counts[r] : 0;
  hipLaunchKernelGGL(cost_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, T, R, moments, rc, out);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

extern "C" int mcp_cost_sums(int T, int M, const double* moments, const double* shift, double* sums, void* stream) {
  if (!moments || !sums || T <= 0 || M <= 0) return MCP_ERR_ARG;
  hipLaunchKernelGGL(cost_sums_kernel, dim3((T + 255) / 256), dim3(256), 0, (hipStream_t)stream, T, (double)M, moments, shift, sums);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

extern "C" int mcp_cost_finalize_sums(int T, int64_t n_total, const double* sums, const double* shift, double* out, double* mean_out,
                                      void* stream) {
  if (!sums || !out || T <= 0 || n_total <= 0) return MCP_ERR_ARG;
  hipLaunchKernelGGL(cost_finalize_sums_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, T, (double)n_total, sums, shift, out, mean_out);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

extern "C" int mcp_cost_bwd(const mcp_cost* cost, int T, int M, const double* states, const double* g_cost, double gscale,
                            double* g_states, void* stream) {
  if (!cost_ok(cost) || !states || !g_states || T <= 0 || M <= 0) return MCP_ERR_ARG;
  size_t n = (size_t)T * M;
  hipLaunchKernelGGL(cost_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *cost, T, M, states, g_cost,
                     gscale, g_states);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

extern "C" int mcp_abi_version(void) { return MCP_ABI_VERSION; }
extern "C" const char* mcp_build_info(void) { return "libmcpilco_hip gfx950 fp64 (" __DATE__ " " __TIME__ ")"; }
