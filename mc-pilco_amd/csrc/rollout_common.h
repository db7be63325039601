// Shared by rollout_fwd.hip and rollout_bwd.hip: feature maps of the dynamics model and of the
// policy, descriptor validation, small integer helpers.
#pragma once
#include <string.h>

#include "mcp_device.h"

#define MCP_LDS_LIMIT (160 * 1024)

__host__ __device__ inline int imax(int a, int b) { return a > b ? a : b; }
__host__ __device__ inline int imin(int a, int b) { return a < b ? a : b; }

// GP input z = [x[not_angle], sin x[angle], cos x[angle], u]   (Model_learning.py:670-683)
// policy feature s = [x_nonangle, COS, SIN] (Policy.py:326-333) or [x, x*_t - x] (:397-399)
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

// workgroup barrier that orders LDS traffic only: global stores issued earlier are fire-and-forget in
// the rollout kernels (nobody in the kernel reads them back), so they are not drained at every barrier
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// hand-off buffer of the GP-sharded forward launch (rollout_fwd.hip): [clusters][2][G][P][2] granules of 8 bytes, clusters * P < M + 16 (a launch per chunk, each rounded up to whole clusters)
static inline size_t rollout_xch_bytes(int M, int G) { return ((size_t)(M + 16) * 2 * G * 2 * sizeof(unsigned long long) + 15) & ~(size_t)15; }
// partial policy sums of the GP-sharded 16-particle kernel: clusters x 2 (step parity) x members (<= G) x 16 particles x U x 2 granules
static inline size_t rollout_uxch_bytes(int M, int G, int U) { return ((size_t)(M + 16) * 2 * G * U * 2 * sizeof(unsigned long long) + 15) & ~(size_t)15; }

// Packed phase-J operands of the 16-particle kernel's wide classes (D + 1 > 16; rollout_fwd_tile.hip, tile_xj_pack_kernel): per GP two
// variants ([X^T; 1] and its columns scaled by alpha_j) x two row tiles x (NpadMax / 8) x 64 lanes x 2 doubles, placed behind the hand-off
// granules in the caller's workspace.  0 for narrow models.
static inline size_t rollout_xj_bytes(const mcp_model* m) {
  if (!m || m->D + 1 <= 16 || m->D + 1 > 32) return 0;
  int npad = 0;
  for (int g = 0; g < m->G; ++g) npad = m->gp[g].Npad > npad ? m->gp[g].Npad : npad;
  return (size_t)m->G * 2 * 2 * (size_t)((npad + 7) / 8) * 64 * 2 * sizeof(double);
}

// Kinv as MFMA operand tiles for the lean small-swarm kernel (rollout_fwd.hip, kt_pack_kernel): G x NpadMax^2 doubles behind the
// packed phase-J operands in the caller's workspace, rebuilt by every call.  0 for models that kernel does not take.
#define RL_MAXD_ 8
static inline size_t rollout_kt_bytes(const mcp_model* m) {
  if (!m || m->D > RL_MAXD_ || m->G < 2) return 0;
  int npad = 0;
  for (int g = 0; g < m->G; ++g) npad = m->gp[g].Npad > npad ? m->gp[g].Npad : npad;
  if (npad > 640) return 0;
  return sizeof(double) * ((size_t)m->G * npad * npad + 6 * 128);  // (+ one register buffer of slack: the last stream may be read past its end)
}

// partial phase-F sums of the row-split cluster (FwdArgs.rxch): clusters x 2 (step parity) x G x 2 senders x 16 particles x (D + 1) columns x 2 values x
// 2 granules, last in the caller's workspace; wide models on swarms that can be resident at two workgroups per (tile, GP) only
static inline size_t rollout_rxch_bytes(const mcp_model* m, int M) {
  if (!m || rollout_xj_bytes(m) == 0 || M > 1024) return 0;
  return (size_t)((M + 15) / 16) * 2 * (size_t)m->G * 2 * 16 * (size_t)(m->D + 1) * 4 * sizeof(unsigned long long);  // (two senders: gsh_rs = 3)
}

static inline bool model_ok(const mcp_model* m) {
  if (!m) return false;
  if (m->S <= 0 || m->S > MCP_MAX_STATE || m->U <= 0 || m->U > MCP_MAX_INPUT || m->G <= 0 || m->G > MCP_MAX_GP) return false;
  if (m->D <= 0 || m->D > MCP_MAX_GPDIM) return false;
  if (m->n_angle < 0 || m->n_not_angle < 0 || m->n_not_angle + 2 * m->n_angle + m->U != m->D) return false;
  for (int i = 0; i < m->n_angle; ++i)
    if (m->angle[i] < 0 || m->angle[i] >= m->S) return false;
  for (int i = 0; i < m->n_not_angle; ++i)
    if (m->not_angle[i] < 0 || m->not_angle[i] >= m->S) return false;
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

// policy-only evaluation (T == 1, no dynamics model): a stub model with no GPs
static inline mcp_model policy_only_model(const mcp_policy* p) {
  mcp_model m;
  memset(&m, 0, sizeof(m));
  m.S = p->S;
  m.U = p->U;
  m.G = 0;
  m.D = p->U;  // z = [u]
  m.Ts = 0.0;
  return m;
}

static inline bool policy_ok(const mcp_policy* p, int S, int U, int T) {
  if (!p || p->S != S || p->U != U) return false;
  if (p->B <= 0 || p->B > MCP_MAX_BASIS || p->P <= 0 || p->P > MCP_MAX_PFEAT) return false;
  if (!p->log_ls || !p->centers || !p->weight || !p->u_max) return false;
  if (!(p->p_drop >= 0.0 && p->p_drop < 1.0)) return false;
  if (p->meas.n < 0 || 2 * p->meas.n > S) return false;
  for (int i = 0; i < p->meas.n; ++i)
    if (p->meas.pos[i] < 0 || p->meas.pos[i] >= S || p->meas.vel[i] < 0 || p->meas.vel[i] >= S || p->meas.pos[i] == p->meas.vel[i]) return false;
  if (p->meas.n > 0 && !(p->meas.a0 != 0.0)) return false;
  if (p->kind == MCP_POLICY_PLAIN) return p->P == S;
  if (p->kind == MCP_POLICY_ANGLES) {
    if (p->n_non_angle + 2 * p->n_angle != p->P) return false;
    for (int i = 0; i < p->n_angle; ++i)
      if (p->angle[i] < 0 || p->angle[i] >= S) return false;
    for (int i = 0; i < p->n_non_angle; ++i)
      if (p->non_angle[i] < 0 || p->non_angle[i] >= S) return false;
    return true;
  }
  if (p->kind == MCP_POLICY_TRAJ) return p->P == 2 * S && p->target_traj && p->traj_len >= T;
  return false;
}
