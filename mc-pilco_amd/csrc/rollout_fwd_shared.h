// Shared by the two forward rollout kernels (rollout_fwd.hip: 1/2/4 particles per workgroup, VALU matvec;
// rollout_fwd_tile.hip: 16 particles per workgroup, matrix-core contractions): kernel arguments, the LDS copy of
// the GP descriptors, kernel hyper-parameter tables.
#pragma once
#include "rollout_common.h"

#ifndef RF_NT  // (a translation unit may be built with another workgroup size: the 16-particle kernel's 1024-thread experiment, -DRF_NT=1024)
#define RF_NT 512
#endif
#define RF_NW (RF_NT / 64)

namespace mcp {

// LDS-resident copy of what the kernels need from mcp_gp.  The descriptors arrive by value in the
// kernel argument; indexing that copy with a per-lane GP index would make the compiler spill the
// whole argument to scratch (global-latency loads in every phase), so it is staged here once.
typedef const double __attribute__((address_space(1))) * gptr_t;  // explicit global pointer: loads become global_load, not flat_load
typedef double v2d __attribute__((ext_vector_type(2)));            // native vector: loadable through an address-space pointer
typedef const v2d __attribute__((address_space(1))) * gptr2_t;
struct GpL {
  const double* Kinv;
  const double* Xt;
  const double* X;
  const double* alpha;
  double lambda, mean, var_scale;
  int N, Npad, deg, pad_;
};
#define GPL_DOUBLES ((int)(sizeof(GpL) / sizeof(double)))
// kernel hyper-parameters per GP in LDS: inv_ls[D] | w1[D+1] | w20[D] | w21[D] | aX[D]
#define KP_INVLS(D) 0
#define KP_W1(D) (D)
#define KP_W20(D) (2 * (D) + 1)
#define KP_W21(D) (3 * (D) + 1)
#define KP_AX(D) (4 * (D) + 1)
#define KP_STRIDE(D) (5 * (D) + 1)


struct FwdArgs {
  mcp_model model;
  mcp_policy pol;
  mcp_noise nz;
  int M, T, particle_pred;
  int NpadMax, maxdeg, GB, NCmax;
  const double* x0;
  double* states;
  double* inputs;
  double* jac;
  uint32_t* status;
  unsigned long long* stamps;  // diagnostic only (mcp_dispatch.fwd_stamps): per-phase cycle totals of workgroup `stamp_block`
  unsigned stamp_block;        // (mcp_dispatch.stamp_block; 0 by default)
  double* xj;                  // packed phase-J operand of the wide 16-particle classes (workspace; null: not available), see rollout_xj_bytes
  int xj_stride;               // doubles per GP
  const double* kt;            // Kinv as MFMA operand tiles (lean small-swarm kernel, rollout_fwd.hip; workspace; null: not available)
  int kt_stride;               // doubles per GP
  // GP-sharded launch (rollout_fwd.hip, GSH): the G workgroups of a particle cluster hand each other their GP's sampled
  // increment once per step through 8-byte {tag, value} granules  xch[cluster][t & 1][g][p][half]  (zeroed per launch)
  unsigned long long* xch;
  int nclusters;
  int gsh_cs;        // workgroups per cluster in the GP-sharded 16-particle kernel (each takes G / gsh_cs consecutive GPs)
  // GP-sharded 16-particle kernel, round 4: the policy's basis functions are SPLIT over the gsh_cs workgroups of a cluster instead of being
  // evaluated by each of them; the partial pre-squash sums W phi (16 particles x U) go round through granules
  //   uxch[cluster][t & 1][member][p * U + k][half]   (zeroed per launch; null: every member evaluates the whole policy)
  unsigned long long* uxch;
  // GP-sharded 16-particle kernel, round 5: gsh_rs = 2 puts TWO workgroups on every (tile, GP range), each with one half of the rows of Kinv
  // (phases V and J over its own rows; phase K, the small one, in both).  The half that does not finish the GP (half 0) sends its partial sums
  // of phase F -- two doubles per (particle, column c <= D) -- to the one that does (half 1 = gsh_rs - 1, which adds own + partner in that order):
  //   rxch[cluster][t & 1][g][sender][2 values][half][p * (D + 1) + c]   (granules, zeroed per launch; gsh_rs <= 1: unused)
  // Round 6: gsh_rs = 3 -- three row parts, two senders (the part with the last rows finishes and adds own + sender 0 + sender 1) -- where three
  // times the grid is still resident (the UR5 script's M = 200: 13 tiles x 6 GPs x 3 = 234 workgroups).  gsh_map = 1 deals the workgroups
  // ROW PART major: the ceil(n / 8) blocks of an XCD are consecutive items of the order (member, row part, tile) -- an XCD's L2 then holds the
  // rows of Kinv of two or three (GP, row part) pairs instead of every GP's (gsh_map = 0: all members of a tile on one XCD, round 5).
  int gsh_rs;
  int gsh_map;
  unsigned long long* rxch;
  int operands_packed;  // host side only: bit 0 -- `kt` holds this model's tiles already, bit 1 -- `xj` does (MCP_FWD_KT_PACKED / MCP_FWD_XJ_PACKED)
  int m_off, m_cnt;  // the particles [m_off, m_off + m_cnt) of the swarm that this launch covers (a swarm too large for one
                     // resident GP-sharded grid goes out as a few launches back to back on the stream)
};

#define RF_STAMP(k)                                 \
  do {                                              \
    if (a.stamps && tid == 0 && blockIdx.x == a.stamp_block) {  \
      unsigned long long now_ = clock64();          \
      a.stamps[k] += now_ - last_stamp;             \
      last_stamp = now_;                            \
    }                                               \
  } while (0)


// one-time staging of the GP descriptors and kernel hyper-parameters into LDS (uniform indices only)
__device__ __forceinline__ void stage_gp_tables(const mcp_gp* gps, const double* var_scale, int G, int D, GpL* gpl, double* kpar, int tid) {
  for (int g = 0; g < G; ++g) {
    const mcp_gp& gp = gps[g];
    if (tid == 0) {
      GpL e;
      e.Kinv = gp.Kinv;
      e.Xt = gp.Xt;
      e.X = gp.X;
      e.alpha = gp.alpha;
      e.lambda = kern_lambda(gp.kern);
      e.mean = kern_mean(gp.kern);
      e.var_scale = var_scale ? var_scale[g] : 1.0;
      e.N = gp.N;
      e.Npad = gp.Npad;
      e.deg = gp.kern.poly_deg;
      e.pad_ = 0;
      gpl[g] = e;
    }
    double* kp = kpar + (size_t)g * KP_STRIDE(D);
    const int deg = gp.kern.poly_deg;
    for (int it = tid; it < KP_STRIDE(D); it += RF_NT) {
      double v = 0.0;
      if (it < D)
        v = gp.kern.inv_ls[it];
      else if (it < 2 * D + 1)
        v = deg >= 1 ? gp.kern.w1[it - D] : 0.0;
      else if (it < 3 * D + 1)
        v = deg >= 2 ? gp.kern.w20[it - KP_W20(D)] : 0.0;
      else if (it < 4 * D + 1)
        v = deg >= 2 ? gp.kern.w21[it - KP_W21(D)] : 0.0;
      else
        v = deg >= 1 ? gp.aX[it - KP_AX(D)] : 0.0;
      kp[it] = v;
    }
  }
}


// hand-off of the GP-sharded launches (protocol: rollout_fwd.hip): granule store, slot of (cluster, step parity, GP, particle)
typedef unsigned long long __attribute__((address_space(1))) * gu64_t;
#define RF_SPIN_LIMIT (1u << 22)  // polls of ~1-2 us each: several seconds -- far beyond any delay a partner can have while the device makes progress
__device__ __forceinline__ void store_granule(gu64_t g, unsigned epoch, unsigned value) {
  __hip_atomic_store(g, ((unsigned long long)epoch << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ size_t xch_slot(int cluster, int t, int G, int g, int P) { return ((((size_t)cluster * 2 + (t & 1)) * G + g) * P) * 2; }

// Trajectory outputs of the 16-particle kernel (states, inputs, d delta/dz: up to 0.9 GB per launch at the UR5 shape) are written once and
// never read back by the launch.  Stored plainly they are write-allocated in the XCD's L2 and push out the Kinv the launch re-reads every
// step (C5: 3 x 1.28 MB per XCD of a 4 MB L2 -- FETCH_SIZE 2.1 GB per launch, 22 x the algorithmic bytes).  An agent-scope relaxed store
// (`global_store ... sc1`) goes through and DROPS the line (MI355X_MICROARCH.md, "stores of each flavour").
__device__ __forceinline__ void store_through(double* p, double v) {
#ifdef TLX_SC1_STORES
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *p = v;
#endif
}

typedef double v4d_t __attribute__((ext_vector_type(4)));
// sum over the 4 lanes l, l^16, l^32, l^48 (the 4 feature groups of an MFMA operand column)
__device__ __forceinline__ double fold_kk(double v) {
  return sum_xor32(sum_xor16(v));
}

// forward rollout with 16 particles per workgroup (rollout_fwd_tile.hip); MCP_ERR_LIMIT when the problem does not fit it
int launch_fwd_tile(const FwdArgs& a, hipStream_t st);
// the same kernel GP-sharded (G workgroups per 16-particle tile; a.xch / a.nclusters set by the caller); MCP_ERR_LIMIT when
// the shape has no sharded instantiation
int launch_fwd_tile_sharded(const FwdArgs& a, hipStream_t st);
bool fwd_tile_fits(const mcp_model* model, const mcp_policy* policy);
// the latency-lean GP-sharded kernel of small swarms (rollout_fwd_lean.hip): dynamic LDS it needs for this shape at P particles per
// workgroup (0: it does not take the shape), the packing of Kinv into its operand tiles (a.kt), the launch itself
size_t fwd_lean_lds_bytes(const mcp_model* model, const mcp_policy* policy, int P, int NpadMax, int maxdeg);
int launch_fwd_lean_pack(const FwdArgs& a, hipStream_t st);
int launch_fwd_lean(const FwdArgs& a, int P, size_t lds, hipStream_t st);

}  // namespace mcp
