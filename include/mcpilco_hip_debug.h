/*
 * mcpilco_hip_debug.h -- test and diagnostic entry points of libmcpilco_hip.so.  NOT part of the drop-in boundary
 * (include/mcpilco_hip.h): the product path never needs them.  They exist so that the parity tests can force every kernel variant the
 * automatic dispatch of mcp_rollout_fwd / mcp_rollout_bwd / mcp_chol_* may choose, so that bench.py / tools can report which variant ran,
 * and so that tools/phase_stamps.py can read per-phase cycle counters.
 *
 * Round 5: the request travels WITH THE CALL.  Every `_ex` entry point is its plain namesake plus a `mcp_dispatch*` (NULL or all zero =
 * automatic: the plain entry points pass NULL); the library keeps no dispatch state of its own -- no setters, nothing process-wide.
 */
#ifndef MCPILCO_HIP_DEBUG_H
#define MCPILCO_HIP_DEBUG_H

#include "mcpilco_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcp_dispatch {
  /* ---- requests (0 = automatic) ---- */
  int32_t fwd_particles; /* forward: particles per workgroup 1 / 2 / 4 (small-tile kernels) or 16 (matrix-core tile kernel)                 */
  int32_t gp_sharding;   /* GP-sharded launch forms: 1 never, 2 whenever the grid fits the device                                        */
  int32_t fwd_lean;      /* the latency-lean GP-sharded kernel (rollout_fwd_lat_kernel): 1 never                                        */
  int32_t policy_split;  /* GP-sharded 16-particle kernel: 1 every member evaluates the whole policy, 2 split whenever the shape allows */
  int32_t row_split;     /* GP-sharded 16-particle kernel: 1 one workgroup per (tile, GP range), 2 / 3 that many (row parts of Kinv) whenever allowed */
  int32_t cluster_map;   /* ... its row-split form: 1 the workgroups of a tile on one XCD, 2 dealt row part major                        */
  int32_t fwd_no_xlds;   /* small-tile kernel: 1 never stage the small operands in LDS                                                  */
  int32_t fwd_gb;        /* small-tile kernel: GPs per pass (0 = as many as fit)                                                        */
  int32_t bwd_particles; /* backward sweep: particles per workgroup 1 / 2 / 4 / 8 (forces the general sweep)                            */
  int32_t bwd_lean;      /* the latency-lean sweep (rollout_bwd_lat_kernel): 1 never                                                    */
  int32_t bwd_pipe;      /* general sweep, one particle per workgroup on the wide classes: 1 never the pipelined form (chain beside the RBF stage) */
  int32_t chol_form;     /* mcp_chol_factor / _inverse: 1 the round-1/2 kernels, 2 the round-3 one-workgroup forms, 3 the round-4 forms
                            with one-wave inverse columns (0: left-looking / panel factorisation, column-parallel / blocked inverse)     */
  uint32_t stamp_block;  /* which workgroup of the forward launch writes its stamps                                                     */
  void* fwd_stamps;      /* device buffer of 32 uint64 per-phase cycle totals of that workgroup (NULL = off)                            */
  void* bwd_stamps;      /* device buffer of 16 uint64 (backward sweep)                                                                 */
  /* ---- report (written by the call) ---- */
  int32_t ran_particles;  /* forward / posterior: particles per workgroup launched (16 = tile kernel) */
  int32_t ran_gp_sharded; /* number of GP-sharded launches the forward call made (0 = unsharded)      */
  int32_t ran_fwd_lean;   /* 1: the lean forward kernel ran                                           */
  int32_t ran_bwd_lean;   /* 1: the lean backward sweep ran                                           */
  int32_t ran_row_split;  /* 2 / 3: the forward launch put that many workgroups on every (tile, GP range) */
  int32_t ran_bwd_pipe;   /* 1: the general sweep ran in its pipelined form                              */
} mcp_dispatch;

int mcp_rollout_fwd_ex(const mcp_model* model, const mcp_policy* policy, const mcp_noise* noise, int M, int T, int particle_pred,
                       const double* x0, double* states, double* inputs, double* jac, uint32_t* status, void* workspace,
                       size_t workspace_bytes, void* stream, mcp_dispatch* d);
int mcp_rollout_bwd_ex(const mcp_model* model, const mcp_policy* policy, const mcp_noise* noise, int M, int T, const double* states,
                       const double* inputs, const double* jac, const double* g_states, const double* g_inputs, double* g_log_ls,
                       double* g_centers, double* g_weight, double* g_x0, void* workspace, size_t workspace_bytes, void* stream,
                       mcp_dispatch* d);
int mcp_posterior_fwd_ex(const mcp_gp* gp, int M, const double* Z, double* mu, double* var, double* Jmu, double* Jvar, uint32_t* status,
                         void* stream, mcp_dispatch* d);
int mcp_chol_factor_ex(int N, double* A, int lda, double* logdet, uint32_t* status, void* stream, const mcp_dispatch* d);
int mcp_chol_inverse_ex(int N, const double* U, int ldu, double* Uinv, int ldi, double* Kinv, int ldk, void* stream, const mcp_dispatch* d);

#ifdef __cplusplus
}
#endif
#endif /* MCPILCO_HIP_DEBUG_H */
