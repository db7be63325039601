/*
 * mcpilco_hip_debug.h -- test and diagnostic hooks of libmcpilco_hip.so.  NOT part of the drop-in
 * boundary (include/mcpilco_hip.h): the product path never needs them.  They exist so that the
 * parity tests can force every kernel variant the automatic dispatch of mcp_rollout_fwd /
 * mcp_rollout_bwd may choose, so that bench.py / tools can report which variant ran, and so that
 * tools/phase_stamps.py can read per-phase cycle counters.  Process-wide settings, held in atomics (round 5:
 * no data race when another thread launches meanwhile -- PyTorch runs the adjoint sweep on its autograd thread, so a thread-local setting would not
 * reach it); they are meant for a test driver that forces ONE variant at a time, not for concurrent use with different settings.
 */
#ifndef MCPILCO_HIP_DEBUG_H
#define MCPILCO_HIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* forward rollout: particles per workgroup 1 / 2 / 4 (small-tile kernel) or 16 (matrix-core tile kernel); 0 = automatic */
void mcp_debug_set_particles_per_wg(int p);
/* what the last mcp_rollout_fwd / mcp_posterior_fwd launched (16 = tile kernel) */
int mcp_debug_last_particles_per_wg(void);
/* GP-sharded launch forms: -1 automatic, 0 never, 1 whenever the grid fits the device */
void mcp_debug_set_gp_sharding(int mode);
/* the policy of the GP-sharded 16-particle kernel: -1 automatic, 0 every member of a cluster evaluates all of it, 1 split over the members
 * whenever the shape allows it (partial sums W phi exchanged per step) */
void mcp_debug_set_policy_split(int mode);
/* number of GP-sharded launches the last forward call made (0 = unsharded) */
int mcp_debug_last_gp_sharded(void);
/* the latency-lean GP-sharded kernel of narrow SE-only models (rollout_fwd_lat_kernel): -1 / 1 wherever it applies, 0 never */
void mcp_debug_set_fwd_lean(int mode);
/* 1 when the last mcp_rollout_fwd ran that kernel */
int mcp_debug_last_fwd_lean(void);
/* small-tile kernel: xlds -1 automatic / 0 never stage the small operands in LDS; gb = GPs per pass (0 = as many as fit) */
void mcp_debug_set_fwd_mode(int xlds, int gb);
/* Cholesky / triangular inverse: 1 (default) the round-4 MFMA kernels (left-looking factorisation, four-wave inverse columns), 3 the same with
 * one wave per inverse column, 2 the round-3 forms (right-looking factorisation, block-diagonal sweep of the inverse), 0 the round-1/2 forms */
void mcp_debug_set_chol_mfma(int on);
/* backward sweep: particles per workgroup 1 / 2 / 4 (wide 512-thread class: also 8); 0 = automatic */
void mcp_debug_set_bwd_particles(int pb);
/* backward sweep of small swarms: -1 (default) the latency-lean kernel where it applies (automatic particle count only), 0 never;
   1 when the last mcp_rollout_bwd ran it */
void mcp_debug_set_bwd_lean(int mode);
int mcp_debug_last_bwd_lean(void);
/* device buffers of per-phase cycle totals (forward: 32 uint64, backward: 16) of one workgroup (NULL = off); the forward kernels stamp workgroup
   `block` (0 by default; the partner of workgroup 0 in a 2-way GP-sharded launch of the small-tile kernel is workgroup 8) */
void mcp_debug_set_stamp_buffer(void* device_u64x32); /* 32 uint64: every forward kernel writes per-phase totals to slots 0..15 and per-wave
                                                         phase totals to slots 16..31 */
void mcp_debug_set_stamp_block(int block);
void mcp_debug_set_bwd_stamp_buffer(void* device_u64x16);

#ifdef __cplusplus
}
#endif
#endif /* MCPILCO_HIP_DEBUG_H */
