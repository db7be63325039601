/*
 * mcpilco_hip.h -- C ABI of libmcpilco_hip.so (gfx950 / MI355X).
 *
 * The reference (merlresearch/MC-PILCO) is pure Python on PyTorch and has no FFI of its own;
 * its extension mechanism is constructor injection of Python classes.  This header is the
 * boundary a maintainer would bind with ctypes underneath those classes:
 * every entry point names the reference function(s) it replaces as file:line relative to
 * the reference root.
 *
 * Conventions
 *   - all matrices float64, row-major, contiguous; every pointer is a DEVICE pointer unless
 *     the parameter is a `const mcp_*` descriptor struct (host memory, copied at launch);
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it; nothing here
 *     allocates, frees or synchronises (graph-capture safe) -- scratch comes from the caller
 *     through `workspace` (size from the matching *_workspace_bytes query);
 *   - return value: MCP_OK (0) or a negative MCP_ERR_* for arguments the kernels cannot take;
 *     never throws.  Numerical trouble is data, not an error: kernels OR bit flags into the
 *     device word `status` (MCP_STATUS_*), which the host may read after synchronising
 *     (reference behaviour: NaN cost -> retry, policy_learning/MC_PILCO.py:479-501,573-607);
 *   - re-entrant per stream; the caller owns every buffer.
 */
#ifndef MCPILCO_HIP_H
#define MCPILCO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCP_ABI_VERSION 6 /* 5: mcp_kernel.scal, MCP_FWD_NO_GP_SHARDING, MCP_STATUS_NONPOS_VAR now means a FINITE variance <= 0, mcp_nll_epoch, \
                             mcp_adam_step_guarded, mcp_policy_step_commit (round 4).                                                            \
                             6 (round 6; the round-5 contract changes, which had kept the number 5, are part of it): mcp_sod_select needs       \
                             mcp_sod_workspace_bytes(N) = 8 (N^2 + 2 N) + the exchange area (was 8 N^2) and may report *n_out = -1;             \
                             MCP_MAX_TRAIN 1024 -> 4096; the process-wide mcp_debug_* setters are gone (mcpilco_hip_debug.h: per-call           \
                             mcp_dispatch); mcp_adam_step_guarded skips a not-SPD epoch; new: mcp_sym_sandwich, mcp_noise.call_dev, MCP_FWD_KT_PACKED / MCP_FWD_XJ_PACKED */

#define MCP_OK 0
#define MCP_ERR_ARG (-1)       /* null pointer / non-positive size                       */
#define MCP_ERR_LIMIT (-2)     /* a dimension exceeds a compiled limit (MCP_MAX_*)        */
#define MCP_ERR_WORKSPACE (-3) /* workspace too small                                     */
#define MCP_ERR_LAUNCH (-4)    /* hipLaunchKernel reported an error                       */
#define MCP_ERR_COMM (-5)      /* RCCL is not loadable / no communicator / a collective failed */

#define MCP_STATUS_NAN 1u         /* a NaN was produced in a state / input / cost          */
#define MCP_STATUS_NONPOS_VAR 2u  /* a FINITE GP posterior variance <= 0 (torch Normal raises ValueError); a NaN variance sets MCP_STATUS_NAN only */
#define MCP_STATUS_NOT_SPD 4u     /* Cholesky met a non-positive pivot                     */
#define MCP_STATUS_SYNC 8u        /* GP-sharded rollout: a partner workgroup never arrived   */

#define MCP_MAX_GP 8
#define MCP_MAX_STATE 16
#define MCP_MAX_INPUT 8
#define MCP_MAX_GPDIM 32   /* D  : GP input dimension                                   */
#define MCP_MAX_PFEAT 32   /* P  : policy feature dimension                             */
#define MCP_MAX_BASIS 1024 /* B  : policy basis functions                               */
#define MCP_MAX_TRAIN 4096 /* N  : training points kept per GP (fused rollout kernels; round 5: was 1024) */

/* Kernel hyper-parameters of one GP: squared-exponential (+ Volterra polynomial of degree
 * 0, 1 or 2) -- gpr_lib/GP_prior/Stationary_GP.py:112-181 (RBF), gpr_lib/GP_prior/Sparse_GP.py:
 * 559-737 (MPK_GP, get_Volterra_MPK_GP), gpr_lib/GP_prior/GP_prior.py:299-347 (Sum_Independent_GP).
 * k(a,b) = lambda * exp(-sum_d ((a_d-b_d)/l_d)^2)                     (no factor 1/2)
 *        + [deg>=1] sum_d w1_d a_d b_d + w1_D                         (MPK_1, offset feature)
 *        + [deg==2] (sum_d w20_d a_d b_d) * (sum_d w21_d a_d b_d)     (MPK_2)
 * where w = s^2 and s_d=(k-d)*exp(par) as the reference's get_Sigma builds it (host side). */
typedef struct mcp_kernel {
  int32_t D;
  int32_t poly_deg;       /* 0, 1, 2                                                   */
  double lambda;          /* exp(log_lambda_par)                                       */
  double sigma_n2;        /* exp(sigma_n_log)^2 + sigma_n_num^2 (GP_prior.py:87-89)    */
  double mean;            /* constant prior mean (RBF.mean_par)                        */
  const double* inv_ls;   /* [D]      1/lengthscale                                    */
  const double* w1;       /* [D+1]    MPK_1 weights s^2 (NULL when poly_deg==0)        */
  const double* w20;      /* [D]      MPK_2 first-factor weights  (NULL unless deg==2) */
  const double* w21;      /* [D]      MPK_2 second-factor weights                      */
  const double* scal;     /* optional, device: [lambda, sigma_n2, mean] -- when non-NULL these override the three by-value
                             fields above (GP training keeps its hyper-parameters on the device: GP_prior.fit_model's epoch
                             then needs no device->host round trip to fill this descriptor)                              */
} mcp_kernel;

/* One pretrained GP = what Model_learning.pretrain_gp caches (model_learning/Model_learning.py:
 * 163-208: gp_inputs_tr_list, alpha_list, K_X_inv_list) in the kernels' layout.            */
typedef struct mcp_gp {
  mcp_kernel kern;
  int32_t N;              /* training points kept (all, or the SOD subset)             */
  int32_t Npad;           /* row pitch of X^T / Kinv, multiple of 16, >= N             */
  const double* Xt;       /* [D][Npad]    training inputs, transposed, zero padded     */
  const double* X;        /* [Npad][D]    same, row-major, zero padded                 */
  const double* alpha;    /* [Npad]       K^-1 (Y - m), zero padded                    */
  const double* Kinv;     /* [Npad][Npad] (K + sigma_n^2 I)^-1, symmetric, zero padded */
  const double* aX;       /* [D]          sum_j alpha_j X_jd (used when poly_deg>=1)   */
} mcp_gp;

/* Speed-integration dynamics model -- Speed_Model_learning_RBF(_MPK)_angle_state,
 * model_learning/Model_learning.py:619-760.  GP g predicts the change of state vel[g];
 * not_vel[g] is the matching position.  GP input z=[x[not_angle], sin x[angle], cos x[angle], u]. */
typedef struct mcp_model {
  int32_t S, U, G, D;
  int32_t n_angle, n_not_angle;
  int32_t angle[MCP_MAX_STATE];
  int32_t not_angle[MCP_MAX_STATE];
  int32_t vel[MCP_MAX_GP];
  int32_t not_vel[MCP_MAX_GP];
  double Ts;
  double var_scale[MCP_MAX_GP]; /* norm_list[g]^2 (Model_learning.py:220-221), 1 by default */
  mcp_gp gp[MCP_MAX_GP];
} mcp_model;

#define MCP_POLICY_PLAIN 0  /* Sum_of_gaussians                        policy_learning/Policy.py:153-265 */
#define MCP_POLICY_ANGLES 1 /* Sum_of_gaussians_with_angles            Policy.py:268-335  s=[x_na,cos,sin] */
#define MCP_POLICY_TRAJ 2   /* Sum_of_gaussians_with_target_trajectory Policy.py:338-403  s=[x, x*_t-x]    */

/* Measurement model of MC_PILCO4PMS.apply_policy (policy_learning/MC_PILCO.py:808-906): the particles evolve on their true
 * states, the policy is evaluated on a simulated measurement -- positions plus Gaussian noise (:881-885), velocities by
 * backward difference of the noisy positions (:888-891) passed through the first-order filter (b0 nv_t + b1 nv_{t-1} -
 * a1 mv_{t-1}) / a0 (:895-899); at t = 0 the measurement is the true state (:856).  n == 0: the policy sees the true state
 * (MC_PILCO.apply_policy). */
typedef struct mcp_meas {
  int32_t n;                      /* (position, velocity) pairs                                        */
  int32_t pos[MCP_MAX_STATE];     /* pos_indeces                                                       */
  int32_t vel[MCP_MAX_STATE];     /* vel_indeces                                                       */
  double std_pos[MCP_MAX_STATE];  /* std of the position measurement noise, std_meas_noise_sim[pos]    */
  double b0, b1, a0, a1;          /* scipy.signal.butter(1, fc)                                        */
  const double* pos_noise;        /* [T-1][M][n] standard normals (step t at row t-1), or NULL: Philox  */
  double* meas;                   /* [T][M][S] measured states: written by mcp_rollout_fwd, read by
                                     mcp_rollout_bwd (caller-owned, required when n > 0)               */
} mcp_meas;

typedef struct mcp_policy {
  int32_t kind;
  int32_t S;              /* state dim of the system                                   */
  int32_t P;              /* feature dim (state_dim of the RBF network)                */
  int32_t B, U;
  int32_t squash;         /* flg_squash                                                */
  int32_t n_angle, n_non_angle;
  int32_t angle[MCP_MAX_STATE];
  int32_t non_angle[MCP_MAX_STATE];
  int32_t traj_len;       /* rows of target_traj                                       */
  double p_drop;          /* dropout probability (0 -> no mask, no draw)               */
  const double* log_ls;   /* [P]     log_lengthscales                                  */
  const double* centers;  /* [B][P]                                                    */
  const double* weight;   /* [U][B]  f_linear.weight (no bias)                         */
  const double* u_max;    /* [U]                                                       */
  const double* target_traj; /* [traj_len][S] or NULL                                  */
  const double* bias;     /* [U]     f_linear.bias (flg_bias, Policy.py:203-212) or NULL: u = squash(W phi + bias); not dropped out */
  double* g_bias;         /* [U]     OUT of mcp_rollout_bwd when bias != NULL: dJ/dbias (this rank's particles); may be NULL */
  mcp_meas meas;          /* what the policy is evaluated on (n == 0: the true state)  */
} mcp_policy;

/* Where the rollout's random numbers come from.  Parity mode: the host draws them with the
 * reference's own torch calls (SURVEY 8c order) and passes buffers.  Performance mode:
 * eps==NULL / masks==NULL -> Philox4x32-10 keyed by (seed, call) and counted by GLOBAL particle
 * id (m + particle_offset), so a sharded run draws the same numbers as a single-GPU run. */
typedef struct mcp_noise {
  const double* eps;      /* [T-1][M][G] standard normals, or NULL                     */
  const uint8_t* masks;   /* [T][M][B]   dropout keep-masks {0,1}, or NULL             */
  uint64_t seed;
  uint64_t call;          /* increments once per rollout so draws never repeat         */
  int64_t particle_offset;
  const uint64_t* call_dev; /* optional DEVICE counter added to `call` when the kernels start (NULL: none): a rollout recorded into a HIP graph
                               draws fresh numbers on every replay when the graph also advances this word -- the by-value `call` is frozen
                               in the recorded kernel arguments (round 6: MC_PILCO.reinforce_policy replays its attempts) */
} mcp_noise;

/* ---- library ------------------------------------------------------------------------ */
int mcp_abi_version(void);
const char* mcp_build_info(void);

/* ---- pretrain: Gram / Cholesky / inverse / alpha -------------------------------------- */
/* K[i][j] = k(X1_i, X2_j) (+ sigma_n2 on the diagonal when add_noise and X2==X1 semantics).
 * Replaces RBF.get_covariance (Stationary_GP.py:162-170), MPK_GP/Linear_GP.get_covariance
 * (Sparse_GP.py:426-441,625-646), Sum_Independent_GP.get_covariance (GP_prior.py:314-335). */
int mcp_cov_build(const mcp_kernel* kern, int N1, const double* X1, int N2, const double* X2, int add_noise,
                  double* K, int ldk, void* stream);
/* diag k(x_i,x_i) -- get_diag_covariance (Stationary_GP.py:172-181, Sparse_GP.py:443-453,658-668,
 * GP_prior.py:337-347). */
int mcp_cov_diag(const mcp_kernel* kern, int N, const double* X, int add_noise, double* diag, void* stream);
/* In place: A (symmetric, upper triangle read) -> U upper with A = U^T U; strictly-lower part
 * zeroed (from 600 rows on it serves as scratch on the way); logdet = 2 sum log U_ii.  N <= 8192.
 * torch.cholesky(K, upper=True) + log_det, GP_prior.py:106-107. */
int mcp_chol_factor(int N, double* A, int lda, double* logdet, uint32_t* status, void* stream);
/* Uinv = U^-1 (upper) and Kinv = Uinv Uinv^T.  torch.inverse(U), GP_prior.py:109-110.  Only the upper 16 x 16 blocks of Uinv are written
 * (and, beyond 1152 rows, its strictly-lower 128-row block rows are used as scratch and zeroed again): pass Uinv zero-filled for a clean lower
 * triangle.  N <= 16384. */
int mcp_chol_inverse(int N, const double* U, int ldu, double* Uinv, int ldi, double* Kinv, int ldk, void* stream);
/* out = A G A for a symmetric A [N][N] (K^-1) and any G [N][N]; scratch: N * N doubles; out, scratch, A, G distinct.  The chain rule through
 * torch.inverse in GP_prior.forward's autograd graph (GP_prior.py:109-110: d K^-1 = - K^-1 dK K^-1) for criteria other than the marginal
 * likelihood in GP_prior.fit_model (GP_prior.py:179-230).  N <= 16384. */
int mcp_sym_sandwich(int N, const double* A, int lda, const double* G, int ldg, double* out, int ldo, double* scratch, void* stream);
/* alpha = Kinv (Y - mean).  GP_prior.get_alpha, GP_prior.py:130-135. */
int mcp_gp_alpha(int N, const double* Kinv, int ldk, const double* Y, double mean, double* alpha, void* stream);
/* Greedy subset-of-data selection on the device, GP_prior.get_SOD (GP_prior.py:232-257):
 * idx_out[0..*n_out) receives the kept sample indices (ascending, bit-exact contract).
 * workspace: mcp_sod_workspace_bytes(N) -- W [N][N] and two running sums [N]; from 256 candidates on also the exchange area and the Gram matrix of
 * the form that runs across workgroups (one per 64 candidates, all resident; N <= 4096).  With 8 (N^2 + 2 N) bytes only, the call keeps to one
 * workgroup.  *n_out = -1: the workgroups never met (the device could not hold the grid) -- repeat with the smaller workspace. */
size_t mcp_sod_workspace_bytes(int N);
int mcp_sod_select(const mcp_kernel* kern, int N, const double* X, double threshold, int32_t* idx_out, int32_t* n_out,
                   void* workspace, size_t workspace_bytes, void* stream);
/* Gradient of the marginal likelihood  L = 1/2 ((Y-m)^T Kinv (Y-m) + logdet K)  w.r.t. the kernel's log-parameters --
 * what autograd computes through GP_prior.forward + Marginal_log_likelihood in GP_prior.fit_model (GP_prior.py:179-230,
 * gpr_lib/Likelihood/Gaussian_likelihood.py:15-24):  dL/dtheta = 1/2 tr((Kinv - alpha alpha^T) dK/dtheta).
 * grad [4D+3]: [0,D) d/d log lengthscale | D d/d log lambda | D+1  1/2 tr(Kinv - alpha alpha^T) (multiply by
 * d sigma_n^2/d sigma_n_log = 2 exp(2 sigma_n_log)) | [D+2,2D+3) MPK_1 Sigma_pos_par | [2D+3,3D+3), [3D+3,4D+3) the two
 * factors of MPK_2's Sigma_pos_par.  workspace: mcp_nll_workspace_bytes(N, D). */
size_t mcp_nll_workspace_bytes(int N, int D);
int mcp_nll_grad(const mcp_kernel* kern, int N, const double* X, const double* Kinv, int ldk, const double* alpha, double* grad,
                 void* workspace, size_t workspace_bytes, void* stream);
/* One epoch of hyper-parameter training for the G GPs of a model at once, from the optimizer's RAW parameters to their gradients and the
 * loss, without a host round trip -- what GP_prior.fit_model does per epoch through forward + Marginal_log_likelihood + autograd
 * (GP_prior.py:91-115,179-230; Gaussian_likelihood.py:15-24; Model_learning.train_gp_likelihood, Model_learning.py:398-421).  The GPs of
 * a model are independent (the reference trains them one after the other, Model_learning.py:149-161): every stage -- Gram, Cholesky,
 * U^-1, K^-1, alpha, gradient -- is ONE launch whose grid carries the GP index.  All GPs share the inputs X [N][D] and the kernel
 * structure: one squared-exponential term (lengthscales per dimension or, ard == 0, one shared) plus poly_deg = 0 / 1 / 2 Volterra
 * terms (MPK_1 with the offset feature, MPK_2 without), noise from the squared-exponential term.  mcp_nll_gp: device pointers to the
 * raw parameters as the reference stores them (log_lengthscales_par, log_lambda_par, sigma_n_log, mean_par, MPK Sigma_pos_par) and to
 * the gradient buffers (NULL: not wanted -- a frozen parameter).  16 < N <= 1152.  status: one word, OR of MCP_STATUS_NOT_SPD. */
typedef struct mcp_nll_gp {
  const double* log_ls;       /* [D], or [1] when ard == 0                                        */
  const double* log_lambda;   /* [1]                                                              */
  const double* sigma_n_log;  /* [1] or NULL: no noise parameter                                  */
  const double* mean;         /* [1] or NULL: zero prior mean                                     */
  const double* mpk1;         /* [D+1] MPK_1 Sigma_pos_par (log), or NULL                         */
  const double* mpk2;         /* [2D]  MPK_2 Sigma_pos_par (log): factor 0 | factor 1, or NULL    */
  const double* Y;            /* [N] targets                                                      */
  double y_scale;             /* the targets enter as Y * y_scale (1 / norm_list[g])              */
  double sigma_n_num2;        /* sigma_n_num^2                                                    */
  double* g_log_ls;           /* gradients w.r.t. the raw parameters, same shapes (NULL: skip)    */
  double* g_log_lambda;
  double* g_sigma_n_log;
  double* g_mean;
  double* g_mpk1;
  double* g_mpk2;
  double* loss;               /* [1]  1/2 ((Y-m)^T K^-1 (Y-m) + logdet K), or NULL                */
} mcp_nll_gp;
size_t mcp_nll_epoch_workspace_bytes(int G, int N, int D);
int mcp_nll_epoch(int G, const mcp_nll_gp* gps, int N, int D, int poly_deg, int ard, const double* X, uint32_t* status, void* workspace,
                  size_t workspace_bytes, void* stream);
/* Packs pretrain outputs into the mcp_gp layout (padding, transposes, aX). */
int mcp_gp_pack(int N, int D, const double* X, const double* alpha, const double* Kinv, int ldk, int Npad, double* Xt_out,
                double* X_out, double* alpha_out, double* Kinv_out, double* aX_out, void* stream);

/* ---- single-step GP posterior (GP_prior.get_estimate_from_alpha, GP_prior.py:137-155) -- */
/* mu[M], var[M] at test inputs Z [M][D]; Jmu/Jvar [M][D] = d mu/dz, d var/dz (NULL to skip). */
int mcp_posterior_fwd(const mcp_gp* gp, int M, const double* Z, double* mu, double* var, double* Jmu, double* Jvar,
                      uint32_t* status, void* stream);
/* gZ[m][d] = gmu[m]*Jmu[m][d] + gvar[m]*Jvar[m][d]  (adjoint of the above). */
int mcp_posterior_bwd(int M, int D, const double* gmu, const double* gvar, const double* Jmu, const double* Jvar, double* gZ,
                      void* stream);

/* ---- fused particle rollout (MC_PILCO.apply_policy, policy_learning/MC_PILCO.py:615-674:
 * T-loop of Model_learning.get_next_state (Model_learning.py:210-229,685-718) and the policy
 * forward (Policy.py:242-265,323-335,389-403)) ----------------------------------------- */
#define MCP_FWD_NO_GP_SHARDING 2 /* flag in mcp_rollout_fwd's particle_pred argument */
#define MCP_FWD_KT_PACKED 4      /* flags in the same argument: the workspace still holds the packed operand copies an EARLIER call built from    */
#define MCP_FWD_XJ_PACKED 8      /* this same model in this same workspace (KT: the lean small-swarm kernel's Kinv tiles; XJ: the wide classes'   */
                                 /* phase-J operands) -- the call does not rebuild them (8 us per rollout at the cart-pole size).  The caller's  */
                                 /* promise: nothing else wrote there and the model's arrays are unchanged.  Without the flags every call packs. */
size_t mcp_rollout_workspace_bytes(const mcp_model* model, const mcp_policy* policy, int M, int T);
/* x0 [M][S] -> states [T][M][S], inputs [T][M][U].  jac [T-1][M][G][D] (d delta_g/d z, sampling
 * included) is written when non-NULL and is what mcp_rollout_bwd consumes.
 * particle_pred: bit 0 clear -> delta = posterior mean (Model_learning.py:707-708); bit 1 (MCP_FWD_NO_GP_SHARDING) -> never launch
 * GP-sharded even though a workspace is passed (the repeat of a step that reported MCP_STATUS_SYNC: the workspace still carries the
 * packed operand copies of the wide / narrow kernels, only the hand-off between workgroups is given up).  T==1 evaluates the
 * policy only (Policy.forward); in that case `model` may be NULL.
 * workspace (optional, mcp_rollout_workspace_bytes): with it, small swarms run GP-sharded -- the G
 * workgroups of a particle cluster each evaluate one GP and hand each other the sampled increments
 * once per step through this buffer (zeroed by the call, on `stream`); results agree with the
 * unsharded launch to rounding.  MCP_STATUS_SYNC in `status` reports a hand-off that timed out
 * (the trajectories are then invalid).  Without a workspace the launch is never sharded.  For models with more
 * than 15 GP-input dimensions the workspace also takes the packed operand copies of the 16-particle kernel's
 * moment / Jacobian contraction (rebuilt by every call, on `stream`, unless MCP_FWD_XJ_PACKED says an earlier call's copy stands); without it
 * that phase keeps its slower form. */
int mcp_rollout_fwd(const mcp_model* model, const mcp_policy* policy, const mcp_noise* noise, int M, int T, int particle_pred,
                    const double* x0, double* states, double* inputs, double* jac, uint32_t* status, void* workspace,
                    size_t workspace_bytes, void* stream);
/* Reverse-time adjoint of the rollout: given dJ/dstates, dJ/dinputs (either may be NULL) returns
 * dJ/d{log_lengthscales [P], centers [B][P], f_linear.weight [U][B]} (overwritten, this rank's
 * particles only; with policy->bias also dJ/dbias into policy->g_bias) and optionally dJ/dx0 [M][S].  Replaces autograd's backward through
 * MC_PILCO.py:522. */
int mcp_rollout_bwd(const mcp_model* model, const mcp_policy* policy, const mcp_noise* noise, int M, int T,
                    const double* states, const double* inputs, const double* jac, const double* g_states,
                    const double* g_inputs, double* g_log_ls, double* g_centers, double* g_weight, double* g_x0,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ---- cost (policy_learning/Cost_function.py) ------------------------------------------ */
#define MCP_COST_CARTPOLE 0 /* cart_pole_cost, Cost_function.py:170-182                          */
#define MCP_COST_TRAJ 1     /* saturated_distance_from_trajectory, Cost_function.py:124-147      */
typedef struct mcp_cost {
  int32_t kind;
  int32_t S;
  int32_t angle_index, pos_index;       /* cart-pole                                    */
  double target_angle, target_pos;      /* target_state = [theta*, x*]                  */
  double ls_angle, ls_pos;              /* lengthscales = [l_theta, l_x]                */
  int32_t n_used;                       /* trajectory cost: used_indeces                */
  int32_t used[MCP_MAX_STATE];
  const double* target_traj;            /* [T][S]                                       */
  const double* lengthscales;           /* [n_used]                                     */
} mcp_cost;
/* costs[t][m] = c(x_{t,m});  moments[t] = {mean_m c, sum_m (c-mean)^2} over THIS rank's M
 * particles (two-pass, as torch.mean/torch.std do; Cost_function.py:32-36). */
int mcp_cost_fwd(const mcp_cost* cost, int T, int M, const double* states, double* costs, double* moments, uint32_t* status,
                 void* stream);
/* Pools R ranks' moments [R][T][2] with counts[R] (host array) and writes out[0]=sum_t mean,
 * out[1]=sum_t unbiased std.  R==1 on a single GPU. */
int mcp_cost_finalize(int T, int R, const double* moments, const int64_t* counts, double* out, void* stream);
/* g_states[t][m][:] = (*g_cost) * gscale * d c/d x.  g_cost: DEVICE scalar with the upstream
 * gradient of the cost (NULL = 1), so no host synchronisation is needed; gscale = 1/M_total. */
int mcp_cost_bwd(const mcp_cost* cost, int T, int M, const double* states, const double* g_cost, double gscale, double* g_states,
                 void* stream);

/* Summable form of one rank's cost moments, for the SINGLE all-reduce of a particle-sharded step:
 * sums[t] = sum_m (c_tm - shift_t), sums[T+t] = sum_m (c_tm - shift_t)^2 over this rank's M particles, from the
 * moments of mcp_cost_fwd.  shift [T] (the same on every rank; NULL = 0; the previous step's pooled means are the
 * natural choice) keeps the pooled variance free of cancellation. */
int mcp_cost_sums(int T, int M, const double* moments, const double* shift, double* sums, void* stream);
/* sums [2T] added over all ranks (n_total particles) -> out[0] = sum_t mean_m c, out[1] = sum_t unbiased std_m c
 * (Cost_function.py:32-36 on the pooled swarm); mean_out [T] (optional) receives the pooled mean per time step. */
int mcp_cost_finalize_sums(int T, int64_t n_total, const double* sums, const double* shift, double* out, double* mean_out,
                           void* stream);

/* ---- the optimizer loop's bookkeeping on the device (MC_PILCO.reinforce_policy, policy_learning/MC_PILCO.py:475-607) ----------
 * The reference decides on the host, from torch.isnan(cost), whether a rollout counts, and so reads every step's cost back before
 * it can launch the next one.  These two entry points take the same decisions from device memory after each ATTEMPT (rollout +
 * cost + adjoint sweep), so the host may enqueue the next attempt at once and read the outcome (a small record) one attempt late:
 *   - an attempt FAILS when its cost is NaN (or flags / status say so: hand-off time-out, non-positive variance): nothing is
 *     updated, the next attempt is the retry (:479-501: "Cost is NaN: try sampling again", at most MCP_OPT_MAX_ATTEMPTS);
 *   - it is VOID while the loop waits for the host: after the tenth failure in a row (re-initialisation, :573-607), after the
 *     lr / exit condition fired (`pending`, :540-567), after step n_steps;
 *   - otherwise it COUNTS: parameters updated, cost_list / std_list [step] written, the cost-difference monitors advanced
 *     (ES1, ES2, diff_cost_ratio, :503-519), step += 1.
 * mcp_opt_state lives in device memory, zero-initialised (+ cost_prev = the warm-up cost, :462) by the caller; the caller resets
 * it (memset on the stream) when it builds a new optimizer or re-initialises the policy. */
#define MCP_OPT_MAX_ATTEMPTS 10
#define MCP_OPT_MAX_TENSORS 32
#define MCP_OPT_RECORD_DOUBLES 12
typedef struct mcp_opt_state {
  int64_t step;           /* optimizer steps taken since the last (re-)initialisation = next index of cost_list            */
  int64_t attempt;        /* failed attempts of the current step                                                          */
  int64_t pending;        /* 1: the lr / exit condition fired at the last counted attempt; cleared by the host            */
  int64_t adam_t;         /* steps of the CURRENT optimizer (Adam's bias correction); the host zeroes it with a new one   */
  int64_t total_attempts; /* every attempt seen, whatever became of it                                                    */
  double es2;             /* ES2_diff_cost                                                                                */
  double cost_prev;       /* cost_tm1                                                                                     */
} mcp_opt_state;
/* torch.optim.Adam's update (weight_decay 0, no amsgrad) of up to MCP_OPT_MAX_TENSORS parameter tensors in ONE launch, applied only
 * when the attempt counts (state == NULL: always, as step number `step` >= 1 of the optimizer -- GP training, where the host knows it;
 * with a state the step number is state->adam_t + 1).  params / grads / exp_avg / exp_avg_sq: HOST arrays of n_tensors device pointers,
 * numel their sizes; a NULL grad skips that tensor.  Must be enqueued BEFORE mcp_policy_step_commit of the same attempt (it reads
 * the state that call advances).  Without a state, a non-NULL `status` whose MCP_STATUS_NOT_SPD bit is set skips the update (GP
 * training: the epoch's Cholesky failed; the bit is sticky, so the parameters stay those of the last good epoch).
 * Replaces optimizer.step(), MC_PILCO.py:525 and GP_prior.py:209. */
int mcp_adam_step_guarded(int n_tensors, double* const* params, const double* const* grads, double* const* exp_avg,
                          double* const* exp_avg_sq, const int64_t* numel, double lr, double beta1, double beta2, double eps,
                          const mcp_opt_state* state, int64_t step, int n_steps, const double* cost, const double* flags,
                          const uint32_t* status, void* stream);
/* The loop's decisions for one attempt.  cost / std_cost: device scalars of the attempt; flags: optional device [3] doubles (> 0 =
 * NaN cost, hand-off time-out, non-positive variance -- the all-reduced form of a particle-sharded step), status: optional device
 * status word of the rollout (either or both may be NULL; a NaN cost always fails).  cost_list / std_list [n_steps], es1 / ratio
 * [n_steps + 1] (zero-initialised).  record (optional, device, MCP_OPT_RECORD_DOUBLES): [counted, void, step, attempts failed so
 * far, pending, cost, std, |ratio|, nan, time-out, non-positive variance, total attempts] for the host to read late. */
int mcp_policy_step_commit(mcp_opt_state* state, int n_steps, const double* cost, const double* std_cost, const double* flags,
                           const uint32_t* status, double* cost_list, double* std_list, double* es1, double* ratio,
                           double alpha_diff_cost, double min_step, double min_diff_cost, int num_min_diff_cost, double* record,
                           void* stream);

/* ---- particle sharding: the one collective of an optimizer step --------------------------
 * The reference is single-process (no collective anywhere); sharding the particles over the GPUs of a node adds ONE
 * exchange between `cost.backward()` and `optimizer.step()` (policy_learning/MC_PILCO.py:522-525): an in-place
 * all-reduce(sum) of the flat fp64 message [dJ/dlog_lengthscales | dJ/dcenters | dJ/dweight | mcp_cost_sums (2T) | flags].
 * Thin wrapper over RCCL (bound at run time; MCP_ERR_COMM when it cannot be loaded), ONE communicator per process created
 * once by mcp_comm_init and reused by every step.  Rank 0 obtains `id` (MCP_COMM_ID_BYTES) from mcp_comm_unique_id and
 * the host distributes it to the other ranks (any side channel: torch.distributed's store, MPI, a file). */
#define MCP_COMM_ID_BYTES 128
int mcp_comm_unique_id(void* id_out);
int mcp_comm_init(int world, int rank, const void* id);
int mcp_comm_world(void); /* 0 when no communicator exists */
int mcp_allreduce_grad(double* flat, size_t n, void* stream);
int mcp_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* MCPILCO_HIP_H */
