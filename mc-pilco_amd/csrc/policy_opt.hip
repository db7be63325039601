// The bookkeeping of MC_PILCO.reinforce_policy's optimizer loop (policy_learning/MC_PILCO.py:475-607) on the device, so that the
// host never has to wait for a step's outcome before it enqueues the next one:
//
//   while opt_step < num_opt_steps:                                  reference, host side, per optimizer step
//       <= 10 attempts: apply_policy -> cost; NaN -> "try sampling again"            (:479-501)
//       cost_list[opt_step] = cost; ES1 / ES2 / diff_cost_ratio monitors            (:503-519)
//       cost.backward(); optimizer.step()                                            (:522-525)
//       opt_step > min_step and |ratio| < min_diff for num_min_diff_cost steps -> halve lr / exit   (:540-567)
//
// The reference decides each of these on the host from `torch.isnan(cost)`, i.e. after a device -> host read of every step's cost.
// Here one ATTEMPT (rollout, cost, adjoint sweep) is followed by two tiny launches that take the same decisions from device memory:
//   mcp_adam_step_guarded   the parameter update, applied only when the attempt counts (torch.optim.Adam's arithmetic);
//   mcp_policy_step_commit  cost list, monitors, counters, the lr / exit condition; and a small RECORD of what was decided, which the
//                           host copies back asynchronously and reads one attempt late (printing, lr changes, re-initialisation).
// An attempt does not count ("void") while the loop waits for the host: after ten failed attempts of one step (the host re-initialises
// the policy), after the lr / exit condition fired (the host builds the new optimizer), after the last step.  A failed attempt needs no
// host action at all: the next attempt IS the retry (same parameters, fresh noise).
#include "mcp_device.h"

namespace {

__device__ __forceinline__ bool attempt_failed(const double* cost, const double* flags, const uint32_t* status) {
  bool fail = cost[0] != cost[0];  // NaN cost (MC_PILCO.py:497)
  if (flags) fail = fail || flags[0] > 0.0 || flags[1] > 0.0 || flags[2] > 0.0;
  if (status) fail = fail || (status[0] & (MCP_STATUS_SYNC | MCP_STATUS_NONPOS_VAR)) != 0u;
  return fail;
}
__device__ __forceinline__ bool loop_frozen(const mcp_opt_state* st, int n_steps) {
  return st->pending != 0 || st->attempt >= MCP_OPT_MAX_ATTEMPTS || st->step >= n_steps;
}

struct AdamSegs {
  double* p[MCP_OPT_MAX_TENSORS];
  const double* g[MCP_OPT_MAX_TENSORS];
  double* m[MCP_OPT_MAX_TENSORS];
  double* v[MCP_OPT_MAX_TENSORS];
  long long end[MCP_OPT_MAX_TENSORS];  // running element count
  int n;
};

// torch.optim.Adam (weight_decay = 0, amsgrad = False, maximize = False), one thread per element, in torch's order of operations:
//   exp_avg.lerp_(grad, 1 - beta1);  exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
//   denom = exp_avg_sq.sqrt() / sqrt(1 - beta2^t) + eps;  param.addcdiv_(exp_avg, denom, value = -lr / (1 - beta1^t))
__global__ void adam_guarded_kernel(AdamSegs s, double lr, double beta1, double beta2, double eps, const mcp_opt_state* st, long long step,
                                    int n_steps, const double* cost, const double* flags, const uint32_t* status) {
  if (st && (loop_frozen(st, n_steps) || attempt_failed(cost, flags, status))) return;  // (uniform over the grid)
  // GP training (no state): an epoch whose Cholesky met a matrix that is not positive definite leaves NaN gradients behind; the flag is
  // sticky, so this and every later epoch's update is skipped and the parameters stay those of the last good step (the reference raises
  // at the failing epoch with them intact, GP_prior.py:106)
  if (!st && status && (status[0] & MCP_STATUS_NOT_SPD) != 0u) return;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.end[s.n - 1]) return;
  int k = 0;
  while (i >= s.end[k]) ++k;
  const long long e = i - (k ? s.end[k - 1] : 0);
  const double t = st ? (double)(st->adam_t + 1) : (double)step;
  const double g = s.g[k][e];
  double m = s.m[k][e], v = s.v[k][e];
  m = m + (1.0 - beta1) * (g - m);
  v = v * beta2 + (1.0 - beta2) * g * g;
  const double bc1 = 1.0 - pow(beta1, t), bc2 = 1.0 - pow(beta2, t);
  const double denom = sqrt(v) / sqrt(bc2) + eps;
  s.m[k][e] = m;
  s.v[k][e] = v;
  s.p[k][e] = s.p[k][e] + (-(lr / bc1)) * (m / denom);
}

// one thread: the loop's decisions for this attempt
__global__ void step_commit_kernel(mcp_opt_state* st, int n_steps, const double* cost, const double* std_, const double* flags,
                                   const uint32_t* status, double* cost_list, double* std_list, double* es1, double* ratio, double alpha,
                                   double min_step, double min_diff, int num_min_diff_cost, double* rec) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double c = cost[0], sd = std_ ? std_[0] : 0.0;
  const bool frozen = loop_frozen(st, n_steps), fail = attempt_failed(cost, flags, status);
  uint32_t sw = status ? status[0] : 0u;
  double committed = 0.0, voided = 0.0, rabs = 0.0;
  const long long k = st->step;
  st->total_attempts += 1;
  if (frozen) {
    voided = 1.0;
  } else if (fail) {
    st->attempt += 1;
    if (st->attempt >= MCP_OPT_MAX_ATTEMPTS) {
      // the tenth failure: the reference takes the step on the failed cost (MC_PILCO.py:503-525) before it re-initialises the policy;
      // of that step only the two monitor values that survive the re-initialisation are kept here (ES2 and cost_tm1 are NOT reset
      // there, :580-603 -- after a NaN step they stay NaN and the lr / exit condition never fires again: reproduced on purpose)
      const double diff = c - st->cost_prev;
      st->es2 = alpha * (st->es2 + (1.0 - alpha) * ((diff - es1[k]) * (diff - es1[k])));
      st->cost_prev = c;
    }
  } else {
    cost_list[k] = c;
    std_list[k] = sd;
    const double diff = c - st->cost_prev;
    const double e1 = alpha * es1[k] + (1.0 - alpha) * diff;
    es1[k + 1] = e1;
    const double dd = diff - es1[k];
    st->es2 = alpha * (st->es2 + (1.0 - alpha) * (dd * dd));
    st->cost_prev = c;
    const double r = alpha * ratio[k] + (1.0 - alpha) * (e1 / sqrt(st->es2));
    ratio[k + 1] = r;
    rabs = fabs(r);
    if ((double)k > min_step) {  // :540-547: the last num_min_diff_cost ratios BEFORE this step's (ratio[k + 1 - n .. k]) all below the bound
      long long cnt = 0;         // (a window that would reach before the start cannot hold n entries: the condition never fires there)
      const long long lo = k + 1 - num_min_diff_cost;
      if (lo >= 0)
        for (long long j = lo; j <= k; ++j) cnt += fabs(ratio[j]) < min_diff ? 1 : 0;
      if (lo >= 0 && cnt >= num_min_diff_cost) st->pending = 1;
    }
    st->step = k + 1;
    st->attempt = 0;
    st->adam_t += 1;
    committed = 1.0;
  }
  if (rec) {
    rec[0] = committed;
    rec[1] = voided;
    rec[2] = (double)k;  // the step this attempt belonged to
    rec[3] = (double)st->attempt;
    rec[4] = (double)st->pending;
    rec[5] = c;
    rec[6] = sd;
    rec[7] = rabs;
    rec[8] = (c != c || (flags && flags[0] > 0.0)) ? 1.0 : 0.0;
    rec[9] = ((sw & MCP_STATUS_SYNC) || (flags && flags[1] > 0.0)) ? 1.0 : 0.0;
    rec[10] = ((sw & MCP_STATUS_NONPOS_VAR) || (flags && flags[2] > 0.0)) ? 1.0 : 0.0;
    rec[11] = (double)st->total_attempts;
  }
}

}  // namespace

extern "C" int mcp_adam_step_guarded(int n_tensors, double* const* params, const double* const* grads, double* const* exp_avg,
                                     double* const* exp_avg_sq, const int64_t* numel, double lr, double beta1, double beta2, double eps,
                                     const mcp_opt_state* state, int64_t step, int n_steps, const double* cost, const double* flags,
                                     const uint32_t* status, void* stream) {
  if (n_tensors <= 0 || !params || !grads || !exp_avg || !exp_avg_sq || !numel) return MCP_ERR_ARG;
  if (n_tensors > MCP_OPT_MAX_TENSORS) return MCP_ERR_LIMIT;
  if (state && !cost) return MCP_ERR_ARG;
  if (!state && step < 1) return MCP_ERR_ARG;
  AdamSegs s;
  long long tot = 0;
  s.n = 0;
  for (int i = 0; i < n_tensors; ++i) {
    if (numel[i] <= 0 || !grads[i]) continue;  // (a parameter the cost does not reach: torch's Adam skips it too)
    if (!params[i] || !exp_avg[i] || !exp_avg_sq[i]) return MCP_ERR_ARG;
    s.p[s.n] = params[i];
    s.g[s.n] = grads[i];
    s.m[s.n] = exp_avg[i];
    s.v[s.n] = exp_avg_sq[i];
    tot += numel[i];
    s.end[s.n] = tot;
    ++s.n;
  }
  if (s.n == 0) return MCP_OK;
  for (int i = s.n; i < MCP_OPT_MAX_TENSORS; ++i) {
    s.p[i] = nullptr;
    s.g[i] = nullptr;
    s.m[i] = s.v[i] = nullptr;
    s.end[i] = tot;
  }
  hipLaunchKernelGGL(adam_guarded_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s, lr, beta1, beta2, eps, state,
                     (long long)step, n_steps, cost, flags, status);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

extern "C" int mcp_policy_step_commit(mcp_opt_state* state, int n_steps, const double* cost, const double* std_cost, const double* flags,
                                      const uint32_t* status, double* cost_list, double* std_list, double* es1, double* ratio,
                                      double alpha_diff_cost, double min_step, double min_diff_cost, int num_min_diff_cost, double* record,
                                      void* stream) {
  if (!state || !cost || !cost_list || !std_list || !es1 || !ratio || n_steps <= 0 || num_min_diff_cost < 0) return MCP_ERR_ARG;
  hipLaunchKernelGGL(step_commit_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, n_steps, cost, std_cost, flags, status, cost_list,
                     std_list, es1, ratio, alpha_diff_cost, min_step, min_diff_cost, num_min_diff_cost, record);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}
