"""Fixture arrays -> packed HIP descriptors (GPU tests only)."""
import numpy as np
import torch

from helpers import fixture_poly, kind_cfg
from mc_pilco_amd import ops

DT = torch.float64


def dev():
    return torch.device("cuda", 0)


def G(a):
    return torch.as_tensor(np.asarray(a), dtype=DT).to(dev()).contiguous()


def spec_from(ls, sigma_n, lam=1.0, poly_w=None):
    """poly_w: list of raw (positive) weight vectors as given to the reference's
    Sigma_pos_par_init_list (their log is the parameter)."""
    w1 = w20 = w21 = None
    if poly_w:
        w1 = ops.mpk_weights(np.log(poly_w[0]), 1)[0]
        if len(poly_w) > 1:
            w20, w21 = ops.mpk_weights(np.log(poly_w[1]), 2)
    return ops.KernelSpec(torch.as_tensor(np.asarray(ls), dtype=DT), float(lam), float(sigma_n) ** 2, 0.0, w1, w20, w21)


def packed_model(fx, kind):
    c = kind_cfg(kind)
    gps = []
    for g in range(c["G"]):
        sp = spec_from(c["lengthscales"], float(fx["sigma_n"]), c["lam"], fixture_poly(fx, g))
        gps.append(ops.PackedGP(sp, G(fx["Xtr%d" % g]), G(fx["alpha%d" % g]), G(fx["Kinv%d" % g])))
    return ops.PackedModel(gps, c["S"], c["U"], c["Ts"], c["angle"], c["not_angle"], c["vel"], c["not_vel"])


def packed_policy(fx, kind, requires_grad=True):
    c = kind_cfg(kind)
    log_ls = torch.log(G(fx["pol_ls"])).requires_grad_(requires_grad)
    centers = G(fx["pol_centers"]).requires_grad_(requires_grad)
    weight = G(fx["pol_weight"]).requires_grad_(requires_grad)
    if kind == "ur5":
        return ops.PackedPolicy("traj", c["S"], log_ls, centers, weight, c["u_max"], True, target_traj=fx["target_traj"])
    return ops.PackedPolicy("angles", c["S"], log_ls, centers, weight, c["u_max"], True, angle=[2], non_angle=[0, 1, 3])


def packed_cost(fx, kind):
    c = kind_cfg(kind)
    if kind == "ur5":
        return ops.PackedCost("traj", c["S"], dev(), target_traj=fx["target_traj"], lengthscales=c["cost_ls"], used=None)
    return ops.PackedCost("cartpole", c["S"], dev(), target_state=c["cost_target"], lengthscales=c["cost_ls"], angle_index=c["cost_angle_index"],
                          pos_index=c["cost_pos_index"])


def noise_from(fx):
    eps = G(fx["eps"])
    masks = torch.as_tensor(fx["masks"]).to(dev()).contiguous() if "masks" in fx else None
    return ops.NoiseSpec(eps=eps, masks=masks)


class forced_variant:
    """Context manager over the library's test hooks: ``code`` 0 = automatic dispatch; 1 / 2 / 4 = that many particles per
    workgroup, all GPs in the workgroup; 16 = the 16-particle matrix-core kernel; 101 / 102 / 104 / 116 = the GP-sharded launch
    (G workgroups per cluster of 1 / 2 / 4 / 16 particles, met by a per-step hand-off) on the GENERAL kernels; 201 / 202 / 204 =
    the GP-sharded launch on the latency-lean kernel of narrow SE-only models (``rollout_fwd_lat_kernel``; models it does not
    cover run the general sharded kernel).  The backward sweep follows: forced codes run the general sweep with 1 / 2 / 4 particles
    per workgroup, codes 0 and 2xx the automatic one (the latency-lean sweep ``rollout_bwd_lat_kernel`` where it applies).
    ``check()`` asserts that the forced variant is the one that ran."""

    def __init__(self, code, bwd_particles=None):
        self.code = code
        self.ppw = code % 100
        self.sharded = code >= 100
        self.lean = code >= 200
        self.pb = bwd_particles if bwd_particles is not None else (0 if self.lean else {0: 0, 1: 1, 2: 2, 4: 4, 16: 4}[self.ppw])

    def __enter__(self):
        from mc_pilco_amd import hipabi

        L = hipabi.lib()
        L.mcp_debug_set_particles_per_wg(self.ppw)
        L.mcp_debug_set_bwd_particles(self.pb)
        L.mcp_debug_set_gp_sharding(1 if self.sharded else (-1 if self.code == 0 else 0))
        L.mcp_debug_set_fwd_lean(-1 if (self.lean or self.code == 0) else 0)
        return self

    def check(self, sharding_optional=False, lean_expected=None):
        """``sharding_optional``: wide shapes whose operands do not fit the LDS beside the policy's cannot be GP-sharded; the
        library then runs the unsharded kernel of the same tile size.  ``lean_expected``: for codes 2xx, whether the model is
        one the lean kernel covers (None: not checked)."""
        from mc_pilco_amd import hipabi

        L = hipabi.lib()
        if self.code:
            assert L.mcp_debug_last_particles_per_wg() == self.ppw, "forced kernel variant was not the one launched"
            if not (sharding_optional and self.sharded):
                assert bool(L.mcp_debug_last_gp_sharded()) == self.sharded, "GP sharding was not what the test forced"
            if not self.lean:
                assert L.mcp_debug_last_fwd_lean() == 0, "the general kernel was forced, the lean one ran"
            elif lean_expected is not None:
                assert bool(L.mcp_debug_last_fwd_lean()) == bool(lean_expected), "lean kernel: expected %s" % lean_expected

    def __exit__(self, *exc):
        from mc_pilco_amd import hipabi

        L = hipabi.lib()
        L.mcp_debug_set_particles_per_wg(0)
        L.mcp_debug_set_bwd_particles(0)
        L.mcp_debug_set_gp_sharding(-1)
        L.mcp_debug_set_fwd_lean(-1)
        return False


VARIANTS = [0, 1, 2, 4, 16, 101, 102, 104, 116, 201, 202, 204]
