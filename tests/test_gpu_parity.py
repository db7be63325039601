"""GPU parity: the HIP path (through the C ABI) against the reference's golden vectors and the
CPU oracle on the same seeded inputs.  Tolerances (fp64) are the ones SURVEY.md 8c states:
per-op rel 1e-12 (Gram, cost, policy), Cholesky-derived quantities rel 1e-9 (cond-scaled),
short rollouts abs 1e-9, T=60 rollout abs 1e-6 / gradients rel 1e-6, indices exact.
"""
import numpy as np
import pytest
import torch

from helpers import ROLLOUT_FIXTURES, T, hyper, oracle_cost_fn, oracle_model, oracle_policy
from oracle import mcpilco_oracle as orc

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=float)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def abserr(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=float)
    return float(np.max(np.abs(a - np.asarray(b, dtype=float))))


def _kernel_spec(fx):
    from gpu_helpers import spec_from

    pw = [fx[k] for k in ("poly_w1", "poly_w2") if k in fx]
    return spec_from(fx["lengthscales"], float(fx["sigma_n"]), float(fx["lam"]), pw or None)


@pytest.mark.parametrize("name", ["kern_se", "kern_se_poly2", "kern_se_poly1_d24"])
def test_gram_cholesky_alpha_posterior(golden, name):
    from gpu_helpers import G
    from mc_pilco_amd import ops

    fx = golden(name)
    sp = _kernel_spec(fx)
    X, Y, Xs = G(fx["X"]), G(fx["Y"]), G(fx["Xs"])
    K = ops.cov_build(sp, X, None, noise=True)
    assert relerr(K, fx["K_noise"]) < 1e-12
    assert relerr(ops.cov_build(sp, Xs, X), fx["K_cross"]) < 1e-12
    assert relerr(ops.cov_diag(sp, Xs), fx["diag"]) < 1e-12
    U, logdet, status = ops.chol_factor(K)
    assert int(status.item()) == 0
    assert relerr(U.t() @ U, fx["K_noise"]) < 1e-13  # it is a Cholesky factor
    assert torch.equal(U, torch.triu(U))
    assert abs(float(logdet) - float(fx["logdet"])) < 1e-11 * abs(float(fx["logdet"]))
    Ui, Kinv = ops.chol_inverse(U)
    assert relerr(Kinv, fx["Kinv"]) < 1e-9
    alpha = ops.gp_alpha(Kinv, Y, 0.0)
    assert relerr(alpha, fx["alpha"]) < 1e-9
    # posterior from the reference's cached operands: isolates the per-step kernel
    gp = ops.PackedGP(sp, X, G(fx["alpha"]), G(fx["Kinv"]))
    mu, var = ops.posterior(gp, Xs)
    # mu = k.alpha cancels heavily (sum|k alpha| >> |mu|): the stated posterior tolerance is rel 1e-10
    assert relerr(mu, fx["mu"]) < 1e-10
    assert abserr(var, fx["var"]) < 1e-10 * max(1.0, float(np.max(np.abs(fx["diag"]))))


def test_not_spd_is_flagged():
    from gpu_helpers import G
    from mc_pilco_amd import ops

    A = G(np.array([[1.0, 2.0], [2.0, 1.0]]))
    _, _, status = ops.chol_factor(A)
    assert ops.status_flags(status)["not_spd"]


def test_cholesky_sizes():
    """Edge sizes around the 16-wide block and the 64-wide column chunk; N=300 as in the benchmark."""
    from gpu_helpers import G
    from mc_pilco_amd import ops

    rs = np.random.RandomState(0)
    for N in (1, 2, 15, 16, 17, 63, 64, 65, 300):
        A = rs.randn(N, N + 3)
        K = A @ A.T / (N + 3) + 0.1 * np.eye(N)
        U, logdet, status = ops.chol_factor(G(K))
        assert int(status.item()) == 0
        Ui, Kinv = ops.chol_inverse(U)
        assert relerr(U.t() @ U, K) < 1e-13
        assert relerr(Kinv, np.linalg.inv(K)) < 1e-10
        assert abs(float(logdet) - np.linalg.slogdet(K)[1]) < 1e-10 * max(1.0, abs(np.linalg.slogdet(K)[1]))


@pytest.mark.parametrize("mfma", [1, 3, 2, 0])
def test_cholesky_and_inverse_kernels_both_forms(mfma):
    """The MFMA-blocked Cholesky / triangular inverse (1: the left-looking factorization and column-parallel inverse of round 4, 2: the
    right-looking / block-diagonal forms of round 3) and the round-1/2 forms behind them (mcp_debug_set_chol_mfma), at
    sizes around every block edge (16-wide blocks, partial last block, one block only), at the benchmark's N = 300, the UR5 model's
    N = 400 and near the limit; a Gram-like ill-conditioned matrix (cond ~1e6) as the GP training sees it; upper-triangular outputs
    (zeros below the diagonal, as torch.cholesky(upper=True) / torch.inverse(U) return them); the not-SPD flag from a late pivot."""
    from gpu_helpers import G
    from mc_pilco_amd import hipabi, ops

    hipabi.lib().mcp_debug_set_chol_mfma(mfma)
    try:
        rs = np.random.RandomState(1)
        for N in (17, 31, 32, 33, 47, 48, 100, 129, 300, 400, 401, 500, 576, 577, 784, 785, 1000, 1152):
            A = rs.randn(N, N + 3)
            K = A @ A.T / (N + 3) + 0.1 * np.eye(N)
            U, logdet, status = ops.chol_factor(G(K))
            assert int(status.item()) == 0
            Ui, Kinv = ops.chol_inverse(U)
            assert float(torch.tril(U, -1).abs().max()) == 0.0 and float(torch.tril(Ui, -1).abs().max()) == 0.0
            assert relerr(U.t() @ U, K) < 1e-13
            assert relerr(Ui @ U, np.eye(N)) < 1e-11
            assert relerr(Kinv, np.linalg.inv(K)) < 1e-10
            assert abs(float(logdet) - np.linalg.slogdet(K)[1]) < 1e-10 * max(1.0, abs(np.linalg.slogdet(K)[1]))
        x = np.linspace(0.0, 4.0, 300)[:, None]
        K = np.exp(-((x - x.T) / 0.6) ** 2) + 1e-4 * np.eye(300)  # SE Gram + noise: cond ~ 1e6
        U, logdet, status = ops.chol_factor(G(K))
        assert int(status.item()) == 0
        _, Kinv = ops.chol_inverse(U)
        ref = np.linalg.inv(K)
        assert relerr(Kinv, ref) < 1e-8 and abs(float(logdet) - np.linalg.slogdet(K)[1]) < 1e-8 * abs(np.linalg.slogdet(K)[1])
        K = np.eye(40)
        K[37, 37] = -1.0  # the pivot of a late block
        assert ops.status_flags(ops.chol_factor(G(K))[2])["not_spd"]
    finally:
        hipabi.lib().mcp_debug_set_chol_mfma(1)


def test_sod_indices_exact(golden):
    from gpu_helpers import G, spec_from
    from mc_pilco_amd import ops

    fx = golden("sod")
    sp = spec_from(fx["lengthscales"], float(fx["sigma_n"]))
    X = G(fx["X"])
    assert ops.sod_select(sp, X, float(fx["thr_rel"])) == [int(i) for i in fx["idx_rel"]]
    assert ops.sod_select(sp, X, float(fx["thr_abs"])) == [int(i) for i in fx["idx_abs"]]


@pytest.mark.parametrize("N", [600, 601, 729, 1000, 1153, 1500, 2048, 4096])
def test_blocked_cholesky_across_workgroups(N):
    """Round 5: from 600 rows on `mcp_chol_factor` factorises by panels of 128 rows across the chip (the diagonal block by the one-workgroup
    kernel, U_kj = W^T A_kj and the trailing update A_ij -= U_ki^T U_kj as MFMA products of one wave per tile; scratch in the lower triangle,
    which comes back zero) -- sizes around the panel edges, beyond the old 1152-row limit, up to 4096: U^T U = K to 1e-13, upper
    triangular, logdet against numpy to 1e-10; and the not-positive-definite flag from a pivot in a late panel."""
    from gpu_helpers import G
    from mc_pilco_amd import ops

    rs = np.random.RandomState(N)
    A = rs.randn(N, N + 3)
    K = A @ A.T / (N + 3) + 0.1 * np.eye(N)
    Kg = G(K)
    U, logdet, status = ops.chol_factor(Kg)
    assert int(status.item()) == 0
    assert float(torch.tril(U, -1).abs().max()) == 0.0
    assert float((U.t() @ U - Kg).abs().max()) < 1e-13 * float(Kg.abs().max())
    ref = np.linalg.slogdet(K)[1]
    assert abs(float(logdet) - ref) < 1e-10 * max(1.0, abs(ref))
    # (beyond 1152 rows: U^-T by block forward substitution in the Kinv buffer, U^-1 = its transpose, K^-1 = U^-1 U^-T by tiles)
    Ui, Kinv = ops.chol_inverse(U)
    assert float(torch.tril(Ui, -1).abs().max()) == 0.0
    assert float((Ui @ U - torch.eye(N, dtype=U.dtype, device=U.device)).abs().max()) < 1e-11
    assert relerr(Kinv, np.linalg.inv(K)) < 1e-10
    if N == 729:
        K2 = K.copy()
        K2[700, 700] = -5.0  # a pivot of the last panel
        assert ops.status_flags(ops.chol_factor(G(K2))[2])["not_spd"]


@pytest.mark.parametrize("pre,kind", [("plain", "plain"), ("ang", "angles"), ("traj", "traj")])
def test_policy_forward(golden, pre, kind):
    """Policy.forward == the fused kernel with T=1."""
    from gpu_helpers import G, dev
    from mc_pilco_amd import ops

    fx = golden("policy")
    um = fx[pre + "_umax"]
    um = float(um) if um.ndim == 0 else [float(v) for v in um]
    x = G(fx[pre + "_x"])
    S = x.shape[1]
    M = x.shape[0]
    tt = fx["traj_target"] if kind == "traj" else None
    pol = ops.PackedPolicy(kind, S, torch.log(G(fx[pre + "_ls"])), G(fx[pre + "_centers"]), G(fx[pre + "_weight"]), um, True, angle=[2],
                           non_angle=[0, 1, 3], target_traj=None if tt is None else tt[int(fx["traj_t"]):])
    U = pol.U
    # a dummy one-GP model is not needed for T=1, but the ABI wants a valid descriptor
    from gpu_helpers import spec_from

    D = S + U
    sp = spec_from(np.ones(D), 0.1)
    gp = ops.PackedGP(sp, G(np.zeros((16, D))), G(np.zeros(16)), G(np.eye(16)))
    model = ops.PackedModel([gp] * 1, S, U, 0.05, [], list(range(S)), [0], [1])
    st, u0, _ = ops.rollout(model, pol, ops.NoiseSpec(), x, 1, 0.0)
    assert relerr(u0[0], fx[pre + "_u0"]) < 1e-12
    masks = torch.as_tensor(fx[pre + "_mask"].astype(np.uint8)).reshape(1, M, -1).to(dev()).contiguous()
    st, u25, _ = ops.rollout(model, pol, ops.NoiseSpec(masks=masks), x, 1, 0.25)
    assert relerr(u25[0], fx[pre + "_u25"]) < 1e-12


def test_costs(golden):
    from gpu_helpers import G, dev
    from mc_pilco_amd import ops

    fx = golden("cost")
    st = G(fx["cp_states"]).requires_grad_(True)
    cost = ops.PackedCost("cartpole", 4, dev(), target_state=[np.pi, 0.0], lengthscales=[3.0, 1.0], angle_index=2, pos_index=0)
    c, s = ops.expected_cost(cost, st)
    c.backward()
    assert abs(float(c) - float(fx["cp_cost"])) < 1e-12 * abs(float(fx["cp_cost"]))
    assert abs(float(s) - float(fx["cp_std"])) < 1e-12 * abs(float(fx["cp_std"]))
    assert relerr(st.grad, fx["cp_grad"]) < 1e-12
    st = G(fx["tr_states"]).requires_grad_(True)
    cost = ops.PackedCost("traj", 12, dev(), target_traj=fx["tr_target"], lengthscales=fx["tr_ls"], used=None)
    c, s = ops.expected_cost(cost, st)
    c.backward()
    assert abs(float(c) - float(fx["tr_cost"])) < 1e-12 * abs(float(fx["tr_cost"]))
    assert abs(float(s) - float(fx["tr_std"])) < 1e-12 * abs(float(fx["tr_std"]))
    assert relerr(st.grad, fx["tr_grad"]) < 1e-12


def test_next_state_step(golden):
    """One get_next_state step = the fused kernel with T=2 (x1 from x0,u0 with injected eps)."""
    from gpu_helpers import G, packed_model
    from mc_pilco_amd import ops

    fx = golden("step_se")
    model = packed_model(fx, "se")
    gp0, gp1 = model.gps
    z = orc.gp_features(T(fx["x"]), T(fx["u"]), [2], [0, 1, 3])
    mu0, var0 = ops.posterior(gp0, G(z.numpy()))
    mu1, var1 = ops.posterior(gp1, G(z.numpy()))
    assert abserr(torch.cat([mu0, mu1], 1), fx["mu"]) < 1e-11
    assert abserr(torch.stack([var0, var1], 1), fx["var"]) < 1e-11


@pytest.mark.parametrize("ppw", [0, 1, 2, 4, 16, 101, 102, 104, 116, 201, 202, 204])
@pytest.mark.parametrize("name,kind", ROLLOUT_FIXTURES)
def test_rollout_cost_gradient_vs_reference(golden, name, kind, ppw):
    """apply_policy + expected cost + backward on the reference's recorded noise, on every kernel variant (gpu_helpers.forced_variant)."""
    from gpu_helpers import G, forced_variant, noise_from, packed_cost, packed_model, packed_policy
    from mc_pilco_amd import ops

    fx = golden(name)
    if ppw and name == "rollout_se_long" and ppw not in (2, 104, 204):
        pytest.skip("long rollout checked at one forced tile size per launch kind")
    model = packed_model(fx, kind)
    pol = packed_policy(fx, kind)
    cost = packed_cost(fx, kind)
    x0 = G(fx["states"][0])
    Tn = fx["states"].shape[0]
    p = float(fx["p_drop"])
    with forced_variant(ppw) as fv:
        st, inp, status = ops.rollout(model, pol, noise_from(fx), x0, Tn, p)
        c, s = ops.expected_cost(cost, st)
        c.backward()
        # (the sharded 16-particle kernel exists for G <= 3; the lean kernel takes every narrow model: SE, SE + polynomial, SOD subsets)
        fv.check(sharding_optional=(kind == "ur5" and ppw == 116), lean_expected=(kind != "ur5") if ppw >= 200 else None)
    assert int(status.item()) == 0
    long = Tn > 12
    assert abserr(st, fx["states"]) < (1e-6 if long else 1e-9)
    assert abserr(inp, fx["inputs"]) < (1e-6 if long else 1e-9)
    assert abs(float(c) - float(fx["cost"])) < (1e-8 if long else 1e-11) * abs(float(fx["cost"]))
    assert abs(float(s) - float(fx["std"])) < (1e-7 if long else 1e-10) * max(abs(float(fx["std"])), 1e-3)
    gt = 1e-6 if long else 1e-8
    assert relerr(pol.log_ls.grad, fx["g_log_ls"]) < gt
    assert relerr(pol.centers.grad, fx["g_centers"]) < gt
    assert relerr(pol.weight.grad, fx["g_weight"]) < gt


@pytest.mark.parametrize("angle_shift,B", [(0.0, 200), (2.0 * np.pi * 50000.0, 200), (0.0, 300)])
def test_rollout_vs_oracle_seeded_c1_shape(angle_shift, B):
    """Config-1 shape at a size the oracle finishes in seconds: N=300, M=64, T=20, oracle-drawn noise.  Second case: the pole angle
    starts 50 000 turns away -- beyond the range of the kernels' own sincos (|x| < 1e5), so every wave takes the library routine.
    Third case: 300 basis functions -- the backward sweep's 1024-thread form (more than 256 basis functions, 128 registers per thread)."""
    from gpu_helpers import G, dev, spec_from
    from mc_pilco_amd import ops
    from mc_pilco_amd import synthetic as sy

    c = sy.CARTPOLE
    Z, Ys = sy.gp_io(sy.cartpole_rollouts(), c["angle"], c["not_angle"], c["vel"])
    hyp = [hyper(c["lengthscales"], c["sigma_n"]) for _ in range(2)]
    caches = [orc.pretrain_gp(hyp[g], T(Z), T(Ys[g])) for g in range(2)]
    m = orc.SpeedModel(hyp, caches, c["Ts"], c["angle"], c["not_angle"], c["vel"], c["not_vel"])
    pi = sy.cartpole_policy_init(B=B)
    pp = orc.PolicyPar(torch.log(T(pi["lengthscales"])).reshape(1, -1), T(pi["centers"]), T(pi["weight"]), c["u_max"], "angles", angle=[2],
                       non_angle=[0, 1, 3])
    M, Tn, p = 64, 20, 0.25
    torch.manual_seed(5)
    e0, eps, masks = orc.draw_noise(M, 4, 2, B, Tn, p)
    x0 = orc.sample_x0(T(c["x0_mean"]), T(c["x0_var"]), M, e0)
    x0[:, 2] += angle_shift
    cost_fn = lambda st: orc.cart_pole_cost(st, T(c["cost_target"]), T(c["cost_ls"]), 2, 0)
    oc, os_, og, ost, oin = orc.policy_grad_step(m, pp, x0, Tn, cost_fn, p, eps, masks)
    # HIP path, pretrain included (Gram -> Cholesky -> inverse -> alpha on the device)
    gps = []
    for g in range(2):
        sp = spec_from(c["lengthscales"], c["sigma_n"])
        K = ops.cov_build(sp, G(Z), None, noise=True)
        U, _, stt = ops.chol_factor(K)
        assert int(stt.item()) == 0
        _, Kinv = ops.chol_inverse(U)
        alpha = ops.gp_alpha(Kinv, G(Ys[g]), 0.0)
        gps.append(ops.PackedGP(sp, G(Z), alpha, Kinv))
    model = ops.PackedModel(gps, 4, 1, c["Ts"], c["angle"], c["not_angle"], c["vel"], c["not_vel"])
    pol = ops.PackedPolicy("angles", 4, torch.log(G(pi["lengthscales"])).reshape(1, -1).requires_grad_(True), G(pi["centers"]).requires_grad_(True),
                           G(pi["weight"]).requires_grad_(True), c["u_max"], True, angle=[2], non_angle=[0, 1, 3])
    cost = ops.PackedCost("cartpole", 4, dev(), target_state=c["cost_target"], lengthscales=c["cost_ls"], angle_index=2, pos_index=0)
    nz = ops.NoiseSpec(eps=G(eps.numpy()), masks=masks.to(torch.uint8).to(dev()).contiguous())
    st, inp, status = ops.rollout(model, pol, nz, G(x0.numpy()), Tn, p)
    cc, ss = ops.expected_cost(cost, st)
    cc.backward()
    assert int(status.item()) == 0
    assert abserr(st, ost.numpy()) < 1e-7
    assert abserr(inp, oin.numpy()) < 1e-7
    assert abs(float(cc) - float(oc)) < 1e-9 * abs(float(oc))
    assert relerr(pol.centers.grad, og["centers"].numpy()) < 1e-6
    assert relerr(pol.weight.grad, og["weight"].numpy()) < 1e-6
    assert relerr(pol.log_ls.grad, og["log_ls"].numpy()) < 1e-6


def test_philox_mode_properties():
    """Performance-mode noise: reproducible for a fixed (seed, call), different across calls,
    invariant to particle sharding (global particle id), and statistically sane."""
    from gpu_helpers import G, dev, packed_model, packed_policy
    from conftest import load_golden
    from mc_pilco_amd import ops

    fx = load_golden("rollout_se")
    model = packed_model(fx, "se")
    pol = packed_policy(fx, "se", requires_grad=False)
    M, Tn, p = 512, 6, 0.25
    x0 = G(np.zeros((M, 4)))
    a = ops.rollout(model, pol, ops.NoiseSpec(seed=7, call=1), x0, Tn, p)[0]
    b = ops.rollout(model, pol, ops.NoiseSpec(seed=7, call=1), x0, Tn, p)[0]
    c = ops.rollout(model, pol, ops.NoiseSpec(seed=7, call=2), x0, Tn, p)[0]
    assert torch.equal(a, b)
    assert not torch.equal(a, c)
    lo = ops.rollout(model, pol, ops.NoiseSpec(seed=7, call=1, particle_offset=0), x0[:256], Tn, p)[0]
    hi = ops.rollout(model, pol, ops.NoiseSpec(seed=7, call=1, particle_offset=256), x0[256:], Tn, p)[0]
    assert torch.equal(torch.cat([lo, hi], 1), a)
    # all particles start at the same point: the spread after one step is the GP's predictive std
    d = a[1, :, 1] - a[1, :, 1].mean()
    assert 0.5 < float(d.std() / a[1, :, 1].std()) < 1.5 and float(a[1, :, 1].std()) > 0


@pytest.mark.parametrize("ppw", [0, 1, 2, 4, 16, 101, 102, 104, 116, 201, 202, 204])
def test_pms_rollout_cost_gradient_vs_reference(golden, ppw):
    """MC_PILCO4PMS.apply_policy + cost + backward through the C ABI (mcp_meas): the measurement filter between particles and
    policy is carried inside the fused kernels; the reference's recorded eps / position noise / masks are injected."""
    from gpu_helpers import G, dev, forced_variant, packed_cost, packed_model, packed_policy
    from mc_pilco_amd import ops

    fx = golden("rollout_pms")
    model, pol, cost = packed_model(fx, "se"), packed_policy(fx, "se"), packed_cost(fx, "se")
    pos = [int(i) for i in fx["pos_indeces"]]
    meas = ops.MeasSpec(pos=pos, vel=[int(i) for i in fx["vel_indeces"]], std_pos=[float(v) for v in fx["std_meas_noise"][pos]],
                        b=fx["butter_b"], a=fx["butter_a"], pos_noise=G(fx["pos_noise"]))
    nz = ops.NoiseSpec(eps=G(fx["eps"]), masks=torch.as_tensor(fx["masks"]).to(dev()).contiguous())
    Tn, p = fx["states"].shape[0], float(fx["p_drop"])
    with forced_variant(ppw) as fv:
        st, inp, status = ops.rollout(model, pol, nz, G(fx["x0"]), Tn, p, meas=meas)
        fv.check(lean_expected=True if ppw >= 200 else None)  # (round 4: the lean kernel carries the measurement model too)
        c, s = ops.expected_cost(cost, st)
        c.backward()
    assert int(status.item()) == 0
    assert abserr(st, fx["states"]) < 1e-9
    assert abserr(inp, fx["inputs"]) < 1e-9
    assert abs(float(c) - float(fx["cost"])) < 1e-11 * abs(float(fx["cost"]))
    assert relerr(pol.log_ls.grad, fx["g_log_ls"]) < 1e-8
    assert relerr(pol.centers.grad, fx["g_centers"]) < 1e-8
    assert relerr(pol.weight.grad, fx["g_weight"]) < 1e-8


def test_pms_philox_mode_is_reproducible_and_shard_invariant():
    """Performance-mode position noise: Philox stream counted by the global particle id, like eps and the masks."""
    from gpu_helpers import G, packed_model, packed_policy
    from conftest import load_golden
    from mc_pilco_amd import ops

    fx = load_golden("rollout_pms")
    model, pol = packed_model(fx, "se"), packed_policy(fx, "se", requires_grad=False)
    meas = ops.MeasSpec(pos=[0, 2], vel=[1, 3], std_pos=[0.01, 0.015], b=fx["butter_b"], a=fx["butter_a"])
    M, Tn = 96, 7
    x0 = G(0.01 * np.random.RandomState(0).randn(M, 4))
    a = ops.rollout(model, pol, ops.NoiseSpec(seed=5, call=2), x0, Tn, 0.25, meas=meas)
    b = ops.rollout(model, pol, ops.NoiseSpec(seed=5, call=2), x0, Tn, 0.25, meas=meas)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    h = M // 2
    lo = ops.rollout(model, pol, ops.NoiseSpec(seed=5, call=2, particle_offset=0), x0[:h].contiguous(), Tn, 0.25, meas=meas)
    hi = ops.rollout(model, pol, ops.NoiseSpec(seed=5, call=2, particle_offset=h), x0[h:].contiguous(), Tn, 0.25, meas=meas)
    assert torch.equal(torch.cat([lo[0], hi[0]], 1), a[0])
    plain = ops.rollout(model, pol, ops.NoiseSpec(seed=5, call=2), x0, Tn, 0.25)
    assert float((plain[1] - a[1]).abs().max()) > 1e-6  # the measurement noise does reach the policy
    # the 16-particle tile kernel draws the same numbers (same trajectories to rounding)
    from mc_pilco_amd import hipabi

    hipabi.lib().mcp_debug_set_particles_per_wg(16)
    try:
        t16 = ops.rollout(model, pol, ops.NoiseSpec(seed=5, call=2), x0, Tn, 0.25, meas=meas)
        assert hipabi.lib().mcp_debug_last_particles_per_wg() == 16
    finally:
        hipabi.lib().mcp_debug_set_particles_per_wg(0)
    assert float((t16[0] - a[0]).abs().max()) < 1e-9 and float((t16[1] - a[1]).abs().max()) < 1e-9


@pytest.mark.parametrize("case", [("cartpole", 0, 20, 17, 3), ("cartpole", 2, 33, 5, 2), ("cartpole", 1, 16, 1, 4), ("ur5", 1, 17, 3, 3), ("ur5", 2, 40, 19, 3),
                                  ("cartpole", 0, 130, 35, 4)])
def test_kernel_variants_agree_on_odd_shapes(case):
    """Every forward variant (1, 2, 4, 16 particles per workgroup; GP-sharded clusters of 1, 2, 4) and backward sweep width (1, 2, 4) on shapes that exercise
    the edges: N not a multiple of 16 or 32, a single 16-row block, M smaller than / not a multiple of the tile, T = 2."""
    from gpu_helpers import dev, forced_variant
    from mc_pilco_amd import ops, workloads

    workloads.CONFIGS["edge"] = case
    w = workloads.build("edge", device=dev())
    torch.manual_seed(11)
    x0 = w.sample_x0()
    ref = None
    for code in [1, 2, 4, 16, 101, 102, 104, 116, 201, 202, 204]:
        with forced_variant(code) as fv:
            for q in w.params:
                q.grad = None
            st, inp, status = ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=9, call=3), x0, w.T, w.p_drop)
            c, s = ops.expected_cost(w.cost, st)
            c.backward()
            fv.check(sharding_optional=(case[0] == "ur5"))
        assert int(status.item()) == 0
        got = (st.detach().clone(), inp.detach().clone(), [q.grad.detach().clone() for q in w.params])
        if ref is None:
            ref = got
            continue
        assert float((got[0] - ref[0]).abs().max()) < 1e-9 and float((got[1] - ref[1]).abs().max()) < 1e-9
        for ga, gb in zip(got[2], ref[2]):
            assert float((ga - gb).abs().max()) <= 1e-8 * max(1e-30, float(gb.abs().max()))


@pytest.mark.parametrize("name,M", [("c1", 400), ("c3", 4096)])
def test_full_size_properties(name, M):
    """BASELINE.json's full shapes (N=300, T=150; c1: M=400 on the small-tile kernels, c3: SE+poly(2) on the 16-particle MFMA
    tile kernel and the 4-particle backward sweep), checked through size-independent properties: bitwise determinism, shard
    invariance (two half swarms with their global particle offsets reproduce the full swarm bit for bit), and the adjoint
    gradient against a central finite difference of the expected cost along a random direction (noise held fixed)."""
    from gpu_helpers import dev
    from mc_pilco_amd import ops, workloads

    w = workloads.build(name, device=dev(), M=M)
    torch.manual_seed(21)
    x0 = w.sample_x0()
    nz = lambda off=0: ops.NoiseSpec(seed=77, call=5, particle_offset=off)

    def cost_of():
        st, inp, status = ops.rollout(w.model, w.policy, nz(), x0, w.T, w.p_drop)
        c, s = ops.expected_cost(w.cost, st)
        return st, c, status

    st_a, c_a, status = cost_of()
    st_b, c_b, _ = cost_of()
    assert int(status.item()) == 0
    assert torch.equal(st_a, st_b) and torch.equal(c_a.detach(), c_b.detach())
    h = M // 2
    lo = ops.rollout(w.model, w.policy, nz(0), x0[:h].contiguous(), w.T, w.p_drop)[0]
    hi = ops.rollout(w.model, w.policy, nz(h), x0[h:].contiguous(), w.T, w.p_drop)[0]
    # (the halves may run another launch form than the whole -- c1: clusters of 2 instead of 4 particles; c3: the GP-sharded instead of
    #  the unsharded 16-particle kernel -- every form adds a GP's sums in the same order, so they reproduce the whole bit for bit)
    assert torch.equal(torch.cat([lo, hi], 1), st_a)
    for q in w.params:
        q.grad = None
    c_a.backward()
    g = [q.grad.detach().clone() for q in w.params]
    gen = torch.Generator(device=dev())
    gen.manual_seed(5)
    dirs = [torch.randn(q.shape, dtype=q.dtype, device=q.device, generator=gen) for q in w.params]
    gd = sum(float((a * b).sum()) for a, b in zip(g, dirs))
    eps = 1e-6
    with torch.no_grad():
        for q, d in zip(w.params, dirs):
            q.add_(eps * d)
        cp = float(cost_of()[1])
        for q, d in zip(w.params, dirs):
            q.sub_(2 * eps * d)
        cm = float(cost_of()[1])
        for q, d in zip(w.params, dirs):
            q.add_(eps * d)
    fd = (cp - cm) / (2 * eps)
    # over 150 steps the expected cost is violently nonlinear in the policy parameters (the difference quotient changes sign
    # at eps = 1e-4 and is itself only good to ~1e-3): a coarse bound here, the tight one on the 40-step horizon below
    assert abs(fd - gd) < 5e-3 * max(abs(gd), 1e-3), (fd, gd)


@pytest.mark.parametrize("name,M", [("c1", 400), ("c3", 2048)])
def test_adjoint_matches_finite_difference_at_full_width(name, M):
    """Same swarms and training sets as above on a 40-step horizon, where a central difference is accurate: the adjoint
    gradient's directional derivative agrees to 1e-5."""
    from gpu_helpers import dev
    from mc_pilco_amd import ops, workloads

    w = workloads.build(name, device=dev(), M=M, T=40)
    torch.manual_seed(21)
    x0 = w.sample_x0()

    def cost_of():
        st, inp, status = ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=77, call=5), x0, w.T, w.p_drop)
        return ops.expected_cost(w.cost, st)[0]

    for q in w.params:
        q.grad = None
    cost_of().backward()
    g = [q.grad.detach().clone() for q in w.params]
    gen = torch.Generator(device=dev())
    gen.manual_seed(5)
    dirs = [torch.randn(q.shape, dtype=q.dtype, device=q.device, generator=gen) for q in w.params]
    gd = sum(float((a * b).sum()) for a, b in zip(g, dirs))
    eps = 1e-6
    with torch.no_grad():
        for q, d in zip(w.params, dirs):
            q.add_(eps * d)
        cp = float(cost_of())
        for q, d in zip(w.params, dirs):
            q.sub_(2 * eps * d)
        cm = float(cost_of())
    fd = (cp - cm) / (2 * eps)
    assert abs(fd - gd) < 1e-5 * max(abs(gd), 1e-3), (fd, gd)


@pytest.mark.parametrize("name,M,expect,sharded", [("c1", 400, 4, True), ("c1", 256, 2, True), ("c1", 1000, 16, True), ("c1", 2048, 16, True), ("c1", 2100, 16, False), ("c3", 4000, 16, False),
                                                   ("c5", 2000, 16, True)])
def test_baseline_shapes_run_on_the_intended_kernel(name, M, expect, sharded):
    """The automatic dispatch puts BASELINE.json's shapes where DESIGN.md says they run (a shape that overflows the tile kernel's
    LDS budget would silently fall back to the 4-particle kernel; a swarm beyond one resident GP-sharded grid of the small-tile kernel runs on the
    GP-sharded 16-particle kernel up to 2048 particles, beyond that unsharded)."""
    from gpu_helpers import dev
    from mc_pilco_amd import hipabi, ops, workloads

    w = workloads.build(name, device=dev(), M=M, T=3)
    st, inp, status = ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=1, call=1), w.sample_x0(), w.T, w.p_drop)
    assert int(status.item()) == 0
    assert hipabi.lib().mcp_debug_last_particles_per_wg() == expect
    assert bool(hipabi.lib().mcp_debug_last_gp_sharded()) == sharded


def test_gp_sharded_launch_needs_its_workspace():
    """Without the hand-off workspace the C ABI must not shard (and must still run on the device)."""
    from gpu_helpers import dev
    from mc_pilco_amd import hipabi, ops, workloads
    import ctypes as C

    w = workloads.build("tiny", device=dev())
    x0 = w.sample_x0()
    M, T = x0.shape[0], w.T
    pc = w.policy.bind(w.p_drop)
    nz = ops.NoiseSpec(seed=3, call=1).to_c()
    out = []
    for with_ws in (True, False):
        states = torch.empty(T, M, w.policy.S, dtype=torch.float64, device=dev())
        inputs = torch.empty(T, M, w.policy.U, dtype=torch.float64, device=dev())
        status = torch.zeros(1, dtype=torch.int32, device=dev())
        nbytes = hipabi.lib().mcp_rollout_workspace_bytes(C.byref(w.model.c), C.byref(pc), M, T)
        ws = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=dev())
        # (the `_ex` entry point: the plain call plus the report of what was launched)
        rc = hipabi.lib().mcp_rollout_fwd_ex(C.byref(w.model.c), C.byref(pc), C.byref(nz), M, T, 1, hipabi.ptr(x0), hipabi.ptr(states), hipabi.ptr(inputs),
                                             None, hipabi.ptr(status), hipabi.ptr(ws) if with_ws else None, nbytes if with_ws else 0, hipabi.stream(),
                                             C.byref(hipabi.DISPATCH))
        assert rc == 0 and int(status.item()) == 0
        assert bool(hipabi.lib().mcp_debug_last_gp_sharded()) == with_ws
        out.append(states.clone())
    assert float((out[0] - out[1]).abs().max()) < 1e-9


def test_two_launch_sharding_matches_the_unsharded_kernels():
    """A swarm beyond one resident GP-sharded grid of the small-tile kernel can go out as two sharded launches over disjoint particle
    ranges (the fallback where the sharded 16-particle kernel does not apply; forced here): same trajectories and gradient as the unsharded kernels, Philox noise keyed by the global particle index."""
    from gpu_helpers import dev
    from mc_pilco_amd import hipabi, ops, workloads

    w = workloads.build("c1", device=dev(), M=700, T=10)
    torch.manual_seed(5)
    x0 = w.sample_x0()
    out = []
    for mode in (1, 0):
        hipabi.lib().mcp_debug_set_gp_sharding(mode)
        hipabi.lib().mcp_debug_set_particles_per_wg(4 if mode == 1 else 0)  # (the automatic choice at this size is the sharded 16-particle kernel)
        try:
            for q in w.params:
                q.grad = None
            st, inp, status = ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=4, call=2), x0, w.T, w.p_drop)
            launches = hipabi.lib().mcp_debug_last_gp_sharded()
            c, s = ops.expected_cost(w.cost, st)
            c.backward()
        finally:
            hipabi.lib().mcp_debug_set_gp_sharding(-1)
            hipabi.lib().mcp_debug_set_particles_per_wg(0)
        assert int(status.item()) == 0
        assert launches == (2 if mode == 1 else 0)
        out.append((st.detach().clone(), inp.detach().clone(), [q.grad.detach().clone() for q in w.params]))
    # (the two kernels add the V partial sums in different orders: rounding-level differences, grown by 10 steps of dynamics
    #  over the most sensitive of 700 particles)
    assert float((out[0][0] - out[1][0]).abs().max()) < 2e-8 and float((out[0][1] - out[1][1]).abs().max()) < 2e-8
    for ga, gb in zip(out[0][2], out[1][2]):
        assert float((ga - gb).abs().max()) <= 1e-7 * max(1e-30, float(gb.abs().max()))


def test_posterior_operator_takes_more_than_1024_test_points(golden):
    """GP_prior.get_estimate_from_alpha on a large batch of test inputs (round 2: the operator refused M > 1024)."""
    from gpu_helpers import G
    from mc_pilco_amd import ops

    fx = golden("kern_se")
    sp = _kernel_spec(fx)
    gp = ops.PackedGP(sp, G(fx["X"]), G(fx["alpha"]), G(fx["Kinv"]))
    Xs = G(np.tile(fx["Xs"], (40, 1)))  # 1280 points
    mu, var = ops.posterior(gp, Xs)
    assert relerr(mu[:32], fx["mu"]) < 1e-10 and relerr(mu[-32:], fx["mu"]) < 1e-10
    assert abserr(var[-32:], fx["var"]) < 1e-10


@pytest.mark.parametrize("ppw", [0, 1, 2, 4, 16, 101, 102, 104, 116, 201, 202, 204])
def test_rollout_with_policy_bias_vs_reference(golden, ppw):
    """flg_bias (Policy.py:203-212) through the C ABI: mcp_policy.bias enters the linear layer in every forward variant, the adjoint
    sweep returns dJ/dbias (mcp_policy.g_bias) next to the three other gradients."""
    from gpu_helpers import G, dev, forced_variant, noise_from, packed_cost, packed_model
    from mc_pilco_amd import ops
    from mc_pilco_amd import synthetic as sy

    fx = golden("rollout_bias")
    model, cost = packed_model_from_training(fx), packed_cost(fx, "se")
    c = sy.CARTPOLE
    prm = [torch.log(G(fx["pol_ls"])).reshape(1, -1).requires_grad_(True), G(fx["pol_centers"]).requires_grad_(True),
           G(fx["pol_weight"]).requires_grad_(True), G(fx["pol_bias"]).requires_grad_(True)]
    pol = ops.PackedPolicy("angles", c["S"], prm[0], prm[1], prm[2], c["u_max"], True, angle=[2], non_angle=[0, 1, 3], bias=prm[3])
    with forced_variant(ppw) as fv:
        st, inp, status = ops.rollout(model, pol, noise_from(fx), G(fx["states"][0]), fx["states"].shape[0], float(fx["p_drop"]))
        cc, ss = ops.expected_cost(cost, st)
        cc.backward()
        fv.check()
    assert int(status.item()) == 0
    assert abserr(st, fx["states"]) < 1e-8 and abserr(inp, fx["inputs"]) < 1e-8
    assert abs(float(cc) - float(fx["cost"])) < 1e-10 * abs(float(fx["cost"]))
    for q, k in zip(prm, ["g_log_ls", "g_centers", "g_weight", "g_bias"]):
        assert relerr(q.grad.reshape(fx[k].shape), fx[k]) < 1e-7, k


def packed_model_from_training(fx):
    """HIP pretrain (Gram -> Cholesky -> inverse -> alpha) on a fixture's training data, cart-pole speed model."""
    from gpu_helpers import G, spec_from
    from mc_pilco_amd import ops
    from mc_pilco_amd import synthetic as sy

    c = sy.CARTPOLE
    Z, Ys = orc.speed_model_io(fx["states_tr"], fx["inputs_tr"], c["angle"], c["not_angle"], c["vel"])
    gps = []
    for g in range(2):
        sp = spec_from(c["lengthscales"], float(fx["sigma_n"]))
        K = ops.cov_build(sp, G(Z.numpy()), None, noise=True)
        U, _, stt = ops.chol_factor(K)
        assert int(stt.item()) == 0
        _, Kinv = ops.chol_inverse(U)
        gps.append(ops.PackedGP(sp, G(Z.numpy()), ops.gp_alpha(Kinv, G(Ys[g].numpy()), 0.0), Kinv))
    return ops.PackedModel(gps, 4, 1, c["Ts"], c["angle"], c["not_angle"], c["vel"], c["not_vel"])


def test_recovery_path_runs_unsharded():
    """The step that follows an MCP_STATUS_SYNC is repeated with gp_sharding=False (no hand-off workspace is passed): the library
    must then launch unsharded, and give the same trajectories (to rounding) as the GP-sharded launch the dispatch would pick."""
    from gpu_helpers import dev
    from mc_pilco_amd import hipabi, ops, workloads

    w = workloads.build("c1", device=dev(), M=400, T=12)
    torch.manual_seed(2)
    x0 = w.sample_x0()
    with torch.no_grad():
        a, ua, sa = ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=6, call=1), x0, w.T, w.p_drop)
        assert hipabi.lib().mcp_debug_last_gp_sharded() >= 1
        b, ub, sb = ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=6, call=1), x0, w.T, w.p_drop, gp_sharding=False)
        assert hipabi.lib().mcp_debug_last_gp_sharded() == 0
    assert int(sa.item()) == 0 and int(sb.item()) == 0
    assert float((a - b).abs().max()) < 1e-8 and float((ua - ub).abs().max()) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("N,M", [(40, 48), (100, 64)])
def test_wide_class_phase_j_forms_agree(N, M):
    """UR5-shaped models (two row tiles of [X^T; 1]) run phase J one output tile per wave from the packed operand copy in the workspace;
    without a workspace (gp_sharding=False) the library keeps the split-j form.  Same trajectories, Jacobian-fed gradients and status."""
    from gpu_helpers import dev, forced_variant
    from mc_pilco_amd import ops, workloads

    w = workloads.build("tiny_ur5", device=dev(), N=N, M=M, T=6)
    torch.manual_seed(5)
    x0 = w.sample_x0()
    outs = []
    for sharding in (True, False):
        for p in w.params:
            p.grad = None
        with forced_variant(16) as fv:
            st, inp, status = ops.rollout(w.model, w.policy, ops.NoiseSpec(seed=9, call=2), x0, w.T, w.p_drop, gp_sharding=sharding)
            fv.check()
            c, _ = ops.expected_cost(w.cost, st)
            c.backward()
        assert int(status.item()) == 0
        outs.append((st.detach().clone(), inp.detach().clone(), [p.grad.detach().clone() for p in w.params]))
    (sa, ua, ga), (sb, ub, gb) = outs
    assert float((sa - sb).abs().max()) < 1e-9 and float((ua - ub).abs().max()) < 1e-9
    for x, y in zip(ga, gb):
        assert float((x - y).abs().max()) <= 1e-8 * (1.0 + float(y.abs().max()))


@pytest.mark.gpu
def test_wide_class_with_a_different_subset_size_per_gp():
    """Subset-of-data pretraining keeps a different number of points per GP (Model_learning.py:176-199): the packed phase-J operands are
    laid out on the largest Npad, every GP runs its own batch count.  All forward variants agree on such a model."""
    from gpu_helpers import dev, forced_variant
    from mc_pilco_amd import ops, workloads

    w = workloads.build("tiny_ur5", device=dev(), N=80, M=32, T=5)
    pb, c = w.problem, w.problem["cfg"]
    sizes = [40, 56, 80, 72, 48, 64]
    gps = []
    for g in range(c["G"]):
        spec = workloads.spec_for(c, c["sigma_n"], None if pb["poly"] is None else pb["poly"][g])
        gps.append(workloads.pretrain_packed(spec, pb["Z"][: sizes[g]], pb["Ys"][g][: sizes[g]], dev()))
    model = ops.PackedModel(gps, c["S"], c["U"], c["Ts"], c["angle"], c["not_angle"], c["vel"], c["not_vel"])
    torch.manual_seed(8)
    x0 = w.sample_x0()
    outs = {}
    with torch.no_grad():
        for code, sharding in ((16, True), (16, False), (4, True), (1, True)):
            with forced_variant(code) as fv:
                st, inp, status = ops.rollout(model, w.policy, ops.NoiseSpec(seed=4, call=1), x0, w.T, w.p_drop, gp_sharding=sharding)
                fv.check()
            assert int(status.item()) == 0
            outs[(code, sharding)] = (st.clone(), inp.clone())
    ref = outs[(1, True)]
    for key, (st, inp) in outs.items():
        assert float((st - ref[0]).abs().max()) < 1e-9, key
        assert float((inp - ref[1]).abs().max()) < 1e-9, key


@pytest.mark.gpu
@pytest.mark.parametrize("case", [("c1", 400, 150, 0.25, "philox"), ("c1", 37, 2, 0.25, "philox"), ("c1", 64, 9, 0.0, "philox"), ("c1", 48, 12, 0.25, "masks"),
                                  ("c3", 96, 20, 0.25, "philox"), ("c1", 1333, 7, 0.25, "philox"), ("c1", 600, 5, 0.25, "masks"),
                                  ("pms_script", 400, 90, 0.25, "philox"), ("pms_script", 37, 2, 0.25, "philox"), ("pms_script", 64, 11, 0.0, "philox"),
                                  ("pms_script", 515, 6, 0.25, "philox")])
def test_lean_backward_sweep_matches_the_general_one(case):
    """rollout_bwd_lat_kernel (small swarms: wave 0 runs the adjoint chain from registers, the RBF waves prepare their step ahead of the
    barrier) against the general sweep on the same rollout: all three policy gradients and dJ/dx0 to 1e-11 relative (different
    summation order only), with in-kernel dropout bits, with mask buffers, without dropout, at T = 2, at the headline size, and beyond 512
    particles, where 256 workgroups walk several particles per slot (an uneven number of rounds at M = 1333 and 600).  Round 4: the same
    with the measurement model of MC_PILCO4PMS between particles and policy (the filter's adjoint recursion carried in the chain's
    registers) at the launch script's size (M = 400, T = 90), at T = 2 (only the t = 0 special case and one filtered step), without
    dropout, and beyond one round."""
    from gpu_helpers import dev
    from mc_pilco_amd import hipabi, ops, workloads

    name, M, Tn, p, mode = case
    w = workloads.build(name, device=dev(), M=M, T=Tn, p_drop=p)
    torch.manual_seed(3)
    x0 = w.sample_x0().requires_grad_(True)
    if mode == "masks":
        B = w.policy.centers.shape[0]
        nz = ops.NoiseSpec(eps=torch.randn(Tn - 1, M, len(w.model.gps), dtype=torch.float64, device=dev()),
                           masks=(torch.rand(Tn, M, B, device=dev()) >= p).to(torch.uint8).contiguous())
    else:
        nz = ops.NoiseSpec(seed=21, call=4)
    L = hipabi.lib()
    res = {}
    try:
        for lean in (1, 0):
            L.mcp_debug_set_bwd_lean(-1 if lean else 0)
            for q in w.params:
                q.grad = None
            x0.grad = None
            st, inp, status = ops.rollout(w.model, w.policy, nz, x0, w.T, w.p_drop, meas=w.meas)
            c, s = ops.expected_cost(w.cost, st)
            (c + 0.3 * s + 1e-3 * (inp ** 2).sum()).backward()  # (a cost on the inputs too: dJ/du_t enters the chain)
            assert L.mcp_debug_last_bwd_lean() == lean
            res[lean] = [q.grad.detach().clone() for q in w.params] + [x0.grad.detach().clone()]
    finally:
        L.mcp_debug_set_bwd_lean(-1)
    for ga, gb in zip(res[1], res[0]):
        assert torch.isfinite(gb).all()
        assert float((ga - gb).abs().max()) <= 1e-11 * float(gb.abs().max()), (float((ga - gb).abs().max()), float(gb.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("G_,angles", [(3, False), (3, True), (4, False)])
def test_lean_kernels_on_models_with_three_and_four_gps(G_, angles):
    """The latency-lean forward and backward kernels on narrow models with more than two GPs (their 4-GP instantiations: clusters of G
    workgroups meeting in the hand-off forward, rollout_bwd_lat_kernel<4> backward), against the general kernels on the same rollout:
    a synthetic G-joint system (G positions + G velocities, one input; optionally the first position an angle), plain policy."""
    from gpu_helpers import G, dev, forced_variant, spec_from
    from mc_pilco_amd import hipabi, ops

    rng = np.random.RandomState(40 + G_)
    S, U, N, B, M, Tn, p = 2 * G_, 1, 48, 40, 24, 9, 0.25
    angle = [0] if angles else []
    not_angle = [i for i in range(S) if i not in angle]
    D = len(not_angle) + 2 * len(angle) + U
    fwd_lean_ok = D <= 7 and D - U <= 6  # (the forward kernel's lane roles and the ones row of its phase-J operand; the backward one takes all three shapes)
    Z = rng.randn(N, D) * 0.8
    gps = []
    for g in range(G_):
        sp = spec_from(1.0 + rng.rand(D), 0.05, lam=0.5 + 0.1 * g)
        K = ops.cov_build(sp, G(Z), None, noise=True)
        Uc, _, stt = ops.chol_factor(K)
        assert int(stt.item()) == 0
        _, Kinv = ops.chol_inverse(Uc)
        alpha = ops.gp_alpha(Kinv, G(0.05 * rng.randn(N)), 0.0)
        gps.append(ops.PackedGP(sp, G(Z), alpha, Kinv))
    model = ops.PackedModel(gps, S, U, 0.05, angle, not_angle, list(range(G_, 2 * G_)), list(range(G_)))
    mk = lambda: ops.PackedPolicy("plain", S, G(np.zeros((1, S))).requires_grad_(True), G(rng.randn(B, S)).requires_grad_(True),
                                  G(0.3 * rng.randn(U, B)).requires_grad_(True), [2.0], True)
    rs = rng.get_state()
    x0 = G(0.3 * rng.randn(M, S))
    nz = ops.NoiseSpec(seed=5, call=2)
    L = hipabi.lib()
    res = {}
    try:
        for lean in (1, 0):
            rng.set_state(rs)
            pol = mk()
            L.mcp_debug_set_bwd_lean(-1 if lean else 0)
            with forced_variant(0 if lean else 104):
                x0v = x0.clone().requires_grad_(True)
                st, inp, status = ops.rollout(model, pol, nz, x0v, Tn, p)
                ((st ** 2).mean() + 0.1 * (inp ** 2).mean()).backward()
                assert bool(L.mcp_debug_last_fwd_lean()) == bool(lean and fwd_lean_ok) and bool(L.mcp_debug_last_bwd_lean()) == bool(lean)
            assert int(status.item()) == 0
            res[lean] = [st.detach().clone(), inp.detach().clone(), pol.log_ls.grad.clone(), pol.centers.grad.clone(), pol.weight.grad.clone(),
                         x0v.grad.clone()]
    finally:
        L.mcp_debug_set_bwd_lean(-1)
    for a_, b_ in zip(res[1], res[0]):
        assert torch.isfinite(b_).all()
        assert float((a_ - b_).abs().max()) <= 1e-10 * max(1e-30, float(b_.abs().max())), (float((a_ - b_).abs().max()), float(b_.abs().max()))
