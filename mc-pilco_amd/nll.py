"""Marginal-likelihood loss and analytic gradient for GP hyper-parameter training on the device.

Replaces what autograd does in the reference's ``GP_prior.fit_model`` (gpr_lib/GP_prior/GP_prior.py:179-230) through
``forward`` (Gram, Cholesky, inverse) and ``Marginal_log_likelihood`` (gpr_lib/Likelihood/Gaussian_likelihood.py:15-24):
    L = 1/2 ((Y-m)^T K^-1 (Y-m) + log det K),      dL/dtheta = 1/2 tr((K^-1 - a a^T) dK/dtheta),  a = K^-1 (Y-m).
The HIP kernel ``mcp_nll_grad`` returns the gradient w.r.t. the kernel's log-parameters; this module scatters it into
``param.grad`` of the GP object's own parameters (RBF: log_lengthscales_par, log_lambda_par, sigma_n_log, mean_par;
MPK_k: Sigma_pos_par), honouring ``requires_grad``.
"""
import ctypes as C

import torch

from . import hipabi as abi
from . import ops

DT = torch.float64


def _leaves(gp):
    from .gpr_lib.GP_prior.GP_prior import Combine_GP

    return list(gp._leaves()) if isinstance(gp, Combine_GP) else [gp]


def nll_loss_and_grad(gp, X, Y):
    """Returns the loss (0-dim tensor) and sets ``.grad`` of every trainable parameter of ``gp``."""
    from .gpr_lib.GP_prior import Sparse_GP, Stationary_GP

    dev = gp.device
    spec = gp.kernel_spec_dev()  # hyper-parameters stay on the device: the epoch has no device->host round trip
    Xc = gp._cols(X)
    N, D = Xc.shape
    K = ops.cov_build(spec, Xc, None, noise=gp.GP_with_noise)
    U, logdet, status = ops.chol_factor(K)
    # the not-positive-definite flag is checked by the caller at its next host sync point (``check_status``): reading it
    # here would stall the launch queue in the middle of every epoch
    prev = getattr(gp, "_nll_status", None)
    gp._nll_status = status if prev is None else torch.bitwise_or(prev, status)
    _, Kinv = ops.chol_inverse(U)
    r = (Y.to(dev) - gp.get_mean(X)).reshape(-1).contiguous()
    alpha = ops.gp_alpha(Kinv, r, 0.0).reshape(-1).contiguous()
    loss = 0.5 * (torch.dot(r, alpha) + logdet)
    nbytes = abi.lib().mcp_nll_workspace_bytes(N, D)
    ws = torch.empty((nbytes + 7) // 8, dtype=DT, device=dev)
    g = torch.empty(4 * D + 3, dtype=DT, device=dev)
    kc = spec.to_c(dev)
    abi.check(abi.lib().mcp_nll_grad(C.byref(kc), N, abi.ptr(Xc), abi.ptr(Kinv), N, abi.ptr(alpha), abi.ptr(g), abi.ptr(ws), nbytes,
                                     abi.stream()), "mcp_nll_grad")

    for p, val in kernel_param_grads(gp, g, D, spec=spec):
        p.grad = val
    first = _leaves(gp)[0]
    for leaf in _leaves(gp):  # only the first child's mean enters the model (GP_prior.py:306-312)
        if hasattr(leaf, "mean_par") and isinstance(leaf, Stationary_GP.RBF) and leaf.mean_par.requires_grad:
            leaf.mean_par.grad = (-alpha.sum()).reshape(leaf.mean_par.shape).to(leaf.mean_par.dtype).clone() if leaf is first else torch.zeros_like(leaf.mean_par)
    return loss.detach()


def kernel_param_grads(gp, g, D, spec=None):
    """[(parameter, gradient)] of the trainable KERNEL hyper-parameters of ``gp`` from the library's gradient vector ``g`` (layout of
    mcp_nll_grad: [0, D) log lengthscales | D log lambda | D + 1 noise variance | MPK_1 (D + 1) | MPK_2 factor 0 (D) | factor 1 (D)).
    ``spec``: the kernel descriptor ``g`` was evaluated with (default: the current one) -- a plain Linear_GP leaf needs its w1."""
    from .gpr_lib.GP_prior import Sparse_GP, Stationary_GP

    out = []

    def put(p, val):
        if p.requires_grad:
            out.append((p, val.reshape(p.shape).to(p.dtype).clone()))

    for leaf in _leaves(gp):
        if leaf.GP_with_noise:
            put(leaf.sigma_n_log, g[D + 1] * 2.0 * torch.exp(2.0 * leaf.sigma_n_log.detach()))
        if isinstance(leaf, Stationary_GP.RBF):
            put(leaf.log_lengthscales_par, g[0:D] if leaf.flg_ARD else g[0:D].sum())
            put(leaf.log_lambda_par, g[D])
        elif isinstance(leaf, Sparse_GP.MPK_GP):
            if leaf.poly_deg == 1:
                put(leaf.Sigma_pos_par, g[D + 2:2 * D + 3] if leaf.flg_offset else g[D + 2:2 * D + 2])
            else:
                put(leaf.Sigma_pos_par, torch.cat([g[2 * D + 3:3 * D + 3], g[3 * D + 3:4 * D + 3]]))
        elif isinstance(leaf, Sparse_GP.Linear_GP):
            # phi^T diag(w) phi' rides in the kernels' degree-1 slot: the library's entries are d/d log w_c = 2 w_c dL/dw_c of the TOTAL weight
            # vector; dL/dw_c goes on to this leaf's own parameters through its Sigma_function by autograd (w = diag(Sigma), torch ops)
            ps = [q for q in (leaf.Sigma_pos_par, leaf.Sigma_free_par) if q is not None and q.requires_grad]
            if ps:
                n = D + 1 if leaf.flg_offset else D
                w_tot = (gp.kernel_spec_dev() if spec is None else spec).w1[:n].to(g.device)
                dLdw = torch.where(w_tot != 0, g[D + 2:D + 2 + n] / (2.0 * w_tot), torch.zeros_like(w_tot))
                with torch.enable_grad():
                    w = torch.diag(leaf.get_Sigma()).to(DT)
                    got = torch.autograd.grad(w, ps, grad_outputs=dLdw, allow_unused=True)
                for q, v in zip(ps, got):
                    put(q, torch.zeros_like(q) if v is None else v)
        else:
            raise NotImplementedError("no gradient mapping for the hyper-parameters of a %s leaf on the HIP path" % type(leaf).__name__)
    return out


def cov_weighted_grad(gp, Xc, Wm, spec=None):
    """sum_ij Wm_ij dK_ij/dtheta for every kernel log-parameter (the library's vector, layout above): the chain rule's last step for ANY
    function of the Gram matrix.  Through mcp_nll_grad, which evaluates 1/2 sum_ij (Kinv - a a^T)_ij dK_ij/dtheta: called with 2 Wm in the
    place of Kinv and a = 0.  ``spec``: the descriptor of the forward pass this is the backward of (default: the current hyper-parameters)."""
    dev = gp.device
    N, D = Xc.shape
    spec = gp.kernel_spec_dev() if spec is None else spec
    nbytes = abi.lib().mcp_nll_workspace_bytes(N, D)
    ws = torch.empty((nbytes + 7) // 8, dtype=DT, device=dev)
    g = torch.empty(4 * D + 3, dtype=DT, device=dev)
    W2 = (2.0 * Wm).to(DT).contiguous()
    zero = torch.zeros(N, dtype=DT, device=dev)
    kc = spec.to_c(dev)
    abi.check(abi.lib().mcp_nll_grad(C.byref(kc), N, abi.ptr(Xc), abi.ptr(W2), N, abi.ptr(zero), abi.ptr(g), abi.ptr(ws), nbytes, abi.stream()),
              "mcp_nll_grad")
    return g


def check_status(gp):
    """Raises if any Cholesky since the last check met a matrix that is not positive definite (torch.cholesky's error in the
    reference, GP_prior.py:106)."""
    st = getattr(gp, "_nll_status", None)
    gp._nll_status = None
    if st is not None and ops.status_flags(st)["not_spd"]:
        raise RuntimeError("cholesky: the covariance matrix is not positive-definite")


# --------------------------------------------------------------------------------------------------------------------------------
# batched training: the G GPs of a model, epoch by epoch, two C calls per epoch (mcp_nll_epoch + mcp_adam_step_guarded)
# --------------------------------------------------------------------------------------------------------------------------------
def plain_adam_options(opt):
    """(lr, beta1, beta2, eps) when ``opt`` is a torch.optim.Adam whose update is the textbook one (what every launch script builds:
    "lambda p : torch.optim.Adam(p, lr=0.01)"); None for anything else."""
    if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
        return None
    g = opt.param_groups[0]
    if (g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False) or g.get("differentiable", False)
            or g.get("decoupled_weight_decay", False) or isinstance(g["lr"], torch.Tensor)):
        return None
    return float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])


def _structure(gp):
    """(rbf, mpk1, mpk2, degree) when ``gp`` is one squared-exponential term (+ MPK_1 with the offset feature (+ MPK_2 without)) whose
    noise comes from the squared-exponential term alone -- every model the launch scripts build; None otherwise."""
    from .gpr_lib.GP_prior import Sparse_GP, Stationary_GP

    leaves = _leaves(gp)
    if not leaves or not isinstance(leaves[0], Stationary_GP.RBF):
        return None
    rbf, mpk1, mpk2 = leaves[0], None, None
    for leaf in leaves[1:]:
        if not isinstance(leaf, Sparse_GP.MPK_GP) or leaf.GP_with_noise:
            return None
        if leaf.poly_deg == 1 and leaf.flg_offset and mpk1 is None:
            mpk1 = leaf
        elif leaf.poly_deg == 2 and not leaf.flg_offset and mpk2 is None:
            mpk2 = leaf
        else:
            return None
    if mpk2 is not None and mpk1 is None:
        return None
    for leaf in leaves:
        ad = leaf._active()
        if ad is None or not bool((ad.detach().cpu() == torch.arange(ad.numel())).all()) or ad.numel() != rbf.num_features:
            return None
    return rbf, mpk1, mpk2, (0 if mpk1 is None else (1 if mpk2 is None else 2))


class BatchedFit:
    """Epoch-synchronous training of G structurally identical GPs on shared inputs (they are independent: the result is what training
    them one after the other gives, Model_learning.py:149-161).  ``eligible`` says whether the batched path applies; ``run`` trains
    and returns, per GP, the text ``GP_prior.fit_model`` would have printed."""

    def __init__(self, gps, X, Ys, y_scales, optimizers, N_epoch, N_epoch_print):
        self.gps, self.N_epoch, self.N_epoch_print = list(gps), int(N_epoch), int(N_epoch_print)
        self.eligible = False
        st = [_structure(g) for g in self.gps]
        ad = [plain_adam_options(o) for o in optimizers]
        if any(v is None for v in st) or any(a is None or a != ad[0] for a in ad) or len(self.gps) > abi.MAX_GP:
            return
        # the caller's optimizer is replaced, not driven: that is only the same thing when it holds exactly this GP's trainable
        # parameters and has not stepped yet (a subset would train too much here, a used one would lose its moments)
        for g, o in zip(self.gps, optimizers):
            want = {id(p) for p in g.parameters() if p.requires_grad}
            have = {id(p) for p in o.param_groups[0]["params"] if p.requires_grad}
            if want != have or len(o.state) != 0:
                return
        rbf0 = st[0][0]
        if any(v[3] != st[0][3] or v[0].flg_ARD != rbf0.flg_ARD or v[0].num_features != rbf0.num_features for v in st):
            return
        dev = rbf0.device
        Xc = self.gps[0]._cols(X)
        N, D = Xc.shape
        if not (16 < N <= 1152) or dev.type != "cuda":
            return
        self.adam, self.dev, self.X, self.N, self.D, self.deg, self.ard = ad[0], dev, Xc, N, D, st[0][3], int(rbf0.flg_ARD)
        self.Ys = [Y.to(dev).to(DT).reshape(-1).contiguous() for Y in Ys]
        self.loss = torch.zeros(len(self.gps), dtype=DT, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        desc = (abi.NllGP * len(self.gps))()
        self.params, self.grads = [], []

        def slot(p):
            """(data pointer, gradient buffer pointer or None) of parameter ``p``; trainable ones join the optimizer's list."""
            if p is None:
                return None, None
            if p.dtype != DT or not p.is_cuda or not p.is_contiguous():
                raise RuntimeError("GP hyper-parameters must be contiguous float64 GPU tensors")
            if not p.requires_grad:
                return p.data_ptr(), None
            g = torch.zeros_like(p)
            self.params.append(p)
            self.grads.append(g)
            return p.data_ptr(), g.data_ptr()

        for i, (gp, (rbf, m1, m2, _)) in enumerate(zip(self.gps, st)):
            d = desc[i]
            d.log_ls, d.g_log_ls = slot(rbf.log_lengthscales_par)
            d.log_lambda, d.g_log_lambda = slot(rbf.log_lambda_par)
            d.sigma_n_log, d.g_sigma_n_log = slot(rbf.sigma_n_log if rbf.GP_with_noise else None)
            d.mean, d.g_mean = slot(rbf.mean_par)
            d.mpk1, d.g_mpk1 = slot(None if m1 is None else m1.Sigma_pos_par)
            d.mpk2, d.g_mpk2 = slot(None if m2 is None else m2.Sigma_pos_par)
            d.Y = self.Ys[i].data_ptr()
            d.y_scale = float(y_scales[i])
            d.sigma_n_num2 = float(rbf.sigma_n_num) ** 2 if rbf.GP_with_noise else 0.0
            d.loss = self.loss[i:i + 1].data_ptr()
        self.desc = desc
        nbytes = abi.lib().mcp_nll_epoch_workspace_bytes(len(self.gps), N, D)
        self.ws, self.nbytes = torch.empty((nbytes + 7) // 8, dtype=DT, device=dev), nbytes
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.eligible = True

    def run(self):
        import io
        import time

        lib, n = abi.lib(), len(self.params)
        # Adam over all trainable tensors of all GPs, at most OPT_MAX_TENSORS per launch
        chunks = []
        for a in range(0, n, abi.OPT_MAX_TENSORS):
            b = min(n, a + abi.OPT_MAX_TENSORS)
            arr = lambda ts: (abi.dptr * (b - a))(*[t.data_ptr() for t in ts[a:b]])
            chunks.append((b - a, arr([p.data for p in self.params]), arr(self.grads), arr(self.m), arr(self.v),
                           (C.c_int64 * (b - a))(*[p.numel() for p in self.params[a:b]])))
        lr, b1, b2, eps = self.adam
        snaps = []  # (epoch, [state_dict clones per GP], losses clone, wall time) at the print epochs

        def snapshot(epoch):
            snaps.append((epoch, [{k: v.detach().clone() for k, v in gp.named_parameters()} for gp in self.gps], self.loss.clone(), time.time()))

        snapshot(-1)  # the initial parameters
        for epoch in range(self.N_epoch):
            abi.check(lib.mcp_nll_epoch(len(self.gps), C.cast(self.desc, C.c_void_p), self.N, self.D, self.deg, self.ard, abi.ptr(self.X),
                                        abi.ptr(self.status), abi.ptr(self.ws), self.nbytes, abi.stream()), "mcp_nll_epoch")
            for cnt, ps, gs, ms, vs, ne in chunks:
                abi.check(lib.mcp_adam_step_guarded(cnt, ps, gs, ms, vs, ne, lr, b1, b2, eps, None, epoch + 1, 0, None, None, abi.ptr(self.status), abi.stream()),
                          "mcp_adam_step_guarded")
            if epoch % self.N_epoch_print == 0:
                snapshot(epoch)
        snapshot(self.N_epoch)  # the final parameters
        if ops.status_flags(self.status)["not_spd"]:  # (the one host sync of the training)
            # the failing epoch and every later one skipped their update on the device (mcp_adam_step_guarded reads the sticky flag): the
            # hyper-parameters are those of the last good step, as when the reference raises from torch.cholesky (GP_prior.py:106)
            raise RuntimeError("cholesky: the covariance matrix is not positive-definite")
        for p, g in zip(self.params, self.grads):
            p.grad = g  # (what the last epoch left, as after the reference's last backward)
        # the text GP_prior.fit_model prints, GP by GP
        texts = []
        for i, gp in enumerate(self.gps):
            out = io.StringIO()

            def show(params):
                saved = {k: v.detach().clone() for k, v in gp.named_parameters()}
                with torch.no_grad():
                    for k, v in gp.named_parameters():
                        v.copy_(params[k])
                import contextlib

                with contextlib.redirect_stdout(out):
                    gp.print_model()
                with torch.no_grad():
                    for k, v in gp.named_parameters():
                        v.copy_(saved[k])

            out.write("\nInitial parameters:\n")
            show(snaps[0][1][i])
            t_prev = snaps[0][3]
            for epoch, pars, losses, tm in snaps[1:-1]:
                out.write("\nEPOCH: %d\n" % epoch)
                show(pars[i])
                out.write("Running loss: %s\n" % float(losses[i]))
                out.write("Time elapsed: %s\n" % (tm - t_prev))
                t_prev = tm
            out.write("\nFinal parameters:\n")
            show(snaps[-1][1][i])
            texts.append(out.getvalue())
        return texts
