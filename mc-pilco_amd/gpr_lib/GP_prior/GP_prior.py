"""GP base classes on the HIP path -- drop-in for the reference's ``gpr_lib/GP_prior/GP_prior.py``
(class and method names, argument order and return arity follow it; numerics run in
libmcpilco_hip.so through ``mc_pilco_amd.ops``).

Reference lines replaced:
  GP_prior.forward                GP_prior.py:91-115   Gram (+noise) -> Cholesky (upper) -> U^-1 -> K^-1, logdet
  GP_prior.get_alpha              GP_prior.py:130-135
  GP_prior.get_estimate_from_alpha  GP_prior.py:137-155  posterior mean / variance at test points
  GP_prior.get_estimate           GP_prior.py:157-171
  GP_prior.get_SOD                GP_prior.py:232-257  greedy subset-of-data (index-exact)
  GP_prior.fit_model              GP_prior.py:179-230  hyper-parameter training (analytic NLL gradient, HIP)
  Sum_Independent_GP              GP_prior.py:299-347  kernel sum; noise added once; mean of the first child
Out of scope (unused by every launch script): Multiply_GP_prior, Scale_GP_prior.

Every GP object describes its covariance to the kernels through ``kernel_spec()``: squared
exponential (+ Volterra polynomial of degree <= 2) over the columns ``active_dims``.
"""
import time

import numpy as np
import torch

from mc_pilco_amd import ops

__all__ = ["GP_prior", "Combine_GP", "Sum_Independent_GP"]


class GP_prior(torch.nn.Module):
    """Base class: noise parameter, device/dtype bookkeeping and every operation that only needs
    ``kernel_spec()``."""

    def __init__(self, active_dims, sigma_n_init=None, flg_train_sigma_n=True, name="", dtype=torch.float64, sigma_n_num=None, device=None):
        super().__init__()
        self.name = name
        self.dtype = dtype
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.active_dims = None if active_dims is None else torch.as_tensor(np.asarray(active_dims), dtype=torch.long, device=self.device)
        self.GP_with_noise = sigma_n_init is not None
        if self.GP_with_noise:
            self.sigma_n_log = torch.nn.Parameter(torch.tensor(np.log(sigma_n_init), dtype=dtype, device=self.device),
                                                  requires_grad=flg_train_sigma_n)
        self.sigma_n_num = torch.tensor(0.0 if sigma_n_num is None else float(np.asarray(sigma_n_num)), dtype=dtype, device=self.device)
        self._packed_cache = {}

    # ---- bookkeeping ---------------------------------------------------------------------------
    def to(self, dev):
        super().to(dev)
        self.device = torch.device(dev)
        self.sigma_n_num = self.sigma_n_num.to(dev)
        if self.active_dims is not None:
            self.active_dims = self.active_dims.to(dev)

    def set_eval_mode(self):
        """Freezes every parameter (remembering which ones were trainable)."""
        self.flg_trainable_list = [p.requires_grad for p in self.parameters()]
        for p in self.parameters():
            p.requires_grad = False

    def set_training_mode(self):
        for flag, p in zip(self.flg_trainable_list, self.parameters()):
            p.requires_grad = flag

    def get_sigma_n_2(self):
        """exp(sigma_n_log)^2 + sigma_n_num^2"""
        return torch.exp(self.sigma_n_log) ** 2 + self.sigma_n_num ** 2

    def print_model(self):
        print(self.name + " parameters:")
        for n, v in self.named_parameters():
            print("-", n, ":", v.data)

    # ---- what the HIP kernels need -----------------------------------------------------------------
    def kernel_spec(self) -> ops.KernelSpec:
        raise NotImplementedError()

    def _cols(self, X):
        """The active columns of X as a contiguous float64 GPU matrix."""
        X = X.to(device=self.device, dtype=torch.float64)
        ad = self._active()
        if ad is not None:
            # "the active dimensions are all columns, in order" is decided ONCE per (index tensor, width): the comparison reads a device
            # tensor back -- a host sync -- and this runs in every epoch of GP training and every pretrain
            # (the cache HOLDS the index tensor and compares with `is` + its in-place version counter, like _packed(): an id() alone
            # could be a freed tensor's address handed to a new one, and an in-place edit of active_dims must not go unnoticed)
            cached = self.__dict__.get("_cols_identity")
            if cached is not None and cached[0] is ad and cached[1] == getattr(ad, "_version", 0) and cached[2] == int(X.shape[1]):
                ident = cached[3]
            else:
                adt = torch.as_tensor(ad)
                ident = adt.numel() == X.shape[1] and bool((adt.detach().cpu() == torch.arange(X.shape[1])).all())
                self.__dict__["_cols_identity"] = (ad, getattr(ad, "_version", 0), int(X.shape[1]), ident)
            if not ident:
                X = X[:, ad]
        return X.contiguous()

    def _active(self):
        return self.active_dims

    # ---- covariance / mean ----------------------------------------------------------------------------
    def get_mean(self, X):
        raise NotImplementedError()

    def get_covariance(self, X1, X2=None, flg_noise=False):
        noise = bool(flg_noise) and self.GP_with_noise
        return ops.cov_build(self.kernel_spec(), self._cols(X1), None if X2 is None else self._cols(X2), noise=noise)

    def get_diag_covariance(self, X, flg_noise=False):
        noise = bool(flg_noise) and self.GP_with_noise
        return ops.cov_diag(self.kernel_spec(), self._cols(X), noise=noise)

    # ---- Gram / Cholesky / inverse ------------------------------------------------------------------------
    def forward(self, X):
        """(m_X, K_X, K_X_inv, log_det): K_X includes the noise, K_X_inv = U^-1 U^-T with U the upper
        Cholesky factor, log_det = 2 sum log diag U.  With autograd on and trainable hyper-parameters the outputs carry a graph, as in the
        reference (GP_prior.py:91-115): any criterion of them can be differentiated (``_ForwardFunction``)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_autograd(X)
        K = self.get_covariance(X, flg_noise=self.GP_with_noise)
        U, log_det, status = ops.chol_factor(K)
        if ops.status_flags(status)["not_spd"]:
            raise RuntimeError("cholesky: the covariance matrix is not positive-definite")
        _, K_inv = ops.chol_inverse(U)
        return self.get_mean(X), K, K_inv, log_det

    def _forward_autograd(self, X):
        params = [p for p in self.parameters() if p.requires_grad]
        K, K_inv, log_det = _ForwardFunction.apply(self, X, *params)
        first = next(iter(self._leaves())) if isinstance(self, Combine_GP) else self
        mp = getattr(first, "mean_par", None)
        if mp is not None and mp.requires_grad and not getattr(first, "flg_no_mean", False) and mp.numel() == 1:
            m_X = mp.reshape(1, -1).repeat(X.shape[0], 1)  # (the constant prior mean of the first child, differentiable)
        else:
            m_X = self.get_mean(X)
        return m_X, K, K_inv, log_det

    def get_alpha(self, X, Y):
        m_X, _, K_X_inv, _ = self(X)
        alpha = ops.gp_alpha(K_X_inv, (Y.to(self.device) - m_X).contiguous(), 0.0)
        return alpha, m_X, K_X_inv

    def _packed(self, X, alpha, K_X_inv):
        """Packed operands of (X, alpha, K_X_inv) under the CURRENT hyper-parameters.  The one-entry cache holds strong
        references to the three tensors and compares them by identity (an address can be reused by a new tensor once the old
        one is freed), together with their in-place version counters and those of every hyper-parameter (optimizer steps,
        load_state_dict and reinit all write in place and bump them)."""
        ver = tuple(int(t._version) for t in (X, alpha, K_X_inv)) + tuple(int(p._version) for p in self.parameters())
        hit = self._packed_cache.get("last")
        if hit is not None and hit[0] is X and hit[1] is alpha and hit[2] is K_X_inv and hit[3] == ver:
            return hit[4]
        gp = ops.PackedGP(self.kernel_spec(), self._cols(X), alpha, K_X_inv)
        self._packed_cache["last"] = (X, alpha, K_X_inv, ver, gp)
        return gp

    def get_estimate_from_alpha(self, X, X_test, alpha, m_X, K_X_inv=None, Y_test=None):
        """Posterior at X_test from the cached alpha (and variance when K_X_inv is given).  Differentiable
        with respect to X_test."""
        N = X.shape[0]
        Kinv = K_X_inv if K_X_inv is not None else torch.zeros(N, N, dtype=torch.float64, device=self.device)
        gp = self._packed(X, alpha, Kinv)
        Y_hat, var = ops.posterior(gp, self._cols(X_test))
        if Y_test is not None:
            print("MSE:", torch.sum((Y_test.to(self.device) - Y_hat) ** 2) / Y_test.shape[0])
        if K_X_inv is None:
            return Y_hat
        return Y_hat, var

    def get_estimate(self, X, Y, X_test, Y_test=None, flg_return_K_X_inv=False):
        alpha, m_X, K_X_inv = self.get_alpha(X, Y)
        self._packed_cache.clear()
        Y_hat, var = self.get_estimate_from_alpha(X, X_test, alpha, m_X, K_X_inv=K_X_inv, Y_test=Y_test)
        if flg_return_K_X_inv:
            return Y_hat, var, alpha, m_X, K_X_inv
        return Y_hat, var, alpha

    # ---- subset of data ---------------------------------------------------------------------------------------
    def get_SOD(self, X, Y, threshold, flg_permutation=False):
        """Indices (python ints, in visiting order) of the greedy subset: sample i joins when the
        posterior std given the current subset exceeds ``threshold``."""
        print("\nSelection of the inducing inputs...")
        Xc = self._cols(X)
        n = Xc.shape[0]
        thr = float(threshold)
        if flg_permutation:
            order = torch.cat([torch.zeros(1, dtype=torch.long), torch.arange(1, n)[torch.randperm(n - 1)]])
            picked = ops.sod_select(self.kernel_spec(), Xc[order.to(self.device)].contiguous(), thr)
            idx = [int(order[i]) for i in picked]
        else:
            idx = ops.sod_select(self.kernel_spec(), Xc, thr)
        print("Shape inducing inputs selected:", torch.Size([len(idx), X.shape[1]]))
        return idx

    # ---- hyper-parameter training --------------------------------------------------------------------------------
    def fit_model(self, trainloader=None, optimizer=None, criterion=None, N_epoch=1, N_epoch_print=1, f_saving_model=None, f_print=None):
        """Optimisation of ``criterion(self(inputs), labels)`` with the caller's optimizer, as in the reference.  For the
        Marginal_log_likelihood the gradient is the analytic NLL derivative evaluated by the HIP kernels (one full batch: the whole epoch on
        the device); any other criterion runs the reference's loop on the differentiable forward."""
        from mc_pilco_amd.gpr_lib.Likelihood.Gaussian_likelihood import Marginal_log_likelihood

        from mc_pilco_amd import nll

        if not isinstance(criterion, Marginal_log_likelihood):
            # any other criterion of [m_X, K_X, K_X_inv, log_det]: the reference's own loop (GP_prior.py:179-230) on the differentiable forward
            print("\nInitial parameters:")
            self.print_model()
            t0 = time.time()
            for epoch in range(N_epoch):
                running, nb = 0.0, 0
                optimizer.zero_grad()
                for inputs, labels in trainloader:
                    optimizer.zero_grad()
                    loss = criterion(self(inputs), labels.to(self.device))
                    loss.backward()
                    optimizer.step()
                    running += loss.item()
                    nb += 1
                if epoch % N_epoch_print == 0:
                    print("\nEPOCH:", epoch)
                    self.print_model()
                    print("Running loss:", running / max(nb, 1))
                    print("Time elapsed:", time.time() - t0)
                    t0 = time.time()
                    if f_saving_model is not None:
                        f_saving_model(epoch)
                    if f_print is not None:
                        f_print()
            print("\nFinal parameters:")
            self.print_model()
            return

        # one full batch, the textbook Adam, a squared-exponential (+ Volterra) kernel: the whole epoch is two C calls (mcp_nll_epoch +
        # mcp_adam_step_guarded), no torch op and no host sync in between (mc_pilco_amd/nll.py: BatchedFit, here with one GP)
        batches = list(trainloader) if (f_saving_model is None and f_print is None and isinstance(trainloader, (list, tuple))) else None
        if batches is not None and len(batches) == 1:
            fit = nll.BatchedFit([self], batches[0][0], [batches[0][1]], [1.0], [optimizer], N_epoch, N_epoch_print)
            if fit.eligible:
                print(fit.run()[0], end="")
                return
        print("\nInitial parameters:")
        self.print_model()
        t0 = time.time()
        for epoch in range(N_epoch):
            running, nb = None, 0  # the loss stays on the device: one host sync per epoch (the kernel descriptor's scalars), not three
            for inputs, labels in trainloader:
                optimizer.zero_grad()
                loss = criterion.loss_and_grad(self, inputs, labels)
                optimizer.step()
                running = loss if running is None else running + loss
                nb += 1
            if epoch % N_epoch_print == 0:
                nll.check_status(self)
                print("\nEPOCH:", epoch)
                self.print_model()
                print("Running loss:", (float(running) if running is not None else 0.0) / max(nb, 1))
                print("Time elapsed:", time.time() - t0)
                t0 = time.time()
                if f_saving_model is not None:
                    f_saving_model(epoch)
                if f_print is not None:
                    f_print()
        nll.check_status(self)
        print("\nFinal parameters:")
        self.print_model()


class _ForwardFunction(torch.autograd.Function):
    """(K_X, K_X_inv, log_det) of ``GP_prior.forward`` as a differentiable function of the GP's trainable hyper-parameters -- what autograd
    records through get_covariance / torch.cholesky / torch.inverse in the reference (GP_prior.py:91-115).  Forward: the HIP Gram,
    factorisation and inverse.  Backward: with the upstream gradients G_K, G_Kinv, g_logdet the Gram matrix's own gradient is
        Wm = G_K - Kinv G_Kinv Kinv + g_logdet Kinv          (d Kinv = -Kinv dK Kinv,  d logdet = tr(Kinv dK); Kinv symmetric)
    -- two products on the library's MFMA GEMM (`mcp_sym_sandwich`, the `tn_gemm_kernel` of the factorisation) -- and
    dL/dtheta = sum_ij Wm_ij dK_ij/dtheta  comes from the HIP gradient kernel (nll.cov_weighted_grad)."""

    @staticmethod
    def forward(ctx, gp, X, *params):
        Xc = gp._cols(X)
        spec = gp.kernel_spec_dev()  # a snapshot of the hyper-parameters (new tensors): backward differentiates the kernel THIS forward evaluated
        K = ops.cov_build(spec, Xc, None, noise=gp.GP_with_noise)
        U, log_det, status = ops.chol_factor(K)
        if ops.status_flags(status)["not_spd"]:
            raise RuntimeError("cholesky: the covariance matrix is not positive-definite")
        _, K_inv = ops.chol_inverse(U)
        ctx.gp, ctx.Xc, ctx.params, ctx.spec = gp, Xc, params, spec
        # (the chain rule's last factors -- d sigma_n^2 / d sigma_n_log, d w / d Sigma_pos_par -- are formed from the live parameters in
        #  backward: an in-place change in between must not go unnoticed, as torch's own version check would not let it in the reference)
        ctx.versions = tuple(int(p._version) for p in gp.parameters())
        ctx.save_for_backward(K_inv)
        ctx.set_materialize_grads(False)
        return K, K_inv, log_det.reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_K, g_Kinv, g_ld):
        from mc_pilco_amd import nll

        if tuple(int(p._version) for p in ctx.gp.parameters()) != ctx.versions:
            raise RuntimeError("one of the GP's hyper-parameters was modified by an inplace operation between GP_prior.forward and the backward "
                               "of its outputs (e.g. optimizer.step() before a second backward): run forward again")
        (K_inv,) = ctx.saved_tensors
        Wm = torch.zeros_like(K_inv)
        if g_K is not None:
            Wm = Wm + g_K
        if g_Kinv is not None:
            Wm = Wm - ops.sym_sandwich(K_inv, g_Kinv)  # Kinv G Kinv on the library's own MFMA GEMM (mcp_sym_sandwich)
        if g_ld is not None:
            Wm = Wm + g_ld * K_inv
        g = nll.cov_weighted_grad(ctx.gp, ctx.Xc, Wm, spec=ctx.spec)
        got = {id(p): v for p, v in nll.kernel_param_grads(ctx.gp, g, ctx.Xc.shape[1], spec=ctx.spec)}
        return (None, None) + tuple(got.get(id(p)) for p in ctx.params)


class Combine_GP(GP_prior):
    """Common part of kernels built from several GP objects."""

    def __init__(self, *gp_priors_obj):
        first = gp_priors_obj[0]
        super().__init__(active_dims=None, sigma_n_num=float(first.sigma_n_num), dtype=first.dtype, device=first.device)
        self.gp_list = torch.nn.ModuleList(gp_priors_obj)
        self.GP_with_noise = any(gp.GP_with_noise for gp in self.gp_list)

    def to(self, dev):
        super().to(dev)
        for gp in self.gp_list:
            gp.to(dev)

    def print_model(self):
        for gp in self.gp_list:
            gp.print_model()

    def get_sigma_n_2(self):
        s = torch.zeros(1, dtype=self.dtype, device=self.device)
        for gp in self.gp_list:
            if gp.GP_with_noise:
                s = s + gp.get_sigma_n_2()
        return s

    def _leaves(self):
        for gp in self.gp_list:
            if isinstance(gp, Combine_GP):
                yield from gp._leaves()
            else:
                yield gp

    def _active(self):
        ads = [gp._active() for gp in self._leaves()]
        for a in ads[1:]:
            if a is None or ads[0] is None or a.numel() != ads[0].numel() or not bool((a == ads[0]).all()):
                raise NotImplementedError("summed kernels must share their active_dims on the HIP path")
        return ads[0]


class Sum_Independent_GP(Combine_GP):
    """Sum of independent GPs: covariances add, the measurement noise (that of the noisy children) is
    added once, and -- as in the reference -- the prior mean is the FIRST child's mean."""

    def get_mean(self, X):
        return self.gp_list[0].get_mean(X)

    def kernel_spec_dev(self) -> ops.KernelSpec:
        """kernel_spec without any device->host transfer (see RBF.kernel_spec_dev)."""
        from .Stationary_GP import RBF

        se, w1, w20, w21 = None, None, None, None
        leaves = list(self._leaves())
        for gp in leaves:
            part = gp.kernel_spec_dev()
            if isinstance(gp, RBF):
                if se is not None:
                    raise NotImplementedError("at most one squared-exponential term per GP on the HIP path")
                se = part
            if part.w1 is not None:
                w1 = part.w1 if w1 is None else w1 + part.w1
            if part.w20 is not None:
                if w20 is not None:
                    raise NotImplementedError("at most one degree-2 polynomial term per GP on the HIP path")
                w20, w21 = part.w20, part.w21
        dev = self.device
        D = (se or leaves[0].kernel_spec_dev()).D
        z = torch.zeros(1, dtype=torch.float64, device=dev)
        if w20 is not None and w1 is None:
            w1 = torch.zeros(D + 1, dtype=torch.float64, device=dev)
        sig2 = self.get_sigma_n_2().detach().reshape(-1)[:1].to(torch.float64) if self.GP_with_noise else z
        first = self.gp_list[0]
        mean = first.mean_par.detach().reshape(-1)[:1].to(torch.float64) if hasattr(first, "mean_par") else z
        lam = se.scal[0:1] if se is not None else z
        scal = torch.cat([lam, sig2, mean]).contiguous()
        nan = float("nan")
        ls = se.lengthscales if se is not None else torch.ones(D, dtype=torch.float64, device=dev)
        return ops.KernelSpec(ls, nan, nan, nan, w1, w20, w21, scal=scal)

    def kernel_spec(self) -> ops.KernelSpec:
        se, w1, w20, w21 = None, None, None, None
        for gp in self._leaves():
            part = gp.kernel_spec()
            if part.lam != 0.0:
                if se is not None:
                    raise NotImplementedError("at most one squared-exponential term per GP on the HIP path")
                se = part
            if part.w1 is not None:
                w1 = part.w1 if w1 is None else w1 + part.w1
            if part.w20 is not None:
                if w20 is not None:
                    raise NotImplementedError("at most one degree-2 polynomial term per GP on the HIP path")
                w20, w21 = part.w20, part.w21
        D = next(self._leaves()).kernel_spec().D
        if w20 is not None and w1 is None:
            w1 = torch.zeros(D + 1, dtype=torch.float64)
        sig2 = float(self.get_sigma_n_2()) if self.GP_with_noise else 0.0
        first = self.gp_list[0]
        mean = float(first.mean_par.reshape(-1)[0]) if hasattr(first, "mean_par") else 0.0
        if se is None:
            return ops.KernelSpec(torch.ones(D, dtype=torch.float64), 0.0, sig2, mean, w1, w20, w21)
        return ops.KernelSpec(se.lengthscales, se.lam, sig2, mean, w1, w20, w21)
