"""GP library with the reference's module layout (gpr_lib/GP_prior, Likelihood, Utils)."""
if __name__ == "gpr_lib":  # imported through the reference's top-level path
    import _alias

    _alias.alias_package("gpr_lib", ["GP_prior", "GP_prior.GP_prior", "GP_prior.Stationary_GP", "GP_prior.Sparse_GP", "Likelihood",
                                     "Likelihood.Gaussian_likelihood", "Utils", "Utils.Parameters_covariance_functions"])
else:
    from . import GP_prior, Likelihood, Utils  # noqa: F401
