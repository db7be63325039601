from . import GP_prior, Sparse_GP, Stationary_GP  # noqa: F401
