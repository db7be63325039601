from . import Gaussian_likelihood  # noqa: F401
