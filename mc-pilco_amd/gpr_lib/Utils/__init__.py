from . import Parameters_covariance_functions  # noqa: F401
