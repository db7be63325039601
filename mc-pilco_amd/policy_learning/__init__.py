if __name__ == "policy_learning":  # imported through the reference's top-level path
    import _alias

    _alias.alias_package("policy_learning", ["Policy", "Cost_function", "MC_PILCO"])
else:
    from . import Cost_function, MC_PILCO, Policy  # noqa: F401
