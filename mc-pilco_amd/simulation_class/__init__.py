if __name__ == "simulation_class":  # imported through the reference's top-level path
    import _alias

    _alias.alias_package("simulation_class", ["model", "ode_systems"])
else:
    from . import model, ode_systems  # noqa: F401
