if __name__ == "model_learning":  # imported through the reference's top-level path
    import _alias

    _alias.alias_package("model_learning", ["Model_learning"])
else:
    from . import Model_learning  # noqa: F401
