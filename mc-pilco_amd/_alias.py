"""Lets launch scripts keep the reference's own import paths.

With ``mc-pilco_amd/`` on ``sys.path`` a script can write, exactly as with the reference,

    import gpr_lib.Likelihood.Gaussian_likelihood as Likelihood
    import model_learning.Model_learning as ML
    import policy_learning.MC_PILCO as MC_PILCO

Those top-level names are aliases of the canonical ``mc_pilco_amd.<...>`` modules (one module
object per file, so ``isinstance`` checks agree whichever path was used to import a class).
"""
import importlib
import importlib.util
import os
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))


def ensure_canonical():
    if "mc_pilco_amd" not in sys.modules:
        spec = importlib.util.spec_from_file_location("mc_pilco_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                      submodule_search_locations=[PKG_DIR])
        mod = importlib.util.module_from_spec(spec)
        sys.modules["mc_pilco_amd"] = mod
        spec.loader.exec_module(mod)
    return sys.modules["mc_pilco_amd"]


def alias_package(top_name, submodules):
    """Called from ``<top_name>/__init__.py`` when it is imported as a TOP-LEVEL package."""
    ensure_canonical()
    canon = importlib.import_module("mc_pilco_amd." + top_name)
    sys.modules[top_name] = canon
    for sub in submodules:
        m = importlib.import_module("mc_pilco_amd.%s.%s" % (top_name, sub))
        sys.modules["%s.%s" % (top_name, sub)] = m
    return canon
