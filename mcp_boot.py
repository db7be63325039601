"""Registers the in-tree package directory ``mc-pilco_amd/`` under the importable name
``mc_pilco_amd`` (a hyphen cannot appear in a Python module name).

    import mcp_boot            # idempotent
    import mc_pilco_amd as mcp

Used by tests, ``bench.py`` and ``__graft_entry__.py``; launch scripts that want the
reference's own import paths (``import gpr_lib...``, ``model_learning.Model_learning``,
``policy_learning.MC_PILCO``) can instead put ``mc-pilco_amd/`` on ``sys.path``
(see INTEGRATION.md).
"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "mc-pilco_amd")

if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

if "mc_pilco_amd" not in sys.modules:
    _spec = importlib.util.spec_from_file_location(
        "mc_pilco_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR]
    )
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules["mc_pilco_amd"] = _mod
    _spec.loader.exec_module(_mod)
