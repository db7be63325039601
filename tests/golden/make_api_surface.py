#!/usr/bin/env python3
"""Records the PUBLIC SURFACE of the reference's in-scope modules -- every class with its public methods and their
argument lists, every module-level function with its argument list -- into tests/golden/api_surface.json.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_api_surface.py            (needs /root/reference; build container only)

The reference is parsed with `ast` (no import, nothing executed): names and signatures only, no source text is stored.
tests/test_api_surface.py diffs the drop-in package against this file, so a method or keyword the reference has and the
drop-in lacks is a test failure, not something a user finds with an AttributeError.
"""
import ast
import json
import os
import sys

REF = os.environ.get("MCPILCO_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

# module path relative to the reference root (= relative to mc-pilco_amd/ in the drop-in)
MODULES = [
    "gpr_lib/GP_prior/GP_prior.py",
    "gpr_lib/GP_prior/Stationary_GP.py",
    "gpr_lib/GP_prior/Sparse_GP.py",
    "gpr_lib/Likelihood/Gaussian_likelihood.py",
    "gpr_lib/Utils/Parameters_covariance_functions.py",
    "model_learning/Model_learning.py",
    "policy_learning/Policy.py",
    "policy_learning/Cost_function.py",
    "policy_learning/MC_PILCO.py",
    "simulation_class/model.py",
    "simulation_class/ode_systems.py",
]


def signature(fn):
    """[positional names..., '*args'?, keyword-only names..., '**kw'?] and which of them have defaults."""
    a = fn.args
    pos = [x.arg for x in getattr(a, "posonlyargs", [])] + [x.arg for x in a.args]
    n_def = len(a.defaults)
    out = []
    for i, name in enumerate(pos):
        out.append({"name": name, "default": i >= len(pos) - n_def})
    if a.vararg:
        out.append({"name": "*" + a.vararg.arg, "default": True})
    for x, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append({"name": x.arg, "default": d is not None, "kwonly": True})
    if a.kwarg:
        out.append({"name": "**" + a.kwarg.arg, "default": True})
    return out


def surface(path):
    tree = ast.parse(open(path).read(), filename=path)
    classes, functions = {}, {}
    for node in tree.body:
        if isinstance(node, ast.ClassDef):
            methods = {}
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and (not item.name.startswith("_") or item.name == "__init__"):
                    methods[item.name] = signature(item)
            bases = []
            for b in node.bases:
                bases.append(b.attr if isinstance(b, ast.Attribute) else getattr(b, "id", "?"))
            classes[node.name] = {"bases": bases, "methods": methods}
        elif isinstance(node, ast.FunctionDef) and not node.name.startswith("_"):
            functions[node.name] = signature(node)
    return {"classes": classes, "functions": functions}


def main():
    out = {}
    for m in MODULES:
        out[m] = surface(os.path.join(REF, m))
    dst = os.path.join(HERE, "api_surface.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    n_c = sum(len(v["classes"]) for v in out.values())
    n_m = sum(len(c["methods"]) for v in out.values() for c in v["classes"].values())
    n_f = sum(len(v["functions"]) for v in out.values())
    print("%s: %d modules, %d classes, %d methods, %d functions" % (dst, len(out), n_c, n_m, n_f))


if __name__ == "__main__":
    sys.exit(main())
