"""Loads the product package, whose directory name (`sparse-conv-gpis-tungsten_amd`) is not a
valid Python identifier, under the module name ``scgt_amd``."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "sparse-conv-gpis-tungsten_amd")


def load_package():
    if "scgt_amd" in sys.modules:
        return sys.modules["scgt_amd"]
    spec = importlib.util.spec_from_file_location(
        "scgt_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["scgt_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
