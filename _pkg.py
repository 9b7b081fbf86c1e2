"""Import helper: the product package directory is named 'vit.cpp_amd' (after the
reference repo), which is not a legal Python identifier, so it is registered in
sys.modules under the alias 'vitcpp_amd'."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "vit.cpp_amd")
ALIAS = "vitcpp_amd"


def load():
    if ALIAS in sys.modules:
        return sys.modules[ALIAS]
    spec = importlib.util.spec_from_file_location(ALIAS, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod
