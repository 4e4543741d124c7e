"""Import alias for the package directory `slsforasvspoof-2021-df_amd/`.

The directory name is not a valid Python identifier, so it is loaded by path
and registered in sys.modules as `slsmi`; `import slsmi` then behaves like a
normal package import (sub-modules: slsmi.config, slsmi.model, ...).
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "slsforasvspoof-2021-df_amd")


def _load():
    spec = importlib.util.spec_from_file_location(
        "slsmi", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["slsmi"] = mod
    spec.loader.exec_module(mod)
    return mod


if sys.modules.get("slsmi") is None or getattr(sys.modules["slsmi"], "__path__", None) is None:
    _load()
