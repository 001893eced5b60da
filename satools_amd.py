"""Import alias: the product package lives in the directory `sa-toolkit_amd/` (the name the
project layout fixes); a hyphen is not importable, so `import satools_amd` resolves here and
this shim loads that directory as the package `satools_amd`."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sa-toolkit_amd")
_spec = importlib.util.spec_from_file_location(
    "satools_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["satools_amd"] = _mod
_spec.loader.exec_module(_mod)
