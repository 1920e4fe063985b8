"""Import shim: the product package lives in the directory `oldoceananigans.jl_amd/` (a name Python cannot import
directly because of the dot). `import oldoceananigans_jl_amd as ocn` loads that directory as a package."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oldoceananigans.jl_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_pkg_dir, "__init__.py"),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
