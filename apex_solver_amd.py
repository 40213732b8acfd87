"""Import shim: the package directory is `apex-solver_amd/` (hyphen, as the repo layout
requires), which is not a valid Python identifier.  `import apex_solver_amd` loads this
file, which loads the real package from that directory under the same module name."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "apex-solver_amd")
_spec = _u.spec_from_file_location(
    "apex_solver_amd", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = _u.module_from_spec(_spec)
_sys.modules["apex_solver_amd"] = _mod
_spec.loader.exec_module(_mod)
