"""Import shim: the package directory is named `ptudes-lab_amd/` (a hyphen cannot appear in a Python
identifier), so `import ptudes_lab_amd` loads it from there."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "ptudes-lab_amd")
_spec = _ilu.spec_from_file_location("ptudes_lab_amd", _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["ptudes_lab_amd"] = _mod
_spec.loader.exec_module(_mod)
