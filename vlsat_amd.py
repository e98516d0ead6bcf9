"""Import shim: the package directory is ``cvpr2023-vlsat_amd`` (not a valid Python
identifier), so ``import vlsat_amd`` loads it from that path under this name."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "cvpr2023-vlsat_amd")
_spec = _ilu.spec_from_file_location(
    "vlsat_amd", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["vlsat_amd"] = _mod
_spec.loader.exec_module(_mod)
