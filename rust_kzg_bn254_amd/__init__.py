"""Import shim: the package lives in `rust-kzg-bn254_amd/` (not a valid Python identifier)."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "rust-kzg-bn254_amd"))
from importlib import util as _util

_spec = _util.spec_from_file_location(__name__, _os.path.join(__path__[0], "__init__.py"), submodule_search_locations=__path__)
_mod = _util.module_from_spec(_spec)
import sys as _sys

_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
