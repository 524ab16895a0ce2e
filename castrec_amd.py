"""Import shim: exposes the product package directory
``context-aware-sequential-recommendation_amd/`` (not a valid Python identifier)
under the importable name ``castrec_amd``.

    import castrec_amd
    from castrec_amd.models import SASRec
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)),
                         "context-aware-sequential-recommendation_amd")
_spec = _ilu.spec_from_file_location(
    "castrec_amd", _os.path.join(_PKG_DIR, "__init__.py"),
    submodule_search_locations=[_PKG_DIR])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["castrec_amd"] = _mod          # replace this shim by the real package
_spec.loader.exec_module(_mod)
