"""Import shim: ``import revisiting_at_amd`` loads the package kept in ``revisiting-at_amd/``.

The directory name is fixed by the project layout and is not a valid Python identifier;
this module swaps itself for the real package in ``sys.modules`` at import time.
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "revisiting-at_amd")
_spec = _ilu.spec_from_file_location(__name__, _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_pkg = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _pkg
_spec.loader.exec_module(_pkg)
