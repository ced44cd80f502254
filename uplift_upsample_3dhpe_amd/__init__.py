"""Importable alias for the hyphenated package directory ``uplift-upsample-3dhpe_amd/``.

Python cannot ``import`` a directory with a hyphen in its name, so this shim loads the real
package (one level up) under the importable name ``uplift_upsample_3dhpe_amd`` and replaces
itself in ``sys.modules``.  All code lives in ``uplift-upsample-3dhpe_amd/``.
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "uplift-upsample-3dhpe_amd")
_spec = _ilu.spec_from_file_location(__name__, _os.path.join(_real, "__init__.py"),
                                     submodule_search_locations=[_real])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
