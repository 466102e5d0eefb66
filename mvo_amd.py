"""Importable alias for the package directory ``monocular-visual-odometry_amd`` (its name is not a
Python identifier).  ``import mvo_amd`` and ``from mvo_amd.ProjectiveICP import ProjectiveICP`` give
the same module objects as importing the directory by its real name through importlib."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

_REAL = "monocular-visual-odometry_amd"
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name == _REAL or _name.startswith(_REAL + "."):
        sys.modules[__name__ + _name[len(_REAL):]] = _mod
