"""MI355X-native projective-ICP pose loop and appearance association behind the Python interface
of ValerioSpagnoli/Monocular-Visual-Odometry.

Module names follow the reference's ``src`` package, so its driver code can switch by changing
the package name in its imports:  ProjectiveICP, data_association, Camera, Data, geometry_utils,
VisualOdometry.  The compute path is libpicp_hip.so (csrc/, C ABI in include/picp_hip.h); there is
no CPU fallback for it.
"""
from . import _lib, geometry_utils, two_view, evaluation, Camera as _Camera, Data as _Data, data_association  # noqa: F401
from . import ProjectiveICP as _ProjectiveICP, VisualOdometry as _VisualOdometry, sharded  # noqa: F401
from ._lib import Context, PicpError, get_context, load_library  # noqa: F401
from .data_association import data_association_on_appearance  # noqa: F401

__all__ = ["Context", "PicpError", "get_context", "load_library", "data_association_on_appearance"]
