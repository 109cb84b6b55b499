"""splatapult_amd -- MI355X-native (gfx950, HIP) 3D Gaussian splat sort + render path behind the
reference's SplatRenderer / GaussianCloud surface (hyperlogic/splatapult).

Only the hot path lives here: csrc/ (HIP kernels + C ABI), host/ (C++ scene-data surface),
and this thin Python mirror of the reference interface.  The CPU oracle is test infrastructure
under /oracle and is never imported from this package."""
from ._capi import MsplatError, lib  # noqa: F401
from .renderer import SplatRenderer, SplatRendererGroup  # noqa: F401
from .scene import GaussianCloud  # noqa: F401
from .points import PointCloud, PointRenderer  # noqa: F401

__all__ = ["SplatRenderer", "SplatRendererGroup", "GaussianCloud", "PointCloud", "PointRenderer", "MsplatError", "lib"]
