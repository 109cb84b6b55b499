"""PointCloud / PointRenderer: Python mirror of the reference's SfM point-cloud view (SURVEY.md 8f-4;
/root/reference/src/pointcloud.h:15-48, src/pointrenderer.h:23-57).  Same method names and argument meaning;
everything is computed by libmsplat.so (host C++ for the cloud, HIP kernels for the rendering)."""
import ctypes as C

import numpy as np

from . import _capi
from .renderer import SplatRenderer


class PointCloud:
    def __init__(self, useLinearColors=False):
        self._lib = _capi.lib()
        self._h = self._lib.msplat_points_create(1 if useLinearColors else 0)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.msplat_points_destroy(h)

    @property
    def handle(self):
        return self._h

    def ImportPly(self, plyFilename):
        """pointcloud.cpp:33-131; False (after logging) on failure"""
        return self._lib.msplat_points_import_ply(self._h, str(plyFilename).encode()) == _capi.OK

    def ExportPly(self, plyFilename):
        return self._lib.msplat_points_export_ply(self._h, str(plyFilename).encode()) == _capi.OK

    def InitDebugCloud(self):
        self._lib.msplat_points_init_debug(self._h)

    def GetNumPoints(self):
        return int(self._lib.msplat_points_num(self._h))

    def GetStride(self):
        return int(self._lib.msplat_points_stride(self._h))

    def GetTotalSize(self):
        return self.GetNumPoints() * self.GetStride()

    def as_array(self):
        """(N, 8) float32 copy: position.xyzw, color.rgba (pointcloud.cpp:19-23)"""
        n = self.GetNumPoints()
        if n == 0:
            return np.zeros((0, 8), np.float32)
        buf = (C.c_float * (n * 8)).from_address(self._lib.msplat_points_data(self._h))
        return np.frombuffer(buf, np.float32).reshape(n, 8).copy()


class PointRenderer(SplatRenderer):
    """PointRenderer::Init(pointCloud, isFramebufferSRGBEnabled) / Render(cameraMat, projMat, viewport, nearFar)
    (pointrenderer.cpp:48-196).  Render sorts and draws in one call, like the reference."""

    def Init(self, pointCloud, isFramebufferSRGBEnabled=False, sprite=None):
        """pointCloud: PointCloud or an (N, 8) float32 array.  sprite: (h, w, 4) uint8 RGBA, top row first (e.g.
        camera.read_image('texture/sphere.png')); None = the library's built-in sphere sprite."""
        if not self._create(isFramebufferSRGBEnabled):
            return False
        if isinstance(pointCloud, PointCloud):
            rc = self._lib.msplat_upload_point_cloud(self._ctx, pointCloud.handle)
            self._n = pointCloud.GetNumPoints()
        else:
            a = np.ascontiguousarray(pointCloud, np.float32)
            assert a.ndim == 2 and a.shape[1] == 8
            rc = self._lib.msplat_upload_points(self._ctx, a.ctypes.data, a.shape[0], 32, 0, 16)
            self._n = a.shape[0]
        if rc != _capi.OK:
            self._err = self._lib.msplat_last_error(self._ctx).decode()
            return False
        if not self._attach_all():
            return False
        return self.set_sprite(sprite)

    def set_sprite(self, sprite):
        for h in self._ctxs:
            if sprite is None:
                rc = self._lib.msplat_set_point_sprite(h, None, 0, 0)
            else:
                t = np.ascontiguousarray(sprite, np.uint8)
                assert t.ndim == 3 and t.shape[2] == 4
                rc = self._lib.msplat_set_point_sprite(h, t.ctypes.data, t.shape[1], t.shape[0])
            if rc != _capi.OK:
                self._err = self._lib.msplat_last_error(h).decode()
                return False
        return True

    def Render(self, cameraMat, projMat, viewport, nearFar, out=None, out_ptr=None, pitch_bytes=0):
        self.Sort(cameraMat, projMat, viewport, nearFar)
        return SplatRenderer.Render(self, cameraMat, projMat, viewport, nearFar, out=out, out_ptr=out_ptr,
                                    pitch_bytes=pitch_bytes)
