"""GaussianCloud: Python face of the C++ GaussianCloud/Ply re-implementation
(splatapult_amd/host/gaussian_scene.*), mirroring /root/reference/src/gaussiancloud.h:17-91
(same method names, argument meaning, bool-return error convention)."""
import ctypes as C

import numpy as np

from . import _capi


class GaussianCloud:
    class Options:
        def __init__(self, importFullSH=True, exportFullSH=True):
            self.importFullSH = bool(importFullSH)
            self.exportFullSH = bool(exportFullSH)

    def __init__(self, options=None):
        self.opt = options or GaussianCloud.Options()
        self._lib = _capi.lib()
        self._h = self._lib.msplat_cloud_create(1 if self.opt.importFullSH else 0)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.msplat_cloud_destroy(h)

    # -- reference surface ------------------------------------------------------------------
    def ImportPly(self, plyFilename):
        """gaussiancloud.cpp:138-365; returns False (after logging) on failure"""
        return self._lib.msplat_cloud_import_ply(self._h, str(plyFilename).encode()) == _capi.OK

    def ExportPly(self, plyFilename):
        return self._lib.msplat_cloud_export_ply(self._h, str(plyFilename).encode()) == _capi.OK

    def InitDebugCloud(self):
        self._lib.msplat_cloud_init_debug(self._h)

    def PruneSplats(self, origin, numGaussians):
        o = (C.c_float * 3)(*[float(v) for v in origin])
        self._lib.msplat_cloud_prune(self._h, o, int(numGaussians))

    def GetNumGaussians(self):
        return int(self._lib.msplat_cloud_num_gaussians(self._h))

    def GetStride(self):
        return int(self._lib.msplat_cloud_stride(self._h))

    def GetTotalSize(self):
        return int(self._lib.msplat_cloud_total_size(self._h))

    def GetRawDataPtr(self):
        return self._lib.msplat_cloud_raw_data(self._h)

    def HasFullSH(self):
        return bool(self._lib.msplat_cloud_has_full_sh(self._h))

    def GetAttribOffsets(self):
        off = _capi.AttrOffsets()
        self._lib.msplat_cloud_attr_offsets(self._h, C.byref(off))
        return off

    # -- extensions -------------------------------------------------------------------------
    def FromAttributes(self, xyz, f_dc, f_rest, opacity, log_scale, rot):
        """Same per-vertex math as ImportPly's lambda (gaussiancloud.cpp:254-361) on arrays."""
        def f(a):
            a = np.ascontiguousarray(a, np.float32)
            return a, a.ctypes.data_as(C.POINTER(C.c_float))
        xyz, p0 = f(xyz); f_dc, p1 = f(f_dc); opacity, p3 = f(opacity); log_scale, p4 = f(log_scale); rot, p5 = f(rot)
        p2 = None
        if f_rest is not None:
            f_rest, p2 = f(f_rest)
        rc = self._lib.msplat_cloud_from_attributes(self._h, xyz.shape[0], p0, p1, p2, p3, p4, p5)
        return rc == _capi.OK

    def as_array(self):
        """(N, stride/4) float32 copy of the interleaved records (GetRawDataPtr/GetStride/GetTotalSize)."""
        n, stride = self.GetNumGaussians(), self.GetStride()
        if n == 0:
            return np.zeros((0, stride // 4), np.float32)
        buf = (C.c_float * (n * stride // 4)).from_address(self.GetRawDataPtr())
        return np.frombuffer(buf, np.float32).reshape(n, stride // 4).copy()

    @property
    def handle(self):
        return self._h
