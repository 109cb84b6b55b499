"""Camera / projection helpers for headless drivers (bench, tests).  Matrices are float32[16],
column-major like glm.  Constants from /root/reference/src/app.cpp:73-75."""
import ctypes as C
import json
import math

import numpy as np

from . import _capi

Z_NEAR = 0.1
Z_FAR = 1000.0
FOVY = math.radians(45.0)


def _out16():
    a = np.zeros(16, np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def perspective(fovy, aspect, zn=Z_NEAR, zf=Z_FAR):
    """glm::perspective as called at app.cpp:1042"""
    a, p = _out16()
    _capi.lib().msplat_perspective(fovy, aspect, zn, zf, p)
    return a


def create_projection(tanL, tanR, tanU, tanD, zn=Z_NEAR, zf=Z_FAR):
    """asymmetric XR frustum, util.cpp:420-480"""
    a, p = _out16()
    _capi.lib().msplat_create_projection(tanL, tanR, tanU, tanD, zn, zf, p)
    return a


def pose(position=(0.0, 0.0, 0.0), yaw=0.0, pitch=0.0):
    """camera-to-world matrix: rotation yaw (about +Y) then pitch (about camera X), looks down -Z"""
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], np.float64)
    rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]], np.float64)
    m = np.eye(4)
    m[:3, :3] = ry @ rx
    m[:3, 3] = position
    return m.T.astype(np.float32).reshape(16).copy()      # column-major


def orbit(radius, angle, height=0.0):
    """camera on a circle of `radius` around the origin in the XZ plane, looking at the origin"""
    x, z = radius * math.sin(angle), radius * math.cos(angle)
    pitch = -math.atan2(height, radius)
    return pose((x, height, z), yaw=angle, pitch=pitch)


def translate_local(cam, dx=0.0, dy=0.0, dz=0.0):
    """move a camera-to-world matrix along its own axes (stereo eye offsets)"""
    m = np.asarray(cam, np.float32).reshape(4, 4).copy()     # m[c] = column c
    m[3, :3] = m[3, :3] + dx * m[0, :3] + dy * m[1, :3] + dz * m[2, :3]
    return m.reshape(16)


def load_vr_json(path):
    """VrConfig::ImportJson (vrconfig.cpp:20-46): floor matrix, float32[16] column-major"""
    a, p = _out16()
    rc = _capi.lib().msplat_vrconfig_import_json(str(path).encode(), p)
    if rc != _capi.OK:
        raise IOError("cannot read vr config %s" % path)
    return a


def camera_from_vr_json(path, raise_by=1.5):
    """default desktop camera when only a *_vr.json exists: floorMat raised 1.5 along its Y (app.cpp:486-497)"""
    cam = load_vr_json(path).reshape(4, 4).copy()            # cam[c] = column c
    cam[3, :3] = cam[3, :3] + raise_by * cam[1, :3]          # pos += mat3(floorMat) * (0, 1.5, 0)
    return cam.reshape(16)


def load_cameras_json(path):
    """CamerasConfig::ImportJson (camerasconfig.cpp:20-67): list of (camera-to-world float32[16], fov float32[2])"""
    L = _capi.lib()
    n = C.c_uint32()
    if L.msplat_cameras_import_json(str(path).encode(), None, None, 0, C.byref(n)) != _capi.OK:
        raise IOError("cannot read cameras config %s" % path)
    mats = np.zeros((max(n.value, 1), 16), np.float32)
    fovs = np.zeros((max(n.value, 1), 2), np.float32)
    L.msplat_cameras_import_json(str(path).encode(), mats.ctypes.data_as(C.POINTER(C.c_float)),
                                 fovs.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(n))
    return [(mats[i].copy(), fovs[i].copy()) for i in range(n.value)]


def find_config_file(ply_path, name):
    """FindConfigFile (app.cpp:89-119); returns '' when nothing is found"""
    buf = C.create_string_buffer(4096)
    rc = _capi.lib().msplat_find_config_file(str(ply_path).encode(), name.encode(), buf, 4096)
    return buf.value.decode() if rc == _capi.OK else ""


def write_image(path, rgba, encode_srgb=False):
    """float RGBA framebuffer (row 0 = bottom) -> PNG (or .ppm), 8-bit, top row first"""
    a = np.ascontiguousarray(rgba, np.float32)
    rc = _capi.lib().msplat_write_image(str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[1], a.shape[0],
                                        1 if encode_srgb else 0)
    if rc != _capi.OK:
        raise IOError("cannot write %s" % path)


def read_image(path):
    """8-bit PNG (gray / gray+alpha / RGB / RGBA, non-interlaced) -> (h, w, 4) uint8, top row first"""
    L = _capi.lib()
    w, h = C.c_uint32(), C.c_uint32()
    if L.msplat_read_image(str(path).encode(), None, 0, C.byref(w), C.byref(h)) != _capi.OK:
        raise IOError("cannot read %s" % path)
    out = np.zeros((h.value, w.value, 4), np.uint8)
    rc = L.msplat_read_image(str(path).encode(), out.ctypes.data, out.nbytes, C.byref(w), C.byref(h))
    if rc != _capi.OK:
        raise IOError("cannot read %s" % path)
    return out

