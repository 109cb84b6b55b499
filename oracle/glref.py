"""The reference's OWN shaders, executed (test infrastructure; see oracle/glref/glref.c).

`presort` runs /root/reference/shader/presort_compute.glsl, `render` runs splat_vert / splat_geom / splat_frag through Mesa
llvmpipe's rasteriser and blender with the GL state of the reference's App -- on the machine where the reference checkout and
Mesa's software rasteriser exist (the build container).  The GPU box has neither: there the committed fixtures
tests/golden/glref_*.npz (made by tests/golden/make_glref_golden.py from these calls) stand in."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libglref.so")
SHADER_DIR = os.environ.get("MSPLAT_REFERENCE_SHADERS", "/root/reference/shader")
_lib = None


def available():
    return os.path.exists(LIB_PATH) and os.path.exists(os.path.join(SHADER_DIR, "splat_vert.glsl"))


def _f(a):
    return np.ascontiguousarray(a, np.float32).ctypes.data_as(C.POINTER(C.c_float))


def _u(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def init(full_sh, srgb=False):
    """compiles the reference's programs with SplatRenderer::Init's defines (FULL_SH, FRAMEBUFFER_SRGB); returns the GL version string"""
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB_PATH)
        _lib.glref_last_error.restype = C.c_char_p
        _lib.glref_gl_version.restype = C.c_char_p
    if _lib.glref_init(SHADER_DIR.encode(), 1 if full_sh else 0, 1 if srgb else 0) != 0:
        raise RuntimeError("glref_init: " + _lib.glref_last_error().decode())
    return _lib.glref_gl_version().decode()


def presort(aos, mvp, near_far, raw=False):
    """SplatRenderer::Sort's pre-sort on the reference's compute shader: (keys, indices) of the visible splats, ordered by index
    (the shader hands out slots with an atomic counter: its own order is not deterministic); raw: in the shader's slot order --
    what the reference's sorter is fed"""
    n = aos.shape[0]
    pos4 = np.ascontiguousarray(np.c_[aos[:, :3], np.ones(n, np.float32)], np.float32)     # posVec, splatrenderer.cpp:106-111
    keys, idx, cnt = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32), C.c_uint32()
    if _lib.glref_presort(_f(pos4), n, _f(mvp), _f(near_far), _u(keys), _u(idx), C.byref(cnt)) != 0:
        raise RuntimeError("glref_presort: " + _lib.glref_last_error().decode())
    if raw:
        return keys[:cnt.value].copy(), idx[:cnt.value].copy()
    o = np.argsort(idx[:cnt.value], kind="stable")
    return keys[:cnt.value][o].copy(), idx[:cnt.value][o].copy()


RGC_HPP = os.path.join(os.path.dirname(SHADER_DIR.rstrip("/")), "src", "radix_sort.hpp")


def rgc_sort(keys, values):
    """the reference's fallback sorter (rgc::radix_sort::sorter::sort, src/radix_sort.hpp:340-485) on its own compute shaders -- the
    string literals of that header, read where it lies: (sorted keys, values carried along), uint32"""
    k = np.ascontiguousarray(keys, np.uint32).copy()
    v = np.ascontiguousarray(values, np.uint32).copy()
    assert k.shape == v.shape and k.ndim == 1
    if _lib.glref_rgc_sort(RGC_HPP.encode(), _u(k), _u(v), k.shape[0]) != 0:
        raise RuntimeError("glref_rgc_sort: " + _lib.glref_last_error().decode())
    return k, v


def render(aos, sorted_idx, view_mat, proj_mat, viewport, near_far, eye, target="fp32", depth_bits=0):
    """SplatRenderer::Render on the reference's vertex / geometry / fragment shaders: (H, W, 4) float32, row 0 = bottom.
    target: "fp32" = the colour-only RGBA32F FBO of --fp32, "rgba8" = an 8-bit target like the default back buffer (GL's ROP
    clamps + quantises after every blend), "fp16" = the --fp16 target; depth_bits 24 / 32: with a depth attachment (GL_DEPTH_TEST live)"""
    W, H = int(viewport[2]), int(viewport[3])
    img = np.zeros((H, W, 4), np.float32)
    aos = np.ascontiguousarray(aos, np.float32)
    si = np.ascontiguousarray(sorted_idx, np.uint32)
    tgt = {"fp32": 0, "rgba8": 1, "fp16": 2}[target]
    if _lib.glref_render_target(_f(aos), aos.shape[0], _u(si), si.shape[0], _f(view_mat), _f(proj_mat), _f(viewport), _f(near_far), _f(eye),
                                tgt, int(depth_bits), img.ctypes.data_as(C.POINTER(C.c_float))) != 0:
        raise RuntimeError("glref_render: " + _lib.glref_last_error().decode())
    return img


def uploaded_sprite(sprite_rgba8):
    """what Image::Load leaves in Image::data for an RGBA PNG decoded top row first (src/core/image.cpp:108-114, 128-158): rows
    flipped, colour bytes replaced by uint8((c / 255.0f * (a / 255.0f)) * 255.0f) in float arithmetic (truncating)"""
    t = np.ascontiguousarray(sprite_rgba8[::-1], np.uint8).copy()
    a = t[..., 3].astype(np.float32) / np.float32(255.0)
    for c in range(3):
        t[..., c] = ((t[..., c].astype(np.float32) / np.float32(255.0)) * a * np.float32(255.0)).astype(np.uint8)
    return t


def points_render(points, sorted_idx, model_view, proj_mat, viewport, sprite_rgba8, srgb=False, depth_bits=0, want_mips=False):
    """PointRenderer::Render's draw on the reference's point shaders (pointrenderer.cpp:168-195).  points: (N, 8) float32
    (position.xyzw, color.rgba); sprite_rgba8: (h, w, 4) uint8 as decoded from the PNG (top row first).  Returns the (H, W, 4)
    float32 image, row 0 = bottom (and the list of mip levels llvmpipe's glGenerateMipmap made, when asked)."""
    W, H = int(viewport[2]), int(viewport[3])
    img = np.zeros((H, W, 4), np.float32)
    pts = np.ascontiguousarray(points, np.float32)
    si = np.ascontiguousarray(sorted_idx, np.uint32)
    up = uploaded_sprite(sprite_rgba8)
    th, tw = up.shape[:2]
    sizes, w, h = [], tw, th
    while True:
        sizes.append((h, w))
        if w == 1 and h == 1:
            break
        w, h = max(1, w // 2), max(1, h // 2)
    mips = np.zeros(sum(a * b for a, b in sizes) * 4, np.float32) if want_mips else None
    if _lib.glref_points_render(_f(pts), pts.shape[0], _u(si), si.shape[0], _f(model_view), _f(proj_mat), _f(viewport),
                                up.ctypes.data_as(C.POINTER(C.c_uint8)), tw, th, 1 if srgb else 0, int(depth_bits),
                                img.ctypes.data_as(C.POINTER(C.c_float)),
                                mips.ctypes.data_as(C.POINTER(C.c_float)) if want_mips else None) != 0:
        raise RuntimeError("glref_points_render: " + _lib.glref_last_error().decode())
    if not want_mips:
        return img
    out, o = [], 0
    for (h, w) in sizes:
        out.append(mips[o:o + h * w * 4].reshape(h, w, 4).copy())
        o += h * w * 4
    return img, out
