"""ctypes face of the C oracle (oracle/msplat_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.  PARITY PINNED against the reference's shaders run on llvmpipe (see msplat_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Splat2D(C.Structure):
    _fields_ = [
        ("px", C.c_float), ("py", C.c_float),
        ("cov", C.c_float * 4), ("inv", C.c_float * 4),
        ("rgb", C.c_float * 3), ("alpha", C.c_float),
        ("ndc", C.c_float * 3), ("depth", C.c_float),
        ("hx", C.c_float), ("hy", C.c_float),
        ("reject", C.c_int32), ("index", C.c_uint32),
    ]


SPLAT2D_DTYPE = np.dtype([
    ("px", "<f4"), ("py", "<f4"), ("cov", "<f4", (4,)), ("inv", "<f4", (4,)),
    ("rgb", "<f4", (3,)), ("alpha", "<f4"), ("ndc", "<f4", (3,)), ("depth", "<f4"),
    ("hx", "<f4"), ("hy", "<f4"), ("reject", "<i4"), ("index", "<u4"),
])
assert SPLAT2D_DTYPE.itemsize == C.sizeof(Splat2D)


def build():
    """Compile liboracle.so (and oracle/_ref when the reference checkout exists)."""
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    fp = C.POINTER(C.c_float)
    u32p = C.POINTER(C.c_uint32)
    L.orc_mat4_mul.argtypes = [fp, fp, fp]
    L.orc_mat4_inverse.argtypes = [fp, fp]
    L.orc_perspective.argtypes = [C.c_float] * 4 + [fp]
    L.orc_create_projection.argtypes = [C.c_float] * 6 + [fp]
    L.orc_build_cloud.argtypes = [C.c_size_t, fp, fp, fp, fp, fp, fp, C.c_int, fp]
    L.orc_presort.argtypes = [C.c_size_t, fp, C.c_size_t, fp, C.c_float, u32p, u32p]
    L.orc_presort.restype = C.c_uint32
    L.orc_cull_key.argtypes = [fp, fp, C.c_float, u32p]
    L.orc_cull_key.restype = C.c_int
    L.orc_sort.argtypes = [C.c_uint32, u32p, u32p]
    L.orc_project.argtypes = [C.c_uint32, u32p, fp, C.c_size_t, C.c_int, C.c_int, fp, fp, fp, fp, fp,
                              C.c_void_p]
    L.orc_composite.argtypes = [C.c_uint32, C.c_void_p, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_int]
    L.orc_composite_flip.argtypes = [C.c_uint32, C.c_void_p, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_int, fp, C.c_float]
    L.orc_composite_depth.argtypes = [C.c_uint32, C.c_void_p, C.c_int, C.c_int, fp, C.c_int, C.c_int]
    L.orc_composite_rop.argtypes = [C.c_uint32, C.c_void_p, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_int]
    L.orc_quantise_depth.argtypes = [C.c_float, C.c_int]
    L.orc_quantise_depth.restype = C.c_uint32
    L.orc_build_sprite.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, fp, u32p]
    L.orc_build_sprite.restype = C.c_int
    L.orc_points_project.argtypes = [C.c_uint32, u32p, fp, fp, fp, fp, C.c_int, C.c_int, C.c_void_p]
    L.orc_points_composite.argtypes = [C.c_uint32, C.c_void_p, fp, u32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp,
                                       C.c_int]
    L.orc_composite_f64.argtypes = [C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]
    L.orc_render_frame.argtypes = [C.c_size_t, fp, C.c_size_t, C.c_int, C.c_int, fp, fp, fp, fp, fp, fp,
                                   fp, u32p, u32p, C.c_void_p, C.c_int]
    L.orc_render_frame.restype = C.c_uint32
    L.orc_last_fragment_count.restype = C.c_uint64
    L.orc_render_frame_tiled.argtypes = [C.c_size_t, fp, C.c_size_t, C.c_int, C.c_int, fp, fp, fp, fp, fp, fp, fp, u32p, u32p,
                                         C.c_float, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.orc_render_frame_tiled.restype = C.c_uint32
    _LIB = L
    return L


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _u(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def mat4_mul(a, b):
    a, pa = _f(a); b, pb = _f(b)
    out = np.empty(16, np.float32)
    lib().orc_mat4_mul(pa, pb, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def mat4_inverse(m):
    m, pm = _f(m)
    out = np.empty(16, np.float32)
    lib().orc_mat4_inverse(pm, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def perspective(fovy, aspect, zn, zf):
    out = np.empty(16, np.float32)
    lib().orc_perspective(fovy, aspect, zn, zf, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def create_projection(tanL, tanR, tanU, tanD, zn, zf):
    out = np.empty(16, np.float32)
    lib().orc_create_projection(tanL, tanR, tanU, tanD, zn, zf, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def build_cloud(xyz, f_dc, f_rest, opacity, log_scale, rot, full_sh):
    n = xyz.shape[0]
    stride = 61 if full_sh else 25
    xyz, p0 = _f(xyz); f_dc, p1 = _f(f_dc)
    if f_rest is None:
        f_rest = np.zeros((n, 45), np.float32)
    f_rest, p2 = _f(f_rest); opacity, p3 = _f(opacity); log_scale, p4 = _f(log_scale); rot, p5 = _f(rot)
    out = np.empty((n, stride), np.float32)
    lib().orc_build_cloud(n, p0, p1, p2, p3, p4, p5, int(bool(full_sh)),
                          out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def presort(aos, mvp, zfar):
    aos, pa = _f(aos)
    mvp, pm = _f(mvp)
    n, stride = aos.shape
    keys = np.empty(max(n, 1), np.uint32)
    idx = np.empty(max(n, 1), np.uint32)
    v = lib().orc_presort(n, pa, stride, pm, zfar, _u(keys), _u(idx))
    return keys[:v].copy(), idx[:v].copy()


def sort(keys, idx):
    keys = np.ascontiguousarray(keys, np.uint32).copy()
    idx = np.ascontiguousarray(idx, np.uint32).copy()
    lib().orc_sort(keys.shape[0], _u(keys), _u(idx))
    return keys, idx


def project(idx, aos, full_sh, srgb, viewMat, projMat, viewport, nearFar, eye):
    aos, pa = _f(aos)
    idx = np.ascontiguousarray(idx, np.uint32)
    v = idx.shape[0]
    out = np.zeros(max(v, 1), SPLAT2D_DTYPE)
    vm, pv = _f(viewMat); pm_, pp = _f(projMat); vp, pvp = _f(viewport); nf, pnf = _f(nearFar); e, pe = _f(eye)
    lib().orc_project(v, _u(idx), pa, aos.shape[1], int(bool(full_sh)), int(bool(srgb)), pv, pp, pvp, pnf, pe,
                      out.ctypes.data)
    return out[:v]


def composite(splats, W, H, nthreads=1, row0=0, row1=None):
    splats = np.ascontiguousarray(splats)
    assert splats.dtype == SPLAT2D_DTYPE
    rgba = np.zeros((H, W, 4), np.float32)
    lib().orc_composite(splats.shape[0], splats.ctypes.data, W, H,
                        rgba.ctypes.data_as(C.POINTER(C.c_float)), row0, H if row1 is None else row1, nthreads)
    return rgba


def composite_flip(splats, W, H, nthreads=1, row0=0, row1=None, flip_rel=1e-4):
    """composite() plus the per-pixel threshold-flip budget: returns (rgba, budget) where budget[y, x] bounds how far
    the pixel can move when fragments with |w - 1/256| <= flip_rel/256 land on the other side of the discard test"""
    splats = np.ascontiguousarray(splats)
    assert splats.dtype == SPLAT2D_DTYPE
    rgba = np.zeros((H, W, 4), np.float32)
    budget = np.zeros((H, W), np.float32)
    lib().orc_composite_flip(splats.shape[0], splats.ctypes.data, W, H, rgba.ctypes.data_as(C.POINTER(C.c_float)),
                             row0, H if row1 is None else row1, nthreads,
                             budget.ctypes.data_as(C.POINTER(C.c_float)), flip_rel)
    return rgba, budget


def composite_depth(splats, W, H, depth_bits=24, nthreads=1):
    """orc_composite with the reference's enabled GL_LESS depth test against an emulated depth buffer"""
    splats = np.ascontiguousarray(splats)
    assert splats.dtype == SPLAT2D_DTYPE
    rgba = np.zeros((H, W, 4), np.float32)
    lib().orc_composite_depth(splats.shape[0], splats.ctypes.data, W, H,
                              rgba.ctypes.data_as(C.POINTER(C.c_float)), depth_bits, nthreads)
    return rgba


def composite_rop(splats, W, H, rop, depth_bits=0, nthreads=1):
    """the blend as the render target performs it after every splat: rop 1 = RGBA8 (clamp, 8-bit unorm), 2 = RGBA16F"""
    splats = np.ascontiguousarray(splats)
    assert splats.dtype == SPLAT2D_DTYPE
    rgba = np.zeros((H, W, 4), np.float32)
    lib().orc_composite_rop(splats.shape[0], splats.ctypes.data, W, H, rgba.ctypes.data_as(C.POINTER(C.c_float)),
                            depth_bits, rop, nthreads)
    return rgba


POINT2D_DTYPE = np.dtype([("cx", "<f4"), ("cy", "<f4"), ("hx", "<f4"), ("hy", "<f4"), ("rgba", "<f4", (4,)),
                          ("lambda", "<f4"), ("ndcz", "<f4"), ("reject", "<i4"), ("index", "<u4")])


def points_frame(points, sprite_rgba8, cam, proj, viewport, nearFar, srgb=False, depth_bits=0):
    """PointRenderer::Render (pointrenderer.cpp:113-196): presort + stable sort of the positions, then the sprites
    in draw order.  points = (N, 8) float32 (position.xyzw, color.rgba); sprite_rgba8 = (h, w, 4) uint8, top row
    first as decoded from the PNG.  Returns dict(V, image, sorted_idx, pts)."""
    points, pp = _f(points)
    n = points.shape[0]
    L = lib()
    view = mat4_inverse(cam)
    mvp = mat4_mul(proj, view)
    keys = np.empty(max(n, 1), np.uint32); idx = np.empty(max(n, 1), np.uint32)
    _, pm = _f(mvp)
    v = L.orc_presort(n, pp, 8, pm, float(nearFar[1]), _u(keys), _u(idx))
    L.orc_sort(v, _u(keys), _u(idx))
    tex = np.ascontiguousarray(sprite_rgba8, np.uint8)
    th, tw = tex.shape[:2]
    chain = np.zeros((tw * th * 4 // 3 + 64) * 4, np.float32)
    off = np.zeros(14, np.uint32)
    levels = L.orc_build_sprite(tex.ctypes.data, tw, th, int(bool(srgb)), chain.ctypes.data_as(C.POINTER(C.c_float)), _u(off))
    pts = np.zeros(max(v, 1), POINT2D_DTYPE)
    _, pv = _f(view); _, ppj = _f(proj); _, pvp = _f(viewport)
    L.orc_points_project(v, _u(idx), pp, pv, ppj, pvp, tw, th, pts.ctypes.data)
    W, H = int(viewport[2]), int(viewport[3])
    img = np.zeros((H, W, 4), np.float32)
    L.orc_points_composite(v, pts.ctypes.data, chain.ctypes.data_as(C.POINTER(C.c_float)), _u(off), tw, th, levels, W, H,
                           img.ctypes.data_as(C.POINTER(C.c_float)), depth_bits)
    return dict(V=v, image=img, sorted_idx=idx[:v].copy(), pts=pts[:v], levels=levels)


def composite_f64(splats, W, H, nthreads=1):
    splats = np.ascontiguousarray(splats)
    rgba = np.zeros((H, W, 4), np.float64)
    lib().orc_composite_f64(splats.shape[0], splats.ctypes.data, W, H,
                            rgba.ctypes.data_as(C.POINTER(C.c_double)), nthreads)
    return rgba


def render_frame(aos, full_sh, sort_cam, sort_proj, viewport, nearFar, render_cam=None, render_proj=None,
                 srgb=False, nthreads=1, want_image=True, want_splats=False):
    """Whole Sort()+Render().  Returns dict(V, image, sorted_idx, sorted_keys, splats)."""
    aos, pa = _f(aos)
    n, stride = aos.shape
    if render_cam is None:
        render_cam = sort_cam
    if render_proj is None:
        render_proj = sort_proj
    sc, p_sc = _f(sort_cam); sp, p_sp = _f(sort_proj); rc, p_rc = _f(render_cam); rp, p_rp = _f(render_proj)
    vp, p_vp = _f(viewport); nf, p_nf = _f(nearFar)
    W, H = int(viewport[2]), int(viewport[3])
    img = np.zeros((H, W, 4), np.float32) if want_image else None
    sidx = np.empty(max(n, 1), np.uint32)
    skeys = np.empty(max(n, 1), np.uint32)
    splats = np.zeros(max(n, 1), SPLAT2D_DTYPE) if want_splats else None
    v = lib().orc_render_frame(n, pa, stride, int(bool(full_sh)), int(bool(srgb)), p_sc, p_sp, p_rc, p_rp,
                               p_vp, p_nf,
                               img.ctypes.data_as(C.POINTER(C.c_float)) if want_image else None,
                               _u(sidx), _u(skeys),
                               splats.ctypes.data if want_splats else None, nthreads)
    return dict(V=v, image=img, sorted_idx=sidx[:v].copy(), sorted_keys=skeys[:v].copy(),
                splats=splats[:v] if want_splats else None,
                fragments=int(lib().orc_last_fragment_count()) if want_image else 0)


def render_frame_tiled(aos, full_sh, sort_cam, sort_proj, viewport, nearFar, render_cam=None, render_proj=None,
                       srgb=False, nthreads=1, t_eps=2.0 ** -14, row0=0, row1=None, image=None):
    """The timed CPU baseline (msplat_cpu_tiled.c): tile-binned, front-to-back, multi-threaded Sort()+Render().
    Returns dict(V, image, sorted_idx, sorted_keys, stages_ms).  `image` may be a preallocated (H, W, 4) float32 array."""
    aos, pa = _f(aos)
    n, stride = aos.shape
    if render_cam is None:
        render_cam = sort_cam
    if render_proj is None:
        render_proj = sort_proj
    sc, p_sc = _f(sort_cam); sp, p_sp = _f(sort_proj); rc, p_rc = _f(render_cam); rp, p_rp = _f(render_proj)
    vp, p_vp = _f(viewport); nf, p_nf = _f(nearFar)
    W, H = int(viewport[2]), int(viewport[3])
    img = np.zeros((H, W, 4), np.float32) if image is None else image
    assert img.dtype == np.float32 and img.shape == (H, W, 4) and img.flags.c_contiguous
    sidx = np.empty(max(n, 1), np.uint32)
    skeys = np.empty(max(n, 1), np.uint32)
    st = (C.c_double * 6)()
    v = lib().orc_render_frame_tiled(n, pa, stride, int(bool(full_sh)), int(bool(srgb)), p_sc, p_sp, p_rc, p_rp, p_vp, p_nf,
                                     img.ctypes.data_as(C.POINTER(C.c_float)), _u(sidx), _u(skeys), float(t_eps),
                                     int(nthreads), int(row0), int(H if row1 is None else row1), st)
    if v == 0xFFFFFFFF:
        raise MemoryError("orc_render_frame_tiled: out of memory")
    return dict(V=v, image=img, sorted_idx=sidx[:v].copy(), sorted_keys=skeys[:v].copy(),
                stages_ms=dict(zip(("cull", "sort", "project", "bin", "composite", "total"), list(st))))


# ---- reference PLY parser (oracle/_ref, built from /root/reference sources) ------------------
_REF = None


def ref_ply_lib():
    """Returns the ctypes handle of oracle/_ref/libref_ply.so or None if it was never built."""
    global _REF
    if _REF is not None:
        return _REF
    path = os.path.join(_HERE, "_ref", "libref_ply.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.ref_ply_open.argtypes = [C.c_char_p]
    L.ref_ply_open.restype = C.c_void_p
    L.ref_ply_close.argtypes = [C.c_void_p]
    L.ref_ply_vertex_count.argtypes = [C.c_void_p]
    L.ref_ply_vertex_count.restype = C.c_ulonglong
    L.ref_ply_vertex_size.argtypes = [C.c_void_p]
    L.ref_ply_vertex_size.restype = C.c_ulonglong
    L.ref_ply_get_property.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int),
                                       C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.ref_ply_get_property.restype = C.c_int
    L.ref_ply_copy_vertices.argtypes = [C.c_void_p, C.c_void_p]
    _REF = L
    return L


def ref_ply_read(path, names):
    """Parse `path` with the REAL reference parser.  Returns (count, vertex_size, props, raw bytes)."""
    L = ref_ply_lib()
    assert L is not None
    h = L.ref_ply_open(path.encode())
    if not h:
        return None
    try:
        cnt = L.ref_ply_vertex_count(h)
        vs = L.ref_ply_vertex_size(h)
        props = {}
        for nm in names:
            t = C.c_int(); s = C.c_ulonglong(); o = C.c_ulonglong()
            if L.ref_ply_get_property(h, nm.encode(), C.byref(t), C.byref(s), C.byref(o)):
                props[nm] = (t.value, s.value, o.value)
        raw = np.empty(cnt * vs, np.uint8)
        if cnt:
            L.ref_ply_copy_vertices(h, raw.ctypes.data)
        return cnt, vs, props, raw
    finally:
        L.ref_ply_close(h)
