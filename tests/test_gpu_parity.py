"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, stage by stage.

Tolerances, stated once:
  keys, visible set, sorted permutation, tile lists ........ exact
  projected centre ......................................... <= 1e-3 px
  conic / colour ........................................... rel 1e-5 relative to max(|x|, 1e-3); measured: 0
                                                             (_check_projection prints the worst case of every call)
  fp32 framebuffer ......................................... >= 99.9 % of values within 1e-4,
                                                             mean |diff| <= 1e-4, max |diff| <= 5e-3
     A value above 5e-3 is accepted only when the oracle EXPLAINS it: orc_composite_flip reports per pixel
     how far fragments within 1e-4 (relative) of the w = 1/256 discard threshold can move it (one flip is
     worth <= w (|c| + |dst|), colours are unclamped SH so |c| can exceed 1); the test asserts
     |diff| <= 5e-3 + that budget for every pixel and prints how many pixels needed it.
     Early termination adds <= t_eps * |c| (t_eps = 2^-14).
  alpha channel ............................................ == 1 exactly
  fp16 framebuffer ......................................... vs fp32 oracle: 2e-3 + 1 fp16 ulp
"""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from splatapult_amd import SplatRenderer, camera
from tests import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KK = -0.5 * 1.4426950408889634


def bin_px():
    from splatapult_amd import _capi
    return _capi.lib().msplat_tile_size()


def make_renderer(cloud, srgb=False, **kw):
    r = SplatRenderer(device=0, **kw)
    assert r.Init(cloud, srgb, False), r.last_error()
    return r


TIGHT = 5e-3          # max abs per channel
# SURVEY 8c states rel 1e-5 for conic and colour: asserted as stated, relative to max(|x|, floor) because conic entries and SH
# colours pass through zero.  Measured on the MI355X (r4, every _check_projection call of the suite, 176 k splats): worst relative
# error 0 -- project_kernel evaluates the oracle's operation order without contraction, so these values are bit-identical.
CONIC_RTOL, CONIC_FLOOR = 1e-5, 1e-3
RGB_RTOL, RGB_FLOOR = 1e-5, 1e-3
WORST = {"conic": 0.0, "rgb": 0.0}


def check_image(img, ref, max_abs=TIGHT, mean_abs=1e-4, frac=0.999, tol=1e-4, budget=None):
    """budget: per-pixel threshold-flip allowance from orc.composite_flip (None = plain max_abs bound)"""
    assert img.shape == ref.shape
    d = np.abs(img.astype(np.float64) - ref.astype(np.float64))
    rgb = d[..., :3]
    assert np.isfinite(img).all()
    within = (rgb <= tol).mean()
    assert within >= frac, "only %.5f of values within %g (max %.3g)" % (within, tol, rgb.max())
    assert rgb.mean() <= mean_abs, "mean |diff| %.3g" % rgb.mean()
    worst = rgb.max(axis=-1)
    over = worst > max_abs
    if over.any():
        assert budget is not None, "max |diff| %.3g in %d pixel(s), no threshold-flip budget given" % (worst.max(), over.sum())
        unexplained = over & (worst > max_abs + budget.astype(np.float64))
        assert not unexplained.any(), "%d pixel(s) above %g not explained by a w ~ 1/256 flip (max %.3g, budget there %.3g)" % (
            unexplained.sum(), max_abs, worst[unexplained].max(), budget[unexplained].max())
        print("check_image: %d pixel(s) above %g, all within the oracle's threshold-flip budget (max %.3g)"
              % (over.sum(), max_abs, worst.max()))
    assert (img[..., 3] == 1.0).all()              # the kernels write alpha = 1 exactly (app.cpp:158-160: cleared to 1, blended to 1)


def check_fp16_image(img, ref, budget):
    """RGBA16F target against the fp32 oracle: 2e-3 + 1 fp16 ulp (the target is rounded once; the reference's
    ROP rounds after every blend), threshold flips explained like in check_image"""
    assert img.dtype == np.float16
    ulp = np.abs(ref[..., :3]) * 2.0 ** -10
    d = np.abs(img[..., :3].astype(np.float32) - ref[..., :3])
    assert (d <= 2e-3 + ulp + budget[..., None]).all(), "max excess %.3g" % (d - 2e-3 - ulp - budget[..., None]).max()
    assert (d <= 2e-3 + ulp).mean() > 0.999
    assert (img[..., 3] == 1).all()


_STORAGE_CACHE = {}


def storage_aos(r, aos):
    """(the cloud in the renderer's STORAGE order, the permutation slot -> upload index or None).  Large clouds are stored in
    Morton order (msplat_config.spatial_order) and splats with EQUAL depth keys are drawn in ascending storage slot: the
    reference's tie order is undefined (atomic slots, presort_compute.glsl:50); ours is reproduced by feeding the oracle the
    cloud in that order.  Everything the renderer reports is in upload numbering, so oracle indices are mapped with order[idx]."""
    order = r.storage_order()
    if order is None:
        return aos, None
    key = (aos.shape, float(aos[0, 0]), float(aos[-1, 5]), float(aos[aos.shape[0] // 2, 17]), int(order[:97].sum()), int(order[-97:].sum()))
    hit = _STORAGE_CACHE.get("last")
    if hit is None or hit[0] != key:
        _STORAGE_CACHE["last"] = hit = (key, np.ascontiguousarray(aos[order]))
    return hit[1], order


def oracle_frame(aos, full_sh, cam, proj, vp, nf, render_cam=None, render_proj=None, srgb=False, nthreads=16,
                 row0=0, row1=None, r=None):
    """oracle Sort + Render with the per-pixel threshold-flip budget: dict(V, image, budget, splats, sorted_*).
    r: the renderer whose storage order the oracle is to follow (storage_aos); sorted_idx comes back in upload numbering"""
    order = None
    if r is not None:
        aos, order = storage_aos(r, aos)
    ref = orc.render_frame(aos, full_sh, cam, proj, vp, nf, render_cam=render_cam, render_proj=render_proj, srgb=srgb,
                           nthreads=nthreads, want_image=False, want_splats=True)
    if order is not None:
        ref["sorted_idx"] = order[ref["sorted_idx"]]
    W, H = int(vp[2]), int(vp[3])
    ref["image"], ref["budget"] = orc.composite_flip(ref["splats"], W, H, nthreads=nthreads, row0=row0, row1=row1)
    return ref


def run_frame(cloud, view, full_sh=True, srgb=False, **kw):
    cam, proj, vp, nf = view
    r = make_renderer(cloud, srgb=srgb, **kw)
    r.Sort(cam, proj, vp, nf)
    img = r.Render(cam, proj, vp, nf)
    ref = oracle_frame(cloud.as_array(), full_sh, cam, proj, vp, nf, srgb=srgb, r=r)
    return r, img, ref


# ------------------------------------------------------------------------------------------------
# stage 1: cull + key + sort
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("n,seed", [(1, 1), (63, 2), (2048, 3), (2049, 4), (50000, 5)])
def test_sort_matches_oracle_exactly(n, seed):
    cloud = scenes.synth_cloud(n, seed)
    cam, proj, vp, nf = scenes.default_view(640, 480)
    r = make_renderer(cloud)
    r.Sort(cam, proj, vp, nf)
    aos = cloud.as_array()
    mvp = orc.mat4_mul(proj, orc.mat4_inverse(cam))
    keys, idx = orc.presort(aos, mvp, nf[1])
    keys, idx = orc.sort(keys, idx)
    assert r.sort_count() == keys.shape[0]
    np.testing.assert_array_equal(r.sorted_keys(), keys)
    np.testing.assert_array_equal(r.sorted_indices(), idx)


def test_sort_hard_cases_and_ties():
    a = scenes.hard_attrs()
    cloud = scenes.cloud_from_attrs(a)
    for yaw in (0.0, 0.7, 3.0):
        cam, proj, vp, nf = scenes.default_view(800, 600, yaw=yaw)
        r = make_renderer(cloud)
        r.Sort(cam, proj, vp, nf)
        mvp = orc.mat4_mul(proj, orc.mat4_inverse(cam))
        keys, idx = orc.sort(*orc.presort(cloud.as_array(), mvp, nf[1]))
        assert 0 < keys.shape[0] < a["xyz"].shape[0]         # something culled, something kept
        np.testing.assert_array_equal(r.sorted_keys(), keys)
        np.testing.assert_array_equal(r.sorted_indices(), idx)   # includes the equal-key tie rule


def test_sort_everything_culled_and_empty_cloud():
    cloud = scenes.synth_cloud(500, 9)
    cam, proj, vp, nf = scenes.default_view(320, 240, z=7.0, yaw=np.pi)   # looking away
    r = make_renderer(cloud)
    r.Sort(cam, proj, vp, nf)
    assert r.sort_count() == 0
    img = r.Render(cam, proj, vp, nf)
    assert (img[..., :3] == 0).all() and (img[..., 3] == 1).all()
    empty = np.zeros((0, 61), np.float32)
    r2 = SplatRenderer()
    assert r2.Init(empty)
    r2.Sort(cam, proj, vp, nf)
    assert r2.sort_count() == 0
    img = r2.Render(cam, proj, vp, nf)
    assert (img[..., :3] == 0).all() and (img[..., 3] == 1).all()


# ------------------------------------------------------------------------------------------------
# stage 2: projection
# ------------------------------------------------------------------------------------------------

def _check_projection(r, ref, W, H):
    rec, rect = r.debug_projected()
    sp = ref["splats"]
    assert rec.shape[0] == sp.shape[0]
    tx0, ty0, tx1, ty1 = rect & 255, (rect >> 8) & 255, (rect >> 16) & 255, rect >> 24
    drawn = tx0 <= tx1
    # a drawn splat is never one the geometry stage rejected
    assert not (drawn & (sp["reject"] != 0)).any()
    ok = sp["reject"] == 0
    np.testing.assert_allclose(rec[ok, 0], sp["px"][ok], atol=1e-3, rtol=0)
    np.testing.assert_allclose(rec[ok, 1], sp["py"][ok], atol=1e-3, rtol=0)
    inv = sp["inv"][ok]
    exp_conic = np.stack([KK * inv[:, 0], KK * (inv[:, 1] + inv[:, 2]), KK * inv[:, 3]], axis=1)
    # worst relative error over the splats (|x| below the absolute floor counts from the floor): printed (-rP shows it), so the
    # tolerance below is a measured one, not a guess
    def rel(a, b, floor):
        a, b = a.astype(np.float64), b.astype(np.float64)
        return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max()) if a.size else 0.0
    rc, rg = rel(rec[ok, 2:5], exp_conic, CONIC_FLOOR), rel(rec[ok, 6:9], sp["rgb"][ok], RGB_FLOOR)
    WORST["conic"], WORST["rgb"] = max(WORST["conic"], rc), max(WORST["rgb"], rg)
    print("projection parity: worst rel conic %.3g, colour %.3g over %d splats (suite so far: %.3g / %.3g)"
          % (rc, rg, int(ok.sum()), WORST["conic"], WORST["rgb"]))
    assert rc <= CONIC_RTOL, "conic rel error %.3g > %g" % (rc, CONIC_RTOL)
    assert rg <= RGB_RTOL, "colour rel error %.3g > %g" % (rg, RGB_RTOL)
    np.testing.assert_array_equal(rec[ok, 9], sp["alpha"][ok])
    # footprint: every pixel with w > 1/256 lies within rho*sqrt(cov_xx) of the centre
    alpha = sp["alpha"].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        rho2 = 2.0 * np.log(256.0 * alpha)
    vis = ok & (rho2 > 0)
    ex = np.sqrt(np.maximum(rho2, 0) * sp["cov"][:, 0])
    ey = np.sqrt(np.maximum(rho2, 0) * sp["cov"][:, 3])
    x0 = np.ceil(sp["px"] - ex - 0.5); x1 = np.floor(sp["px"] + ex - 0.5)
    y0 = np.ceil(sp["py"] - ey - 0.5); y1 = np.floor(sp["py"] + ey - 0.5)
    onscreen = vis & (x1 >= 0) & (y1 >= 0) & (x0 <= W - 1) & (y0 <= H - 1) & (x0 <= x1) & (y0 <= y1)
    # anything the oracle can light up must be drawn ...
    assert drawn[onscreen].all()
    B = bin_px()
    xs0 = np.clip(x0, 0, W - 1) // B; xs1 = np.clip(x1, 0, W - 1) // B
    ys0 = np.clip(y0, 0, H - 1) // B; ys1 = np.clip(y1, 0, H - 1) // B
    m = onscreen
    assert (tx0[m] <= xs0[m]).all() and (tx1[m] >= xs1[m]).all()
    assert (ty0[m] <= ys0[m]).all() and (ty1[m] >= ys1[m]).all()
    # ... and the rectangle is tight to within one tile
    assert (xs0[m] - tx0[m] <= 1).all() and (tx1[m] - xs1[m] <= 1).all()
    assert (ys0[m] - ty0[m] <= 1).all() and (ty1[m] - ys1[m] <= 1).all()
    return rect


@pytest.mark.parametrize("full_sh", [True, False])
def test_projection_matches_oracle(full_sh):
    cloud = scenes.synth_cloud(6000, 21, full_sh=full_sh, log_scale_mean=-3.2)
    view = scenes.default_view(640, 360, yaw=0.4, pitch=-0.1, x=3.0)
    r, img, ref = run_frame(cloud, view, full_sh=full_sh)
    _check_projection(r, ref, 640, 360)


def test_projection_hard_cases():
    cloud = scenes.cloud_from_attrs(scenes.hard_attrs())
    view = scenes.default_view(800, 600)
    r, img, ref = run_frame(cloud, view)
    _check_projection(r, ref, 800, 600)
    st = r.stats()
    assert st["drawn"] < st["sort_count"]          # guard band / near plane / alpha rejects exist


# ------------------------------------------------------------------------------------------------
# stage 3: tile lists (binning must reproduce the draw order inside every tile, exactly)
# ------------------------------------------------------------------------------------------------

def _expected_tile_lists(rect, tiles_x, tiles_y):
    lists = [[] for _ in range(tiles_x * tiles_y)]
    for rank, rc in enumerate(rect.tolist()):
        tx0, ty0, tx1, ty1 = rc & 255, (rc >> 8) & 255, (rc >> 16) & 255, rc >> 24
        if tx0 > tx1:
            continue
        for ty in range(ty0, ty1 + 1):
            for tx in range(tx0, tx1 + 1):
                lists[ty * tiles_x + tx].append(rank)
    return lists


@pytest.mark.parametrize("W,H", [(640, 360), (333, 211), (16, 16)])
def test_tile_lists_exact(W, H):
    cloud = scenes.cloud_from_attrs(scenes.hard_attrs(2500, 5))
    view = scenes.default_view(W, H)
    r, img, ref = run_frame(cloud, view)
    rec, rect = r.debug_projected()
    st = r.stats()
    ts, pairs = r.debug_tile_lists()
    exp = _expected_tile_lists(rect, st["tiles_x"], st["tiles_y"])
    assert st["pairs"] == sum(len(e) for e in exp)
    assert ts[-1] == st["pairs"]
    for t, e in enumerate(exp):
        got = pairs[ts[t]:ts[t + 1]] & 0xFFFFFF
        assert got.tolist() == e, "tile %d" % t


# ------------------------------------------------------------------------------------------------
# stage 4: framebuffer
# ------------------------------------------------------------------------------------------------

def test_image_test_ply_config1(golden_dir):
    """BASELINE config 1: data/test.ply, 640x480, --nosh, camera from test_vr.json"""
    import os
    from splatapult_amd import GaussianCloud
    gc = GaussianCloud(GaussianCloud.Options(False, False))
    assert gc.ImportPly(os.path.join(golden_dir, "test.ply"))
    cam = camera.camera_from_vr_json(os.path.join(golden_dir, "test_vr.json"))
    W, H = 640, 480
    view = (cam, camera.perspective(camera.FOVY, W / H), [0, 0, W, H], scenes.NF)
    r, img, ref = run_frame(gc, view, full_sh=False)
    assert ref["V"] == 16
    np.testing.assert_array_equal(r.sorted_indices(), ref["sorted_idx"])
    check_image(img, ref["image"], budget=ref["budget"])
    assert img[..., :3].max() > 0.9


@pytest.mark.parametrize("n,W,H,seed,full_sh", [(3000, 320, 240, 31, True), (20000, 640, 360, 32, True),
                                               (20000, 500, 301, 33, False)])
def test_image_matches_oracle(n, W, H, seed, full_sh):
    cloud = scenes.synth_cloud(n, seed, full_sh=full_sh, log_scale_mean=-3.3)
    view = scenes.default_view(W, H, yaw=0.2)
    r, img, ref = run_frame(cloud, view, full_sh=full_sh)
    check_image(img, ref["image"], budget=ref["budget"])


def test_image_exact_mode_no_early_termination():
    """t_epsilon = 0: no early-out; the only differences left are fp32 association + exp ulps"""
    cloud = scenes.synth_cloud(8000, 41, log_scale_mean=-3.0)
    view = scenes.default_view(400, 300)
    r, img, ref = run_frame(cloud, view, t_epsilon=0.0)
    check_image(img, ref["image"], mean_abs=2e-5, budget=ref["budget"])


def test_image_hard_cases():
    cloud = scenes.cloud_from_attrs(scenes.hard_attrs())
    for yaw, z in ((0.0, 7.0), (0.9, 3.0)):
        view = scenes.default_view(640, 480, yaw=yaw, z=z)
        r, img, ref = run_frame(cloud, view)
        check_image(img, ref["image"], budget=ref["budget"])


def test_image_srgb_flag():
    cloud = scenes.synth_cloud(4000, 51, log_scale_mean=-3.0)
    view = scenes.default_view(320, 240)
    r, img, ref = run_frame(cloud, view, srgb=True)
    ok = np.isfinite(ref["image"]).all(axis=-1)          # pow() of a negative colour is NaN in both
    assert ok.mean() > 0.5
    d = np.abs(img[ok] - ref["image"][ok])
    assert (d[..., :3] <= 1e-4).mean() >= 0.999
    assert (d[..., :3].max(axis=-1) <= TIGHT + ref["budget"][ok]).all()


def test_two_views_share_one_sort():
    """XR contract (app.cpp:603-607): Sort with view 0 only, Render both eyes"""
    cloud = scenes.synth_cloud(10000, 61, log_scale_mean=-3.2)
    W, H = 504, 560
    cam0 = camera.pose((0.0, 0.0, 7.0))
    eyes = [camera.translate_local(cam0, dx=-0.032), camera.translate_local(cam0, dx=+0.032)]
    projs = [camera.create_projection(-1.0, 0.8, 0.95, -0.95), camera.create_projection(-0.8, 1.0, 0.95, -0.95)]
    vp, nf = [0, 0, W, H], scenes.NF
    r = make_renderer(cloud)
    r.Sort(eyes[0], projs[0], vp, nf)
    for e in range(2):
        img = r.Render(eyes[e], projs[e], vp, nf)
        ref = oracle_frame(cloud.as_array(), True, eyes[0], projs[0], vp, nf, render_cam=eyes[e], render_proj=projs[e])
        check_image(img, ref["image"], budget=ref["budget"])


# ------------------------------------------------------------------------------------------------
# depth-buffer emulation: the GL_DEPTH_TEST the reference leaves enabled
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("bits", [24, 32])
def test_depth_test_emulation_matches_oracle(bits):
    cloud = scenes.synth_cloud(20000, 101, log_scale_mean=-3.0)
    W, H = 640, 360
    cam, proj, vp, nf = scenes.default_view(W, H, yaw=0.2)
    r = make_renderer(cloud)
    r.Sort(cam, proj, vp, nf)
    plain = r.Render(cam, proj, vp, nf)
    r.set_depth_test(bits)
    img = r.Render(cam, proj, vp, nf)                      # same sort, depth-tested composite
    ref = orc.render_frame(cloud.as_array(), True, cam, proj, vp, nf, nthreads=8, want_splats=True)
    check_image(plain, ref["image"])
    check_image(img, orc.composite_depth(ref["splats"], W, H, bits, nthreads=8))
    r.set_depth_test(0)
    np.testing.assert_array_equal(r.Render(cam, proj, vp, nf), plain)
    from splatapult_amd import MsplatError
    with pytest.raises(MsplatError):
        r.set_depth_test(16)


def test_depth_test_second_eye_artifact_and_bands():
    """the case the flag exists for: the second XR eye is drawn in the FIRST eye's depth order, so GL_LESS
    rejects fragments of splats that are nearer for eye 0 but farther for eye 1"""
    cloud = scenes.synth_cloud(15000, 102, log_scale_mean=-2.8)
    W, H = 504, 560
    cam0 = camera.pose((0.0, 0.0, 6.0))
    eye1 = camera.translate_local(camera.pose((0.0, 0.0, 6.0), yaw=0.35), dx=0.4)     # exaggerated second view
    proj = camera.create_projection(-0.8, 1.0, 0.95, -0.95)
    vp, nf = [0, 0, W, H], scenes.NF
    r = make_renderer(cloud)
    r.set_depth_test(24)
    r.Sort(cam0, proj, vp, nf)
    img = r.Render(eye1, proj, vp, nf)
    ref = orc.render_frame(cloud.as_array(), True, cam0, proj, vp, nf, render_cam=eye1, render_proj=proj, nthreads=8,
                           want_splats=True)
    exp = orc.composite_depth(ref["splats"], W, H, 24, nthreads=8)
    check_image(img, exp)
    assert np.abs(exp - ref["image"])[..., :3].max() > 0.05      # the artifact is really there
    # tile-row bands reassemble bit-exactly in depth mode too
    acc = np.zeros_like(img)
    for g in range(3):
        rb = make_renderer(cloud)
        rb.set_depth_test(24)
        rb.set_band(3, g)
        rb.Sort(cam0, proj, vp, nf)
        part = rb.Render(eye1, proj, vp, nf)
        rows = np.arange(H) // bin_px() % 3 == g
        acc[rows] = part[rows]
    np.testing.assert_array_equal(acc, img)


@pytest.mark.parametrize("rop,depth_bits,fb", [("rgba8", 0, "fp32"), ("rgba8", 24, "fp32"), ("fp16", 0, "fp16"), ("fp16", 0, "fp32")])
def test_render_target_emulation_matches_oracle(rop, depth_bits, fb):
    """SURVEY.md 8a-12 / VERDICT r1 missing #4: the GL app's default RGBA8 target clamps to [0,1] and stores 8-bit unorm after
    EVERY blend, its --fp16 target rounds to fp16 after every blend (src/app.cpp:1012-1020); msplat_set_target_emulation
    reproduces both on the draw-order compositor, against orc_composite_rop.  One rounding-boundary flip (w differs by an
    ulp between exp2 on the GPU and expf in the oracle) moves a value by one 8-bit / fp16 step."""
    cloud = scenes.synth_cloud(20000, 131, log_scale_mean=-3.0)
    W, H = 640, 360
    cam, proj, vp, nf = scenes.default_view(W, H, yaw=0.2)
    r = make_renderer(cloud, fb_format=fb)
    r.Sort(cam, proj, vp, nf)
    plain = r.Render(cam, proj, vp, nf)
    r.set_target_emulation(rop)
    if depth_bits:
        r.set_depth_test(depth_bits)
    img = r.Render(cam, proj, vp, nf).astype(np.float32)
    ref = orc.render_frame(cloud.as_array(), True, cam, proj, vp, nf, nthreads=8, want_splats=True)
    want = orc.composite_rop(ref["splats"], W, H, 1 if rop == "rgba8" else 2, depth_bits=depth_bits, nthreads=8)
    d = np.abs(img - want)[..., :3]
    step = 1.0 / 255.0 if rop == "rgba8" else np.maximum(np.abs(want[..., :3]) * 2.0 ** -10, 2.0 ** -14)
    assert (d <= 1e-6).mean() > 0.99, (d > 1e-6).mean()
    assert (d <= 2.0 * step + 1e-6).all(), d.max()
    assert (img[..., 3] == 1).all()
    if rop == "rgba8":
        assert np.allclose(img * 255.0, np.round(img * 255.0), atol=1e-4) and img.min() >= 0 and img.max() <= 1
        assert np.abs(img[..., :3] - np.clip(plain[..., :3].astype(np.float32), 0, 1)).max() > 0.02     # not the same picture
    r.set_target_emulation(None)
    r.set_depth_test(0)
    np.testing.assert_array_equal(r.Render(cam, proj, vp, nf), plain)
    from splatapult_amd import MsplatError, _capi
    with pytest.raises(MsplatError):
        _capi.check(r._ctx, _capi.lib().msplat_set_target_emulation(r._ctx, 7))


# ------------------------------------------------------------------------------------------------
# point-cloud renderer: PointRenderer::Render = presort + sort + textured sprites
# ------------------------------------------------------------------------------------------------

def random_points(n, seed):
    rng = np.random.default_rng(seed)
    pts = np.zeros((n, 8), np.float32)
    pts[:, :3] = rng.normal(0, 1.2, size=(n, 3))
    pts[:, 3] = 1.0
    pts[:, 4:7] = rng.integers(0, 256, size=(n, 3)).astype(np.float32) / np.float32(255.0)
    pts[:, 7] = 1.0
    return pts


@pytest.mark.parametrize("srgb,depth_bits,z", [(False, 0, 5.0), (True, 0, 5.0), (False, 24, 5.0), (False, 0, 1.5)])
def test_point_renderer_matches_oracle(srgb, depth_bits, z):
    from splatapult_amd import PointRenderer
    from tests.test_points import smooth_sprite
    pts = random_points(6000, 111)
    tex = smooth_sprite(64, 48, seed=2)                     # not square, not 2^k x 2^k all the way down
    W, H = 640, 360
    cam, proj, vp, nf = scenes.default_view(W, H, z=z, yaw=0.3)     # z = 1.5: inside the cloud, magnified sprites
    r = PointRenderer(device=0)
    assert r.Init(pts, srgb, sprite=tex), r.last_error()
    if depth_bits:
        r.set_depth_test(depth_bits)
    img = r.Render(cam, proj, vp, nf)
    ref = orc.points_frame(pts, tex, cam, proj, vp, nf, srgb=srgb, depth_bits=depth_bits)
    assert r.sort_count() == ref["V"]
    np.testing.assert_array_equal(r.sorted_indices(), ref["sorted_idx"])
    assert np.isfinite(img).all() and (img[..., :3].sum(axis=-1) > 0).mean() > 0.01
    d = np.abs(img - ref["image"])[..., :3]
    assert (d <= 1e-5).mean() >= 0.9999, (d > 1e-5).mean()
    assert d.max() <= 1e-3
    assert np.abs(img[..., 3] - 1.0).max() == 0


def test_point_renderer_debug_cloud_builtin_sprite_and_bands(tmp_path):
    from splatapult_amd import PointCloud, PointRenderer
    pc = PointCloud(False)
    pc.InitDebugCloud()
    W, H = 320, 240
    cam = camera.pose((0.4, 0.4, 2.5))
    proj = camera.perspective(camera.FOVY, W / H)
    vp, nf = [0, 0, W, H], scenes.NF
    r = PointRenderer(device=0)
    assert r.Init(pc, False)                                # built-in sphere sprite
    img = r.Render(cam, proj, vp, nf)
    assert r.sort_count() == 15
    lit = img[..., :3].sum(axis=-1) > 0
    assert 15 <= lit.sum() < 15 * 200                       # 15 small discs (half size 0.01 H / w ~ 1 px here)
    assert img[..., 0].max() > 0.5 and img[..., 1].max() > 0.5 and img[..., 2].max() > 0.5      # the three axis colours
    acc = np.zeros_like(img)
    for g in range(2):
        rb = PointRenderer(device=0)
        assert rb.Init(pc, False)
        rb.set_band(2, g)
        part = rb.Render(cam, proj, vp, nf)
        rows = np.arange(H) // bin_px() % 2 == g
        acc[rows] = part[rows]
    np.testing.assert_array_equal(acc, img)
    # a context goes back to splats with a splat upload
    cloud = scenes.synth_cloud(500, 7)
    assert SplatRenderer.Init(r, cloud, False, False)
    r.Sort(cam, proj, vp, nf)
    check_image(SplatRenderer.Render(r, cam, proj, vp, nf), orc.render_frame(cloud.as_array(), True, cam, proj, vp, nf)["image"])


def test_8f4_golden_fixtures(golden_dir):
    """the HIP path against the committed 8f-4 fixtures (depth-tested second eye; point sprites)"""
    import os
    from splatapult_amd import PointRenderer
    g = np.load(os.path.join(golden_dir, "fixtures_8f4.npz"))
    s = np.load(os.path.join(golden_dir, "synth_sh3.npz"))
    cloud = scenes.cloud_from_attrs({k[3:]: s[k] for k in s.files if k.startswith("in_")})
    W, H = int(s["W"]), int(s["H"])
    vp = [0, 0, W, H]
    r = make_renderer(cloud)
    r.Sort(s["cam"], s["proj"], vp, scenes.NF)
    check_image(r.Render(g["eye1"], s["proj"], vp, scenes.NF), g["exp_eye1_plain"])
    for bits in (24, 32):
        r.set_depth_test(bits)
        check_image(r.Render(g["eye1"], s["proj"], vp, scenes.NF), g["exp_eye1_depth%d" % bits])
    PW, PH = int(g["PW"]), int(g["PH"])
    for srgb in (0, 1):
        for bits in (0, 24):
            pr = PointRenderer(device=0)
            assert pr.Init(g["points"], bool(srgb), sprite=g["sprite"])
            pr.set_depth_test(bits)
            img = pr.Render(g["pcam"], g["pproj"], [0, 0, PW, PH], scenes.NF)
            d = np.abs(img - g["exp_points_srgb%d_depth%d" % (srgb, bits)])[..., :3]
            assert (d <= 1e-5).mean() >= 0.9999 and d.max() <= 1e-3


def test_fp16_framebuffer():
    cloud = scenes.synth_cloud(8000, 71, log_scale_mean=-3.2)
    view = scenes.default_view(320, 240)
    cam, proj, vp, nf = view
    r = make_renderer(cloud, fb_format="fp16")
    r.Sort(cam, proj, vp, nf)
    img = r.Render(cam, proj, vp, nf)
    assert img.dtype == np.float16
    o = oracle_frame(cloud.as_array(), True, cam, proj, vp, nf)
    check_fp16_image(img, o["image"], o["budget"])


def test_row_bands_reassemble_bit_exact():
    """multi-GPU sharding: interleaved tile rows rendered by separate contexts == single-context image"""
    cloud = scenes.synth_cloud(12000, 81, log_scale_mean=-3.2)
    view = scenes.default_view(640, 360, yaw=0.3)
    cam, proj, vp, nf = view
    r = make_renderer(cloud)
    r.Sort(cam, proj, vp, nf)
    full = r.Render(cam, proj, vp, nf)
    for G in (2, 3, 8):
        acc = np.zeros_like(full)
        for g in range(G):
            rb = make_renderer(cloud)
            rb.set_band(G, g)
            rb.Sort(cam, proj, vp, nf)
            part = rb.Render(cam, proj, vp, nf)
            rows = np.arange(360) // bin_px() % G == g
            assert (part[~rows] == 0).all()
            acc[rows] = part[rows]
        np.testing.assert_array_equal(acc, full)
    # band-restricted cull (Sort drops splats that cannot reach an owned row): same pixels, smaller V
    G = 4
    acc = np.zeros_like(full)
    vs = []
    for g in range(G):
        rb = make_renderer(cloud)
        rb.set_band(G, g, band_cull=True)
        rb.Sort(cam, proj, vp, nf)
        vs.append(rb.sort_count())
        part = rb.Render(cam, proj, vp, nf)
        rows = np.arange(360) // bin_px() % G == g
        acc[rows] = part[rows]
    np.testing.assert_array_equal(acc, full)
    assert max(vs) < 0.8 * r.sort_count(), (vs, r.sort_count())
    # the general layouts (msplat_set_band_layout / msplat_band_plan): contiguous bands, blocks of rows dealt round-robin;
    # 360 px = 11.25 bin rows, so bands are uneven and the last row is ragged; with and without the band cull
    from splatapult_amd import _capi
    R = (360 + bin_px() - 1) // bin_px()
    for kind, k, G, cull in (("contiguous", 1, 3, False), ("contiguous", 1, 8, True), ("block", 2, 3, True),
                             ("block", 4, 2, False), ("interleaved", 1, 5, True), ("contiguous", 1, 16, True)):
        acc = np.zeros_like(full)
        covered = np.zeros(360, bool)
        for g in range(G):
            rb = make_renderer(cloud)
            lay = rb.set_band_plan(kind, R, G, g, block_rows=k, band_cull=cull)
            mine = _capi.band_rows(*lay, rows_full=R)
            rb.Sort(cam, proj, vp, nf)
            part = rb.Render(cam, proj, vp, nf)
            rows = np.isin(np.arange(360) // bin_px(), mine)
            assert (part[~rows] == 0).all() and not (covered & rows).any()
            covered |= rows
            acc[rows] = part[rows]
            if kind == "contiguous" and cull and 0 < len(mine) <= 2:
                assert rb.sort_count() < 0.7 * r.sort_count()
        assert covered.all()
        np.testing.assert_array_equal(acc, full)
    # back to the whole image on the same context
    rb.set_band(1, 0)
    rb.Sort(cam, proj, vp, nf)
    np.testing.assert_array_equal(rb.Render(cam, proj, vp, nf), full)


def test_frames_in_flight_bit_identical_to_serial_frames():
    """frames overlapped on several contexts sharing one cloud (msplat_attach_cloud) == the same frames
    rendered one after the other on a single context; stereo re-uses the slot of its Sort"""
    import torch
    cloud = scenes.synth_cloud(60000, 91, log_scale_mean=-3.4)
    W, H = 640, 360
    Hpad = (H + bin_px() - 1) // bin_px() * bin_px()
    views = [scenes.default_view(W, H, yaw=0.1 * k, x=0.05 * k) for k in range(7)]
    r1 = make_renderer(cloud)
    serial = []
    for cam, proj, vp, nf in views:
        r1.Sort(cam, proj, vp, nf)
        serial.append((r1.Render(cam, proj, vp, nf), r1.sort_count(), r1.sorted_indices()))
    P = 3
    rp = make_renderer(cloud, frames_in_flight=P)
    assert rp.frames_in_flight == P
    dev = torch.device("cuda", 0)
    fbs = [torch.zeros((Hpad, W, 4), dtype=torch.float32, device=dev) for _ in views]
    fb2 = [torch.zeros((Hpad, W, 4), dtype=torch.float32, device=dev) for _ in views]
    torch.cuda.synchronize()
    slots = []
    for k, (cam, proj, vp, nf) in enumerate(views):       # nothing synchronises inside this loop
        rp.Sort(cam, proj, vp, nf)
        slots.append(rp.frame_slot)
        rp.Render(cam, proj, vp, nf, out_ptr=fbs[k].data_ptr(), pitch_bytes=W * 16)
        cam2 = camera.translate_local(cam, dx=0.03)        # second eye, same sort
        rp.Render(cam2, proj, vp, nf, out_ptr=fb2[k].data_ptr(), pitch_bytes=W * 16)
    assert slots == [k % P for k in range(len(views))]
    rp.wait_on_stream(None)                                # device-side join of the last frame ...
    rp.synchronize()                                       # ... and a host-side wait for all of them
    for k in range(len(views)):
        np.testing.assert_array_equal(fbs[k][:H].cpu().numpy(), serial[k][0])
    # getters refer to the latest Sort's context
    assert rp.sort_count() == serial[-1][1]
    np.testing.assert_array_equal(rp.sorted_indices(), serial[-1][2])
    cam, proj, vp, nf = views[-1]
    r1.Sort(cam, proj, vp, nf)
    eye2 = r1.Render(camera.translate_local(cam, dx=0.03), proj, vp, nf)
    np.testing.assert_array_equal(fb2[-1][:H].cpu().numpy(), eye2)
    # re-upload into the owner detaches it (new store); the attached contexts keep rendering the old cloud
    small = scenes.synth_cloud(500, 92)
    assert rp.Init(small, False, False)
    rp.Sort(cam, proj, vp, nf)
    img = rp.Render(cam, proj, vp, nf)
    ref = oracle_frame(small.as_array(), True, cam, proj, vp, nf)
    check_image(img, ref["image"], budget=ref["budget"])


def test_frames_in_flight_edge_cases():
    """attach without a cloud fails; empty and fully culled clouds render black on every context"""
    import ctypes as C
    from splatapult_amd import _capi
    L = _capi.lib()
    a, b = C.c_void_p(), C.c_void_p()
    assert L.msplat_create(C.byref(a), None) == _capi.OK and L.msplat_create(C.byref(b), None) == _capi.OK
    assert L.msplat_attach_cloud(b, a) == _capi.ERR_NO_CLOUD
    assert L.msplat_attach_cloud(None, a) == _capi.ERR_INVALID_ARG
    assert L.msplat_stream_wait(None, None) == _capi.ERR_INVALID_ARG and L.msplat_wait_event(a, None) == _capi.ERR_INVALID_ARG
    assert L.msplat_stream_wait(a, None) == _capi.OK            # default stream waits for the (idle) context stream
    L.msplat_destroy(a); L.msplat_destroy(b)
    cam, proj, vp, nf = scenes.default_view(160, 96)
    for cloud, behind in ((np.zeros((0, 61), np.float32), False), (scenes.synth_cloud(300, 5), True)):
        r = make_renderer(cloud, frames_in_flight=3)
        c = camera.pose((0.0, 0.0, -50.0)) if behind else cam          # looking away: everything culled
        for _ in range(4):
            r.Sort(c, proj, vp, nf)
            img = r.Render(c, proj, vp, nf)
            assert r.sort_count() == 0
            assert (img[..., :3] == 0).all() and (img[..., 3] == 1).all()


def test_large_viewport_4096_matches_oracle():
    """BASELINE config 4's framebuffer size (4096x4096: 128x128 bins, 65536 compositor work items) on a small cloud"""
    cloud = scenes.synth_cloud(30000, 77, log_scale_mean=-3.6)
    W = H = 4096
    cam, proj, vp, nf = scenes.default_view(W, H, z=6.0, yaw=0.15)
    r = make_renderer(cloud)
    r.Sort(cam, proj, vp, nf)
    img = r.Render(cam, proj, vp, nf)
    ref = oracle_frame(cloud.as_array(), True, cam, proj, vp, nf)
    assert r.sort_count() == ref["V"]
    check_image(img, ref["image"], budget=ref["budget"])
    st = r.stats()
    assert st["tiles_x"] == 128 and st["tiles_y"] == 128


def test_ballot_rank_fallback_matches_lds_atomic_rank(monkeypatch):
    """the stable ranking uses the return values of lane-ordered LDS atomics (probed at msplat_create); the ballot
    path (msplat_config.rank_mode = MSPLAT_RANK_BALLOT) is the fallback.  Both must give the oracle's permutation, the same bin lists and pixels,
    and pass the on-device order check (msplat_debug_verify_order)."""
    cloud = scenes.cloud_from_attrs(scenes.hard_attrs(9000, 23))             # duplicates: equal keys, tie order matters
    cam, proj, vp, nf = scenes.default_view(640, 400, yaw=0.4)
    mvp = orc.mat4_mul(proj, orc.mat4_inverse(cam))
    keys, idx = orc.sort(*orc.presort(cloud.as_array(), mvp, nf[1]))
    outs = []
    # the ballot path is selected by msplat_config.rank_mode (r6: the process-wide MSPLAT_BALLOT_RANK switch is gone)
    for kw, env in (({}, {}), ({"rank_mode": 1}, {}), ({"rank_mode": 1}, {"MSPLAT_SCAN_KERNELS": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = make_renderer(cloud, **kw)
        for k in env:
            monkeypatch.delenv(k)
        for rep in range(2):
            r.Sort(cam, proj, vp, nf)
            img = r.Render(cam, proj, vp, nf)
        np.testing.assert_array_equal(r.sorted_keys(), keys)
        np.testing.assert_array_equal(r.sorted_indices(), idx)
        assert r.verify_order() == (0, 0)
        ts, pairs = r.debug_tile_lists()
        outs.append((ts, pairs, img))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            np.testing.assert_array_equal(a, b)


def test_scan_free_and_scan_kernel_passes_agree(monkeypatch):
    """the radix / binning passes exist in two forms -- scan-free (group tables, 2 launches per pass; the default for
    chunk tables of up to 2048 rows) and upsweep + scan + downsweep (MSPLAT_SCAN_KERNELS=1, and automatically for
    large tables): same keys, permutation, bin lists and pixels, frame after frame (the row pass only turns
    scan-free from the second frame on: it sizes itself with an earlier frame's pair count)"""
    cloud = scenes.synth_cloud(70000, 201, log_scale_mean=-3.4)
    cam, proj, vp, nf = scenes.default_view(800, 450, yaw=0.2)

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = make_renderer(cloud)              # the switches are read at msplat_create
        for k in env:
            monkeypatch.delenv(k)
        outs = []
        for rep in range(3):
            c = camera.translate_local(cam, dx=0.01 * rep)
            r.Sort(c, proj, vp, nf)
            img = r.Render(c, proj, vp, nf)
            ts, pairs = r.debug_tile_lists()
            outs.append((r.sorted_keys(), r.sorted_indices(), ts, pairs, img))
        return outs
    a, b = run({}), run({"MSPLAT_SCAN_KERNELS": "1"})
    for fa, fb in zip(a, b):
        for xa, xb in zip(fa, fb):
            np.testing.assert_array_equal(xa, xb)
    mvp = orc.mat4_mul(proj, orc.mat4_inverse(cam))
    keys, idx = orc.sort(*orc.presort(cloud.as_array(), mvp, nf[1]))
    np.testing.assert_array_equal(a[0][0], keys)
    np.testing.assert_array_equal(a[0][1], idx)


def test_compositor_both_targets_probe_and_work_counters():
    """the compositor (one wave per 16x16 tile; the other formulations measured in r2 / r3 were removed in r4) on ragged right /
    top tiles, both target formats, with and without the per-item probe"""
    cloud = scenes.cloud_from_attrs(scenes.hard_attrs(6000, 17))
    cam, proj, vp, nf = scenes.default_view(701, 397, yaw=0.3, z=5.0)          # ragged right / top tiles
    ref = oracle_frame(cloud.as_array(), True, cam, proj, vp, nf)
    for fmt in ("fp32", "fp16"):
        r = make_renderer(cloud, fb_format=fmt)
        r.Sort(cam, proj, vp, nf)
        img = r.Render(cam, proj, vp, nf)
        if fmt == "fp32":
            check_image(img, ref["image"], budget=ref["budget"])
            r.set_tile_probe(True)
            np.testing.assert_array_equal(r.Render(cam, proj, vp, nf), img)      # the probe does not change pixels
            wk = r.composite_work()
            assert wk["work_items"] > 0 and wk["records_composited"] > 0
            assert wk["records_fetched"] >= wk["records_composited"]
            assert wk["pair_words_fetched"] <= wk["list_entries"] and wk["pixel_evals"] > 0
        else:
            check_fp16_image(img, ref["image"], ref["budget"])


@pytest.mark.parametrize("case", range(24))
def test_seeded_random_scenes_sort_exact_and_image_in_tolerance(case):
    """seeded sweep over cloud sizes around the chunk / group boundaries of the scan-free passes (2048-key chunks, 32-chunk
    groups, 1024-rank binning chunks), ragged viewports, cameras inside and outside the cloud, SH0 / SH3, fp32 / fp16:
    exact permutation, on-device order check, framebuffer inside the oracle tolerance, second frame identical"""
    rng = np.random.default_rng(1000 + case)
    sizes = [1, 2, 63, 64, 65, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4097, 65535, 65536, 65537, 66000, 70001]
    n = int(sizes[case % len(sizes)] if case < 18 else rng.integers(100, 90000))
    W, H = int(rng.integers(16, 900)), int(rng.integers(16, 700))
    full_sh = bool(case % 3)
    fb = "fp16" if case % 5 == 0 else "fp32"
    cloud = scenes.synth_cloud(n, 5000 + case, full_sh=full_sh, log_scale_mean=float(rng.uniform(-4.2, -2.6)),
                               pos_sigma=float(rng.uniform(0.8, 2.5)))
    z = float(rng.choice([0.5, 2.0, 5.0, 9.0]))
    cam, proj, vp, nf = scenes.default_view(W, H, z=z, yaw=float(rng.uniform(-3.1, 3.1)), pitch=float(rng.uniform(-0.6, 0.6)),
                                            x=float(rng.uniform(-1, 1)))
    r = make_renderer(cloud, fb_format=fb)
    r.Sort(cam, proj, vp, nf)
    img = r.Render(cam, proj, vp, nf)
    ref = oracle_frame(cloud.as_array(), full_sh, cam, proj, vp, nf)
    assert r.sort_count() == ref["V"]
    np.testing.assert_array_equal(r.sorted_indices(), ref["sorted_idx"])
    np.testing.assert_array_equal(r.sorted_keys(), ref["sorted_keys"])
    assert r.verify_order() == (0, 0)
    if fb == "fp16":
        check_fp16_image(img, ref["image"], ref["budget"])
    else:
        check_image(img, ref["image"], budget=ref["budget"])
    r.Sort(cam, proj, vp, nf)                              # second frame: the row pass is scan-free now
    np.testing.assert_array_equal(r.Render(cam, proj, vp, nf), img)


@pytest.mark.parametrize("fovy_deg,zn,zf,z,wh", [(20.0, 0.1, 1000.0, 12.0, (640, 360)),      # long lens
                                                  (100.0, 0.1, 1000.0, 3.0, (512, 512)),     # wide angle, inside the cloud's halo
                                                  (45.0, 0.5, 9.0, 7.0, (480, 300)),         # splats beyond the far plane
                                                  (60.0, 2.0, 40.0, 5.0, (333, 211)),        # near plane cuts the cloud
                                                  (45.0, 0.001, 1.0e6, 7.0, (400, 240))])    # key uses 12 of its 32 bits
def test_other_projections_match_the_oracle(fovy_deg, zn, zf, z, wh):
    """every other test uses the application's projection (45 degrees, near 0.1, far 1000: app.cpp:73-75); the depth key
    scales with 1 / far (presort_compute.glsl:53), the geometry stage rejects on ndc.z < 0.25 and clips at the far
    plane (splat_geom.glsl:46-54): other frusta move all three.  Keys at or beyond the far plane saturate."""
    W, H = wh
    cloud = scenes.synth_cloud(40000, 77, log_scale_mean=-3.3)
    cam = camera.pose((0.3, -0.2, z), 0.25, -0.1)
    proj = camera.perspective(np.radians(fovy_deg), W / H, zn, zf)
    vp, nf = [0, 0, W, H], [zn, zf]
    r = make_renderer(cloud)
    r.Sort(cam, proj, vp, nf)
    img = r.Render(cam, proj, vp, nf)
    ref = oracle_frame(cloud.as_array(), True, cam, proj, vp, nf)
    assert r.sort_count() == ref["V"] and ref["V"] > 1000
    np.testing.assert_array_equal(r.sorted_keys(), ref["sorted_keys"])
    np.testing.assert_array_equal(r.sorted_indices(), ref["sorted_idx"])
    assert r.verify_order() == (0, 0)
    _check_projection(r, ref, W, H)                      # centres, conics, colours, rectangles; no rejected splat drawn
    drawn_ref = int((ref["splats"]["reject"] == 0).sum())
    assert r.stats()["drawn"] <= drawn_ref
    if zf < 100.0:
        assert drawn_ref < ref["V"]                      # the case really has far- or near-rejected splats
    check_image(img, ref["image"], budget=ref["budget"])


def test_device_output_pair_overflow_is_reported_on_the_next_call():
    """VERDICT r1 / ADVICE: a device-output render cannot know that the (splat, bin) pair buffer overflowed; the
    binning kernel leaves the needed count in host-mapped memory and the next call on the context reports
    it once, after growing the buffer (unless msplat_config.pair_capacity fixed it): msplat_synchronize fails with
    MSPLAT_ERR_PAIR_OVERFLOW, msplat_sort / msplat_render -- whose own work is done -- return the warning code
    MSPLAT_ERR_PAIR_OVERFLOW_EARLIER"""
    import torch
    from splatapult_amd import MsplatError, _capi
    # screen-filling splats: ~10 M pairs at 1024x1024 (32 x 32 bins), above the 4 M-pair minimum capacity
    cloud = scenes.synth_cloud(12000, 123, log_scale_mean=-0.5, pos_sigma=1.0)
    W = H = 1024
    cam, proj, vp, nf = scenes.default_view(W, H, z=4.0)
    dev = torch.device("cuda", 0)
    fb = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    want = make_renderer(cloud)
    want.Sort(cam, proj, vp, nf)
    expect = want.Render(cam, proj, vp, nf)                 # host output: grows synchronously and retries
    need = want.stats()["pairs"]
    assert need > (1 << 22), need

    # (a) automatic capacity, error surfaces at msplat_synchronize
    r = make_renderer(cloud)
    cap0 = 1 << 22
    r.Sort(cam, proj, vp, nf)
    r.Render(cam, proj, vp, nf, out_ptr=fb.data_ptr(), pitch_bytes=W * 16)      # OK: nothing is known yet
    with pytest.raises(MsplatError) as e:
        r.synchronize()
    assert e.value.code == _capi.ERR_PAIR_OVERFLOW and "capacity grown" in str(e.value)
    assert not np.array_equal(fb.cpu().numpy(), expect)     # that frame really was truncated
    r.synchronize()                                         # reported once
    r.Sort(cam, proj, vp, nf)
    r.Render(cam, proj, vp, nf, out_ptr=fb.data_ptr(), pitch_bytes=W * 16)
    r.synchronize()
    assert r.stats()["pair_capacity"] >= need > cap0
    np.testing.assert_array_equal(fb.cpu().numpy(), expect)

    # (b) ... or at the next Sort, which is still performed
    r = make_renderer(cloud)
    r.Sort(cam, proj, vp, nf)
    r.Render(cam, proj, vp, nf, out_ptr=fb.data_ptr(), pitch_bytes=W * 16)
    torch.cuda.synchronize()
    with pytest.warns(_capi.EarlierFrameOverflow, match="earlier device-output render"):
        r.Sort(cam, proj, vp, nf)          # MSPLAT_ERR_PAIR_OVERFLOW_EARLIER: a warning, the sort itself is valid
    r.Render(cam, proj, vp, nf, out_ptr=fb.data_ptr(), pitch_bytes=W * 16)      # uses the sort that just ran
    r.synchronize()
    np.testing.assert_array_equal(fb.cpu().numpy(), expect)

    # (c) capacity fixed by the caller: reported, cannot grow, reported again for the next overflowing frame
    r = make_renderer(cloud, pair_capacity=100000)
    for _ in range(2):
        r.Sort(cam, proj, vp, nf)
        r.Render(cam, proj, vp, nf, out_ptr=fb.data_ptr(), pitch_bytes=W * 16)
        with pytest.raises(MsplatError) as e:
            r.synchronize()
        assert e.value.code == _capi.ERR_PAIR_OVERFLOW and "fixed by msplat_config.pair_capacity" in str(e.value)
    with pytest.raises(MsplatError):                        # host output with a fixed capacity fails at once
        r.Render(cam, proj, vp, nf)

    # (d) async_submit (the default with frames in flight): the worker thread's Sort finds the earlier frame's
    # overflow, grows the buffer and does its work; the warning is kept and handed out by the next synchronize / stream wait, once
    r = make_renderer(cloud, frames_in_flight=2, async_submit=True)
    for k in range(2):                                      # both contexts: frame, then a wait so that the count has landed
        r.Sort(cam, proj, vp, nf)
        r.Render(cam, proj, vp, nf, out_ptr=fb.data_ptr(), pitch_bytes=W * 16)
    with pytest.raises(MsplatError) as e:                   # synchronize itself sees the overflow of context 0's frame
        r.synchronize()
    assert e.value.code == _capi.ERR_PAIR_OVERFLOW
    r2 = make_renderer(cloud, frames_in_flight=1, async_submit=True)
    r2.Sort(cam, proj, vp, nf)
    r2.Render(cam, proj, vp, nf, out_ptr=fb.data_ptr(), pitch_bytes=W * 16)
    r2.wait_on_stream(None)
    torch.cuda.synchronize()                                # the truncated frame is done, nobody has looked at the flag yet
    r2.Sort(cam, proj, vp, nf)                              # queued: returns OK; the worker gets MSPLAT_ERR_PAIR_OVERFLOW_EARLIER
    r2.Render(cam, proj, vp, nf, out_ptr=fb.data_ptr(), pitch_bytes=W * 16)
    with pytest.warns(_capi.EarlierFrameOverflow, match="earlier device-output render"):
        r2.synchronize()
    np.testing.assert_array_equal(fb.cpu().numpy(), expect)  # the frame after the growth is complete
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        r2.synchronize()                                    # reported once


def test_render_before_sort_and_bad_viewport_errors():
    from splatapult_amd import MsplatError
    cloud = scenes.synth_cloud(10, 1)
    cam, proj, vp, nf = scenes.default_view(64, 64)
    r = make_renderer(cloud)
    with pytest.raises(MsplatError):
        r.Render(cam, proj, vp, nf)
    with pytest.raises(MsplatError):
        r.Sort(cam, proj, [0, 0, 9000, 100], nf)


# ------------------------------------------------------------------------------------------------
# full size (BASELINE config 2): exact sort parity + size-independent properties
# ------------------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def cloud_1m():
    """BASELINE configs[1] / configs[4] cloud (bench.py WORKLOADS cfg2 / cfg5)"""
    return scenes.synth_cloud(1_000_000, 0x5EED1234)


@pytest.fixture(scope="module")
def cloud_6m():
    """BASELINE configs[2] fallback / configs[3] cloud (bench.py WORKLOADS cfg3 / cfg4): 6 M splats, SH3"""
    return scenes.synth_cloud(6_000_000, 0x5EED6000, pos_sigma=3.0)


def _check_sort_exact(r, aos, cam, proj, nf):
    aos_s, order = storage_aos(r, aos)
    mvp = orc.mat4_mul(proj, orc.mat4_inverse(cam))
    keys, idx = orc.sort(*orc.presort(aos_s, mvp, nf[1]))
    if order is not None:
        idx = order[idx]
    gk, gi = r.sorted_keys(), r.sorted_indices()
    np.testing.assert_array_equal(gk, keys)
    np.testing.assert_array_equal(gi, idx)                        # stable: ties in ascending storage slot (= upload index unless reordered)
    assert (np.diff(gk.astype(np.int64)) >= 0).all()              # sortedness
    assert np.unique(gi).shape[0] == gi.shape[0]                  # a permutation of the visible set
    return keys.shape[0]


def _check_tile_lists_ascending(r):
    """every bin list is strictly increasing in rank (draw order preserved inside bins), lists tile the pair array"""
    assert r.verify_order() == (0, 0)                 # the on-device self-check agrees
    st = r.stats()
    ts, pairs = r.debug_tile_lists()
    assert (np.diff(ts.astype(np.int64)) >= 0).all() and ts[-1] == st["pairs"]
    ranks = (pairs & 0xFFFFFF).astype(np.int64)
    d = np.diff(ranks)
    starts = ts[1:-1][(ts[1:-1] > 0) & (ts[1:-1] < ranks.shape[0])]
    d[starts - 1] = 1
    assert (d > 0).all()
    return st


def _check_window(r, img, aos, W, H, cam, proj, nf, y0, y1, render_cam=None, render_proj=None, fp16=False):
    """bounded oracle comparison at full workload: rows [y0, y1) of the frame (the oracle follows r's storage order)"""
    import os
    aos, _ = storage_aos(r, aos)
    nt = max(32, min(128, os.cpu_count() or 32)) if y1 - y0 > 512 else 32         # whole frames: more row bands
    ref = orc.render_frame(aos, True, cam, proj, [0, 0, W, H], nf, render_cam=render_cam, render_proj=render_proj,
                           nthreads=32, want_image=False, want_splats=True)
    win, bud = orc.composite_flip(ref["splats"], W, H, nthreads=nt, row0=y0, row1=y1)
    if fp16:
        check_fp16_image(img[y0:y1], win[y0:y1], bud[y0:y1])
    else:
        check_image(img[y0:y1], win[y0:y1], budget=bud[y0:y1])
    return ref


def _check_whole_frame(r, img, aos, W, H, cam, proj, nf, render_cam=None, render_proj=None, fp16=False):
    """EVERY pixel of a full-size frame (the oracle windows leave > 75 % of configs 3 / 4 / 5 to properties).
    Checker: oracle/msplat_cpu_tiled.c -- the tiled front-to-back CPU renderer that tests/test_oracle.py holds against the literal
    oracle -- fed the cloud in the renderer's storage order, SURVEY 8c's tolerance.  A value beyond 5e-3 (2e-3 + 1 ulp for fp16)
    must be explained by the LITERAL oracle's threshold-flip budget for its row (orc_composite_flip over just those rows)."""
    import os
    import time
    aos_s, _ = storage_aos(r, aos)
    t0 = time.time()
    ref = orc.render_frame_tiled(aos_s, True, cam, proj, [0, 0, W, H], nf, render_cam=render_cam, render_proj=render_proj,
                                 nthreads=max(8, min(64, os.cpu_count() or 16)))
    want = ref["image"]
    assert ref["V"] == r.sort_count()
    d = np.abs(img[..., :3].astype(np.float64) - want[..., :3].astype(np.float64))
    lim = (2e-3 + np.abs(want[..., :3]) * 2.0 ** -10) if fp16 else np.full(d.shape, TIGHT)
    tol = lim if fp16 else 1e-4
    within = float((d <= tol).mean())
    print("_check_whole_frame %dx%d: %.2f M values, max |diff| %.3g, mean %.3g, %.5f within tolerance, checker %.1f s"
          % (W, H, d.size / 1e6, d.max(), d.mean(), within, time.time() - t0))
    assert within >= 0.999 and d.mean() <= 1e-4, (within, d.mean())
    assert (img[..., 3] == 1).all() and np.isfinite(img.astype(np.float32)).all()
    over = (d > lim).any(axis=-1)
    if over.any():
        rows = np.unique(np.nonzero(over)[0])
        assert len(rows) <= 64, "%d rows hold values beyond the bound (max %.3g)" % (len(rows), d.max())
        lit = orc.render_frame(aos_s, True, cam, proj, [0, 0, W, H], nf, render_cam=render_cam, render_proj=render_proj, nthreads=32,
                               want_image=False, want_splats=True)
        for y in rows:
            win, bud = orc.composite_flip(lit["splats"], W, H, nthreads=8, row0=int(y), row1=int(y) + 1)
            dl = np.abs(img[y, :, :3].astype(np.float64) - win[y, :, :3])
            ll = (2e-3 + np.abs(win[y, :, :3]) * 2.0 ** -10) if fp16 else TIGHT
            assert (dl <= ll + bud[y][:, None]).all(), "row %d: max excess %.3g over the literal oracle's flip budget" % (y, (dl - ll - bud[y][:, None]).max())
        print("_check_whole_frame: %d pixel(s) in %d row(s) beyond the bound, all inside the literal oracle's threshold-flip budget" % (over.sum(), len(rows)))


def test_full_size_config2_sort_and_properties(cloud_1m):
    n, W, H = 1_000_000, 1920, 1080
    cloud = cloud_1m
    cam, proj, vp, nf = scenes.default_view(W, H, z=7.0)
    r = make_renderer(cloud)
    r.Sort(cam, proj, vp, nf)
    aos = cloud.as_array()
    V = _check_sort_exact(r, aos, cam, proj, nf)
    img = r.Render(cam, proj, vp, nf)
    st = _check_tile_lists_ascending(r)
    assert st["sort_count"] == V and st["pairs"] > st["drawn"] > 0
    assert np.isfinite(img).all() and (img[..., 3] == 1).all()
    # idempotence: rendering again from the same sort is bit-identical
    np.testing.assert_array_equal(r.Render(cam, proj, vp, nf), img)
    _check_window(r, img, aos, W, H, cam, proj, nf, 0, H)          # the WHOLE frame, incl. the ragged top bin row (33.75 bins)


@pytest.mark.parametrize("step", [17, 40])
def test_full_size_config2_whole_frame_at_rotated_orbit_poses(cloud_1m, step):
    """BASELINE configs[1] at bench.py's orbit steps 17 and 40 (rotated view matrices): exact sort, ordered bin lists and
    the whole 1920x1080 frame against the oracle"""
    import math
    W, H = 1920, 1080
    cam = camera.orbit(7.0, 2.0 * math.pi * step / 64.0)
    proj, vp, nf = camera.perspective(camera.FOVY, W / H), [0, 0, W, H], scenes.NF
    r = make_renderer(cloud_1m)
    r.Sort(cam, proj, vp, nf)
    aos = cloud_1m.as_array()
    V = _check_sort_exact(r, aos, cam, proj, nf)
    img = r.Render(cam, proj, vp, nf)
    st = _check_tile_lists_ascending(r)
    assert st["sort_count"] == V
    _check_window(r, img, aos, W, H, cam, proj, nf, 0, H)


def _check_two_pass_at_full_size(init, cam, proj, vp, nf, single, window=None):
    """the two-pass frames behind the 6 M-splat bench numbers, at BASELINE size, against the single-pass frame
    (itself checked against the oracle windows) bit for bit: forced with a pinned share, forced with the share left to the feedback
    loop after 12 warm frames, and AUTO (what bench.py runs) after 16.  `init(**kw)` makes an initialised renderer;
    `window(img)` repeats an oracle-window check on the two-pass image.  One Render = one image, however it is scheduled
    (src/splatrenderer.cpp:315-343)."""
    import torch
    from splatapult_amd import _capi
    W, H = int(vp[2]), int(vp[3])
    Hpad = (H + bin_px() - 1) // bin_px() * bin_px()
    scratch = torch.zeros((Hpad, W, 4), dtype=torch.float32, device=torch.device("cuda", 0))
    for mode, share, warm in ((_capi.TWO_PASS_ON, 0.15, 0), (_capi.TWO_PASS_ON, 0.0, 12), (_capi.TWO_PASS_AUTO, 0.0, 16)):
        b = init(two_pass=mode)
        if share > 0.0:
            b.two_pass_state(share)
        for k in range(warm):
            b.Sort(cam, proj, vp, nf)
            b.Render(cam, proj, vp, nf, out_ptr=scratch.data_ptr(), pitch_bytes=W * 16)
            b.synchronize()                     # the frame's feedback has landed before the next plan is made
        b.Sort(cam, proj, vp, nf)
        img = b.Render(cam, proj, vp, nf)
        info = b.two_pass_info()
        frames, now = b.two_pass_state(share)
        print("two-pass at full size: mode %d share %.3f -> %.3f, two-pass frames %d, latest %s" % (mode, share, now, frames, info))
        if mode == _capi.TWO_PASS_ON:
            assert info is not None and info["splats_pass1"] > 0 and frames == warm + 1
            assert info["splats_pass1"] + info["splats_pass2"] < info["visible"]       # work really was skipped
        else:
            assert frames > 0                   # AUTO engaged (probe frames at least): these are the frames bench.py times
        np.testing.assert_array_equal(img, single)
        if window is not None and share > 0.0:
            window(img)
        assert b.verify_order() == (0, 0)
        b.close()


def test_full_size_config3_6m_1080p(cloud_6m):
    """BASELINE configs[2] (synthetic stand-in for the 6 M-splat Inria scene), 1920x1080 fp32: exact keys and
    permutation, ordered bin lists, 256-row oracle window"""
    W, H = 1920, 1080
    cam, proj, vp, nf = scenes.default_view(W, H, z=12.0)
    r = make_renderer(cloud_6m)
    r.Sort(cam, proj, vp, nf)
    aos = cloud_6m.as_array()
    V = _check_sort_exact(r, aos, cam, proj, nf)
    assert V > 5_000_000
    img = r.Render(cam, proj, vp, nf)
    st = _check_tile_lists_ascending(r)
    assert st["sort_count"] == V and st["pairs"] > st["drawn"] > 4_000_000
    assert np.isfinite(img).all() and (img[..., 3] == 1).all()
    _check_window(r, img, aos, W, H, cam, proj, nf, 412, 668)
    _check_window(r, img, aos, W, H, cam, proj, nf, 1056, 1080)      # the ragged top bin row (1080 = 33.75 bins)
    _check_whole_frame(r, img, aos, W, H, cam, proj, nf)              # r6: every pixel
    _check_two_pass_at_full_size(lambda **kw: make_renderer(cloud_6m, **kw), cam, proj, vp, nf, img,
                                 window=lambda im: _check_window(r, im, aos, W, H, cam, proj, nf, 412, 668))


def test_full_size_config4_6m_4096_and_8_bands(cloud_6m):
    """BASELINE configs[3]: the 6 M cloud at 4096x4096 fp32 -- single-context frame vs a 128-row oracle window,
    and the 8-GPU sharding (interleaved 32-px bin rows, band-restricted cull) reassembling bit-exactly"""
    W = H = 4096
    cam, proj, vp, nf = scenes.default_view(W, H, z=12.0)
    r = make_renderer(cloud_6m)
    r.Sort(cam, proj, vp, nf)
    full = r.Render(cam, proj, vp, nf)
    V = r.sort_count()
    st = _check_tile_lists_ascending(r)
    assert st["tiles_x"] == 128 and st["tiles_y"] == 128 and st["pairs"] > 20_000_000
    assert np.isfinite(full).all() and (full[..., 3] == 1).all()
    aos = cloud_6m.as_array()
    _check_window(r, full, aos, W, H, cam, proj, nf, 1984, 2112)
    _check_window(r, full, aos, W, H, cam, proj, nf, 4064, 4096)     # the top bin row
    _check_whole_frame(r, full, aos, W, H, cam, proj, nf)             # r6: every one of the 16.8 M pixels
    _check_two_pass_at_full_size(lambda **kw: make_renderer(cloud_6m, **kw), cam, proj, vp, nf, full,
                                 window=lambda im: _check_window(r, im, aos, W, H, cam, proj, nf, 1984, 2112))
    from splatapult_amd import _capi
    G = 8
    part = np.zeros_like(full)
    for kind, k, vmax in (("interleaved", 1, 0.6), ("contiguous", 1, 0.45), ("block", 4, 0.5)):
        acc = np.zeros_like(full)
        vs = []
        for g in range(G):
            lay = r.set_band_plan(kind, 128, G, g, block_rows=k, band_cull=True)
            r.Sort(cam, proj, vp, nf)
            vs.append(r.sort_count())
            r.Render(cam, proj, vp, nf, out=part)
            rows = np.isin(np.arange(H) // bin_px(), _capi.band_rows(*lay, rows_full=128))
            acc[rows] = part[rows]
        np.testing.assert_array_equal(acc, full)
        # the band cull really shrinks the per-rank sort (r2 measured 0.50 V at G = 8 for interleaved rows)
        assert max(vs) < vmax * V, (kind, vs, V)


def test_full_size_config5_stereo_fp16(cloud_1m):
    """BASELINE configs[4]: 1 M splats, two asymmetric views of 2016x2240, RGBA16F target, ONE sort with view 0
    (app.cpp:603-607); both eyes against a 256-row oracle window with the fp16 tolerance"""
    W, H = 2016, 2240
    cam0 = camera.pose((0.0, 0.0, 7.0))
    eyes = [camera.translate_local(cam0, dx=-0.032), camera.translate_local(cam0, dx=+0.032)]
    projs = [camera.create_projection(-1.0, 0.8, 0.95, -0.95), camera.create_projection(-0.8, 1.0, 0.95, -0.95)]
    vp, nf = [0, 0, W, H], scenes.NF
    r = make_renderer(cloud_1m, fb_format="fp16")
    r.Sort(eyes[0], projs[0], vp, nf)
    aos = cloud_1m.as_array()
    _check_sort_exact(r, aos, eyes[0], projs[0], nf)
    for e in range(2):
        img = r.Render(eyes[e], projs[e], vp, nf)
        assert img.dtype == np.float16 and img.shape == (H, W, 4)
        _check_tile_lists_ascending(r)
        _check_window(r, img, aos, W, H, eyes[0], projs[0], nf, 992, 1248, render_cam=eyes[e], render_proj=projs[e], fp16=True)
        _check_window(r, img, aos, W, H, eyes[0], projs[0], nf, 2208, 2240, render_cam=eyes[e], render_proj=projs[e], fp16=True)
        _check_whole_frame(r, img, aos, W, H, eyes[0], projs[0], nf, render_cam=eyes[e], render_proj=projs[e], fp16=True)     # r6: every pixel of both eyes


def test_large_cloud_uses_the_wide_scan_path():
    """above 2 M splats the histogram tables are scanned by the one-workgroup-per-digit kernel (radix_scan) instead
    of radix_scan_small: exact sort and tile lists + image parity on a 2.2 M-splat SH0 cloud"""
    n, W, H = 2_200_000, 480, 270
    cloud = scenes.synth_cloud(n, 909, full_sh=False, log_scale_mean=-5.2, pos_sigma=2.0)
    cam, proj, vp, nf = scenes.default_view(W, H, z=8.0, yaw=0.1)
    r = make_renderer(cloud)
    r.Sort(cam, proj, vp, nf)
    img = r.Render(cam, proj, vp, nf)
    ref = oracle_frame(cloud.as_array(), False, cam, proj, vp, nf, r=r)
    assert r.storage_order() is not None                 # 2.2 M splats: stored in Morton order (msplat_config.spatial_order = AUTO)
    assert r.sort_count() == ref["V"] and ref["V"] > 2_100_000
    np.testing.assert_array_equal(r.sorted_indices(), ref["sorted_idx"])
    np.testing.assert_array_equal(r.sorted_keys(), ref["sorted_keys"])
    ts, pairs = r.debug_tile_lists()
    for b in np.random.default_rng(0).integers(0, len(ts) - 1, size=40):
        ranks = pairs[ts[b]:ts[b + 1]] & 0xFFFFFF
        assert (np.diff(ranks.astype(np.int64)) > 0).all()
    check_image(img, ref["image"], budget=ref["budget"])
    # a view that culls most of the cloud: from the second such frame on, passes 1 and 2 of the sort take 4096-key chunks
    # (the V an earlier frame left in host-mapped memory is below 2 M) while pass 0 keeps 8192-key chunks: same exact order
    cam_in = camera.orbit(0.5, 1.3)
    mvp = orc.mat4_mul(proj, orc.mat4_inverse(cam_in))
    aos_s, order = storage_aos(r, cloud.as_array())
    keys_in, idx_in = orc.sort(*orc.presort(aos_s, mvp, nf[1]))
    idx_in = order[idx_in]
    assert 1000 < keys_in.shape[0] < 1_500_000
    for _ in range(3):
        r.Sort(cam_in, proj, vp, nf)
        r.Render(cam_in, proj, vp, nf)
        assert r.sort_count() == keys_in.shape[0]
        np.testing.assert_array_equal(r.sorted_keys(), keys_in)
        np.testing.assert_array_equal(r.sorted_indices(), idx_in)
    assert r.verify_order() == (0, 0)
    # a context configured for frames in flight takes the four 8-bit passes beyond 2 M splats (msplat.h, frame_mode): same frame
    from splatapult_amd import _capi
    r2 = make_renderer(cloud, frame_mode=_capi.FRAMES_IN_FLIGHT)
    r2.Sort(cam, proj, vp, nf)
    img2 = r2.Render(cam, proj, vp, nf)
    np.testing.assert_array_equal(r2.sorted_keys(), ref["sorted_keys"])
    np.testing.assert_array_equal(r2.sorted_indices(), ref["sorted_idx"])
    ts2, pairs2 = r2.debug_tile_lists()
    np.testing.assert_array_equal(ts2, ts)
    np.testing.assert_array_equal(pairs2, pairs)
    np.testing.assert_array_equal(img2, img)


def test_cpp_shim_renders_like_the_python_mirror(tmp_path, golden_dir):
    """C++ drop-in surface (msplat_host.hpp: GaussianCloud::ImportPly -> SplatRenderer::Init/Sort/Render)"""
    import os
    import subprocess
    from splatapult_amd import GaussianCloud, _capi
    from tests.conftest import ROOT
    exe = str(tmp_path / "example_render")
    libdir = os.path.dirname(_capi.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-I", ROOT, os.path.join(ROOT, "splatapult_amd", "host", "example_render.cpp"),
                    "-L", libdir, "-lmsplat", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    W, H = 320, 200
    out = str(tmp_path / "o.f32")
    ply = os.path.join(golden_dir, "test.ply")
    subprocess.run([exe, ply, out, str(W), str(H)], check=True)
    img = np.fromfile(out, np.float32).reshape(H, W, 4)
    subprocess.run([exe, ply, out + "3", str(W), str(H), "--frames-in-flight", "3"], check=True)
    np.testing.assert_array_equal(np.fromfile(out + "3", np.float32).reshape(H, W, 4), img)
    # SplatRenderer::ConfigureDevices: the same frame from a device group (three contexts on this box's one GPU)
    subprocess.run([exe, ply, out + "g", str(W), str(H), "--devices", "0,0,0"], check=True)
    np.testing.assert_array_equal(np.fromfile(out + "g", np.float32).reshape(H, W, 4), img)
    gc = GaussianCloud()
    assert gc.ImportPly(ply)
    cam = camera.pose((0.0, 0.0, 5.0))
    proj = camera.perspective(np.float32(45.0 * 3.14159265358979 / 180.0), W / H)
    ref = oracle_frame(gc.as_array(), True, cam, proj, [0, 0, W, H], scenes.NF)
    check_image(img, ref["image"], budget=ref["budget"])
    assert img[..., :3].max() > 0.5


def test_cpp_point_renderer_shim_matches_python(tmp_path):
    """C++ PointCloud / PointRenderer (msplat_host.hpp) == the Python mirror, with a PNG sprite read by ReadPNG"""
    import os
    import subprocess
    from PIL import Image
    from splatapult_amd import PointCloud, PointRenderer, _capi
    from tests.conftest import ROOT
    from tests.test_points import smooth_sprite
    exe = str(tmp_path / "example_points")
    libdir = os.path.dirname(_capi.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-I", ROOT, "-I", os.path.join(ROOT, "splatapult_amd", "host"),
                    os.path.join(ROOT, "splatapult_amd", "host", "example_points.cpp"),
                    "-L", libdir, "-lmsplat", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    W, H = 320, 240
    tex = smooth_sprite(48, 48, seed=9)
    png = str(tmp_path / "sprite.png")
    Image.fromarray(tex, "RGBA").save(png)
    pc = PointCloud(False)
    pc.InitDebugCloud()
    ply = str(tmp_path / "input.ply")
    assert pc.ExportPly(ply)
    out = str(tmp_path / "p.f32")
    subprocess.run([exe, out, str(W), str(H), ply, png], check=True)
    img = np.fromfile(out, np.float32).reshape(H, W, 4)
    back = PointCloud(False)
    assert back.ImportPly(ply)
    r = PointRenderer(device=0)
    assert r.Init(back, False, sprite=camera.read_image(png))
    cam = camera.pose((0.4, 0.4, 2.5))
    ref = r.Render(cam, camera.perspective(np.float32(45.0 * 3.14159265358979 / 180.0), W / H), [0, 0, W, H], scenes.NF)
    np.testing.assert_array_equal(img, ref)
    assert (img[..., :3].sum(axis=-1) > 0).sum() >= 15


# ------------------------------------------------------------------------------------------------
# GPU ingest: ImportPly's per-vertex math as a HIP kernel
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("n,full_sh", [(1, True), (63, True), (64, False), (5000, True), (5000, False)])
def test_gpu_ingest_matches_host_import(tmp_path, n, full_sh):
    """device cloud built by ingest_kernel == GaussianCloud::ImportPly (host) == oracle load-time math.
    Tolerance: alpha rel 1e-6, covariance 3e-6 of the splat's largest entry (device expf / sqrt vs glibc
    differ by <= 1 ulp); positions and SH coefficients are copies: exact."""
    from splatapult_amd import GaussianCloud, synthetic
    a = synthetic.generate(n, seed=1000 + n, full_sh=True)
    path = str(tmp_path / "in.ply")
    synthetic.write_ply(path, a)
    host = GaussianCloud(GaussianCloud.Options(full_sh, full_sh))
    assert host.ImportPly(path)
    r = SplatRenderer()
    assert r.InitFromPly(path, importFullSH=full_sh), r.last_error()
    dev = r.download_cloud(full_sh)
    ref = host.as_array()
    assert dev.shape == ref.shape
    np.testing.assert_array_equal(dev[:, :3], ref[:, :3])
    np.testing.assert_array_equal(dev[:, 4:16], ref[:, 4:16])
    if full_sh:
        np.testing.assert_array_equal(dev[:, 25:], ref[:, 25:])
    np.testing.assert_allclose(dev[:, 3], ref[:, 3], rtol=1e-6)
    # covariance: relative to the splat's own scale (off-diagonals are differences of near-equal products)
    scale = np.abs(ref[:, 16:25]).max(axis=1, keepdims=True)
    assert (np.abs(dev[:, 16:25] - ref[:, 16:25]) <= 3e-6 * scale).all()
    # and the ingested cloud renders like the host-built one
    cam, proj, vp, nf = scenes.default_view(256, 160)
    r.Sort(cam, proj, vp, nf)
    img = r.Render(cam, proj, vp, nf)
    refimg = orc.render_frame(ref, full_sh, cam, proj, vp, nf)["image"]
    check_image(img, refimg)


def test_gpu_ingest_covariance_by_rodrigues_rotation(tmp_path):
    """VERDICT r4 item 7c on the device: ingest_kernel's R S S^T R^T for random non-normalised quaternions against the property
    Sigma = sum_i s_i^2 (R e_i)(R e_i)^T with R e_i from Rodrigues' formula in float64 (tests/test_host.py) -- no shared formula"""
    from splatapult_amd import synthetic
    from tests.test_host import _covariance_by_rodrigues
    rng = np.random.default_rng(78)
    n = 5000
    a = synthetic.generate(n, seed=6, full_sh=True)
    a["rot"] = (rng.normal(size=(n, 4)) * rng.uniform(0.2, 5.0, size=(n, 1))).astype(np.float32)
    a["rot"][:4] = np.array([[0, 1, 0, 0], [1, 1, 0, 0], [1, 1, 1, 1], [-1, 2, -3, 4]], np.float32)
    a["log_scale"] = rng.uniform(-9.0, 0.5, size=(n, 3)).astype(np.float32)
    path = str(tmp_path / "q.ply")
    synthetic.write_ply(path, a)
    r = SplatRenderer()
    assert r.InitFromPly(path, importFullSH=True), r.last_error()
    dev = r.download_cloud(True)
    got = dev[:, 16:25].reshape(n, 3, 3).transpose(0, 2, 1).astype(np.float64)
    want = _covariance_by_rodrigues(a["rot"], a["log_scale"])
    scale = np.abs(want).max(axis=(1, 2), keepdims=True)
    err = (np.abs(got - want) / scale).max()
    print("ingest_kernel covariance vs Rodrigues construction: worst error %.3g of the splat's largest entry" % err)
    assert err < 6e-6


def test_gpu_ingest_test_ply_and_errors(golden_dir, tmp_path):
    import os
    r = SplatRenderer()
    assert r.InitFromPly(os.path.join(golden_dir, "test.ply"), importFullSH=False)
    g = np.load(os.path.join(golden_dir, "test_ply_cfg1.npz"))
    np.testing.assert_allclose(r.download_cloud(False), g["aos_nosh"], rtol=1e-6)
    assert r.stats()["num_splats"] == 16
    assert not SplatRenderer().InitFromPly(str(tmp_path / "missing.ply"))           # false after logging
    bad = tmp_path / "bad.ply"
    bad.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 0\nend_header\n")
    assert not SplatRenderer().InitFromPly(str(bad))


# ------------------------------------------------------------------------------------------------
# round 3: the three-pass sort chooses its digit widths from the visible set's largest quantised depth
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("zf,z,n", [(8.0, 7.0, 30000),        # depth up to and beyond far: keys saturate at 0 -> 10 + 11 + 11 bits
                                    (40.0, 7.0, 30000),       # q up to 2^30: 10 + 10 + 10
                                    (1.0e6, 7.0, 30000),      # tiny q (B < 26): the minimum widths 10 + 8 + 8
                                    (1000.0, 7.0, 5000),      # the reference's far plane
                                    (64.0, 30.0, 200000)])    # many chunks, B = 31
@pytest.mark.parametrize("frame_mode", ["serial", "in_flight"])    # 512-thread workgroups / the 256-thread form for frames in flight
def test_sort_exact_for_every_key_range(zf, z, n, frame_mode):
    from splatapult_amd import _capi
    fm = _capi.FRAMES_IN_FLIGHT if frame_mode == "in_flight" else _capi.FRAMES_SERIAL
    cloud = scenes.synth_cloud(n, 1234 + int(zf))
    cam, proj, vp, _ = scenes.default_view(640, 480, z=z, yaw=0.4)
    nf = [0.1, zf]
    proj = camera.perspective(camera.FOVY, 640 / 480, 0.1, zf)
    r = make_renderer(cloud, frame_mode=fm)
    for _ in range(2):                      # twice: the second frame runs on the tables the first one left behind
        r.Sort(cam, proj, vp, nf)
        mvp = orc.mat4_mul(proj, orc.mat4_inverse(cam))
        keys, idx = orc.sort(*orc.presort(cloud.as_array(), mvp, nf[1]))
        assert r.sort_count() == keys.shape[0] > 0
        np.testing.assert_array_equal(r.sorted_keys(), keys)
        np.testing.assert_array_equal(r.sorted_indices(), idx)
    assert r.verify_order()[0] == 0


def test_wide_sort_and_legacy_sort_and_tile_tables_agree(monkeypatch):
    """MSPLAT_SORT=lsd8 (four 8-bit passes: the fallback without lane-ordered LDS atomics) and the in-flight kernel selection
    (256-thread sort workgroups, tile_start_kernel / tile_order_kernel instead of the row pass's counts; msplat_config.frame_mode)
    against the default: bit-identical keys, permutation, bin lists and pixels"""
    from splatapult_amd import _capi
    cloud = scenes.synth_cloud(150000, 77, log_scale_mean=-3.6)
    cam, proj, vp, nf = scenes.default_view(800, 450, yaw=-0.3)
    res = []
    for env, kw in (({}, {}), ({"MSPLAT_SORT": "lsd8"}, {}), ({}, {"frame_mode": _capi.FRAMES_IN_FLIGHT}),
                    ({"MSPLAT_SORT": "lsd8"}, {"frame_mode": _capi.FRAMES_IN_FLIGHT})):
        monkeypatch.delenv("MSPLAT_SORT", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = make_renderer(cloud, **kw)
        r.Sort(cam, proj, vp, nf)
        img = r.Render(cam, proj, vp, nf)
        ts, pairs = r.debug_tile_lists()
        res.append((r.sorted_keys(), r.sorted_indices(), ts, pairs, img))
        assert r.verify_order() == (0, 0)
    for other in res[1:]:
        for a, b in zip(res[0], other):
            np.testing.assert_array_equal(a, b)


def test_frame_modes_give_identical_frames():
    """msplat_config.frame_mode only chooses kernels (msplat.h): a context configured for frames in flight returns the keys,
    permutation, bin lists and pixels of one configured for a single frame at a time"""
    from splatapult_amd import _capi
    cloud = scenes.synth_cloud(120000, 91, log_scale_mean=-3.5)
    cam, proj, vp, nf = scenes.default_view(800, 450, yaw=0.5)
    res = []
    for mode in (_capi.FRAMES_AUTO, _capi.FRAMES_SERIAL, _capi.FRAMES_IN_FLIGHT, 77):       # 77: stale padding = AUTO
        r = make_renderer(cloud, frame_mode=mode)
        for _ in range(2):
            r.Sort(cam, proj, vp, nf)
            img = r.Render(cam, proj, vp, nf)
        ts, pairs = r.debug_tile_lists()
        res.append((r.sorted_keys(), r.sorted_indices(), ts, pairs, img))
        assert r.verify_order() == (0, 0)
    for other in res[1:]:
        for a, b in zip(res[0], other):
            np.testing.assert_array_equal(a, b)


# ------------------------------------------------------------------------------------------------
# round 3: several GPUs, one process (msplat_group_*).  On a one-GPU box the group runs with one context, and with two
# contexts on the same device -- the band plan, the worker threads, the joins and both exchange forms execute; only
# the xGMI peer mapping itself needs a second GPU.
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("devices,layout,k,cull", [([0], "contiguous", 1, False), ([0, 0], "contiguous", 1, True),
                                                   ([0, 0, 0], "block", 2, True), ([0, 0], "interleaved", 1, False)])
def test_device_group_renders_the_single_context_frame(devices, layout, k, cull, monkeypatch):
    import torch
    from splatapult_amd import SplatRendererGroup
    cloud = scenes.synth_cloud(40000, 55, log_scale_mean=-3.3)
    W, H = 640, 360
    cam, proj, vp, nf = scenes.default_view(W, H, yaw=0.2)
    r = make_renderer(cloud)
    r.Sort(cam, proj, vp, nf)
    full = r.Render(cam, proj, vp, nf)
    for exchange in ("peer", "copy"):
        if exchange == "copy":
            monkeypatch.setenv("MSPLAT_GROUP_EXCHANGE", "copy")
        else:
            monkeypatch.delenv("MSPLAT_GROUP_EXCHANGE", raising=False)
        g = SplatRendererGroup(devices, layout=layout, block_rows=k, band_cull=cull)
        assert g.Init(cloud), g.last_error()
        assert g.size == len(devices) and g.peer_store(0)
        if len(devices) > 1:
            assert g.peer_store(1) == (exchange == "peer")
        # host framebuffer
        g.Sort(cam, proj, vp, nf)
        np.testing.assert_array_equal(g.Render(cam, proj, vp, nf), full)
        if cull and len(devices) > 1:
            assert max(g.sort_count(i) for i in range(g.size)) < r.sort_count()
        # device framebuffer on devices[0]: complete once context 0's stream is (msplat_group_synchronize)
        fb = torch.full((H, W, 4), -1.0, dtype=torch.float32, device="cuda:0")
        for _ in range(3):
            g.Sort(cam, proj, vp, nf)
            g.Render(cam, proj, vp, nf, out_ptr=fb.data_ptr(), pitch_bytes=W * 16)
        g.synchronize()
        np.testing.assert_array_equal(fb.cpu().numpy(), full)
        # another viewport: the rows are re-planned
        vp2 = [0, 0, 333, 211]
        proj2 = camera.perspective(camera.FOVY, 333 / 211)
        r.Sort(cam, proj2, vp2, nf)
        g.Sort(cam, proj2, vp2, nf)
        np.testing.assert_array_equal(g.Render(cam, proj2, vp2, nf), r.Render(cam, proj2, vp2, nf))
        r.Sort(cam, proj, vp, nf)
        g.close()


def _hip_device_count():
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        c = ctypes.c_int(0)
        return c.value if hip.hipGetDeviceCount(ctypes.byref(c)) != 0 else c.value
    except OSError:
        return 0


@pytest.mark.parametrize("exchange", ["peer", "copy", "rccl"])
def test_device_group_over_distinct_devices(exchange, monkeypatch):
    """msplat_group over REAL distinct GPUs (skipped on a one-GPU box): the peer mapping towards device 0 and the
    cross-device stream waits, the per-device kernel attributes (the three-pass sort's dynamic LDS on devices 1..), both exchange
    forms; and the ordering contract of msplat_group_render -- DISTINCT consecutive frames into ONE framebuffer with a consumer
    queued on context 0's stream between them: every consumer sees its own frame complete, no rank overwrites rows early"""
    import torch
    from splatapult_amd import SplatRendererGroup
    ndev = min(_hip_device_count(), torch.cuda.device_count())
    if ndev < 2:
        pytest.skip("needs two GPUs (found %d)" % ndev)
    if exchange != "peer":
        monkeypatch.setenv("MSPLAT_GROUP_EXCHANGE", exchange)            # rccl: ncclSend / ncclRecv over ncclCommInitAll's communicators
    devices = list(range(min(ndev, 4)))
    cloud = scenes.synth_cloud(300000, 56, log_scale_mean=-3.6)          # large enough for the spatial storage order (AUTO)
    W, H = 1280, 720
    proj, vp, nf = camera.perspective(camera.FOVY, W / H), [0, 0, W, H], scenes.NF
    cams = [camera.orbit(7.0, 0.3 * k) for k in range(6)]
    r = make_renderer(cloud)
    refs = []
    for cam in cams:
        r.Sort(cam, proj, vp, nf)
        refs.append(r.Render(cam, proj, vp, nf))
    g = SplatRendererGroup(devices, layout="block", block_rows=2, band_cull=True)
    assert g.Init(cloud), g.last_error()
    assert all(g.peer_store(i) == (exchange == "peer") for i in range(1, g.size))
    fb = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    keep = [torch.empty_like(fb) for _ in cams]
    import ctypes as C
    from splatapult_amd import _capi
    s0 = torch.cuda.ExternalStream(_capi.lib().msplat_get_stream(g.context(0)), device="cuda:0")
    for k, cam in enumerate(cams):                                       # nothing synchronises inside this loop
        g.Sort(cam, proj, vp, nf)
        g.Render(cam, proj, vp, nf, out_ptr=fb.data_ptr(), pitch_bytes=W * 16)
        with torch.cuda.stream(s0):                                      # the consumer of frame k, on context 0's stream
            keep[k].copy_(fb, non_blocking=True)
    g.synchronize()
    torch.cuda.synchronize()
    for k in range(len(cams)):
        np.testing.assert_array_equal(keep[k].cpu().numpy(), refs[k])
    assert g.exchange() == {"peer": "peer_store", "copy": "copy", "rccl": "rccl"}[exchange]
    g.close()


def test_band_exchange_through_rccl_on_one_rank_and_the_group_switch():
    """the row gather over RCCL behind the C ABI.  A one-GPU box cannot hold two ranks (RCCL refuses a device
    twice), so the real calls are exercised on a ONE-rank communicator: the runs of bin rows that rank g of 8 owns travel through
    ncclSend / ncclRecv to the rank itself, from one framebuffer into another (msplat_debug_band_exchange_loopback: the same run
    enumeration and the same group of calls as msplat_band_exchange) -- every layout, both pixel sizes; world == 1 is a no-op;
    the one-process group accepts the switch where it can (one device) and refuses it with a message where RCCL cannot (a device
    listed twice), staying on its previous exchange."""
    import torch
    from splatapult_amd import MsplatError, SplatRendererGroup, _capi
    from splatapult_amd.dist import RcclComm, owned_rows
    comm = RcclComm(0, 1, 0)
    r = make_renderer(scenes.synth_cloud(2000, 5))
    r16 = make_renderer(scenes.synth_cloud(2000, 5), fb_format="fp16")      # the pixel size is the context's target format (r6)
    T = bin_px()
    dev = torch.device("cuda", 0)
    for rr, W, H, dtype, bpp in ((r, 640, 360, torch.float32, 16), (r16, 517, 293, torch.float16, 8)):
        tiles_y = (H + T - 1) // T
        Hpad = tiles_y * T
        src = torch.randn((Hpad, W, 4), dtype=torch.float32, device=dev).to(dtype)
        for kind, name, k in ((_capi.BANDS_CONTIGUOUS, "contiguous", 1), (_capi.BANDS_INTERLEAVED, "interleaved", 1),
                              (_capi.BANDS_BLOCK_INTERLEAVED, "block", 2), (_capi.BANDS_ROOT_WEIGHTED, "weighted", 300)):
            for g in (0, 3, 7):
                dst = torch.zeros_like(src)
                rr.band_exchange(comm.handle, g, 8, 0, kind, k, dst.data_ptr(), W * bpp, W, Hpad, loopback_src=src.data_ptr())
                rr.synchronize()
                torch.cuda.synchronize()
                want = torch.zeros_like(src)
                rows = np.isin(np.arange(Hpad) // T, owned_rows(name, tiles_y, 8, g, k))
                want[torch.from_numpy(rows).to(dev)] = src[torch.from_numpy(rows).to(dev)]
                assert torch.equal(dst, want), (name, g, W, H)
    with pytest.raises(MsplatError):         # an fp16-sized pitch on an RGBA32F context: refused, not reinterpreted
        r.band_exchange(comm.handle, 0, 8, 0, _capi.BANDS_CONTIGUOUS, 1, dst.data_ptr(), 517 * 8, 517, Hpad, loopback_src=src.data_ptr())
    # a target that is a WINDOW of a wider surface (pitch > width x 16): the run travels row by row, the surface's other pixels
    # keep their values on both sides and nothing behind the last row's last pixel is touched
    Ws, Ww, Hh = 640, 500, 96
    surf_src = torch.randn((Hh, Ws, 4), dtype=torch.float32, device=dev)
    surf_dst = torch.full((Hh, Ws, 4), -7.0, dtype=torch.float32, device=dev)
    for g in (0, 1):
        surf_dst.fill_(-7.0)
        r.band_exchange(comm.handle, g, 2, 0, _capi.BANDS_ROOT_WEIGHTED, 200, surf_dst.data_ptr(), Ws * 16, Ww, Hh, loopback_src=surf_src.data_ptr())
        r.synchronize()
        torch.cuda.synchronize()
        rows = torch.from_numpy(np.isin(np.arange(Hh) // T, owned_rows("weighted", Hh // T, 2, g, 200))).to(dev)
        want = torch.full_like(surf_dst, -7.0)
        want[rows, :Ww] = surf_src[rows, :Ww]
        assert torch.equal(surf_dst, want), g
    # fp16 on the wire (MSPLAT_EXCHANGE_WIRE_FP16, fp32 targets): the rows arrive rounded once to fp16 -- |d| <= 2^-11 |value| --
    # packed and unpacked by the library around the same ncclSend / ncclRecv; an fp16-sized pitch is refused
    W, H = 640, 360
    tiles_y = (H + T - 1) // T
    Hpad = tiles_y * T
    src = torch.randn((Hpad, W, 4), dtype=torch.float32, device=dev) * 3.0
    for kind, name, k in ((_capi.BANDS_CONTIGUOUS, "contiguous", 1), (_capi.BANDS_BLOCK_INTERLEAVED, "block", 2)):
        for g in (0, 5):
            dst = torch.zeros_like(src)
            r.band_exchange(comm.handle, g, 8, 0, kind, k, dst.data_ptr(), W * 16, W, Hpad, loopback_src=src.data_ptr(), wire_fp16=True)
            r.synchronize()
            torch.cuda.synchronize()
            rows = torch.from_numpy(np.isin(np.arange(Hpad) // T, owned_rows(name, tiles_y, 8, g, k))).to(dev)
            want = torch.zeros_like(src)
            want[rows] = src[rows].half().float()
            assert torch.equal(dst, want), (name, g)
            assert float((dst[rows] - src[rows]).abs().max()) <= 2.0 ** -11 * float(src.abs().max())
    with pytest.raises(MsplatError):
        r.band_exchange(comm.handle, 0, 8, 0, _capi.BANDS_CONTIGUOUS, 1, dst.data_ptr(), W * 8, W, Hpad, loopback_src=src.data_ptr(), wire_fp16=True)
    # ordering behind a frame whose launches a worker thread issues (async_submit): the exchange waits for them to be ISSUED
    cloud_a = scenes.synth_cloud(30000, 91, log_scale_mean=-3.0)
    Wa, Ha = 640, 352
    cam, proj, vp, nf = scenes.default_view(Wa, Ha)
    ra = make_renderer(cloud_a, frames_in_flight=1, async_submit=True)
    ref_img = make_renderer(cloud_a)
    ref_img.Sort(cam, proj, vp, nf)
    want_img = ref_img.Render(cam, proj, vp, nf)
    fa = torch.zeros((Ha, Wa, 4), dtype=torch.float32, device=dev)
    fb2 = torch.zeros_like(fa)
    for _ in range(3):
        ra.Sort(cam, proj, vp, nf)
        fa.zero_(); fb2.zero_()
        torch.cuda.synchronize()
        ra.Render(cam, proj, vp, nf, out_ptr=fa.data_ptr(), pitch_bytes=Wa * 16)       # queued: returns at once
        ra.band_exchange(comm.handle, 0, 1, 0, _capi.BANDS_CONTIGUOUS, 1, fb2.data_ptr(), Wa * 16, Wa, Ha, loopback_src=fa.data_ptr())
        ra.synchronize()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(fb2.cpu().numpy(), want_img)       # the whole frame (rank 0 of 1 owns every row) arrived
    # world == 1: the whole image is this rank's, nothing to exchange (no communicator needed)
    r.band_exchange(None, 0, 1, 0, _capi.BANDS_CONTIGUOUS, 1, 0, 0, 0, 0)
    with pytest.raises(MsplatError):
        r.band_exchange(None, 0, 2, 0, _capi.BANDS_CONTIGUOUS, 1, src.data_ptr(), W * bpp, W, Hpad)      # no communicator
    comm.close()
    # the one-process group
    cloud = scenes.synth_cloud(20000, 77, log_scale_mean=-3.2)
    W, H = 640, 360
    cam, proj, vp, nf = scenes.default_view(W, H)
    one = make_renderer(cloud)
    one.Sort(cam, proj, vp, nf)
    ref = one.Render(cam, proj, vp, nf)
    g1 = SplatRendererGroup([0])
    assert g1.Init(cloud), g1.last_error()
    g1.set_exchange("rccl")                                    # one device: accepted, nothing to exchange
    g2 = SplatRendererGroup([0, 0], layout="interleaved")
    assert g2.Init(cloud), g2.last_error()
    with pytest.raises(MsplatError) as e:
        g2.set_exchange("rccl")                                # RCCL refuses a device listed twice
    assert e.value.code == _capi.ERR_UNSUPPORTED and "ncclCommInitAll" in str(e.value)
    fb = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    for g in (g1, g2):
        g.Sort(cam, proj, vp, nf)
        fb.zero_()
        g.Render(cam, proj, vp, nf, out_ptr=fb.data_ptr(), pitch_bytes=W * 16)
        g.synchronize()
        np.testing.assert_array_equal(fb.cpu().numpy(), ref)
        g.close()


def test_exchange_without_librccl_is_refused_not_a_crash():
    """with no loadable librccl the RCCL entry points return MSPLAT_ERR_UNSUPPORTED with the loader's message (the
    first version called dlerror() twice and crashed on the NULL the second call returns); a mistyped MSPLAT_GROUP_EXCHANGE is
    refused instead of silently selecting the default.  Child processes: the library resolves librccl once per process"""
    import subprocess
    import sys
    code = r"""
import sys
sys.path.insert(0, %r)
import torch
from splatapult_amd import SplatRenderer, SplatRendererGroup, MsplatError, _capi
import numpy as np
r = SplatRenderer(device=0)
assert r.Init(np.zeros((8, 61), np.float32), False, False)
fb = torch.zeros((64, 64, 4), dtype=torch.float32, device="cuda")
import ctypes as C
try:
    r.band_exchange(C.c_void_p(1234), 1, 2, 0, _capi.BANDS_CONTIGUOUS, 1, fb.data_ptr(), 64 * 16, 64, 64)
    print("NO ERROR")
except MsplatError as e:
    print("CODE", e.code, str(e))
g = SplatRendererGroup([0, 0])
assert g.Init(np.zeros((8, 61), np.float32))
try:
    g.set_exchange("rccl")
    print("NO ERROR")
except MsplatError as e:
    print("CODE", e.code, str(e))
""" % ROOT
    env = dict(os.environ, MSPLAT_RCCL_LIB="/nonexistent/librccl_not_here.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-800:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("CODE")]
    assert len(lines) == 2 and all(("CODE %d" % _capi_mod().ERR_UNSUPPORTED) in ln and "librccl not found" in ln for ln in lines), out.stdout
    code2 = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from splatapult_amd import SplatRendererGroup
g = SplatRendererGroup([0, 0])
ok = g.Init(np.zeros((8, 61), np.float32))
print("INIT", ok, g.last_error())
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code2], env=dict(os.environ, MSPLAT_GROUP_EXCHANGE="nccl"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-800:]
    assert "INIT False" in out.stdout and "MSPLAT_GROUP_EXCHANGE=nccl" in out.stdout, out.stdout


def _capi_mod():
    from splatapult_amd import _capi
    return _capi


def test_device_group_errors():
    from splatapult_amd import SplatRendererGroup, _capi
    g = SplatRendererGroup([0, 99])
    assert g.Init(np.zeros((4, 61), np.float32)) is False and "99" in g.last_error()
    g = SplatRendererGroup([0, 0])
    assert g.Init(np.zeros((0, 61), np.float32))           # an empty cloud is a cloud
    cam, proj, vp, nf = scenes.default_view(64, 48)
    with pytest.raises(_capi.MsplatError):
        g.Render(cam, proj, vp, nf)                         # no sort yet
    g.Sort(cam, proj, vp, nf)
    img = g.Render(cam, proj, vp, nf)
    assert (img[..., :3] == 0).all() and (img[..., 3] == 1).all()


# ------------------------------------------------------------------------------------------------
# round 3: a scene-LIKE 6 M workload (bench.py cfg3s): surfaces, heavy-tailed anisotropic scales, 1 % background splats a
# quarter of the view wide, bimodal opacity, cameras inside the cloud -- through the FILE path (PLY + cameras.json)
# ------------------------------------------------------------------------------------------------
def test_scene_like_6m_file_replay_matches_the_oracle(tmp_path):
    from splatapult_amd import synthetic
    from splatapult_amd.scene import GaussianCloud
    n, W, H = 6_000_000, 1920, 1080
    ply = str(tmp_path / "point_cloud" / "iteration_30000" / "point_cloud.ply")
    import os
    os.makedirs(os.path.dirname(ply))
    synthetic.write_ply(ply, synthetic.generate_scene(n, seed=0x5CE11E))
    synthetic.write_cameras_json(str(tmp_path / "cameras.json"), synthetic.scene_cameras(64), W, H, camera.FOVY)
    cj = camera.find_config_file(ply, "cameras.json")            # app.cpp:418-461: two directories above the PLY
    assert cj == str(tmp_path / "cameras.json")
    cams = [m for m, _ in camera.load_cameras_json(cj)]
    assert len(cams) == 64
    r = SplatRenderer(device=0)
    assert r.InitFromPly(ply, True, False), r.last_error()      # Ply::Parse + GaussianCloud::ImportPly's math on the GPU
    host = GaussianCloud()
    assert host.ImportPly(ply)                                   # the same file through the host importer, for the oracle
    aos = host.as_array()
    assert aos.shape == (n, 61)
    proj, vp, nf = camera.perspective(camera.FOVY, W / H), [0, 0, W, H], scenes.NF
    cap0 = r.stats()["pair_capacity"]
    for k, (y0, y1) in ((0, (500, 564)), (23, (1040, 1080))):   # a mid-frame window; the ragged top bin row at another pose
        cam = cams[k]
        r.Sort(cam, proj, vp, nf)
        V = _check_sort_exact(r, aos, cam, proj, nf)
        assert 1_000_000 < V < n                                 # the camera is inside: much of the cloud is behind it
        img = r.Render(cam, proj, vp, nf)
        st = _check_tile_lists_ascending(r)
        assert st["sort_count"] == V
        ts, _ = r.debug_tile_lists(want_pairs=False)
        longest = int(np.diff(ts.astype(np.int64)).max())
        print("scene-like 6M pose %d: V %d  drawn %d  pairs(32 px bins) %d = %.1f per splat  longest bin list %d  pair capacity %d -> %d"
              % (k, V, st["drawn"], st["pairs"], st["pairs"] / n, longest, cap0, st["pair_capacity"]))
        assert st["pairs"] > 5 * st["drawn"]                      # big footprints: many bins per splat
        assert np.isfinite(img).all() and (img[..., 3] == 1).all()
        _check_window(r, img, aos, W, H, cam, proj, nf, y0, y1)

        def init(**kw):
            b = SplatRenderer(device=0, **kw)
            assert b.InitFromPly(ply, True, False), b.last_error()
            return b
        if k == 0:
            _check_two_pass_at_full_size(init, cam, proj, vp, nf, img,
                                         window=lambda im: _check_window(r, im, aos, W, H, cam, proj, nf, y0, y1))


def test_heavy_chunks_of_the_column_pass_are_split_without_changing_anything():
    """depth order puts a scene's huge far splats first: a few chunks of the column pass hold most of the pairs and are given
    eight workgroups (one per block of columns).  The helper workgroups are sized from an EARLIER frame's count of heavy chunks,
    so a context's first frame runs every chunk unsplit and the later ones split them: same bin lists and pixels, and the
    oracle's image."""
    a = scenes.synthetic.generate(9000, seed=404, pos_sigma=1.2, log_scale_mean=-3.4, log_scale_sigma=0.8)
    a["xyz"][:3500, 2] -= 14.0                      # a far layer ...
    a["log_scale"][:3500] = 1.0 + 0.1 * a["log_scale"][:3500]             # ... of screen-sized splats: ~140 bins each
    a["opacity"][:3500] = 1.5
    cloud = scenes.cloud_from_attrs(a)
    W, H = 800, 450
    cam, proj, vp, nf = scenes.default_view(W, H, z=6.0, yaw=0.1)
    res = []
    r = make_renderer(cloud)
    for frame in range(3):                           # frame 0: no helpers yet (unsplit); frames 1, 2: split, both list parities
        r.Sort(cam, proj, vp, nf)
        img = r.Render(cam, proj, vp, nf)
        st = _check_tile_lists_ascending(r)
        ts, pairs = r.debug_tile_lists()
        res.append((ts, pairs, img))
    assert st["pairs"] > 400_000                     # the first chunks of 1024 ranks hold > 100 k pairs each: above the 49 152 threshold
    for other in res[1:]:
        for x, y in zip(res[0], other):
            np.testing.assert_array_equal(x, y)
    ref = oracle_frame(cloud.as_array(), True, cam, proj, vp, nf)
    check_image(res[0][2], ref["image"], budget=ref["budget"])


# ------------------------------------------------------------------------------------------------
# round 4: spatial storage order (Morton-sorted cloud, one bounding box per 1024 stored splats) and the chunk-level cull.
# Contract (include/msplat.h, msplat_config.spatial_order): the visible set, the keys and the draw order are those of the
# per-splat test with ties in ascending STORAGE slot; everything reported is in upload numbering.
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,seed,hard", [(1, 3, False), (1023, 4, False), (5000, 5, True), (40000, 6, False)])
def test_spatial_order_small_scenes_exact_sort_image_and_download(n, seed, hard):
    from splatapult_amd import _capi
    cloud = scenes.cloud_from_attrs(scenes.hard_attrs(n, seed)) if hard else scenes.synth_cloud(n, seed, log_scale_mean=-3.0)
    aos = cloud.as_array()
    r = make_renderer(cloud, spatial_order=_capi.SPATIAL_ON)
    order = r.storage_order()
    if n == 1:
        assert order is None                                # nothing to reorder
    else:
        assert order is not None and np.array_equal(np.sort(order), np.arange(n))          # a permutation of the upload indices
        assert not np.array_equal(order, np.arange(n))
    np.testing.assert_array_equal(r.download_cloud(True), aos)      # the API speaks upload numbering
    off = make_renderer(cloud, spatial_order=_capi.SPATIAL_OFF)
    assert off.storage_order() is None
    for yaw, z in ((0.0, 7.0), (0.7, 5.0), (3.0, 0.5)):               # outside looking in; oblique; inside the cloud
        cam, proj, vp, nf = scenes.default_view(640, 360, yaw=yaw, z=z)
        r.Sort(cam, proj, vp, nf)
        off.Sort(cam, proj, vp, nf)
        V = _check_sort_exact(r, aos, cam, proj, nf)                  # incl. the tie rule (hard_attrs has exact duplicates)
        assert off.sort_count() == V                                  # the chunk-level cull never changes the visible set
        np.testing.assert_array_equal(np.sort(r.sorted_indices()), np.sort(off.sorted_indices()))
        np.testing.assert_array_equal(r.sorted_keys(), off.sorted_keys())
        img = r.Render(cam, proj, vp, nf)
        ref = oracle_frame(aos, True, cam, proj, vp, nf, r=r)
        check_image(img, ref["image"], budget=ref["budget"])
        assert r.verify_order() == (0, 0)
        live, total, listed = r.cull_boxes()
        assert total == (n + 255) // 256 * (order is not None) and live <= total
        # the hint for the next Sort (V of a rendered frame) is in place now: a view that sees < 70 % of the cloud is sorted again,
        # this time over the listed live boxes only -- same keys, same order
        r.Sort(cam, proj, vp, nf)
        assert r.cull_boxes()[2] == (order is not None and 0 < V < 0.7 * n)
        assert _check_sort_exact(r, aos, cam, proj, nf) == V
        np.testing.assert_array_equal(r.Render(cam, proj, vp, nf), img)


@pytest.mark.parametrize("mode", ["serial", "in_flight", "lsd8", "ballot"])
def test_chunk_level_cull_skips_boxes_and_keeps_everything_exact(mode, monkeypatch):
    """400 k splats (stored in Morton order by default), views that see little of the cloud: most bounding boxes are dead, and
    V, keys, permutation, bin lists and pixels are what the per-splat cull gives -- for every pass-0 kernel (three-pass sort in
    its 512- and 256-thread forms, the 8-bit passes, the ballot fallback)"""
    from splatapult_amd import _capi
    kw = {}
    if mode == "in_flight":
        kw["frame_mode"] = _capi.FRAMES_IN_FLIGHT
    if mode == "lsd8":
        monkeypatch.setenv("MSPLAT_SORT", "lsd8")
    if mode == "ballot":
        kw["rank_mode"] = _capi.RANK_BALLOT
    n = 400_000
    cloud = scenes.synth_cloud(n, 77, log_scale_mean=-4.5, pos_sigma=2.0)
    aos = cloud.as_array()
    r = make_renderer(cloud, **kw)
    ref_r = make_renderer(cloud, spatial_order=_capi.SPATIAL_OFF)     # per-splat cull over the whole cloud, upload order
    assert r.storage_order() is not None and ref_r.storage_order() is None
    W, H = 640, 360
    seen = []
    for k, (pos, yaw, pitch) in enumerate((((0.0, 0.0, 7.0), 0.0, 0.0),          # everything in view
                                           ((0.3, 0.2, 0.5), 1.3, 0.2),          # inside, looking sideways
                                           ((4.0, 0.0, 4.0), 2.6, 0.0),          # outside, the cloud mostly behind / beside
                                           ((0.0, 0.0, 7.0), 3.14159, 0.0))):    # looking away: nothing
        cam = camera.pose(pos, yaw, pitch)
        proj, vp, nf = camera.perspective(camera.FOVY, W / H), [0, 0, W, H], scenes.NF
        # frame 1 of a view runs on the hint of the previous view; its Render leaves THIS view's V for frames 2 and 3, whose
        # pass 0 walks the listed live boxes when the view sees less than 70 % of the cloud.  All three must agree exactly.
        ref_r.Sort(cam, proj, vp, nf)
        img = None
        for f in range(3):
            r.Sort(cam, proj, vp, nf)
            V = _check_sort_exact(r, aos, cam, proj, nf)
            assert ref_r.sort_count() == V
            np.testing.assert_array_equal(r.sorted_keys(), ref_r.sorted_keys())
            live, total, listed = r.cull_boxes()
            assert total == (n + 255) // 256
            if f >= 1:
                assert listed == (0 < V < 0.7 * n), (f, V, listed)
            im = r.Render(cam, proj, vp, nf)
            if img is not None:
                np.testing.assert_array_equal(im, img)
            img = im
        seen.append((V, live))
        _check_tile_lists_ascending(r)
        if V:
            ref = oracle_frame(aos, True, cam, proj, vp, nf, r=r)
            check_image(img, ref["image"], budget=ref["budget"])
        else:
            assert (img[..., :3] == 0).all() and live == 0
    print("chunk-level cull (%s): (V, live boxes of %d) per view: %s" % (mode, (n + 255) // 256, seen))
    assert seen[0][1] > 0.9 * total                                   # everything in view: (almost) every box is live
    assert seen[1][1] < 0.8 * total and seen[2][1] < 0.8 * total      # partial views skip boxes
    assert seen[1][0] > 1000


def test_chunk_level_cull_with_row_bands_reassembles_bit_exact():
    """the band-restricted cull at box level (box_live: the box's reach in y against the owned bin rows): every rank's Sort
    keeps what the per-splat band test keeps, the bands reassemble the unbanded frame bit for bit, a rank tests fewer boxes"""
    from splatapult_amd import _capi
    n, W, H = 400_000, 800, 800
    cloud = scenes.synth_cloud(n, 78, log_scale_mean=-4.3, pos_sigma=2.0)
    cam, proj, vp, nf = scenes.default_view(W, H, z=6.5, yaw=0.2)
    r = make_renderer(cloud)
    r.Sort(cam, proj, vp, nf)
    full = r.Render(cam, proj, vp, nf)
    live_full, total, _ = r.cull_boxes()
    off = make_renderer(cloud, spatial_order=_capi.SPATIAL_OFF)
    R = (H + bin_px() - 1) // bin_px()
    for kind, k, G in (("contiguous", 1, 8), ("block", 2, 4), ("interleaved", 1, 3)):
        acc = np.zeros_like(full)
        lives = []
        for g in range(G):
            lay = r.set_band_plan(kind, R, G, g, block_rows=k, band_cull=True)
            off.set_band_plan(kind, R, G, g, block_rows=k, band_cull=True)
            off.Sort(cam, proj, vp, nf)
            for f in range(2):                                        # the second frame of a rank runs over the listed boxes
                r.Sort(cam, proj, vp, nf)
                assert r.sort_count() == off.sort_count()             # box-level band test == per-splat band test
                np.testing.assert_array_equal(r.sorted_keys(), off.sorted_keys())
                part = r.Render(cam, proj, vp, nf)
            live, _, listed = r.cull_boxes()
            assert listed
            lives.append(live)
            rows = np.isin(np.arange(H) // bin_px(), _capi.band_rows(*lay, rows_full=R))
            acc[rows] = part[rows]
        np.testing.assert_array_equal(acc, full)
        print("band cull at box level, %s x %d: live boxes per rank %s of %d (unbanded %d)" % (kind, G, lives, total, live_full))
        assert max(lives) < live_full
    r.set_band(1, 0)


def test_spatial_order_through_the_file_path_and_shared_clouds(tmp_path, monkeypatch):
    """GPU ingest (InitFromPly) followed by the reordering: download_cloud returns the file's order; contexts that share the
    cloud (frames in flight) see the same storage order and render identical frames"""
    from splatapult_amd import synthetic
    from splatapult_amd.scene import GaussianCloud
    from splatapult_amd import _capi
    n = 6000
    ply = str(tmp_path / "c.ply")
    synthetic.write_ply(ply, synthetic.generate(n, seed=91, pos_sigma=1.5, log_scale_mean=-3.0))
    r = SplatRenderer(device=0, frames_in_flight=3, spatial_order=_capi.SPATIAL_ON)      # (small cloud: ask for the Morton storage order)
    assert r.InitFromPly(ply, True, False), r.last_error()
    host = GaussianCloud()
    assert host.ImportPly(ply)
    aos = host.as_array()
    order = r.storage_order()
    assert order is not None and np.array_equal(np.sort(order), np.arange(n))
    dl = r.download_cloud(True)
    np.testing.assert_array_equal(dl[:, :3], aos[:, :3])              # positions pass through the ingest untouched
    np.testing.assert_allclose(dl, aos, rtol=2e-5, atol=2e-6)         # (device expf / sqrtf vs glibc, see test_gpu_ingest_*)
    cam, proj, vp, nf = scenes.default_view(480, 270, z=6.0, yaw=0.4)
    imgs = []
    for _ in range(3):                                                # one frame on each of the three contexts
        r.Sort(cam, proj, vp, nf)
        assert np.array_equal(r.storage_order(), order)
        _check_sort_exact(r, dl, cam, proj, nf)                       # (the oracle gets the DEVICE cloud: exact keys)
        imgs.append(r.Render(cam, proj, vp, nf))
    np.testing.assert_array_equal(imgs[0], imgs[1])
    np.testing.assert_array_equal(imgs[0], imgs[2])
    ref = oracle_frame(dl, True, cam, proj, vp, nf, r=r)
    check_image(imgs[0], ref["image"], budget=ref["budget"])


def test_async_submit_queues_calls_and_defers_their_errors():
    """msplat_config.async_submit (on by default with frames in flight): Sort / device-output Render return at once and a worker
    thread of the context issues the launches; a queued call that fails is reported by the next synchronize, once; getters wait
    for the worker; the frames are those of a context that issues its own launches"""
    import torch
    from splatapult_amd import _capi
    cloud = scenes.synth_cloud(30000, 93, log_scale_mean=-3.4)
    W, H = 480, 270
    Hpad = (H + bin_px() - 1) // bin_px() * bin_px()
    cam, proj, vp, nf = scenes.default_view(W, H, yaw=0.2)
    dev = torch.device("cuda", 0)
    ra = make_renderer(cloud, frames_in_flight=2)                       # async by default
    rs = make_renderer(cloud, frames_in_flight=2, async_submit=False)   # the calling thread issues the launches
    fa = [torch.zeros((Hpad, W, 4), dtype=torch.float32, device=dev) for _ in range(6)]
    fs = [torch.zeros((Hpad, W, 4), dtype=torch.float32, device=dev) for _ in range(6)]
    for k in range(6):
        c = camera.translate_local(cam, dx=0.02 * k)
        for rr, f in ((ra, fa), (rs, fs)):
            rr.Sort(c, proj, vp, nf)
            rr.Render(c, proj, vp, nf, out_ptr=f[k].data_ptr(), pitch_bytes=W * 16)
    assert ra.sort_count() == rs.sort_count() > 0                        # a getter waits for the queued Sort
    ra.synchronize(); rs.synchronize()
    for k in range(6):
        assert torch.equal(fa[k], fs[k])
    # a queued Render with an unsupported viewport returns OK and surfaces at synchronize ...
    ra.Sort(cam, proj, vp, nf)
    ra.Render(cam, proj, [0, 0, 9000, 9000], nf, out_ptr=fa[0].data_ptr(), pitch_bytes=9000 * 16)
    with pytest.raises(_capi.MsplatError) as ei:
        ra.synchronize()
    assert ei.value.code == _capi.ERR_UNSUPPORTED and "queued" in str(ei.value)
    ra.synchronize()                                                      # ... once
    # ... while the synchronous context reports it at the call
    rs.Sort(cam, proj, vp, nf)
    with pytest.raises(_capi.MsplatError):
        rs.Render(cam, proj, [0, 0, 9000, 9000], nf, out_ptr=fs[0].data_ptr(), pitch_bytes=9000 * 16)
    # the context keeps working
    ra.Sort(cam, proj, vp, nf)
    ra.Render(cam, proj, vp, nf, out_ptr=fa[1].data_ptr(), pitch_bytes=W * 16)
    ra.synchronize()
    img = ra.Render(cam, proj, vp, nf)                                    # host output: synchronous, after the queue
    np.testing.assert_array_equal(img, fa[1][:H].cpu().numpy())


@pytest.mark.parametrize("fb", ["fp32", "fp16"])
def test_render_stereo_is_bit_identical_to_two_renders(fb):
    """msplat_render_stereo (r4): both eyes of one Sort in ONE chain of launches -- view 1's splats are the ranks behind view 0's,
    its bin rows are stacked on view 0's -- against the reference's call pattern, one Render per eye (app.cpp:603-607): the same
    pixels bit for bit, on a ragged viewport (360 = 11.25 bin rows), repeated (the tables it leaves), with frames in flight, and
    through the view-after-view fallbacks (host targets, a banded context)"""
    import torch
    cloud = scenes.synth_cloud(50000, 95, log_scale_mean=-3.3)
    W, H = 648, 360
    Hpad = (H + bin_px() - 1) // bin_px() * bin_px()
    cam0 = camera.pose((0.0, 0.0, 7.0), 0.15)
    eyes = [camera.translate_local(cam0, dx=-0.032), camera.translate_local(cam0, dx=+0.032)]
    projs = [camera.create_projection(-1.0, 0.8, 0.95, -0.95), camera.create_projection(-0.8, 1.0, 0.95, -0.95)]
    vp, nf = [0, 0, W, H], scenes.NF
    tdt, bpp = (torch.float16, 8) if fb == "fp16" else (torch.float32, 16)
    dev = torch.device("cuda", 0)
    r = make_renderer(cloud, fb_format=fb)
    r.Sort(eyes[0], projs[0], vp, nf)
    ref = [r.Render(eyes[e], projs[e], vp, nf) for e in range(2)]                 # one Render per eye (host images)
    assert not np.array_equal(ref[0], ref[1])
    for depth in (1, 2):
        rs = make_renderer(cloud, fb_format=fb, frames_in_flight=depth)
        fbs = [[torch.zeros((Hpad, W, 4), dtype=tdt, device=dev) for _ in range(2)] for _ in range(3)]
        for k in range(3):                                                       # three frames: the self-cleaning tables, both parities
            rs.Sort(eyes[0], projs[0], vp, nf)
            rs.RenderStereo(eyes, projs, vp, nf, out_ptrs=[t.data_ptr() for t in fbs[k]], pitch_bytes=W * bpp)
        rs.synchronize()
        for k in range(3):
            for e in range(2):
                np.testing.assert_array_equal(fbs[k][e][:H].cpu().numpy(), ref[e])
        assert rs.verify_order() == (0, 0)
        st = rs.stats()
        assert st["tiles_y"] == 2 * ((H + bin_px() - 1) // bin_px())              # the two views' bin rows, stacked
        # a mono Render on the same context afterwards is unaffected
        rs.Sort(eyes[0], projs[0], vp, nf)
        np.testing.assert_array_equal(rs.Render(eyes[1], projs[1], vp, nf), ref[1])
    # fallbacks: host targets and a banded context render view after view -- same pixels
    r.Sort(eyes[0], projs[0], vp, nf)
    both = r.RenderStereo(eyes, projs, vp, nf)
    np.testing.assert_array_equal(both[0], ref[0])
    np.testing.assert_array_equal(both[1], ref[1])
    rb = make_renderer(cloud, fb_format=fb)
    rb.set_band(2, 1)
    rb.Sort(eyes[0], projs[0], vp, nf)
    fb2 = [torch.zeros((Hpad, W, 4), dtype=tdt, device=dev) for _ in range(2)]
    rb.RenderStereo(eyes, projs, vp, nf, out_ptrs=[t.data_ptr() for t in fb2], pitch_bytes=W * bpp)
    rb.synchronize()
    rows = np.arange(H) // bin_px() % 2 == 1
    for e in range(2):
        np.testing.assert_array_equal(fb2[e][:H].cpu().numpy()[rows], ref[e][rows])


# ---- two-pass frames with occlusion feedback (msplat_config.two_pass, r4) ----
def _two_pass_pair(cloud, **kw):
    from splatapult_amd import _capi
    a = make_renderer(cloud, two_pass=_capi.TWO_PASS_OFF, **kw)
    b = make_renderer(cloud, two_pass=_capi.TWO_PASS_ON, **kw)
    return a, b


@pytest.mark.parametrize("share", [1.0 / 64.0, 0.1, 0.3, 0.75, 1.0])
def test_two_pass_frames_are_bit_identical_to_single_pass(share):
    """any share of the visible splats in the first pass gives the pixels of the single pass, bit for bit: a dense cloud seen from
    outside (most tiles saturate), the same from inside (near splats cover the screen), a sparse one (hardly any tile saturates:
    pass 2 redoes nearly everything), a ragged viewport"""
    cases = [(scenes.synth_cloud(120000, 301, log_scale_mean=-2.6), dict(z=5.0, yaw=0.3), 640, 360),
             (scenes.synth_cloud(120000, 301, log_scale_mean=-2.6), dict(z=0.8, yaw=2.0), 640, 360),
             (scenes.synth_cloud(20000, 302, log_scale_mean=-4.2), dict(z=7.0, yaw=0.0), 517, 293),
             (scenes.cloud_from_attrs(scenes.hard_attrs(6000, 17)), dict(z=6.0, yaw=0.7), 400, 300)]
    frames = 0
    for cloud, view, W, H in cases:
        a, b = _two_pass_pair(cloud)
        b.two_pass_state(share)
        for k in range(3):
            cam, proj, vp, nf = scenes.default_view(W, H, z=view["z"], yaw=view["yaw"] + 0.4 * k)
            a.Sort(cam, proj, vp, nf); b.Sort(cam, proj, vp, nf)
            ia, ib = a.Render(cam, proj, vp, nf), b.Render(cam, proj, vp, nf)
            np.testing.assert_array_equal(ia, ib)
            assert a.sort_count() == b.sort_count()
            frames += 1
        assert b.two_pass_state(share)[0] == 3 and a.two_pass_state()[0] == 0
        assert b.verify_order() == (0, 0)
    assert frames == 12


def test_two_pass_with_bands_fp16_frames_in_flight_and_the_feedback_loop():
    """the same identity for a rank of a row-sharded frame (block layout, band-culled sort: virtual rows in the mask), the fp16
    target, device output with four frames in flight, and with the share left to the feedback loop over a moving camera"""
    import torch
    from splatapult_amd import _capi
    cloud = scenes.synth_cloud(150000, 303, log_scale_mean=-2.8)
    W, H = 800, 448
    tiles_y = (H + bin_px() - 1) // bin_px()
    for g in range(3):
        a, b = _two_pass_pair(cloud)
        for r in (a, b):
            r.set_band_plan("block", tiles_y, 3, g, block_rows=2, band_cull=True)
        b.two_pass_state(0.2)
        for k in range(2):
            cam, proj, vp, nf = scenes.default_view(W, H, z=4.0, yaw=0.5 + k)
            a.Sort(cam, proj, vp, nf); b.Sort(cam, proj, vp, nf)
            np.testing.assert_array_equal(a.Render(cam, proj, vp, nf), b.Render(cam, proj, vp, nf))
        assert b.two_pass_state(0.2)[0] == 2
    a, b = _two_pass_pair(cloud, fb_format="fp16")
    cam, proj, vp, nf = scenes.default_view(W, H, z=4.0, yaw=0.2)
    a.Sort(cam, proj, vp, nf); b.Sort(cam, proj, vp, nf)
    np.testing.assert_array_equal(a.Render(cam, proj, vp, nf), b.Render(cam, proj, vp, nf))
    # four frames in flight, the share steered by the feedback of earlier frames
    dev = torch.device("cuda", 0)
    Hpad = tiles_y * bin_px()
    a, b = _two_pass_pair(cloud, frames_in_flight=4)
    n = 48
    fa = [torch.zeros((Hpad, W, 4), dtype=torch.float32, device=dev) for _ in range(n)]
    fb = [torch.zeros((Hpad, W, 4), dtype=torch.float32, device=dev) for _ in range(n)]
    for k in range(n):
        cam, proj, vp, nf = scenes.default_view(W, H, z=4.0 - 0.05 * k, yaw=0.05 * k)
        for r, f in ((a, fa), (b, fb)):
            r.Sort(cam, proj, vp, nf)
            r.Render(cam, proj, vp, nf, out_ptr=f[k].data_ptr(), pitch_bytes=W * 16)
    a.synchronize(); b.synchronize()
    for k in range(n):
        assert torch.equal(fa[k], fb[k]), k
    frames, share = b.two_pass_state()
    assert frames == n and 1.0 / 256.0 <= share <= 0.75
    print("two-pass frames in flight: share of the visible splats in pass 1 after %d frames: %.3f" % (n, share))
    # AUTO: off for a small cloud and during a context's first frames
    c = make_renderer(cloud)
    for k in range(12):
        cam, proj, vp, nf = scenes.default_view(W, H, z=4.0, yaw=0.1 * k)
        c.Sort(cam, proj, vp, nf)
        c.Render(cam, proj, vp, nf)
    assert c.two_pass_state()[0] == 0                       # 150 k splats: below the AUTO size
    # ... on for a large dense one after the context's first 8 frames (and off again for the probe / stats frames)
    big = scenes.synth_cloud(300000, 304, log_scale_mean=-2.9)
    a = make_renderer(big, two_pass=_capi.TWO_PASS_OFF)
    c = make_renderer(big, frame_mode=_capi.FRAMES_IN_FLIGHT)      # (a context that shares the GPU with other frames)
    lat = make_renderer(big)                                       # one frame at a time, few visible splats: stays in one pass
    for k in range(24):
        cam, proj, vp, nf = scenes.default_view(W, H, z=3.0, yaw=0.07 * k)
        a.Sort(cam, proj, vp, nf); c.Sort(cam, proj, vp, nf); lat.Sort(cam, proj, vp, nf)
        ref = a.Render(cam, proj, vp, nf)
        np.testing.assert_array_equal(ref, c.Render(cam, proj, vp, nf))
        np.testing.assert_array_equal(ref, lat.Render(cam, proj, vp, nf))
    assert lat.two_pass_state()[0] == 0
    frames, share = c.two_pass_state()
    info = c.two_pass_info()
    assert frames >= 8 and info is not None and info["splats_pass1"] < info["visible"]
    print("two-pass AUTO on 300 k splats: %d of 24 frames, share %.3f, latest frame: %s" % (frames, share, info))
    c.set_tile_probe(True)
    c.Sort(cam, proj, vp, nf)
    np.testing.assert_array_equal(a.Render(cam, proj, vp, nf), c.Render(cam, proj, vp, nf))
    assert c.two_pass_info() is None and c.stats()["pairs"] == a.stats()["pairs"]


def test_two_pass_nothing_visible_and_tiny_viewports():
    """forced two-pass frames with no visible splat (camera looking away), one visible splat, a viewport smaller than a bin"""
    from splatapult_amd import _capi
    cloud = scenes.synth_cloud(5000, 311, log_scale_mean=-3.0)
    for W, H, z, yaw in ((320, 200, -30.0, 0.0), (320, 200, 6.0, 0.0), (17, 9, 6.0, 0.3), (33, 31, 2.0, 1.0)):
        a, b = _two_pass_pair(cloud)
        b.two_pass_state(0.3)
        cam, proj, vp, nf = scenes.default_view(W, H, z=z, yaw=yaw)
        if z < 0:
            cam = camera.pose((0.0, 0.0, 30.0))            # the cloud is behind the camera (it looks down -z from z = 30 ... away: flip below)
            cam = np.array(cam, np.float32).reshape(4, 4).copy()
            cam[2, :3] *= -1.0; cam[0, :3] *= -1.0         # rotate 180 degrees about y: now it looks away from the cloud
        for _ in range(2):
            a.Sort(cam, proj, vp, nf); b.Sort(cam, proj, vp, nf)
            np.testing.assert_array_equal(a.Render(cam, proj, vp, nf), b.Render(cam, proj, vp, nf))
        assert a.sort_count() == b.sort_count()
        if z < 0:
            assert a.sort_count() == 0
    one = scenes.synth_cloud(1, 312, log_scale_mean=-2.0)
    a, b = _two_pass_pair(one)
    b.two_pass_state(1.0 / 256.0)
    cam, proj, vp, nf = scenes.default_view(200, 120, z=5.0)
    a.Sort(cam, proj, vp, nf); b.Sort(cam, proj, vp, nf)
    np.testing.assert_array_equal(a.Render(cam, proj, vp, nf), b.Render(cam, proj, vp, nf))


def test_two_pass_fuzz():
    """tools/two_pass_fuzz.py, shortened: random clouds / cameras / viewports / band plans / targets / shares"""
    import os
    import subprocess
    import sys
    from tests.conftest import ROOT
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "two_pass_fuzz.py"), "--cases", "40", "--seed", "3"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "0 mismatches" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_two_pass_with_a_pair_buffer_that_overflows():
    """two-pass frames and the (splat, bin) pair capacity: the host-output Render grows the buffer and retries until the frame is
    the single pass's frame; a device-output Render that overflowed in EITHER pass reports it on the next call and renders
    correctly after the growth"""
    import torch
    from splatapult_amd import MsplatError, _capi
    cloud = scenes.synth_cloud(12000, 123, log_scale_mean=-0.5, pos_sigma=1.0)      # ~10 M pairs at 1024 x 1024, capacity starts at 4 M
    W = H = 1024
    cam, proj, vp, nf = scenes.default_view(W, H, z=4.0)
    want = make_renderer(cloud, two_pass=_capi.TWO_PASS_OFF)
    want.Sort(cam, proj, vp, nf)
    expect = want.Render(cam, proj, vp, nf)
    for share in (0.05, 0.5, 1.0):
        r = make_renderer(cloud, two_pass=_capi.TWO_PASS_ON)
        r.two_pass_state(share)
        r.Sort(cam, proj, vp, nf)
        np.testing.assert_array_equal(r.Render(cam, proj, vp, nf), expect)       # host output: grows synchronously and retries
        assert r.two_pass_state(share)[0] >= 1
        dev = torch.device("cuda", 0)
        fb = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
        r = make_renderer(cloud, two_pass=_capi.TWO_PASS_ON)
        r.two_pass_state(share)
        r.Sort(cam, proj, vp, nf)
        r.Render(cam, proj, vp, nf, out_ptr=fb.data_ptr(), pitch_bytes=W * 16)
        try:
            r.synchronize()
            overflowed = False
        except MsplatError as e:
            assert e.code == _capi.ERR_PAIR_OVERFLOW
            overflowed = True
        if overflowed:
            r.Sort(cam, proj, vp, nf)
            r.Render(cam, proj, vp, nf, out_ptr=fb.data_ptr(), pitch_bytes=W * 16)
            r.synchronize()
        np.testing.assert_array_equal(fb.cpu().numpy(), expect)


def test_every_combination_of_scheduling_modes_gives_the_same_pixels():
    """the modes were tested in pairs, not as a lattice.  Target format x row bands x two-pass frames x
    frames in flight x (one view | both eyes in one chain): 32 combinations of one small dense scene, every one against the plain
    fp32 context's two views -- bit for bit (fp16 targets: the fp32 frame rounded once to fp16, which is what the compositor
    stores).  Combinations the library schedules differently (two passes are not used for two views in one chain; banded
    contexts render two views one after the other) are in the product on purpose: the CALLER'S pixels must not depend on that."""
    import itertools
    import torch
    from splatapult_amd import _capi
    cloud = scenes.synth_cloud(60000, 611, log_scale_mean=-2.9)
    W, H = 640, 352                                          # 11 bin rows exactly: device targets need no padding
    T = bin_px()
    tiles_y = H // T
    cam0 = camera.pose((0.1, -0.05, 5.5), 0.1, 0.03)
    cams = [camera.translate_local(cam0, dx=-0.032), camera.translate_local(cam0, dx=+0.032)]
    projs = [camera.create_projection(-1.0, 0.8, 0.95, -0.95), camera.create_projection(-0.8, 1.0, 0.95, -0.95)]
    vp, nf = [0, 0, W, H], scenes.NF
    plain = make_renderer(cloud)
    plain.Sort(cams[0], projs[0], vp, nf)                    # one sort with the first eye (app.cpp:603-607)
    ref = [plain.Render(cams[v], projs[v], vp, nf) for v in range(2)]
    assert all((im[..., :3] != 0).any() for im in ref)
    dev = torch.device("cuda", 0)
    combos = 0
    for fmt, bands, two_pass, fif, stereo in itertools.product(("fp32", "fp16"), (1, 3), (False, True), (1, 3), (False, True)):
        tdt, bpp, ndt = (torch.float16, 8, np.float16) if fmt == "fp16" else (torch.float32, 16, np.float32)
        r = make_renderer(cloud, fb_format=fmt, frames_in_flight=fif, two_pass=_capi.TWO_PASS_ON if two_pass else _capi.TWO_PASS_OFF)
        if two_pass:
            r.two_pass_state(0.2)
        got = [np.zeros((H, W, 4), ndt) for _ in range(2)]
        fbs = [torch.zeros((H, W, 4), dtype=tdt, device=dev) for _ in range(2)]
        for rep in range(2 if fif > 1 else 1):               # frames in flight: a second round lands on other contexts
            for g in range(bands):
                lay = None
                if bands > 1:
                    lay = r.set_band_plan("block", tiles_y, bands, g, block_rows=2, band_cull=False)
                r.Sort(cams[0], projs[0], vp, nf)
                for t in fbs:
                    t.zero_()
                torch.cuda.synchronize()
                if stereo:
                    r.RenderStereo(cams, projs, vp, nf, out_ptrs=[t.data_ptr() for t in fbs], pitch_bytes=W * bpp)
                else:
                    for v in range(2):
                        r.Render(cams[v], projs[v], vp, nf, out_ptr=fbs[v].data_ptr(), pitch_bytes=W * bpp)
                r.synchronize()
                torch.cuda.synchronize()
                rows = np.ones(H, bool) if lay is None else np.isin(np.arange(H) // T, _capi.band_rows(*lay, rows_full=tiles_y))
                for v in range(2):
                    got[v][rows] = fbs[v].cpu().numpy()[rows]
            for v in range(2):
                want = ref[v] if fmt == "fp32" else ref[v].astype(np.float16)
                np.testing.assert_array_equal(got[v], want, err_msg="target %s, %d band(s), two_pass %s, %d in flight, stereo %s, view %d"
                                              % (fmt, bands, two_pass, fif, stereo, v))
        r.close()
        combos += 1
    assert combos == 32


def test_every_combination_of_the_render_target_emulations_gives_the_same_pixels():
    """the other half of the lattice: emulated depth buffer (0 / 24 bits) x render-target rounding (none / RGBA8 / RGBA16F) x row
    bands x frames in flight, the second eye drawn in the first eye's order (where the depth test matters): every combination
    against the same emulation on a plain context.  The draw-order compositor takes over whenever one of the two is set; two-pass
    frames requested on such a context run in one pass."""
    import itertools
    import torch
    from splatapult_amd import _capi
    cloud = scenes.synth_cloud(25000, 612, log_scale_mean=-2.9)
    W, H = 480, 288
    T = bin_px()
    tiles_y = H // T
    cam0 = camera.pose((0.0, 0.0, 5.5), 0.05, 0.0)
    cams = [camera.translate_local(cam0, dx=-0.032), camera.translate_local(cam0, dx=+0.032)]
    proj = camera.perspective(camera.FOVY, W / H)
    vp, nf = [0, 0, W, H], scenes.NF
    dev = torch.device("cuda", 0)
    combos = 0
    for depth_bits, rop in itertools.product((0, 24), (None, "rgba8", "fp16")):
        if depth_bits == 0 and rop is None:
            continue                                         # the plain compositor: the test above
        plain = make_renderer(cloud)
        plain.set_depth_test(depth_bits)
        plain.set_target_emulation(rop)
        plain.Sort(cams[0], proj, vp, nf)
        ref = plain.Render(cams[1], proj, vp, nf)            # the second eye in the first eye's order
        for bands, fif in itertools.product((1, 3), (1, 3)):
            r = make_renderer(cloud, frames_in_flight=fif, two_pass=_capi.TWO_PASS_ON)
            r.set_depth_test(depth_bits)
            r.set_target_emulation(rop)
            fb = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
            got = np.zeros((H, W, 4), np.float32)
            for rep in range(2 if fif > 1 else 1):
                for g in range(bands):
                    lay = r.set_band_plan("interleaved", tiles_y, bands, g) if bands > 1 else None
                    r.Sort(cams[0], proj, vp, nf)
                    fb.zero_()
                    torch.cuda.synchronize()
                    r.Render(cams[1], proj, vp, nf, out_ptr=fb.data_ptr(), pitch_bytes=W * 16)
                    r.synchronize()
                    torch.cuda.synchronize()
                    rows = np.ones(H, bool) if lay is None else np.isin(np.arange(H) // T, _capi.band_rows(*lay, rows_full=tiles_y))
                    got[rows] = fb.cpu().numpy()[rows]
                np.testing.assert_array_equal(got, ref, err_msg="depth %d, rop %s, %d band(s), %d in flight" % (depth_bits, rop, bands, fif))
            assert r.two_pass_state()[0] == 0               # the draw-order compositor never runs in two passes
            r.close()
            combos += 1
        plain.close()
    assert combos == 20


@pytest.mark.gpu
def test_cu_partition_halves_for_four_frames_in_flight():
    """r6, msplat_config.cu_partition: with four frames in flight the shims' contexts alternate between the even and the odd CU
    positions of every XCD (hipExtStreamCreateWithCUMask).  The streams really carry the masks -- 128 CUs each, disjoint, together
    all 256 --, the frames are the serial frames bit for bit, one context on a half renders the same pixels as one on every CU, a
    caller's stream is never masked, and a bad value is refused."""
    import torch
    from splatapult_amd import _capi
    cloud = scenes.synth_cloud(80000, 92, log_scale_mean=-3.4)
    W, H = 800, 450
    Hpad = (H + bin_px() - 1) // bin_px() * bin_px()
    views = [scenes.default_view(W, H, yaw=0.07 * k, x=0.04 * k) for k in range(9)]
    r1 = make_renderer(cloud)
    assert r1.cu_partitions()[0][0] == _capi.CU_ALL
    serial = []
    for cam, proj, vp, nf in views:
        r1.Sort(cam, proj, vp, nf)
        serial.append(r1.Render(cam, proj, vp, nf))
    rp = make_renderer(cloud, frames_in_flight=4)
    parts = rp.cu_partitions()
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        # (a partitioned MI355X: the library keeps every stream on every CU; the rest of this test is about the masks)
        assert [p for p, _ in parts] == [_capi.CU_ALL] * 4
        pytest.skip("not a 256-CU device: msplat_config.cu_partition falls back to every CU")
    assert [p for p, _ in parts] == [_capi.CU_EVEN, _capi.CU_ODD, _capi.CU_EVEN, _capi.CU_ODD]
    bits = [int.from_bytes(np.asarray(m, np.uint32).tobytes(), "little") for _, m in parts]
    assert all(bin(b).count("1") == 128 for b in bits), [bin(b).count("1") for b in bits]
    assert bits[0] == bits[2] and bits[1] == bits[3] and bits[0] & bits[1] == 0 and bits[0] | bits[1] == (1 << 256) - 1
    assert make_renderer(cloud, frames_in_flight=3).cu_partitions()[1][0] == _capi.CU_ALL        # the rule: an even depth >= 4
    assert make_renderer(cloud, frames_in_flight=4, cu_partition=False).cu_partitions()[3][0] == _capi.CU_ALL
    dev = torch.device("cuda", 0)
    fbs = [torch.zeros((Hpad, W, 4), dtype=torch.float32, device=dev) for _ in views]
    torch.cuda.synchronize()
    for k, (cam, proj, vp, nf) in enumerate(views):
        rp.Sort(cam, proj, vp, nf)
        rp.Render(cam, proj, vp, nf, out_ptr=fbs[k].data_ptr(), pitch_bytes=W * 16)
    rp.synchronize()
    for k in range(len(views)):
        assert np.array_equal(fbs[k][:H].cpu().numpy(), serial[k]), k
    # one context on one half: the same frame
    for part in (_capi.CU_EVEN, _capi.CU_ODD):
        rh = make_renderer(cloud, cu_partition=part)
        assert rh.cu_partitions()[0][0] == part
        cam, proj, vp, nf = views[3]
        rh.Sort(cam, proj, vp, nf)
        assert np.array_equal(rh.Render(cam, proj, vp, nf), serial[3])
    # a caller's stream is used as it is
    st = torch.cuda.Stream()
    rc = make_renderer(cloud, stream=st.cuda_stream, cu_partition=_capi.CU_ODD)
    assert rc.cu_partitions()[0][0] == _capi.CU_ALL
    bad = SplatRenderer(device=0, cu_partition=3)
    assert not bad.Init(cloud, False, False) and "cu_partition" in bad.last_error()
