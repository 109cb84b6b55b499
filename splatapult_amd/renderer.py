"""SplatRenderer: Python mirror of the reference's renderer seam
(/root/reference/src/splatrenderer.h:23-67): Init(gaussianCloud, isFramebufferSRGBEnabled,
useRgcSortOverride) -> bool, Sort(cameraMat, projMat, viewport, nearFar), Render(same four).
The reference renders into the currently bound GL framebuffer; here Render takes/returns an explicit
RGBA framebuffer.  Everything runs in libmsplat.so's HIP kernels -- there is no CPU path."""
import ctypes as C

import numpy as np

from . import _capi
from .scene import GaussianCloud


class _FrameArgs:
    """cameraMat, projMat, viewport, nearFar of one Sort / Render call, marshalled into ONE preallocated float buffer whose
    four pointers are made once: building four ctypes pointers per call cost 20 us, a third of the time it takes the host
    to issue a frame's launches (r3, bench.py's host_enqueue_ms_per_frame).  The library copies what it needs
    during the call, so the buffer is free again when the call returns."""
    __slots__ = ("buf", "cam", "proj", "vp", "nf", "p_cam", "p_proj", "p_vp", "p_nf")

    def __init__(self):
        self.buf = np.zeros(38, np.float32)
        self.cam, self.proj, self.vp, self.nf = self.buf[0:16], self.buf[16:32], self.buf[32:36], self.buf[36:38]
        base = self.buf.ctypes.data
        fp = C.POINTER(C.c_float)
        self.p_cam, self.p_proj = C.cast(base, fp), C.cast(base + 64, fp)
        self.p_vp, self.p_nf = C.cast(base + 128, fp), C.cast(base + 144, fp)

    @staticmethod
    def _flat(a, n):
        a = np.asarray(a, np.float32).reshape(-1)
        if a.shape[0] != n:
            raise AssertionError("expected %d floats" % n)
        return a

    def load(self, cameraMat, projMat, viewport, nearFar):
        f = self._flat
        self.cam[:] = f(cameraMat, 16)
        self.proj[:] = f(projMat, 16)
        self.vp[:] = f(viewport, 4)
        self.nf[:] = f(nearFar, 2)
        return self.p_cam, self.p_proj, self.p_vp, self.p_nf


class SplatRenderer:
    def __init__(self, device=0, fb_format="fp32", t_epsilon=-1.0, pair_capacity=0, stream=None,
                 enable_timing=False, frames_in_flight=1, rank_mode=_capi.RANK_AUTO, frame_mode=None,
                 spatial_order=_capi.SPATIAL_AUTO, async_submit=None, two_pass=_capi.TWO_PASS_AUTO, compositor_waves=None,
                 cu_partition=None):
        """frames_in_flight > 1: every Sort moves on to the next of that many contexts (own stream and
        per-frame buffers, ONE shared cloud -- msplat_attach_cloud), so successive frames overlap on the
        GPU; Render and the getters use the context of the latest Sort.  `stream` is only used with depth 1;
        consume a frame after wait_on_stream() / synchronize(), one framebuffer per frame in flight."""
        self.numBlocksPerWorkgroup = 1024      # accepted and ignored (splatrenderer.h:39)
        self._lib = _capi.lib()
        self._ctx = None
        self._ctxs = []
        self._cur = 0
        self._depth = max(1, int(frames_in_flight))
        self._args = _FrameArgs()
        self._device = device
        self._fb_format = {"fp32": _capi.FB_RGBA32F, "fp16": _capi.FB_RGBA16F}[fb_format]
        self._t_eps = t_epsilon
        self._pair_cap = pair_capacity
        self._stream = stream
        self._timing = enable_timing
        self._rank_mode = int(rank_mode)       # msplat_config.rank_mode (RANK_AUTO / RANK_BALLOT)
        self._spatial = int(spatial_order)     # msplat_config.spatial_order (SPATIAL_AUTO / _ON / _OFF)
        # msplat_config.async_submit: Sort / device-output Render are queued to a worker thread of their context (default: on
        # with frames in flight -- the frames' launches are then issued concurrently instead of one context after the other)
        self._async = (self._depth > 1) if async_submit is None else bool(async_submit)
        # msplat_config.two_pass: may a Render run as two passes with occlusion feedback (same pixels, less work)?
        self._two_pass = int(two_pass)
        # msplat_config.compositor_waves: persistent compositor waves per render (None: every item its own wave for one frame at a
        # time, 1280 with frames in flight -- measured r3: pool sweep 768 .. 2048)
        self._comp_waves = compositor_waves
        # msplat_config.cu_partition (r6): None = the shims' rule -- with an even number >= 4 of frames in flight the contexts' streams
        # alternate between the even and the odd CU positions of every XCD (two frames per half: +3-5 %), else every CU;
        # False / 0 = every CU; or one MSPLAT_CU_* per context
        self._cu_partition = cu_partition
        # msplat_config.frame_mode: kernels for one frame at a time, or for contexts that share the GPU with other frames in
        # flight (msplat.h, MSPLAT_FRAMES_*)
        self._frame_mode = int(frame_mode) if frame_mode is not None else (_capi.FRAMES_IN_FLIGHT if self._depth > 1 else _capi.FRAMES_AUTO)
        self._n = 0

    def __del__(self):
        self.close()

    def close(self):
        ctxs, self._ctxs, self._ctx = getattr(self, "_ctxs", []), [], None
        for ctx in ctxs:
            self._lib.msplat_destroy(ctx)

    # -- reference surface ------------------------------------------------------------------
    def Init(self, gaussianCloud, isFramebufferSRGBEnabled=False, useRgcSortOverride=False):
        """splatrenderer.cpp:50-151.  Returns False (error text via last_error()) on failure, like the reference."""
        del useRgcSortOverride   # one HIP sort replaces both GL sorters
        if not self._create(isFramebufferSRGBEnabled):
            return False
        if isinstance(gaussianCloud, GaussianCloud):
            rc = self._lib.msplat_upload_gaussian_cloud(self._ctx, gaussianCloud.handle)
            self._n = gaussianCloud.GetNumGaussians()
        else:   # (N, 25|61) float32 array in the reference AoS layout
            aos = np.ascontiguousarray(gaussianCloud, np.float32)
            full = aos.shape[1] == 61
            assert aos.shape[1] in (25, 61)
            off = _capi.AttrOffsets(0, 16, 32, 48, 64, 76, 88, 100, 116, 132, 148, 164, 180, 196, 212, 228)
            rc = self._lib.msplat_upload_cloud(self._ctx, aos.ctypes.data, aos.shape[0], aos.shape[1] * 4,
                                               C.byref(off), 1 if full else 0)
            self._n = aos.shape[0]
        if rc != _capi.OK:
            self._err = self._lib.msplat_last_error(self._ctx).decode()
            return False
        return self._attach_all()

    def _create(self, isFramebufferSRGBEnabled):
        """one context per frame in flight; context 0 receives the cloud"""
        self.close()
        cfg = _capi.Config()
        cfg.struct_size = C.sizeof(_capi.Config)
        cfg.device = self._device
        cfg.fb_format = self._fb_format
        cfg.srgb = 1 if isFramebufferSRGBEnabled else 0
        cfg.t_epsilon = self._t_eps
        cfg.pair_capacity = self._pair_cap
        cfg.enable_timing = int(self._timing)
        cfg.compositor_waves = int(self._comp_waves) if self._comp_waves is not None else (0 if self._depth == 1 else 1280)
        cfg.rank_mode = self._rank_mode
        cfg.frame_mode = self._frame_mode
        cfg.spatial_order = self._spatial
        cfg.async_submit = 1 if self._async else 0
        cfg.two_pass = self._two_pass
        halves = self._depth >= 4 and self._depth % 2 == 0 if self._cu_partition is None else False
        for k in range(self._depth):
            if isinstance(self._cu_partition, (list, tuple)):
                cfg.cu_partition = int(self._cu_partition[k])
            else:
                cfg.cu_partition = (_capi.CU_EVEN + (k & 1)) if halves else int(self._cu_partition or 0)
            if isinstance(self._stream, (list, tuple)):       # one caller-owned stream per frame in flight
                cfg.stream = self._stream[k]
            else:
                cfg.stream = self._stream if self._depth == 1 else None
            h = C.c_void_p()
            rc = self._lib.msplat_create(C.byref(h), C.byref(cfg))
            if rc != _capi.OK:
                self._err = self._lib.msplat_last_error(None).decode()
                self.close()
                return False
            self._ctxs.append(h)
        self._ctx = self._ctxs[0]
        self._cur = self._depth - 1          # the first Sort lands on context 0
        return True

    def _attach_all(self):
        for h in self._ctxs[1:]:
            if self._lib.msplat_attach_cloud(h, self._ctxs[0]) != _capi.OK:
                self._err = self._lib.msplat_last_error(h).decode()
                return False
        return True

    def InitFromPly(self, plyFilename, importFullSH=True, isFramebufferSRGBEnabled=False):
        """GPU ingest: Ply::Parse on the host, then GaussianCloud::ImportPly's per-vertex
        math (gaussiancloud.cpp:254-361) as a HIP kernel straight into the renderer's device layout."""
        if not self._create(isFramebufferSRGBEnabled):
            return False
        rc = self._lib.msplat_upload_ply(self._ctx, str(plyFilename).encode(), 1 if importFullSH else 0)
        if rc != _capi.OK:
            self._err = self._lib.msplat_last_error(self._ctx).decode() or "PLY open/parse failure"
            return False
        self._n = self.stats()["num_splats"]
        return self._attach_all()

    def download_cloud(self, full_sh):
        """device cloud as (N, 25|61) float32 in the reference record layout (parity tests)"""
        out = np.zeros((max(self._n, 1), 61 if full_sh else 25), np.float32)
        _capi.check(self._ctx, self._lib.msplat_download_cloud(self._ctx, out.ctypes.data, out.nbytes))
        return out[:self._n]

    def last_error(self):
        if self._ctx:
            return self._lib.msplat_last_error(self._ctx).decode()
        return getattr(self, "_err", "")

    def Sort(self, cameraMat, projMat, viewport, nearFar):
        """splatrenderer.cpp:153-312"""
        c, p, v, nf = self._args.load(cameraMat, projMat, viewport, nearFar)
        self._cur = (self._cur + 1) % len(self._ctxs)
        self._ctx = self._ctxs[self._cur]
        _capi.check(self._ctx, self._lib.msplat_sort(self._ctx, c, p, v, nf))

    def Render(self, cameraMat, projMat, viewport, nearFar, out=None, out_ptr=None, pitch_bytes=0):
        """splatrenderer.cpp:315-343 + the GL pipeline behind it.
        out=None      -> returns a new (H, W, 4) numpy array (float32 or float16), row 0 = GL bottom row
        out=ndarray   -> filled in place
        out_ptr=int   -> device pointer (e.g. torch tensor .data_ptr()); asynchronous on the stream"""
        c, p, v, nf = self._args.load(cameraMat, projMat, viewport, nearFar)
        vp = self._args.vp
        if out_ptr is not None:
            _capi.check(self._ctx, self._lib.msplat_render(self._ctx, c, p, v, nf, C.c_void_p(out_ptr),
                                                           pitch_bytes, 1))
            return None
        W, H = int(vp[2]), int(vp[3])
        dt = np.float16 if self._fb_format == _capi.FB_RGBA16F else np.float32
        if out is None:
            out = np.zeros((H, W, 4), dt)
        assert out.dtype == dt and out.shape == (H, W, 4) and out.flags["C_CONTIGUOUS"]
        _capi.check(self._ctx, self._lib.msplat_render(self._ctx, c, p, v, nf, out.ctypes.data, 0, 0))
        return out

    def RenderStereo(self, cameraMats, projMats, viewport, nearFar, out_ptrs=None, pitch_bytes=0):
        """both eyes of the latest Sort in ONE chain of launches (msplat_render_stereo; the reference renders them one after the
        other, app.cpp:603-607): same pixels as two Render calls.  out_ptrs: two device pointers (asynchronous), else two host
        arrays are returned"""
        a0, a1 = self._args, getattr(self, "_args1", None)
        if a1 is None:
            a1 = self._args1 = _FrameArgs()
        c0, p0, v, nf = a0.load(cameraMats[0], projMats[0], viewport, nearFar)
        c1, p1, _, _ = a1.load(cameraMats[1], projMats[1], viewport, nearFar)
        if out_ptrs is not None:
            _capi.check(self._ctx, self._lib.msplat_render_stereo(self._ctx, c0, p0, c1, p1, v, nf, C.c_void_p(out_ptrs[0]),
                                                                  C.c_void_p(out_ptrs[1]), pitch_bytes, 1))
            return None
        W, H = int(a0.vp[2]), int(a0.vp[3])
        dt = np.float16 if self._fb_format == _capi.FB_RGBA16F else np.float32
        outs = [np.zeros((H, W, 4), dt), np.zeros((H, W, 4), dt)]
        _capi.check(self._ctx, self._lib.msplat_render_stereo(self._ctx, c0, p0, c1, p1, v, nf, outs[0].ctypes.data,
                                                              outs[1].ctypes.data, 0, 0))
        return outs

    # -- extensions -------------------------------------------------------------------------
    def set_band(self, row_mod, row_rem, band_cull=False):
        """interleaved rows: this renderer owns the bin rows t with t % row_mod == row_rem"""
        for h in self._ctxs:
            _capi.check(h, self._lib.msplat_set_band(h, row_mod, row_rem))
            _capi.check(h, self._lib.msplat_set_band_cull(h, 1 if band_cull else 0))

    def set_band_layout(self, first_row, row_count, block, stride, band_cull=False):
        """general form (msplat_set_band_layout): blocks of `block` bin rows at first_row, first_row + stride, ..."""
        for h in self._ctxs:
            _capi.check(h, self._lib.msplat_set_band_layout(h, first_row, row_count, block, stride))
            _capi.check(h, self._lib.msplat_set_band_cull(h, 1 if band_cull else 0))

    def set_band_plan(self, kind, rows_full, world, rank, block_rows=1, band_cull=False):
        """rank `rank` of `world` under the layout `kind` ("contiguous" | "interleaved" | "block"); returns the parameters"""
        lay = _capi.band_plan(kind, rows_full, world, rank, block_rows)
        self.set_band_layout(*lay, band_cull=band_cull)
        return lay

    def set_depth_test(self, depth_bits):
        """emulated depth buffer: 0 = colour-only target (default), 24 = default back buffer,
        32 = float depth attachment"""
        for h in self._ctxs:
            _capi.check(h, self._lib.msplat_set_depth_test(h, int(depth_bits)))

    def set_target_emulation(self, rop):
        """the blend as the GL app's render target performs it: "rgba8" = clamp + 8-bit unorm after every
        blend (default back buffer), "fp16" = fp16 rounding after every blend (--fp16), None = float accumulation"""
        mode = {None: _capi.ROP_NONE, "none": _capi.ROP_NONE, "rgba8": _capi.ROP_RGBA8, "fp16": _capi.ROP_RGBA16F}[rop]
        for h in self._ctxs:
            _capi.check(h, self._lib.msplat_set_target_emulation(h, mode))

    def synchronize(self):
        """blocks until every frame in flight has been issued (async_submit) AND has finished on the GPU"""
        for h in self._ctxs:
            _capi.check(h, self._lib.msplat_synchronize(h))

    def wait_on_stream(self, stream):
        """device-side join: `stream` (hipStream_t handle, e.g. torch.cuda.Stream.cuda_stream; None or 0 = the
        default stream) waits for the frame issued last"""
        _capi.check(self._ctx, self._lib.msplat_stream_wait(self._ctx, C.c_void_p(stream or 0)))

    def band_exchange(self, comm, rank, world, root, kind, block_rows, fb_ptr, pitch_bytes, width, height, loopback_src=None,
                      wire_fp16=False):
        """msplat_band_exchange on the current context's stream (comm = ncclComm_t handle, e.g. dist.RcclComm().handle): rank
        `root` receives every other rank's runs of bin rows straight into its framebuffer, the owners send theirs.
        loopback_src: one-rank test form (msplat_debug_band_exchange_loopback): this rank's runs travel from loopback_src to fb_ptr;
        wire_fp16 (RGBA32F targets): the rows cross the link as RGBA16F (MSPLAT_EXCHANGE_WIRE_FP16)"""
        flags = _capi.EXCHANGE_WIRE_FP16 if wire_fp16 else 0
        if loopback_src is not None:
            rc = self._lib.msplat_debug_band_exchange_loopback(self._ctx, comm, kind, block_rows, world, rank, C.c_void_p(loopback_src),
                                                               C.c_void_p(fb_ptr), pitch_bytes, width, height, flags)
        else:
            rc = self._lib.msplat_band_exchange(self._ctx, comm, rank, world, root, kind, block_rows, C.c_void_p(fb_ptr), pitch_bytes,
                                                width, height, flags)
        if rc != _capi.OK:
            raise _capi.MsplatError(rc, self._lib.msplat_group_last_error(None).decode())

    def next_frame_wait_event(self, event):
        """the context the NEXT Sort will use waits for `event` (hipEvent_t handle, e.g. torch.cuda.Event
        .cuda_event recorded after the consumer of the framebuffer that frame is going to overwrite)"""
        h = self._ctxs[(self._cur + 1) % len(self._ctxs)]
        _capi.check(h, self._lib.msplat_wait_event(h, C.c_void_p(event)))

    @property
    def frames_in_flight(self):
        return self._depth

    @property
    def frame_slot(self):
        """index of the context the latest Sort ran on (0 .. frames_in_flight-1)"""
        return self._cur

    def sort_count(self):
        v = C.c_uint32()
        _capi.check(self._ctx, self._lib.msplat_sort_count(self._ctx, C.byref(v)))
        return v.value

    def sorted_indices(self):
        out = np.empty(max(self._n, 1), np.uint32)
        _capi.check(self._ctx, self._lib.msplat_get_sorted_indices(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                                   out.shape[0]))
        return out[:self.sort_count()].copy()

    def storage_order(self):
        """None when the cloud is stored in upload order, else the permutation slot -> upload index of the library's spatial
        storage order (msplat_config.spatial_order): equal depth keys are drawn in ascending storage slot"""
        ro = C.c_int(0)
        _capi.check(self._ctx, self._lib.msplat_get_storage_order(self._ctx, None, 0, C.byref(ro)))
        if not ro.value:
            return None
        out = np.empty(max(self._n, 1), np.uint32)
        _capi.check(self._ctx, self._lib.msplat_get_storage_order(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                                  out.shape[0], None))
        return out[:self._n]

    def cull_boxes(self):
        """(live, total, listed): bounding boxes that can hold a visible splat for the latest Sort's camera / boxes in the cloud
        ((0, 0) for a cloud in upload order); listed: that Sort's first pass walked only the listed live boxes"""
        a, b, l = C.c_uint32(), C.c_uint32(), C.c_int()
        _capi.check(self._ctx, self._lib.msplat_debug_get_cull_boxes(self._ctx, C.byref(a), C.byref(b), C.byref(l)))
        return a.value, b.value, bool(l.value)

    def sorted_keys(self):
        out = np.empty(max(self._n, 1), np.uint32)
        _capi.check(self._ctx, self._lib.msplat_get_sorted_keys(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                                out.shape[0]))
        return out[:self.sort_count()].copy()

    def stats(self):
        s = _capi.Stats()
        _capi.check(self._ctx, self._lib.msplat_get_stats(self._ctx, C.byref(s)))
        return {k: getattr(s, k) for k, _ in _capi.Stats._fields_}

    def timings(self):
        """stage times averaged over the sampled frames of every context (frames in flight)"""
        keys = ("sort_total", "render_total", "project", "binning", "composite")
        acc = {k: 0.0 for k in keys + ("composite_kernel",)}
        frames = 0.0
        for h in self._ctxs:
            t = _capi.Timings()
            _capi.check(h, self._lib.msplat_get_timings(h, C.byref(t)))
            nf = t.reserved[0]
            frames += nf
            for k in keys:
                acc[k] += getattr(t, k) * nf
            acc["composite_kernel"] += t.reserved[1] * nf    # exact dispatch begin/end of composite_kernel
        d = {k: (v / frames if frames > 0 else 0.0) for k, v in acc.items()}
        d["frames_averaged"] = frames
        return d

    def debug_projected(self):
        v = self.sort_count()
        rec = np.zeros((max(v, 1), 12), np.float32)
        rect = np.zeros(max(v, 1), np.uint32)
        _capi.check(self._ctx, self._lib.msplat_debug_get_projected(
            self._ctx, rec.ctypes.data_as(C.POINTER(C.c_float)), rect.ctypes.data_as(C.POINTER(C.c_uint32)), rec.shape[0]))
        return rec[:v], rect[:v]

    def set_tile_probe(self, enable=True):
        """per-work-item compositor counters for the following renders (off by default: a few clock reads per batch)"""
        for h in self._ctxs:
            _capi.check(h, self._lib.msplat_set_tile_probe(h, 1 if enable else 0))

    def composite_work(self):
        """what the compositor fetched and evaluated in the last render (needs set_tile_probe)"""
        w = _capi.CompositeWork()
        _capi.check(self._ctx, self._lib.msplat_get_composite_work(self._ctx, C.byref(w)))
        return {k: int(getattr(w, k)) for k, _ in _capi.CompositeWork._fields_}

    def two_pass_state(self, share=0.0):
        """(two-pass Renders so far, share of the visible splats the next one puts into its first pass) summed / taken over the
        contexts; share > 0 pins the share (msplat_debug_two_pass), 0 leaves it to the feedback loop"""
        frames, now = 0, 0.0
        for h in self._ctxs:
            a, b = C.c_uint64(), C.c_float()
            _capi.check(h, self._lib.msplat_debug_two_pass(h, float(share), C.byref(a), C.byref(b)))
            frames += a.value
            now = b.value
        return frames, now

    def two_pass_info(self):
        """what the latest two-pass Render of the current context did (msplat_get_two_pass_info), None if its latest Render ran in
        one pass"""
        out = (C.c_uint64 * 8)()
        _capi.check(self._ctx, self._lib.msplat_get_two_pass_info(self._ctx, out))
        if out[0] == 0:
            return None
        keys = ("frames", "splats_pass1", "splats_pass2", "pairs_pass1", "pairs_pass2", "bins_unfinished", "bins", "visible")
        return {k: int(out[i]) for i, k in enumerate(keys)}

    def verify_order(self):
        """on-device self-check: (violations of the sorted-key / tie order, violations of the bin-list order); (0, 0) = healthy"""
        a, b = C.c_uint32(), C.c_uint32()
        _capi.check(self._ctx, self._lib.msplat_debug_verify_order(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def cu_partitions(self):
        """per context: (MSPLAT_CU_* its stream got, the stream's CU mask as the runtime reports it: 8 words)"""
        out = []
        for h in self._ctxs:
            m = (C.c_uint32 * 8)()
            rc = self._lib.msplat_debug_cu_partition(h, m)
            _capi.check(h, rc if rc < 0 else 0)
            out.append((rc, list(m)))
        return out

    def debug_tile_probe(self):
        st = self.stats()
        nt = st["tiles_x"] * st["tiles_y"] * 8        # one slot per work item: (bin, quadrant[, half])
        out = np.zeros((max(nt, 1), 8), np.uint32)
        _capi.check(self._ctx, self._lib.msplat_debug_get_tile_probe8(
            self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint32)), out.shape[0]))
        return out[out[:, 7] > 0]

    def debug_tile_lists(self, want_pairs=True):
        st = self.stats()
        nt = st["tiles_x"] * st["tiles_y"]
        ts = np.zeros(nt + 1, np.uint32)
        if not want_pairs:          # only the list offsets
            _capi.check(self._ctx, self._lib.msplat_debug_get_tile_lists(
                self._ctx, ts.ctypes.data_as(C.POINTER(C.c_uint32)), ts.shape[0], None, 0))
            return ts, None
        pairs = np.zeros(max(int(st["pairs"]), 1), np.uint32)
        _capi.check(self._ctx, self._lib.msplat_debug_get_tile_lists(
            self._ctx, ts.ctypes.data_as(C.POINTER(C.c_uint32)), ts.shape[0],
            pairs.ctypes.data_as(C.POINTER(C.c_uint32)), pairs.shape[0]))
        return ts, pairs[:int(st["pairs"])]


class SplatRendererGroup:
    """Several GPUs, one process: the Python mirror of msplat_group_* (include/msplat.h).  Same Init / Sort / Render
    surface as SplatRenderer; the screen's bin rows are partitioned over `devices`, every device renders its rows, and
    with a device framebuffer (memory of devices[0]) the other devices' compositors write into it directly over xGMI."""

    def __init__(self, devices, fb_format="fp32", t_epsilon=-1.0, layout="contiguous", block_rows=1, band_cull=False,
                 enable_timing=False):
        self._lib = _capi.lib()
        self._g = None
        self._args = _FrameArgs()
        self._devices = [int(d) for d in devices]
        self._fb_format = {"fp32": _capi.FB_RGBA32F, "fp16": _capi.FB_RGBA16F}[fb_format]
        self._t_eps = t_epsilon
        self._layout, self._block_rows, self._band_cull = layout, int(block_rows), bool(band_cull)
        self._timing = enable_timing
        self._err = ""

    def __del__(self):
        self.close()

    def close(self):
        g, self._g = getattr(self, "_g", None), None
        if g:
            self._lib.msplat_group_destroy(g)

    def last_error(self):
        return self._lib.msplat_group_last_error(self._g).decode() if self._g else self._err

    def _check(self, rc):
        if rc == _capi.ERR_PAIR_OVERFLOW_EARLIER:
            import warnings
            warnings.warn(_capi.EarlierFrameOverflow(self._lib.msplat_group_last_error(self._g).decode()), stacklevel=3)
            return
        if rc != _capi.OK:
            raise _capi.MsplatError(rc, self._lib.msplat_group_last_error(self._g).decode())

    def Init(self, gaussianCloud, isFramebufferSRGBEnabled=False, useRgcSortOverride=False):
        del useRgcSortOverride
        self.close()
        cfg = _capi.Config()
        cfg.struct_size = C.sizeof(_capi.Config)
        cfg.fb_format = self._fb_format
        cfg.srgb = 1 if isFramebufferSRGBEnabled else 0
        cfg.t_epsilon = self._t_eps
        cfg.enable_timing = int(self._timing)
        devs = (C.c_int32 * len(self._devices))(*self._devices)
        g = C.c_void_p()
        rc = self._lib.msplat_group_create(C.byref(g), devs, len(self._devices), C.byref(cfg))
        if rc != _capi.OK:
            self._err = self._lib.msplat_group_last_error(None).decode()
            return False
        self._g = g
        self._check(self._lib.msplat_group_set_layout(g, _capi.BAND_KINDS[self._layout], self._block_rows))
        self._check(self._lib.msplat_group_set_band_cull(g, 1 if self._band_cull else 0))
        if isinstance(gaussianCloud, GaussianCloud):
            rc = self._lib.msplat_group_upload_gaussian_cloud(g, gaussianCloud.handle)
        else:
            aos = np.ascontiguousarray(gaussianCloud, np.float32)
            assert aos.shape[1] in (25, 61)
            off = _capi.AttrOffsets(0, 16, 32, 48, 64, 76, 88, 100, 116, 132, 148, 164, 180, 196, 212, 228)
            rc = self._lib.msplat_group_upload_cloud(g, aos.ctypes.data, aos.shape[0], aos.shape[1] * 4, C.byref(off),
                                                     1 if aos.shape[1] == 61 else 0)
        if rc != _capi.OK:
            self._err = self._lib.msplat_group_last_error(g).decode()
            return False
        return True

    @property
    def size(self):
        return int(self._lib.msplat_group_size(self._g))

    def peer_store(self, i):
        return bool(self._lib.msplat_group_peer_store(self._g, i))

    def context(self, i):
        """borrowed msplat_ctx handle of rank i (for the C-ABI getters)"""
        return C.c_void_p(self._lib.msplat_group_context(self._g, i))

    def sort_count(self, i):
        v = C.c_uint32()
        h = self.context(i)
        _capi.check(h, self._lib.msplat_sort_count(h, C.byref(v)))
        return v.value

    def Sort(self, cameraMat, projMat, viewport, nearFar):
        c, p, v, nf = self._args.load(cameraMat, projMat, viewport, nearFar)
        self._check(self._lib.msplat_group_sort(self._g, c, p, v, nf))

    def Render(self, cameraMat, projMat, viewport, nearFar, out=None, out_ptr=None, pitch_bytes=0):
        """out_ptr: device pointer ON devices[0] (asynchronous; synchronize() or wait on context 0's stream);
        otherwise a host array is filled / returned"""
        c, p, v, nf = self._args.load(cameraMat, projMat, viewport, nearFar)
        vp = self._args.vp
        if out_ptr is not None:
            self._check(self._lib.msplat_group_render(self._g, c, p, v, nf, C.c_void_p(out_ptr), pitch_bytes, 1))
            return None
        W, H = int(vp[2]), int(vp[3])
        dt = np.float16 if self._fb_format == _capi.FB_RGBA16F else np.float32
        if out is None:
            out = np.zeros((H, W, 4), dt)
        assert out.dtype == dt and out.shape == (H, W, 4) and out.flags["C_CONTIGUOUS"]
        self._check(self._lib.msplat_group_render(self._g, c, p, v, nf, out.ctypes.data, 0, 0))
        return out

    def synchronize(self):
        self._check(self._lib.msplat_group_synchronize(self._g))

    EXCHANGES = ("peer_store", "rccl", "copy")      # MSPLAT_EXCHANGE_*

    def set_exchange(self, name):
        """how the other devices' rows reach devices[0]'s framebuffer: "peer_store" (default), "rccl" (ncclSend / ncclRecv over
        communicators from ncclCommInitAll) or "copy" (hipMemcpy2DAsync per run)"""
        self._check(self._lib.msplat_group_set_exchange(self._g, self.EXCHANGES.index(name)))

    def exchange(self):
        """the exchange the latest device-output Render used"""
        return self.EXCHANGES[self._lib.msplat_group_get_exchange(self._g)]
