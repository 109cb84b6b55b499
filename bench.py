#!/usr/bin/env python
"""bench.py -- frames/s (+ Gsplats/s) of the splat Sort+Render hot path on N MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one frame = SplatRenderer::Sort + SplatRenderer::Render of the resident cloud from a
camera on a 64-step orbit (stereo workloads: one Sort + two Renders).  With N > 1 the frame's bin
rows are sharded across the ranks (--layout: contiguous | interleaved | block:k | weighted[:percent] | auto = blocks of
rows / (2 N), the best of profiles/r03_cfg4_bands.json) and gathered to rank 0 over RCCL: total work
is fixed, so scaling is "strong".  Rank 0 prints ONE JSON line.

What is measured:
  value / ms_per_step ... EXACTLY `--steps` frames between barrier + synchronize on both sides, `--frames-in-flight`
                          frames overlapped on the GPU (default 4).  The block is repeated until >= 2 s of frames
                          have been timed and the MEDIAN block is reported (a 20-frame block lasts 3.3 ms).
  serial ................ the same frames strictly one after the other on ONE stream (a second renderer, 8192 compositor
                          waves): single-frame latency, and every kernel has the GPU to itself, so its launch duration
                          is a clean per-kernel number.  `roofline` is computed from THIS phase; the rocprofv3 summary
                          to compare it with is the one of `bench.py --frames-in-flight 1` (profiles/).
  roofline .............. dominant kernel (composite_kernel): bytes it fetches under its front-to-back early-termination
                          contract (4 B per list entry + 48 B per projected record actually loaded, counted by the
                          kernel's own probe on extra frames, + the framebuffer write) / serial launch duration vs
                          8 TB/s; beside it the SURVEY 8d formula bytes (52 D + W H bpp), the PMC traffic, and the
                          VALU view -- (pixel, splat) evaluations/s, the flops the kernel EXECUTES for them (14.5 + 1
                          v_exp_f32 per evaluation, counted in the ISA) as a share of the 157.3 TFLOP/s fp32 vector peak
                          (the nominal 20 flop of SURVEY 8d beside it), and the share of the evaluations whose weight
                          survives the discard (useful_eval_frac) -- because VALU, not HBM, bounds this kernel.
  cpu_baseline .......... the tiled CPU renderer of SURVEY 8d(ii) (oracle/msplat_cpu_tiled.c, kind "port-tiled": parallel
                          radix sort, tile binning, front-to-back with early termination; thread count from the cgroup
                          CPU quota) on a bounded sample of the same workload; rank 0, N = 1 only.
  cpu_baseline_literal .. the literal oracle (C restatement of the reference shaders, every OpenMP row band walks all
                          visible splats back to front), one frame.

PyTorch is plumbing only (device selection, the framebuffer tensor, torch.distributed); all compute
is libmsplat.so's HIP kernels.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Frames in flight run on one HIP stream each; the ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES
# hardware queues (default 4) and streams that share a queue serialise.  With torch's own streams in the
# process, 4 frames in flight need more than 4 queues (measured: 3.8 k fps with 4 queues, 5.1 k with 8).
# Must be set before the HIP runtime initialises, i.e. before torch is imported.
# N > 1: RCCL brings its own streams; 16 queues keep the frame streams from sharing one with them (a shared queue costs a
# quarter of the frame rate; 8 / 12 / 16 / 24 queues measure the same on one GPU).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "8")

WORKLOADS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "cfg2": dict(n=1_000_000, seed=0x5EED1234, pos_sigma=1.5, W=1920, H=1080, cam_z=7.0, fb="fp32", views=1,
                 desc="1M synthetic Gaussians, SH3, 1920x1080 fp32, 64-step orbit (BASELINE configs[1])"),
    # configs[2] fallback (no Inria scene on the box)
    "cfg3": dict(n=6_000_000, seed=0x5EED6000, pos_sigma=3.0, W=1920, H=1080, cam_z=12.0, fb="fp32", views=1,
                 desc="6M synthetic Gaussians, SH3, 1920x1080 fp32 (BASELINE configs[2] fallback)"),
    "cfg4": dict(n=6_000_000, seed=0x5EED6000, pos_sigma=3.0, W=4096, H=4096, cam_z=12.0, fb="fp32", views=1,
                 desc="6M synthetic Gaussians, SH3, 4096x4096 fp32 (BASELINE configs[3])"),
    "cfg5": dict(n=1_000_000, seed=0x5EED1234, pos_sigma=1.5, W=2016, H=2240, cam_z=7.0, fb="fp16", views=2,
                 desc="1M synthetic Gaussians, stereo 2x2016x2240 fp16, one sort (BASELINE configs[4])"),
    # configs[2] again, scene-LIKE (r3): surface-concentrated positions, heavy-tailed anisotropic scales, 1 % background splats
    # that span a quarter of the view, bimodal opacity, cameras INSIDE the cloud; written as an Inria-style PLY + cameras.json and
    # replayed through the file path (Ply::Parse -> GPU ingest, CamerasConfig::ImportJson: camerasconfig.cpp:20-67)
    "cfg3s": dict(n=6_000_000, seed=0x5CE11E, scene=True, W=1920, H=1080, cam_z=0.0, fb="fp32", views=1,
                  desc="6M scene-like synthetic splats (surfaces, log-scale sigma 1.2, 1% background, camera inside), SH3, "
                       "1920x1080 fp32, PLY + cameras.json replay (BASELINE configs[2] stand-in)"),
    # experiment (r6): BASELINE configs[1] drawn as two views of one Sort -- what a chain shared by two frames' renders buys
    "cfg2v2": dict(n=1_000_000, seed=0x5EED1234, pos_sigma=1.5, W=1920, H=1080, cam_z=7.0, fb="fp32", views=2,
                   desc="1M synthetic Gaussians, SH3, two 1920x1080 fp32 views of one sort (experiment)"),
    "tiny": dict(n=20_000, seed=7, pos_sigma=1.5, W=640, H=360, cam_z=7.0, fb="fp32", views=1,
                 desc="20k synthetic Gaussians, 640x360 (debug)"),
    "tinys": dict(n=60_000, seed=0x5CE11E, scene=True, W=640, H=360, cam_z=0.0, fb="fp32", views=1,
                  desc="60k scene-like synthetic splats, 640x360, PLY + cameras.json replay (debug)"),
}
HBM_PEAK = 8.0e12        # B/s, MI355X_MICROARCH.md
VALU_PEAK = 157.3e12     # fp32 vector FLOP/s, MI355X_MICROARCH.md
FLOP_PER_EVAL = 20.0     # ~20 flop + 1 transcendental per (pixel, splat): the NOMINAL figure
# what composite_kernel's inner loop executes per (pixel, record) evaluation, counted in its ISA (msplat_composite.hip.h: per
# record and lane, i.e. per 4 evaluations: 3 scalar FMAs + per strip pair 2 packed FMAs for e, 1 packed multiply, 4 packed FMAs
# = 58 flop, and 4 v_exp_f32): 14.5 flop + 1 transcendental
FLOP_PER_EVAL_EXECUTED = 14.5
MIN_TIMED_SECONDS = 2.0  # r6: the timed GPU phase lasts long enough for an outside observer (the driver's SMI sampler) to see it
MAX_BLOCKS = 4000


class Env:
    """process-wide state shared by the measurements of one bench invocation"""
    _comm = None

    def rccl_comm(self):
        """this process's ncclComm_t for msplat_band_exchange (ncclCommInitRank is a collective: every rank calls this at the
        same point); made once, the unique id travels over the gloo side group"""
        if self._comm is None:
            from splatapult_amd.dist import RcclComm
            self._comm = RcclComm(self.rank, self.world, self.local_rank, group=self.cpu_group)
        return self._comm


_T0 = time.perf_counter()


def phase(msg):
    """MSPLAT_BENCH_VERBOSE=1: where an invocation's wall time goes (stderr)"""
    if os.environ.get("MSPLAT_BENCH_VERBOSE") == "1":
        print("[bench %8.2f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def hbm_delivered(dev):
    """what this box's HBM delivers to a plain copy, measured in this process after the timed region (r6): `roofline.peak` has to quote
    the nominal 8 TB/s; a 1-GiB device-to-device copy (2 GiB moved, beyond the 256 MB of MALL) says what a stream gets here.  Context
    for the fractions, never part of `value`.  (tools/ubench_stream.hip: plain reads 5.2-5.75 TB/s, copies 4.5-4.9 TB/s on this pool.)"""
    import torch
    try:
        n = 1 << 28
        src = torch.zeros(n, dtype=torch.float32, device=dev)
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            dst.copy_(src)
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b))
        del src, dst
        ms = float(np.median(ts))
        return {"copy_GBps": 2.0 * n * 4 / (ms * 1e-3) / 1e9, "copy_frac_of_peak": 2.0 * n * 4 / (ms * 1e-3) / HBM_PEAK,
                "how": "torch Tensor.copy_ of 1 GiB on this device (2 GiB moved), median of 10, after the timed region"}
    except Exception as e:      # noqa: BLE001 -- context only
        return {"copy_GBps": None, "how": "not measured: %s" % e}


def measure(E, args, key, ply=None, primary=True):
    """one workload on the current process group; returns the JSON dict (rank 0 uses it)"""
    import torch
    import torch.distributed as dist
    from splatapult_amd import SplatRenderer, camera, synthetic, _capi

    rank, world, dev, stream = E.rank, E.world, E.dev, E.stream
    wl = dict(WORKLOADS[key])
    wl["key"] = key
    W, H, views = wl["W"], wl["H"], wl["views"]
    scene_cams = None
    scene_dir = None
    if wl.get("scene") and not ply:
        # the scene-like workload goes through the FILE path: attributes -> PLY + cameras.json in a scratch directory
        import tempfile
        scene_dir = tempfile.mkdtemp(prefix="msplat_scene_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        ply = os.path.join(scene_dir, "point_cloud", "iteration_30000", "point_cloud.ply")
        # every rank writes its own copy (ranks do not share the scratch directory)
        os.makedirs(os.path.dirname(ply))
        synthetic.write_ply(ply, synthetic.generate_scene(wl["n"], seed=wl["seed"]))
        synthetic.write_cameras_json(os.path.join(scene_dir, "cameras.json"), synthetic.scene_cameras(64), W, H, camera.FOVY)
    from_file = ply is not None
    if ply:
        from splatapult_amd import GaussianCloud
        cloud = GaussianCloud()
        if not cloud.ImportPly(ply):
            raise SystemExit("cannot import " + ply)
        wl["n"] = cloud.GetNumGaussians()
        cj = camera.find_config_file(ply, "cameras.json")          # app.cpp:418-461
        if cj:
            scene_cams = [m for m, _ in camera.load_cameras_json(cj)]
        if not wl.get("scene"):
            # a real scene (BASELINE configs[2]: the Inria PLY): the line names the file and where the cameras came from
            # (camerasconfig.cpp:20-67 when a cameras.json was found next to / above it, else the synthetic orbit)
            wl["desc"] = "%s (%d splats, SH%d), %dx%d %s, cameras: %s" % (
                os.path.abspath(ply), wl["n"], 3 if cloud.HasFullSH() else 0, W, H, wl["fb"],
                ("%s (%d poses, CamerasConfig::ImportJson)" % (os.path.abspath(cj), len(scene_cams))) if cj else "64-step orbit at z = %g (no cameras.json found)" % wl["cam_z"])
    else:
        cloud = synthetic.make_cloud(wl["n"], seed=wl["seed"], full_sh=True, pos_sigma=wl["pos_sigma"])
    n = wl["n"]

    P = max(1, args.frames_in_flight)
    two_pass = {"auto": _capi.TWO_PASS_AUTO, "on": _capi.TWO_PASS_ON, "off": _capi.TWO_PASS_OFF}[args.two_pass]
    r = SplatRenderer(device=E.local_rank, fb_format=wl["fb"], stream=stream.cuda_stream,
                      enable_timing=args.timing_stride, frames_in_flight=P,
                      async_submit=None if args.async_submit < 0 else bool(args.async_submit), two_pass=two_pass,
                      compositor_waves=args.compositor_waves, cu_partition=None if args.cu_partition == "auto" else False)

    def init(rr):
        # a file is rendered the way the app would: Ply::Parse on the host + GaussianCloud::ImportPly's math on the GPU
        ok = rr.InitFromPly(ply, True, False) if from_file else rr.Init(cloud, False, False)
        if not ok:
            raise SystemExit("Init failed: " + rr.last_error())
        if args.two_pass_share > 0.0:
            rr.two_pass_state(args.two_pass_share)

    init(r)
    phase("cloud uploaded")
    pair_cap0 = int(max(4 << 20, 32 * n))           # the library's initial (splat, bin) pair capacity: max(4 M, 32 N)
    TILE = _capi.lib().msplat_tile_size()
    tiles_y = (H + TILE - 1) // TILE
    # bin-row layout over the ranks (msplat_band_plan): "contiguous" | "interleaved" | "block:k"; auto = blocks of
    # ~rows / (2 ranks) rows dealt round-robin -- two blocks per rank balance the load of a centred scene while the
    # band-restricted cull still drops most of the splats of the other ranks' rows (measured on the 6 M / 4096^2 workload, 8
    # ranks: blocks of 8 rows 0.387 ms per rank, contiguous bands 0.417, interleaved rows 0.528: profiles/r03_cfg4_bands.json)
    lay = args.layout
    lay_model = None
    bpp_ = 8 if wl["fb"] == "fp16" else 16
    if world > 1 and lay in ("auto", "weighted") and views == 1:
        # r6: where the row gather bounds the frame (BASELINE configs[3]: 2 MiB per bin row over a 153 GB/s link against ~6 us of
        # compute per row) equal bands make N = 2 SLOWER than one GPU.  Contiguous bands with rank 0 -- the gather's root, which
        # sends nothing -- weighted by msplat_band_root_weight's linear cost model, calibrated here on rank 0's GPU with the
        # timed protocol's own frames in flight: the whole image, and one equal band in the middle of it.
        cal_proj = camera.perspective(camera.FOVY, W / H)

        def _orbit_pose(_wl, k):
            return scene_cams[k % len(scene_cams)] if scene_cams else camera.orbit(_wl["cam_z"], 2.0 * math.pi * (k % 64) / 64.0)

        def _proj0(_wl, _W, _H):
            return cal_proj

        def ms_per_frame(nframes=3 * P + 8):
            for s_ in range(P + 2):
                cs = _orbit_pose(wl, s_)
                r.Sort(cs, _proj0(wl, W, H), [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR])
                r.Render(cs, _proj0(wl, W, H), [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR], out_ptr=cal_fb[r.frame_slot].data_ptr(), pitch_bytes=W * bpp_)
            r.synchronize()
            t0_ = time.perf_counter()
            for s_ in range(nframes):
                cs = _orbit_pose(wl, 7 + s_)
                r.Sort(cs, _proj0(wl, W, H), [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR])
                r.Render(cs, _proj0(wl, W, H), [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR], out_ptr=cal_fb[r.frame_slot].data_ptr(), pitch_bytes=W * bpp_)
            r.synchronize()
            return 1e3 * (time.perf_counter() - t0_) / nframes
        cal_fb = [torch.zeros((tiles_y * TILE, W, 4), dtype=torch.float16 if bpp_ == 8 else torch.float32, device=dev) for _ in range(P)]
        for _ in range(2):
            ms_per_frame(40)                              # (runtime warm-up)
        rows_eq = max(1, tiles_y // world)
        r.set_band(1, 0)
        t_full = ms_per_frame()
        r.set_band_layout((tiles_y - rows_eq) // 2, rows_eq, rows_eq, max(tiles_y, 1), band_cull=(views == 1))
        t_part = ms_per_frame()
        r.set_band(1, 0)
        del cal_fb
        per_row = max(1e-6, (t_full - t_part) / max(1, tiles_y - rows_eq))
        fixed = max(0.0, t_part - per_row * rows_eq)
        cal = torch.tensor([fixed, per_row], dtype=torch.float64, device=dev)
        dist.broadcast(cal, src=0)                        # every rank plans with rank 0's numbers
        fixed, per_row = float(cal[0].item()), float(cal[1].item())
        pct = _capi.band_root_weight(tiles_y, world, fixed, per_row, TILE * W * bpp_ * views, 153.0, True)
        lay_model = {"fixed_ms": fixed, "ms_per_bin_row": per_row, "row_bytes": TILE * W * bpp_ * views, "link_GBps": 153.0,
                     "root_weight_percent": pct, "calibrated_on": "rank 0, %d frames in flight, whole image %.4f ms / an equal band of %d rows %.4f ms per frame" % (P, t_full, rows_eq, t_part)}
        # auto: the weighted bands only where the model says the link binds (root weight well above an equal share); else r3's blocks
        lay = "weighted:%d" % pct if (lay == "weighted" or pct >= 125) else "block:%d" % max(1, tiles_y // (2 * world))
    elif lay in ("auto", "weighted"):
        lay = "block:%d" % max(1, tiles_y // (2 * world))
    lay_kind, lay_k = (lay.split(":")[0], int(lay.split(":")[1])) if ":" in lay else (lay, 1)
    if lay_kind == "block" and lay_k == 1:
        lay_kind = "interleaved"
    if lay_kind == "weighted" and ":" not in lay:
        lay_k = 100
    if world > 1:
        # mono workloads may also restrict the cull to the band (every Render uses its Sort's camera)
        r.set_band_plan(lay_kind, tiles_y, world, rank, block_rows=lay_k, band_cull=(views == 1))

    Hpad = tiles_y * TILE
    fdt = torch.float16 if wl["fb"] == "fp16" else torch.float32
    bpp = 8 if wl["fb"] == "fp16" else 16
    fb_sets = [[torch.zeros((Hpad, W, 4), dtype=fdt, device=dev) for _ in range(views)] for _ in range(P)]
    fb_free = [None] * P      # N > 1: event recorded after the gather that read / filled the slot's framebuffers
    vp, nf = [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR]
    if views == 1:
        projs = [camera.perspective(camera.FOVY, W / H)]
    elif key == "cfg2v2":
        projs = [camera.perspective(camera.FOVY, W / H)] * 2
    else:   # BASELINE config 5: asymmetric XR frusta (util.cpp:420-480)
        projs = [camera.create_projection(-1.0, 0.8, 0.95, -0.95), camera.create_projection(-0.8, 1.0, 0.95, -0.95)]

    pose_cache = {}      # the orbit has 64 poses: built once (a pose costs the host 8 us, a frame's launches ~40)

    def cams_for(step):
        k = step % (len(scene_cams) if scene_cams else 64)
        cs = pose_cache.get(k)
        if cs is None:
            c = scene_cams[k] if scene_cams else camera.orbit(wl["cam_z"], 2.0 * math.pi * k / 64.0)
            cs = [c] if views == 1 else [camera.translate_local(c, dx=-0.032), camera.translate_local(c, dx=+0.032)]
            pose_cache[k] = cs
        return cs

    # the only exchange step: every rank's bin rows go straight into rank 0's framebuffer.  r6: under RCCL the timed frames use the
    # PRODUCT's exchange -- msplat_band_exchange (C ABI: one ncclGroupStart / ncclRecv | ncclSend per run / ncclGroupEnd issued by
    # libmsplat on the context's stream, behind the frame's compositor) with a communicator of the bench's own; torch.distributed's
    # batch_isend_irecv (splatapult_amd/dist.py, BandGather) is the labelled fallback (--exchange torch, gloo, or no communicator)
    gathers = None
    comm = None
    exchange_note = None
    if world > 1:
        from splatapult_amd.dist import BandGather, CAbiBandGather
        gathers = [BandGather(tiles_y, W, fdt, dev, rank, world, tile=TILE, layout=lay_kind, block_rows=lay_k) for _ in range(views)]
        want_cabi = args.exchange == "cabi" or (args.exchange == "auto" and dist.get_backend() == "nccl" and not E.one_dev)
        if want_cabi:
            ok, why = 1, ""
            try:
                comm = E.rccl_comm()
            except Exception as e:      # noqa: BLE001 -- reported in the line, the torch gather takes over
                ok, why = 0, "%s: %s" % (type(e).__name__, e)
            t_ok = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN, group=E.cpu_group)
            if int(t_ok.item()) == 0:
                comm = None
                exchange_note = "msplat_band_exchange unavailable on some rank (%s): torch.distributed gather used" % (why or "another rank")
    cabi_gathers = {}

    def gathers_of(rr):
        """the exchange objects of renderer rr (the C-ABI exchange runs on rr's own streams)"""
        if gathers is None or comm is None:
            return gathers
        g = cabi_gathers.get(id(rr))
        if g is None:
            g = cabi_gathers[id(rr)] = [CAbiBandGather(rr, comm, tiles_y, W, fdt, rank, world, tile=TILE, layout=lay_kind, block_rows=lay_k,
                                                       wire_fp16=(os.environ.get("MSPLAT_BENCH_WIRE_FP16") == "1"))
                                        for _ in range(views)]
        return g
    state = {"fbs": fb_sets[0]}
    stereo_batch = views == 2 and gathers is None and not args.no_stereo_batch       # both eyes in one chain of launches
    launches_per_frame = 1 if stereo_batch else views                                 # compositor launches (and render chains) per frame

    def frame(step, rr=r, sets=fb_sets):
        cams = cams_for(step)
        Pn = len(sets)
        if Pn > 1 and gathers is not None and comm is None:
            ev = fb_free[(rr.frame_slot + 1) % Pn]
            if ev is not None:
                rr.next_frame_wait_event(ev.cuda_event)         # the slot's previous frame has been gathered
        rr.Sort(cams[0], projs[0], vp, nf)                      # sort once with view 0 (app.cpp:603-606)
        fbs = sets[rr.frame_slot % Pn]
        state["fbs"] = fbs
        if stereo_batch:
            # both eyes in one chain of launches (msplat_render_stereo): same pixels as the two Render calls below
            rr.RenderStereo(cams, projs, vp, nf, out_ptrs=[fbs[0].data_ptr(), fbs[1].data_ptr()], pitch_bytes=W * bpp)
            return
        gs = gathers_of(rr)
        for v in range(views):
            rr.Render(cams[v], projs[v], vp, nf, out_ptr=fbs[v].data_ptr(), pitch_bytes=W * bpp)
            if gs is not None:
                if comm is not None:
                    gs[v](fbs[v])                               # msplat_band_exchange: on the context's own stream, behind its compositor
                else:
                    if Pn > 1:
                        rr.wait_on_stream(stream.cuda_stream)   # the gather's stream waits for this frame only
                    gs[v](fbs[v])
        if Pn > 1 and gs is not None and comm is None:
            ev = torch.cuda.Event()
            ev.record(stream)
            fb_free[rr.frame_slot] = ev

    live = [r]            # renderers whose queued calls (async_submit: a worker thread per in-flight context) must be ISSUED
                          # before a device synchronisation means "the frames are done"

    def sync_all():
        for x in live:
            x.synchronize()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # The HIP runtime grows its per-queue kernarg / signal pools once, a few thousand launches after start-up
    # (a single 15-40 ms stall around frame 170-210): get past it before the official warm-up
    phase("layout %s; prewarm" % lay)
    for s in range(args.prewarm):
        frame(s)
    sync_all()
    phase("prewarm done")
    for s in range(args.warmup):
        frame(s)

    # ---- the timed region: blocks of EXACTLY --steps frames, barrier + synchronize on both sides ----
    def timed_block(first_step):
        sync_all()
        t0 = time.perf_counter()
        for s in range(args.steps):
            frame(first_step + s)
        enq = time.perf_counter() - t0            # host time to hand over the frames (the launches are issued by worker threads)
        sync_all()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())                  # identical on every rank: the block count below agrees too
        return el, enq

    blocks, enqs = [], []
    el, enq = timed_block(args.warmup)
    blocks.append(el); enqs.append(enq)
    # (MSPLAT_BENCH_ONE_DEVICE: N processes time-share ONE GPU over gloo -- a control-flow check whose blocks escalate to 2.5 s each once
    #  the runtime juggles both processes' queues; it is not a measurement and gets a short timed phase)
    min_s, max_b = (0.25, 12) if E.one_dev else (MIN_TIMED_SECONDS, MAX_BLOCKS)
    nblocks = int(min(max_b, max(1, math.ceil(min_s / max(el, 1e-6)))))
    for b in range(1, nblocks):
        el, enq = timed_block(args.warmup + b * args.steps)
        blocks.append(el); enqs.append(enq)
    elapsed = float(np.median(blocks))
    enqueue = float(np.median(enqs))
    phase("timed region done: %d blocks, first %.4f s, median %.4f s" % (len(blocks), blocks[0], elapsed))
    prof = dict(sort_total=0.0, render_total=0.0, project=0.0, binning=0.0, composite=0.0, composite_kernel=0.0)
    if args.timing_stride > 0:
        prof = r.timings()                        # sampled stage events of the overlapped frames
    tp_flight = r.two_pass_info()                 # the latest frame of the timed region (None: one pass)
    tp_frames_flight, tp_share_flight = r.two_pass_state(args.two_pass_share)

    # ---- serial phase: the same frames one at a time on ONE stream (clean per-kernel durations, latency) ----
    if P == 1:
        rs, rs_sets = r, fb_sets
    else:
        rs = SplatRenderer(device=E.local_rank, fb_format=wl["fb"], stream=stream.cuda_stream, enable_timing=4,
                           frames_in_flight=1, two_pass=two_pass)     # stage events on every 4th frame: they cost a few us each
        init(rs)
        live.append(rs)
        if world > 1:
            rs.set_band_plan(lay_kind, tiles_y, world, rank, block_rows=lay_k, band_cull=(views == 1))
        rs_sets = [fb_sets[0]]
    for s in range(24):
        frame(s, rs, rs_sets)
    sync_all()
    if args.timing_stride > 0 or P > 1:
        rs.timings()                              # drop the warm-up samples
    SER = args.serial_frames
    t0 = time.perf_counter()
    for s in range(SER):
        frame(args.warmup + s, rs, rs_sets)
    sync_all()
    serial_ms = 1e3 * (time.perf_counter() - t0) / SER
    prof_serial = rs.timings() if (args.timing_stride > 0 or P > 1) else None
    tp_serial = rs.two_pass_info()                 # None: the serial frames ran in one pass
    tp_share_serial = rs.two_pass_state(args.two_pass_share)[1]
    lat = []
    for s in range(16):
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        frame(args.warmup + SER + s, rs, rs_sets)
        torch.cuda.synchronize(dev)
        lat.append(time.perf_counter() - t1)
    latency_ms = 1e3 * float(np.median(lat))
    phase("serial phase done")

    # ---- statistics frames (outside every timed region; each read synchronises) ----
    Vs, Ds, drawn, Dbin, works = [], [], [], [], []
    rs.set_tile_probe(True)
    for s in range(max(1, args.profile_frames)):
        # (no gather here: only the compositor's own counters are wanted; every view's launch is probed)
        cams = cams_for(args.warmup + s * 7)
        rs.Sort(cams[0], projs[0], vp, nf)
        if stereo_batch:                   # ONE compositor launch covers both eyes
            rs.RenderStereo(cams, projs, vp, nf, out_ptrs=[t.data_ptr() for t in rs_sets[0]], pitch_bytes=W * bpp)
            works.append(rs.composite_work())
        else:
            for v in range(views):
                rs.Render(cams[v], projs[v], vp, nf, out_ptr=rs_sets[0][v].data_ptr(), pitch_bytes=W * bpp)
                works.append(rs.composite_work())
        st = rs.stats()
        Vs.append(st["sort_count"]); Ds.append(st["pairs_tile16"]); drawn.append(st["drawn"]); Dbin.append(st["pairs"])
    rs.set_tile_probe(False)
    ts_last, _ = rs.debug_tile_lists(want_pairs=False)
    longest_list = int(np.diff(ts_last.astype(np.int64)).max()) if ts_last.shape[0] > 1 else 0
    pair_cap_end = int(rs.stats()["pair_capacity"])
    V, D = float(np.mean(Vs)), float(np.mean(Ds))
    work = {k: float(np.mean([w[k] for w in works])) for k in works[0]} if works else None
    if world > 1:
        t = torch.tensor([V, D], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        D_total = float(t[1].item())
    else:
        D_total = D

    # ---- two-pass frames: are they THE frames?  (outside every timed region; only when the timed frames ran in two passes) ----
    tp_check = None
    if views == 1 and (tp_frames_flight > 0 or tp_serial is not None):
        tp_check = two_pass_check(E, wl, init, cams_for, projs, vp, nf, W, Hpad, bpp, fdt,
                                  tp_share_flight if tp_share_flight > 0.0 else 0.15,
                                  (lay_kind, tiles_y, world, rank, lay_k) if world > 1 else None)

    phase("statistics / two-pass check done")
    # ---- N > 1: is rank 0's gathered frame THE frame?  (outside every timed region) ----
    gcheck = gather_check(E, args, wl, init, frame, rs, rs_sets, cams_for, projs, vp, nf, cloud, lay_kind, lay_k, primary, comm) \
        if (world > 1 and gathers is not None) else None

    phase("gather check done")
    fps = args.steps / elapsed
    ms = 1e3 * elapsed / args.steps
    # algorithmic bytes of the whole frame:  B = 16 N + (8+68+S+48) V + views (52 D + W H bpp)
    S = 244
    B_frame = 16.0 * n + (8 + 68 + S + 48) * V + (52.0 * D_total + W * H * bpp) * views
    Dbin_mean = float(np.mean(Dbin))
    # dominant kernel: composite_kernel, one launch per view, measured with the GPU to itself (serial phase)
    fb_bytes = (W * H * bpp) / world * (views / launches_per_frame)
    B_formula = 52.0 * D * (views / launches_per_frame if views > 1 else 1.0) + fb_bytes      # every (splat, 16x16 tile) pair fetched
    comp_serial_ms = (prof_serial or {}).get("composite_kernel", 0.0) or (prof_serial or {}).get("composite", 0.0)
    comp_overlap_ms = prof.get("composite_kernel", 0.0) or prof.get("composite", 0.0)
    if work:
        B_fetched = 4.0 * work["pair_words_fetched"] + 48.0 * work["records_fetched"] + fb_bytes
    else:
        B_fetched = None
    B_used = B_fetched if B_fetched is not None else B_formula
    comp_s = comp_serial_ms * 1e-3
    achieved = B_used / comp_s if comp_s > 0 else 0.0
    # HBM traffic of the dominant kernel comes from a SEPARATE rocprofv3 --pmc run of `bench.py --frames-in-flight 1`
    # (tools/pmc_traffic.sh: FETCH_SIZE / WRITE_SIZE in their own passes, gfx950 x2 correction on FETCH_SIZE);
    # the committed summary is only quoted for the workload it was measured on
    traffic, tsrc = None, None
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        tpath = os.path.join(ROOT, "profiles", "%s_pmc_traffic_%s.json" % (rnd, key))
        if world == 1 and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # template arguments (r5): <RGBA16F target, pass of a two-pass frame (0 = one pass), two views in one chain, probe>
                f16 = "true" if wl["fb"] == "fp16" else "false"
                if rnd < "r05":
                    names = [[k for k in tj if k.startswith("msplat::composite_kernel<%s" % f16)][0]]
                elif tp_serial:
                    names = ["msplat::composite_kernel<%s, %d, false, false>" % (f16, q) for q in (1, 2)]      # both launches of the frame
                else:
                    names = ["msplat::composite_kernel<%s, 0, %s, false>" % (f16, "true" if stereo_batch else "false")]
                traffic = sum(tj[k]["hbm_bytes_per_launch_corrected"] for k in names)
                tsrc = "profiles/%s (rocprofv3 --pmc of bench.py --frames-in-flight 1, bytes per launch)" % os.path.basename(tpath)
                break
            except (KeyError, ValueError, IndexError):
                traffic = None
    # per-stage view of the serial frame (every kernel alone on the GPU): bytes the stage has to move by its own design --
    # SURVEY 8d's terms for sort (16 N + 8 V + 68 V) and projection ((S + 48) V), 4 V + 16 D32 for the two binning partitions
    # (rectangle read; every (splat, 32-px bin) pair word written, read, written, read), and for the compositor what its
    # front-to-back walk really fetched (probe) + the framebuffer -- over the stage's time; none of them can exceed 1
    stages = None
    vpl = views / launches_per_frame           # views per render chain (2 when both eyes run as one chain)
    if prof_serial and prof_serial.get("sort_total", 0) > 0:
        # (stereo: projection, binning and the compositor run once per view; their stage times and bytes are per Render call)
        sb = {"sort": 16.0 * n + 76.0 * V, "project": vpl * (S + 48.0) * V, "binning": vpl * 4.0 * V + 16.0 * Dbin_mean, "composite": B_used}
        if tp_serial:
            # two-pass frames: pass 1 projects its share, the gate reads 20 B per splat behind the cut (index + centre) and writes
            # / rewrites rectangles, pass 2 projects what passes it; both binning chains read every rectangle and move their own pairs
            sb["project"] = (S + 48.0) * (tp_serial["splats_pass1"] + tp_serial["splats_pass2"]) + 28.0 * tp_serial["visible"]
            sb["binning"] = 2 * 4.0 * tp_serial["visible"] + 16.0 * (tp_serial["pairs_pass1"] + tp_serial["pairs_pass2"])
        st_ms = {"sort": prof_serial["sort_total"], "project": prof_serial["project"], "binning": prof_serial["binning"],
                 "composite": comp_serial_ms}
        stages = {k: {"bytes": sb[k], "us": 1e3 * st_ms[k], "frac": min(1.0, sb[k] / max(st_ms[k] * 1e-3, 1e-12) / HBM_PEAK)}
                  for k in sb}
        stages["bytes_definition"] = ("sort 16 N + 76 V; project 292 V; binning 4 V + 16 D32 (D32 = (splat, 32-px bin) pairs); composite = bytes "
                                      "fetched (probe) + framebuffer; us = stage time of a serial frame (per Render call for stereo)"
                                      + ("; TWO-PASS frames: project 292 (splats of pass 1 + splats that passed the gate) + 28 V, binning 8 V + 16 "
                                         "(pairs of pass 1 + pairs of pass 2), composite = the single pass's fetched bytes (a lower bound: "
                                         "unfinished bins are walked twice)" if tp_serial else ""))
    B_moved = (16.0 * n + 76.0 * V) + launches_per_frame * (vpl * ((S + 48.0) * V + 4.0 * V) + 16.0 * Dbin_mean + B_used)
    if tp_serial and stages:
        B_moved = stages["sort"]["bytes"] + stages["project"]["bytes"] + stages["binning"]["bytes"] + B_used
    roof = {
        "kernel": "composite_kernel", "bound": "hbm", "limiter": "valu (exp + blend per pixel-splat); the HBM fraction is honest-but-low",
        "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
        "traffic": traffic, "traffic_source": tsrc,
        "bytes_per_launch": B_used,
        "bytes_definition": ("fetched under early termination: 4 B x %.0f list entries + 48 B x %.0f records (of %.0f list entries "
                             "in the bins) + %.0f B framebuffer" % (work["pair_words_fetched"], work["records_fetched"],
                                                                   work["list_entries"], fb_bytes)) if work
                            else "SURVEY 8d formula 52 D + W H bpp (no probe data)",
        "stages": stages,
        "avg_launch_ms": comp_serial_ms,
        "avg_launch_source": "hipExtLaunchKernelGGL begin/end events on the launch stream, %d serial frames (kernel alone on the GPU)"
                             % (prof_serial or {}).get("frames_averaged", 0),
        "avg_launch_ms_overlapped": comp_overlap_ms if P > 1 else None,
        "valu": ({"pixel_splat_evals_per_launch": work["pixel_evals"],
                  "gevals_per_sec": work["pixel_evals"] / comp_s / 1e9,
                  "tflops_at_20_flop_per_eval": work["pixel_evals"] * FLOP_PER_EVAL / comp_s / 1e12,
                  "frac_of_fp32_vector_peak_nominal_20_flop": work["pixel_evals"] * FLOP_PER_EVAL / comp_s / VALU_PEAK,
                  # ACHIEVED: the flops the kernel executes (14.5 per evaluation + 1 v_exp_f32, counted in the ISA) over the peak
                  "flop_per_eval_executed": FLOP_PER_EVAL_EXECUTED, "transcendentals_per_eval": 1.0,
                  "tflops_executed": work["pixel_evals"] * FLOP_PER_EVAL_EXECUTED / comp_s / 1e12,
                  "frac_of_fp32_vector_peak": work["pixel_evals"] * FLOP_PER_EVAL_EXECUTED / comp_s / VALU_PEAK,
                  # of the evaluations, the share whose weight survives the fragment shader's discard (w > 1/256,
                  # splat_frag.glsl:37-40): the rest is what evaluating a whole 16x16 tile per record costs
                  "useful_eval_frac": (work["useful_evals"] / work["pixel_evals"]) if work.get("pixel_evals") else None,
                  "records_composited_per_launch": work["records_composited"],
                  "work_items": work["work_items"]} if (work and comp_s > 0) else None),
        # the frame's HBM-bound kernel, for comparison: project_kernel gathers 256 B per visible splat (244 B record padded
        # to 4 lines) + 4 B index and writes 52 B; stage time from stream markers, serial frames
        "project_kernel_frac": ((312.0 * V) / (prof_serial["project"] * 1e-3) / HBM_PEAK)
        if prof_serial and prof_serial.get("project", 0) > 0 else None,
    }
    roof["hbm_delivered"] = hbm_delivered(dev) if (world == 1 and primary) else None      # (N > 1: the N = 1 line of the same box has it)
    out = {
        "metric": "frames_per_sec", "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not ply else "file",
        "gsplats_per_sec": n * fps / 1e9,
        "rccl_ranks": (E.pg["world_size"] if E.pg["backend"] == "nccl" else 0) if world > 1 else 1,
        "process_group": E.pg if world > 1 else None,
        "config": {"workload": wl["desc"], "key": key, "splats": n, "width": W, "height": H,
                   "views": views, "framebuffer": wl["fb"],
                   "sharding": ("bin rows of %d px over %d ranks, layout %s%s" % (TILE, world, lay, " (contiguous bands, rank 0 weighted %d %% of another rank)" % lay_k if lay_kind == "weighted" else "")) if world > 1 else "none (one GPU)",
                   "layout_model": lay_model,
                   "exchange": (("msplat_band_exchange (C ABI, RCCL group of ncclSend / ncclRecv per run of rows on the context's stream)" if comm is not None
                                 else "torch.distributed batch_isend_irecv (%s)" % dist.get_backend()) + ("; " + exchange_note if exchange_note else "")) if world > 1 else None,
                   "frames_in_flight": P, "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                   "cu_partition": [p for p, _ in r.cu_partitions()],        # MSPLAT_CU_* of every context's stream (0 all, 1 even, 2 odd)
                   "async_submit": bool(r._async),
                   "two_pass": {"mode": args.two_pass,
                                "timed_region": dict(tp_flight, frames_total=tp_frames_flight, share_pass1=tp_share_flight) if tp_flight else None,
                                "serial_frames": dict(tp_serial, share_pass1=tp_share_serial) if tp_serial else None},
                   "stereo": ("one chain for both eyes (msplat_render_stereo)" if not args.no_stereo_batch else "one Render per eye") if views == 2 else None,
                   "visible_V": V, "pairs_D": D_total, "pairs_binned_32px": float(np.mean(Dbin)),
                   "drawn": float(np.mean(drawn)), "D_over_N": D_total / max(1, n),
                   "longest_bin_list": longest_list, "pair_capacity": pair_cap_end, "pair_capacity_initial": pair_cap0,
                   "cameras": ("cameras.json, %d poses" % len(scene_cams)) if scene_cams else "64-step orbit"},
        "timed_blocks": len(blocks), "timed_seconds": float(np.sum(blocks)),
        "block_ms": {"median": 1e3 * elapsed, "min": 1e3 * float(np.min(blocks)), "p10": 1e3 * float(np.percentile(blocks, 10)),
                     "p90": 1e3 * float(np.percentile(blocks, 90)), "max": 1e3 * float(np.max(blocks)),
                     "first_64": [round(1e3 * b, 4) for b in blocks[:64]]},
        "serial": {"frames_per_sec": 1e3 / serial_ms, "ms_per_frame": serial_ms, "frames": SER,
                   "single_frame_latency_ms_host_to_host": latency_ms, "stages_ms": prof_serial},
        "stages_ms": prof,
        "host_enqueue_ms_per_frame": 1e3 * enqueue / args.steps,
        # bytes a frame moves by the stages' own design (sum of roofline.stages, compositor = fetched bytes) against the peak;
        # (r1-r3 printed SURVEY 8d's 52 D formula here, which credits bytes an early-terminating compositor never moves)
        "frame_moved_GB": B_moved / 1e9,
        "frame_moved_frac": min(1.0, (B_moved / (elapsed / args.steps)) / HBM_PEAK),
        "frame_moved_frac_serial": min(1.0, (B_moved / (serial_ms * 1e-3)) / HBM_PEAK),
        "survey_8d_formula_GB": B_frame / 1e9,
        "roofline": roof,
    }
    if gcheck is not None:
        out["gather_check"] = gcheck
    if tp_check is not None:
        out["two_pass_check"] = tp_check
    if world > 1 and gathers is not None:
        out["gather"] = {"p2p_ops_per_frame_rank0": (len(gathers[0].plan) * views) if rank == 0 else None,
                         "bytes_into_rank0_per_frame": None}
        t = torch.tensor([float(gathers[0].bytes_per_frame * views) if rank != 0 else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        out["gather"]["bytes_into_rank0_per_frame"] = float(t.item())

    if rank == 0 and world == 1 and primary and not args.no_cpu_baseline:
        out["cpu_baseline"], out["cpu_baseline_literal"] = cpu_baseline(cloud, wl, cams_for, projs, vp, nf, args.cpu_frames)
    if rank == 0 and primary and args.save_image:
        img = state["fbs"][0][:H].float().cpu().numpy()
        camera.write_image(args.save_image, img)
    if rs is not r:
        rs.close()
    r.close()
    if scene_dir:
        import shutil
        shutil.rmtree(scene_dir, ignore_errors=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--also", default=None, help="comma-separated extra workloads measured after the primary one and "
                    "embedded under \"also\" (default: cfg4 when --gpus > 1 -- the row-sharded BASELINE configs[3])")
    ap.add_argument("--ply", default=None, help="render a real scene instead of the synthetic workload (BASELINE configs[2]: "
                    "Inria point_cloud.ply); cameras.json next to it (or up to two directories above) is replayed")
    ap.add_argument("--layout", default="auto", help="N > 1: how the bin rows are dealt to the ranks: contiguous | interleaved | "
                    "block:k (blocks of k rows round-robin) | auto")
    ap.add_argument("--exchange", default="auto", choices=["auto", "cabi", "torch"],
                    help="N > 1: the row gather of the timed frames: cabi = msplat_band_exchange (libmsplat's C ABI, RCCL), torch = "
                         "torch.distributed batch_isend_irecv; auto = cabi under the nccl backend, else torch")
    ap.add_argument("--save-image", default=None, help="write the last frame of rank 0 as PNG")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="CPU baseline frames (0 = auto, about 10-30 s)")
    ap.add_argument("--profile-frames", type=int, default=6, help="extra frames (outside the timed regions) for V/D/work statistics")
    ap.add_argument("--serial-frames", type=int, default=128, help="frames of the serial (one stream, one at a time) phase")
    ap.add_argument("--prewarm", type=int, default=400, help="untimed frames before the warm-up (runtime pool growth)")
    ap.add_argument("--cu-partition", default="auto", choices=["auto", "off"],
                    help="auto: four frames in flight -> the contexts' streams alternate between the even and the odd CU positions of every "
                         "XCD (msplat_config.cu_partition, the shims' rule); off: every stream on every CU (A/B)")
    ap.add_argument("--compositor-waves", type=int, default=None,
                    help="persistent compositor waves per launch (msplat_config.compositor_waves; default: the library's choice -- A/B runs)")
    ap.add_argument("--frames-in-flight", type=int, default=4,
                    help="frames overlapped on the GPU (one context + stream + framebuffer per frame in flight, one shared "
                         "cloud); 1 = strictly serial frames (latency mode)")
    ap.add_argument("--peer-store-check", action="store_true",
                    help="(run by rank 0 of an N > 1 bench in a child process) one process, --gpus devices: msplat_group_* renders two "
                         "poses with peer stores into device 0's framebuffer and compares them bit for bit with a single context")
    ap.add_argument("--no-stereo-batch", action="store_true",
                    help="two-view workloads: one Render per eye (the reference's call pattern) instead of msplat_render_stereo (A/B)")
    ap.add_argument("--two-pass", default="auto", choices=["auto", "on", "off"],
                    help="msplat_config.two_pass: Renders in two passes with occlusion feedback (same pixels; A/B)")
    ap.add_argument("--two-pass-share", type=float, default=0.0,
                    help="pin the share of the visible splats in the first pass (msplat_debug_two_pass; 0 = the library's feedback loop)")
    ap.add_argument("--async-submit", type=int, default=-1,
                    help="msplat_config.async_submit of the in-flight contexts: 1 = a worker thread per context issues its launches, "
                         "0 = the calling thread does (A/B); default: on with frames in flight")
    ap.add_argument("--timing-stride", type=int, default=8,
                    help="record per-stage hipEvents on every n-th frame of the timed region (0 = never)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft

    if args.peer_store_check:
        return peer_store_check(args)
    E = Env()
    E.rank = int(os.environ.get("RANK", "0"))
    E.world = int(os.environ.get("WORLD_SIZE", "1"))
    E.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if E.world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, E.world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    if E.rank == 0:
        graft.build()
    # MSPLAT_BENCH_ONE_DEVICE=1: debug aid for 1-GPU boxes -- every rank uses device 0 and the gather
    # runs over gloo; it exercises the N > 1 control flow (bands, gather, max-over-ranks timing), not xGMI.
    E.one_dev = os.environ.get("MSPLAT_BENCH_ONE_DEVICE") == "1"
    if E.one_dev:
        E.local_rank = 0
    torch.cuda.set_device(E.local_rank)
    E.dev = torch.device("cuda", E.local_rank)
    E.pg, E.cpu_group = {"backend": None, "world_size": 1}, None
    if E.world > 1:
        # a first run on real multi-GPU hardware must not hang silently: if the invocation has not finished after 20 minutes every
        # rank dumps where its threads are (stderr) and exits -- the driver then has a stack, not a timeout
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("MSPLAT_BENCH_WATCHDOG_S", "1200")), exit=True)
        if E.one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=E.dev)
        dist.barrier()
        # host-side barriers (while one rank works alone, the others must not spin in a collective kernel on their GPUs)
        E.cpu_group = dist.new_group(backend="gloo")
        E.pg = process_group_report(E, torch, dist)
    # a dedicated (non-null) torch stream: libmsplat launches on it, so torch copies, RCCL's stream
    # hand-off and torch.cuda.synchronize all order correctly with the HIP kernels
    E.stream = torch.cuda.Stream(device=E.dev)
    torch.cuda.set_stream(E.stream)

    out = measure(E, args, args.workload, ply=args.ply, primary=True)
    also = args.also if args.also is not None else ("cfg4" if (E.world > 1 and args.workload == "cfg2" and not args.ply) else "")
    extra = {}
    for key in [k for k in also.split(",") if k]:
        if key not in WORKLOADS:
            raise SystemExit("unknown workload in --also: " + key)
        sub = measure(E, args, key, primary=False)
        extra[key] = {k: sub[k] for k in ("value", "unit", "ms_per_step", "gsplats_per_sec", "n_gpus", "rccl_ranks", "config",
                                          "timed_blocks", "serial", "stages_ms", "frame_moved_frac", "roofline", "gather",
                                          "gather_check", "two_pass_check")
                      if k in sub}
    if extra:
        out["also"] = extra
    # N > 1: a gathered frame that is not bit-identical to the single-context frame voids the number
    bad = [k for k, d in [(args.workload, out)] + list(extra.items())
           if d.get("gather_check") is not None and not d["gather_check"]["bit_exact"]]
    if bad:
        out["value"] = None
        out["error"] = "gather_check failed for %s: rank 0's gathered frame differs from the unbanded render" % ", ".join(bad)
    # a two-pass frame that is not bit-identical to the single pass voids the number as well
    bad2 = [k for k, d in [(args.workload, out)] + list(extra.items())
            if d.get("two_pass_check") is not None and not d["two_pass_check"]["bit_exact"]]
    if bad2:
        out["value"] = None
        out["error"] = (out.get("error", "") + " two_pass_check failed for %s: a two-pass frame differs from the single pass" % ", ".join(bad2)).strip()
        bad = bad + bad2
    if E.rank == 0:
        print(json.dumps(out))
    if E.world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if bad:
        sys.exit(3)


def process_group_report(E, torch, dist):
    """what the LIVE process group says: backend, world size, and every rank's device (ordinal, PCI bus id, name), so that a
    run on N distinct GPUs can be told from N ranks on one"""
    props = torch.cuda.get_device_properties(E.dev)
    bus = None
    if hasattr(props, "pci_bus_id"):                # asked of the runtime torch already holds (no second HIP runtime by dlopen)
        bus = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, getattr(props, "pci_device_id", 0))
    if getattr(props, "uuid", None) is not None:
        bus = (bus or "") + " " + str(props.uuid)
    mine = {"rank": E.rank, "local_rank": E.local_rank, "device": int(E.dev.index), "pci_bus_id": bus, "name": props.name,
            "host": os.uname().nodename, "pid": os.getpid()}
    allr = [None] * dist.get_world_size()
    dist.all_gather_object(allr, mine, group=E.cpu_group)
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": allr,
            "distinct_devices": len({(r["host"], r["pci_bus_id"] or r["device"]) for r in allr})}


def peer_store_check(args):
    """child process of an N > 1 bench (rank 0 spawns it while the other ranks wait at a host barrier): the single-process
    device group (msplat_group_*: every device's compositor stores its rows into device 0's framebuffer through the peer mapping,
    no RCCL) against a plain single context, two orbit poses, bit for bit.  Prints one JSON line."""
    import torch
    from splatapult_amd import SplatRenderer, SplatRendererGroup, camera, synthetic, _capi
    wl = WORKLOADS[args.workload]
    W, H, views, G = wl["W"], wl["H"], wl["views"], args.gpus
    out = {"devices": list(range(G)), "bit_exact": None, "peer_store_ranks": None, "error": None, "poses": [5, 37]}
    try:
        cloud = synthetic.make_cloud(wl["n"], seed=wl["seed"], full_sh=True, pos_sigma=wl["pos_sigma"])
        lay = args.layout
        kind, k = (lay.split(":")[0], int(lay.split(":")[1])) if ":" in lay else (("block", 1) if lay == "auto" else (lay, 1))
        T = _capi.lib().msplat_tile_size()
        if lay in ("auto", "weighted"):
            kind, k = "block", max(1, ((H + T - 1) // T) // (2 * G))
        dev = torch.device("cuda", 0)
        tdt, bpp = (torch.float16, 8) if wl["fb"] == "fp16" else (torch.float32, 16)
        bits = torch.int16 if wl["fb"] == "fp16" else torch.int32
        vp, nf = [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR]
        projs = [camera.perspective(camera.FOVY, W / H)] if views == 1 else \
            [camera.create_projection(-1.0, 0.8, 0.95, -0.95), camera.create_projection(-0.8, 1.0, 0.95, -0.95)]
        ref = SplatRenderer(device=0, fb_format=wl["fb"])
        g = SplatRendererGroup(list(range(G)), fb_format=wl["fb"], layout=kind, block_rows=k, band_cull=(views == 1))
        if not ref.Init(cloud, False, False) or not g.Init(cloud, False, False):
            raise RuntimeError(ref.last_error() or g.last_error())
        out["peer_store_ranks"] = [i for i in range(g.size) if g.peer_store(i)]
        a, b = torch.zeros((H, W, 4), dtype=tdt, device=dev), torch.zeros((H, W, 4), dtype=tdt, device=dev)

        def compare():
            ok = True
            for step in out["poses"]:
                c = camera.orbit(wl["cam_z"], 2.0 * math.pi * step / 64.0)
                cams = [c] if views == 1 else [camera.translate_local(c, dx=-0.032), camera.translate_local(c, dx=+0.032)]
                ref.Sort(cams[0], projs[0], vp, nf)
                g.Sort(cams[0], projs[0], vp, nf)
                for v in range(views):
                    a.zero_(); b.zero_()
                    torch.cuda.synchronize(dev)
                    ref.Render(cams[v], projs[v], vp, nf, out_ptr=a.data_ptr(), pitch_bytes=W * bpp)
                    g.Render(cams[v], projs[v], vp, nf, out_ptr=b.data_ptr(), pitch_bytes=W * bpp)
                    ref.synchronize(); g.synchronize()
                    ok = ok and bool((a.view(bits) == b.view(bits)).all().item())
            return ok

        def frames_ms(n=12):
            c = camera.orbit(wl["cam_z"], 0.5)
            for k in range(n + 4):
                if k == 4:
                    g.synchronize(); torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                g.Sort(c, projs[0], vp, nf)
                g.Render(c, projs[0], vp, nf, out_ptr=b.data_ptr(), pitch_bytes=W * bpp)
            g.synchronize(); torch.cuda.synchronize(dev)
            return 1e3 * (time.perf_counter() - t0) / n

        out["bit_exact"] = compare()                       # the default exchange: peer stores where the mapping exists
        out["exchanges"] = {g.exchange(): {"bit_exact": out["bit_exact"], "ms_per_frame": frames_ms()}}
        # r5: the same rows over RCCL (ncclCommInitAll + grouped ncclSend / ncclRecv behind msplat_group_render) and as plain copies
        for name in ("rccl", "copy"):
            if G < 2:
                break
            try:
                g.set_exchange(name)
                ok = compare()
                out["exchanges"][name] = {"bit_exact": ok, "ms_per_frame": frames_ms(), "used": g.exchange()}
            except Exception as e:
                out["exchanges"][name] = {"bit_exact": None, "error": "%s: %s" % (type(e).__name__, e)}
        g.close(); ref.close()
    except Exception as e:
        out["error"] = "%s: %s" % (type(e).__name__, e)
    print(json.dumps(out))
    return 0


def two_pass_check(E, wl, init, cams_for, projs, vp, nf, W, Hpad, bpp, fdt, share, band):
    """Two fresh contexts -- two passes forced with the share the timed frames ended on, and one pass -- render orbit poses 5 and 37
    (this rank's bin rows when the frame is row-sharded); the two-pass frames must equal the single-pass ones bit for bit
    (one Render = one image however it is scheduled, src/splatrenderer.cpp:315-343)."""
    import torch
    from splatapult_amd import SplatRenderer, _capi
    res = {"bit_exact": True, "poses": [5, 37], "share": float(share), "values_compared": 0, "values_different": 0, "two_pass_frames": 0}
    rr = []
    for mode in (_capi.TWO_PASS_OFF, _capi.TWO_PASS_ON):
        x = SplatRenderer(device=E.local_rank, fb_format=wl["fb"], stream=E.stream.cuda_stream, frames_in_flight=1, two_pass=mode)
        init(x)
        if band is not None:
            x.set_band_plan(band[0], band[1], band[2], band[3], block_rows=band[4], band_cull=True)
        rr.append(x)
    rr[1].two_pass_state(share)
    fa = torch.zeros((Hpad, W, 4), dtype=fdt, device=E.dev)
    fb = torch.zeros((Hpad, W, 4), dtype=fdt, device=E.dev)
    for step in res["poses"]:
        cam = cams_for(step)[0]
        for x, f in ((rr[0], fa), (rr[1], fb)):
            x.Sort(cam, projs[0], vp, nf)
            x.Render(cam, projs[0], vp, nf, out_ptr=f.data_ptr(), pitch_bytes=W * bpp)
            x.synchronize()
        torch.cuda.synchronize(E.dev)
        diff = int((fa != fb).sum().item())
        res["values_compared"] += fa.numel()
        res["values_different"] += diff
    res["two_pass_frames"] = int(rr[1].two_pass_state(share)[0])
    res["bit_exact"] = res["values_different"] == 0 and res["two_pass_frames"] == len(res["poses"])
    for x in rr:
        x.close()
    return res


def gather_check(E, args, wl, init, frame, rs, rs_sets, cams_for, projs, vp, nf, cloud, lay_kind, lay_k, primary, comm=None):
    """Rank 0 renders two orbit poses UNBANDED on a plain single context and compares them bit for bit with the frame the ranks
    rendered in bands and gathered into its framebuffer (the bench's own exchange: grouped send / receive over RCCL, or gloo in
    the one-device debug mode).  On the primary workload, when rank 0's process can see one device per rank, the single-process
    device group (msplat_group_*: the other devices' compositors store their rows into device 0's framebuffer over the peer
    mapping) renders the same poses too.  Outside every timed region."""
    import torch
    import torch.distributed as dist
    from splatapult_amd import SplatRenderer
    rank, world, dev, stream = E.rank, E.world, E.dev, E.stream
    W, H, views = wl["W"], wl["H"], wl["views"]
    bpp = 8 if wl["fb"] == "fp16" else 16
    poses = [5, 37]
    bits = torch.int16 if wl["fb"] == "fp16" else torch.int32

    def host_barrier():
        torch.cuda.synchronize(dev)
        dist.barrier(group=E.cpu_group)

    res = {"bit_exact": True, "poses": poses, "exchange": dist.get_backend(), "values_compared": 0, "values_different": 0,
           "max_abs_diff": 0.0, "peer_store": None}
    wire16_on = comm is not None and os.environ.get("MSPLAT_BENCH_WIRE_FP16") == "1" and wl["fb"] != "fp16"
    ref, ref_fbs = None, None
    if rank == 0:
        ref = SplatRenderer(device=E.local_rank, fb_format=wl["fb"], stream=stream.cuda_stream, frames_in_flight=1)
        init(ref)
        ref_fbs = [torch.zeros_like(rs_sets[0][v]) for v in range(views)]
    for k in poses:
        for t in rs_sets[0]:
            t.zero_()                               # rows left by an earlier frame must not pass for gathered ones
        host_barrier()
        frame(k, rs, rs_sets)                       # every rank: its bands + the gather into rank 0's framebuffer
        host_barrier()
        if rank == 0:
            cams = cams_for(k)
            ref.Sort(cams[0], projs[0], vp, nf)
            for v in range(views):
                ref.Render(cams[v], projs[v], vp, nf, out_ptr=ref_fbs[v].data_ptr(), pitch_bytes=W * bpp)
            torch.cuda.synchronize(dev)
            for v in range(views):
                a, b = rs_sets[0][v][:H], ref_fbs[v][:H]
                ne = int((a.view(bits) != b.view(bits)).sum().item())
                res["values_compared"] += a.numel()
                res["values_different"] += ne
                if ne:
                    res["max_abs_diff"] = max(res["max_abs_diff"], float((a.float() - b.float()).abs().max().item()))
    # what the exchange costs each rank (r5): frames whose gather is bracketed by events on the gathering stream -- rank 0's
    # interval is the arrival of every foreign run, the others' their sends; reported per rank so that a first multi-GPU run
    # shows whether the rows or the kernels bound the frame
    try:
        n_t = 6
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_t)]
        state_g = {"k": 0}
        from splatapult_amd import dist as _sd
        GatherClass = _sd.CAbiBandGather if comm is not None else _sd.BandGather
        orig_call = GatherClass.__call__
        ev_stream = torch.cuda.ExternalStream(rs._lib.msplat_get_stream(rs._ctx)) if comm is not None else stream      # the exchange's stream

        def timed_call(self, fb):
            k = state_g["k"]
            if k < n_t:
                if comm is not None:
                    self.r.synchronize()          # (the frame's launches are issued and done: the interval is the exchange alone)
                evs[k][0].record(ev_stream)
            out_ = orig_call(self, fb)
            if k < n_t:
                evs[k][1].record(ev_stream)
            state_g["k"] = k + 1
            return out_
        GatherClass.__call__ = timed_call
        try:
            for k in range(n_t // max(1, views)):
                frame(100 + k, rs, rs_sets)
                host_barrier()
        finally:
            GatherClass.__call__ = orig_call
        torch.cuda.synchronize(dev)
        mine = float(np.median([a.elapsed_time(b) for a, b in evs[:state_g["k"]]])) if state_g["k"] else None
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine, group=E.cpu_group)
        res["exchange_ms_per_gather_per_rank"] = per_rank
        res["exchange_call"] = ("msplat_band_exchange (libmsplat's C ABI): ncclGroupStart / ncclRecv | ncclSend per run of rows / ncclGroupEnd on the "
                                "context's stream" + (", rows as RGBA16F on the wire" if wire16_on else "")) if comm is not None else (
            "torch.distributed.batch_isend_irecv = ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd, one run of rows per op"
            if dist.get_backend() == "nccl" else "gloo stand-in (host staging)")
    except Exception as e:
        res["exchange_ms_per_gather_per_rank"] = "%s: %s" % (type(e).__name__, e)
    # opt-in (MSPLAT_BENCH_CABI_EXCHANGE=1, RCCL only): the same gather through the C ABI -- msplat_band_exchange with a communicator
    # of this bench's own (ncclCommInitRank; collective, so it is not on by default on hardware nobody has run it on)
    if comm is None and os.environ.get("MSPLAT_BENCH_CABI_EXCHANGE") == "1" and dist.get_backend() == "nccl" and not E.one_dev and views == 1:
        cab = {"bit_exact": None, "error": None}
        try:
            from splatapult_amd.dist import CAbiBandGather, RcclComm
            comm = E.rccl_comm()
            from splatapult_amd import _capi as _cp
            TILE = _cp.lib().msplat_tile_size()
            tiles_y = rs_sets[0][0].shape[0] // TILE
            wire16 = os.environ.get("MSPLAT_BENCH_WIRE_FP16") == "1" and wl["fb"] != "fp16"
            cab["wire_fp16"] = wire16
            cg = CAbiBandGather(rs, comm, tiles_y, W, rs_sets[0][0].dtype, rank, world, tile=TILE, layout=lay_kind, block_rows=lay_k,
                                wire_fp16=wire16)
            cams = cams_for(poses[0])
            rs_sets[0][0].zero_()
            host_barrier()
            rs.Sort(cams[0], projs[0], vp, nf)
            rs.Render(cams[0], projs[0], vp, nf, out_ptr=rs_sets[0][0].data_ptr(), pitch_bytes=W * bpp)
            cg(rs_sets[0][0])
            rs.synchronize()
            host_barrier()
            if rank == 0:
                ref.Sort(cams[0], projs[0], vp, nf)
                ref.Render(cams[0], projs[0], vp, nf, out_ptr=ref_fbs[0].data_ptr(), pitch_bytes=W * bpp)
                torch.cuda.synchronize(dev)
                cab["bit_exact"] = bool((rs_sets[0][0][:H].view(bits) == ref_fbs[0][:H].view(bits)).all().item())
                cab["max_abs_diff"] = float((rs_sets[0][0][:H].float() - ref_fbs[0][:H].float()).abs().max().item())      # fp16 wire: <= 2^-11 |value|
        except Exception as e:
            cab["error"] = "%s: %s" % (type(e).__name__, e)
        res["c_abi_exchange"] = cab
    # the other exchange form: one process, one context per device, peer stores into device 0's framebuffer.  Run in a CHILD
    # process (bench.py --peer-store-check): a fault on that path must not take the bench line with it
    if rank == 0 and primary and not E.one_dev and torch.cuda.device_count() >= world and not args.ply:
        import subprocess
        ps = {"devices": list(range(world)), "bit_exact": None, "error": None}
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--peer-store-check", "--gpus", str(world), "--workload", wl["key"],
                   "--layout", "%s:%d" % (lay_kind, lay_k) if lay_kind in ("block", "weighted") else lay_kind]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")
                   and not k.startswith("TORCHELASTIC")}
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode == 0 and lines:
                ps = json.loads(lines[-1])
            else:
                ps["error"] = "exit %d: %s" % (p.returncode, (p.stderr or p.stdout)[-400:])
        except Exception as e:                      # reported, not fatal: the bench's own exchange is the gather above
            ps["error"] = "%s: %s" % (type(e).__name__, e)
        res["peer_store"] = ps
    if ref is not None:
        ref.close()
    host_barrier()
    flag = torch.tensor([res["values_different"]], dtype=torch.int64)
    dist.broadcast(flag, src=0, group=E.cpu_group)
    res["values_different"] = int(flag.item())
    res["bit_exact"] = res["values_different"] == 0
    return res


def cpu_baseline(cloud, wl, cams_for, projs, vp, nf, frames):
    """The CPU-side sort + raster path on the host cores (the reference has none of its own: src/sdl_main.cpp:28 is a
    commented-out define): oracle/msplat_cpu_tiled.c -- parallel cull / stable radix sort / projection, 16x16-tile binning,
    front-to-back compositor with early termination -- timed on a bounded sample of the same workload; next to it the
    literal oracle (the reference shaders restated one to one, back-to-front over every pixel row band) on one frame."""
    from oracle import oracle as orc
    try:
        avail = len(os.sched_getaffinity(0))          # the CPUs this process may run on (a container may own fewer than it sees)
    except AttributeError:
        avail = os.cpu_count() or 1
    aos = cloud.as_array()
    views = wl["views"]
    W, H = int(vp[2]), int(vp[3])
    img = np.zeros((H, W, 4), np.float32)

    def one(step, stages=None, nt=None):
        cams = cams_for(step)
        t = time.perf_counter()
        for v in range(views):
            r = orc.render_frame_tiled(aos, True, cams[0], projs[0], vp, nf, render_cam=cams[v], render_proj=projs[v],
                                       nthreads=nt or cores, image=img)
            if stages is not None:
                stages.append(r["stages_ms"])
        return time.perf_counter() - t

    # thread count: the fastest of a few candidates (more threads than the scheduler really grants -- a cgroup CPU quota, SMT
    # siblings -- make every OpenMP barrier slower, so "all the CPUs the OS lists" is not always best: on the r3 GPU box the
    # container sees 256 CPUs and owns a quota of 16, and 256 threads were 50x slower than 32)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    eff = int(min(avail, math.ceil(quota))) if quota else avail
    cores = eff
    one(0, nt=min(eff, 16))                           # first touch of the work buffers
    trial = {}
    for nt in sorted({max(1, eff // 2), eff, min(avail, 2 * eff), min(avail, 4 * eff)} | ({avail} if not quota else set())):
        trial[nt] = float(np.median([one(k, nt=nt) for k in range(3)]))
    cores = min(trial, key=trial.get)
    t_first = trial[cores]
    if frames <= 0:
        frames = int(max(3, min(64, 12.0 // max(t_first, 1e-3))))
    stages = []
    times = [one(s, stages) for s in range(frames)]
    sec = float(np.median(times))
    st = {k: float(np.median([s[k] for s in stages])) for k in stages[0]}
    out = {"value": 1.0 / sec, "unit": "frames/s", "cores": int(min(cores, eff)), "threads": cores, "kind": "port-tiled",
           "sample": "%d frame(s) of the same workload (orbit steps 0..%d), median, %d host threads (OpenMP), "
                     "oracle/msplat_cpu_tiled.c: tile-binned, front-to-back, early termination at T < 2^-14"
                     % (len(times), len(times) - 1, cores),
           "sec_per_frame": sec, "gsplats_per_sec": wl["n"] / sec / 1e9, "stages_ms": st,
           "cpus_available": avail, "cgroup_cpu_quota": quota, "thread_count_trials_sec": {str(k): v for k, v in trial.items()}}
    # the literal restatement of the reference shaders (what cpu_baseline was in rounds 1-2): one frame
    cams = cams_for(0)
    t = time.perf_counter()
    for v in range(views):
        orc.render_frame(aos, True, cams[0], projs[0], vp, nf, render_cam=cams[v], render_proj=projs[v], nthreads=avail)
    lit = time.perf_counter() - t
    cores = avail
    return out, {"value": 1.0 / lit, "unit": "frames/s", "cores": int(min(avail, eff)), "threads": avail, "kind": "port",
                 "sample": "1 frame (orbit step 0), %d host threads, oracle/msplat_oracle.c via OpenMP row bands (every band "
                           "walks all visible splats back to front; single-threaded sort)" % cores,
                 "sec_per_frame": lit}


if __name__ == "__main__":
    main()
