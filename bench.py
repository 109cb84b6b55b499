#!/usr/bin/env python
"""bench.py -- frames/s (+ Gsplats/s) of the splat Sort+Render hot path on N MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one frame = SplatRenderer::Sort + SplatRenderer::Render of the resident cloud from a
camera on a 64-step orbit (stereo workloads: one Sort + two Renders).  With N > 1 the frame's tile
rows are sharded across the ranks (interleaved, row % N == rank) and gathered to rank 0 over RCCL:
total work is fixed, so scaling is "strong".  Rank 0 prints ONE JSON line.

PyTorch is plumbing only (device selection, the framebuffer tensor, torch.distributed); all compute
is libmsplat.so's HIP kernels launched on torch's current stream.
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
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
    "tiny": dict(n=20_000, seed=7, pos_sigma=1.5, W=640, H=360, cam_z=7.0, fb="fp32", views=1,
                 desc="20k synthetic Gaussians, 640x360 (debug)"),
}
HBM_PEAK = 8.0e12   # B/s, MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--ply", default=None, help="render a real scene instead of the synthetic workload (BASELINE configs[2]: "
                    "Inria point_cloud.ply); cameras.json next to it (or up to two directories above) is replayed")
    ap.add_argument("--save-image", default=None, help="write the last frame of rank 0 as PNG")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="CPU baseline frames (0 = auto, about 10-30 s)")
    ap.add_argument("--profile-frames", type=int, default=8, help="extra frames (outside the timed region) for V/D statistics")
    ap.add_argument("--prewarm", type=int, default=400, help="untimed frames before the warm-up (runtime pool growth)")
    ap.add_argument("--frames-in-flight", type=int, default=4,
                    help="frames overlapped on the GPU (one context + stream + framebuffer per frame in flight, one shared "
                         "cloud); 1 = strictly serial frames (latency mode)")
    ap.add_argument("--timing-stride", type=int, default=8,
                    help="record per-stage hipEvents on every n-th frame of the timed region (0 = never)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    from splatapult_amd import SplatRenderer, camera, synthetic

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    if rank == 0:
        graft.build()
    # MSPLAT_BENCH_ONE_DEVICE=1: debug aid for 1-GPU boxes -- every rank uses device 0 and the gather
    # runs over gloo; it exercises the N > 1 control flow (bands, gather, max-over-ranks timing), not xGMI.
    one_dev = os.environ.get("MSPLAT_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        dist.barrier()

    wl = dict(WORKLOADS[args.workload])
    W, H, views = wl["W"], wl["H"], wl["views"]
    t0 = time.time()
    scene_cams = None
    if args.ply:
        from splatapult_amd import GaussianCloud
        cloud = GaussianCloud()
        if not cloud.ImportPly(args.ply):
            raise SystemExit("cannot import " + args.ply)
        wl["n"] = cloud.GetNumGaussians()
        wl["desc"] = "%s (%d splats), %dx%d %s" % (os.path.basename(args.ply), wl["n"], W, H, wl["fb"])
        cj = camera.find_config_file(args.ply, "cameras.json")          # app.cpp:418-461
        if cj:
            scene_cams = [m for m, _ in camera.load_cameras_json(cj)]
    else:
        cloud = synthetic.make_cloud(wl["n"], seed=wl["seed"], full_sh=True, pos_sigma=wl["pos_sigma"])
    n = wl["n"]
    t_gen = time.time() - t0

    # a dedicated (non-null) torch stream: libmsplat launches on it, so torch copies, RCCL's stream
    # hand-off and torch.cuda.synchronize all order correctly with the HIP kernels
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    P = max(1, args.frames_in_flight)
    r = SplatRenderer(device=local_rank, fb_format=wl["fb"], stream=stream.cuda_stream, enable_timing=args.timing_stride,
                      frames_in_flight=P)
    if not r.Init(cloud, False, False):
        raise SystemExit("Init failed: " + r.last_error())
    if world > 1:
        # mono workloads may also restrict the cull to the band (every Render uses its Sort's camera)
        r.set_band(world, rank, band_cull=(wl["views"] == 1))

    from splatapult_amd import _capi
    TILE = _capi.lib().msplat_tile_size()
    tiles_y = (H + TILE - 1) // TILE
    Hpad = tiles_y * TILE
    fdt = torch.float16 if wl["fb"] == "fp16" else torch.float32
    bpp = 8 if wl["fb"] == "fp16" else 16
    # one framebuffer set per frame in flight
    fb_sets = [[torch.zeros((Hpad, W, 4), dtype=fdt, device=dev) for _ in range(views)] for _ in range(P)]
    fbs = fb_sets[0]
    fb_free = [None] * P      # N > 1: event recorded after the gather that read the slot's framebuffers
    vp, nf = [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR]
    if views == 1:
        projs = [camera.perspective(camera.FOVY, W / H)]
    else:   # BASELINE config 5: asymmetric XR frusta (util.cpp:420-480)
        projs = [camera.create_projection(-1.0, 0.8, 0.95, -0.95), camera.create_projection(-0.8, 1.0, 0.95, -0.95)]

    def cams_for(step):
        if scene_cams:
            c = scene_cams[step % len(scene_cams)]
        else:
            c = camera.orbit(wl["cam_z"], 2.0 * math.pi * (step % 64) / 64.0)
        if views == 1:
            return [c]
        return [camera.translate_local(c, dx=-0.032), camera.translate_local(c, dx=+0.032)]

    # the only exchange step: gather each rank's tile rows to rank 0 (splatapult_amd/dist.py)
    gathers = None
    if world > 1:
        from splatapult_amd.dist import BandGather
        gathers = [BandGather(tiles_y, W, fdt, dev, rank, world, tile=TILE) for _ in range(views)]

    def frame(step):
        nonlocal fbs
        cams = cams_for(step)
        if P > 1 and gathers is not None:
            ev = fb_free[(r.frame_slot + 1) % P]
            if ev is not None:
                r.next_frame_wait_event(ev.cuda_event)         # the slot's previous frame has been gathered
        r.Sort(cams[0], projs[0], vp, nf)                      # sort once with view 0 (app.cpp:603-606)
        fbs = fb_sets[r.frame_slot]
        for v in range(views):
            r.Render(cams[v], projs[v], vp, nf, out_ptr=fbs[v].data_ptr(), pitch_bytes=W * bpp)
            if gathers is not None:
                if P > 1:
                    r.wait_on_stream(stream.cuda_stream)       # the gather's stream waits for this frame only
                gathers[v](fbs[v])
        if P > 1 and gathers is not None:
            ev = torch.cuda.Event()
            ev.record(stream)
            fb_free[r.frame_slot] = ev

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # The HIP runtime grows its per-queue kernarg / signal pools once, a few thousand launches after start-up
    # (measured: a single 15-40 ms stall around frame 170-210, scratch timeline in DESIGN.md 6): get past it
    # before the official warm-up so that short --steps runs measure the steady state too.
    for s in range(args.prewarm):
        frame(s)
    sync_all()
    for s in range(args.warmup):
        frame(s)
    sync_all()
    t0 = time.perf_counter()
    for s in range(args.steps):
        frame(args.warmup + s)
    enqueue = time.perf_counter() - t0        # host time to issue the frames (launch-rate bound check)
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-stage / per-kernel times: hipEvents recorded on the launch stream on every `timing_stride`-th
    # frame INSIDE the timed region (the markers cost a few us each, hence sampled); averaged here
    prof = dict(sort_total=0.0, render_total=0.0, project=0.0, binning=0.0, composite=0.0, composite_kernel=0.0)
    if args.timing_stride > 0:
        prof = r.timings()
    # V / D statistics on a few extra frames outside the timed region (each read synchronises)
    Vs, Ds, drawn, Dbin = [], [], [], []
    for s in range(max(1, args.profile_frames)):
        frame(args.warmup + args.steps + s)
        st = r.stats()
        Vs.append(st["sort_count"]); Ds.append(st["pairs_tile16"]); drawn.append(st["drawn"]); Dbin.append(st["pairs"])
    st = r.stats()
    V, D = float(np.mean(Vs)), float(np.mean(Ds))
    # latency of ONE frame with nothing else in flight (outside the timed region)
    lat = []
    for s in range(16 * P):
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        frame(args.warmup + args.steps + 64 + s)
        torch.cuda.synchronize(dev)
        lat.append(time.perf_counter() - t1)
    latency_ms = 1e3 * float(np.median(lat))
    prof_serial = r.timings() if args.timing_stride > 0 else None     # the same events, frames not overlapped
    if world > 1:
        t = torch.tensor([V, D, prof["composite"]], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        D_total = float(t[1].item())
    else:
        D_total = D

    fps = args.steps / elapsed
    ms = 1e3 * elapsed / args.steps
    # algorithmic bytes (SURVEY.md 8d / BASELINE.md):  B = 16 N + (8+68+S+48) V + views (52 D + W H bpp)
    S = 244
    B_frame = 16.0 * n + (8 + 68 + S + 48) * V + (52.0 * D_total + W * H * bpp) * views
    # dominant kernel: composite.  per launch: 52 B per (splat,tile) pair + the framebuffer write
    B_comp = 52.0 * D + (W * H * bpp) / world
    comp_ms = prof.get("composite_kernel", 0.0) or prof["composite"]   # exact kernel begin/end events
    comp_s = comp_ms * 1e-3
    achieved = B_comp / comp_s if comp_s > 0 else 0.0
    # HBM traffic of the dominant kernel comes from a SEPARATE rocprofv3 --pmc run (tools/pmc_traffic.sh:
    # FETCH_SIZE / WRITE_SIZE in their own passes, gfx950 x2 correction on FETCH_SIZE); the committed
    # summary is only quoted for the workload it was measured on
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic_%s.json" % args.workload)
    if world == 1 and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = "msplat::composite_kernel<%s>" % ("true" if wl["fb"] == "fp16" else "false")
            traffic = tj[key]["hbm_bytes_per_launch_corrected"]
        except (KeyError, ValueError):
            traffic = None

    out = {
        "metric": "frames_per_sec", "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not args.ply else "file",
        "gsplats_per_sec": n * fps / 1e9,
        "config": {"workload": wl["desc"], "key": args.workload, "splats": n, "width": W, "height": H,
                   "views": views, "framebuffer": wl["fb"], "sharding": "tile rows, row %% %d == rank" % world,
                   "frames_in_flight": P, "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                   "visible_V": V, "pairs_D": D_total, "pairs_binned_32px": float(np.mean(Dbin)),
                   "drawn": float(np.mean(drawn))},
        "stages_ms": prof,
        "single_frame_latency_ms": latency_ms,
        "host_enqueue_ms_per_frame": 1e3 * enqueue / args.steps,
        "frame_algorithmic_GB": B_frame / 1e9,
        "frame_hbm_frac": (B_frame / (elapsed / args.steps)) / HBM_PEAK / world,
        "roofline": {"kernel": "composite_kernel", "bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                     "traffic_source": "profiles/r01_pmc_traffic_%s.json (rocprofv3 --pmc, bytes per launch)" % args.workload if traffic else None,
                     "algorithmic_bytes_per_launch": B_comp,
                     "avg_launch_ms": comp_ms,
                     "avg_launch_ms_one_frame_at_a_time": prof_serial["composite_kernel"] if prof_serial else None,
                     "frac_one_frame_at_a_time": (B_comp / (prof_serial["composite_kernel"] * 1e-3) / HBM_PEAK)
                     if prof_serial and prof_serial["composite_kernel"] > 0 else None,
                     # the frame's HBM-bound kernel, for comparison: project_kernel gathers 256 B per visible splat (244 B
                     # record padded to 4 lines) + 4 B index and writes 52 B; stage time from stream markers, serial frames
                     "project_kernel_frac_one_frame_at_a_time": ((312.0 * V) / (prof_serial["project"] * 1e-3) / HBM_PEAK)
                     if prof_serial and prof_serial["project"] > 0 else None,
                     "note": "composite is VALU/LDS bound (exp + blend per pixel-splat); HBM fraction is honest-but-low"
                             + ("; launch duration measured while %d frames share the GPU" % P if P > 1 else "")},
    }

    if rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cloud, wl, cams_for, projs, vp, nf, args.cpu_frames)

    if rank == 0 and args.save_image:
        img = (gathers[0].final.view(Hpad, W, 4) if gathers else fbs[0])[:H].float().cpu().numpy()
        camera.write_image(args.save_image, img)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(cloud, wl, cams_for, projs, vp, nf, frames):
    """The oracle (C restatement of the reference shaders; the reference has no CPU path of its own)
    timed on the host cores on a bounded sample: `frames` whole frames of the same workload."""
    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    aos = cloud.as_array()
    views = wl["views"]

    def one(step):
        cams = cams_for(step)
        t = time.perf_counter()
        for v in range(views):
            orc.render_frame(aos, True, cams[0], projs[0], vp, nf, render_cam=cams[v], render_proj=projs[v],
                             nthreads=cores)
        return time.perf_counter() - t

    t_first = one(0)
    if frames <= 0:
        frames = int(max(1, min(8, 15.0 // max(t_first, 1e-3))))
    times = [t_first] + [one(s) for s in range(1, frames)]
    sec = float(np.median(times))
    return {"value": 1.0 / sec, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d frame(s) of the same workload (orbit steps 0..%d), median, all %d host threads, "
                      "oracle/msplat_oracle.c via OpenMP row bands" % (len(times), len(times) - 1, cores),
            "sec_per_frame": sec, "gsplats_per_sec": wl["n"] / sec / 1e9}


if __name__ == "__main__":
    main()
