#!/usr/bin/env python
"""Predicted multi-GPU frame from ONE GPU: every rank's share of a workload, rendered in turn, for several bin-row layouts.

  python tools/band_table.py --workload cfg4 --world 8 --out gpurun_out/r03_cfg4_bands.json

For each layout (contiguous bands, interleaved rows, blocks of k rows dealt round-robin) and each rank it sets the rank's band
(msplat_band_plan + band-restricted cull), renders serial frames on one stream and records V, the (splat, bin) pairs, the stage
times and the frame time; then the max over the ranks (= the frame a node of `world` GPUs would take before the gather), the
messages / bytes that reach rank 0, and the gather modelled at 153 GB/s per xGMI link (every rank has its own link to rank 0,
MI355X_MICROARCH.md).  The exchange itself needs the real node; this table is what chooses the layout bench.py defaults to.

--fif P (r4): every rank's share rendered the way bench.py times it -- P frames in flight (P contexts, one shared cloud, the
in-flight kernel selection, 1280 compositor waves), blocks of --block frames between two synchronisations (bench.py's --steps;
20 under the driver) -- and reported as frames/s: max over the ranks of the block time = what a node of `world` GPUs does
per block before the exchange.  --worlds 2,4,8 tabulates several rank counts (layout "auto" = bench.py's default for each)."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
XGMI_LINK = 153e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg4")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--layouts", default="contiguous,interleaved,block:2,block:4,block:8")
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--out", default=None)
    ap.add_argument("--fif", type=int, default=1, help="frames in flight per rank (bench.py's default protocol: 4)")
    ap.add_argument("--block", type=int, default=20, help="--fif > 1: frames per timed block (bench.py --steps)")
    ap.add_argument("--worlds", default=None, help="--fif > 1: comma-separated rank counts (default: --world)")
    args = ap.parse_args()
    if args.fif > 1:
        return main_fif(args)
    import torch
    import bench
    from splatapult_amd import SplatRenderer, camera, synthetic, _capi
    from splatapult_amd.dist import owned_rows, row_runs

    wl = bench.WORKLOADS[args.workload]
    W, H, G = wl["W"], wl["H"], args.world
    cloud = synthetic.make_cloud(wl["n"], seed=wl["seed"], full_sh=True, pos_sigma=wl["pos_sigma"])
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    r = SplatRenderer(device=0, fb_format=wl["fb"], stream=stream.cuda_stream, enable_timing=1, frames_in_flight=1)
    assert r.Init(cloud, False, False), r.last_error()
    T = _capi.lib().msplat_tile_size()
    R = (H + T - 1) // T
    bpp = 8 if wl["fb"] == "fp16" else 16
    fb = torch.zeros((R * T, W, 4), dtype=torch.float16 if bpp == 8 else torch.float32, device=dev)
    proj = camera.perspective(camera.FOVY, W / H)
    vp, nf = [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR]

    def frames(n, first=0):
        for s in range(n):
            cam = camera.orbit(wl["cam_z"], 2.0 * math.pi * ((first + s) % 64) / 64.0)
            r.Sort(cam, proj, vp, nf)
            r.Render(cam, proj, vp, nf, out_ptr=fb.data_ptr(), pitch_bytes=W * bpp)

    def measure():
        frames(6)
        torch.cuda.synchronize()
        r.timings()
        blocks = []
        for b in range(3):          # the fastest of three blocks: the runtime's one-off stalls (pool growth) must not count
            t0 = time.perf_counter()
            frames(args.frames, 6 + b * args.frames)
            torch.cuda.synchronize()
            blocks.append(1e3 * (time.perf_counter() - t0) / args.frames)
        ms = min(blocks)
        tm = r.timings()
        st = r.stats()
        return dict(frame_ms=ms, sort_ms=tm["sort_total"], project_ms=tm["project"], binning_ms=tm["binning"],
                    composite_ms=tm["composite"], V=st["sort_count"], pairs=st["pairs"])

    r.set_band(1, 0)
    frames(300)                 # get past the runtime's start-up stall (bench.py --prewarm)
    whole = measure()
    out = {"workload": wl["desc"], "world": G, "bin_rows": R, "bin_px": T, "single_gpu": whole, "layouts": {},
           "xgmi_link_GBps": XGMI_LINK / 1e9, "frames_per_point": args.frames}
    print("single GPU: frame %.3f ms  V %d  pairs %d" % (whole["frame_ms"], whole["V"], whole["pairs"]))
    for lay in args.layouts.split(","):
        kind, k = (lay.split(":")[0], int(lay.split(":")[1])) if ":" in lay else (lay, 1)
        if kind == "block" and k == 1:
            kind = "interleaved"
        ranks = []
        for g in range(G):
            r.set_band_plan(kind, R, G, g, block_rows=k, band_cull=True)
            m = measure()
            rows = owned_rows(kind, R, G, g, k)
            runs = row_runs(rows)
            px_rows = sum(min(T * c, max(0, H - t * T)) for t, c in runs)
            m.update(rows=len(rows), messages=len(runs) if g else 0, bytes=px_rows * W * bpp if g else 0)
            ranks.append(m)
        mx = max(x["frame_ms"] for x in ranks)
        gather_ms = 1e3 * max(x["bytes"] for x in ranks) / XGMI_LINK       # every rank's rows on its own link, concurrently
        out["layouts"][lay] = {
            "ranks": ranks, "max_rank_frame_ms": mx, "mean_rank_frame_ms": float(np.mean([x["frame_ms"] for x in ranks])),
            "max_V_frac": max(x["V"] for x in ranks) / max(1, whole["V"]),
            "messages_into_rank0": sum(x["messages"] for x in ranks), "bytes_into_rank0": sum(x["bytes"] for x in ranks),
            "modelled_gather_ms": gather_ms, "predicted_frame_ms_overlapped": max(mx, gather_ms),
            "predicted_frame_ms_serialised": mx + gather_ms,
            "speedup_vs_single_gpu": whole["frame_ms"] / max(mx, gather_ms)}
        L = out["layouts"][lay]
        print("%-12s max-rank frame %.3f ms (mean %.3f)  max V %.2f  msgs %d  gather %.3f ms  -> x%.2f" % (
            lay, mx, L["mean_rank_frame_ms"], L["max_V_frac"], L["messages_into_rank0"], gather_ms, L["speedup_vs_single_gpu"]))
        for g, x in enumerate(ranks):
            print("    rank %d: frame %.3f  sort %.3f  proj %.3f  bin %.3f  comp %.3f  V %d  pairs %d  rows %d" % (
                g, x["frame_ms"], x["sort_ms"], x["project_ms"], x["binning_ms"], x["composite_ms"], x["V"], x["pairs"], x["rows"]))
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


def main_fif(args):
    import torch
    import bench
    from splatapult_amd import SplatRenderer, camera, synthetic, _capi
    from splatapult_amd.dist import owned_rows, row_runs

    wl = bench.WORKLOADS[args.workload]
    W, H, P = wl["W"], wl["H"], args.fif
    cloud = synthetic.make_cloud(wl["n"], seed=wl["seed"], full_sh=True, pos_sigma=wl["pos_sigma"])
    dev = torch.device("cuda", 0)
    r = SplatRenderer(device=0, fb_format=wl["fb"], frames_in_flight=P)
    assert r.Init(cloud, False, False), r.last_error()
    T = _capi.lib().msplat_tile_size()
    R = (H + T - 1) // T
    bpp = 8 if wl["fb"] == "fp16" else 16
    fbs = [torch.zeros((R * T, W, 4), dtype=torch.float16 if bpp == 8 else torch.float32, device=dev) for _ in range(P)]
    proj = camera.perspective(camera.FOVY, W / H)
    vp, nf = [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR]
    poses = [camera.orbit(wl["cam_z"], 2.0 * math.pi * k / 64.0) for k in range(64)]

    def frames(n, first=0):
        for s in range(n):
            cam = poses[(first + s) % 64]
            r.Sort(cam, proj, vp, nf)
            r.Render(cam, proj, vp, nf, out_ptr=fbs[r.frame_slot].data_ptr(), pitch_bytes=W * bpp)

    def measure():
        frames(2 * P + 4)
        r.synchronize()
        blocks = []
        nblk = max(5, min(40, int(0.15 / max(1e-4, args.block * 2e-4))))
        for b in range(nblk):
            t0 = time.perf_counter()
            frames(args.block, 11 + b * args.block)
            r.synchronize()                       # (issued by the contexts' worker threads, then finished on the GPU)
            blocks.append(time.perf_counter() - t0)
        med = float(np.median(blocks))
        st = r.stats()
        return dict(block_ms=1e3 * med, frames_per_sec=args.block / med, ms_per_frame=1e3 * med / args.block,
                    V=st["sort_count"], pairs=st["pairs"])

    r.set_band(1, 0)
    frames(400)
    r.synchronize()
    whole = measure()
    out = {"workload": wl["desc"], "frames_in_flight": P, "block_frames": args.block, "bin_rows": R, "bin_px": T,
           "single_gpu": whole, "worlds": {}, "xgmi_link_GBps": XGMI_LINK / 1e9,
           "protocol": "bench.py's: blocks of %d frames, %d in flight, synchronise on both sides; median block" % (args.block, P)}
    print("single GPU: %.0f frames/s (%.3f ms per frame in blocks of %d)  V %d" % (whole["frames_per_sec"], whole["ms_per_frame"], args.block, whole["V"]))
    for G in [int(x) for x in (args.worlds or str(args.world)).split(",")]:
        out["worlds"][str(G)] = {}
        for lay in args.layouts.split(","):
            if lay == "auto":
                lay = "block:%d" % max(1, R // (2 * G))
            if lay == "weighted":
                # r6: contiguous bands, rank 0 (the gather's root: sends nothing) weighted by the linear cost model of
                # msplat_band_root_weight, calibrated on THIS GPU: a rank with r rows takes fixed + per_row r ms per frame (from
                # the whole frame and one equal band in the middle of the image); the others also move r rows over their link
                r.set_band_layout((R - R // G) // 2, R // G, R // G, max(R, 1), band_cull=True)
                part = measure()
                per_row = max(1e-6, (whole["ms_per_frame"] - part["ms_per_frame"]) / max(1, R - R // G))
                fixed = max(0.0, part["ms_per_frame"] - per_row * (R // G))
                pct = _capi.band_root_weight(R, G, fixed, per_row, T * W * bpp, XGMI_LINK / 1e9, True)
                print("G = %d  weighted: calibration fixed %.4f ms + %.5f ms per bin row, %.0f KB per row over the link -> root weight %d %%" % (
                    G, fixed, per_row, T * W * bpp / 1024.0, pct))
                lay = "weighted:%d" % pct
            kind, k = (lay.split(":")[0], int(lay.split(":")[1])) if ":" in lay else (lay, 1)
            if kind == "block" and k == 1:
                kind = "interleaved"
            ranks = []
            for g in range(G):
                r.set_band_plan(kind, R, G, g, block_rows=k, band_cull=True)
                m = measure()
                rows = owned_rows(kind, R, G, g, k)
                runs = row_runs(rows)
                px_rows = sum(min(T * c, max(0, H - t * T)) for t, c in runs)
                m.update(rows=len(rows), messages=len(runs) if g else 0, bytes=px_rows * W * bpp if g else 0)
                ranks.append(m)
            slow = max(ranks, key=lambda x: x["block_ms"])
            gather_ms = 1e3 * max(x["bytes"] for x in ranks) / XGMI_LINK        # per frame, every rank on its own link
            fps = slow["frames_per_sec"]
            fps_g = min(fps, 1e3 / gather_ms) if gather_ms > 0 else fps           # the gather of frame k overlaps frame k + 1's compute
            out["worlds"][str(G)][lay] = {
                "ranks": ranks, "max_rank_block_ms": slow["block_ms"], "predicted_frames_per_sec": fps,
                "predicted_frames_per_sec_with_modelled_gather": fps_g, "modelled_gather_ms_per_frame": gather_ms,
                "speedup_vs_single_gpu": fps_g / whole["frames_per_sec"], "scaling_efficiency": fps_g / whole["frames_per_sec"] / G,
                # r5: the same with the rows of an fp32 target on the wire as fp16 (msplat_band_exchange, MSPLAT_EXCHANGE_WIRE_FP16)
                "predicted_frames_per_sec_with_fp16_wire": (min(fps, 2e3 / gather_ms) if (gather_ms > 0 and bpp == 16) else fps_g),
                "max_V_frac": max(x["V"] for x in ranks) / max(1, whole["V"])}
            L = out["worlds"][str(G)][lay]
            print("G = %d  %-12s slowest rank %.0f frames/s (ranks %s)  gather %.3f ms  -> %.0f frames/s = x%.2f (efficiency %.2f)" % (
                G, lay, fps, " ".join("%.0f" % x["frames_per_sec"] for x in ranks), gather_ms, fps_g,
                L["speedup_vs_single_gpu"], L["scaling_efficiency"]))
    r.set_band(1, 0)
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
