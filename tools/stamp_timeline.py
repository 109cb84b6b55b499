#!/usr/bin/env python3
"""usage (GPU box): MSPLAT_LIB_PATH=tools/bin/variants/libmsplat_stamps.so python tools/stamp_timeline.py [frames_in_flight=4] [frames=48]

What do frames in flight do to each other?  ("counters that say why".)  rocprofv3 serialises dispatches while
it collects counters and this pool's agents do not support PC sampling, so the library's DIAGNOSTIC build (-DMSPLAT_STAMPS) makes
every workgroup of the frame's kernels record {kernel, block, HW_ID, XCC_ID, start, end} (s_memrealtime, 10 ns) while a flag is on.
This script renders BASELINE configs[1] one frame at a time and with N frames in flight, reads the stamps of a window of frames
and prints, per kernel:

  wg_us             mean lifetime of a workgroup (persistent compositor waves: of the whole wave)
  x serial          the same in flight / one frame at a time: > 1 = the workgroup itself runs slower (memory system, issue slots)
  resident          workgroups of the kernel resident on the GPU, averaged over the window (sum of lifetimes / window)
  busy share        share of the window during which at least one workgroup of the kernel is resident

and for the whole GPU the time-averaged resident WAVES per SIMD and LDS per CU by kernel -- what the frames compete for."""
import ctypes as C
import math
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from splatapult_amd import SplatRenderer, _capi, camera, synthetic  # noqa: E402

NAMES = {1: "ws_upsweep<cull>", 2: "ws_downsweep<cull>", 3: "ws_upsweep", 4: "ws_downsweep", 5: "project_kernel", 6: "bin1_upsweep",
         7: "bin1_downsweep", 8: "radix_upsweep<pair>", 9: "radix_downsweep<pair>", 10: "tile_start_kernel (removed at the end of r6)", 11: "composite_kernel",
         12: "box_cull_kernel", 13: "radix_upsweep", 14: "radix_downsweep"}
# waves per workgroup and LDS bytes per workgroup in each frame mode (serial / in flight), from the launch code and tools/kres.sh
WAVES = {1: (16, 4), 2: (8, 4), 3: (16, 4), 4: (8, 4), 5: (1, 1), 6: (4, 4), 7: (4, 4), 8: (4, 4), 9: (4, 4), 10: (4, 4), 11: (1, 1)}
REC = np.dtype([("kid", "<u4"), ("blk", "<u4"), ("hwid", "<u4"), ("xcc", "<u4"), ("t0", "<u8"), ("t1", "<u8")])


def run(P, frames, cloud, W, H, L, pool=None):
    dev = torch.device("cuda:0")
    # STAMP_CU_PARTITION=off: every context's stream on every CU (the shims' default for 4 frames in flight: two per half, r6)
    r = SplatRenderer(device=0, fb_format="fp32", frames_in_flight=P, compositor_waves=pool,
                      cu_partition=False if os.environ.get("STAMP_CU_PARTITION") == "off" else None)
    assert r.Init(cloud, False, False), r.last_error()
    fbs = [torch.zeros((((H + 31) // 32) * 32, W, 4), dtype=torch.float32, device=dev) for _ in range(P)]
    proj = camera.perspective(camera.FOVY, W / H)
    vp, nf = [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR]

    poses = [camera.orbit(7.0, 2.0 * math.pi * k / 64.0) for k in range(64)]      # (a pose costs the host more than a frame's enqueue)

    def frame(s):
        c = poses[s % 64]
        r.Sort(c, proj, vp, nf)
        r.Render(c, proj, vp, nf, out_ptr=fbs[r.frame_slot].data_ptr(), pitch_bytes=W * 16)

    for s in range(300):
        frame(s)
    r.synchronize(); torch.cuda.synchronize()
    slots = L.msplat_debug_stamps(1, 23)
    assert slots > 0, slots
    import time
    t0 = time.perf_counter()
    for s in range(frames):
        frame(300 + s)
    r.synchronize(); torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / frames
    assert L.msplat_debug_stamps(0, 23) > 0
    buf = np.zeros(slots, REC)
    assert L.msplat_debug_stamps_read(buf.ctypes.data, buf.nbytes) == 0
    r.close()
    rec = buf[buf["t1"] != 0]
    return ms, rec


def table(rec, frames, P):
    t_lo, t_hi = rec["t0"].min(), rec["t1"].max()
    # drop the fill and the drain of the window: keep the middle 70 %
    a, b = t_lo + 0.15 * (t_hi - t_lo), t_hi - 0.15 * (t_hi - t_lo)
    win_us = (b - a) * 0.01
    out = {}
    for kid in sorted(set(rec["kid"].tolist())):
        rk = rec[rec["kid"] == kid]
        life = (rk["t1"] - rk["t0"]).astype(np.float64) * 0.01
        c0, c1 = np.clip(rk["t0"].astype(np.float64), a, b), np.clip(rk["t1"].astype(np.float64), a, b)
        inside = (c1 - c0) * 0.01
        # busy share: union of the intervals inside the window
        order = np.argsort(c0)
        s0, s1 = c0[order], c1[order]
        reach = np.maximum.accumulate(s1)
        gaps = np.maximum(0.0, s0[1:] - reach[:-1]).sum() + (s0[0] - a) + (b - reach[-1]) if len(s0) else (b - a)
        out[kid] = dict(n=len(rk), wg_us=float(life.mean()), wg_p90=float(np.percentile(life, 90)), resident=float(inside.sum() / win_us),
                        busy=float(1.0 - gaps / (b - a)))
    return out, win_us


def launches(rec, kid, nmin):
    """groups the workgroup records of one kernel into launches (ascending start, a block index occurs once per launch, starts of a
    launch lie within 60 us): per launch (first start, last start, last end, records)"""
    rk = rec[rec["kid"] == kid]
    rk = rk[np.argsort(rk["t0"])]
    open_, done = [], []
    for t0, t1, blk in zip(rk["t0"].tolist(), rk["t1"].tolist(), rk["blk"].tolist()):
        hit = None
        for c in reversed(open_):
            if blk not in c["blks"] and t0 - c["last"] < 6000:
                hit = c
                break
        if hit is None:
            hit = dict(first=t0, last=t0, end=t1, blks=set())
            open_.append(hit)
            if len(open_) > 12:
                done.append(open_.pop(0))
        hit["blks"].add(blk)
        hit["last"] = t0
        hit["end"] = max(hit["end"], t1)
    done += open_
    return [(c["first"], c["last"], c["end"], len(c["blks"])) for c in done if len(c["blks"]) >= nmin]


def launch_table(rec, label):
    print("## %s: per LAUNCH (median over the launches found): workgroups recorded, first-to-last workgroup START, first start to last END" % label)
    for kid in sorted(set(rec["kid"].tolist())):
        n_typ = np.bincount(rec["blk"][rec["kid"] == kid]).size
        ls = launches(rec, kid, max(2, n_typ // 3))
        if not ls:
            continue
        a = np.asarray(ls, np.float64)
        print("%-24s launches %4d  workgroups %6.0f  start spread %7.2f us (p90 %7.2f)  span %7.2f us (p90 %7.2f)"
              % (NAMES.get(kid, str(kid)), len(ls), np.median(a[:, 3]), 0.01 * np.median(a[:, 1] - a[:, 0]), 0.01 * np.percentile(a[:, 1] - a[:, 0], 90),
                 0.01 * np.median(a[:, 2] - a[:, 0]), 0.01 * np.percentile(a[:, 2] - a[:, 0], 90)))


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    L = C.CDLL(_capi.LIB_PATH)
    L.msplat_debug_stamps.argtypes = [C.c_int, C.c_uint32]
    L.msplat_debug_stamps_read.argtypes = [C.c_void_p, C.c_uint64]
    W, H = 1920, 1080
    cloud = synthetic.make_cloud(1_000_000, seed=0x5EED1234, full_sh=True, pos_sigma=1.5)
    ms1, rec1 = run(1, frames, cloud, W, H, L)
    t1, w1 = table(rec1, frames, 1)
    launch_table(rec1, "one frame at a time")
    variants = [("%d in flight" % P, None)] + [("%d in flight, compositor pool %s" % (P, w), int(w)) for w in sys.argv[3:]]
    for label, env in variants:
        msP, recP = run(P, frames, cloud, W, H, L, pool=env)
        tP, wP = table(recP, frames, P)
        print("# config 2 (1 M splats, 1920x1080), stamps build: one frame at a time %.4f ms/frame (%d records), %s %.4f ms/frame (%d records; "
              "hash-table losses expected: a few %%)" % (ms1, len(rec1), label, msP, len(recP)))
        print("%-24s | serial: %7s %8s %8s %6s | in flight: %7s %8s %8s %8s %6s" % ("kernel", "wg_us", "p90", "resident", "busy", "wg_us", "p90", "x serial", "resident", "busy"))
        wavesum = [0.0, 0.0]
        for kid in sorted(tP):
            s, f = t1.get(kid), tP[kid]
            name = NAMES.get(kid, str(kid))
            if s:
                print("%-24s | %15.2f %8.2f %8.1f %5.0f%% | %18.2f %8.2f %8.2f %8.1f %5.0f%%" % (name, s["wg_us"], s["wg_p90"], s["resident"], 100 * s["busy"], f["wg_us"], f["wg_p90"],
                                                                                           f["wg_us"] / s["wg_us"], f["resident"], 100 * f["busy"]))
                wv = WAVES.get(kid, (4, 4))
                wavesum[0] += s["resident"] * wv[0]
                wavesum[1] += f["resident"] * wv[1]
        print("# resident waves per SIMD (1024 SIMDs), time average over the window: serial %.2f, in flight %.2f" % (wavesum[0] / 1024.0, wavesum[1] / 1024.0))
        launch_table(recP, label)
        if not env:
            # a 1.5 ms slice of the in-flight window for offline viewing (kernel, block, cu, start, end in units of 10 ns)
            mid = (recP["t0"].min() + recP["t1"].max()) // 2
            sl = recP[(recP["t0"] >= mid) & (recP["t0"] < mid + 150000)]
            cu = ((sl["xcc"] & 7).astype(np.uint16) << 6) | (((sl["hwid"] >> 13) & 7).astype(np.uint16) << 4) | ((sl["hwid"] >> 8) & 15).astype(np.uint16)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r06_stamps_slice.npz"), kid=sl["kid"].astype(np.uint8), blk=sl["blk"].astype(np.uint32),
                                cu=cu, simd=((sl["hwid"] >> 4) & 3).astype(np.uint8), t0=(sl["t0"] - mid).astype(np.uint32), t1=(sl["t1"] - mid).astype(np.uint32))
    # per-CU crowding in flight: how many workgroups of ANY kernel share a CU with a starting compositor / chain workgroup
    cu = (recP["xcc"].astype(np.int64) & 7) * 64 + ((recP["hwid"].astype(np.int64) >> 8) & 0xF) + (((recP["hwid"].astype(np.int64) >> 13) & 0x7) << 4)
    print("# distinct (xcc, se, cu) ids seen: %d" % len(np.unique(cu)))


if __name__ == "__main__":
    main()
