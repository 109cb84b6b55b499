#!/usr/bin/env python
"""tools/soak.py -- a few thousand frames over poses that make the frame change its kernel selection from frame to frame
(visible count above / below 2 M: chunk size of sort passes 1 and 2; heavy chunks of the column pass present / absent:
helper workgroups; pair counts above / below the scan-free limit), one frame at a time, with four frames in flight, and as one
rank of a row-sharded frame (block layout, band-culled sort), and with both eyes of a stereo pair in one chain of launches
(four in flight).

Large clouds render in two passes with occlusion feedback after a context's first frames (msplat_config.two_pass AUTO): most of
these frames do, with a share of the splats in the first pass that moves from frame to frame -- the pixels may not.
Every stateful shortcut of the frame is exercised across those switches: the self-cleaning group tables, the per-parity
minimum-key and heavy-chunk words, the host-mapped hints of an earlier frame.  Checked: every render of a pose is
bit-identical to the first render of that pose (same context kind), the on-device order checks stay at (0, 0), the visible
count and pair count of a pose never change, no overflow is reported.

  python tools/soak.py [--frames 3000] [--splats 2400000]          (GPU box; ~1 minute)
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3000)
    ap.add_argument("--splats", type=int, default=2_400_000)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    args = ap.parse_args(argv)
    import torch
    from splatapult_amd import GaussianCloud, SplatRenderer, camera, synthetic

    t0 = time.time()
    a = synthetic.generate_scene(args.splats, seed=0x50A4, full_sh=False)
    gc = GaussianCloud(GaussianCloud.Options(False, False))
    assert gc.FromAttributes(a["xyz"], a["f_dc"], None, a["opacity"], a["log_scale"], a["rot"])
    W, H = args.width, args.height
    proj = camera.perspective(camera.FOVY, W / H)
    vp, nf = [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR]
    inside = synthetic.scene_cameras(8)
    poses = [inside[0], inside[3], inside[5],                       # camera inside: V ~ 40 %, background splats = heavy chunks
             camera.orbit(30.0, 0.3), camera.orbit(45.0, 2.1),      # far outside: everything visible, tiny footprints
             camera.orbit(9.0, 1.0), camera.orbit(6.0, 4.0),        # near the objects
             camera.pose((0.0, 0.2, 0.0), 1.0, -1.2)]               # looking at the ground from inside: few visible
    print("scene built in %.1f s, %d poses" % (time.time() - t0, len(poses)), flush=True)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(7)
    order = rng.integers(0, len(poses), size=args.frames)
    failures = 0
    for kind in ("serial", "in flight", "band", "stereo in flight"):
        depth = 4 if "in flight" in kind else 1
        stereo = kind.startswith("stereo")
        r = SplatRenderer(device=0, frames_in_flight=depth)
        assert r.Init(gc, False, False), r.last_error()
        if kind == "band":        # rank 1 of 4 under the block layout, band-culled sort: virtual rows, V a fraction of the cloud's
            r.set_band_plan("block", (H + 31) // 32, 4, 1, block_rows=2, band_cull=True)
        Hpad = (H + 31) // 32 * 32           # the compositor writes whole bins
        fbs = [torch.zeros(((2 if stereo else 1) * Hpad, W, 4), dtype=torch.float32, device=dev) for _ in range(depth)]      # stereo: the eyes one above the other
        first, counts, two_pass_seen = {}, {}, 0
        t0 = time.time()
        pending = []                      # (pose index, framebuffer slot) of frames not yet checked
        for f, p in enumerate(order):
            slot = f % depth
            if len(pending) == depth:     # the slot's previous frame must be consumed before it is overwritten
                r.synchronize()
                for pp, ss in pending:
                    img = fbs[ss]
                    if pp not in first:
                        first[pp] = img.clone()
                    elif not torch.equal(first[pp], img):
                        failures += 1
                        print("MISMATCH %s frame %d pose %d: %d pixels differ" % (kind, f, pp, int((first[pp] != img).any(-1).sum())))
                pending = []
            r.Sort(poses[p], proj, vp, nf)
            if stereo:        # both eyes in one chain of launches (msplat_render_stereo), in the first eye's order
                eyes = [camera.translate_local(poses[p], dx=-0.032), camera.translate_local(poses[p], dx=+0.032)]
                r.RenderStereo(eyes, [proj, proj], vp, nf, out_ptrs=[fbs[slot].data_ptr(), fbs[slot][Hpad:].data_ptr()], pitch_bytes=W * 16)
            else:
                r.Render(poses[p], proj, vp, nf, out_ptr=fbs[slot].data_ptr(), pitch_bytes=W * 16)
            pending.append((int(p), slot))
            if f % 97 == 0:
                r.synchronize()
                vo = r.verify_order()
                st = r.stats()
                # (after a two-pass Render the pair count is that of its second pass, which depends on the share of the splats the
                #  feedback loop put into the first: only the visible count is a constant of the pose then)
                two = r.two_pass_info() is not None
                two_pass_seen += int(two)
                key = (st["sort_count"], -1 if two else st["pairs"])
                if vo != (0, 0):
                    failures += 1
                    print("ORDER %s frame %d pose %d: %s" % (kind, f, p, vo))
                if counts.setdefault((int(p), two), key) != key:
                    failures += 1
                    print("COUNTS %s frame %d pose %d: %s != %s" % (kind, f, p, key, counts[(int(p), two)]))
        r.synchronize()
        el = time.time() - t0
        print("%s: %d frames in %.1f s (%.0f frames/s incl. checks), poses seen %d, two-pass frames among the %d checked: %d, V/pairs per pose: %s"
              % (kind, args.frames, el, args.frames / el, len(first), (args.frames + 96) // 97, two_pass_seen,
                 {k[0]: v for k, v in sorted(counts.items()) if not k[1]}), flush=True)
    print("soak: %s" % ("OK" if failures == 0 else "%d FAILURES" % failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
