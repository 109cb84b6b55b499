#!/usr/bin/env python
"""tools/two_pass_fuzz.py [--cases 150] [--seed 1] -- random clouds, cameras, viewports, row-band plans, targets and shares of the
splats in the first pass: a two-pass Render (msplat_config.two_pass = ON) must give the pixels of the single pass bit for bit.
GPU box only.  Prints one line per failure and a summary; exit code 1 on any mismatch."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from splatapult_amd import SplatRenderer, _capi
    from tests import scenes
    rng = np.random.default_rng(args.seed)
    T = _capi.lib().msplat_tile_size()
    bad = 0
    two_pass_frames = 0
    for case in range(args.cases):
        n = int(rng.choice([3000, 20000, 80000, 300000]))
        ls = float(rng.uniform(-4.6, -2.2))
        full = bool(rng.integers(0, 2))
        cloud = scenes.synth_cloud(n, 1000 + case, full_sh=full, log_scale_mean=ls) if rng.random() < 0.85 else \
            scenes.cloud_from_attrs(scenes.hard_attrs(min(n, 20000), 2000 + case))
        W, H = int(rng.integers(65, 1500)), int(rng.integers(33, 900))
        z = float(rng.choice([0.3, 1.0, 3.0, 6.0, 12.0]))
        fb = "fp16" if rng.random() < 0.25 else "fp32"
        share = float(rng.choice([1.0 / 256.0, 0.02, 0.1, 0.25, 0.5, 0.75, 1.0]))
        kw = dict(device=0, fb_format=fb, spatial_order=int(rng.choice([_capi.SPATIAL_AUTO, _capi.SPATIAL_ON, _capi.SPATIAL_OFF])))
        a = SplatRenderer(two_pass=_capi.TWO_PASS_OFF, **kw)
        b = SplatRenderer(two_pass=_capi.TWO_PASS_ON, **kw)
        assert a.Init(cloud, False, False) and b.Init(cloud, False, False), a.last_error() + b.last_error()
        b.two_pass_state(share)
        tiles_y = (H + T - 1) // T
        band = None
        if rng.random() < 0.35 and tiles_y >= 3:
            world = int(rng.integers(2, min(4, tiles_y) + 1))
            band = (str(rng.choice(["contiguous", "interleaved", "block"])), world, int(rng.integers(0, world)), int(rng.integers(1, 4)), bool(rng.integers(0, 2)))
            for r in (a, b):
                r.set_band_plan(band[0], tiles_y, band[1], band[2], block_rows=band[3], band_cull=band[4])
        for k in range(2):
            cam, proj, vp, nf = scenes.default_view(W, H, z=z, yaw=float(rng.uniform(0, 6.28)))
            a.Sort(cam, proj, vp, nf); b.Sort(cam, proj, vp, nf)
            ia, ib = a.Render(cam, proj, vp, nf), b.Render(cam, proj, vp, nf)
            if not np.array_equal(ia, ib):
                bad += 1
                print("MISMATCH case %d frame %d: n %d ls %.2f full %s %dx%d z %.1f %s share %.4f band %s: %d values differ, max %.3g"
                      % (case, k, n, ls, full, W, H, z, fb, share, band, int((ia != ib).sum()), float(np.abs(ia.astype(np.float32) - ib.astype(np.float32)).max())))
        two_pass_frames += b.two_pass_state(share)[0]
        a.close(); b.close()
    print("two-pass fuzz: %d cases, %d two-pass frames, %d mismatches" % (args.cases, two_pass_frames, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
