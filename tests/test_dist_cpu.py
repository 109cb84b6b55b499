"""N > 1 path on CPU: two / three gloo ranks shard an image by interleaved bin rows (the exact ownership
rule of msplat_set_band) and rank 0 receives the foreign rows straight into its own framebuffer with
splatapult_amd.dist.BandGather (grouped send/recv, no pack or unpack copy) -- the one exchange step of the
multi-GPU path.  The per-band pixels come from the oracle (no GPU here)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from splatapult_amd import synthetic
    from splatapult_amd.dist import BandGather
    from tests import scenes
    cloud = synthetic.make_cloud(1500, seed=5, log_scale_mean=-3.0)
    cam, proj, vp, nf = scenes.default_view(W, H)
    res = orc.render_frame(cloud.as_array(), True, cam, proj, vp, nf, want_image=False, want_splats=True)
    T = 32                                   # msplat_tile_size()
    tiles_y = (H + T - 1) // T
    fb = np.zeros((tiles_y * T, W, 4), np.float32)
    # this rank "renders" only its own tile rows (rows of tile t with t % world == rank)
    for t in range(rank, tiles_y, world):
        y0, y1 = t * T, min(t * T + T, H)
        fb[y0:y1] = orc.composite(res["splats"], W, H, row0=y0, row1=y1)[y0:y1]
    g = BandGather(tiles_y, W, torch.float32, torch.device("cpu"), rank, world, tile=T)
    own = fb.copy()
    t_fb = torch.from_numpy(fb)
    for frame in range(2):                   # the object is reused frame after frame
        out = g(t_fb)
    if rank == 0:
        full = orc.composite(res["splats"], W, H)
        assert out.data_ptr() == t_fb.data_ptr()               # rank 0's framebuffer IS the final image
        q.put(bool(np.array_equal(out.numpy()[:H], full)))
    else:
        assert out is None
        assert np.array_equal(fb, own)                          # senders' framebuffers are untouched
    assert len(g.plan) == (sum(len(range(s, tiles_y, world)) for s in range(1, world)) if rank == 0
                           else len(range(rank, tiles_y, world)))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, W, H):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return ok


def test_two_ranks_reassemble_the_frame_bit_exact():
    assert _run(2, 160, 120)        # 120 rows = 3.75 tiles: ragged last tile row


def test_three_ranks_uneven_bands():
    assert _run(3, 96, 208)         # 7 tile rows over 3 ranks: 3 + 2 + 2
