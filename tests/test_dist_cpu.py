"""N > 1 path on CPU: two / three gloo ranks shard an image by bin rows -- contiguous bands, interleaved rows or blocks of
rows dealt round-robin (the ownership rules of msplat_band_plan) -- and rank 0 receives the foreign rows straight into its own framebuffer with
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


def _worker(rank, world, port, W, H, q, layout, block_rows):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from splatapult_amd import synthetic
    from splatapult_amd import _capi
    from splatapult_amd.dist import BandGather, owned_rows, row_runs
    from tests import scenes
    cloud = synthetic.make_cloud(1500, seed=5, log_scale_mean=-3.0)
    cam, proj, vp, nf = scenes.default_view(W, H)
    res = orc.render_frame(cloud.as_array(), True, cam, proj, vp, nf, want_image=False, want_splats=True)
    T = 32                                   # msplat_tile_size()
    tiles_y = (H + T - 1) // T
    fb = np.zeros((tiles_y * T, W, 4), np.float32)
    # this rank "renders" only its own bin rows; the Python ownership rule must be the library's (msplat_band_plan)
    mine = owned_rows(layout, tiles_y, world, rank, block_rows)
    assert mine == _capi.band_rows(*_capi.band_plan(layout, tiles_y, world, rank, block_rows), rows_full=tiles_y)
    for t in mine:
        y0, y1 = t * T, min(t * T + T, H)
        fb[y0:y1] = orc.composite(res["splats"], W, H, row0=y0, row1=y1)[y0:y1]
    g = BandGather(tiles_y, W, torch.float32, torch.device("cpu"), rank, world, tile=T, layout=layout, block_rows=block_rows)
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
    nruns = [len(row_runs(owned_rows(layout, tiles_y, world, r, block_rows))) for r in range(world)]
    assert len(g.plan) == (sum(nruns[1:]) if rank == 0 else nruns[rank])
    if layout == "contiguous":
        assert all(n <= 1 for n in nruns)                        # one message per rank and frame
    dist.barrier()
    dist.destroy_process_group()


def _run(world, W, H, layout="interleaved", block_rows=1):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, q, layout, block_rows)) for r in range(world)]
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


def test_contiguous_bands_one_message_per_rank():
    assert _run(3, 96, 208, layout="contiguous")          # 7 rows: 2 + 2 + 3


def test_block_interleaved_bands():
    assert _run(2, 64, 300, layout="block", block_rows=2)   # 10 rows, blocks of 2: ragged last row, 5 blocks over 2 ranks


def test_weighted_contiguous_bands_two_and_three_ranks():
    """MSPLAT_BANDS_ROOT_WEIGHTED (r6): contiguous bands, rank 0 -- the gather's root, which sends nothing -- takes more rows.
    Rows disjoint, cover, one message per rank, the gathered frame bit-exact"""
    assert _run(2, 160, 300, layout="weighted", block_rows=220)      # 10 rows: 7 + 3
    assert _run(3, 96, 208, layout="weighted", block_rows=300)       # 7 rows: 4 + 2 + 1 (largest remainders)


def test_weighted_band_plan_and_the_cost_model_behind_the_root_weight():
    sys.path.insert(0, ROOT)
    import ctypes as C
    from splatapult_amd import _capi
    from splatapult_amd.dist import owned_rows, weighted_bounds
    L = _capi.lib()
    rng = np.random.default_rng(11)
    # general weights: bounds ascend, cover [0, R], proportional within one row, nobody with a positive weight is left empty
    # while a larger band can spare a row; the Python restatement agrees
    for R in (0, 1, 5, 34, 128, 256):
        for world in (1, 2, 3, 8):
            for _ in range(6):
                w = rng.uniform(0.2, 4.0, world).astype(np.float32)
                b = _capi.band_plan_weighted(R, list(w))
                assert b == weighted_bounds(R, w), (R, world, w)
                assert b[0] == 0 and b[-1] == R and all(b[i] <= b[i + 1] for i in range(world))
                ideal = R * w.astype(np.float64) / w.astype(np.float64).sum()
                rows = np.diff(b)
                assert np.all(np.abs(rows - ideal) < 1.0 + 1e-9) or R < world or rows.min() == 1
                if R >= 2 * world:
                    assert rows.min() >= 1
    # the one-parameter family behind msplat_band_plan: weight 100 = equal bands up to rounding, larger = more rows for rank 0
    for R, world in ((128, 2), (128, 8), (34, 4)):
        prev0 = -1
        for pct in (100, 150, 250, 400, 1000):
            seen = []
            for rank in range(world):
                first, count, block, stride = _capi.band_plan("weighted", R, world, rank, pct)
                rows = _capi.band_rows(first, count, block, stride, rows_full=R)
                assert rows == owned_rows("weighted", R, world, rank, pct) and len(rows) == count
                assert rows == list(range(rows[0], rows[0] + count)) if rows else True     # contiguous
                seen += rows
                if rank == 0:
                    assert count >= prev0
                    prev0 = count
            assert sorted(seen) == list(range(R))
        eq = [_capi.band_plan("weighted", R, world, r, 100)[1] for r in range(world)]
        assert max(eq) - min(eq) <= 1
    o = [C.c_int32() for _ in range(4)]
    assert L.msplat_band_plan(3, 8, 2, 0, 0, *[C.byref(x) for x in o]) == _capi.ERR_INVALID_ARG       # weight must be >= 1
    assert L.msplat_band_plan_weighted(8, 2, None, None) == _capi.ERR_INVALID_ARG
    bad = (C.c_float * 2)(1.0, -1.0)
    out = (C.c_int32 * 3)()
    assert L.msplat_band_plan_weighted(8, 2, bad, out) == _capi.ERR_INVALID_ARG
    zero = (C.c_float * 2)(0.0, 0.0)
    assert L.msplat_band_plan_weighted(8, 2, zero, out) == _capi.ERR_INVALID_ARG
    # cost model (msplat_band_root_weight): BASELINE configs[3] -- 128 bin rows of 4096 x 32 px x 16 B = 2 MiB, 153 GB/s per xGMI
    # link, a rank with r rows computes 0.046 + 0.00564 r ms (profiles/r05_cfg4_bands_fif4.json: 1 GPU 0.768 ms, 2 GPUs 0.407 ms)
    row_bytes = 4096 * 32 * 16
    for world, overlap in ((2, True), (2, False), (4, True), (8, True)):
        pct = _capi.band_root_weight(128, world, 0.046, 0.00564, row_bytes, 153.0, overlap)
        b = _capi.band_plan_weighted(128, [float(pct)] + [100.0] * (world - 1))
        rows = np.diff(b)

        def cost(r, root):
            comp, link = 0.046 + 0.00564 * r, r * row_bytes / 153e6
            return comp if root else (max(comp, link) if overlap else comp + link)
        t = max(cost(rows[i], i == 0) for i in range(world))
        t_equal = max(cost(128 // world, i == 0) for i in range(world))
        t_single = 0.046 + 0.00564 * 128
        assert rows[0] > rows[1] and t < t_equal
        assert t < t_single, (world, overlap, t, t_single)          # N GPUs beat one GPU on paper (with equal bands 2 do not)
        if world == 2:
            assert t_equal > t_single
        if world == 8:
            assert t_single / t >= 3.5
    assert _capi.band_root_weight(128, 1, 0.046, 0.00564, row_bytes) == 100


def test_band_plan_partitions_every_row_exactly_once():
    """msplat_band_plan / msplat_set_band_layout arithmetic (host only): for every layout the ranks' rows are disjoint,
    cover 0..R-1, and agree with the Python restatement the gather plan uses"""
    sys.path.insert(0, ROOT)
    from splatapult_amd import _capi
    from splatapult_amd.dist import owned_rows
    for R in (0, 1, 7, 34, 128, 256):
        for world in (1, 2, 3, 8, 9):
            for kind, k in (("contiguous", 1), ("interleaved", 1), ("block", 2), ("block", 5), ("block", 32)):
                seen = []
                for rank in range(world):
                    first, count, block, stride = _capi.band_plan(kind, R, world, rank, k)
                    rows = _capi.band_rows(first, count, block, stride, rows_full=R)
                    assert len(rows) == count and rows == owned_rows(kind, R, world, rank, k), (R, world, kind, k, rank)
                    assert block >= 1 and stride >= block
                    seen += rows
                assert sorted(seen) == list(range(R)), (R, world, kind, k)
    # argument checks
    import ctypes as C
    L = _capi.lib()
    o = [C.c_int32() for _ in range(4)]
    refs = [C.byref(x) for x in o]
    assert L.msplat_band_plan(0, 8, 0, 0, 1, *refs) == _capi.ERR_INVALID_ARG
    assert L.msplat_band_plan(0, 8, 2, 2, 1, *refs) == _capi.ERR_INVALID_ARG
    assert L.msplat_band_plan(2, 8, 2, 0, 0, *refs) == _capi.ERR_INVALID_ARG
    assert L.msplat_band_plan(7, 8, 2, 0, 1, *refs) == _capi.ERR_INVALID_ARG
    assert L.msplat_set_band_layout(None, 0, 0, 1, 1) == _capi.ERR_INVALID_ARG
    assert L.msplat_group_create(None, None, 0, None) == _capi.ERR_INVALID_ARG
    g = C.c_void_p()
    assert L.msplat_group_create(C.byref(g), None, 2, None) == _capi.ERR_INVALID_ARG and not g.value
    assert L.msplat_group_sort(None, None, None, None, None) == _capi.ERR_INVALID_ARG
    L.msplat_group_destroy(None)
