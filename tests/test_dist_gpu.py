"""The RCCL code path of the multi-GPU exchange on a ONE-GPU box (-m gpu): splatapult_amd.dist.BandGather over the
"nccl" backend (= RCCL on ROCm).  The driver's 8-GPU run is the real measurement; these tests only make sure the
NCCL branch (grouped isend/irecv of device rows, stream hand-off) has executed at least once before it.

  * one rank: RCCL initialises, a collective runs, BandGather is a no-op that returns the framebuffer;
  * two ranks sharing device 0: the real exchange.  RCCL may refuse two ranks on one device ("Duplicate GPU");
    that refusal is reported as a skip with RCCL's own message, not hidden."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        t = torch.full((4,), float(rank + 1), device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        assert float(t[0].item()) == world * (world + 1) / 2
        from splatapult_amd.dist import BandGather
        T, W, tiles_y = 32, 96, 7                       # 7 bin rows: uneven bands
        fb = torch.zeros((tiles_y * T, W, 4), dtype=torch.float32, device=dev)
        rows = fb.view(tiles_y, T, W, 4)
        for tr in range(rank, tiles_y, world):
            rows[tr] = float(100 * (rank + 1) + tr)     # what this rank "rendered"
        g = BandGather(tiles_y, W, torch.float32, dev, rank, world, tile=T)
        for _ in range(3):                              # reused frame after frame
            out = g(fb)
        torch.cuda.synchronize()
        if rank == 0:
            got = out.view(tiles_y, T, W, 4)[:, 0, 0, 0].cpu().numpy()
            want = np.array([100 * (tr % world + 1) + tr for tr in range(tiles_y)], np.float32)
            q.put(("ok", bool(np.array_equal(got, want)) and out.data_ptr() == fb.data_ptr()))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                               # noqa: BLE001 -- report RCCL's own words to the parent
        if rank == 0:
            q.put(("error", "%s: %s" % (type(e).__name__, e)))
        raise


def _run(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        kind, val = q.get(timeout=150)
    except Exception:                                    # noqa: BLE001
        kind, val = "error", "timeout waiting for the RCCL ranks"
    for p in procs:
        p.join(30)
        if p.is_alive():
            p.kill()
    return kind, val


def test_rccl_single_rank_group_and_noop_gather():
    kind, val = _run(1)
    assert kind == "ok" and val, val


def test_rccl_two_ranks_on_one_device_exchange_bands():
    kind, val = _run(2)
    if kind == "error" and any(s in val.lower() for s in ("duplicate gpu", "invalid usage", "invalid device ordinal")):
        pytest.skip("RCCL refuses two ranks on one device here: " + val[:300])
    assert kind == "ok" and val, val
