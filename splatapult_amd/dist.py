"""Multi-GPU tile-row sharding (one process per GPU, torch.distributed = RCCL over xGMI on ROCm).

The path shards by screen tiles: rank g of G renders a set of bin rows (rows of msplat_tile_size() = 32
pixels) given by a layout -- "contiguous" bands [g R / G, (g+1) R / G) (the north star's "tiles row-sharded"), "interleaved"
rows (t % G == g) or "block": blocks of k rows dealt round-robin (msplat_band_plan; per-pixel results are bit-identical
to single-GPU rendering for every layout because every bin still sees its splats in global depth order).  The only
exchange is the final row gather to rank 0, and it moves every band exactly once:

  * every rank renders into a full-size framebuffer (it writes only its own bin rows);
  * rank 0's framebuffer IS the final image: it posts one receive per RUN of consecutive foreign bin rows, straight into
    the run's place (a run is one contiguous block of pixels), and the other ranks send their runs from where the
    compositor left them -- no pack, no staging buffer, no unpack.  Contiguous bands: ONE message per rank and frame;
  * all of a frame's sends/receives are issued as ONE group (ncclGroupStart/End via batch_isend_irecv), so
    each rank's band travels over its own direct xGMI link concurrently (7 links into rank 0), not around a ring.

No collective is used anywhere else.  (One process driving several GPUs uses msplat_group_* instead: the other devices'
compositors store into device 0's framebuffer through the peer mapping and there is no message at all.)"""
import torch.distributed as dist


def owned_rows(kind, tiles_y, world, rank, block_rows=1):
    """the bin rows rank `rank` owns under the layout (the ownership rule of msplat_band_plan, restated in Python so
    that the gather plan needs no library call): ascending list"""
    if kind == "contiguous":
        return list(range((tiles_y * rank) // world, (tiles_y * (rank + 1)) // world))
    if kind == "weighted":         # contiguous bands, rank 0 weighted block_rows percent of another rank (MSPLAT_BANDS_ROOT_WEIGHTED)
        b = weighted_bounds(tiles_y, [float(int(block_rows))] + [100.0] * (world - 1))
        return list(range(b[rank], b[rank + 1]))
    k = 1 if kind == "interleaved" else int(block_rows)
    assert kind in ("interleaved", "block") and k >= 1
    return [t for t in range(tiles_y) if (t // k) % world == rank]


def weighted_bounds(tiles_y, weights):
    """msplat_band_plan_weighted restated (largest remainders; nobody with a positive weight stays empty while a larger band can
    spare a row): bounds[i] .. bounds[i + 1] = the bin rows of rank i.  float32 weights like the C ABI's"""
    import numpy as np
    w = np.asarray(weights, np.float32).astype(np.float64)
    x = tiles_y * w / w.sum()
    rows = np.floor(x).astype(np.int64)
    frac = x - np.floor(x)
    for _ in range(int(tiles_y - rows.sum())):
        i = int(np.argmax(frac))               # (first of equals, like the library)
        rows[i] += 1
        frac[i] = -1.0
    for i in range(len(w)):
        if rows[i] == 0 and w[i] > 0:
            j = int(np.argmax(rows))
            if rows[j] >= 2:
                rows[j] -= 1
                rows[i] = 1
    return [0] + [int(v) for v in np.cumsum(rows)]


def row_runs(rows):
    """[(first_row, count)] of the maximal runs of consecutive rows"""
    runs = []
    for t in rows:
        if runs and runs[-1][0] + runs[-1][1] == t:
            runs[-1] = (runs[-1][0], runs[-1][1] + 1)
        else:
            runs.append((t, 1))
    return runs


class BandGather:
    """Gathers the owned bin rows of every rank's full-size framebuffer into rank `dst`'s framebuffer.

    fb layout on every rank: (tiles_y * tile, W, 4) (height padded to a multiple of the bin size);
    rank g has written its own bin rows only.  On `dst` the call returns fb itself, completed."""

    def __init__(self, tiles_y, width, dtype, device, rank, world, dst=0, tile=32, layout="interleaved", block_rows=1):
        self.tiles_y, self.W, self.rank, self.world, self.dst, self.tile = tiles_y, width, rank, world, dst, tile
        self.dtype, self.device, self.layout, self.block_rows = dtype, device, layout, block_rows
        self.rows = owned_rows(layout, tiles_y, world, rank, block_rows)
        # runs of bin rows this rank sends, or (on dst) receives from each peer -- fixed for the lifetime of the object
        if rank == dst:
            self.plan = [(t, c, src) for src in range(world) if src != dst
                         for t, c in row_runs(owned_rows(layout, tiles_y, world, src, block_rows))]
        else:
            self.plan = [(t, c, dst) for t, c in row_runs(self.rows)]
        self.bytes_per_frame = sum(c for _, c, _ in self.plan) * tile * width * 4 * (2 if str(dtype).endswith("float16") else 4)

    def owned(self, fb):
        """this rank's bin rows inside a (tiles_y*tile, W, 4) framebuffer (a copy when they are not evenly strided)"""
        return fb.view(self.tiles_y, self.tile, self.W, 4)[self.rows]

    def __call__(self, fb):
        """returns the assembled (tiles_y*tile, W, 4) image on rank dst (fb itself), None elsewhere.
        Asynchronous on the current stream with RCCL; the tensors must stay alive until the stream has passed."""
        rows = fb.view(self.tiles_y, self.tile, self.W, 4)
        op = dist.irecv if self.rank == self.dst else dist.isend
        if fb.is_cuda and dist.get_backend() == "gloo":
            # debug path only (bench.py MSPLAT_BENCH_ONE_DEVICE=1: several ranks on one GPU, no RCCL): gloo moves host
            # memory, so the rows are staged through the CPU here; RCCL sends / receives the device rows in place
            host = {t: rows[t:t + c].cpu() for t, c, _ in self.plan}
            for req in dist.batch_isend_irecv([dist.P2POp(op, host[t], peer) for t, _, peer in self.plan]) if self.plan else []:
                req.wait()
            if self.rank == self.dst:
                for t, c, _ in self.plan:
                    rows[t:t + c].copy_(host[t])
            return fb if self.rank == self.dst else None
        ops = [dist.P2POp(op, rows[t:t + c], peer) for t, c, peer in self.plan]      # contiguous slices of fb, in place
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return fb if self.rank == self.dst else None


# ------------------------------------------------------------------------------------------------------------------------------
# The same gather behind the C ABI (r5): msplat_band_exchange(ctx, ncclComm_t, ...) -- what a C++ host with one process per GPU
# calls (include/msplat.h).  Python only has to own a communicator: RcclComm makes one with ncclGetUniqueId / ncclCommInitRank
# through ctypes on the librccl the process already uses (the id travels over the torch.distributed group that exists anyway).
# ------------------------------------------------------------------------------------------------------------------------------
import ctypes as _C
import os as _os


def loaded_rccl_path():
    """the librccl mapped into this process (PyTorch's bundled copy once torch is imported), else the system's"""
    try:
        for ln in open("/proc/self/maps"):
            p = ln.rsplit(" ", 1)[-1].strip()
            if "librccl" in _os.path.basename(p):
                return p
    except OSError:
        pass
    for p in ("/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"):
        if _os.path.exists(p):
            return p
    return None


class _UniqueId(_C.Structure):
    _fields_ = [("internal", _C.c_char * 128)]      # NCCL_UNIQUE_ID_BYTES (rccl.h:40-43)


class RcclComm:
    """an ncclComm_t of this process's rank (rccl.h:187-260): `handle` is what msplat_band_exchange takes.
    world == 1 needs no process group; otherwise the unique id is broadcast from rank 0 over torch.distributed."""

    def __init__(self, rank=0, world=1, device=0, group=None):
        path = loaded_rccl_path()
        if path is None:
            raise RuntimeError("librccl not found")
        _os.environ.setdefault("MSPLAT_RCCL_LIB", path)       # libmsplat resolves the same copy (communicators belong to one copy)
        self._lib = L = _C.CDLL(path)
        L.ncclGetErrorString.restype = _C.c_char_p
        L.ncclGetUniqueId.argtypes = [_C.POINTER(_UniqueId)]
        L.ncclCommInitRank.argtypes = [_C.POINTER(_C.c_void_p), _C.c_int, _UniqueId, _C.c_int]
        L.ncclCommDestroy.argtypes = [_C.c_void_p]
        import torch
        torch.cuda.set_device(device)
        uid = _UniqueId()
        if rank == 0:
            self._ok(L.ncclGetUniqueId(_C.byref(uid)), "ncclGetUniqueId")
        if world > 1:
            box = [bytes(uid)] if rank == 0 else [None]          # the raw 128 bytes (uid.internal would stop at a NUL)
            dist.broadcast_object_list(box, src=0, group=group)       # (group: e.g. a gloo side group; None = the default group)
            _C.memmove(_C.byref(uid), box[0], 128)
        self.handle = _C.c_void_p()
        self._ok(L.ncclCommInitRank(_C.byref(self.handle), world, uid, rank), "ncclCommInitRank")
        self.rank, self.world, self.path = rank, world, path

    def _ok(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s: %s" % (what, self._lib.ncclGetErrorString(rc).decode()))

    def close(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self._lib.ncclCommDestroy(h)

    def __del__(self):
        self.close()


_KIND = {"contiguous": 0, "interleaved": 1, "block": 2, "weighted": 3}


class CAbiBandGather:
    """BandGather's job through msplat_band_exchange: one ncclGroupStart/End of ncclSend / ncclRecv issued by libmsplat on the
    renderer's stream.  Same plan (runs of consecutive bin rows, straight into rank dst's framebuffer)."""

    def __init__(self, renderer, comm, tiles_y, width, dtype, rank, world, dst=0, tile=32, layout="interleaved", block_rows=1,
                 wire_fp16=False):
        self.r, self.comm, self.rank, self.world, self.dst = renderer, comm, rank, world, dst
        self.kind, self.block_rows = _KIND[layout], int(block_rows)
        self.W, self.H = width, tiles_y * tile
        self.pitch = width * (8 if str(dtype).endswith("float16") else 16)
        self.wire_fp16 = bool(wire_fp16) and not str(dtype).endswith("float16")       # fp32 targets: RGBA16F on the wire

    def __call__(self, fb):
        self.r.band_exchange(self.comm.handle, self.rank, self.world, self.dst, self.kind, self.block_rows, fb.data_ptr(),
                             self.pitch, self.W, self.H, wire_fp16=self.wire_fp16)
        return fb if self.rank == self.dst else None
