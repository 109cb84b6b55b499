"""Multi-GPU tile-row sharding (one process per GPU, torch.distributed = RCCL over xGMI on ROCm).

The path shards by screen tiles: rank g of G renders bin rows t with t % G == g
(rows of msplat_tile_size() = 32 pixels, interleaved for load balance; per-pixel results are bit-identical to single-GPU
rendering because every bin still sees its splats in global depth order).  The only exchange is the final row
gather to rank 0, and it moves every band exactly once:

  * every rank renders into a full-size framebuffer (it writes only its own bin rows);
  * rank 0's framebuffer IS the final image: it posts one receive per foreign bin row, straight into that row's
    place (a bin row is a contiguous block of tile x W pixels), and the other ranks send their rows from where the
    compositor left them -- no pack, no staging buffer, no unpack;
  * all of a frame's sends/receives are issued as ONE group (ncclGroupStart/End via batch_isend_irecv), so
    each rank's band travels over its own direct xGMI link concurrently (7 links into rank 0), not around a ring.

No collective is used anywhere else."""
import torch.distributed as dist


class BandGather:
    """Gathers the owned bin rows of every rank's full-size framebuffer into rank `dst`'s framebuffer.

    fb layout on every rank: (tiles_y * tile, W, 4) (height padded to a multiple of the bin size);
    rank g has written rows of bins g, g+G, ... only.  On `dst` the call returns fb itself, completed."""

    def __init__(self, tiles_y, width, dtype, device, rank, world, dst=0, tile=32):
        self.tiles_y, self.W, self.rank, self.world, self.dst, self.tile = tiles_y, width, rank, world, dst, tile
        self.dtype, self.device = dtype, device
        # bin rows this rank sends, or (on dst) receives from each peer -- fixed for the lifetime of the object
        if rank == dst:
            self.plan = [(t, src) for src in range(world) if src != dst for t in range(src, tiles_y, world)]
        else:
            self.plan = [(t, dst) for t in range(rank, tiles_y, world)]
        self.bytes_per_frame = len(self.plan) * tile * width * 4 * (2 if str(dtype).endswith("float16") else 4)

    def owned(self, fb):
        """view of this rank's bin rows inside a (tiles_y*tile, W, 4) framebuffer"""
        return fb.view(self.tiles_y, self.tile, self.W, 4)[self.rank::self.world]

    def __call__(self, fb):
        """returns the assembled (tiles_y*tile, W, 4) image on rank dst (fb itself), None elsewhere.
        Asynchronous on the current stream with RCCL; the tensors must stay alive until the stream has passed."""
        rows = fb.view(self.tiles_y, self.tile, self.W, 4)
        op = dist.irecv if self.rank == self.dst else dist.isend
        if fb.is_cuda and dist.get_backend() == "gloo":
            # debug path only (bench.py MSPLAT_BENCH_ONE_DEVICE=1: several ranks on one GPU, no RCCL): gloo moves host
            # memory, so the rows are staged through the CPU here; RCCL sends / receives the device rows in place
            host = {t: rows[t].cpu() for t, _ in self.plan}
            for req in dist.batch_isend_irecv([dist.P2POp(op, host[t], peer) for t, peer in self.plan]) if self.plan else []:
                req.wait()
            if self.rank == self.dst:
                for t, _ in self.plan:
                    rows[t].copy_(host[t])
            return fb if self.rank == self.dst else None
        ops = [dist.P2POp(op, rows[t], peer) for t, peer in self.plan]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return fb if self.rank == self.dst else None
