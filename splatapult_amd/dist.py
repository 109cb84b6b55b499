"""Multi-GPU tile-row sharding (one process per GPU, torch.distributed = RCCL over xGMI on ROCm).

The path shards by screen tiles: rank g of G renders tile rows t with t % G == g
(16-pixel rows, interleaved for load balance; per-pixel results are bit-identical to single-GPU
rendering because every tile still sees its splats in global depth order).  The only exchange is
the final row gather to rank 0: one message of ceil(rows/G) x 16 x W pixels per rank per view.
No collective is used anywhere else."""
import torch
import torch.distributed as dist


class BandGather:
    """Gathers the owned tile rows of every rank's full-size framebuffer into rank 0's image.

    fb layout on every rank: (tiles_y * 16, W, 4) (height padded to a multiple of 16);
    rank g has written rows of tiles g, g+G, ... only."""

    def __init__(self, tiles_y, width, dtype, device, rank, world, dst=0):
        self.tiles_y, self.W, self.rank, self.world, self.dst = tiles_y, width, rank, world, dst
        self.max_rows = (tiles_y + world - 1) // world
        self.send = torch.zeros((self.max_rows, 16, width, 4), dtype=dtype, device=device)
        self.recv = None
        self.final = None
        if rank == dst:
            self.recv = [torch.zeros((self.max_rows, 16, width, 4), dtype=dtype, device=device) for _ in range(world)]
            self.final = torch.zeros((tiles_y, 16, width, 4), dtype=dtype, device=device)

    def owned(self, fb):
        """view of this rank's tile rows inside a (tiles_y*16, W, 4) framebuffer"""
        return fb.view(self.tiles_y, 16, self.W, 4)[self.rank::self.world]

    def __call__(self, fb):
        """returns the assembled (tiles_y*16, W, 4) image on rank dst, None elsewhere"""
        mine = self.owned(fb)
        self.send[:mine.shape[0]].copy_(mine)
        dist.gather(self.send, self.recv if self.rank == self.dst else None, dst=self.dst)
        if self.rank != self.dst:
            return None
        for g in range(self.world):
            rows = self.final[g::self.world]
            rows.copy_(self.recv[g][:rows.shape[0]])
        return self.final.view(self.tiles_y * 16, self.W, 4)
