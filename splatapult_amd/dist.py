"""Multi-GPU tile-row sharding (one process per GPU, torch.distributed = RCCL over xGMI on ROCm).

The path shards by screen tiles: rank g of G renders tile rows t with t % G == g
(rows of msplat_tile_size() = 32 pixels, interleaved for load balance; per-pixel results are bit-identical to single-GPU
rendering because every tile still sees its splats in global depth order).  The only exchange is
the final row gather to rank 0: one message of ceil(rows/G) x 32 x W pixels per rank per view.
No collective is used anywhere else."""
import torch
import torch.distributed as dist


class BandGather:
    """Gathers the owned tile rows of every rank's full-size framebuffer into rank 0's image.

    fb layout on every rank: (tiles_y * tile, W, 4) (height padded to a multiple of the tile size);
    rank g has written rows of tiles g, g+G, ... only."""

    def __init__(self, tiles_y, width, dtype, device, rank, world, dst=0, tile=32):
        self.tiles_y, self.W, self.rank, self.world, self.dst, self.tile = tiles_y, width, rank, world, dst, tile
        self.max_rows = (tiles_y + world - 1) // world
        self.send = torch.zeros((self.max_rows, tile, width, 4), dtype=dtype, device=device)
        self.recv = None
        self.final = None
        if rank == dst:
            # one allocation for all ranks' bands: the interleave back into image order is then a single
            # strided copy (rank g's r-th band is tile row r * world + g) instead of one copy per rank
            self.recv_all = torch.zeros((world, self.max_rows, tile, width, 4), dtype=dtype, device=device)
            self.recv = list(self.recv_all.unbind(0))
            self.final_padded = torch.zeros((self.max_rows * world, tile, width, 4), dtype=dtype, device=device)
            self.final = self.final_padded[:tiles_y]

    def owned(self, fb):
        """view of this rank's tile rows inside a (tiles_y*tile, W, 4) framebuffer"""
        return fb.view(self.tiles_y, self.tile, self.W, 4)[self.rank::self.world]

    def __call__(self, fb):
        """returns the assembled (tiles_y*tile, W, 4) image on rank dst, None elsewhere"""
        mine = self.owned(fb)
        self.send[:mine.shape[0]].copy_(mine)
        dist.gather(self.send, self.recv if self.rank == self.dst else None, dst=self.dst)
        if self.rank != self.dst:
            return None
        self.final_padded.view(self.max_rows, self.world, self.tile, self.W, 4).copy_(self.recv_all.permute(1, 0, 2, 3, 4))
        return self.final.reshape(self.tiles_y * self.tile, self.W, 4)
