"""Multi-GPU frame assembly: one process per GPU, scene replicated, interleaved row tiles, one gather.

Pixels are independent, so the only exchange step of a frame is the gather of every rank's rows to
rank 0 (RCCL over xGMI when the process group backend is "nccl"; the same code runs over gloo on CPU
tensors, which is how the N > 1 path is tested without GPUs).  Tile t (``tile_rows`` consecutive image
rows) belongs to rank ``t % world``: interleaving balances scenes whose cost is concentrated mid-frame.
The reference has no counterpart: its only parallelism is OpenMP over rows (src/main.cpp:126).
"""
import numpy as np
import torch
import torch.distributed as dist


def rank_rows(height, tile_rows, rank, world):
    """Global row numbers owned by `rank`, ascending (pure index arithmetic; same rule as rtx_local_rows)."""
    n_tiles = (height + tile_rows - 1) // tile_rows
    rows = []
    for t in range(rank, n_tiles, world):
        rows.extend(range(t * tile_rows, min((t + 1) * tile_rows, height)))
    return np.asarray(rows, np.int64)


def max_rows_per_rank(height, tile_rows, world):
    return max(len(rank_rows(height, tile_rows, r, world)) for r in range(world))


class FrameGather:
    """Pre-allocates the padded send/receive buffers and the row permutation for one image geometry, so that a
    frame costs one collective and one gather-by-index kernel on rank 0 (nothing at all on a single rank, whose
    rows already are the frame)."""

    def __init__(self, height, width, channels, tile_rows, rank, world, device, dtype=torch.float32):
        self.h, self.w, self.c = height, width, channels
        self.rank, self.world = rank, world
        self.pad_rows = max_rows_per_rank(height, tile_rows, world)
        self.n_local = len(rank_rows(height, tile_rows, rank, world))
        # every rank sends the same number of rows (ranks with one tile fewer send padding)
        self.send = torch.zeros((self.pad_rows, width, channels), dtype=dtype, device=device)
        self.recv = None
        self.full = None
        if rank == 0 and world > 1:
            self.recv = torch.zeros((world, self.pad_rows, width, channels), dtype=dtype, device=device)
            self.recv_list = list(self.recv.unbind(0))          # contiguous views, one per rank
            self.full = torch.zeros((height, width, channels), dtype=dtype, device=device)
            # image row y comes from row src[y] of the receive buffer seen as [world * pad_rows, W, C]
            src = np.zeros(height, np.int64)
            for r in range(world):
                rows = rank_rows(height, tile_rows, r, world)
                src[rows] = r * self.pad_rows + np.arange(len(rows))
            self.src_index = torch.as_tensor(src, device=device)

    def local_view(self):
        """The part of the send buffer the renderer writes: [n_local_rows, W, C], contiguous."""
        return self.send[:self.n_local]

    def gather(self):
        """Gathers all ranks' rows; returns the assembled [H, W, C] image on rank 0, None elsewhere."""
        if self.world == 1:
            return self.send[:self.n_local]                     # one rank owns every row, in order
        dist.gather(self.send, self.recv_list if self.rank == 0 else None, dst=0)
        if self.rank != 0:
            return None
        torch.index_select(self.recv.view(self.world * self.pad_rows, self.w, self.c), 0, self.src_index, out=self.full)
        return self.full
