"""Multi-GPU frame assembly: one process per GPU, scene replicated, interleaved row tiles, one gather.

Pixels are independent, so the only exchange step of a frame is the gather of every rank's rows to
rank 0 (RCCL over xGMI when the process group backend is "nccl"; the same code runs over gloo on CPU
tensors, which is how the N > 1 path is tested without GPUs).  Tile t (``tile_rows`` consecutive image
rows) belongs to rank ``t % world``: interleaving balances scenes whose cost is concentrated mid-frame.
The reference has no counterpart: its only parallelism is OpenMP over rows (src/main.cpp:126).

A frame's gather may stay in flight while the next frame renders: the buffers exist ``depth`` times and
``gather_async(slot)`` / ``finish(slot)`` split the exchange from the point where its result is needed.
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
    rows already are the frame).  ``depth`` independent buffer sets ("slots") allow that many frames in flight."""

    def __init__(self, height, width, channels, tile_rows, rank, world, device, dtype=torch.float32, depth=1):
        self.h, self.w, self.c = height, width, channels
        self.rank, self.world, self.depth = rank, world, depth
        self.pad_rows = max_rows_per_rank(height, tile_rows, world)
        self.n_local = len(rank_rows(height, tile_rows, rank, world))
        # every rank sends the same number of rows (ranks with one tile fewer send padding)
        self.send = [torch.zeros((self.pad_rows, width, channels), dtype=dtype, device=device) for _ in range(depth)]
        self.recv = self.recv_list = self.full = None
        self.work = [None] * depth
        if rank == 0 and world > 1:
            self.recv = [torch.zeros((world, self.pad_rows, width, channels), dtype=dtype, device=device) for _ in range(depth)]
            self.recv_list = [list(r.unbind(0)) for r in self.recv]          # contiguous views, one per rank
            self.full = [torch.zeros((height, width, channels), dtype=dtype, device=device) for _ in range(depth)]
            # image row y comes from row src[y] of the receive buffer seen as [world * pad_rows, W, C]
            src = np.zeros(height, np.int64)
            for r in range(world):
                rows = rank_rows(height, tile_rows, r, world)
                src[rows] = r * self.pad_rows + np.arange(len(rows))
            self.src_index = torch.as_tensor(src, device=device)

    def local_view(self, slot=0):
        """The part of the send buffer the renderer writes: [n_local_rows, W, C], contiguous."""
        return self.send[slot][:self.n_local]

    def gather_async(self, slot=0):
        """Starts the exchange of slot `slot` (its rows must have been enqueued on the current stream)."""
        if self.world == 1:
            return
        assert self.work[slot] is None, "slot still in flight: call finish() first"
        self.work[slot] = dist.gather(self.send[slot], self.recv_list[slot] if self.rank == 0 else None, dst=0, async_op=True)

    def finish(self, slot=0):
        """Completes the exchange of `slot`; returns the assembled [H, W, C] image on rank 0, None elsewhere.  The
        slot's buffers may be written again afterwards (the returned image lives until the slot's next finish)."""
        if self.world == 1:
            return self.send[slot][:self.n_local]                  # one rank owns every row, in order
        if self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None
        if self.rank != 0:
            return None
        torch.index_select(self.recv[slot].view(self.world * self.pad_rows, self.w, self.c), 0, self.src_index, out=self.full[slot])
        return self.full[slot]

    def in_flight(self, slot=0):
        return self.work[slot] is not None

    def gather(self, slot=0):
        """Exchange and completion in one call."""
        self.gather_async(slot)
        return self.finish(slot)
