"""The N > 1 frame assembly (interleaved row tiles gathered to rank 0) over gloo on CPU tensors, world_size 2 and 3.
The rendering itself needs a GPU; here every rank fabricates its rows from a known image."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd.multi_gpu import FrameGather, max_rows_per_rank, rank_rows


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _image(h, w):
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([y * 1000 + x, y - x, (y * x) % 7], axis=2).astype(np.float32)


def _pipelined_worker(rank, world, port, h, w, tile_rows, result):
    """Two frames in flight: frame k's exchange is completed only when its slot is needed again (frame k + 2)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.from_numpy(_image(h, w))
        fg = FrameGather(h, w, 3, tile_rows, rank, world, torch.device("cpu"), depth=2)
        rows = torch.as_tensor(rank_rows(h, tile_rows, rank, world))
        seen = []
        for k in range(5):
            b = k % 2
            if fg.in_flight(b):
                out = fg.finish(b)
                if rank == 0:
                    assert torch.equal(out, full + (k - 2))
                    seen.append(k - 2)
            fg.local_view(b).copy_(full[rows] + k)
            fg.gather_async(b)
        for k in (3, 4):
            out = fg.finish(k % 2)
            if rank == 0:
                assert torch.equal(out, full + k)
                seen.append(k)
            assert not fg.in_flight(k % 2)
        if rank == 0:
            assert seen == [0, 1, 2, 3, 4]
            result.put("ok")
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, h, w, tile_rows, dtype, result):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.from_numpy(_image(h, w)).to(dtype)       # 8-bit frames wrap around: still a known image
        fg = FrameGather(h, w, 3, tile_rows, rank, world, torch.device("cpu"), dtype=dtype)
        rows = rank_rows(h, tile_rows, rank, world)
        for frame in range(2):      # buffers are reused frame after frame
            fg.local_view().copy_(full[torch.as_tensor(rows)] + frame)
            out = fg.gather()
            if rank == 0:
                assert out.dtype == dtype and torch.equal(out, full + frame)
            else:
                assert out is None
        if rank == 0:
            result.put("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype", [torch.float32, torch.uint8], ids=["f32", "u8"])
@pytest.mark.parametrize("world,h,w,tile_rows", [(2, 37, 16, 8), (3, 50, 9, 5), (2, 8, 4, 8)])
def test_gather_reassembles_the_frame(world, h, w, tile_rows, dtype):
    ctx = mp.get_context("spawn")
    result = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, h, w, tile_rows, dtype, result)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert result.get(timeout=5) == "ok"


@pytest.mark.parametrize("world,h,w,tile_rows", [(2, 37, 16, 8), (3, 50, 9, 5)])
def test_two_frames_in_flight(world, h, w, tile_rows):
    ctx = mp.get_context("spawn")
    result = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, world, port, h, w, tile_rows, result)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert result.get(timeout=5) == "ok"


def test_single_rank_needs_no_process_group():
    h, w = 21, 6
    fg = FrameGather(h, w, 3, 8, 0, 1, torch.device("cpu"))
    full = torch.from_numpy(_image(h, w))
    fg.local_view().copy_(full)
    assert torch.equal(fg.gather(), full)
    fg.gather_async()
    assert not fg.in_flight() and torch.equal(fg.finish(), full)


def test_padding_covers_the_largest_rank():
    assert max_rows_per_rank(2160, 8, 8) == 34 * 8          # 270 tiles over 8 ranks: ranks 0..5 hold 34 tiles
    assert len(rank_rows(2160, 8, 7, 8)) == 33 * 8
    assert max_rows_per_rank(7, 8, 2) == 7 and len(rank_rows(7, 8, 1, 2)) == 0
