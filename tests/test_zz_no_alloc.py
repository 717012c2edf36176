"""rtx_render_device_noalloc: the entry point for callers that cannot allocate, against rtx_render_device.

Through the C ABI, two calls in a row per scene (the second with a smaller frame, so that it meets whatever the first
one left in the library's per-GPU scratch array), outputs compared bit for bit with the existing entry point's; free
device memory must not move across the calls; sizes beyond the scratch array's limits are refused."""
import numpy as np
import pytest
import torch

import raytracer_amd as rt
from conftest import scene_path

pytestmark = pytest.mark.gpu


def _both(r, W, H, depth, mode, **kw):
    """(existing entry point, no-alloc entry point) outputs for one frame, as host arrays."""
    want = torch.full((H, W, 3), -1.0, device="cuda")
    want_ids = torch.full((H, W), -9, dtype=torch.int32, device="cuda")
    got = torch.full((H, W, 3), -2.0, device="cuda")
    got_ids = torch.full((H, W), -8, dtype=torch.int32, device="cuda")
    r.render_device(want.data_ptr(), want_ids.data_ptr(), W, H, depth, mode, **kw)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    r.render_device_noalloc(got.data_ptr(), got_ids.data_ptr(), W, H, depth, mode, **kw)
    torch.cuda.synchronize()
    r.frame_status()
    assert torch.cuda.mem_get_info()[0] == free_before, "the call changed the amount of free device memory"
    return want.cpu().numpy(), want_ids.cpu().numpy(), got.cpu().numpy(), got_ids.cpu().numpy()


@pytest.mark.parametrize("name,sub,sizes,depth", [("scene6", 0, [(640, 480), (200, 117)], 5), ("scene7", 6, [(3840, 2160), (1280, 720)], 5),
                                                  ("scene4-specular", 0, [(320, 240), (37, 23)], 6), ("scene5", 0, [(320, 240), (160, 120)], 5)])
def test_no_alloc_entry_point_matches_render_device(name, sub, sizes, depth):
    r = rt.Renderer(0)
    try:
        r.set_scene(rt.SceneReader().readScene(scene_path(name)))
        r.build_grid(15, sub)
        r.render(64, 48, depth)                                   # the context's own workspace exists before the comparison
        for (W, H) in sizes:                                      # two calls in a row, the second one smaller
            for mode in (rt.SEARCH_NESTED, rt.SEARCH_REFERENCE_LISTS):
                if name == "scene7" and mode == rt.SEARCH_REFERENCE_LISTS and W > 2000:
                    continue                                      # a quarter of a second of GPU time buys nothing here
                want, want_ids, got, got_ids = _both(r, W, H, depth, mode)
                assert np.array_equal(got_ids, want_ids), (W, H, mode)
                assert np.array_equal(got.view(np.int32), want.view(np.int32)), (W, H, mode)
        # a rank's share of the frame, both register budgets, a second context of the same GPU in between (it takes the
        # scratch over, so the first context's scheduling hint in it is void)
        other = rt.Renderer(0)
        try:
            other.set_scene(rt.SceneReader().readScene(scene_path("scene4-specular")))
            other.build_grid(15, 0)
            W, H = sizes[1]
            for kw in (dict(rank=1, world=3, tile_rows=5), dict(frame_waves=3), dict(frame_waves=4, tile_hint=False)):
                d = torch.zeros((H, W, 3), device="cuda")
                other.render_device_noalloc(d.data_ptr(), 0, W, H, 5)
                torch.cuda.synchronize()
                want, want_ids, got, got_ids = _both(r, W, H, depth, rt.SEARCH_NESTED, **kw)
                rows = rt.local_rows(H, kw.get("tile_rows", 8), kw.get("rank", 0), kw.get("world", 1))
                n = len(rows)                                      # the device buffers hold this rank's rows first
                assert np.array_equal(got_ids.reshape(-1, W)[:n], want_ids.reshape(-1, W)[:n]), kw
                assert np.array_equal(got.view(np.int32).reshape(-1, W, 3)[:n], want.view(np.int32).reshape(-1, W, 3)[:n]), kw
        finally:
            other.close()
    finally:
        r.close()


def test_no_alloc_entry_point_refuses_what_its_scratch_cannot_hold():
    r = rt.Renderer(0)
    try:
        r.set_scene(rt.SceneReader().readScene(scene_path("scene6")))
        r.build_grid(15, 0)
        d = torch.zeros((64, 64, 3), device="cuda")
        with pytest.raises(rt.RtxError) as e:
            r.render_device_noalloc(d.data_ptr(), 0, 64, 64, max_depth=7)
        assert e.value.code == -6 and "max_depth" in str(e.value)
        with pytest.raises(rt.RtxError) as e:                        # 8192 x 4096 pixels: 524288 tiles
            r.render_device_noalloc(d.data_ptr(), 0, 8192, 4096)
        assert e.value.code == -6
        for flags in (rt.api.RENDER_COUNT_WORK, rt.api.RENDER_LEVEL_QUEUES, rt.api.RENDER_SERIAL_LANES):
            with pytest.raises(rt.RtxError) as e:
                r.render_device_noalloc(d.data_ptr(), 0, 64, 64, flags=flags)
            assert e.value.code == -1
        r.render_device_noalloc(d.data_ptr(), 0, 64, 64, max_depth=6)
        torch.cuda.synchronize()
        r.frame_status()
    finally:
        r.close()
