"""Development: the same frame many times over, every copy compared bit for bit on the device with the level-queue
pipeline's frame (a rare race in the persistent kernel would show as a frame that differs).

    python tools/gpu_soak.py [frames per case]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
r = rt.Renderer(0)
bad = 0
cases = [("scene7", 6, 3840, 2160, rt.SEARCH_NESTED), ("scene7", 6, 640, 480, rt.SEARCH_NESTED), ("scene7", 6, 1280, 720, rt.SEARCH_NESTED),
         ("scene6", 0, 640, 480, rt.SEARCH_NESTED), ("scene6", 0, 640, 480, rt.SEARCH_REFERENCE_LISTS),
         ("scene4-specular", 0, 640, 480, rt.SEARCH_REFERENCE_LISTS), ("scene4-specular", 0, 1920, 1080, rt.SEARCH_NESTED),
         ("scene5", 0, 640, 480, rt.SEARCH_NESTED), ("scene5", 0, 320, 240, rt.SEARCH_REFERENCE_LISTS)]
for name, sub, W, H, mode in cases:
    r.set_scene(rt.SceneReader().readScene(os.path.join(ROOT, "scenes", name + ".test")))
    r.build_grid(15, sub)
    want = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    want_ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
    got = torch.empty_like(want)
    got_ids = torch.empty_like(want_ids)
    r.render_device(want.data_ptr(), want_ids.data_ptr(), W, H, 5, mode, level_queues=True)
    torch.cuda.synchronize()
    differ = 0
    for waves in (3, 4):
        for k in range(n):
            got.fill_(-1.0)
            # every third frame with helper lanes from the start and a varying number of paths per wave
            kw = dict(helpers="early", path_lanes=8 + (5 * k) % 56) if k % 3 == 2 else {}
            if k % 4 == 1:                             # every fourth frame in generations, the endgame starting at varying points
                kw = dict(generations=True, endgame_eighths=(-1, 0, 12, 40)[(k // 4) % 4])
            r.render_device(got.data_ptr(), got_ids.data_ptr(), W, H, 5, mode, frame_waves=waves, tile_hint=k % 2 == 0, **kw)
            torch.cuda.synchronize()
            if not (torch.equal(got.view(torch.int32), want.view(torch.int32)) and torch.equal(got_ids, want_ids)):
                differ += 1
    bad += differ
    print(f"{name} {W}x{H} mode {mode}: {2 * n} frames, {differ} differ", flush=True)
r.close()
sys.exit(1 if bad else 0)
