"""Minimal driver for profiler runs: a few frames of one scene with the default pipeline, nothing else.  Every frame is
scheduled as a first frame (no tile order from the previous frame), as the benchmark's timed frames are.

    python tools/prof_frame.py [scene] [width height] [sub] [frames] [levels]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = sys.argv[1] if len(sys.argv) > 1 else "scene7"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
sub = int(sys.argv[4]) if len(sys.argv) > 4 else 6
frames = int(sys.argv[5]) if len(sys.argv) > 5 else 5
lq = len(sys.argv) > 6 and sys.argv[6] == "levels"
sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", scene + ".test"))
r = rt.Renderer(0)
r.set_scene(sr)
r.build_grid(15, sub, rt.GRID_NESTED)
d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for _ in range(frames):
    r.render_device(d_rgb.data_ptr(), 0, W, H, level_queues=lq, tile_hint=False)
torch.cuda.synchronize()
r.close()
