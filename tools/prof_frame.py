"""Minimal driver for profiler runs: a few frames of scene7 @ 4K with the default pipeline, nothing else."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sub = int(sys.argv[1]) if len(sys.argv) > 1 else 6
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lq = len(sys.argv) > 3 and sys.argv[3] == "levels"
W, H = 3840, 2160
sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", "scene7.test"))
r = rt.Renderer(0)
r.set_scene(sr)
r.build_grid(15, sub, rt.GRID_NESTED)
d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for _ in range(frames):
    r.render_device(d_rgb.data_ptr(), 0, W, H, level_queues=lq)
torch.cuda.synchronize()
r.close()
