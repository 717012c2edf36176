"""Development (GPU box): lone first frames of the small configurations over the number of finished rays a wave waits
for before a state update (ready_threshold) and its share while the wave drains (drain_eighths).
python tools/threshold_sweep_small.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401

r = rt.Renderer(0)
cases = [("scene4-specular", 640, 480, 0, 1), ("scene5", 640, 480, 0, 1), ("scene6", 640, 480, 0, 1), ("scene7", 3840, 2160, 6, 8)]
for name, W, H, sub, world in cases:
    sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", name + ".test"))
    r.set_scene(sr)
    r.build_grid(15, sub)
    d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    line = [f"{name} {W}x{H} world {world} | threshold"]
    for th in (4, 8, 16, 24, 32, 48, 56):
        ts = [r.render_device(d_rgb.data_ptr(), 0, W, H, rank=0, world=world, want_stats=True, tile_hint=False, ready_threshold=th).frame_ms for _ in range(7)]
        line.append(f"{th}: {np.median(ts):.3f}")
    line.append("| drain eighths")
    for dr in (1, 2, 3, 4, 6, 8):
        ts = [r.render_device(d_rgb.data_ptr(), 0, W, H, rank=0, world=world, want_stats=True, tile_hint=False, drain_eighths=dr).frame_ms for _ in range(7)]
        line.append(f"{dr}: {np.median(ts):.3f}")
    print("  ".join(line), flush=True)
r.close()
