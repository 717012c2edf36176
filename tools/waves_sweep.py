"""Development (GPU box): three against four waves per SIMD of the per-lane frame kernel on frames of few pixels per
lane, lone first frames.  python tools/waves_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401

r = rt.Renderer(0)
cases = [("scene1", 640, 480, 0, 1), ("scene4-specular", 640, 480, 0, 1), ("scene5", 640, 480, 0, 1), ("scene6", 640, 480, 0, 1),
         ("scene7", 640, 480, 6, 1), ("scene7", 1280, 720, 6, 1), ("scene7", 2560, 1440, 6, 1), ("scene7", 3840, 2160, 6, 8),
         ("scene7", 3840, 2160, 6, 4), ("scene7", 3840, 2160, 6, 2)]
for name, W, H, sub, world in cases:
    sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", name + ".test"))
    r.set_scene(sr)
    r.build_grid(15, sub)
    d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    line = [f"{name} {W}x{H} world {world}"]
    for waves in (0, 3, 4):
        ts = [r.render_device(d_rgb.data_ptr(), 0, W, H, rank=0, world=world, want_stats=True, tile_hint=False, frame_waves=waves, generations=False).frame_ms for _ in range(9)]
        line.append(f"waves {waves if waves else 'auto'}: {np.median(ts):.3f}")
    print("  ".join(line), flush=True)
r.close()
