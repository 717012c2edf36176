"""Development (GPU box): lone first frames of the small configurations over image sizes and paths per wave.
python tools/size_sweep_small.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401

r = rt.Renderer(0)
for name in ("scene4-specular", "scene5", "scene6"):
    sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", name + ".test"))
    r.set_scene(sr)
    r.build_grid(15, 0)
    line = [name]
    for W, H in ((160, 120), (320, 240), (640, 480), (1280, 960), (2560, 1920)):
        d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        ts = [r.render_device(d_rgb.data_ptr(), 0, W, H, want_stats=True, tile_hint=False) for _ in range(7)]
        line.append(f"{W}x{H}: {np.median([t.frame_ms for t in ts]):.3f} (kernel {np.median([t.trace_ms for t in ts]):.3f})")
    print("  ".join(line), flush=True)
    line = [name + " 640x480 by paths per wave"]
    d_rgb = torch.empty((480, 640, 3), dtype=torch.float32, device="cuda")
    for pl in (8, 16, 24, 32, 48, 64):
        for waves in (3, 4):
            ts = [r.render_device(d_rgb.data_ptr(), 0, 640, 480, want_stats=True, tile_hint=False, path_lanes=pl, frame_waves=waves).frame_ms for _ in range(7)]
            line.append(f"{pl}/w{waves}: {np.median(ts):.3f}")
    print("  ".join(line), flush=True)
r.close()
