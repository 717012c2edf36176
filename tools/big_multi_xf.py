"""Development (GPU box): the scenes with several transforms at 3840x2160 (throughput-bound frames), lone first frames.
python tools/big_multi_xf.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401

W, H = 3840, 2160
r = rt.Renderer(0)
d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
out = []
for name in ("scene4-specular", "scene5", "scene6"):
    sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", name + ".test"))
    r.set_scene(sr)
    r.build_grid(15, 0)
    for gen in (False, True):
        ts = [r.render_device(d_rgb.data_ptr(), 0, W, H, want_stats=True, tile_hint=False, generations=gen).frame_ms for _ in range(7)]
        out.append(f"{name} {'generations' if gen else 'per-lane paths'} {np.median(ts):.3f}")
print(" | ".join(out), flush=True)
r.close()
