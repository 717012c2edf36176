"""Development (GPU box): frame times of consecutive lone 4K frames of scene7 with the previous frame's hint."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401

W, H = 3840, 2160
sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", "scene7.test"))
r = rt.Renderer(0)
r.set_scene(sr)
r.build_grid(15, 6)
d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for hint in (False, True):
    ts = []
    for _ in range(10):
        st = r.render_device(d_rgb.data_ptr(), 0, W, H, want_stats=True, tile_hint=hint, generations=False)
        ts.append(round(st.frame_ms, 3))
    print("hint", hint, ts, flush=True)
r.close()
