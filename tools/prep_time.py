"""Development (GPU box): time of the frame-start kernel (frame time minus frame kernel time) of lone frames.
python tools/prep_time.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401

r = rt.Renderer(0)
out = []
for name, W, H, sub in (("scene1", 640, 480, 0), ("scene5", 640, 480, 0), ("scene6", 640, 480, 0), ("scene7", 3840, 2160, 6)):
    sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", name + ".test"))
    r.set_scene(sr)
    r.build_grid(15, sub)
    d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    ts = []
    for _ in range(9):
        st = r.render_device(d_rgb.data_ptr(), 0, W, H, want_stats=True, tile_hint=False)
        ts.append((st.frame_ms, st.frame_ms - st.trace_ms))
    t = np.median(np.array(ts), axis=0)
    out.append(f"{name} frame {t[0]:.3f} start {1000 * t[1]:.0f} us")
print(" | ".join(out), flush=True)
r.close()
