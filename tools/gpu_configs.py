"""Development measurement (GPU box): the benchmark configurations at their own sizes -- GPU frame time (default
pipeline, nested search) next to the CPU oracle's time for the same frame on this box's host cores."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)
from oracle import ref_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = rt.Renderer(0)
for name in ("scene1", "scene4-specular", "scene5", "scene6"):
    path = os.path.join(ROOT, "scenes", name + ".test")
    sr = rt.SceneReader().readScene(path)
    r.set_scene(sr)
    gi = r.build_grid(15, 0)
    W, H = sr.width, sr.height
    d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    ms = []
    for _ in range(8):
        st = r.render_device(d_rgb.data_ptr(), 0, W, H, want_stats=True)
        ms.append(st.frame_ms)
    gpu_ms = float(np.median(ms[3:]))
    ref = ref_oracle.RefScene(path)
    rgb_ref, _, sec, _ = ref.render(W, H, 5)
    ref.close()
    same = np.abs(d_rgb.cpu().numpy() - rgb_ref).max()
    print(f"{name} {W}x{H}: GPU {gpu_ms:.3f} ms ({st.total_rays / gpu_ms / 1e3:.0f} Mrays/s), CPU oracle {sec * 1e3:.0f} ms on "
          f"{ref_oracle.threads()} threads ({st.total_rays / sec / 1e6:.2f} Mrays/s), ratio {sec * 1e3 / gpu_ms:.0f}x, "
          f"nested sub {gi.sub}, max |colour difference| {same:.1e}", flush=True)
r.close()
