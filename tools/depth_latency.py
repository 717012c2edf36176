import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)
W, H = 3840, 2160
sr = rt.SceneReader().readScene('/root/repo/scenes/scene7.test')
r = rt.Renderer(0); r.set_scene(sr); r.build_grid(15, 6)
d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for world in (8, 1):
    for depth in (0, 1, 2, 3, 5):
        ms = []
        for _ in range(6):
            st = r.render_device(d_rgb.data_ptr(), 0, W, H, max_depth=depth, rank=0, world=world, want_stats=True)
            ms.append(st.trace_ms)
        print(f"world {world} max_depth {depth}: frame kernel {np.median(ms[2:]):.3f} ms, hits per level {list(st.hits_per_level)[:6]}", flush=True)
