"""Development measurement (GPU box): frame time of one rank's share of scene7 @ 4K for 1, 2, 4, 8 ranks, each
rendered alone on this GPU (what every GPU of an N-GPU run does per frame, without the gather)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 3840, 2160
sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", "scene7.test"))
r = rt.Renderer(0)
r.set_scene(sr)
r.build_grid(15, 6)
d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for world in (1, 2, 4, 8):
    worst = 0.0
    for rank in range(world):
        ms = []
        for _ in range(6):
            st = r.render_device(d_rgb.data_ptr(), 0, W, H, rank=rank, world=world, want_stats=True)
            ms.append((st.frame_ms, st.trace_ms))
        m = np.median(np.array(ms[2:]), axis=0)
        worst = max(worst, m[0])
        if rank == 0:
            first = m
    print(f"world {world}: rank 0 frame {first[0]:.3f} ms (frame kernel {first[1]:.3f}), slowest rank {worst:.3f} ms, "
          f"ideal {2.75 / world:.3f} ms", flush=True)
r.close()
