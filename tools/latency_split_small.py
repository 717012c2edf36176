"""Development (GPU box): cycles per event of the instrumented frame kernel on the 640x480 configurations, lone frames.
python tools/latency_split_small.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)

W, H = 640, 480
r = rt.Renderer(0)
d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for name in ("scene1", "scene4-specular", "scene5", "scene6"):
    sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", name + ".test"))
    r.set_scene(sr)
    r.build_grid(15, 0)
    r.render_device(d_rgb.data_ptr(), 0, W, H, tile_hint=False)
    st = r.render_device(d_rgb.data_ptr(), 0, W, H, want_stats=True, count_work=True, tile_hint=False)
    ev = list(st.wave_events)
    cy = np.array(list(st.wave_cycles), float)
    print(f"{name}: kernel {st.trace_ms:.3f} ms | rounds {ev[0]} walk iters {ev[1]} batches {ev[2]} updates {ev[3]} | "
          f"cycles per event: update {cy[0] / max(ev[3], 1):.0f} walk-iter {cy[1] / max(ev[1], 1):.0f} batch {cy[2] / max(ev[2], 1):.0f} "
          f"fold/round {cy[3] / max(ev[0], 1):.0f} | split {np.round(cy / cy.sum(), 3)}", flush=True)
r.close()
