"""Development (GPU box): host time of one rtx_render_device call through the python binding (no statistics, frames
enqueued back to back on one stream), and of the C call alone.  python tools/host_overhead.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import raytracer_amd as rt

sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", "scene7.test"))
r = rt.Renderer(0)
r.set_scene(sr)
r.build_grid(15, 6)
W, H = 3840, 2160
d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for world in (1, 8):
    for _ in range(5):
        r.render_device(d_rgb.data_ptr(), 0, W, H, rank=0, world=world, workgroups_per_cu=1)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        r.render_device(d_rgb.data_ptr(), 0, W, H, rank=0, world=world, workgroups_per_cu=1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"world {world}: host {1e6 * (t1 - t0) / n:.1f} us per call to enqueue, {1e3 * (t2 - t0) / n:.3f} ms per frame in all", flush=True)
r.close()
