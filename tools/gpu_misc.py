"""Development timing of the pieces around the frame kernel (GPU box): host-output render (PCIe inclusive), the
8-bit conversion kernel, scene parse + upload + grid build.

    python tools/gpu_misc.py [scene] [W H]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = sys.argv[1] if len(sys.argv) > 1 else "scene7"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
t = time.perf_counter()
sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", scene + ".test"))
t_parse = time.perf_counter() - t
r = rt.Renderer(0)
t = time.perf_counter()
r.set_scene(sr)
t_upload = time.perf_counter() - t
for k in range(3):
    t = time.perf_counter()
    gi = r.build_grid(15, 6)
    t_build = time.perf_counter() - t
print(f"parse {t_parse * 1e3:.1f} ms, set_scene {t_upload * 1e3:.1f} ms, build_grid {t_build * 1e3:.1f} ms (device part {gi.build_ms:.1f} ms)")

r.render(W, H, want_ids=False)
ts = []
for _ in range(3):
    t = time.perf_counter()
    rgb, _, st = r.render(W, H, want_ids=False)
    ts.append(time.perf_counter() - t)
print(f"rtx_render to host memory: {np.median(ts) * 1e3:.1f} ms per frame ({st.total_rays / np.median(ts) / 1e6:.0f} Mrays/s PCIe inclusive)")

d_rgb = torch.from_numpy(rgb).cuda()
d_q = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for bgr in (False, True):
    r.quantize_device(d_rgb.data_ptr(), W * H, d_q.data_ptr(), bgr=bgr, stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        r.quantize_device(d_rgb.data_ptr(), W * H, d_q.data_ptr(), bgr=bgr, stream=stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"k_quantize bgr={bgr}: {ms * 1e3:.1f} us per 4K frame, {15.0 * W * H / (ms * 1e-3) / 1e9:.0f} GB/s of 15 B/pixel")
assert np.array_equal(d_q.cpu().numpy(), rt.quantize(rgb)[:, :, ::-1])
r.close()
