"""Development measurement (GPU box): a camera orbit around scene7 at 4K, frame times with and without the
previous-frame scheduling hint.  python tools/gpu_orbit.py [degrees per frame] [frames]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
step = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
W, H = 3840, 2160
sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", "scene7.test"))
r = rt.Renderer(0)
r.set_scene(sr)
r.build_grid(15, 6)
d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
eye = np.array(sr.camera["eye"], np.float64)
centre = np.array(sr.camera["center"], np.float64)
radius = np.linalg.norm((eye - centre)[[0, 2]])


def orbit(hint):
    ms = []
    for k in range(n):
        a = math.radians(step * k)
        cam = sr._camera()
        cam.eye[0] = float(centre[0] + radius * math.sin(a))
        cam.eye[2] = float(centre[2] + radius * math.cos(a))
        st = r.render_device(d_rgb.data_ptr(), 0, W, H, camera=cam, want_stats=True, tile_hint=hint)
        ms.append(st.frame_ms)
    return np.array(ms)


for hint in (False, True):
    ms = orbit(hint)
    print(f"{step} degrees per frame, hint {'on ' if hint else 'off'}: first frame {ms[0]:.2f} ms, "
          f"frames 2..{n}: mean {ms[1:].mean():.2f} ms (min {ms[1:].min():.2f}, max {ms[1:].max():.2f})")
r.close()
