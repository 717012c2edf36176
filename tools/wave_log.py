"""Development: per-wave log of one instrumented frame of the persistent kernel (GPU box).

    python tools/wave_log.py [world] [workgroups_per_cu] [width height] [scene sub] -> summary of gpurun_out/wave_log.csv
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wg = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160)
scene, sub = (sys.argv[5], int(sys.argv[6])) if len(sys.argv) > 6 else ("scene7", 6)
out = os.path.join(ROOT, "gpurun_out", f"wave_log_w{world}_g{wg}.csv")
os.makedirs(os.path.dirname(out), exist_ok=True)
sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", scene + ".test"))
r = rt.Renderer(0)
r.set_scene(sr)
r.build_grid(15, sub)
d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for _ in range(2):
    r.render_device(d_rgb.data_ptr(), 0, W, H, rank=0, world=world, workgroups_per_cu=wg)
st = r.render_device(d_rgb.data_ptr(), 0, W, H, rank=0, world=world, want_stats=True, count_work=True, workgroups_per_cu=wg)
log = r.wave_log().astype(np.float64)
a = np.concatenate([np.arange(len(log))[:, None], log], axis=1)      # column 0: wave number
np.savetxt(out, a, delimiter=",", header="wave,start,exhausted,end,rounds,updates,pixels,deepest,helper_tasks", comments="")
a = a[a[:, 3] > 0]                                  # waves that ran
t0 = a[:, 1].min()
start, exh, end = (a[:, 1] - t0) / 100.0, (a[:, 2] - t0) / 100.0, (a[:, 3] - t0) / 100.0      # microseconds
print(f"{scene}, world {world}, {wg} workgroups per CU, {W}x{H}: instrumented kernel {st.trace_ms:.3f} ms, {len(a)} waves ran")
print(f"  start   : min {start.min():.0f} median {np.median(start):.0f} max {start.max():.0f} us")
print(f"  no more pixels after: median {np.median(exh[exh > 0]):.0f} max {exh.max():.0f} us")
print(f"  end     : percentiles 10/50/90/99/100 = {np.percentile(end, [10, 50, 90, 99, 100]).round(0)} us")
print(f"  rounds  : percentiles 50/90/99/100 = {np.percentile(a[:, 4], [50, 90, 99, 100])}; updates {np.percentile(a[:, 5], [50, 90, 99, 100])}")
print(f"  pixels per wave: mean {a[:, 6].mean():.0f} max {a[:, 6].max():.0f}; helper tasks per wave mean {a[:, 8].mean():.1f}")
late = a[end > np.percentile(end, 99)]
print(f"  last 1 % of the waves: rounds {late[:, 4].mean():.0f}, updates {late[:, 5].mean():.0f}, pixels {late[:, 6].mean():.0f}, deepest {late[:, 7].mean():.1f}, "
      f"us per round {((late[:, 3] - late[:, 1]) / 100.0 / late[:, 4]).mean():.2f}")
print(f"  all waves: us per round {((a[:, 3] - a[:, 1]) / 100.0 / a[:, 4]).mean():.2f}")
r.close()
