"""Development measurement (GPU box): every benchmark configuration at its own size -- rays, counted work, kernel time of
the frame kernel (first-frame schedule, HIP events, median of nine launches), algorithmic flops and the fraction of the
fp32 vector peak, one JSON object per configuration on stdout.

    python tools/config_roofline.py > gpurun_out/configs.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)
from bench import PEAK_FP32_VALU_TFLOPS, PEAK_PACKED_UNFUSED_TFLOPS, algorithmic_flops

CONFIGS = [("scene1", 640, 480, 0), ("scene4-specular", 640, 480, 0), ("scene5", 640, 480, 0), ("scene6", 640, 480, 0),
           ("scene7", 3840, 2160, 6)]
r = rt.Renderer(0)
for name, W, H, sub in CONFIGS:
    sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", name + ".test"))
    r.set_scene(sr)
    gi = r.build_grid(15, sub, rt.GRID_NESTED)
    d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    for _ in range(3):
        r.render_device(d_rgb.data_ptr(), 0, W, H, tile_hint=False)
    st = r.render_device(d_rgb.data_ptr(), 0, W, H, want_stats=True, count_work=True, tile_hint=False)
    cold = [r.render_device(d_rgb.data_ptr(), 0, W, H, want_stats=True, tile_hint=False) for _ in range(9)]
    warm = [r.render_device(d_rgb.data_ptr(), 0, W, H, want_stats=True) for _ in range(9)]
    k_ms = float(np.median([s.trace_ms for s in cold]))
    f_ms = float(np.median([s.frame_ms for s in cold]))
    work = [a + b for a, b in zip(st.trace_work, st.shade_work)]
    flops = algorithmic_flops(work)
    print(json.dumps({
        "config": f"{name} @ {W}x{H}, depth 5, nested sub {gi.sub}", "rays": st.total_rays, "rays_walked": st.walked_rays,
        "frame_ms": f_ms, "k_frame_ms": k_ms, "frame_ms_with_previous_frame_hint": float(np.median([s.frame_ms for s in warm])),
        "Mrays_per_s": st.total_rays / f_ms / 1e3, "Mrays_walked_per_s": st.walked_rays / f_ms / 1e3,
        "work": dict(zip(("tri_tests", "sphere_tests", "xform_evals", "cell_steps", "fine_steps", "light_evals", "nearest_hits", "rays"),
                         (int(v) for v in work))),
        "wave_events": dict(zip(("rounds", "walk_iterations", "test_batches", "updates"), (int(v) for v in st.wave_events))),
        "algorithmic_flops": flops, "achieved_TFLOPs": flops / (k_ms * 1e-3) / 1e12,
        "frac_of_fp32_peak": flops / (k_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS,
        "frac_of_packed_unfused_peak": flops / (k_ms * 1e-3) / 1e12 / PEAK_PACKED_UNFUSED_TFLOPS,
        "grid": {"sub": gi.sub, "fine_refs": gi.n_fine_refs, "occupied_fine_cells": gi.n_occupied_fine_cells, "build_ms": gi.build_ms}}), flush=True)
r.close()
