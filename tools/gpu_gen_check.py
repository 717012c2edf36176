"""Development check run on the GPU box: the generation pipeline against the default persistent kernel, bit for bit.

    python tools/gpu_gen_check.py [4k]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import raytracer_amd as rt
from raytracer_amd import api

api.DEFAULTS["round_limit"] = 1 << 23
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = [("scene1", 0, 0, 0), ("scene2", 0, 0, 0), ("scene3", 0, 0, 0), ("scene4-specular", 0, 0, 0), ("scene6", 0, 0, 0),
         ("scene5", 320, 240, 0), ("scene7", 160, 90, 0), ("scene7", 640, 480, 6)]
if "4k" in sys.argv:
    cases.append(("scene7", 3840, 2160, 6))
r = rt.Renderer(0)
bad = 0
for name, w, h, sub in cases:
    sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", name + ".test"))
    r.set_scene(sr)
    r.build_grid(15, sub)
    for mode in (rt.SEARCH_NESTED, rt.SEARCH_REFERENCE_LISTS):
        if mode == rt.SEARCH_REFERENCE_LISTS and w * h > 700000:
            continue
        rgb0, ids0, st0 = r.render(w, h, 5, mode)
        for waves in (4, 3):
            for count in (False, True):
                rgb, ids, st = r.render(w, h, 5, mode, generations=True, frame_waves=waves, count_work=count)
                ok = (np.array_equal(ids, ids0) and np.array_equal(rgb.view(np.int32), rgb0.view(np.int32)) and
                      (st.primary_rays, st.shadow_rays, st.reflection_rays) == (st0.primary_rays, st0.shadow_rays, st0.reflection_rays) and
                      list(st.hits_per_level) == list(st0.hits_per_level))
                bad += not ok
                print(f"{name} {rgb.shape[1]}x{rgb.shape[0]} search {mode} waves {waves} count {count}: identical={ok} "
                      f"frame {st.frame_ms:.3f} ms (default pipeline {st0.frame_ms:.3f})"
                      + ("" if ok else f"  ids differ {int((ids != ids0).sum())} rgb differ {int((rgb.view(np.int32) != rgb0.view(np.int32)).any(axis=2).sum())}"), flush=True)
r.close()
print("ALL IDENTICAL" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
