"""Development (GPU box): the 640x480 configurations as lone first frames over the subdivision of the nested grid.
python tools/sweep_sub_small.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import raytracer_amd as rt
import _knobs  # noqa: F401

r = rt.Renderer(0)
for name in ("scene1", "scene4-specular", "scene5", "scene6"):
    sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", name + ".test"))
    r.set_scene(sr)
    line = [name]
    for sub in (0, 1, 2, 3, 4, 5, 6, 8):
        gi = r.build_grid(15, sub)
        for _ in range(2):
            r.render(640, 480, 5, tile_hint=False)
        ms = float(np.median([r.render(640, 480, 5, tile_hint=False)[2].frame_ms for _ in range(7)]))
        line.append(f"sub {sub}{'(auto ' + str(gi.sub) + ')' if sub == 0 else ''}: {ms:.3f}")
    print("  ".join(line), flush=True)
r.close()
