"""Development sweep on the GPU box: the 640x480 configurations as lone frames, default pipeline against the generation
pipeline (register budget, start of the endgame).  python tools/ab_small_gen.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import raytracer_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = rt.Renderer(0)
for name in ("scene1", "scene4-specular", "scene5", "scene6"):
    sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", name + ".test"))
    r.set_scene(sr)
    r.build_grid(15, 0)

    def ms(**kw):
        for _ in range(2):
            r.render(640, 480, 5, tile_hint=False, **kw)
        return float(np.median([r.render(640, 480, 5, tile_hint=False, **kw)[2].frame_ms for _ in range(7)]))

    line = [f"{name}: default {ms(generations=False):.3f}"]
    for waves in (3,):
        for e in (8,):
            line.append(f"gen w{waves} e{e} {ms(generations=True, frame_waves=waves, endgame_eighths=e):.3f}")
    print("  ".join(line), flush=True)
r.close()
