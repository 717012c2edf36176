"""Development measurement (GPU box): grid build times of one scene, nested grid only and with the reference lists,
first build (allocations included) and rebuilds.  python tools/prof_build.py [scene] [sub] [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = sys.argv[1] if len(sys.argv) > 1 else "scene7"
sub = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", scene + ".test"))
for flags, label in ((rt.GRID_NESTED, "nested only"), (rt.GRID_REFERENCE | rt.GRID_NESTED, "reference lists + nested")):
    r = rt.Renderer(0)
    r.set_scene(sr)
    ms = [r.build_grid(15, sub, flags).build_ms for _ in range(n + 1)]
    print(f"{scene} sub={sub} {label}: first build {ms[0]:.2f} ms, rebuilds median {np.median(ms[1:]):.2f} ms (min {min(ms[1:]):.2f})", flush=True)
    r.close()
