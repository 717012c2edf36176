"""Development differential run on the GPU box: random scenes (the generator of tests/test_gpu_parity.py) rendered by
the persistent pipelines (per-lane paths with and without helpers, generations) in both builds of the frame kernel, against the CPU oracle and against
the level-queue pipeline bit for bit.

    python tools/gpu_fuzz.py [first_seed] [n_seeds]
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)
from oracle.ref_oracle import RefScene
from test_gpu_parity import _random_scene, FLOAT_TOL

first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
r = rt.Renderer(0)
bad = 0
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "random.test")
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        text = _random_scene(rng, int(rng.integers(0, 200)), int(rng.integers(0, 40)), int(rng.integers(0, 4)))
        open(path, "w").write(text)
        depth = int(rng.integers(0, 10))
        rgb_ref, ids_ref, _, cnt = RefScene(path, counting=True).render(0, 0, depth)
        r.set_scene(rt.SceneReader().readScene(path))
        r.build_grid(15, int(rng.choice([0, 1, 2, 3, 5, 6])))
        problems = []
        for mode, mode_name in ((rt.SEARCH_NESTED, "nested"), (rt.SEARCH_REFERENCE_LISTS, "reference lists")):
            want_rgb, want_ids, _ = r.render(0, 0, depth, mode, level_queues=True)
            if not np.array_equal(want_ids, ids_ref) or np.abs(want_rgb - rgb_ref).max() > FLOAT_TOL:
                problems.append(f"{mode_name}: levels vs oracle")
            for waves in (3, 4):
                for helpers, lanes in ((True, 0), (False, 0), ("early", 0), ("early", 24), ("early", 5), ("late", 0)):
                    for k in range(2):
                        rgb, ids, st = r.render(0, 0, depth, mode, frame_waves=waves, helpers=helpers, path_lanes=lanes)
                        if not np.array_equal(ids, want_ids) or not np.array_equal(rgb.view(np.int32), want_rgb.view(np.int32)):
                            problems.append(f"{mode_name}: frame waves={waves} helpers={helpers} lanes={lanes} pass={k}")
                        if (st.primary_rays, st.shadow_rays, st.reflection_rays) != (cnt["primary"], cnt["shadow"], cnt["reflect"]):
                            problems.append(f"{mode_name}: ray counts waves={waves} helpers={helpers}")
                for endgame in (-1, 0, 40):             # the generation pipeline: no endgame, the default, the whole frame
                    rgb, ids, st = r.render(0, 0, depth, mode, frame_waves=waves, generations=True, endgame_eighths=endgame)
                    if not np.array_equal(ids, want_ids) or not np.array_equal(rgb.view(np.int32), want_rgb.view(np.int32)):
                        problems.append(f"{mode_name}: generations waves={waves} endgame={endgame}")
                    if (st.primary_rays, st.shadow_rays, st.reflection_rays) != (cnt["primary"], cnt["shadow"], cnt["reflect"]):
                        problems.append(f"{mode_name}: ray counts generations waves={waves}")
        if problems:
            bad += 1
            print(f"seed {seed}: {problems}", flush=True)
        elif (seed - first) % 25 == 0:
            print(f"seed {seed} ok", flush=True)
print(f"{n} scenes, {bad} with differences")
r.close()
sys.exit(1 if bad else 0)
