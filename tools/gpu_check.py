"""Development check run on the GPU box: GPU render vs CPU oracle for every config scene."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)
from oracle.ref_oracle import RefScene, quantize as ref_quantize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = [("scene1", 0, 0), ("scene2", 0, 0), ("scene3", 0, 0), ("scene4-specular", 0, 0), ("scene6", 0, 0),
         ("scene5", 320, 240), ("scene7", 160, 90)]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
r = rt.Renderer(0)
for name, w, h in cases:
    path = os.path.join(ROOT, "scenes", name + ".test")
    ref = RefScene(path, counting=True)
    t0 = time.time()
    rgb_ref, ids_ref, sec, cnt = ref.render(w, h)
    sr = rt.SceneReader().readScene(path)
    r.set_scene(sr)
    gi = r.build_grid(15, 0)
    bounds, rstart, rids = ref.grid()
    gstart, gids = r.download_grid(rt.GRID_REFERENCE)
    grid_ok = (np.array_equal(bounds, np.array(list(gi.lo) + list(gi.hi), np.float32)) and
               np.array_equal(rstart, gstart) and np.array_equal(rids, gids))
    print(f"{name}: prims {sr.n_prims} grid refs {gi.n_refs} fine(sub={gi.sub}) refs {gi.n_fine_refs} "
          f"build {gi.build_ms:.2f} ms grid_identical={grid_ok}  oracle {sec:.2f}s", flush=True)
    for mode, mname, serial, lq in [(rt.SEARCH_REFERENCE_LISTS, "ref-frame", False, False), (rt.SEARCH_NESTED, "nest-frame", False, False),
                                    (rt.SEARCH_REFERENCE_LISTS, "ref-levels", False, True), (rt.SEARCH_NESTED, "nest-levels", False, True),
                                    (rt.SEARCH_NESTED, "nest-serial", True, True)]:
        rgb, ids, st = r.render(w, h, 5, mode, count_work=True, serial_lanes=serial, level_queues=lq)
        rgb2, ids2, st2 = r.render(w, h, 5, mode, serial_lanes=serial, level_queues=lq)
        same_ids = np.array_equal(ids, ids_ref)
        n_id_diff = int((ids != ids_ref).sum())
        bit_same = np.array_equal(rgb.view(np.int32), rgb_ref.view(np.int32))
        q, qr = rt.quantize(rgb), ref_quantize(rgb_ref)
        d = np.abs(q.astype(int) - qr.astype(int)).max(axis=2)
        print(f"   {mname:9s} ids_identical={same_ids} ({n_id_diff} differ) float_bit_identical={bit_same} "
              f"pixels>0LSB={int((d > 0).sum())} pixels>1LSB={int((d > 1).sum())} maxabs={np.abs(rgb - rgb_ref).max():.3g} "
              f"frame {st2.frame_ms:.3f} ms (trace {st2.trace_ms:.3f} shade {st2.shade_ms:.3f} resolve {st2.resolve_ms:.3f}) "
              f"Mrays/s {st2.total_rays / st2.frame_ms / 1e3:.1f}", flush=True)
        print(f"      rays gpu p/s/r {st.primary_rays}/{st.shadow_rays}/{st.reflection_rays} oracle "
              f"{cnt['primary']}/{cnt['shadow']}/{cnt['reflect']}; tri {st.tri_tests} vs {cnt['tri_tests']}; sph "
              f"{st.sphere_tests} vs {cnt['sph_tests']}; cells {st.cell_steps} vs {cnt['cell_steps']}; light_evals "
              f"{st.light_evals} vs {cnt['light_evals']}; fine {st.fine_steps} xform {st.xform_evals}", flush=True)
        assert np.array_equal(rgb.view(np.int32), rgb2.view(np.int32)), "counting and plain kernels disagree"
r.close()
