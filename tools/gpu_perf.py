"""Development timing run on the GPU box: scene7 @ 4K for several nested sub-resolutions and the reference lists."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raytracer_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = sys.argv[1] if len(sys.argv) > 1 else "scene7"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
subs = [int(s) for s in sys.argv[4].split(",")] if len(sys.argv) > 4 else [4, 6, 8, 12, 16]
do_ref = (sys.argv[5] != "noref") if len(sys.argv) > 5 else True
sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", scene + ".test"))
r = rt.Renderer(0)
r.set_scene(sr)
d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
d_ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
base_ids = None
base_rgb = None
for sub in subs:
    gi = r.build_grid(15, sub)
    st = r.render_device(d_rgb.data_ptr(), d_ids.data_ptr(), W, H, want_stats=True, count_work=True)
    torch.cuda.synchronize()
    times = []
    for it in range(5):
        st2 = r.render_device(d_rgb.data_ptr(), d_ids.data_ptr(), W, H, want_stats=True)
        times.append((st2.frame_ms, st2.trace_ms, st2.shade_ms, st2.resolve_ms))
    t = np.median(np.array(times), axis=0)
    ids = d_ids.cpu().numpy()
    rgb = d_rgb.cpu().numpy()
    if base_ids is None:
        base_ids, base_rgb = ids, rgb
    print(f"sub={sub:2d} fine refs {gi.n_fine_refs} build {gi.build_ms:.1f} ms | frame {t[0]:.3f} ms trace {t[1]:.3f} shade {t[2]:.3f} "
          f"resolve {t[3]:.3f} | rays {st.total_rays} ({st.primary_rays}/{st.shadow_rays}/{st.reflection_rays}) "
          f"Mrays/s {st.total_rays / t[0] / 1e3:.1f} | tri {st.tri_tests} sph {st.sphere_tests} cells {st.cell_steps} fine {st.fine_steps} "
          f"| same ids {np.array_equal(ids, base_ids)} same rgb {np.array_equal(rgb.view(np.int32), base_rgb.view(np.int32))}", flush=True)
if do_ref:
    t0 = time.time()
    st = r.render_device(d_rgb.data_ptr(), d_ids.data_ptr(), W, H, search=rt.SEARCH_REFERENCE_LISTS, want_stats=True, count_work=True)
    ids = d_ids.cpu().numpy()
    rgb = d_rgb.cpu().numpy()
    st2 = r.render_device(d_rgb.data_ptr(), d_ids.data_ptr(), W, H, search=rt.SEARCH_REFERENCE_LISTS, want_stats=True)
    print(f"ref-lists frame {st2.frame_ms:.3f} ms trace {st2.trace_ms:.3f} shade {st2.shade_ms:.3f} | Mrays/s {st.total_rays / st2.frame_ms / 1e3:.1f} "
          f"| tri {st.tri_tests} sph {st.sphere_tests} cells {st.cell_steps} | same ids as nested {np.array_equal(ids, base_ids)} "
          f"({int((ids != base_ids).sum())} differ) same rgb {np.array_equal(rgb.view(np.int32), base_rgb.view(np.int32))} "
          f"({int((rgb.view(np.int32) != base_rgb.view(np.int32)).any(axis=2).sum())} pixels differ)", flush=True)
    print("hits per level", list(st.hits_per_level)[:6])
r.close()
