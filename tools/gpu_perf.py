"""Development timing run on the GPU box: one scene at one size for several nested sub-resolutions, both kernel
families (wave-cooperative / one lane per ray), optionally the reference lists.

    python tools/gpu_perf.py scene7 3840 2160 8,12 [noref] [noserial]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = sys.argv[1] if len(sys.argv) > 1 else "scene7"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
subs = [int(s) for s in sys.argv[4].split(",")] if len(sys.argv) > 4 else [8, 12]
do_ref = "noref" not in sys.argv
do_serial = "noserial" not in sys.argv
do_levels = "nolevels" not in sys.argv
sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", scene + ".test"))
r = rt.Renderer(0)
r.set_scene(sr)
d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
d_ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
base = None


def run(search, serial, label, lq=True):
    global base
    r.render_device(d_rgb.data_ptr(), d_ids.data_ptr(), W, H, search=search, serial_lanes=serial, level_queues=lq)   # warm
    st = r.render_device(d_rgb.data_ptr(), d_ids.data_ptr(), W, H, search=search, want_stats=True, count_work=True,
                         serial_lanes=serial, level_queues=lq)
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        s2 = r.render_device(d_rgb.data_ptr(), d_ids.data_ptr(), W, H, search=search, want_stats=True, serial_lanes=serial, level_queues=lq)
        times.append((s2.frame_ms, s2.trace_ms, s2.shade_ms, s2.resolve_ms))
    t = np.median(np.array(times), axis=0)
    ids, rgb = d_ids.cpu().numpy(), d_rgb.cpu().numpy()
    if base is None:
        base = (ids, rgb)
    same_ids = np.array_equal(ids, base[0])
    same_rgb = np.array_equal(rgb.view(np.int32), base[1].view(np.int32))
    print(f"{label}: frame {t[0]:.3f} ms trace {t[1]:.3f} shade {t[2]:.3f} resolve {t[3]:.3f} | rays {st.total_rays} "
          f"Mrays/s {st.total_rays / t[0] / 1e3:.1f} | tri {st.tri_tests} sph {st.sphere_tests} cells {st.cell_steps} "
          f"fine {st.fine_steps} | wave events (rounds, walk iters, test batches, updates) {list(st.wave_events)} cycles (update, walk, test, fold) {[round(c / max(1, sum(st.wave_cycles)), 3) for c in st.wave_cycles]} | same ids {same_ids} same rgb {same_rgb}", flush=True)
    print("     per level trace ms", [round(v, 3) for v in list(s2.level_trace_ms)[:6]], "shade ms",
          [round(v, 3) for v in list(s2.level_shade_ms)[:6]], "hits", list(s2.hits_per_level)[:6], flush=True)


for sub in subs:
    gi = r.build_grid(15, sub)
    print(f"sub={gi.sub}: fine cells {gi.n_fine_cells} occupied {gi.n_occupied_fine_cells} refs {gi.n_fine_refs} "
          f"build {gi.build_ms:.1f} ms", flush=True)
    run(rt.SEARCH_NESTED, False, f"  nested sub={gi.sub} frame ", lq=False)
    if do_levels:
        run(rt.SEARCH_NESTED, False, f"  nested sub={gi.sub} levels")
    if do_serial:
        run(rt.SEARCH_NESTED, True, f"  nested sub={gi.sub} serial")
if do_ref:
    run(rt.SEARCH_REFERENCE_LISTS, False, "  reference lists frame ", lq=False)
    run(rt.SEARCH_REFERENCE_LISTS, False, "  reference lists levels")
    if do_serial:
        run(rt.SEARCH_REFERENCE_LISTS, True, "  reference lists serial")
r.close()
