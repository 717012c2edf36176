import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)
W, H = 3840, 2160
sr = rt.SceneReader().readScene('/root/repo/scenes/scene7.test')
r = rt.Renderer(0); r.set_scene(sr); r.build_grid(15, 6)
d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for world, wg in ((8, 1), (8, 4), (1, 4)):
    r.render_device(d_rgb.data_ptr(), 0, W, H, rank=0, world=world, workgroups_per_cu=wg)
    st = r.render_device(d_rgb.data_ptr(), 0, W, H, rank=0, world=world, want_stats=True, count_work=True, workgroups_per_cu=wg)
    ev = list(st.wave_events); cy = np.array(list(st.wave_cycles), float)
    print(f"world {world} wg/cu {wg}: kernel {st.trace_ms:.3f} ms | rounds {ev[0]} walk iters {ev[1]} batches {ev[2]} updates {ev[3]} | "
          f"cycles per event: update {cy[0]/max(ev[3],1):.0f} walk-iter {cy[1]/max(ev[1],1):.0f} batch {cy[2]/max(ev[2],1):.0f} fold/round {cy[3]/max(ev[0],1):.0f} | split {np.round(cy/cy.sum(),3)}", flush=True)
