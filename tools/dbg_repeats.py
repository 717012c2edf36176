"""Development hack (needs a variant library that counts them): how many of the lists a ray finds repeat the list it
tested just before."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import raytracer_amd as rt
r = rt.Renderer(0)
for name, W, H, sub in (("scene1", 640, 480, 0), ("scene4-specular", 640, 480, 0), ("scene5", 640, 480, 0), ("scene6", 640, 480, 0), ("scene7", 3840, 2160, 6)):
    d_rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", name + ".test"))
    r.set_scene(sr); r.build_grid(15, sub)
    st = r.render_device(d_rgb.data_ptr(), 0, W, H, want_stats=True, count_work=True, tile_hint=False, generations=False)
    print(name, "lists found", st.light_evals, "repeats of the list before", st.xform_evals, round(st.xform_evals / max(st.light_evals, 1), 3), "rounds", list(st.wave_events)[0], flush=True)
r.close()
