"""Development: what bit-parity costs.  Renders the benchmark frame (scene7 @ 3840x2160, depth 5, sub 6) with the library
named by RTX_LIB_PATH (e.g. a variant built with fused multiply-adds, tools/variant.sh fma -ffp-contract=fast) and
compares it with the committed oracle frame: primary-hit ids that flip, pixels off by more than one 8-bit step; then
times three first frames in flight.   RTX_LIB_PATH=.../variants/fma.so python tools/parity_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as rt
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(ROOT, "tests", "golden", "golden_scene7_4k.npz"))
W, H = int(g["width"]), int(g["height"])
sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", "scene7.test"))
r = rt.Renderer(0); r.set_scene(sr); r.build_grid(15, 6, rt.GRID_NESTED)
rgb, ids, st = r.render(W, H, 5, rt.SEARCH_NESTED, tile_hint=False)
d = np.abs(rt.quantize(rgb).astype(np.int16) - g["rgb8"].astype(np.int16)).max(axis=2)
print(f"library {rt.lib_path()}")
print(f"primary-hit ids that differ from the oracle: {int((ids != g['ids']).sum())} of {W * H}; pixels off by > 1 step: {int((d > 1).sum())}, by 1 step: {int((d == 1).sum())}; "
      f"rays {st.total_rays} (oracle {int(g['counters'][list(g['counter_names']).index('primary')]) + int(g['counters'][list(g['counter_names']).index('shadow')]) + int(g['counters'][list(g['counter_names']).index('reflect')])})")
rs, streams, bufs = [r], [torch.cuda.Stream()], [torch.empty((H, W, 3), dtype=torch.float32, device="cuda")]
for _ in range(2):
    x = rt.Renderer(0); x.set_scene(sr); x.build_grid(15, 6, rt.GRID_NESTED)
    rs.append(x); streams.append(torch.cuda.Stream()); bufs.append(torch.empty((H, W, 3), dtype=torch.float32, device="cuda"))
def run(n):
    for k in range(n):
        b = k % 3
        rs[b].render_device(bufs[b].data_ptr(), 0, W, H, stream=streams[b].cuda_stream, workgroups_per_cu=2, tile_hint=False)
    torch.cuda.synchronize()
run(9)
t = time.perf_counter(); run(60); dt = (time.perf_counter() - t) / 60 * 1e3
print(f"three first frames in flight: {dt:.3f} ms per frame")
