"""Development: one rank's share of a small frame rendered by several contexts in flight, compared with the same share
rendered by one context alone; prints where they differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as rt
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
world, rank, depth, wpc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
W, H = 640, 360
sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", "scene7.test"))
rows = rt.local_rows(H, 8, rank, world)
rs, streams, bufs = [], [], []
for _ in range(depth):
    r = rt.Renderer(0); r.set_scene(sr); r.build_grid(15, 6, rt.GRID_NESTED)
    rs.append(r); streams.append(torch.cuda.Stream())
    bufs.append(torch.full((len(rows), W, 3), 7.0, dtype=torch.float32, device="cuda"))
ref = torch.full((len(rows), W, 3), 9.0, dtype=torch.float32, device="cuda")
rs[0].render_device(ref.data_ptr(), 0, W, H, rank=rank, world=world, tile_hint=False)
torch.cuda.synchronize()
ref = ref.cpu().numpy()
print("reference share: min", ref.min(), "max", ref.max(), flush=True)
for it in range(int(sys.argv[5]) if len(sys.argv) > 5 else 20):
    b = it % depth
    rs[b].render_device(bufs[b].data_ptr(), 0, W, H, rank=rank, world=world, stream=streams[b].cuda_stream, workgroups_per_cu=wpc, tile_hint=False)
torch.cuda.synchronize()
for b in range(depth):
    got = bufs[b].cpu().numpy()
    bad = np.argwhere((got != ref).any(axis=2))
    print(f"slot {b}: {len(bad)} pixels differ", "" if len(bad) == 0 else f"first {bad[:5].tolist()} values {got[tuple(bad[0])]} want {ref[tuple(bad[0])]}; rows {sorted(set(bad[:,0].tolist()))[:20]}", flush=True)
