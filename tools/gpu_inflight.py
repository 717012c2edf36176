"""Development measurement (GPU box): throughput of one rank's share of scene7 @ 4K when several frames are in flight
on their own streams and contexts.  python tools/gpu_inflight.py <world> <frames in flight> <workgroups per CU> [sub]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < 8 and "KEEP_HW_QUEUES" not in os.environ:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"        # as bench.py does: streams beyond the hardware queues share one
import torch

import raytracer_amd as rt
import _knobs  # noqa: F401  (RTX_* environment knobs -> render-call defaults)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
world, depth, wpc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sub = int(sys.argv[4]) if len(sys.argv) > 4 else 6
W, H = 3840, 2160
sr = rt.SceneReader().readScene(os.path.join(ROOT, "scenes", "scene7.test"))
rs, streams, bufs = [], [], []
for _ in range(depth):
    r = rt.Renderer(0)
    r.set_scene(sr)
    r.build_grid(15, sub, rt.GRID_NESTED)
    rs.append(r)
    streams.append(torch.cuda.Stream())
    bufs.append(torch.empty((H, W, 3), dtype=torch.float32, device="cuda"))


def run(n):
    for k in range(n):
        b = k % depth
        rs[b].render_device(bufs[b].data_ptr(), 0, W, H, rank=0, world=world, stream=streams[b].cuda_stream, workgroups_per_cu=wpc)
    torch.cuda.synchronize()


run(3 * depth)
t = time.perf_counter()
n = 60
run(n)
dt = (time.perf_counter() - t) / n * 1e3
print(f"world {world}, {depth} frame(s) in flight, {wpc} workgroup(s) per CU per frame, sub {sub}: {dt:.3f} ms per frame", flush=True)
