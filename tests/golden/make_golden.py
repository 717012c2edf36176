"""Generates tests/golden/golden_<scene>.npz from the CPU oracle (oracle/ref_cpu.cpp).

The reference ships no bit-level goldens and cannot be built in this image (glm, OpenCV absent), so these
vectors pin the ORACLE, not the reference: they catch any later change of the oracle's arithmetic and let
the GPU tests run without recomputing the slow scenes.  Per scene: primary-hit id map (int32), 8-bit image,
float framebuffer and the ray / work counters, at the sizes below.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.ref_oracle import RefScene, quantize  # noqa: E402

# (scene, width, height): the four 640x480 configs at a quarter of their size, scene7 at 4K aspect
CASES = [("scene1", 160, 120), ("scene4-specular", 160, 120), ("scene5", 160, 120), ("scene6", 160, 120),
         ("scene7", 192, 108)]

for name, w, h in CASES:
    s = RefScene(os.path.join(ROOT, "scenes", name + ".test"), counting=True)
    rgb, ids, sec, cnt = s.render(w, h, 5)
    bounds, start, cell_ids = s.grid()
    out = os.path.join(HERE, f"golden_{name}.npz")
    np.savez_compressed(out, rgb=rgb, ids=ids, rgb8=quantize(rgb), width=w, height=h,
                        counters=np.array([cnt[k] for k in sorted(cnt)], np.int64), counter_names=np.array(sorted(cnt)),
                        grid_bounds=bounds, grid_refs=np.int64(len(cell_ids)),
                        grid_cell_count=np.diff(start).astype(np.int32))
    print(name, w, h, f"{sec:.2f}s", os.path.getsize(out), "bytes")
