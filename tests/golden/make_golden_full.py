"""Generates tests/golden/golden_<scene>_full.npz from the CPU oracle (oracle/ref_cpu.cpp): scene5 and scene7 at the
size their own scene files give (640x480, `size` line of each file), depth 5 -- primary-hit ids and 8-bit image of every
pixel, SHA-256 and per-row sums of the float framebuffer, ray counters.  Like the other goldens these pin the ORACLE
(the reference cannot be built in this image); the GPU tests compare the HIP path with them at full size.

    python tests/golden/make_golden_full.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.ref_oracle import RefScene, quantize  # noqa: E402

for name in ("scene5", "scene7"):
    s = RefScene(os.path.join(ROOT, "scenes", name + ".test"), counting=True)
    w, h = 640, 480
    rgb, ids, sec, cnt = s.render(w, h, 5)
    out = os.path.join(HERE, f"golden_{name}_full.npz")
    np.savez_compressed(out, ids=ids, rgb8=quantize(rgb), width=w, height=h, max_depth=5,
                        rgb_sha256=hashlib.sha256(np.ascontiguousarray(rgb).tobytes()).hexdigest(),
                        rgb_row_sums=rgb.astype(np.float64).sum(axis=(1, 2)),
                        counters=np.array([cnt[k] for k in sorted(cnt)], np.int64), counter_names=np.array(sorted(cnt)))
    print(name, w, h, f"{sec:.2f}s", os.path.getsize(out), "bytes")
