"""Generates tests/golden/golden_scene7_4k.npz from the CPU oracle (oracle/ref_cpu.cpp): the whole scene7 frame at
the benchmark's size, 3840x2160, maximum depth 5 -- the pixel loop of src/main.cpp:126-133 on res/scene7.test with
the width and height overridden.

Like the small goldens these vectors pin the ORACLE, not the reference (which cannot be built in this image).  Held
per pixel: the primary-hit primitive index (int32) and the 8-bit image; of the float framebuffer only a SHA-256 and the
per-row sums (float64) are kept, the 99.5 MB themselves would not be a small fixture.

    python tests/golden/make_golden_4k.py        (about four minutes on eight cores)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.ref_oracle import RefScene, quantize  # noqa: E402

W, H, DEPTH = 3840, 2160, 5

s = RefScene(os.path.join(ROOT, "scenes", "scene7.test"), counting=True)
rgb, ids, sec, cnt = s.render(W, H, DEPTH)
out = os.path.join(HERE, "golden_scene7_4k.npz")
np.savez_compressed(out, ids=ids, rgb8=quantize(rgb), width=W, height=H, max_depth=DEPTH,
                    rgb_sha256=np.array(hashlib.sha256(rgb.tobytes()).hexdigest()),
                    rgb_row_sums=rgb.astype(np.float64).sum(axis=(1, 2)),
                    counters=np.array([cnt[k] for k in sorted(cnt)], np.int64), counter_names=np.array(sorted(cnt)))
print("scene7", W, H, f"{sec:.1f}s", os.path.getsize(out), "bytes")
