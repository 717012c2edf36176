"""ctypes wrapper of the CPU oracle (oracle/libref_cpu*.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg as the checker / the timed CPU baseline.  The product package never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
COUNTER_NAMES = ("rays", "tri_tests", "sph_tests", "cell_steps", "light_evals", "hits",
                 "primary", "shadow", "reflect")


def build():
    """Compile the oracle libraries with the Makefile next to this file (g++ only, no GPU needed)."""
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


def _load(counting):
    name = "libref_cpu_count.so" if counting else "libref_cpu.so"
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.ref_load.restype = C.c_void_p
    lib.ref_load.argtypes = [C.c_char_p, C.c_int]
    lib.ref_free.argtypes = [C.c_void_p]
    lib.ref_info.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_output_name.restype = C.c_char_p
    lib.ref_output_name.argtypes = [C.c_void_p]
    lib.ref_camera.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_lights.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_prims.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    lib.ref_grid.restype = C.c_long
    lib.ref_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ref_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    lib.ref_prim_hit.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ref_camera_ray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.ref_render.restype = C.c_double
    lib.ref_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ref_quantize.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
    lib.ref_threads.restype = C.c_int
    lib.ref_set_schedule.argtypes = [C.c_int]
    return lib


_libs = {}


def _lib(counting=False):
    if counting not in _libs:
        _libs[counting] = _load(counting)
    return _libs[counting]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class RefScene:
    """One parsed scene held by the oracle (reference: SceneReader, src/readScene.h:25-183)."""

    def __init__(self, path, use_grid=True, counting=False):
        self.lib = _lib(counting)
        self.h = self.lib.ref_load(os.fsencode(path), 1 if use_grid else 0)
        if not self.h:
            raise FileNotFoundError(path)
        info = np.zeros(8, np.int32)
        self.lib.ref_info(self.h, _p(info))
        self.width, self.height, self.n_prims, self.n_lights, self.n_verts = (int(v) for v in info[:5])
        self.output = self.lib.ref_output_name(self.h).decode()

    def close(self):
        if self.h:
            self.lib.ref_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def camera(self):
        out = np.zeros(19, np.float32)
        self.lib.ref_camera(self.h, _p(out))
        return dict(eye=out[0:3], center=out[3:6], up=out[6:9], fov=float(out[9]),
                    u=out[10:13], v=out[13:16], w=out[16:19])

    def lights(self):
        out = np.zeros((max(self.n_lights, 1), 10), np.float32)
        self.lib.ref_lights(self.h, _p(out))
        return out[:self.n_lights]

    def prims(self):
        n = max(self.n_prims, 1)
        kind = np.zeros(n, np.int32)
        geom = np.zeros((n, 9), np.float32)
        mat = np.zeros((n, 13), np.float32)
        xf = np.zeros((n, 16), np.float32)
        inv = np.zeros((n, 16), np.float32)
        box = np.zeros((n, 6), np.float32)
        self.lib.ref_prims(self.h, _p(kind), _p(geom), _p(mat), _p(xf), _p(inv), _p(box))
        k = self.n_prims
        return dict(kind=kind[:k], geom=geom[:k], mat=mat[:k], xf=xf[:k], inv=inv[:k], box=box[:k])

    def grid(self):
        """(bounds[6], cell_start[3376], ids[total]) of the reference 15^3 grid, lists in file order."""
        bounds = np.zeros(6, np.float32)
        count = np.zeros(3375, np.int32)
        total = self.lib.ref_grid(self.h, _p(bounds), _p(count), None)
        ids = np.zeros(max(total, 1), np.int32)
        self.lib.ref_grid(self.h, _p(bounds), _p(count), _p(ids))
        start = np.zeros(3376, np.int64)
        np.cumsum(count, out=start[1:])
        return bounds, start, ids[:total]

    def intersect(self, O, D, t_limit=np.finfo(np.float32).max):
        O = np.asarray(O, np.float32)
        D = np.asarray(D, np.float32)
        out = np.zeros(9, np.float32)
        self.lib.ref_intersect(self.h, _p(O), _p(D), C.c_float(t_limit), _p(out))
        return dict(ok=bool(out[0]), t=out[1], p=out[2:5].copy(), n=out[5:8].copy(), prim=int(out[8]))

    def prim_hit(self, prim, O, D):
        O = np.asarray(O, np.float32)
        D = np.asarray(D, np.float32)
        out = np.zeros(5, np.float32)
        self.lib.ref_prim_hit(self.h, prim, _p(O), _p(D), _p(out))
        return dict(ok=bool(out[0]), t=out[1], n=out[2:5].copy())

    def camera_ray(self, width, height, x, y):
        out = np.zeros(3, np.float32)
        self.lib.ref_camera_ray(self.h, width, height, x, y, _p(out))
        return out

    def render(self, width=0, height=0, max_depth=5, row_start=0, row_step=1):
        """Returns (rgb[H,W,3] float32, ids[H,W] int32, seconds, counters dict).  Rows outside the subset
        row_start::row_step keep rgb = 0 and ids = -2."""
        W = width if width > 0 else self.width
        H = height if height > 0 else self.height
        rgb = np.zeros((H, W, 3), np.float32)
        ids = np.full((H, W), -2, np.int32)
        cnt = np.zeros(9, np.uint64)
        sec = self.lib.ref_render(self.h, W, H, max_depth, row_start, row_step, _p(rgb), _p(ids), _p(cnt))
        return rgb, ids, sec, dict(zip(COUNTER_NAMES, (int(v) for v in cnt)))


def quantize(rgb):
    """8-bit truncating conversion of the reference's writers (src/Image3f.h:72-74)."""
    rgb = np.ascontiguousarray(rgb, np.float32)
    out = np.zeros(rgb.shape, np.uint8)
    _lib().ref_quantize(_p(rgb), rgb.size, _p(out))
    return out


def set_schedule(reference_schedule, counting=False):
    """True: rows go to the threads in contiguous blocks, as the reference's plain `#pragma omp parallel for` hands them
    out (src/main.cpp:126) -- what the timed CPU baseline uses; False (default): one row at a time to whichever thread is
    free."""
    _lib(counting).ref_set_schedule(1 if reference_schedule else 0)


def threads():
    """OpenMP threads the oracle's pixel loop will use."""
    return int(_lib().ref_threads())
