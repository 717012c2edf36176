"""ctypes binding of librtx.so and a host-side mirror of the reference's driver interface.

Reference interface mirrored (paths relative to the reference tree):
  SceneReader.readScene(filename)      src/readScene.h:45-182
  raytrace(scenefilename)              src/main.cpp:109-146
  Image3f(width, height).save(name)    src/Image3f.h:23-133
The arithmetic lives in the library; this module only moves pointers and sizes across the C ABI.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

SEARCH_REFERENCE_LISTS = 0
SEARCH_NESTED = 1
GRID_REFERENCE = 1
GRID_NESTED = 2
GRID_TEST_REPEATED_LISTS = 4
RENDER_COUNT_WORK = 1
RENDER_SERIAL_LANES = 2
ORDER_RGB, ORDER_BGR = 0, 1
RENDER_LEVEL_QUEUES = 4
RENDER_NO_TILE_HINT = 8
RENDER_NO_TILE_CULL = 16
RENDER_NO_HELPERS = 32
RENDER_WAVES_3 = 64
RENDER_WAVES_4 = 128
RENDER_HELPERS_EARLY = 256
RENDER_HELPERS_LATE = 512
RENDER_GENERATIONS = 1024
RENDER_NO_GENERATIONS = 2048

# Scheduling and tuning fields of rtx_render_params applied to every render call of this process unless the call
# names them itself.  The library reads no environment variables; development tools (tools/) fill this from theirs,
# and the test suite sets a round limit here so that a kernel that never ends cannot hang the machine.
# generations: None = the library picks the pipeline of the persistent launch by frame size, True / False force the
# generation pipeline / per-lane paths; endgame_eighths: 0 = default, < 0 = never (see include/rtx.h).
DEFAULTS = dict(tile_hint=True, tile_cull=True, helpers=True, frame_waves=0, walk_budget=0, ready_threshold=0,
                drain_eighths=0, path_lanes=0, round_limit=0, generations=None, endgame_eighths=0)


class RtxError(RuntimeError):
    def __init__(self, code, detail):
        super().__init__(f"rtx error {code}: {detail}")
        self.code = code


class _SceneView(C.Structure):
    _fields_ = [("n_prims", C.c_int32), ("n_xforms", C.c_int32), ("n_materials", C.c_int32), ("n_lights", C.c_int32),
                ("prim_kind", C.c_void_p), ("prim_geom", C.c_void_p), ("prim_xform", C.c_void_p),
                ("prim_material", C.c_void_p), ("xforms", C.c_void_p), ("materials", C.c_void_p),
                ("lights", C.c_void_p)]


class _Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("center", C.c_float * 3), ("up", C.c_float * 3), ("fov_deg", C.c_float)]


class GridInfo(C.Structure):
    _fields_ = [("lo", C.c_float * 3), ("hi", C.c_float * 3), ("rx", C.c_int32), ("ry", C.c_int32), ("rz", C.c_int32),
                ("sub", C.c_int32), ("n_cells", C.c_int64), ("n_refs", C.c_int64), ("n_fine_cells", C.c_int64),
                ("n_fine_refs", C.c_int64), ("n_occupied_fine_cells", C.c_int64), ("build_ms", C.c_float),
                ("skips_repeated_lists", C.c_int32)]


class _RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_depth", C.c_int32), ("eps_bias", C.c_float),
                ("search", C.c_int32), ("tile_rows", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32),
                ("flags", C.c_uint32), ("workgroups_per_cu", C.c_int32),
                ("walk_budget", C.c_int32), ("ready_threshold", C.c_int32), ("drain_eighths", C.c_int32),
                ("path_lanes", C.c_int32), ("round_limit", C.c_int64),
                ("endgame_eighths", C.c_int32), ("reserved1", C.c_int32)]


class RenderStats(C.Structure):
    _fields_ = [("primary_rays", C.c_int64), ("shadow_rays", C.c_int64), ("reflection_rays", C.c_int64),
                ("hits_per_level", C.c_int64 * 16),
                ("tri_tests", C.c_int64), ("sphere_tests", C.c_int64), ("xform_evals", C.c_int64),
                ("cell_steps", C.c_int64), ("fine_steps", C.c_int64), ("light_evals", C.c_int64),
                ("nearest_hits", C.c_int64),
                ("trace_work", C.c_int64 * 8), ("shade_work", C.c_int64 * 8), ("wave_events", C.c_int64 * 4), ("wave_cycles", C.c_int64 * 4),
                ("frame_ms", C.c_float), ("trace_ms", C.c_float), ("shade_ms", C.c_float), ("resolve_ms", C.c_float),
                ("level_trace_ms", C.c_float * 16), ("level_shade_ms", C.c_float * 16),
                ("trace_launches", C.c_int32), ("shade_launches", C.c_int32), ("resolve_launches", C.c_int32),
                ("pipeline", C.c_int32), ("traced_primary_rays", C.c_int64), ("culled_pixels", C.c_int64)]

    @property
    def total_rays(self):
        return self.primary_rays + self.shadow_rays + self.reflection_rays

    @property
    def walked_rays(self):
        """Rays that walked the grid: total_rays minus the camera rays answered by the screen-tile cull."""
        return self.traced_primary_rays + self.shadow_rays + self.reflection_rays

    def as_dict(self):
        arrays = ("hits_per_level", "trace_work", "shade_work", "wave_events", "wave_cycles", "level_trace_ms", "level_shade_ms")
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in arrays}
        for k in arrays:
            d[k] = list(getattr(self, k))
        d["total_rays"] = self.total_rays
        d["walked_rays"] = self.walked_rays
        return d


def lib_path():
    return os.environ.get("RTX_LIB_PATH") or os.path.join(_HERE, "librtx.so")


_lib = None


def _load():
    """Loads librtx.so; a missing library is an error (the HIP extension is the only compute path)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `make -C {_HERE}` (or __graft_entry__.build()); "
                          "there is no fallback implementation")
    lib = C.CDLL(path)
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    sig = {
        "rtx_strerror": (C.c_char_p, [C.c_int]),
        "rtx_last_error": (C.c_char_p, []),
        "rtx_scene_load": (C.c_int, [C.POINTER(vp), C.c_char_p]),
        "rtx_scene_parse": (C.c_int, [C.POINTER(vp), C.c_char_p, C.c_int64]),
        "rtx_scene_free": (None, [vp]),
        "rtx_scene_get_view": (C.c_int, [vp, C.POINTER(_SceneView)]),
        "rtx_scene_get_camera": (C.c_int, [vp, C.POINTER(_Camera)]),
        "rtx_scene_get_size": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32)]),
        "rtx_scene_get_output": (C.c_char_p, [vp]),
        "rtx_create": (C.c_int, [C.POINTER(vp), C.c_int]),
        "rtx_destroy": (None, [vp]),
        "rtx_set_scene": (C.c_int, [vp, C.POINTER(_SceneView)]),
        "rtx_build_grid": (C.c_int, [vp, i32, i32, i32, i32, C.c_uint32]),
        "rtx_get_grid_info": (C.c_int, [vp, C.POINTER(GridInfo)]),
        "rtx_download_grid": (C.c_int, [vp, C.c_uint32, vp, vp]),
        "rtx_local_rows": (C.c_int, [C.POINTER(_RenderParams), C.POINTER(i32), vp]),
        "rtx_render": (C.c_int, [vp, C.POINTER(_Camera), C.POINTER(_RenderParams), vp, vp, C.POINTER(RenderStats)]),
        "rtx_render_u8": (C.c_int, [vp, C.POINTER(_Camera), C.POINTER(_RenderParams), vp, i32, vp, C.POINTER(RenderStats)]),
        "rtx_multi_create": (C.c_int, [C.POINTER(vp), vp, i32]),
        "rtx_multi_destroy": (None, [vp]),
        "rtx_multi_device_count": (C.c_int, [vp]),
        "rtx_multi_set_scene": (C.c_int, [vp, C.POINTER(_SceneView)]),
        "rtx_multi_build_grid": (C.c_int, [vp, i32, i32, i32, i32, C.c_uint32]),
        "rtx_multi_get_grid_info": (C.c_int, [vp, C.POINTER(GridInfo)]),
        "rtx_multi_render": (C.c_int, [vp, C.POINTER(_Camera), C.POINTER(_RenderParams), vp, vp, C.POINTER(RenderStats)]),
        "rtx_multi_render_u8": (C.c_int, [vp, C.POINTER(_Camera), C.POINTER(_RenderParams), vp, i32, vp, C.POINTER(RenderStats)]),
        "rtx_multi_render_device": (C.c_int, [vp, C.POINTER(_Camera), C.POINTER(_RenderParams), vp, vp, C.POINTER(RenderStats)]),
        "rtx_save_image_u8": (C.c_int, [C.c_char_p, vp, i32, i32]),
        "rtx_render_device": (C.c_int, [vp, C.POINTER(_Camera), C.POINTER(_RenderParams), vp, vp, vp,
                                        C.POINTER(RenderStats)]),
        "rtx_render_device_noalloc": (C.c_int, [vp, C.POINTER(_Camera), C.POINTER(_RenderParams), vp, vp, vp]),
        "rtx_frame_status": (C.c_int, [vp]),
        "rtx_frame_times": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
        "rtx_download_wave_log": (C.c_int, [vp, vp, C.c_int64, C.POINTER(C.c_int64)]),
        "rtx_debug_set_frame_counters": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
        "rtx_intersect": (C.c_int, [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
        "rtx_camera_rays": (C.c_int, [vp, C.POINTER(_Camera), i32, i32, i32, vp, vp]),
        "rtx_quantize": (C.c_int, [vp, C.c_int64, vp]),
        "rtx_quantize_device": (C.c_int, [vp, vp, C.c_int64, vp, i32, vp]),
        "rtx_save_image": (C.c_int, [C.c_char_p, vp, i32, i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTED_SYMBOLS = (
    "rtx_strerror", "rtx_last_error", "rtx_scene_load", "rtx_scene_parse", "rtx_scene_free", "rtx_scene_get_view",
    "rtx_scene_get_camera", "rtx_scene_get_size", "rtx_scene_get_output", "rtx_create", "rtx_destroy",
    "rtx_set_scene", "rtx_build_grid", "rtx_get_grid_info", "rtx_download_grid", "rtx_local_rows", "rtx_render",
    "rtx_render_device", "rtx_render_device_noalloc", "rtx_frame_status", "rtx_frame_times", "rtx_download_wave_log", "rtx_debug_set_frame_counters", "rtx_intersect", "rtx_camera_rays", "rtx_quantize", "rtx_quantize_device",
    "rtx_save_image", "rtx_render_u8", "rtx_save_image_u8", "rtx_multi_create", "rtx_multi_destroy", "rtx_multi_device_count",
    "rtx_multi_set_scene", "rtx_multi_build_grid", "rtx_multi_get_grid_info", "rtx_multi_render", "rtx_multi_render_u8",
    "rtx_multi_render_device")


def _check(rc):
    if rc != 0:
        lib = _load()
        raise RtxError(rc, f"{lib.rtx_strerror(rc).decode()}: {lib.rtx_last_error().decode()}")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _np_from(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return np.zeros(shape, dtype)
    buf = (C.c_byte * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


class SceneReader:
    """Host scene layer (mirror of the reference's SceneReader).  ``readScene`` parses a ``.test`` file into flat
    arrays; the arrays are exposed as numpy copies for inspection and can also be supplied directly."""

    epsilonBias = 0.001  # src/readScene.h:37

    def __init__(self):
        self.width = 0
        self.height = 0
        self.outputFilename = ""
        self.camera = dict(eye=np.zeros(3, np.float32), center=np.zeros(3, np.float32),
                           up=np.zeros(3, np.float32), fov=0.0)
        self.prim_kind = np.zeros(0, np.int32)
        self.prim_geom = np.zeros((0, 9), np.float32)
        self.prim_xform = np.zeros(0, np.int32)
        self.prim_material = np.zeros(0, np.int32)
        self.xforms = np.zeros((0, 16), np.float32)
        self.materials = np.zeros((0, 13), np.float32)
        self.lights = np.zeros((0, 10), np.float32)

    def _take(self, handle):
        lib = _load()
        try:
            v = _SceneView()
            _check(lib.rtx_scene_get_view(handle, C.byref(v)))
            self.prim_kind = _np_from(v.prim_kind, (v.n_prims,), np.int32)
            self.prim_geom = _np_from(v.prim_geom, (v.n_prims, 9), np.float32)
            self.prim_xform = _np_from(v.prim_xform, (v.n_prims,), np.int32)
            self.prim_material = _np_from(v.prim_material, (v.n_prims,), np.int32)
            self.xforms = _np_from(v.xforms, (v.n_xforms, 16), np.float32)
            self.materials = _np_from(v.materials, (v.n_materials, 13), np.float32)
            self.lights = _np_from(v.lights, (v.n_lights, 10), np.float32)
            cam = _Camera()
            _check(lib.rtx_scene_get_camera(handle, C.byref(cam)))
            self.camera = dict(eye=np.array(cam.eye, np.float32), center=np.array(cam.center, np.float32),
                               up=np.array(cam.up, np.float32), fov=float(cam.fov_deg))
            w, h = C.c_int32(), C.c_int32()
            _check(lib.rtx_scene_get_size(handle, C.byref(w), C.byref(h)))
            self.width, self.height = w.value, h.value
            self.outputFilename = lib.rtx_scene_get_output(handle).decode()
        finally:
            lib.rtx_scene_free(handle)
        return self

    def readScene(self, filename):
        lib = _load()
        h = C.c_void_p()
        _check(lib.rtx_scene_load(C.byref(h), os.fsencode(filename)))
        return self._take(h)

    def parse(self, text):
        lib = _load()
        data = text.encode() if isinstance(text, str) else bytes(text)
        h = C.c_void_p()
        _check(lib.rtx_scene_parse(C.byref(h), data, len(data)))
        return self._take(h)

    @property
    def n_prims(self):
        return len(self.prim_kind)

    def _view(self):
        """(_SceneView, keep-alive list) over contiguous copies of the arrays."""
        arrs = [np.ascontiguousarray(self.prim_kind, np.int32), np.ascontiguousarray(self.prim_geom, np.float32),
                np.ascontiguousarray(self.prim_xform, np.int32), np.ascontiguousarray(self.prim_material, np.int32),
                np.ascontiguousarray(self.xforms, np.float32), np.ascontiguousarray(self.materials, np.float32),
                np.ascontiguousarray(self.lights, np.float32)]
        v = _SceneView(len(arrs[0]), len(arrs[4]), len(arrs[5]), len(arrs[6]), *[_ptr(a) for a in arrs])
        return v, arrs

    def _camera(self):
        c = _Camera()
        for k in range(3):
            c.eye[k] = float(self.camera["eye"][k])
            c.center[k] = float(self.camera["center"][k])
            c.up[k] = float(self.camera["up"][k])
        c.fov_deg = float(self.camera["fov"])
        return c


def _params(width, height, max_depth, eps_bias, search, tile_rows, rank, world, flags=0, workgroups_per_cu=0, **sched):
    """rtx_render_params from the call's arguments; scheduling / tuning keywords not given fall back to DEFAULTS."""
    unknown = set(sched) - set(DEFAULTS)
    if unknown:
        raise TypeError(f"unknown render option(s): {sorted(unknown)}")
    o = {k: (sched[k] if sched.get(k) is not None else v) for k, v in DEFAULTS.items()}
    flags |= ((0 if o["tile_hint"] else RENDER_NO_TILE_HINT) | (0 if o["tile_cull"] else RENDER_NO_TILE_CULL) |
              {0: RENDER_NO_HELPERS, 1: 0, True: 0, "late": RENDER_HELPERS_LATE, "early": RENDER_HELPERS_EARLY}[o["helpers"]] |
              {0: 0, 3: RENDER_WAVES_3, 4: RENDER_WAVES_4}[int(o["frame_waves"])] |
              {None: 0, True: RENDER_GENERATIONS, False: RENDER_NO_GENERATIONS}[o["generations"]])
    return _RenderParams(width, height, max_depth, eps_bias, search, tile_rows, rank, world, flags, workgroups_per_cu,
                         int(o["walk_budget"]), int(o["ready_threshold"]), int(o["drain_eighths"]), int(o["path_lanes"]), int(o["round_limit"]),
                         int(o["endgame_eighths"]), 0)


def local_rows(height, tile_rows=8, rank=0, world=1, width=1):
    """Global row numbers rendered by `rank` under the interleaved row-tile split."""
    lib = _load()
    p = _params(width, height, 0, 0.0, SEARCH_NESTED, tile_rows, rank, world)
    n = C.c_int32()
    _check(lib.rtx_local_rows(C.byref(p), C.byref(n), None))
    rows = np.zeros(max(n.value, 1), np.int32)
    _check(lib.rtx_local_rows(C.byref(p), C.byref(n), _ptr(rows)))
    return rows[:n.value]


class Renderer:
    """One GPU context: scene upload, grid build, render."""

    def __init__(self, device=0):
        lib = _load()
        self._h = C.c_void_p()
        _check(lib.rtx_create(C.byref(self._h), device))
        self.device = device
        self.scene = None

    def close(self):
        if getattr(self, "_h", None):
            _load().rtx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, scene):
        v, keep = scene._view()
        _check(_load().rtx_set_scene(self._h, C.byref(v)))
        self.scene = scene
        return self

    def build_grid(self, res=15, sub=0, flags=GRID_REFERENCE | GRID_NESTED):
        r = (res, res, res) if np.isscalar(res) else tuple(res)
        _check(_load().rtx_build_grid(self._h, r[0], r[1], r[2], sub, flags))
        return self.grid_info()

    def grid_info(self):
        gi = GridInfo()
        _check(_load().rtx_get_grid_info(self._h, C.byref(gi)))
        return gi

    def download_grid(self, which=GRID_REFERENCE):
        gi = self.grid_info()
        n_cells = gi.n_cells if which == GRID_REFERENCE else gi.n_fine_cells
        n_refs = gi.n_refs if which == GRID_REFERENCE else gi.n_fine_refs
        start = np.zeros(n_cells + 1, np.int32)
        ids = np.zeros(max(n_refs, 1), np.int32)
        _check(_load().rtx_download_grid(self._h, which, _ptr(start), _ptr(ids)))
        return start, ids[:n_refs]

    def render(self, width=0, height=0, max_depth=5, search=SEARCH_NESTED, tile_rows=8, rank=0, world=1,
               want_ids=True, count_work=False, camera=None, eps_bias=SceneReader.epsilonBias, serial_lanes=False, level_queues=False,
               **sched):
        """Renders this rank's rows into full-size host arrays.  Returns (rgb[H,W,3], ids[H,W] or None, stats).
        `sched`: scheduling / tuning keywords of DEFAULTS (tile_hint, tile_cull, helpers, frame_waves, ...)."""
        sc = self.scene
        W = width if width > 0 else sc.width
        H = height if height > 0 else sc.height
        cam = camera if camera is not None else sc._camera()
        flags = ((RENDER_COUNT_WORK if count_work else 0) | (RENDER_SERIAL_LANES if serial_lanes else 0) |
                 (RENDER_LEVEL_QUEUES if level_queues else 0))
        p = _params(W, H, max_depth, eps_bias, search, tile_rows, rank, world, flags, **sched)
        rgb = np.zeros((H, W, 3), np.float32)
        ids = np.full((H, W), -2, np.int32) if want_ids else None
        st = RenderStats()
        _check(_load().rtx_render(self._h, C.byref(cam), C.byref(p), _ptr(rgb), _ptr(ids), C.byref(st)))
        return rgb, ids, st

    def render_u8(self, width=0, height=0, max_depth=5, search=SEARCH_NESTED, bgr=False, want_ids=False, camera=None,
                  eps_bias=SceneReader.epsilonBias, **sched):
        """The frame converted to 8 bits per channel on the device; returns (rgb8[H,W,3] uint8, ids or None, stats)."""
        sc = self.scene
        W = width if width > 0 else sc.width
        H = height if height > 0 else sc.height
        cam = camera if camera is not None else sc._camera()
        p = _params(W, H, max_depth, eps_bias, search, 8, 0, 1, 0, **sched)
        rgb8 = np.zeros((H, W, 3), np.uint8)
        ids = np.full((H, W), -2, np.int32) if want_ids else None
        st = RenderStats()
        _check(_load().rtx_render_u8(self._h, C.byref(cam), C.byref(p), _ptr(rgb8), ORDER_BGR if bgr else ORDER_RGB, _ptr(ids),
                                     C.byref(st)))
        return rgb8, ids, st

    def render_device(self, d_rgb_ptr, d_ids_ptr, width, height, max_depth=5, search=SEARCH_NESTED, tile_rows=8,
                      rank=0, world=1, stream=None, want_stats=False, count_work=False, camera=None,
                      eps_bias=SceneReader.epsilonBias, serial_lanes=False, level_queues=False, workgroups_per_cu=0, **sched):
        """Enqueues one frame writing DEVICE memory (raw pointers, e.g. torch tensor data_ptr())."""
        cam = camera if camera is not None else self.scene._camera()
        flags = ((RENDER_COUNT_WORK if count_work else 0) | (RENDER_SERIAL_LANES if serial_lanes else 0) |
                 (RENDER_LEVEL_QUEUES if level_queues else 0))
        p = _params(width, height, max_depth, eps_bias, search, tile_rows, rank, world, flags, workgroups_per_cu, **sched)
        st = RenderStats() if want_stats else None
        _check(_load().rtx_render_device(self._h, C.byref(cam), C.byref(p), C.c_void_p(d_rgb_ptr),
                                         C.c_void_p(d_ids_ptr) if d_ids_ptr else None,
                                         C.c_void_p(stream) if stream else None,
                                         C.byref(st) if st is not None else None))
        return st

    def render_device_noalloc(self, d_rgb_ptr, d_ids_ptr, width, height, max_depth=5, search=SEARCH_NESTED, tile_rows=8,
                              rank=0, world=1, stream=None, camera=None, eps_bias=SceneReader.epsilonBias, workgroups_per_cu=0,
                              flags=0, **sched):
        """render_device through the entry point that allocates nothing (scratch in the library's per-GPU array)."""
        cam = camera if camera is not None else self.scene._camera()
        p = _params(width, height, max_depth, eps_bias, search, tile_rows, rank, world, flags, workgroups_per_cu, **sched)
        _check(_load().rtx_render_device_noalloc(self._h, C.byref(cam), C.byref(p), C.c_void_p(d_rgb_ptr),
                                                 C.c_void_p(d_ids_ptr) if d_ids_ptr else None,
                                                 C.c_void_p(stream) if stream else None))

    def frame_status(self):
        """Waits for this context's last asynchronous frame; raises RtxError (RTX_E_LIMIT) if it was abandoned."""
        _check(_load().rtx_frame_status(self._h))

    def frame_times(self):
        """(frame_ms, kernel_ms) of this context's last frame from the library's HIP events; waits for that frame."""
        a, b = C.c_float(), C.c_float()
        _check(_load().rtx_frame_times(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def debug_set_frame_counters(self, frames_before_wrap, tickets_before_wrap):
        """Test aid: the context's next frame is this many frames / tickets before its never-reset counters wrap."""
        _check(_load().rtx_debug_set_frame_counters(self._h, frames_before_wrap, tickets_before_wrap))

    def wave_log(self):
        """[n_waves, 8] uint64 log of the last frame rendered with count_work (start, exhausted, end, rounds, updates,
        pixels, deepest level, helper tasks per wave)."""
        n = C.c_int64()
        _check(_load().rtx_download_wave_log(self._h, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 8), np.uint64)
        _check(_load().rtx_download_wave_log(self._h, _ptr(out), n.value, C.byref(n)))
        return out[:n.value]

    def quantize_device(self, d_rgb_ptr, n_pixels, d_out_ptr, bgr=False, stream=None):
        """Enqueues the 8-bit conversion of a DEVICE float framebuffer into DEVICE bytes (raw pointers)."""
        _check(_load().rtx_quantize_device(self._h, C.c_void_p(d_rgb_ptr), n_pixels, C.c_void_p(d_out_ptr),
                                           ORDER_BGR if bgr else ORDER_RGB, C.c_void_p(stream) if stream else None))

    def intersect(self, origins, dirs, t_limit=None, search=SEARCH_NESTED):
        O = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        D = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = len(O)
        tl = np.ascontiguousarray(t_limit, np.float32) if t_limit is not None else None
        t = np.zeros(n, np.float32)
        prim = np.zeros(n, np.int32)
        pos = np.zeros((n, 3), np.float32)
        nrm = np.zeros((n, 3), np.float32)
        _check(_load().rtx_intersect(self._h, search, n, _ptr(O), _ptr(D), _ptr(tl), _ptr(t), _ptr(prim), _ptr(pos),
                                     _ptr(nrm)))
        return t, prim, pos, nrm

    def camera_rays(self, width, height, xy, camera=None):
        xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
        cam = camera if camera is not None else self.scene._camera()
        out = np.zeros((len(xy), 3), np.float32)
        _check(_load().rtx_camera_rays(self._h, C.byref(cam), width, height, len(xy), _ptr(xy), _ptr(out)))
        return out


class MultiRenderer:
    """Several GPUs of one node behind one handle (rtx_multi_*): scene and grid replicated, interleaved row tiles, rows
    copied to the first GPU.  `devices`: list of device ids; an id may repeat (ranks then share that GPU)."""

    def __init__(self, devices):
        lib = _load()
        ids = np.ascontiguousarray(devices, np.int32)
        self._h = C.c_void_p()
        _check(lib.rtx_multi_create(C.byref(self._h), _ptr(ids), len(ids)))
        self.devices = list(int(d) for d in ids)
        self.scene = None

    def close(self):
        if getattr(self, "_h", None):
            _load().rtx_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, scene):
        v, keep = scene._view()
        _check(_load().rtx_multi_set_scene(self._h, C.byref(v)))
        self.scene = scene
        return self

    def build_grid(self, res=15, sub=0, flags=GRID_REFERENCE | GRID_NESTED):
        r = (res, res, res) if np.isscalar(res) else tuple(res)
        _check(_load().rtx_multi_build_grid(self._h, r[0], r[1], r[2], sub, flags))
        gi = GridInfo()
        _check(_load().rtx_multi_get_grid_info(self._h, C.byref(gi)))
        return gi

    def render(self, width=0, height=0, max_depth=5, search=SEARCH_NESTED, tile_rows=8, want_ids=True, camera=None,
               eps_bias=SceneReader.epsilonBias, u8=False, **sched):
        """(rgb[H,W,3] float32 -- or uint8 with u8=True --, ids[H,W] or None, stats summed over the GPUs)."""
        sc = self.scene
        W = width if width > 0 else sc.width
        H = height if height > 0 else sc.height
        cam = camera if camera is not None else sc._camera()
        p = _params(W, H, max_depth, eps_bias, search, tile_rows, 0, 1, 0, **sched)
        rgb = np.zeros((H, W, 3), np.uint8 if u8 else np.float32)
        ids = np.full((H, W), -2, np.int32) if want_ids else None
        st = RenderStats()
        if u8:
            _check(_load().rtx_multi_render_u8(self._h, C.byref(cam), C.byref(p), _ptr(rgb), ORDER_RGB, _ptr(ids), C.byref(st)))
        else:
            _check(_load().rtx_multi_render(self._h, C.byref(cam), C.byref(p), _ptr(rgb), _ptr(ids), C.byref(st)))
        return rgb, ids, st

    def render_device(self, d_rgb_ptr, d_ids_ptr, width, height, max_depth=5, search=SEARCH_NESTED, tile_rows=8, camera=None,
                      eps_bias=SceneReader.epsilonBias, want_stats=False, **sched):
        """The assembled frame left in device memory of the first listed GPU (raw pointers); complete on return."""
        cam = camera if camera is not None else self.scene._camera()
        p = _params(width, height, max_depth, eps_bias, search, tile_rows, 0, 1, 0, **sched)
        st = RenderStats() if want_stats else None
        _check(_load().rtx_multi_render_device(self._h, C.byref(cam), C.byref(p), C.c_void_p(d_rgb_ptr),
                                               C.c_void_p(d_ids_ptr) if d_ids_ptr else None, C.byref(st) if st is not None else None))
        return st


def quantize(rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    out = np.zeros(rgb.shape, np.uint8)
    _check(_load().rtx_quantize(_ptr(rgb), rgb.size, _ptr(out)))
    return out


def save_image(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w = rgb.shape[:2]
    _check(_load().rtx_save_image(os.fsencode(path), _ptr(rgb), w, h))


def save_image_u8(path, rgb8):
    """Writes an 8-bit RGB image [h, w, 3] as PNG (or PPM by extension), as rtx_save_image does after its conversion."""
    rgb8 = np.ascontiguousarray(rgb8, np.uint8)
    h, w = rgb8.shape[:2]
    _check(_load().rtx_save_image_u8(os.fsencode(path), _ptr(rgb8), w, h))


class Image3f:
    """Float RGB framebuffer with the reference's accessors (src/Image3f.h:23-60) and a save() without OpenCV."""

    def __init__(self, width, height):
        self.width, self.height = width, height
        self.data = np.zeros((height, width, 3), np.float32)

    def setAt(self, x, y, color):
        self.data[y, x] = color

    def getAt(self, x, y):
        return self.data[y, x]

    def save(self, filename):
        save_image(filename, self.data)

    def write3fPpm(self, filename):
        save_image(filename if filename.lower().endswith(".ppm") else filename + ".ppm", self.data)


def raytrace(scenefilename, width=0, height=0, max_depth=5, device=0, search=SEARCH_NESTED, save=True):
    """Mirror of raytrace() (src/main.cpp:109-146): parse, render on the GPU, save when the scene names an output.
    Returns (Image3f, RenderStats)."""
    sr = SceneReader().readScene(scenefilename)
    r = Renderer(device)
    try:
        r.set_scene(sr)
        r.build_grid(15, 0, GRID_NESTED if search == SEARCH_NESTED else GRID_REFERENCE)
        rgb, _, st = r.render(width, height, max_depth, search, want_ids=False)
    finally:
        r.close()
    img = Image3f(rgb.shape[1], rgb.shape[0])
    img.data = rgb
    if save and sr.outputFilename:
        img.save(sr.outputFilename)
    return img, st
