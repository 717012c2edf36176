"""MI355X-native renderer for the per-pixel trace loop of the CSE167x-style Whitted ray tracer.

Everything here is a thin ctypes layer over ``librtx.so`` (C ABI in ``include/rtx.h``); the compute
runs in hand-written HIP kernels for gfx950.  There is no CPU path: importing works without a GPU
(so host-side scene parsing and image output can be used and tested), but creating a ``Renderer``
without a HIP device raises.
"""
from .api import (  # noqa: F401
    GridInfo,
    Image3f,
    MultiRenderer,
    Renderer,
    RenderStats,
    RtxError,
    SceneReader,
    SEARCH_NESTED,
    SEARCH_REFERENCE_LISTS,
    GRID_NESTED,
    GRID_TEST_REPEATED_LISTS,
    GRID_REFERENCE,
    lib_path,
    local_rows,
    quantize,
    raytrace,
    save_image,
    save_image_u8,
)
