/*
 * rtx.h -- C ABI of the MI355X renderer for the per-pixel trace loop.
 *
 * The reference renderer has no FFI or plug-in seam: its hot path is the OpenMP pixel loop of
 * raytrace() (src/main.cpp:126-133 in the reference tree), which reads everything a SceneReader
 * holds after readScene() + Camera::updateAxes() (src/main.cpp:110-112) and fills an Image3f
 * (src/main.cpp:119,131).  This header cuts the boundary around that whole loop: plain pointers
 * and sizes in, a float RGB framebuffer (and optionally the primary-hit primitive index map) out.
 * No C++ or torch types cross it; every entry point returns 0 on success or a negative RTX_E_* code,
 * and no exception ever leaves the library.
 *
 * Two layers:
 *   rtx_scene_*  host-side scene layer (the reference's SceneReader: .test parser, transform stack,
 *                materials, lights, camera) producing flat arrays;
 *   rtx_*        the device layer: upload, regular-grid build on the GPU, render.
 * A context is bound to one GPU and is not thread-safe; use one context per GPU (one process per
 * GPU in the multi-GPU launch, each rendering its interleaved row tiles).
 */
#ifndef RTX_H_
#define RTX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_OK            0
#define RTX_E_INVALID    -1   /* bad argument (null pointer, out-of-range index or size)        */
#define RTX_E_IO         -2   /* scene file could not be opened / image could not be written    */
#define RTX_E_DEVICE     -3   /* a HIP runtime call failed (rtx_last_error() has the detail)    */
#define RTX_E_STATE      -4   /* call order violated (e.g. render before scene/grid are set)    */
#define RTX_E_NOMEM      -5   /* host or device allocation failed                               */
#define RTX_E_LIMIT      -6   /* size beyond a documented maximum                               */

const char* rtx_strerror(int code);
/* Detail text of the most recent failing call on this thread ("" if none). */
const char* rtx_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Flat scene description (borrowed host pointers; every rtx_* call that takes a view copies what it
 * needs before returning).  Replaces the object graph SceneReader owns: `geometries` (heap Triangle /
 * Sphere objects, each carrying its own Material and mat4, src/readScene.h:104-123), `lights`
 * (src/readScene.h:79-96).  Primitive order is file order; a primitive's index is its identity
 * (the reference's only identity is the Material* inside each primitive, src/geometries.h:234,308).
 * ---------------------------------------------------------------------------------------------- */
#define RTX_PRIM_TRIANGLE 0
#define RTX_PRIM_SPHERE   1
#define RTX_LIGHT_POINT        0   /* LightType::POINT        src/geometries.h:354-357 */
#define RTX_LIGHT_DIRECTIONAL  1   /* LightType::DIRECTIONAL                           */

typedef struct rtx_scene_view {
    int32_t n_prims, n_xforms, n_materials, n_lights;
    const int32_t* prim_kind;      /* [n_prims] RTX_PRIM_*                                               */
    const float*   prim_geom;      /* [n_prims][9]  triangle: A,B,C (object space); sphere: centre, r, 0 */
    const int32_t* prim_xform;     /* [n_prims] index into xforms                                        */
    const int32_t* prim_material;  /* [n_prims] index into materials                                     */
    const float*   xforms;         /* [n_xforms][16] object->world, column-major (glm::mat4 layout)      */
    const float*   materials;      /* [n_materials][13] ambient, diffuse, specular, emission (rgb), shininess
                                      (struct Material, src/geometries.h:38-56)                           */
    const float*   lights;         /* [n_lights][10] kind, position-or-direction xyz, colour rgb, attenuation
                                      c0 c1 c2 (struct Light, src/geometries.h:359-364)                   */
} rtx_scene_view;

/* struct Camera as parsed (src/geometries.h:321-330); the frame axes of Camera::updateAxes() (:332-337)
 * are derived inside the library. */
typedef struct rtx_camera {
    float eye[3], center[3], up[3];
    float fov_deg;
} rtx_camera;

/* ------------------------------------------------------------------------------------------------
 * Host scene layer -- replaces SceneReader::readScene(filename, useGrid) (src/readScene.h:45-182).
 * Same directives, same defaults (diffuse 1 1 1, others 0, attenuation 1 0 0, shininess 40,
 * :46-51), same silent ignoring of unknown directives (`maxdepth`, `maxverts`, comments, :172-173).
 * Differences: an unreadable file is an error (RTX_E_IO) instead of an empty scene (:57-59), and
 * vertex indices are range-checked (RTX_E_INVALID) instead of read unchecked (:108).
 * ---------------------------------------------------------------------------------------------- */
typedef struct rtx_scene rtx_scene;

int  rtx_scene_load(rtx_scene** out, const char* path);
/* Parses scene text held in memory (same grammar as a .test file). */
int  rtx_scene_parse(rtx_scene** out, const char* text, int64_t n_bytes);
void rtx_scene_free(rtx_scene* s);
/* The arrays stay owned by the scene and valid until rtx_scene_free(). */
int  rtx_scene_get_view(const rtx_scene* s, rtx_scene_view* out);
int  rtx_scene_get_camera(const rtx_scene* s, rtx_camera* out);
/* `size` directive (src/readScene.h:69-72); 0 x 0 if the file has none. */
int  rtx_scene_get_size(const rtx_scene* s, int32_t* width, int32_t* height);
/* `output` directive (:169-171); "" if absent.  Owned by the scene. */
const char* rtx_scene_get_output(const rtx_scene* s);

/* ------------------------------------------------------------------------------------------------
 * Device layer
 * ---------------------------------------------------------------------------------------------- */
typedef struct rtx_ctx rtx_ctx;

/* Binds a context to GPU `device_id`.  Fails with RTX_E_DEVICE when no HIP device is usable:
 * there is no CPU fallback. */
int  rtx_create(rtx_ctx** out, int device_id);
void rtx_destroy(rtx_ctx* ctx);

/* Uploads the scene (replaces the geometry/light state trace() reads through SceneReader&,
 * src/main.cpp:47-107).  Derived per-primitive data (edge vectors, triangle normals, per-transform
 * inverses, glm::inverse call sites src/geometries.h:28,179-180) is computed once here. */
int  rtx_set_scene(rtx_ctx* ctx, const rtx_scene_view* view);

/* Grid::Grid (src/grid.h:36-49) on the GPU: per-primitive world boxes, scene bounds, then count /
 * prefix-scan / scatter of primitive references into rx*ry*rz cells.
 *   RTX_GRID_REFERENCE  cell lists exactly as the reference bins them (boxes whose maximum is seeded with
 *                       FLT_MIN, src/geometries.h:146-152; bounds padded by 0.001, src/grid.h:31);
 *                       needed by RTX_SEARCH_REFERENCE_LISTS.
 *   RTX_GRID_NESTED     additionally a fine grid of (rx*sub, ry*sub, rz*sub) cells nested in the coarse
 *                       cells, binned by tight boxes; needed by RTX_SEARCH_NESTED.  sub in [1, 64];
 *                       sub = 0 lets the library choose (cube root of the mean number of tight-box
 *                       references per occupied coarse cell).
 * The reference always uses rx = ry = rz = 15 (src/grid.h:37). */
#define RTX_GRID_REFERENCE 1u
#define RTX_GRID_NESTED    2u
/* with RTX_GRID_NESTED: rays test every list they meet, also one they tested in the cell before (see
 * rtx_grid_info.skips_repeated_lists); for A/B runs and tests -- the frames are the same either way */
#define RTX_GRID_TEST_REPEATED_LISTS 4u
int  rtx_build_grid(rtx_ctx* ctx, int32_t rx, int32_t ry, int32_t rz, int32_t sub, uint32_t flags);

typedef struct rtx_grid_info {
    float   lo[3], hi[3];          /* Grid::start_pos / end_pos (src/grid.h:12-13)                 */
    int32_t rx, ry, rz, sub;
    int64_t n_cells, n_refs;       /* coarse cells, references in the reference lists (0 if not built) */
    int64_t n_fine_cells, n_fine_refs;   /* fine cells incl. padding (32 * ceil(sub^3 / 32) per coarse cell) */
    int64_t n_occupied_fine_cells;
    float   build_ms;              /* GPU time of the last rtx_build_grid                           */
    int32_t skips_repeated_lists;  /* nested grid: 1 if a ray that has tested a fine cell's list skips the cells after it
                                      that hold the very same list (chosen where at least a quarter of the occupied
                                      cells share their list with another cell and the scene has several transforms:
                                      spheres and large triangles in a coarse grid); same hits either way */
} rtx_grid_info;
int  rtx_get_grid_info(const rtx_ctx* ctx, rtx_grid_info* out);
/* Copies a grid's CSR arrays to the host.  which = RTX_GRID_REFERENCE or RTX_GRID_NESTED.
 * cell_start has n_cells+1 (n_fine_cells+1) entries, prim_ids n_refs (n_fine_refs).  Coarse cell (x,y,z) is
 * entry x + rx*y + rx*ry*z (Grid::getOffsetAtIndices, src/grid.h:61-63); fine cell (lx,ly,lz) of coarse cell c
 * is entry c * (n_fine_cells / n_cells) + lx + sub*(ly + sub*lz).  Within a cell the ids are ascending, i.e. in
 * file order as in the reference (src/grid.h:46-48,65-68). */
int  rtx_download_grid(const rtx_ctx* ctx, uint32_t which, int32_t* cell_start, int32_t* prim_ids);

/* How the primitives of one coarse cell are searched (Container::intersect, src/geometries.h:173-204):
 *   RTX_SEARCH_REFERENCE_LISTS  every primitive in the reference's cell list is tested (the reference's work);
 *   RTX_SEARCH_NESTED           a 3D-DDA over the fine cells nested in the coarse cell visits only primitives
 *                               near the ray; same per-primitive arithmetic and acceptance rule, so the same
 *                               nearest hit.
 * The coarse walk (Grid::traverseGrid, src/grid.h:149-257) is identical in both. */
#define RTX_SEARCH_REFERENCE_LISTS 0
#define RTX_SEARCH_NESTED          1

#define RTX_RENDER_COUNT_WORK 1u   /* run the instrumented kernels and fill the work counters of rtx_stats */
#define RTX_RENDER_LEVEL_QUEUES 4u /* level-synchronous pipeline: global ray / hit queues in device memory, one
                                      trace and one shade launch per recursion level, then resolve launches;
                                      default is one persistent launch per frame (per-wave path pools)        */
#define RTX_RENDER_SERIAL_LANES 2u /* one lane walks and tests its own ray alone (no sharing of candidate tests
                                      across the wave); same results, kept for A/B timing                     */
/* Scheduling switches of the persistent frame kernel.  None of them changes a single bit of the frame: they choose
 * which pixels are answered by a fill instead of a walk, in which order the rest is traced and by which build of the
 * kernel.  All default to "on" / "automatic" (flag clear). */
#define RTX_RENDER_NO_TILE_HINT  8u  /* order the screen tiles without the path depths the previous frame of this
                                        context recorded: what a first frame, or a cut to another camera, costs  */
#define RTX_RENDER_NO_TILE_CULL 16u  /* trace every 8x8 tile of the screen window, also those that can see no
                                        occupied fine cell                                                        */
#define RTX_RENDER_NO_HELPERS   32u  /* a path's own lane traces its shadow rays and its mirror ray one after the
                                        other, as trace() does (src/main.cpp:85-97)                               */
#define RTX_RENDER_HELPERS_EARLY 256u /* lanes without a path take over the shadow rays of fresh hits from the start of
                                        the frame ...                                                             */
#define RTX_RENDER_HELPERS_LATE  512u /* ... or only once their wave can draw no new pixels; default: by frame size  */
#define RTX_RENDER_WAVES_3      64u  /* force the build held to 168 registers (3 waves per SIMD) ...              */
#define RTX_RENDER_WAVES_4     128u  /* ... or the one held to 128 (4 waves per SIMD); default: chosen by frame size */

#define RTX_RENDER_GENERATIONS 1024u  /* generation pipeline: a wave traces 64 rays of one kind at a time (nearest-hit
                                         rays, then the shadow rays of their hits) and keeps its paths in request
                                         stacks in device memory, so that every state update runs on a full wave ... */
#define RTX_RENDER_NO_GENERATIONS 2048u /* ... or never; default: chosen by frame size for launches that share the GPU
                                         with other frames (workgroups_per_cu given), off for a lone frame          */

typedef struct rtx_render_params {
    int32_t  width, height;        /* full image; y = 0 is the top row (src/main.cpp:127-131)           */
    int32_t  max_depth;            /* recursion budget of trace(); the reference always passes 5 (:131)  */
    float    eps_bias;             /* SceneReader::epsilonBias = 0.001f (src/readScene.h:37)            */
    int32_t  search;               /* RTX_SEARCH_*                                                      */
    int32_t  tile_rows;            /* rows per tile of the multi-GPU split; <= 0 means 8                */
    int32_t  rank, world;          /* this context renders tiles t with t % world == rank; 0,1 = all    */
    uint32_t flags;                /* RTX_RENDER_*                                                      */
    int32_t  workgroups_per_cu;    /* persistent frame kernel: workgroups launched per compute unit; <= 0 means 4.
                                      A caller that keeps several frames in flight on their own streams and
                                      contexts gives each a share (e.g. 1) so that their kernels run side by side;
                                      such a launch runs the generation pipeline when its frame is large enough
                                      (see RTX_RENDER_GENERATIONS)                                              */
    /* tuning of the persistent frame kernel (results never depend on it); <= 0 selects the default */
    int32_t  walk_budget;          /* grid steps a lane may take per round (3)                              */
    int32_t  ready_threshold;      /* lanes with a finished ray that trigger a state update (56)            */
    int32_t  drain_eighths;        /* ... once no new pixels are left: that many eighths of the live lanes (2) */
    int32_t  path_lanes;           /* paths a wave carries at a time; its other lanes stay free for the shadow rays of
                                      fresh hits from the start of the frame (default: 64, 48 or 32 by frame size) */
    int64_t  round_limit;          /* rounds a wave of the frame kernel may run before the frame is abandoned with
                                      RTX_E_LIMIT; <= 0 derives it from the frame: the smaller of what no valid frame
                                      can exceed (every ray the frame can hold times every step a ray can take) and
                                      max(2^24, 16 rounds per ray of a wave's even share of the pixels) -- a working
                                      wave stays far below it, a kernel that would never end stops within seconds */
    int32_t  endgame_eighths;      /* generation pipeline: pixels left per lane of the launch, in eighths, from which a
                                      wave stops stocking up on paths and drains (both kinds of ray in one
                                      generation): shortens the end of a frame that has the GPU to itself; < 0 = never;
                                      0 selects the default (8 for a lone frame, never with workgroups_per_cu set) */
    int32_t  reserved1;
} rtx_render_params;

typedef struct rtx_stats {
    /* rays = calls the reference would make to scene_content->intersect (src/main.cpp:57,70,82) */
    int64_t primary_rays, shadow_rays, reflection_rays;
    int64_t hits_per_level[16];
    /* filled only with RTX_RENDER_COUNT_WORK: whole frame, then split by kernel family in the order
     * tri_tests, sphere_tests, xform_evals, cell_steps, fine_steps, light_evals, nearest_hits, rays traced */
    int64_t tri_tests, sphere_tests, xform_evals, cell_steps, fine_steps, light_evals, nearest_hits;
    int64_t trace_work[8], shade_work[8];
    int64_t wave_events[4];        /* wave-cooperative kernels: rounds, walk-loop iterations, 64-pair test batches,
                                      state updates -- each counted once per wave                               */
    int64_t wave_cycles[4];        /* persistent frame kernel: shader-clock cycles summed over waves spent in state
                                      updates, walking, pair tests, folding results (counting builds only)     */
    float   frame_ms;              /* first kernel to framebuffer complete on the device (HIP events)     */
    float   trace_ms, shade_ms, resolve_ms;   /* per kernel family, summed over levels; the persistent frame
                                                 kernel is reported as trace_ms with trace_launches = 1   */
    float   level_trace_ms[16], level_shade_ms[16];   /* the same per level                              */
    int32_t trace_launches, shade_launches, resolve_launches;
    int32_t pipeline;              /* which pipeline rendered the frame: 0 persistent frame kernel, 1 the same in
                                      generations (RTX_RENDER_GENERATIONS or chosen by frame size), 2 level queues */
    /* primary_rays counts every pixel of this rank, as the reference casts one camera ray per pixel
     * (src/main.cpp:126-133).  The persistent frame kernel answers camera rays that cannot meet any primitive
     * (outside the screen window of the grid box / of every occupied fine cell) with the miss colour without
     * walking the grid: traced_primary_rays are the camera rays that were walked, culled_pixels the rest
     * (primary_rays = traced_primary_rays + culled_pixels; every shadow and mirror ray is walked). */
    int64_t traced_primary_rays, culled_pixels;
} rtx_stats;

/* Number of image rows this rank renders, and their global row numbers (ascending). rows may be NULL. */
int  rtx_local_rows(const rtx_render_params* p, int32_t* n_rows, int32_t* rows);

/* The pixel loop (src/main.cpp:126-133) for this rank's rows, results copied to the host.
 * rgb_out: width*height*3 floats, row-major, RGB, y = 0 top -- only this rank's rows are written.
 * prim_id_out (nullable): width*height, primary-hit primitive index, -1 = miss. */
int  rtx_render(rtx_ctx* ctx, const rtx_camera* cam, const rtx_render_params* p,
                float* rgb_out, int32_t* prim_id_out, rtx_stats* stats);

/* The same with the image converted to 8 bits per channel on the device before it crosses to the host -- the image the
 * reference saves (Image3f::save via get_3b_bgr_buffer, src/Image3f.h:128-133,154-170: order RTX_ORDER_BGR; write3fPpm,
 * :63-80: RTX_ORDER_RGB), a quarter of the bytes of the float frame.  rgb8_out: width*height*3 bytes. */
int  rtx_render_u8(rtx_ctx* ctx, const rtx_camera* cam, const rtx_render_params* p,
                   uint8_t* rgb8_out, int32_t order, int32_t* prim_id_out, rtx_stats* stats);

/* rtx_render_device for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc, code that
 * must not disturb another thread's stream capture with an allocation).  It has no counterpart in the reference.  The
 * call itself has not been exercised under stream capture: it records, and for the context's previous frame queries,
 * HIP events of the context.  The call allocates nothing and frees nothing, on
 * the host or on the device: the work cursors, the screen-tile map / classes / list / per-tile depths and the path
 * terms beyond the levels kept in LDS -- which rtx_render_device keeps in buffers the context allocates on first use --
 * live in a 10.6 MB array the library holds on every GPU it runs on for the life of the process.
 *   Limits (RTX_E_LIMIT beyond): max_depth <= 6; ceil(width / 8) * ceil(height / 8) <= 131072 (3840 x 2160 fits).
 *   Not available (RTX_E_INVALID): RTX_RENDER_LEVEL_QUEUES, RTX_RENDER_SERIAL_LANES, RTX_RENDER_COUNT_WORK; no stats.
 *   At most 1024 workgroups are launched (workgroups_per_cu is clamped).
 * The scene and grid must be set before (rtx_set_scene, rtx_build_grid: those do allocate).  Results are those of
 * rtx_render_device bit for bit.  All contexts of a GPU share that GPU's array: calls that overlap in time on one
 * GPU, from whatever contexts or streams, race on it and must be serialised by the caller. */
int  rtx_render_device_noalloc(rtx_ctx* ctx, const rtx_camera* cam, const rtx_render_params* p,
                               float* d_rgb_local, int32_t* d_prim_id_local, void* stream);

/* A frame is abandoned when a wave of the persistent frame kernel exceeds rtx_render_params.round_limit: its
 * remaining pixels keep the fill colour.  rtx_render and rtx_render_device with `stats` report that as RTX_E_LIMIT
 * themselves; after an asynchronous rtx_render_device this call waits for the context's last frame and returns
 * RTX_OK or RTX_E_LIMIT for it (the next render call on the context reports it as well). */
int  rtx_frame_status(rtx_ctx* ctx);

/* Same, leaving the result on the device: d_rgb_local is n_local_rows*width*3 floats of DEVICE memory,
 * d_prim_id_local (nullable) n_local_rows*width int32, rows in the order rtx_local_rows() reports.
 * Work is enqueued on `stream` (a hipStream_t, NULL = default stream) and NOT waited for, except that
 * `stats` (nullable) forces a wait at the end of the call so the numbers are final. */
int  rtx_render_device(rtx_ctx* ctx, const rtx_camera* cam, const rtx_render_params* p,
                       float* d_rgb_local, int32_t* d_prim_id_local, void* stream, rtx_stats* stats);

/* Device times of the context's last frame, from the HIP events the library records on the frame's stream around it:
 * frame_ms from before the fills to the completed framebuffer, kernel_ms the persistent frame kernel alone (the span
 * of the trace / shade launches for the level-queue pipelines).  Waits for that frame.  For callers that keep frames
 * in flight with rtx_render_device and want per-launch durations without a statistics call per frame (which would
 * wait); either pointer may be NULL.  It has no counterpart in the reference (which prints one wall time per run,
 * src/main.cpp:122-139). */
int  rtx_frame_times(rtx_ctx* ctx, float* frame_ms, float* kernel_ms);

/* Development aid: per-wave log of the last frame rendered with RTX_RENDER_COUNT_WORK by the persistent frame
 * kernel.  Eight values per wave: start, cursor exhausted, end (100 MHz clock), rounds, state updates, pixels drawn,
 * deepest level, helper tasks.  out holds 8 * max_waves values; *n_waves receives the number of waves logged. */
int  rtx_download_wave_log(rtx_ctx* ctx, uint64_t* out, int64_t max_waves, int64_t* n_waves);

/* Test aid: moves the never-reset counters of the context's frame start close to where they wrap around, so that a
 * test can render across the wrap.  The persistent pipelines clear nothing per frame: screen tiles and per-tile path
 * depths carry the frame's number, which starts over after 2^26 frames, and the frame-start kernel's ticket counters
 * are 32-bit words that simply wrap (a few thousand tickets per frame: every few million frames).  After this call the
 * context's next frame is `frames_before_wrap` frames before the frame number starts over and `tickets_before_wrap`
 * tickets before the ticket counters wrap.  Needs a frame rendered on the context before (the counters live in
 * buffers that the first frame allocates); waits for the device.  It has no counterpart in the reference. */
int  rtx_debug_set_frame_counters(rtx_ctx* ctx, uint32_t frames_before_wrap, uint32_t tickets_before_wrap);

/* Single-ray probes for tests: the device evaluates Grid::intersect (src/grid.h:259-261) for n rays.
 * origins/dirs: [n][3] host floats (directions are used as given, not normalised); t_limit: [n] or NULL
 * (= FLT_MAX).  out_t [n], out_prim [n] (-1 = miss), out_pos / out_normal [n][3] (nullable). */
int  rtx_intersect(rtx_ctx* ctx, int32_t search, int32_t n, const float* origins, const float* dirs,
                   const float* t_limit, float* out_t, int32_t* out_prim, float* out_pos, float* out_normal);

/* Camera::getRayAt (src/geometries.h:339-351) for n pixels, evaluated on the device: xy [n][2] -> dirs [n][3]. */
int  rtx_camera_rays(rtx_ctx* ctx, const rtx_camera* cam, int32_t width, int32_t height, int32_t n,
                     const int32_t* xy, float* dirs);

/* ------------------------------------------------------------------------------------------------
 * Several GPUs of one node behind one handle.  The reference has no counterpart (its only parallelism is OpenMP over the
 * rows of the pixel loop, src/main.cpp:126).  The scene and its grid are replicated on every GPU, the frame is split
 * into interleaved tiles of `tile_rows` rows (tile t -> the (t mod N)-th device), every GPU renders its tiles and copies
 * them into the frame on the first GPU over its point-to-point link to it; rank and world of rtx_render_params are
 * ignored.  Frames are bit-identical to one GPU's.  A device id may be listed more than once (the ranks then share
 * that GPU).  Not thread-safe per handle.
 * ---------------------------------------------------------------------------------------------- */
typedef struct rtx_multi rtx_multi;
int  rtx_multi_create(rtx_multi** out, const int32_t* device_ids, int32_t n_devices);
void rtx_multi_destroy(rtx_multi* m);
int  rtx_multi_device_count(const rtx_multi* m);
int  rtx_multi_set_scene(rtx_multi* m, const rtx_scene_view* view);
int  rtx_multi_build_grid(rtx_multi* m, int32_t rx, int32_t ry, int32_t rz, int32_t sub, uint32_t flags);
int  rtx_multi_get_grid_info(const rtx_multi* m, rtx_grid_info* out);          /* of the first GPU's copy */
/* rgb_out / rgb8_out / prim_id_out: the whole image in host memory, as for rtx_render / rtx_render_u8.  stats
 * (nullable): ray counts summed over the GPUs, trace_ms of the slowest GPU's kernel, frame_ms host wall time. */
int  rtx_multi_render(rtx_multi* m, const rtx_camera* cam, const rtx_render_params* p,
                      float* rgb_out, int32_t* prim_id_out, rtx_stats* stats);
int  rtx_multi_render_u8(rtx_multi* m, const rtx_camera* cam, const rtx_render_params* p,
                         uint8_t* rgb8_out, int32_t order, int32_t* prim_id_out, rtx_stats* stats);
/* d_rgb (width*height*3 floats) / d_prim_id (nullable): device memory of the FIRST listed GPU; complete on return. */
int  rtx_multi_render_device(rtx_multi* m, const rtx_camera* cam, const rtx_render_params* p,
                             float* d_rgb, int32_t* d_prim_id, rtx_stats* stats);

/* ------------------------------------------------------------------------------------------------
 * Image output -- replaces Image3f::write3fPpm / save (src/Image3f.h:63-80,128-133,154-170) without OpenCV.
 * Conversion to 8 bits truncates, (unsigned char)(clamp(c) * 255), as the reference does (:72-74,163-165).
 * ---------------------------------------------------------------------------------------------- */
int  rtx_quantize(const float* rgb, int64_t n_values, uint8_t* out);
/* The same conversion on the device, asynchronous on `stream` (a hipStream_t, NULL = default stream): d_rgb is
 * n_pixels*3 floats of device memory (16-byte aligned), d_out n_pixels*3 bytes of device memory (4-byte aligned).
 * RTX_ORDER_RGB keeps the channel order (write3fPpm, :63-80); RTX_ORDER_BGR swaps red and blue inside each pixel,
 * the layout get_3b_bgr_buffer hands to OpenCV (:154-170).  A multi-GPU run converts before the gather and so moves
 * a quarter of the bytes. */
#define RTX_ORDER_RGB 0
#define RTX_ORDER_BGR 1
int  rtx_quantize_device(rtx_ctx* ctx, const float* d_rgb, int64_t n_pixels, uint8_t* d_out, int32_t order, void* stream);
/* Format by extension: .png (8-bit RGB, zlib deflate) or .ppm (P6). */
int  rtx_save_image(const char* path, const float* rgb, int32_t width, int32_t height);
/* The same from an image that already is 8 bits per channel, RGB order (e.g. rtx_render_u8's output). */
int  rtx_save_image_u8(const char* path, const uint8_t* rgb8, int32_t width, int32_t height);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H_ */
