// Device layer of librtx: scene upload, regular-grid build on the GPU, and the per-pixel trace loop
// (src/main.cpp:126-133 of the reference tree and everything it calls) for gfx950.  One translation unit:
//   rtx_traverse.h        one ray walked and tested by one lane (3D-DDA, sphere / triangle tests)
//   rtx_wave.h            the same search shared by the 64 lanes of a wave
//   rtx_frame_state.h     what a kernel knows about a frame, work cursors, camera rays, shading terms
//   rtx_kernel_frame.h    default pipeline: one persistent kernel per frame + the screen-tile kernels
//   rtx_kernels_levels.h  level-synchronous pipelines (trace / shade per recursion level, resolve)
//   rtx_kernels_grid.h    grid build kernels
//   rtx_kernels_misc.h    8-bit conversion, probes
//   this file             context, host side of every entry point of rtx.h
// A frame needs no host synchronisation until its end: queues, counters and work lists live in device memory.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <type_traits>
#include <vector>

#include "rtx.h"
#include "rtx_internal.h"
#include "rt_math.h"
#include "rt_xform.h"
#include "rtx_traverse.h"
#include "rtx_wave.h"

using namespace rt;

// ... from which a frame launched with a share of the workgroups (frames in flight) runs the generation pipeline
#define RTX_PIXELS_PER_LANE_FOR_GENERATIONS 10.0
// ... below which a wave carries 32 paths and keeps its other lanes free for the shadow rays of fresh hits from the start
// of the frame (48 paths below three times as many, when the launch has the GPU to itself)
#define RTX_PIXELS_PER_LANE_FOR_EARLY_HELPERS 4.0

// generation pipeline: default start of a frame's endgame, in eighths of a pixel left per lane of the launch
#define RTX_GEN_ENDGAME_EIGHTHS 8

#define RTX_MAX_LEVELS 16          // max_depth <= 15
#define RTX_BLOCK 256

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            rtx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);  \
            return RTX_E_DEVICE;                                                                       \
        }                                                                                              \
    } while (0)

#include "rtx_frame_state.h"
#include "rtx_kernels_levels.h"
#include "rtx_kernel_frame.h"
#include "rtx_kernel_generations.h"
#include "rtx_kernels_misc.h"
#include "rtx_kernels_grid.h"

// ---------------------------------------------------------------------------------------------------
// Scratch of rtx_render_device_noalloc: everything the persistent frame pipeline needs beyond the caller's buffers
// (work cursors and counters, the screen-tile map / classes / list / per-tile path depths, the strips of path terms
// deeper than the levels held in LDS), carved out of one file-scope device array.  Every GPU the library runs on
// holds its own copy until the process ends; host code reaches it only through the address hipGetSymbolAddress
// returns for the current device (kept per context), never through the array's name.
// ---------------------------------------------------------------------------------------------------
#define RTX_NOALLOC_MAX_TILES (1 << 17)            // 8x8 pixel tiles of the whole image: 3840 x 2160 has 129 600
#define RTX_NOALLOC_MAX_BLOCKS 1024                // workgroups of the frame kernel (256 CUs x 4)
#define RTX_NOALLOC_GLOBAL_LEVELS 2                // path levels beyond those in LDS: max_depth <= 6
#define RTX_NOALLOC_MAX_DEPTH 6
#define RTX_NA_ALIGN(x) (((size_t)(x) + 255) & ~(size_t)255)
#define RTX_NA_COUNTERS 0
#define RTX_NA_SYNC (RTX_NA_COUNTERS + RTX_NA_ALIGN(CNT_TOTAL * sizeof(int)))
#define RTX_NA_TILE_MAP (RTX_NA_SYNC + RTX_NA_ALIGN(8 * sizeof(unsigned)))
#define RTX_NA_TILE_CLASS (RTX_NA_TILE_MAP + RTX_NA_ALIGN((RTX_NOALLOC_MAX_TILES + 4) * sizeof(unsigned)))
#define RTX_NA_TILE_LIST (RTX_NA_TILE_CLASS + RTX_NA_ALIGN(RTX_NOALLOC_MAX_TILES))
#define RTX_NA_TILE_COST0 (RTX_NA_TILE_LIST + RTX_NA_ALIGN(RTX_NOALLOC_MAX_TILES * sizeof(int)))
#define RTX_NA_TILE_COST1 (RTX_NA_TILE_COST0 + RTX_NA_ALIGN((RTX_NOALLOC_MAX_TILES + 4) * sizeof(int)))
#define RTX_NA_PATH_STACK (RTX_NA_TILE_COST1 + RTX_NA_ALIGN((RTX_NOALLOC_MAX_TILES + 4) * sizeof(int)))
#define RTX_NOALLOC_SCRATCH_BYTES (RTX_NA_PATH_STACK + (size_t)RTX_NOALLOC_MAX_BLOCKS * RTX_BLOCK * RTX_NOALLOC_GLOBAL_LEVELS * sizeof(float4))
static __device__ __align__(256) unsigned char scratch[RTX_NOALLOC_SCRATCH_BYTES];   // 10.6 MB

// ---------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }      // a context's buffers go with the context (rtx_destroy makes its device current first)
    int reserve(size_t count) {
        if (count <= n && p) return RTX_OK;
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            rtx_set_error("hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
            return RTX_E_NOMEM;
        }
        n = count;
        return RTX_OK;
    }
    // the same, and memory that was allocated by the call is zeroed on `stream`, i.e. in order with the work the caller
    // enqueues there next (a plain hipMemset runs on the default stream, which a non-blocking stream does not wait for)
    int reserve_zeroed(size_t count, hipStream_t stream) {
        if (count <= n && p) return RTX_OK;
        const int rc = reserve(count);
        if (rc) return rc;
        if (hipMemsetAsync(p, 0, n * sizeof(T), stream) != hipSuccess) { rtx_set_error("hipMemsetAsync of a new buffer failed"); return RTX_E_DEVICE; }
        return RTX_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

// Buffers of the persistent frame pipeline beyond the caller's: a context's own allocations, or the device's scratch
struct FrameScratch {
    int* counters = nullptr;
    unsigned long long* work = nullptr;            // instrumented frames only
    float4* path_stack = nullptr;
    unsigned* tile_map = nullptr;
    unsigned* prep_sync = nullptr;                 // ticket and done counters of the frame-start kernel (never reset)
    signed char* tile_class = nullptr;
    int* tile_list = nullptr;
    int* tile_cost[2] = {nullptr, nullptr};
};
// What the per-tile path depths in a pair of cost maps belong to
struct HintState {
    int key[6] = {0, 0, 0, 0, 0, 0};   // W, H, tile_rows, rank, world, grid generation the cost map was recorded for
    rtx_camera cam{};                   // ... and the camera it was recorded under
    int cur = 0;                        // tile_cost[cur] was written by the last frame
    int stamp[2] = {0, 0};              // the stamp its entries carry (shifted past the depth bits)
    bool valid = false;
};
// Where the never-reset words of a set of frame-start buffers stand: the stamp of the last frame that used the tile
// map and the cost maps, and the values of the frame-start kernel's ticket and done counters.
struct PrepState {
    unsigned seq = 0;
    unsigned ticket[3] = {0, 0, 0};
    unsigned done[2] = {0, 0};
};
#define RTX_MAX_STAMP ((1u << (31 - RTX_COST_BITS)) - 2u)

struct rtx_ctx {
    int device = 0;
    int n_cus = 256;
    // scene
    bool have_scene = false;
    int n_prims = 0, n_xforms = 0, n_materials = 0, n_lights = 0;
    int n_triangles = 0, n_spheres = 0;
    DevBuf<float4> prims, tri_normal;
    DevBuf<float> xinv, xfwd, mats, lights, geom;
    DevBuf<int> kind, xform;
    // grid
    bool have_ref = false, have_nested = false;
    rtx_grid_info ginfo{};
    DevBuf<float> ref_box, tight_box, bounds;
    DevBuf<int> cstart, cids, fstart, fids, scratch_ids, cell_tmp, scan_tmp, wide_prims;
    DevBuf<unsigned char> cocc;
    DevBuf<unsigned> wmask, cocc_bits;
    DevBuf<int> wpop, wpre, ostart, ocell;
    DevBuf<int4> oinfo;                        // per occupied fine cell: list start, length, class of equal lists
    DevBuf<unsigned long long> list_key, class_key;
    DevBuf<int> class_rep;
    DevBuf<uint2> words;
    DevBuf<float4> frecs;
    DevGrid dgrid{};
    DevSparse dsparse{};
    // frame workspace
    size_t ws_pixels = 0;
    int ws_levels = 0;
    DevBuf<int> counters;
    DevBuf<unsigned long long> work;
    DevBuf<float4> hP, hN, hD, rO[2], rD[2], hL[RTX_MAX_LEVELS];
    DevBuf<int> pix0, hChild[RTX_MAX_LEVELS];
    DevBuf<float4> path_stack;
    DevBuf<float4> gen_stacks;    // generation pipeline: the request stacks and path slots of every wave of the launch
    DevBuf<unsigned> tile_map, prep_sync;
    PrepState prep;               // of the context's own frame-start buffers
    DevBuf<signed char> tile_class;
    DevBuf<int> tile_list, tile_cost[2];
    HintState hint;               // of the context's own cost maps
    HintState scratch_hint;       // of the device scratch's cost maps, while this context was its last user
    unsigned char* scratch = nullptr;   // device address of this GPU's copy of the scratch array
    int grid_generation = 0;
    float fine_pad = 0.0f;        // margin the nested lists were binned with
    DevBuf<float> rgb_local;
    DevBuf<int> ids_local;
    DevBuf<unsigned char> rgb8_local;
    hipEvent_t ev[4 + 3 * 2 * RTX_MAX_LEVELS];
    bool have_events = false;
    // Set by a wave of the frame kernel that ran into the round limit: one int of pinned host memory the kernel
    // writes through its device address, so that an abandoned frame is noticed without a copy per frame.
    int* abort_host = nullptr;
    int* abort_dev = nullptr;
    bool frame_pending = false;   // a frame kernel was enqueued and its status has not been looked at yet
    bool frame_timed = false;     // the events of the context bracket a frame (rtx_frame_times)
    size_t wave_log_waves = 0;    // waves logged by the last instrumented frame
};

namespace {

// Which context used a device's scratch array last (its cost maps then still hold that context's per-tile depths).
#define RTX_MAX_DEVICES 64
std::mutex g_scratch_mutex;
const rtx_ctx* g_scratch_user[RTX_MAX_DEVICES] = {};
PrepState g_scratch_prep[RTX_MAX_DEVICES];      // of each device's scratch array (calls on one device do not overlap)
// Makes `c` the scratch's user on its device; true if it already was.
bool scratch_take_over(const rtx_ctx* c) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    const int d = c->device;
    if (d < 0 || d >= RTX_MAX_DEVICES) return false;
    const bool mine = g_scratch_user[d] == c;
    g_scratch_user[d] = c;
    return mine;
}
void scratch_forget(const rtx_ctx* c) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    for (auto& u : g_scratch_user)
        if (u == c) u = nullptr;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

DevScene dev_scene(const rtx_ctx* c) {
    DevScene s;
    s.prims = c->prims.p;
    s.tri_normal = c->tri_normal.p;
    s.xinv = c->xinv.p;
    s.mats = c->mats.p;
    s.lights = c->lights.p;
    s.n_prims = c->n_prims;
    s.n_lights = c->n_lights;
    return s;
}

int check_params(const rtx_render_params* p, int* tile_rows, int* rank, int* world) {
    if (!p || p->width <= 0 || p->height <= 0) { rtx_set_error("render params: width/height must be positive"); return RTX_E_INVALID; }
    if ((int64_t)p->width * p->height > (int64_t)1 << 28) { rtx_set_error("render params: more than 2^28 pixels"); return RTX_E_LIMIT; }
    if (p->max_depth < 0 || p->max_depth >= RTX_MAX_LEVELS) { rtx_set_error("render params: max_depth must be in [0, %d]", RTX_MAX_LEVELS - 1); return RTX_E_LIMIT; }
    *tile_rows = p->tile_rows > 0 ? p->tile_rows : 8;
    *world = p->world > 0 ? p->world : 1;
    *rank = p->rank;
    if (*rank < 0 || *rank >= *world) { rtx_set_error("render params: rank %d outside world %d", *rank, *world); return RTX_E_INVALID; }
    if (p->search != RTX_SEARCH_REFERENCE_LISTS && p->search != RTX_SEARCH_NESTED) { rtx_set_error("render params: unknown search mode %d", p->search); return RTX_E_INVALID; }
    return RTX_OK;
}

int count_local_rows(int H, int tile_rows, int rank, int world) {
    const int n_tiles = (H + tile_rows - 1) / tile_rows;
    int rows = 0;
    for (int t = rank; t < n_tiles; t += world) rows += std::min(tile_rows, H - t * tile_rows);
    return rows;
}

// Camera::updateAxes (src/geometries.h:332-337) and the per-frame constants of getRayAt (:342-346)
DevCamera make_camera(const rtx_camera* cam, int W, int H) {
    DevCamera c;
    const f3 eye = mk(cam->eye[0], cam->eye[1], cam->eye[2]);
    const f3 center = mk(cam->center[0], cam->center[1], cam->center[2]);
    const f3 up = mk(cam->up[0], cam->up[1], cam->up[2]);
    const f3 dir = center - eye;
    c.eye = eye;
    c.w = normalize3(-dir);
    c.u = normalize3(cross3(up, c.w));
    c.v = normalize3(cross3(c.w, c.u));
    const float pi = 3.14159265358979323846264338327950288f;
    const float aspect = (float)W / (float)H;
    c.tany = std::tan(cam->fov_deg * pi / 360.f);
    c.tanx = c.tany * aspect;
    c.halfW = (float)W / 2.f;
    c.halfH = (float)H / 2.f;
    return c;
}

// Conservative pixel window [x0, x1] x [y0, y1] outside which no camera ray can meet the box [lo, hi]: the
// bounding rectangle of the box's eight projected corners, grown by two pixels.  Any corner at or behind the eye
// plane (or an empty scene, whose box is inverted) gives the whole frame / an empty window.  Pixel (x, y) looks
// along a*u + b*v - w with a = tanx*((x + .5 - W/2)/(W/2)), b = tany*((H/2 - y - .5)/(H/2)) (src/geometries.h:339-351).
void screen_window(const DevCamera& cam, f3 lo, f3 hi, int W, int H, bool have_prims, int* x0, int* x1, int* y0, int* y1) {
    *x0 = 0; *x1 = W - 1; *y0 = 0; *y1 = H - 1;
    if (!have_prims) { *x1 = -1; *y1 = -1; return; }          // nothing can be hit: the whole frame is the miss colour
    if (!(lo.x <= hi.x && lo.y <= hi.y && lo.z <= hi.z)) return;
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (int k = 0; k < 8; k++) {
        const double px = (k & 1) ? hi.x : lo.x, py = (k & 2) ? hi.y : lo.y, pz = (k & 4) ? hi.z : lo.z;
        const double dx = px - cam.eye.x, dy = py - cam.eye.y, dz = pz - cam.eye.z;
        const double depth = -(dx * cam.w.x + dy * cam.w.y + dz * cam.w.z);
        const double extent = std::fabs(dx) + std::fabs(dy) + std::fabs(dz);
        if (!(depth > 1e-3 * extent) || !(depth > 0)) return;   // corner beside or behind the eye: trace everything
        const double a = (dx * cam.u.x + dy * cam.u.y + dz * cam.u.z) / depth;
        const double b = (dx * cam.v.x + dy * cam.v.y + dz * cam.v.z) / depth;
        const double sx = a / cam.tanx * cam.halfW + cam.halfW - 0.5;
        const double sy = cam.halfH - 0.5 - b / cam.tany * cam.halfH;
        if (!(sx == sx) || !(sy == sy)) return;
        xmin = std::min(xmin, sx); xmax = std::max(xmax, sx);
        ymin = std::min(ymin, sy); ymax = std::max(ymax, sy);
    }
    const double margin = 2.0;
    *x0 = (int)std::max(0.0, std::min((double)W, std::floor(xmin - margin)));
    *y0 = (int)std::max(0.0, std::min((double)H, std::floor(ymin - margin)));
    *x1 = (int)std::min((double)(W - 1), std::max(-1.0, std::ceil(xmax + margin)));
    *y1 = (int)std::min((double)(H - 1), std::max(-1.0, std::ceil(ymax + margin)));
}

// Rounds after which a wave of the frame kernel gives the frame up when the caller names no limit.  Two bounds, the
// smaller one counts:
//   * what no valid frame can exceed -- every round of a wave moves at least one of its unfinished rays one step along
//     its walk (or tests the candidates of the cell it stands on, after which the next round moves it), a ray crosses at
//     most rx + ry + rz coarse cells and, inside each, 3 * sub fine cells, each crossing at most two rounds (look, leave),
//     and a wave can hold every ray of the frame in the worst case;
//   * what a wave that works ends well below, so that a kernel that would never end is cut off in seconds and not in
//     hours: 2^24 rounds, or sixteen rounds for every ray of an even share of the frame's pixels where that is more (a
//     wave advances 64 rays per round and a ray takes six to ten rounds, so a wave that drew four times its share is
//     still a factor of six below it; a 4K frame takes a few hundred rounds per wave).
long long derived_round_limit(const rtx_ctx* c, const DevFrame& f, int max_depth, int blocks) {
    const double rays_per_path = (double)(max_depth + 1) * (double)(c->n_lights + 1);
    const DevGrid& g = c->dgrid;
    const double steps_per_ray = 2.0 * (double)(g.rx + g.ry + g.rz + 3) * (double)(3 * std::max(g.sub, 1) + 2) + 8.0;
    const double bound = ((double)f.n_items0 + 64.0) * rays_per_path * steps_per_ray;
    const double waves = (double)std::max(blocks, 1) * (RTX_BLOCK / 64);
    const double guard = std::max(16777216.0, 16.0 * ((double)f.n_items0 / waves + 64.0) * rays_per_path);
    return (long long)std::min(std::min(bound, guard), 4.0e18);
}

// Looks at the status word of the context's last frame; `wait` synchronises with the frame's end first.
int take_frame_status(rtx_ctx* c, bool wait) {
    if (!c->frame_pending) return RTX_OK;
    if (wait) {
        if (hipEventSynchronize(c->ev[2]) != hipSuccess) { rtx_set_error("waiting for the last frame failed"); return RTX_E_DEVICE; }
    } else if (hipEventQuery(c->ev[2]) != hipSuccess) {
        return RTX_OK;            // still running: whoever looks next finds it
    }
    c->frame_pending = false;
    if (*(volatile int*)c->abort_host != 0) {
        *c->abort_host = 0;
        rtx_set_error("frame abandoned: a wave of the frame kernel reached the round limit; the image is incomplete");
        return RTX_E_LIMIT;
    }
    return RTX_OK;
}

int ensure_workspace(rtx_ctx* c, size_t pixels, int levels) {
    if (pixels <= c->ws_pixels && levels <= c->ws_levels) return RTX_OK;
    pixels = std::max(pixels, c->ws_pixels);
    levels = std::max(levels, c->ws_levels);
    int rc;
    if ((rc = c->counters.reserve(CNT_TOTAL))) return rc;
    if ((rc = c->work.reserve(24))) return rc;
    if ((rc = c->hP.reserve(pixels)) || (rc = c->hN.reserve(pixels)) || (rc = c->hD.reserve(pixels))) return rc;
    if ((rc = c->pix0.reserve(pixels))) return rc;
    for (int k = 0; k < 2; k++)
        if ((rc = c->rO[k].reserve(pixels)) || (rc = c->rD[k].reserve(pixels))) return rc;
    for (int l = 0; l < levels; l++)
        if ((rc = c->hL[l].reserve(pixels)) || (rc = c->hChild[l].reserve(pixels))) return rc;
    c->ws_pixels = pixels;
    c->ws_levels = levels;
    return RTX_OK;
}

template <int MODE, bool COUNT>
void launch_level(rtx_ctx* c, const DevScene& sc, const DevFrame& f, int level, int blocks, hipStream_t st, bool timed,
                  hipEvent_t* ev) {
    if (timed) (void)hipEventRecord(ev[0], st);
    hipLaunchKernelGGL((k_trace<MODE, COUNT>), dim3(blocks), dim3(RTX_BLOCK), 0, st, sc, c->dgrid, f, level);
    if (timed) (void)hipEventRecord(ev[1], st);
    hipLaunchKernelGGL((k_shade<MODE, COUNT>), dim3(blocks), dim3(RTX_BLOCK), 0, st, sc, c->dgrid, f, level);
    if (timed) (void)hipEventRecord(ev[2], st);
}

template <int MODE, bool SINGLE_XF, bool COUNT>
void launch_level_w(rtx_ctx* c, const DevScene& sc, const DevFrame& f, int level, int blocks, hipStream_t st, bool timed,
                    hipEvent_t* ev) {
    if (timed) (void)hipEventRecord(ev[0], st);
    hipLaunchKernelGGL((k_trace_w<MODE, SINGLE_XF, COUNT>), dim3(blocks), dim3(RTX_BLOCK), 0, st, sc, c->dgrid, c->dsparse, f, level);
    if (timed) (void)hipEventRecord(ev[1], st);
    hipLaunchKernelGGL((k_shade_w<MODE, SINGLE_XF, COUNT>), dim3(blocks), dim3(RTX_BLOCK), 0, st, sc, c->dgrid, c->dsparse, f, level);
    if (timed) (void)hipEventRecord(ev[2], st);
}

// gen: the generation pipeline (k_frame_gen; `ps` = its request stacks, `levels` = path terms a request carries)
template <int MODE, bool SINGLE_XF, int PRIMS, int WAVES>
void launch_frame4(rtx_ctx* c, const DevScene& sc, const DevFrame& f, float4* ps, int blocks, hipStream_t st, bool count, int levels, bool gen) {
    if (gen) {
        const int gen_blocks = blocks * (RTX_BLOCK / RTX_GEN_BLOCK);
        if (count) hipLaunchKernelGGL((k_frame_gen<MODE, SINGLE_XF, true, PRIMS, WAVES>), dim3(gen_blocks), dim3(RTX_GEN_BLOCK), 0, st, sc, c->dgrid, c->dsparse, f, ps, levels);
        else hipLaunchKernelGGL((k_frame_gen<MODE, SINGLE_XF, false, PRIMS, WAVES>), dim3(gen_blocks), dim3(RTX_GEN_BLOCK), 0, st, sc, c->dgrid, c->dsparse, f, ps, levels);
        return;
    }
    if (count) hipLaunchKernelGGL((k_frame<MODE, SINGLE_XF, true, PRIMS, WAVES>), dim3(blocks), dim3(RTX_BLOCK), 0, st, sc, c->dgrid, c->dsparse, f, ps, levels);
    else hipLaunchKernelGGL((k_frame<MODE, SINGLE_XF, false, PRIMS, WAVES>), dim3(blocks), dim3(RTX_BLOCK), 0, st, sc, c->dgrid, c->dsparse, f, ps, levels);
}

template <int MODE, bool SINGLE_XF, int PRIMS>
void launch_frame3(rtx_ctx* c, const DevScene& sc, const DevFrame& f, float4* ps, int blocks, hipStream_t st, bool count, int levels, int waves, bool gen) {
    if (waves >= 4) launch_frame4<MODE, SINGLE_XF, PRIMS, 4>(c, sc, f, ps, blocks, st, count, levels, gen);
    else launch_frame4<MODE, SINGLE_XF, PRIMS, 3>(c, sc, f, ps, blocks, st, count, levels, gen);
}

// (`levels` = entries per lane of the strip `ps` of path terms kept in global memory; both builds index it from their
// own first global level, so the strip sized for the four-wave build also serves the three-wave one)
// picks the kernel specialisation for the scene: one transform or many, triangles only / spheres only / both
template <int MODE>
void launch_frame(rtx_ctx* c, const DevScene& sc, const DevFrame& f, float4* ps, int blocks, hipStream_t st, bool count, int levels, int waves, bool gen) {
    const bool single = c->n_xforms == 1;
    const int prims = c->n_spheres == 0 ? 1 : (c->n_triangles == 0 ? 2 : 0);
    if (single) {
        if (prims == 1) launch_frame3<MODE, true, 1>(c, sc, f, ps, blocks, st, count, levels, waves, gen);
        else if (prims == 2) launch_frame3<MODE, true, 2>(c, sc, f, ps, blocks, st, count, levels, waves, gen);
        else launch_frame3<MODE, true, 0>(c, sc, f, ps, blocks, st, count, levels, waves, gen);
    } else {
        if (prims == 1) launch_frame3<MODE, false, 1>(c, sc, f, ps, blocks, st, count, levels, waves, gen);
        else if (prims == 2) launch_frame3<MODE, false, 2>(c, sc, f, ps, blocks, st, count, levels, waves, gen);
        else launch_frame3<MODE, false, 0>(c, sc, f, ps, blocks, st, count, levels, waves, gen);
    }
}

template <int MODE>
void launch_level_any(rtx_ctx* c, const DevScene& sc, const DevFrame& f, int level, int blocks, hipStream_t st, bool timed,
                      hipEvent_t* ev, bool count, bool serial) {
    if (serial) {
        if (count) launch_level<MODE, true>(c, sc, f, level, blocks, st, timed, ev);
        else launch_level<MODE, false>(c, sc, f, level, blocks, st, timed, ev);
        return;
    }
    const bool single = c->n_xforms == 1;
    if (single) {
        if (count) launch_level_w<MODE, true, true>(c, sc, f, level, blocks, st, timed, ev);
        else launch_level_w<MODE, true, false>(c, sc, f, level, blocks, st, timed, ev);
    } else {
        if (count) launch_level_w<MODE, false, true>(c, sc, f, level, blocks, st, timed, ev);
        else launch_level_w<MODE, false, false>(c, sc, f, level, blocks, st, timed, ev);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
extern "C" {

int rtx_create(rtx_ctx** out, int device_id) {
    if (!out) { rtx_set_error("rtx_create: null out pointer"); return RTX_E_INVALID; }
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        rtx_set_error("no usable HIP device (%s); this library has no CPU path", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return RTX_E_DEVICE;
    }
    if (device_id < 0 || device_id >= n_dev) { rtx_set_error("rtx_create: device %d out of range (%d devices)", device_id, n_dev); return RTX_E_INVALID; }
    rtx_ctx* c = new (std::nothrow) rtx_ctx;
    if (!c) return RTX_E_NOMEM;
    c->device = device_id;
    DeviceGuard guard(device_id);
    if (!guard.ok) { delete c; rtx_set_error("hipSetDevice(%d) failed", device_id); return RTX_E_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount;
    for (auto& ev : c->ev) {
        if (hipEventCreate(&ev) != hipSuccess) { delete c; rtx_set_error("hipEventCreate failed"); return RTX_E_DEVICE; }
    }
    c->have_events = true;
    if (hipHostMalloc((void**)&c->abort_host, sizeof(int), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->abort_dev, c->abort_host, 0) != hipSuccess) {
        rtx_set_error("rtx_create: pinned status word could not be allocated");
        rtx_destroy(c);
        return RTX_E_NOMEM;
    }
    *c->abort_host = 0;
    // This GPU's copy of the scratch array of rtx_render_device_noalloc: its address is a property of the device, asked
    // for once per context with that device current.  The array's name itself is never used as an address on the host.
    {
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            rtx_set_error("rtx_create: hipGetSymbolAddress(scratch) failed: %s", hipGetErrorString(e));
            rtx_destroy(c);
            return RTX_E_DEVICE;
        }
        c->scratch = (unsigned char*)ptr;
    }
    *out = c;
    return RTX_OK;
}

void rtx_destroy(rtx_ctx* c) {
    if (!c) return;
    DeviceGuard guard(c->device);
    (void)hipDeviceSynchronize();
    if (c->have_events)
        for (auto& ev : c->ev) (void)hipEventDestroy(ev);
    if (c->abort_host) (void)hipHostFree(c->abort_host);
    scratch_forget(c);
    delete c;                     // every DevBuf releases its memory
}

int rtx_set_scene(rtx_ctx* c, const rtx_scene_view* v) {
    if (!c || !v) { rtx_set_error("rtx_set_scene: null argument"); return RTX_E_INVALID; }
    if (v->n_prims < 0 || v->n_xforms < 0 || v->n_materials < 0 || v->n_lights < 0) { rtx_set_error("rtx_set_scene: negative count"); return RTX_E_INVALID; }
    if (v->n_prims > 0 && (!v->prim_kind || !v->prim_geom || !v->prim_xform || !v->prim_material || !v->xforms || !v->materials)) {
        rtx_set_error("rtx_set_scene: null array with n_prims > 0");
        return RTX_E_INVALID;
    }
    if (v->n_lights > 0 && !v->lights) { rtx_set_error("rtx_set_scene: null lights"); return RTX_E_INVALID; }
    const int n = v->n_prims;
    for (int i = 0; i < n; i++) {
        if (v->prim_kind[i] != RTX_PRIM_TRIANGLE && v->prim_kind[i] != RTX_PRIM_SPHERE) { rtx_set_error("primitive %d: unknown kind %d", i, v->prim_kind[i]); return RTX_E_INVALID; }
        if (v->prim_xform[i] < 0 || v->prim_xform[i] >= v->n_xforms) { rtx_set_error("primitive %d: transform index %d out of range", i, v->prim_xform[i]); return RTX_E_INVALID; }
        if (v->prim_material[i] < 0 || v->prim_material[i] >= v->n_materials) { rtx_set_error("primitive %d: material index %d out of range", i, v->prim_material[i]); return RTX_E_INVALID; }
    }
    for (int l = 0; l < v->n_lights; l++) {
        const int k = (int)v->lights[10 * l];
        if (k != RTX_LIGHT_POINT && k != RTX_LIGHT_DIRECTIONAL) { rtx_set_error("light %d: unknown kind %d", l, k); return RTX_E_INVALID; }
    }
    DeviceGuard guard(c->device);
    if (!guard.ok) { rtx_set_error("hipSetDevice(%d) failed", c->device); return RTX_E_DEVICE; }

    std::vector<float4> prims, tn;
    std::vector<float> xinv, mats, lights;
    try {
        prims.resize(3 * (size_t)std::max(n, 1));
        tn.resize((size_t)std::max(n, 1));
        xinv.resize(16 * (size_t)std::max(v->n_xforms, 1));
        mats.assign(16 * (size_t)std::max(v->n_materials, 1), 0.0f);
        lights.assign(12 * (size_t)std::max(v->n_lights, 1), 0.0f);
    } catch (const std::bad_alloc&) {
        return RTX_E_NOMEM;
    }
    for (int i = 0; i < n; i++) {
        const float* gp = v->prim_geom + 9 * (size_t)i;
        const float xf = as_float(v->prim_xform[i]), mt = as_float(v->prim_material[i]), kd = as_float(v->prim_kind[i]);
        if (v->prim_kind[i] == RTX_PRIM_TRIANGLE) {
            const f3 A = mk(gp[0], gp[1], gp[2]), B = mk(gp[3], gp[4], gp[5]), C = mk(gp[6], gp[7], gp[8]);
            const f3 e1 = A - B, e2 = A - C;                       // src/geometries.h:306
            const f3 nn = normalize3(cross3(B - A, C - A));        // src/geometries.h:308
            prims[3 * (size_t)i] = make_float4(A.x, A.y, A.z, e1.x);
            prims[3 * (size_t)i + 1] = make_float4(e1.y, e1.z, e2.x, e2.y);
            prims[3 * (size_t)i + 2] = make_float4(e2.z, xf, mt, kd);
            tn[i] = make_float4(nn.x, nn.y, nn.z, 0.0f);
        } else {
            prims[3 * (size_t)i] = make_float4(gp[0], gp[1], gp[2], gp[3]);
            prims[3 * (size_t)i + 1] = make_float4(0, 0, 0, 0);
            prims[3 * (size_t)i + 2] = make_float4(0, xf, mt, kd);
            tn[i] = make_float4(0, 0, 0, 0);
        }
    }
    for (int x = 0; x < v->n_xforms; x++) {
        Mat4 m;
        memcpy(m.m, v->xforms + 16 * (size_t)x, sizeof m.m);
        const Mat4 inv = mat_inverse(m);
        memcpy(&xinv[16 * (size_t)x], inv.m, sizeof inv.m);
    }
    for (int m = 0; m < v->n_materials; m++) memcpy(&mats[16 * (size_t)m], v->materials + 13 * (size_t)m, 13 * sizeof(float));
    for (int l = 0; l < v->n_lights; l++) memcpy(&lights[12 * (size_t)l], v->lights + 10 * (size_t)l, 10 * sizeof(float));

    int rc;
    if ((rc = c->prims.reserve(prims.size())) || (rc = c->tri_normal.reserve(tn.size())) || (rc = c->xinv.reserve(xinv.size())) ||
        (rc = c->xfwd.reserve(16 * (size_t)std::max(v->n_xforms, 1))) || (rc = c->mats.reserve(mats.size())) ||
        (rc = c->lights.reserve(lights.size())) || (rc = c->geom.reserve(9 * (size_t)std::max(n, 1))) ||
        (rc = c->kind.reserve((size_t)std::max(n, 1))) || (rc = c->xform.reserve((size_t)std::max(n, 1))))
        return rc;
    HIP_TRY(hipMemcpy(c->prims.p, prims.data(), prims.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->tri_normal.p, tn.data(), tn.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->xinv.p, xinv.data(), xinv.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->mats.p, mats.data(), mats.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->lights.p, lights.data(), lights.size() * sizeof(float), hipMemcpyHostToDevice));
    if (n > 0) {
        HIP_TRY(hipMemcpy(c->geom.p, v->prim_geom, 9 * (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->kind.p, v->prim_kind, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->xform.p, v->prim_xform, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    }
    if (v->n_xforms > 0) HIP_TRY(hipMemcpy(c->xfwd.p, v->xforms, 16 * (size_t)v->n_xforms * sizeof(float), hipMemcpyHostToDevice));
    c->n_prims = n;
    c->n_triangles = 0;
    for (int i = 0; i < n; i++) c->n_triangles += v->prim_kind[i] == RTX_PRIM_TRIANGLE;
    c->n_spheres = n - c->n_triangles;
    c->n_xforms = v->n_xforms;
    c->n_materials = v->n_materials;
    c->n_lights = v->n_lights;
    c->have_scene = true;
    c->have_ref = c->have_nested = false;
    return RTX_OK;
}

// exclusive prefix sum of n counts into start[0..n] (start[n] = total), on the default stream
static int device_scan(rtx_ctx* c, int n, const int* count, int* start) {
    const int n_strips = (n + RTX_SCAN_STRIP - 1) / RTX_SCAN_STRIP;
    int rc;
    if ((rc = c->scan_tmp.reserve((size_t)std::max(n_strips, 1)))) return rc;
    if (n_strips > 0) hipLaunchKernelGGL(k_scan_strips, dim3(n_strips), dim3(1024), 0, 0, n, count, start, c->scan_tmp.p);
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, 0, n_strips, c->scan_tmp.p, n, start);
    if (n_strips > 1) hipLaunchKernelGGL(k_scan_offsets, dim3(n_strips), dim3(1024), 0, 0, n, (const int*)c->scan_tmp.p, start);
    return RTX_OK;
}

// count -> scan -> scatter -> order, for one set of boxes and one resolution
static int build_lists(rtx_ctx* c, const BinParams& bp, const float* boxes, DevBuf<int>& start, DevBuf<int>& ids, int64_t* n_refs) {
    const int n = c->n_prims;
    const int64_t n_cells64 = bp.sub > 0 ? (int64_t)(bp.rx / bp.sub) * (bp.ry / bp.sub) * (bp.rz / bp.sub) * bp.cpb
                                         : (int64_t)bp.rx * bp.ry * bp.rz;
    if (n_cells64 > ((int64_t)1 << 27)) { rtx_set_error("grid of %lld cells exceeds the 2^27 cell limit", (long long)n_cells64); return RTX_E_LIMIT; }
    const int n_cells = (int)n_cells64;
    int rc;
    if ((rc = start.reserve((size_t)n_cells + 1)) || (rc = c->cell_tmp.reserve((size_t)n_cells + 1))) return rc;
    HIP_TRY(hipMemsetAsync(c->cell_tmp.p, 0, ((size_t)n_cells + 1) * sizeof(int), 0));
    const int pb = (n + 255) / 256;
    const bool exact = bp.sub > 0;   // nested grid: triangles are binned by their own extent, not their box
    const float* eg = exact ? c->geom.p : nullptr;
    // primitives spanning many cells are set aside by each pass and binned by a workgroup each: [list, count] twice
    if ((rc = c->wide_prims.reserve(2 * ((size_t)n + 1)))) return rc;
    int* wide[2] = {c->wide_prims.p, c->wide_prims.p + n + 1};
    HIP_TRY(hipMemsetAsync(wide[0] + n, 0, sizeof(int), 0));
    HIP_TRY(hipMemsetAsync(wide[1] + n, 0, sizeof(int), 0));
    const int wide_blocks = std::max(1, std::min(n, c->n_cus * 8));
    if (n > 0) {
        hipLaunchKernelGGL((k_bin<false>), dim3(pb), dim3(256), 0, 0, n, bp, boxes, c->cell_tmp.p, (const int*)nullptr, (int*)nullptr, eg,
                           (const int*)c->kind.p, (const int*)c->xform.p, (const float*)c->xfwd.p, wide[0], wide[0] + n);
        hipLaunchKernelGGL((k_bin_wide<false>), dim3(wide_blocks), dim3(256), 0, 0, (const int*)wide[0], (const int*)(wide[0] + n), bp, boxes,
                           c->cell_tmp.p, (const int*)nullptr, (int*)nullptr, eg, (const int*)c->kind.p, (const int*)c->xform.p,
                           (const float*)c->xfwd.p);
    }
    if ((rc = device_scan(c, n_cells, c->cell_tmp.p, start.p))) return rc;
    int total = 0;
    HIP_TRY(hipMemcpy(&total, start.p + n_cells, sizeof(int), hipMemcpyDeviceToHost));
    if (total < 0) { rtx_set_error("cell reference count overflowed 2^31"); return RTX_E_LIMIT; }
    if ((rc = ids.reserve((size_t)std::max(total, 1))) || (rc = c->scratch_ids.reserve((size_t)std::max(total, 1)))) return rc;
    HIP_TRY(hipMemsetAsync(c->cell_tmp.p, 0, ((size_t)n_cells + 1) * sizeof(int), 0));
    if (n > 0) {
        hipLaunchKernelGGL((k_bin<true>), dim3(pb), dim3(256), 0, 0, n, bp, boxes, c->cell_tmp.p, (const int*)start.p, ids.p, eg,
                           (const int*)c->kind.p, (const int*)c->xform.p, (const float*)c->xfwd.p, wide[1], wide[1] + n);
        hipLaunchKernelGGL((k_bin_wide<true>), dim3(wide_blocks), dim3(256), 0, 0, (const int*)wide[1], (const int*)(wide[1] + n), bp, boxes,
                           c->cell_tmp.p, (const int*)start.p, ids.p, eg, (const int*)c->kind.p, (const int*)c->xform.p,
                           (const float*)c->xfwd.p);
    }
    // the fill cursors are done with: the same array now collects the cells whose lists are too long for one lane
    int* long_cells = c->cell_tmp.p;
    int* n_long = c->cell_tmp.p + n_cells;
    HIP_TRY(hipMemsetAsync(n_long, 0, sizeof(int), 0));
    hipLaunchKernelGGL(k_sort_small, dim3((n_cells + 255) / 256), dim3(256), 0, 0, n_cells, (const int*)start.p, ids.p, 24, long_cells, n_long);
    const int sort_blocks = std::min(n_cells, c->n_cus * 8);
    hipLaunchKernelGGL(k_sort_lds, dim3(sort_blocks), dim3(256), 0, 0, (const int*)long_cells, (const int*)n_long, (const int*)start.p, ids.p);
    hipLaunchKernelGGL(k_sort_large, dim3(sort_blocks), dim3(256), 0, 0, (const int*)long_cells, (const int*)n_long, (const int*)start.p, ids.p,
                       c->scratch_ids.p);
    HIP_TRY(hipGetLastError());
    *n_refs = total;
    return RTX_OK;
}

int rtx_build_grid(rtx_ctx* c, int32_t rx, int32_t ry, int32_t rz, int32_t sub, uint32_t flags) {
    if (!c) { rtx_set_error("rtx_build_grid: null context"); return RTX_E_INVALID; }
    if (!c->have_scene) { rtx_set_error("rtx_build_grid: no scene set"); return RTX_E_STATE; }
    if (rx < 1 || ry < 1 || rz < 1 || rx > 1024 || ry > 1024 || rz > 1024) { rtx_set_error("rtx_build_grid: resolution must be in [1, 1024] per axis"); return RTX_E_INVALID; }
    if (sub < 0 || sub > 64) { rtx_set_error("rtx_build_grid: sub must be in [0, 64]"); return RTX_E_INVALID; }
    if (!(flags & (RTX_GRID_REFERENCE | RTX_GRID_NESTED))) { rtx_set_error("rtx_build_grid: no grid kind requested"); return RTX_E_INVALID; }
    DeviceGuard guard(c->device);
    if (!guard.ok) { rtx_set_error("hipSetDevice(%d) failed", c->device); return RTX_E_DEVICE; }
    const int n = c->n_prims;
    int rc;
    if ((rc = c->ref_box.reserve(6 * (size_t)std::max(n, 1))) || (rc = c->tight_box.reserve(6 * (size_t)std::max(n, 1))) || (rc = c->bounds.reserve(6))) return rc;

    HIP_TRY(hipEventRecord(c->ev[0], 0));
    if (n > 0)
        hipLaunchKernelGGL(k_boxes, dim3((n + 255) / 256), dim3(256), 0, 0, n, (const float*)c->geom.p, (const int*)c->kind.p,
                           (const int*)c->xform.p, (const float*)c->xfwd.p, c->ref_box.p, c->tight_box.p);
    hipLaunchKernelGGL(k_scene_bounds, dim3(1), dim3(1024), 0, 0, n, (const float*)c->ref_box.p, c->bounds.p);
    float b[6];
    HIP_TRY(hipMemcpy(b, c->bounds.p, sizeof b, hipMemcpyDeviceToHost));

    rtx_grid_info& gi = c->ginfo;
    memset(&gi, 0, sizeof gi);
    for (int a = 0; a < 3; a++) { gi.lo[a] = b[a]; gi.hi[a] = b[3 + a]; }
    gi.rx = rx; gi.ry = ry; gi.rz = rz;
    gi.n_cells = (int64_t)rx * ry * rz;

    DevGrid& g = c->dgrid;
    memset(&g, 0, sizeof g);
    g.lo = mk(b[0], b[1], b[2]);
    g.hi = mk(b[3], b[4], b[5]);
    g.size = g.hi - g.lo;                                           // src/grid.h:41
    g.res = mk((float)rx, (float)ry, (float)rz);
    g.rx = rx; g.ry = ry; g.rz = rz;

    c->have_ref = c->have_nested = false;
    c->grid_generation++;
    if (flags & RTX_GRID_REFERENCE) {
        BinParams bp;
        bp.lo = g.lo; bp.size = g.size; bp.res = g.res;
        bp.rx = rx; bp.ry = ry; bp.rz = rz;
        bp.pad = 0.0f;
        bp.sub = 0; bp.cpb = 0;
        if ((rc = build_lists(c, bp, c->ref_box.p, c->cstart, c->cids, &gi.n_refs))) return rc;
        g.cstart = c->cstart.p;
        g.cids = c->cids.p;
        c->have_ref = true;
    }
    if (flags & RTX_GRID_NESTED) {
        if (sub == 0) {
            // Aim at about one primitive per occupied fine cell: count the references the tight boxes make at the
            // coarse resolution, take the mean per occupied coarse cell and split each axis by its cube root.
            // (A count by primitives alone misjudges scenes of few large primitives, whose cells are crowded by
            // the same walls over and over.)
            BinParams cp;
            cp.lo = g.lo; cp.size = g.size; cp.res = g.res;
            cp.rx = rx; cp.ry = ry; cp.rz = rz;
            cp.pad = 0.0f;
            cp.sub = 0; cp.cpb = 0;
            const int n_coarse = rx * ry * rz;
            if ((rc = c->cell_tmp.reserve((size_t)n_coarse + 1))) return rc;
            HIP_TRY(hipMemsetAsync(c->cell_tmp.p, 0, ((size_t)n_coarse + 1) * sizeof(int), 0));
            if (n > 0)
                hipLaunchKernelGGL((k_bin<false>), dim3((n + 255) / 256), dim3(256), 0, 0, n, cp, (const float*)c->tight_box.p, c->cell_tmp.p,
                                   (const int*)nullptr, (int*)nullptr, (const float*)nullptr, (const int*)c->kind.p, (const int*)c->xform.p,
                                   (const float*)c->xfwd.p, (int*)nullptr, (int*)nullptr);
            std::vector<int> counts((size_t)n_coarse);
            HIP_TRY(hipMemcpy(counts.data(), c->cell_tmp.p, counts.size() * sizeof(int), hipMemcpyDeviceToHost));
            long long refs = 0, occupied = 0;
            for (int v : counts) { refs += v; occupied += v > 0; }
            const double per_cell = occupied > 0 ? (double)refs / (double)occupied : 1.0;
            sub = (int)std::min(32.0, std::max(1.0, std::floor(std::cbrt(per_cell) + 0.5)));
        }
        const int frx = rx * sub, fry = ry * sub, frz = rz * sub;
        BinParams bp;
        bp.lo = g.lo; bp.size = g.size; bp.res = mk((float)frx, (float)fry, (float)frz);
        bp.rx = frx; bp.ry = fry; bp.rz = frz;
        bp.pad = 2e-4f * std::max(g.size.x, std::max(g.size.y, g.size.z));
        c->fine_pad = bp.pad;
        const int wpb = (sub * sub * sub + 31) / 32;
        bp.sub = sub; bp.cpb = 32 * wpb;
        if ((rc = build_lists(c, bp, c->tight_box.p, c->fstart, c->fids, &gi.n_fine_refs))) return rc;
        const int n_blocks = (int)gi.n_cells, n_words = n_blocks * wpb;
        if ((rc = c->cocc.reserve((size_t)n_blocks)) || (rc = c->wmask.reserve((size_t)n_words)) || (rc = c->wpop.reserve((size_t)n_words)) ||
            (rc = c->wpre.reserve((size_t)n_words + 1)) || (rc = c->words.reserve((size_t)n_words)))
            return rc;
        hipLaunchKernelGGL(k_words, dim3((n_words + 255) / 256), dim3(256), 0, 0, n_words, (const int*)c->fstart.p, c->wmask.p, c->wpop.p);
        if ((rc = device_scan(c, n_words, c->wpop.p, c->wpre.p))) return rc;
        int n_occ = 0;
        HIP_TRY(hipMemcpy(&n_occ, c->wpre.p + n_words, sizeof(int), hipMemcpyDeviceToHost));
        if ((rc = c->ostart.reserve((size_t)n_occ + 1)) || (rc = c->ocell.reserve((size_t)std::max(n_occ, 1)))) return rc;
        hipLaunchKernelGGL(k_compact, dim3((n_words + 255) / 256), dim3(256), 0, 0, n_words, (const int*)c->fstart.p, (const unsigned*)c->wmask.p,
                           (const int*)c->wpre.p, c->words.p, c->ostart.p, c->ocell.p);
        int n_shared = 0;
        {
            // classes of equal lists (see k_list_keys): a table of twice as many slots as occupied cells, a power of two
            int slots = 64;
            while (slots < 2 * n_occ) slots *= 2;
            if ((rc = c->oinfo.reserve((size_t)std::max(n_occ, 1))) || (rc = c->list_key.reserve((size_t)std::max(n_occ, 1))) ||
                (rc = c->class_key.reserve((size_t)slots)) || (rc = c->class_rep.reserve((size_t)slots + 1)))
                return rc;
            HIP_TRY(hipMemsetAsync(c->class_key.p, 0, (size_t)slots * sizeof(unsigned long long), 0));
            HIP_TRY(hipMemsetAsync(c->class_rep.p, 0x7f, (size_t)slots * sizeof(int), 0));
            HIP_TRY(hipMemsetAsync(c->class_rep.p + slots, 0, sizeof(int), 0));
            if (n_occ > 0) {
                hipLaunchKernelGGL(k_list_keys, dim3((n_occ + 255) / 256), dim3(256), 0, 0, n_occ, (const int*)c->ostart.p, (const int*)c->fids.p,
                                   c->list_key.p, c->class_key.p, c->class_rep.p, slots - 1);
                hipLaunchKernelGGL(k_list_classes, dim3((n_occ + 255) / 256), dim3(256), 0, 0, n_occ, (const int*)c->ostart.p, (const int*)c->fids.p,
                                   (const unsigned long long*)c->list_key.p, (const unsigned long long*)c->class_key.p, (const int*)c->class_rep.p,
                                   slots - 1, c->oinfo.p, c->class_rep.p + slots);
                HIP_TRY(hipMemcpy(&n_shared, c->class_rep.p + slots, sizeof(int), hipMemcpyDeviceToHost));
            }
        }
        // (worth the look-up inside the walk where at least a quarter of the occupied cells repeat another cell's list:
        // spheres and large triangles in a coarse nested grid; the dragon's 47 549 lists are all but unique)
        c->dsparse.skip_repeats = c->n_xforms != 1 && 4 * (int64_t)n_shared >= (int64_t)n_occ && n_shared > 0 &&
                                  !(flags & RTX_GRID_TEST_REPEATED_LISTS) ? 1 : 0;
        hipLaunchKernelGGL(k_occupancy, dim3((n_blocks + 255) / 256), dim3(256), 0, 0, n_blocks, wpb, (const unsigned*)c->wmask.p, c->cocc.p);
        if ((rc = c->cocc_bits.reserve((size_t)(n_blocks + 31) / 32))) return rc;
        hipLaunchKernelGGL(k_occupancy_bits, dim3(((n_blocks + 31) / 32 + 255) / 256), dim3(256), 0, 0, n_blocks, (const unsigned char*)c->cocc.p, c->cocc_bits.p);
        g.cocc_bits = c->cocc_bits.p;
        c->dsparse.words = c->words.p;
        c->dsparse.lists = c->dsparse.skip_repeats ? (const int*)c->oinfo.p : (const int*)c->ostart.p;
        c->dsparse.words_per_block = wpb;
        g.sub = sub;
        g.cpb = 32 * wpb;
        g.frx = frx; g.fry = fry; g.frz = frz;
        g.fcell = mk(g.size.x / (float)frx, g.size.y / (float)fry, g.size.z / (float)frz);
        g.inv_fcell = mk((float)frx / g.size.x, (float)fry / g.size.y, (float)frz / g.size.z);
        g.fine_off = mk(-g.lo.x * g.inv_fcell.x, -g.lo.y * g.inv_fcell.y, -g.lo.z * g.inv_fcell.z);
        g.fstart = c->fstart.p;
        g.fids = c->fids.p;
        // (the kernels of a scene with one transform read 3 fields per record, the others 6: see k_expand_records)
        const int frec_stride = c->n_xforms == 1 ? 3 : 6;
        if ((rc = c->frecs.reserve((size_t)frec_stride * (size_t)std::max<int64_t>(gi.n_fine_refs, 1)))) return rc;
        if (gi.n_fine_refs > 0)
            hipLaunchKernelGGL(k_expand_records, dim3((unsigned)((gi.n_fine_refs + 255) / 256)), dim3(256), 0, 0, (int)gi.n_fine_refs,
                               (const int*)c->fids.p, (const float4*)c->prims.p, (const float*)c->xinv.p, frec_stride, c->frecs.p);
        g.frecs = c->frecs.p;
        g.cocc = c->cocc.p;
        gi.sub = sub;
        gi.n_fine_cells = (int64_t)n_blocks * 32 * wpb;
        gi.n_occupied_fine_cells = n_occ;
        c->have_nested = true;
    }
    HIP_TRY(hipEventRecord(c->ev[1], 0));
    HIP_TRY(hipEventSynchronize(c->ev[1]));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
    gi.build_ms = ms;
    gi.skips_repeated_lists = (flags & RTX_GRID_NESTED) ? c->dsparse.skip_repeats : 0;
    return RTX_OK;
}

int rtx_get_grid_info(const rtx_ctx* c, rtx_grid_info* out) {
    if (!c || !out) { rtx_set_error("rtx_get_grid_info: null argument"); return RTX_E_INVALID; }
    if (!c->have_ref && !c->have_nested) { rtx_set_error("rtx_get_grid_info: no grid built"); return RTX_E_STATE; }
    *out = c->ginfo;
    return RTX_OK;
}

int rtx_download_grid(const rtx_ctx* c, uint32_t which, int32_t* cell_start, int32_t* prim_ids) {
    if (!c || !cell_start) { rtx_set_error("rtx_download_grid: null argument"); return RTX_E_INVALID; }
    DeviceGuard guard(c->device);
    if (which == RTX_GRID_REFERENCE) {
        if (!c->have_ref) { rtx_set_error("rtx_download_grid: reference lists not built"); return RTX_E_STATE; }
        HIP_TRY(hipMemcpy(cell_start, c->cstart.p, ((size_t)c->ginfo.n_cells + 1) * sizeof(int), hipMemcpyDeviceToHost));
        if (prim_ids && c->ginfo.n_refs > 0) HIP_TRY(hipMemcpy(prim_ids, c->cids.p, (size_t)c->ginfo.n_refs * sizeof(int), hipMemcpyDeviceToHost));
    } else if (which == RTX_GRID_NESTED) {
        if (!c->have_nested) { rtx_set_error("rtx_download_grid: nested lists not built"); return RTX_E_STATE; }
        HIP_TRY(hipMemcpy(cell_start, c->fstart.p, ((size_t)c->ginfo.n_fine_cells + 1) * sizeof(int), hipMemcpyDeviceToHost));
        if (prim_ids && c->ginfo.n_fine_refs > 0) HIP_TRY(hipMemcpy(prim_ids, c->fids.p, (size_t)c->ginfo.n_fine_refs * sizeof(int), hipMemcpyDeviceToHost));
    } else {
        rtx_set_error("rtx_download_grid: unknown grid kind %u", which);
        return RTX_E_INVALID;
    }
    return RTX_OK;
}

int rtx_local_rows(const rtx_render_params* p, int32_t* n_rows, int32_t* rows) {
    int tile_rows, rank, world;
    int rc = check_params(p, &tile_rows, &rank, &world);
    if (rc) return rc;
    if (!n_rows) { rtx_set_error("rtx_local_rows: null n_rows"); return RTX_E_INVALID; }
    const int n_tiles = (p->height + tile_rows - 1) / tile_rows;
    int k = 0;
    for (int t = rank; t < n_tiles; t += world)
        for (int r = 0; r < tile_rows && t * tile_rows + r < p->height; r++) {
            if (rows) rows[k] = t * tile_rows + r;
            k++;
        }
    *n_rows = k;
    return RTX_OK;
}

// One frame of this rank's rows, enqueued on `stream`.  no_alloc: the entry point that allocates nothing -- the
// persistent frame pipeline only, its scratch taken from the device's file-scope array instead of the context's own
// buffers, sizes beyond that array rejected, no statistics.
static int render_frame(rtx_ctx* c, const rtx_camera* cam, const rtx_render_params* p, float* d_rgb, int32_t* d_ids,
                        void* stream, rtx_stats* stats, bool no_alloc) {
    if (!c || !cam || !d_rgb) { rtx_set_error("render: null argument"); return RTX_E_INVALID; }
    int tile_rows, rank, world;
    int rc = check_params(p, &tile_rows, &rank, &world);
    if (rc) return rc;
    if (!c->have_scene) { rtx_set_error("render: no scene set"); return RTX_E_STATE; }
    if (p->search == RTX_SEARCH_REFERENCE_LISTS && !c->have_ref) { rtx_set_error("render: reference cell lists not built"); return RTX_E_STATE; }
    if (p->search == RTX_SEARCH_NESTED && !c->have_nested) { rtx_set_error("render: nested grid not built"); return RTX_E_STATE; }
    DeviceGuard guard(c->device);
    if (!guard.ok) { rtx_set_error("hipSetDevice(%d) failed", c->device); return RTX_E_DEVICE; }
    hipStream_t st = (hipStream_t)stream;
    if ((rc = take_frame_status(c, false))) return rc;      // an earlier asynchronous frame that was abandoned

    const int nl = count_local_rows(p->height, tile_rows, rank, world);
    const int levels = p->max_depth + 1;
    const size_t pixels = (size_t)nl * p->width;
    const bool serial = (p->flags & RTX_RENDER_SERIAL_LANES) != 0;
    const bool level_queues = serial || (p->flags & RTX_RENDER_LEVEL_QUEUES) != 0;
    bool generations = !level_queues && (p->flags & RTX_RENDER_GENERATIONS) != 0;      // (or chosen below, by frame size)
    const int gen_levels = p->max_depth + 1;           // terms a path slot of the generation pipeline holds
    int blocks = c->n_cus * 4;   // 3 workgroups of 4 waves fit a CU at a time: the work cursors balance the rest
    if (p->workgroups_per_cu > 0) blocks = c->n_cus * std::min(p->workgroups_per_cu, 16);
    // path levels that do not fit the frame kernel's LDS strips (five levels in its four-wave build) live in global memory
    const int global_levels = std::max(levels - FrameBudget<4>::lds_levels, 1);
    const int tbx = (p->width + 7) / 8, tby = (p->height + 7) / 8;
    FrameScratch ws;
    HintState* hint = &c->hint;
    if (no_alloc) {
        if (level_queues || generations || (p->flags & RTX_RENDER_COUNT_WORK)) {
            rtx_set_error("rtx_render_device_noalloc: the level-queue and generation pipelines and the work counters need allocated workspace");
            return RTX_E_INVALID;
        }
        if (p->max_depth > RTX_NOALLOC_MAX_DEPTH || (long long)tbx * tby > RTX_NOALLOC_MAX_TILES) {
            rtx_set_error("rtx_render_device_noalloc: max_depth %d / %lld screen tiles exceed the scratch array's limits (max_depth <= %d, "
                          "ceil(width/8) * ceil(height/8) <= %d)", p->max_depth, (long long)tbx * tby, RTX_NOALLOC_MAX_DEPTH, RTX_NOALLOC_MAX_TILES);
            return RTX_E_LIMIT;
        }
        if (!c->scratch) { rtx_set_error("rtx_render_device_noalloc: the device scratch array was not found at rtx_create"); return RTX_E_STATE; }
        blocks = std::min(blocks, RTX_NOALLOC_MAX_BLOCKS);
        unsigned char* base = c->scratch;
        ws.counters = (int*)(base + RTX_NA_COUNTERS);
        ws.tile_map = (unsigned*)(base + RTX_NA_TILE_MAP);
        ws.prep_sync = (unsigned*)(base + RTX_NA_SYNC);
        ws.tile_class = (signed char*)(base + RTX_NA_TILE_CLASS);
        ws.tile_list = (int*)(base + RTX_NA_TILE_LIST);
        ws.tile_cost[0] = (int*)(base + RTX_NA_TILE_COST0);
        ws.tile_cost[1] = (int*)(base + RTX_NA_TILE_COST1);
        ws.path_stack = (float4*)(base + RTX_NA_PATH_STACK);
        // the scratch's cost maps only carry a hint for this context if no other context of the device used them since
        if (!scratch_take_over(c)) c->scratch_hint.valid = false;
        hint = &c->scratch_hint;
    } else if (level_queues) {
        if ((rc = ensure_workspace(c, std::max(pixels, (size_t)1), levels))) return rc;
        ws.counters = c->counters.p;
        ws.work = c->work.p;
    } else {
        // (instrumented frames log eight values per wave behind the 24 work counters)
        if ((rc = c->counters.reserve(CNT_TOTAL)) || (rc = c->work.reserve(24 + 8 * (size_t)blocks * (RTX_BLOCK / 64))) ||
            (rc = c->path_stack.reserve((size_t)blocks * RTX_BLOCK * global_levels)))
            return rc;
        ws.counters = c->counters.p;
        ws.work = c->work.p;
        ws.path_stack = c->path_stack.p;
    }
    if (stats) memset(stats, 0, sizeof *stats);
    if (nl == 0) return RTX_OK;

    DevFrame f;
    memset(&f, 0, sizeof f);
    f.W = p->width; f.H = p->height;
    f.n_local_rows = nl;
    f.tile_rows = tile_rows; f.rank = rank; f.world = world;
    f.cam = make_camera(cam, p->width, p->height);
    // Traced window: only pixels whose camera ray can reach the grid's box are traced; the rest of the frame is the
    // miss colour (src/main.cpp:104-106) and is written by two fills.
    int wx0, wx1, wy0, wy1;
    screen_window(f.cam, c->dgrid.lo, c->dgrid.hi, p->width, p->height, c->n_prims > 0, &wx0, &wx1, &wy0, &wy1);
    int lr_begin = 0, lr_end = 0;   // local rows of this rank with wy0 <= global row <= wy1
    {
        const int n_tiles = (p->height + tile_rows - 1) / tile_rows;
        int lr = 0;
        bool any = false;
        for (int t = rank; t < n_tiles; t += world)
            for (int r = 0; r < tile_rows && t * tile_rows + r < p->height; r++, lr++) {
                const int y = t * tile_rows + r;
                if (y >= wy0 && y <= wy1) { if (!any) { lr_begin = lr; any = true; } lr_end = lr + 1; }
            }
    }
    f.lr0 = lr_begin & ~7;
    f.bx0 = wx0 / 8;
    f.nbx = wx1 >= wx0 ? wx1 / 8 - f.bx0 + 1 : 0;
    f.n_items0 = f.nbx * ((lr_end - f.lr0 + 7) / 8) * 64;
    bool whole_frame = f.lr0 == 0 && lr_end == nl && f.bx0 == 0 && f.nbx == (p->width + 7) / 8;
    // nested search, persistent kernel: only the tiles of the window that can see an occupied fine cell are traced
    const int n_row_blocks = f.n_items0 > 0 ? (lr_end - f.lr0 + 7) / 8 : 0;
    // Register budget of the frame kernel: four waves per SIMD (128 registers) unless the caller asks for three (168).
    // With many pixels of the traced window per lane the frame is bound by throughput and the fourth wave wins (4K frame
    // of scene7 2.16 against 2.35 ms); a launch that shares the GPU with other frames fits four of all those launches'
    // workgroups on a CU instead of three (six frames of one workgroup per CU in flight: a quarter of the 4K frame 0.59 ->
    // 0.50 ms per frame, an eighth 0.35 -> 0.31).  Frames of a pixel or two per lane used to run the larger budget (each
    // ray of a chain finished sooner); measured again on this library, lone first frames, three against four waves:
    // scene5 0.434 / 0.413 ms, scene6 0.992 / 0.908, scene7 at 1280x720 0.834 / 0.802, at 2560x1440 1.625 / 1.463, a half /
    // quarter / eighth of its 4K frame 1.77 / 1.55, 1.11 / 1.06, 1.00 / 0.91; only scene4-specular (0.312 / 0.332) and scene1
    // (0.125 / 0.128) still prefer three.
    const double pixels_per_lane = (double)f.n_items0 / ((double)blocks * RTX_BLOCK);
    int frame_waves = 4;
    if (p->flags & RTX_RENDER_WAVES_3) frame_waves = 3;
    if (p->flags & RTX_RENDER_WAVES_4) frame_waves = 4;
    // Pipeline of the persistent launch.  Generations of one kind of ray make every state update dense but end each
    // generation on its slowest ray and each path on a longer chain of generations: they win where a frame is bound by
    // throughput and its end is hidden behind other frames -- a caller that keeps frames in flight (workgroups_per_cu
    // given) with many pixels per lane (scene7 @ 4K, three frames in flight: 1.86 -> 1.64 ms per frame; half a frame per
    // rank, four in flight: 0.96 -> 0.91; a quarter: 0.50 either way; an eighth: 0.31 -> 0.37) -- and lose on a lone
    // frame (2.2 against 1.95 ms).
    if (!level_queues && !no_alloc && !(p->flags & (RTX_RENDER_GENERATIONS | RTX_RENDER_NO_GENERATIONS)))
        generations = p->workgroups_per_cu > 0 && pixels_per_lane >= RTX_PIXELS_PER_LANE_FOR_GENERATIONS;
    if (generations && c->n_materials >= (1 << 27)) {      // a request packs the material index beside the level
        if (p->flags & RTX_RENDER_GENERATIONS) { rtx_set_error("render: the generation pipeline takes fewer than 2^27 materials"); return RTX_E_LIMIT; }
        generations = false;
    }
    if (generations &&
        (rc = c->gen_stacks.reserve((size_t)blocks * (RTX_BLOCK / 64) * rtx_gen_wave_f4(gen_levels))))
        return rc;
    bool tile_cull = !level_queues && p->search == RTX_SEARCH_NESTED && f.n_items0 > 0 && tbx <= 0xffff && (nl + 7) / 8 <= 0x7fff;
    if (p->flags & RTX_RENDER_NO_TILE_CULL) tile_cull = false;
    if (tile_cull) {
        if (!no_alloc) {
            const size_t n_slots = (size_t)tbx * ((nl + 7) / 8);
            // (the tile map, the cost maps and the frame-start kernel's counters are never cleared per frame: they start as
            // zeros, and a map that had to grow starts over -- its hint is then void)
            const bool regrow = n_slots + 4 > c->tile_cost[0].n || n_slots + 4 > c->tile_cost[1].n;
            if ((rc = c->tile_map.reserve_zeroed((size_t)tbx * tby + 4, st)) || (rc = c->prep_sync.reserve_zeroed(8, st)) ||
                (rc = c->tile_list.reserve((size_t)n_row_blocks * f.nbx)) ||
                (rc = c->tile_class.reserve((size_t)n_row_blocks * f.nbx)) || (rc = c->tile_cost[0].reserve_zeroed(n_slots + 4, st)) ||
                (rc = c->tile_cost[1].reserve_zeroed(n_slots + 4, st)))
                return rc;
            if (regrow) c->hint.valid = false;
            ws.tile_map = c->tile_map.p;
            ws.prep_sync = c->prep_sync.p;
            ws.tile_class = c->tile_class.p;
            ws.tile_list = c->tile_list.p;
            ws.tile_cost[0] = c->tile_cost[0].p;
            ws.tile_cost[1] = c->tile_cost[1].p;
        }
        f.tile_list = ws.tile_list;
        f.tbx = tbx;
        whole_frame = false;
    }
    f.max_depth = p->max_depth;
    f.bias = p->eps_bias;
    f.walk_budget = p->walk_budget > 0 ? p->walk_budget : 3;
    f.ready_threshold = p->ready_threshold > 0 ? std::min(p->ready_threshold, 64) : 56;
    f.drain_num = p->drain_eighths > 0 ? std::min(p->drain_eighths, 8) : 2;
    // Paths per wave.  A frame with many pixels per lane is bound by throughput: every lane carries a path and lanes
    // help with shadow rays only once their wave drains.  A frame with few (a small image, a rank's share of a large
    // one) is a handful of chains of dependent rays per lane: a wave then carries fewer paths and keeps the other
    // lanes free for the shadow rays of fresh hits from the start, which shortens every chain (measured on scene7 @ 4K,
    // one frame at a time: an eighth of the frame 1.58 -> 0.96 ms, a quarter 1.63 -> 1.21, a half 2.02 -> 1.89 with 48
    // lanes; scene4 / scene5 at 640x480 0.44 -> 0.33 / 0.57 -> 0.47 ms).  A caller that shares the GPU among several
    // frames in flight (workgroups_per_cu given) is bound by throughput again unless the frame is very small.
    int path_lanes = 64;
    if (pixels_per_lane < RTX_PIXELS_PER_LANE_FOR_EARLY_HELPERS) path_lanes = 32;
    else if (pixels_per_lane < 3 * RTX_PIXELS_PER_LANE_FOR_EARLY_HELPERS && p->workgroups_per_cu <= 0) path_lanes = 48;
    if (p->path_lanes > 0) path_lanes = std::min(p->path_lanes, 64);
    f.helpers = path_lanes < 64 ? 2 : 1;
    if (p->flags & RTX_RENDER_HELPERS_EARLY) f.helpers = 2;
    if (p->flags & RTX_RENDER_HELPERS_LATE) f.helpers = 1;
    if (p->flags & RTX_RENDER_NO_HELPERS) f.helpers = 0;
    f.path_lanes = f.helpers == 2 ? path_lanes : 64;
    {
        // generation pipeline: a frame ends sooner when its waves stop stocking up on paths toward the end (a lone 4K
        // frame 2.74 -> 2.6 ms); with several frames in flight the end of one frame overlaps the next and it matters
        // little (three in flight 1.570 -> 1.563 ms per frame; from two pixels per lane on it costs: 1.60, 1.68)
        const int eighths = p->endgame_eighths != 0 ? std::max(p->endgame_eighths, 0) : RTX_GEN_ENDGAME_EIGHTHS;
        f.endgame_pixels = (int)std::min<long long>((long long)eighths * blocks * RTX_BLOCK / 8, 0x7fffffff);
    }
    f.round_limit = p->round_limit > 0 ? (long long)p->round_limit : derived_round_limit(c, f, p->max_depth, blocks);
    f.abort_host = c->abort_dev;
    f.rgb = d_rgb;
    f.ids = d_ids;
    f.counters = ws.counters;
    f.work = ws.work;
    if (level_queues) { f.hP = c->hP.p; f.hN = c->hN.p; f.hD = c->hD.p; f.pix0 = c->pix0.p; }

    const DevScene sc = dev_scene(c);
    const bool count = (p->flags & RTX_RENDER_COUNT_WORK) != 0;
    const bool timed = stats != nullptr;

    hipEvent_t* ev = c->ev;
    HIP_TRY(hipEventRecord(ev[0], st));              // frame time includes the start of the frame
    if (tile_cull) {
        // Persistent kernel over the tiles that can see geometry: ONE launch before the frame kernel zeroes the counters,
        // marks and classifies the tiles and writes the list (k_frame_prep); the frame kernel itself writes the miss
        // colour over the blocks it does not trace.  Nothing is cleared per frame: the tile map and the cost maps hold
        // stamps that differ from frame to frame.
        PrepState* ps = no_alloc ? &g_scratch_prep[c->device < RTX_MAX_DEVICES ? c->device : 0] : &c->prep;
        std::unique_lock<std::mutex> scratch_lock(g_scratch_mutex, std::defer_lock);
        if (no_alloc) scratch_lock.lock();
        if (ps->seq >= RTX_MAX_STAMP) {
            // the stamps start over (after 2^26 frames): the maps go back to zeros, on the frame's stream
            HIP_TRY(hipMemsetAsync(ws.tile_map, 0, (size_t)tbx * tby * sizeof(unsigned), st));
            for (int k = 0; k < 2; k++) HIP_TRY(hipMemsetAsync(ws.tile_cost[k], 0, (size_t)tbx * ((nl + 7) / 8) * sizeof(int), st));
            HIP_TRY(hipMemsetAsync(ws.counters + CNT_TILES_ALL, 0, sizeof(int), st));   // (holds a stamp as well)
            ps->seq = 0;
            hint->valid = false;
        }
        const unsigned stamp = ++ps->seq;
        // the frame kernel fills what it does not trace when every 16-byte store of the fill is aligned; otherwise the
        // buffers are cleared here, as for the pipelines without a tile list
#ifndef RT_KERNEL_FILL
#define RT_KERNEL_FILL 1
#endif
        const bool fill_ok = RT_KERNEL_FILL && p->width % 4 == 0 && ((uintptr_t)d_rgb & 15u) == 0 && (!d_ids || ((uintptr_t)d_ids & 15u) == 0);
        if (!fill_ok) {
            HIP_TRY(hipMemsetAsync(d_rgb, 0, pixels * 3 * sizeof(float), st));
            if (d_ids) HIP_TRY(hipMemsetAsync(d_ids, 0xff, pixels * sizeof(int32_t), st));
        }
        // the map this frame records its per-block path depths into (the other one orders this frame's list)
        hint->cur ^= 1;
        // the previous frame's per-block path depths order this frame's list if they were recorded for the same
        // image geometry, rank split and grid (the camera may have moved: the order is a hint, never a result)
        const int key[6] = {p->width, p->height, tile_rows, rank, world, c->grid_generation};
        const bool feedback = hint->valid && memcmp(key, hint->key, sizeof key) == 0 && !(p->flags & RTX_RENDER_NO_TILE_HINT);
        // a tile takes the deepest class recorded within hint_radius tiles around it: one tile allows for a camera
        // that moved a little; the very same camera needs no allowance (4K frame of scene7: 2.08 against 2.19 ms)
        f.hint_radius = feedback && memcmp(cam, &hint->cam, sizeof *cam) == 0 ? 0 : 1;
        hint->cam = *cam;
        f.tile_cost = ws.tile_cost[hint->cur];
        f.cost_stamp = (int)(stamp << RTX_COST_BITS);
        hint->stamp[hint->cur] = f.cost_stamp;
        memcpy(hint->key, key, sizeof key);
        hint->valid = true;
        f.fill_cls = fill_ok ? ws.tile_class : nullptr;
        f.fill_window_row_blocks = n_row_blocks;

        PrepArgs a;
        memset(&a, 0, sizeof a);
        a.f = f;
        a.g = c->dgrid;
        a.occupied_cells = c->ocell.p;
        a.pad = c->fine_pad;
        a.n_occ = (int)c->ginfo.n_occupied_fine_cells;
        a.tby = tby;
        a.tile_map = ws.tile_map;
        a.stamp = stamp;
        a.n_row_blocks = n_row_blocks;
        a.prev_cost = feedback ? ws.tile_cost[hint->cur ^ 1] : nullptr;
        a.prev_stamp = hint->stamp[hint->cur ^ 1];
        a.cls = ws.tile_class;
        a.tile_list = ws.tile_list;
        a.sync = ws.prep_sync;
        a.work = count ? ws.work : nullptr;
        const int n_blocks_window = n_row_blocks * f.nbx;
        a.n_chunks[0] = 1 + (a.n_occ + 255) / 256;              // (chunk 0 zeroes the counters)
        a.n_chunks[1] = a.n_chunks[2] = (n_blocks_window + 255) / 256;
        // (a launch that shares the GPU with other frames' persistent workgroups queues for its places: the fewer
        // workgroups the frame start needs, the sooner the last of them has run -- three 4K frames in flight 1.60 -> 1.59
        // ms per frame, six eighths of one 0.313 -> 0.301 with 32; 16 and 8 are slower again, and a lone 4K frame
        // wants them all: 68 us against 113 with 32)
        const int prep_cap = p->workgroups_per_cu > 0 ? std::min(c->n_cus, 32) : c->n_cus;
        const int prep_blocks = std::max(1, std::min(prep_cap, std::max(a.n_chunks[0], a.n_chunks[1])));
        for (int k = 0; k < 3; k++) a.ticket_base[k] = ps->ticket[k];
        for (int k = 0; k < 2; k++) a.done_base[k] = ps->done[k];
        hipLaunchKernelGGL(k_frame_prep, dim3(prep_blocks), dim3(256), 0, st, a);
        // every workgroup draws one ticket beyond a phase's chunks; every chunk of the first two phases is reported once
        for (int k = 0; k < 3; k++) ps->ticket[k] += (unsigned)(a.n_chunks[k] + prep_blocks);
        for (int k = 0; k < 2; k++) ps->done[k] += (unsigned)a.n_chunks[k];
    } else {
        // one launch clears everything a frame starts from; caller buffers whose size or address is not a multiple of 16
        // bytes fall back to memsets
        ClearRanges cr;
        memset(&cr, 0, sizeof cr);
        bool clear_overflow = false;
        auto add = [&](void* ptr, size_t bytes, unsigned pattern) {
            if (cr.n >= RTX_CLEAR_RANGES) { clear_overflow = true; return; }
            cr.ptr[cr.n] = (uint4*)ptr; cr.words[cr.n] = (long long)((bytes + 15) / 16); cr.pattern[cr.n] = pattern; cr.n++;
        };
        if (!whole_frame) {
            const size_t rgb_bytes = pixels * 3 * sizeof(float), id_bytes = pixels * sizeof(int32_t);
            if (((uintptr_t)d_rgb & 15u) == 0 && rgb_bytes % 16 == 0) add(d_rgb, rgb_bytes, 0u);            // 0.0f
            else HIP_TRY(hipMemsetAsync(d_rgb, 0, rgb_bytes, st));
            if (d_ids) {
                if (((uintptr_t)d_ids & 15u) == 0 && id_bytes % 16 == 0) add(d_ids, id_bytes, 0xffffffffu);   // -1
                else HIP_TRY(hipMemsetAsync(d_ids, 0xff, id_bytes, st));
            }
        }
        add(ws.counters, CNT_TOTAL * sizeof(int), 0u);
        if (count) add(ws.work, 24 * sizeof(unsigned long long), 0u);
        if (clear_overflow) { rtx_set_error("render: more ranges to clear than the clear kernel takes"); return RTX_E_STATE; }
        long long total = 0;
        for (int k = 0; k < cr.n; k++) total += cr.words[k];
        const int clear_blocks = (int)std::min<long long>((total + 255) / 256, (long long)c->n_cus * 16);
        hipLaunchKernelGGL(k_frame_clear, dim3(std::max(clear_blocks, 1)), dim3(256), 0, st, cr);
    }
    HIP_TRY(hipEventRecord(ev[3], st));              // start of the kernels proper
    if (!level_queues) {
        float4* const strips = generations ? c->gen_stacks.p : ws.path_stack;
        const int n_levels = generations ? gen_levels : global_levels;
        if (p->search == RTX_SEARCH_REFERENCE_LISTS) launch_frame<RT_SEARCH_REF>(c, sc, f, strips, blocks, st, count, n_levels, frame_waves, generations);
        else launch_frame<RT_SEARCH_NESTED>(c, sc, f, strips, blocks, st, count, n_levels, frame_waves, generations);
        HIP_TRY(hipEventRecord(ev[1], st));
    }
    for (int l = 0; level_queues && l < levels; l++) {
        hipEvent_t* le = ev + 4 + 3 * l;
        f.hL = c->hL[l].p;
        f.hChild = c->hChild[l].p;
        f.hChild_prev = l > 0 ? c->hChild[l - 1].p : nullptr;
        f.rO_in = c->rO[(l + 1) & 1].p; f.rD_in = c->rD[(l + 1) & 1].p;
        f.rO_out = c->rO[l & 1].p; f.rD_out = c->rD[l & 1].p;
        if (p->search == RTX_SEARCH_REFERENCE_LISTS) launch_level_any<RT_SEARCH_REF>(c, sc, f, l, blocks, st, timed, le, count, serial);
        else launch_level_any<RT_SEARCH_NESTED>(c, sc, f, l, blocks, st, timed, le, count, serial);
    }
    if (level_queues) HIP_TRY(hipEventRecord(ev[1], st));
    for (int l = levels - 1; level_queues && l >= 0; l--) {
        f.hL = c->hL[l].p;
        f.hL_next = l + 1 < levels ? c->hL[l + 1].p : nullptr;
        f.hChild = c->hChild[l].p;
        hipLaunchKernelGGL(k_resolve, dim3(c->n_cus * 4), dim3(RTX_BLOCK), 0, st, sc, f, l);
    }
    HIP_TRY(hipEventRecord(ev[2], st));
    HIP_TRY(hipGetLastError());
    c->frame_pending = !level_queues;
    c->frame_timed = true;

    if (stats) {
        if ((rc = take_frame_status(c, true))) return rc;
        HIP_TRY(hipEventSynchronize(ev[2]));
        int cnt[CNT_TOTAL];
        HIP_TRY(hipMemcpyAsync(cnt, ws.counters, sizeof cnt, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        stats->primary_rays = (int64_t)pixels;
        // the level-synchronous pipelines walk every pixel's camera ray
        stats->traced_primary_rays = level_queues ? (int64_t)pixels : (int64_t)cnt[CNT_PIXELS];
        stats->culled_pixels = stats->primary_rays - stats->traced_primary_rays;
        for (int l = 0; l < levels; l++) {
            stats->hits_per_level[l] = cnt[CNT_HITS + l];
            stats->shadow_rays += (int64_t)cnt[CNT_HITS + l] * c->n_lights;
            if (l < p->max_depth) stats->reflection_rays += cnt[CNT_HITS + l];
            float a = 0, b = 0;
            if (level_queues) {
                (void)hipEventElapsedTime(&a, ev[4 + 3 * l], ev[4 + 3 * l + 1]);
                (void)hipEventElapsedTime(&b, ev[4 + 3 * l + 1], ev[4 + 3 * l + 2]);
            }
            stats->trace_ms += a;
            stats->shade_ms += b;
            stats->level_trace_ms[l] = a;
            stats->level_shade_ms[l] = b;
        }
        (void)hipEventElapsedTime(&stats->frame_ms, ev[0], ev[2]);
        if (level_queues) {
            stats->trace_launches = stats->shade_launches = stats->resolve_launches = levels;
            stats->pipeline = 2;
            (void)hipEventElapsedTime(&stats->resolve_ms, ev[1], ev[2]);
        } else {
            stats->trace_launches = 1;                 // the persistent frame kernel does everything
            stats->pipeline = generations ? 1 : 0;
            (void)hipEventElapsedTime(&stats->trace_ms, ev[3], ev[1]);
        }
        if (count && !level_queues) c->wave_log_waves = (size_t)blocks * (RTX_BLOCK / 64);
        if (count) {
            unsigned long long w[24];
            HIP_TRY(hipMemcpy(w, c->work.p, sizeof w, hipMemcpyDeviceToHost));
            for (int k = 0; k < 4; k++) { stats->wave_events[k] = (int64_t)w[16 + k]; stats->wave_cycles[k] = (int64_t)w[20 + k]; }
            for (int k = 0; k < 8; k++) { stats->trace_work[k] = (int64_t)w[k]; stats->shade_work[k] = (int64_t)w[8 + k]; }
            stats->tri_tests = (int64_t)(w[0] + w[8]); stats->sphere_tests = (int64_t)(w[1] + w[9]);
            stats->xform_evals = (int64_t)(w[2] + w[10]); stats->cell_steps = (int64_t)(w[3] + w[11]);
            stats->fine_steps = (int64_t)(w[4] + w[12]); stats->light_evals = (int64_t)(w[5] + w[13]);
            stats->nearest_hits = (int64_t)(w[6] + w[14]);
        }
    }
    return RTX_OK;
}

int rtx_render_device(rtx_ctx* c, const rtx_camera* cam, const rtx_render_params* p, float* d_rgb, int32_t* d_ids,
                      void* stream, rtx_stats* stats) {
    return render_frame(c, cam, p, d_rgb, d_ids, stream, stats, false);
}

int rtx_render_device_noalloc(rtx_ctx* c, const rtx_camera* cam, const rtx_render_params* p, float* d_rgb, int32_t* d_ids, void* stream) {
    return render_frame(c, cam, p, d_rgb, d_ids, stream, nullptr, true);
}

// The pixel loop with the result on the host: floats (rgb_out) or, converted on the device first, bytes (rgb8_out).
// This rank's rows arrive in the order rtx_local_rows() lists them; with one rank that is the image itself and one
// copy moves it, otherwise every tile of rows is copied to its place in the caller's full-size image.
static int render_to_host(rtx_ctx* c, const rtx_camera* cam, const rtx_render_params* p, float* rgb_out, uint8_t* rgb8_out, int32_t order,
                          int32_t* prim_id_out, rtx_stats* stats) {
    int tile_rows, rank, world;
    int rc = check_params(p, &tile_rows, &rank, &world);
    if (rc) return rc;
    DeviceGuard guard(c->device);
    if (!guard.ok) { rtx_set_error("hipSetDevice(%d) failed", c->device); return RTX_E_DEVICE; }
    const int nl = count_local_rows(p->height, tile_rows, rank, world);
    const size_t pixels = (size_t)nl * p->width;
    if ((rc = c->rgb_local.reserve(std::max(pixels, (size_t)1) * 3))) return rc;
    if (prim_id_out && (rc = c->ids_local.reserve(std::max(pixels, (size_t)1)))) return rc;
    if (rgb8_out && (rc = c->rgb8_local.reserve(std::max(pixels, (size_t)1) * 3 + 16))) return rc;
    rtx_stats local_stats;
    rc = render_frame(c, cam, p, c->rgb_local.p, prim_id_out ? c->ids_local.p : nullptr, nullptr, stats ? stats : &local_stats, false);
    if (rc) return rc;
    if (nl == 0) return RTX_OK;
    if (rgb8_out) {
        if ((rc = rtx_quantize_device(c, c->rgb_local.p, (int64_t)pixels, c->rgb8_local.p, order, nullptr))) return rc;
    }
    const size_t W = (size_t)p->width;
    const size_t px_bytes = rgb8_out ? 3 : 3 * sizeof(float);
    const char* src = rgb8_out ? (const char*)c->rgb8_local.p : (const char*)c->rgb_local.p;
    char* dst = rgb8_out ? (char*)rgb8_out : (char*)rgb_out;
    const int n_tiles = (p->height + tile_rows - 1) / tile_rows;
    size_t lr = 0;
    for (int t = rank; t < n_tiles; t += world) {
        // consecutive tiles of one rank are consecutive in the image only when it is the only rank
        const size_t rows_here = world == 1 ? (size_t)p->height : (size_t)std::min(tile_rows, p->height - t * tile_rows);
        const size_t y0 = (size_t)t * tile_rows;
        HIP_TRY(hipMemcpy(dst + y0 * W * px_bytes, src + lr * W * px_bytes, rows_here * W * px_bytes, hipMemcpyDeviceToHost));
        if (prim_id_out) HIP_TRY(hipMemcpy(prim_id_out + y0 * W, c->ids_local.p + lr * W, rows_here * W * sizeof(int32_t), hipMemcpyDeviceToHost));
        lr += rows_here;
        if (world == 1) break;
    }
    return RTX_OK;
}

int rtx_render(rtx_ctx* c, const rtx_camera* cam, const rtx_render_params* p, float* rgb_out, int32_t* prim_id_out, rtx_stats* stats) {
    if (!c || !cam || !rgb_out) { rtx_set_error("rtx_render: null argument"); return RTX_E_INVALID; }
    return render_to_host(c, cam, p, rgb_out, nullptr, RTX_ORDER_RGB, prim_id_out, stats);
}

int rtx_render_u8(rtx_ctx* c, const rtx_camera* cam, const rtx_render_params* p, uint8_t* rgb8_out, int32_t order, int32_t* prim_id_out,
                  rtx_stats* stats) {
    if (!c || !cam || !rgb8_out) { rtx_set_error("rtx_render_u8: null argument"); return RTX_E_INVALID; }
    if (order != RTX_ORDER_RGB && order != RTX_ORDER_BGR) { rtx_set_error("rtx_render_u8: unknown channel order"); return RTX_E_INVALID; }
    return render_to_host(c, cam, p, nullptr, rgb8_out, order, prim_id_out, stats);
}

int rtx_frame_status(rtx_ctx* c) {
    if (!c) { rtx_set_error("rtx_frame_status: null context"); return RTX_E_INVALID; }
    DeviceGuard guard(c->device);
    if (!guard.ok) { rtx_set_error("hipSetDevice(%d) failed", c->device); return RTX_E_DEVICE; }
    return take_frame_status(c, true);
}

int rtx_frame_times(rtx_ctx* c, float* frame_ms, float* kernel_ms) {
    if (!c) { rtx_set_error("rtx_frame_times: null context"); return RTX_E_INVALID; }
    DeviceGuard guard(c->device);
    if (!guard.ok) { rtx_set_error("hipSetDevice(%d) failed", c->device); return RTX_E_DEVICE; }
    if (!c->have_events || !c->frame_timed) { rtx_set_error("rtx_frame_times: the context has rendered no frame"); return RTX_E_STATE; }
    HIP_TRY(hipEventSynchronize(c->ev[2]));
    float a = 0.0f, b = 0.0f;
    HIP_TRY(hipEventElapsedTime(&a, c->ev[0], c->ev[2]));
    HIP_TRY(hipEventElapsedTime(&b, c->ev[3], c->ev[1]));
    if (frame_ms) *frame_ms = a;
    if (kernel_ms) *kernel_ms = b;
    return RTX_OK;
}

int rtx_download_wave_log(rtx_ctx* c, uint64_t* out, int64_t max_waves, int64_t* n_waves) {
    if (!c || !n_waves || max_waves < 0 || (max_waves > 0 && !out)) { rtx_set_error("rtx_download_wave_log: bad argument"); return RTX_E_INVALID; }
    DeviceGuard guard(c->device);
    if (!guard.ok) { rtx_set_error("hipSetDevice(%d) failed", c->device); return RTX_E_DEVICE; }
    const size_t n = std::min<size_t>(c->wave_log_waves, (size_t)max_waves);
    *n_waves = (int64_t)c->wave_log_waves;
    if (n > 0) {
        if (c->work.n < 24 + 8 * c->wave_log_waves) { rtx_set_error("rtx_download_wave_log: no log held"); return RTX_E_STATE; }
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, c->work.p + 24, 8 * n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    return RTX_OK;
}

int rtx_debug_set_frame_counters(rtx_ctx* c, uint32_t frames_before_wrap, uint32_t tickets_before_wrap) {
    if (!c) { rtx_set_error("rtx_debug_set_frame_counters: bad argument"); return RTX_E_INVALID; }
    DeviceGuard guard(c->device);
    if (!guard.ok) { rtx_set_error("hipSetDevice(%d) failed", c->device); return RTX_E_DEVICE; }
    if (!c->prep_sync.p || c->prep_sync.n < 8) { rtx_set_error("rtx_debug_set_frame_counters: no frame rendered on this context yet"); return RTX_E_STATE; }
    HIP_TRY(hipDeviceSynchronize());
    PrepState& ps = c->prep;
    // (frame numbers only ever grow between wraps, which is all the stamps in the maps need)
    ps.seq = std::max(ps.seq, RTX_MAX_STAMP - std::min(frames_before_wrap, RTX_MAX_STAMP));
    const unsigned at = 0u - tickets_before_wrap;
    unsigned words[8] = {at, at, at, at, at, 0u, 0u, 0u};
    for (int k = 0; k < 3; k++) ps.ticket[k] = at;
    for (int k = 0; k < 2; k++) ps.done[k] = at;
    HIP_TRY(hipMemcpy(c->prep_sync.p, words, sizeof words, hipMemcpyHostToDevice));
    HIP_TRY(hipDeviceSynchronize());
    return RTX_OK;
}

int rtx_intersect(rtx_ctx* c, int32_t search, int32_t n, const float* O, const float* D, const float* tl, float* out_t,
                  int32_t* out_prim, float* out_pos, float* out_n) {
    if (!c || n < 0 || (n > 0 && (!O || !D || !out_t || !out_prim))) { rtx_set_error("rtx_intersect: bad argument"); return RTX_E_INVALID; }
    if (!c->have_scene) { rtx_set_error("rtx_intersect: no scene set"); return RTX_E_STATE; }
    if (search == RTX_SEARCH_REFERENCE_LISTS && !c->have_ref) { rtx_set_error("rtx_intersect: reference cell lists not built"); return RTX_E_STATE; }
    if (search == RTX_SEARCH_NESTED && !c->have_nested) { rtx_set_error("rtx_intersect: nested grid not built"); return RTX_E_STATE; }
    if (search != RTX_SEARCH_REFERENCE_LISTS && search != RTX_SEARCH_NESTED) { rtx_set_error("rtx_intersect: unknown search mode"); return RTX_E_INVALID; }
    if (n == 0) return RTX_OK;
    DeviceGuard guard(c->device);
    if (!guard.ok) { rtx_set_error("hipSetDevice(%d) failed", c->device); return RTX_E_DEVICE; }
    DevBuf<float> dO, dD, dT, dOutT, dPos, dN;
    DevBuf<int> dPrim;
    int rc;
    if ((rc = dO.reserve(3 * (size_t)n)) || (rc = dD.reserve(3 * (size_t)n)) || (rc = dT.reserve((size_t)n)) || (rc = dOutT.reserve((size_t)n)) ||
        (rc = dPos.reserve(3 * (size_t)n)) || (rc = dN.reserve(3 * (size_t)n)) || (rc = dPrim.reserve((size_t)n)))
        goto done;
    rc = RTX_E_DEVICE;
    if (hipMemcpy(dO.p, O, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) goto done;
    if (hipMemcpy(dD.p, D, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) goto done;
    if (tl && hipMemcpy(dT.p, tl, (size_t)n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) goto done;
    {
        const DevScene sc = dev_scene(c);
        const int blocks = (n + 63) / 64;
        if (search == RTX_SEARCH_REFERENCE_LISTS)
            hipLaunchKernelGGL((k_probe<RT_SEARCH_REF>), dim3(blocks), dim3(64), 0, 0, sc, c->dgrid, n, (const float*)dO.p, (const float*)dD.p,
                               tl ? (const float*)dT.p : (const float*)nullptr, dOutT.p, dPrim.p, dPos.p, dN.p);
        else
            hipLaunchKernelGGL((k_probe<RT_SEARCH_NESTED>), dim3(blocks), dim3(64), 0, 0, sc, c->dgrid, n, (const float*)dO.p, (const float*)dD.p,
                               tl ? (const float*)dT.p : (const float*)nullptr, dOutT.p, dPrim.p, dPos.p, dN.p);
    }
    if (hipDeviceSynchronize() != hipSuccess) goto done;
    if (hipMemcpy(out_t, dOutT.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) goto done;
    if (hipMemcpy(out_prim, dPrim.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) goto done;
    if (out_pos && hipMemcpy(out_pos, dPos.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) goto done;
    if (out_n && hipMemcpy(out_n, dN.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) goto done;
    rc = RTX_OK;
done:
    if (rc == RTX_E_DEVICE) rtx_set_error("rtx_intersect: HIP call failed: %s", hipGetErrorString(hipGetLastError()));
    dO.release(); dD.release(); dT.release(); dOutT.release(); dPos.release(); dN.release(); dPrim.release();
    return rc;
}

int rtx_camera_rays(rtx_ctx* c, const rtx_camera* cam, int32_t W, int32_t H, int32_t n, const int32_t* xy, float* dirs) {
    if (!c || !cam || W <= 0 || H <= 0 || n < 0 || (n > 0 && (!xy || !dirs))) { rtx_set_error("rtx_camera_rays: bad argument"); return RTX_E_INVALID; }
    if (n == 0) return RTX_OK;
    DeviceGuard guard(c->device);
    if (!guard.ok) { rtx_set_error("hipSetDevice(%d) failed", c->device); return RTX_E_DEVICE; }
    DevBuf<int> dxy;
    DevBuf<float> dd;
    int rc;
    if ((rc = dxy.reserve(2 * (size_t)n)) || (rc = dd.reserve(3 * (size_t)n))) { dxy.release(); dd.release(); return rc; }
    rc = RTX_E_DEVICE;
    if (hipMemcpy(dxy.p, xy, 2 * (size_t)n * sizeof(int), hipMemcpyHostToDevice) == hipSuccess) {
        hipLaunchKernelGGL(k_camera_rays, dim3((n + 255) / 256), dim3(256), 0, 0, make_camera(cam, W, H), n, (const int*)dxy.p, dd.p);
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(dirs, dd.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess) rc = RTX_OK;
    }
    if (rc) rtx_set_error("rtx_camera_rays: HIP call failed: %s", hipGetErrorString(hipGetLastError()));
    dxy.release(); dd.release();
    return rc;
}

int rtx_quantize_device(rtx_ctx* c, const float* d_rgb, int64_t n_pixels, uint8_t* d_out, int32_t order, void* stream) {
    if (!c || n_pixels < 0 || (n_pixels > 0 && (!d_rgb || !d_out)) || (order != RTX_ORDER_RGB && order != RTX_ORDER_BGR)) {
        rtx_set_error("rtx_quantize_device: bad argument");
        return RTX_E_INVALID;
    }
    if (((uintptr_t)d_rgb & 15u) || ((uintptr_t)d_out & 3u)) {
        rtx_set_error("rtx_quantize_device: d_rgb must be 16-byte aligned and d_out 4-byte aligned");
        return RTX_E_INVALID;
    }
    if (n_pixels == 0) return RTX_OK;
    DeviceGuard guard(c->device);
    if (!guard.ok) { rtx_set_error("hipSetDevice(%d) failed", c->device); return RTX_E_DEVICE; }
    const long long quads = std::max<long long>(1, n_pixels >> 2);
    const int blocks = (int)std::min<long long>((quads + 255) / 256, 256 * 16);
    hipStream_t st = (hipStream_t)stream;
    if (order == RTX_ORDER_BGR) hipLaunchKernelGGL(k_quantize<true>, dim3(blocks), dim3(256), 0, st, d_rgb, (long long)n_pixels, d_out);
    else hipLaunchKernelGGL(k_quantize<false>, dim3(blocks), dim3(256), 0, st, d_rgb, (long long)n_pixels, d_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rtx_set_error("rtx_quantize_device: launch failed: %s", hipGetErrorString(e)); return RTX_E_DEVICE; }
    return RTX_OK;
}

}  // extern "C"
