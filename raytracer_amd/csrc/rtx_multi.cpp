// Several GPUs of one node behind one handle: the scene and its grid replicated on every GPU, a frame split into
// interleaved row tiles (tile t of `tile_rows` rows -> rank t mod N), every rank's rows copied into the frame on the
// first GPU, tile by tile, over the point-to-point links (hipMemcpyPeerAsync: a gather to one root uses each peer's own
// link to it, so no collective library is involved).  The reference has no counterpart: its only parallelism is
// OpenMP over the rows of the pixel loop (src/main.cpp:126 of the reference tree).
//
// One worker thread per rank issues that rank's HIP calls, so that the launches of all GPUs go out side by side and
// not one GPU after the other from a single host thread.  A device id may be listed more than once: the ranks then
// share that GPU (how the N > 1 paths are rehearsed on a one-GPU machine).
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "rtx.h"
#include "rtx_internal.h"

namespace {

// Keeps the caller's current device: every entry point that makes another device current puts the caller's back.
struct DeviceKeeper {
    int prev = -1;
    DeviceKeeper() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceKeeper() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceKeeper(const DeviceKeeper&) = delete;
    DeviceKeeper& operator=(const DeviceKeeper&) = delete;
};

struct Rank {
    int device = 0;
    rtx_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    float* rows = nullptr;            // this rank's rows of the frame (device memory on its GPU)
    int32_t* row_ids = nullptr;
    size_t rows_cap = 0, ids_cap = 0; // floats / ints allocated
    bool direct = true;               // this GPU can address the first GPU's memory (itself, or peer access enabled)
    // worker
    std::thread thread;
    std::function<int()> job;
    bool has_job = false, quit = false;
    int rc = 0;
    std::string err;
    rtx_stats stats{};
};

}  // namespace

struct rtx_multi {
    std::vector<Rank> ranks;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    int pending = 0;
    float* d_frame = nullptr;         // assembled frame on the first GPU (rtx_multi_render only)
    int32_t* d_frame_ids = nullptr;
    size_t frame_cap = 0, frame_ids_cap = 0;
    uint8_t* d_frame8 = nullptr;      // ... after the 8-bit conversion (rtx_multi_render_u8)
    size_t frame8_cap = 0;
};

namespace {

void worker(rtx_multi* m, Rank* r) {
    (void)hipSetDevice(r->device);
    for (;;) {
        std::function<int()> job;
        {
            std::unique_lock<std::mutex> lock(m->mu);
            m->cv_job.wait(lock, [&] { return r->has_job || r->quit; });
            if (r->quit) return;
            job = r->job;
        }
        int rc;
        std::string err;
        try {
            rc = job();
        } catch (const std::bad_alloc&) {
            rc = RTX_E_NOMEM;
            rtx_set_error("out of host memory on a rank's worker thread");
        } catch (...) {                                   // nothing may leave a worker thread: it would end the process
            rc = RTX_E_STATE;
            rtx_set_error("unexpected exception on a rank's worker thread");
        }
        if (rc != RTX_OK) err = rtx_last_error();
        {
            std::lock_guard<std::mutex> lock(m->mu);
            r->rc = rc;
            r->err = err;
            r->has_job = false;
            if (--m->pending == 0) m->cv_done.notify_all();
        }
    }
}

// Runs make_job(rank) on every rank's thread and waits; the first failing rank's code and message are reported.
int run_all(rtx_multi* m, const std::function<std::function<int()>(int)>& make_job) {
    {
        std::lock_guard<std::mutex> lock(m->mu);
        for (size_t k = 0; k < m->ranks.size(); k++) {
            m->ranks[k].job = make_job((int)k);
            m->ranks[k].has_job = true;
        }
        m->pending = (int)m->ranks.size();
    }
    m->cv_job.notify_all();
    std::unique_lock<std::mutex> lock(m->mu);
    m->cv_done.wait(lock, [&] { return m->pending == 0; });
    for (size_t k = 0; k < m->ranks.size(); k++)
        if (m->ranks[k].rc != RTX_OK) {
            rtx_set_error("rank %zu (device %d): %s", k, m->ranks[k].device, m->ranks[k].err.c_str());
            return m->ranks[k].rc;
        }
    return RTX_OK;
}

int hip_fail(const char* what, hipError_t e) {
    rtx_set_error("%s failed: %s", what, hipGetErrorString(e));
    return RTX_E_DEVICE;
}

// a rank on the first GPU itself copies inside that GPU; every other rank over its link to it
hipError_t copy_to_root(void* dst, int root, const void* src, int device, size_t bytes, hipStream_t stream) {
    if (device == root) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
    return hipMemcpyPeerAsync(dst, root, src, device, bytes, stream);
}

// All tiles of one rank in one copy.  The rank's rows are its tiles one after the other (`tile_bytes` each) and its
// tiles stand `world` tiles apart in the frame: a pitched copy of `n_full` lines of tile_bytes, destination pitch
// world * tile_bytes.  The frame's last tile may be short and is copied by itself.  A GPU that cannot address the
// first GPU's memory (no peer access) copies tile by tile through the runtime's staging instead.
hipError_t copy_tiles_to_root(char* frame, int root, const char* rows, const Rank& r, int rank, int world, int n_tiles, int tile_rows,
                              int height, size_t row_bytes, hipStream_t stream) {
    const size_t tile_bytes = (size_t)tile_rows * row_bytes;
    int n_mine = 0, n_full = 0;
    for (int t = rank; t < n_tiles; t += world) { n_mine++; if ((t + 1) * tile_rows <= height) n_full++; }
    if (n_mine == 0) return hipSuccess;
    if (!r.direct) {
        size_t off = 0;
        for (int t = rank; t < n_tiles; t += world) {
            const size_t bytes = (size_t)std::min(tile_rows, height - t * tile_rows) * row_bytes;
            const hipError_t e = copy_to_root(frame + (size_t)t * tile_bytes, root, rows + off, r.device, bytes, stream);
            if (e != hipSuccess) return e;
            off += bytes;
        }
        return hipSuccess;
    }
    if (n_full > 0) {
        const hipError_t e = hipMemcpy2DAsync(frame + (size_t)rank * tile_bytes, (size_t)world * tile_bytes, rows, tile_bytes, tile_bytes,
                                              (size_t)n_full, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
    }
    if (n_mine > n_full) {                                // the short last tile of the frame is this rank's
        const int t = rank + n_full * world;
        const size_t bytes = (size_t)(height - t * tile_rows) * row_bytes;
        return hipMemcpyAsync(frame + (size_t)t * tile_bytes, rows + (size_t)n_full * tile_bytes, bytes, hipMemcpyDeviceToDevice, stream);
    }
    return hipSuccess;
}

// device buffer of the current device that only ever grows
template <typename T>
int grow(T** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return RTX_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const hipError_t e = hipMalloc((void**)p, need * sizeof(T));
    if (e != hipSuccess) { *p = nullptr; rtx_set_error("allocation of %zu bytes failed: %s", need * sizeof(T), hipGetErrorString(e)); return RTX_E_NOMEM; }
    *cap = need;
    return RTX_OK;
}

// One frame: every rank renders its rows and copies them, tile by tile, into the frame on the first GPU.
int render_into(rtx_multi* m, const rtx_camera* cam, const rtx_render_params* p, float* d_frame, int32_t* d_ids, rtx_stats* stats) {
    const int world = (int)m->ranks.size();
    const int tile_rows = p->tile_rows > 0 ? p->tile_rows : 8;
    const int root = m->ranks[0].device;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = run_all(m, [&](int k) {
        return std::function<int()>([=]() -> int {
            Rank& r = m->ranks[k];
            rtx_render_params q = *p;
            q.rank = k;
            q.world = world;
            q.tile_rows = tile_rows;
            int32_t n_rows = 0;
            int rc = rtx_local_rows(&q, &n_rows, nullptr);
            if (rc) return rc;
            const size_t W = (size_t)q.width;
            if ((rc = grow(&r.rows, &r.rows_cap, std::max<size_t>((size_t)n_rows * W * 3, 1)))) return rc;
            if (d_ids && (rc = grow(&r.row_ids, &r.ids_cap, std::max<size_t>((size_t)n_rows * W, 1)))) return rc;
            memset(&r.stats, 0, sizeof r.stats);
            if (n_rows == 0) return RTX_OK;
            if ((rc = rtx_render_device(r.ctx, cam, &q, r.rows, d_ids ? r.row_ids : nullptr, r.stream, stats ? &r.stats : nullptr))) return rc;
            // local rows are this rank's tiles one after the other: tile j of the rank is image tile k + j * world
            const int n_tiles = (q.height + tile_rows - 1) / tile_rows;
            hipError_t ce = copy_tiles_to_root((char*)d_frame, root, (const char*)r.rows, r, k, world, n_tiles, tile_rows, q.height,
                                               W * 3 * sizeof(float), r.stream);
            if (ce == hipSuccess && d_ids)
                ce = copy_tiles_to_root((char*)d_ids, root, (const char*)r.row_ids, r, k, world, n_tiles, tile_rows, q.height,
                                        W * sizeof(int32_t), r.stream);
            if (ce != hipSuccess) return hip_fail("copy of a rank's row tiles to the first GPU", ce);
            const hipError_t e = hipStreamSynchronize(r.stream);
            if (e != hipSuccess) return hip_fail("hipStreamSynchronize", e);
            return rtx_frame_status(r.ctx);
        });
    });
    if (rc) return rc;
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (const Rank& r : m->ranks) {
            stats->primary_rays += r.stats.primary_rays;
            stats->shadow_rays += r.stats.shadow_rays;
            stats->reflection_rays += r.stats.reflection_rays;
            stats->traced_primary_rays += r.stats.traced_primary_rays;
            stats->culled_pixels += r.stats.culled_pixels;
            for (int l = 0; l < 16; l++) stats->hits_per_level[l] += r.stats.hits_per_level[l];
            stats->trace_ms = std::max(stats->trace_ms, r.stats.trace_ms);      // slowest rank's kernel
            stats->trace_launches = 1;
        }
        stats->frame_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RTX_OK;
}

}  // namespace

extern "C" {

int rtx_multi_create(rtx_multi** out, const int32_t* device_ids, int32_t n_devices) {
    if (!out || !device_ids || n_devices < 1 || n_devices > 64) { rtx_set_error("rtx_multi_create: bad argument (1..64 devices)"); return RTX_E_INVALID; }
    *out = nullptr;
    DeviceKeeper keep;
    rtx_multi* m = new (std::nothrow) rtx_multi;
    if (!m) return RTX_E_NOMEM;
    m->ranks.resize((size_t)n_devices);
    int rc = RTX_OK;
    for (int k = 0; k < n_devices && rc == RTX_OK; k++) {
        Rank& r = m->ranks[k];
        r.device = device_ids[k];
        rc = rtx_create(&r.ctx, r.device);
        if (rc) break;
        if (hipSetDevice(r.device) != hipSuccess || hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking) != hipSuccess) {
            rtx_set_error("rtx_multi_create: stream creation on device %d failed", r.device);
            rc = RTX_E_DEVICE;
            break;
        }
        // direct copies into the first GPU's memory where the link allows; otherwise the runtime stages them
        if (r.device != device_ids[0]) {
            int can = 0;
            r.direct = false;
            if (hipDeviceCanAccessPeer(&can, r.device, device_ids[0]) == hipSuccess && can) {
                const hipError_t pe = hipDeviceEnablePeerAccess(device_ids[0], 0);
                r.direct = pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled;
            }
            (void)hipGetLastError();      // "already enabled" is not an error here
        }
    }
    if (rc) { rtx_multi_destroy(m); return rc; }
    for (Rank& r : m->ranks) r.thread = std::thread(worker, m, &r);
    *out = m;
    return RTX_OK;
}

void rtx_multi_destroy(rtx_multi* m) {
    if (!m) return;
    DeviceKeeper keep;
    {
        std::lock_guard<std::mutex> lock(m->mu);
        for (Rank& r : m->ranks) r.quit = true;
    }
    m->cv_job.notify_all();
    for (Rank& r : m->ranks)
        if (r.thread.joinable()) r.thread.join();
    for (Rank& r : m->ranks) {
        if (!r.ctx) continue;                                     // creation stopped before this rank
        if (hipSetDevice(r.device) == hipSuccess) {
            if (r.stream) { (void)hipStreamSynchronize(r.stream); (void)hipStreamDestroy(r.stream); }
            if (r.rows) (void)hipFree(r.rows);
            if (r.row_ids) (void)hipFree(r.row_ids);
        }
        if (r.ctx) rtx_destroy(r.ctx);
    }
    if (!m->ranks.empty() && m->ranks[0].ctx && hipSetDevice(m->ranks[0].device) == hipSuccess) {
        if (m->d_frame) (void)hipFree(m->d_frame);
        if (m->d_frame_ids) (void)hipFree(m->d_frame_ids);
        if (m->d_frame8) (void)hipFree(m->d_frame8);
    }
    (void)hipGetLastError();                                      // nothing of the teardown may stick to the thread
    delete m;
}

int rtx_multi_device_count(const rtx_multi* m) { return m ? (int)m->ranks.size() : 0; }

int rtx_multi_set_scene(rtx_multi* m, const rtx_scene_view* view) {
    if (!m || !view) { rtx_set_error("rtx_multi_set_scene: null argument"); return RTX_E_INVALID; }
    return run_all(m, [&](int k) { return std::function<int()>([=]() { return rtx_set_scene(m->ranks[k].ctx, view); }); });
}

int rtx_multi_build_grid(rtx_multi* m, int32_t rx, int32_t ry, int32_t rz, int32_t sub, uint32_t flags) {
    if (!m) { rtx_set_error("rtx_multi_build_grid: null argument"); return RTX_E_INVALID; }
    // every GPU builds its own copy: the build is deterministic, so the copies are identical
    return run_all(m, [&](int k) { return std::function<int()>([=]() { return rtx_build_grid(m->ranks[k].ctx, rx, ry, rz, sub, flags); }); });
}

int rtx_multi_get_grid_info(const rtx_multi* m, rtx_grid_info* out) {
    if (!m || !out) { rtx_set_error("rtx_multi_get_grid_info: null argument"); return RTX_E_INVALID; }
    return rtx_get_grid_info(m->ranks[0].ctx, out);
}

int rtx_multi_render_device(rtx_multi* m, const rtx_camera* cam, const rtx_render_params* p, float* d_rgb, int32_t* d_prim_id, rtx_stats* stats) {
    if (!m || !cam || !p || !d_rgb) { rtx_set_error("rtx_multi_render_device: null argument"); return RTX_E_INVALID; }
    if (p->width <= 0 || p->height <= 0) { rtx_set_error("rtx_multi_render_device: width/height must be positive"); return RTX_E_INVALID; }
    return render_into(m, cam, p, d_rgb, d_prim_id, stats);
}

// the assembled frame on the first GPU, as floats or (quantised there) as bytes, copied to the caller's host memory
static int render_to_host(rtx_multi* m, const rtx_camera* cam, const rtx_render_params* p, float* rgb_out, uint8_t* rgb8_out, int32_t order,
                          int32_t* prim_id_out, rtx_stats* stats) {
    if (p->width <= 0 || p->height <= 0) { rtx_set_error("rtx_multi_render: width/height must be positive"); return RTX_E_INVALID; }
    const size_t pixels = (size_t)p->width * (size_t)p->height;
    const int root = m->ranks[0].device;
    DeviceKeeper keep;
    if (hipSetDevice(root) != hipSuccess) { rtx_set_error("hipSetDevice(%d) failed", root); return RTX_E_DEVICE; }
    int rc;
    if ((rc = grow(&m->d_frame, &m->frame_cap, pixels * 3))) return rc;
    if (prim_id_out && (rc = grow(&m->d_frame_ids, &m->frame_ids_cap, pixels))) return rc;
    if (rgb8_out && (rc = grow(&m->d_frame8, &m->frame8_cap, pixels * 3 + 16))) return rc;
    if ((rc = render_into(m, cam, p, m->d_frame, prim_id_out ? m->d_frame_ids : nullptr, stats))) return rc;
    if (hipSetDevice(root) != hipSuccess) { rtx_set_error("hipSetDevice(%d) failed", root); return RTX_E_DEVICE; }
    hipError_t e;
    if (rgb8_out) {
        if ((rc = rtx_quantize_device(m->ranks[0].ctx, m->d_frame, (int64_t)pixels, m->d_frame8, order, m->ranks[0].stream))) return rc;
        e = hipMemcpyAsync(rgb8_out, m->d_frame8, pixels * 3, hipMemcpyDeviceToHost, m->ranks[0].stream);
        if (e == hipSuccess) e = hipStreamSynchronize(m->ranks[0].stream);
        if (e != hipSuccess) return hip_fail("copy of the 8-bit frame to the host", e);
    } else {
        e = hipMemcpy(rgb_out, m->d_frame, pixels * 3 * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return hip_fail("copy of the frame to the host", e);
    }
    if (prim_id_out) {
        e = hipMemcpy(prim_id_out, m->d_frame_ids, pixels * sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return hip_fail("copy of the id map to the host", e);
    }
    return RTX_OK;
}

int rtx_multi_render(rtx_multi* m, const rtx_camera* cam, const rtx_render_params* p, float* rgb_out, int32_t* prim_id_out, rtx_stats* stats) {
    if (!m || !cam || !p || !rgb_out) { rtx_set_error("rtx_multi_render: null argument"); return RTX_E_INVALID; }
    return render_to_host(m, cam, p, rgb_out, nullptr, RTX_ORDER_RGB, prim_id_out, stats);
}

int rtx_multi_render_u8(rtx_multi* m, const rtx_camera* cam, const rtx_render_params* p, uint8_t* rgb8_out, int32_t order, int32_t* prim_id_out,
                        rtx_stats* stats) {
    if (!m || !cam || !p || !rgb8_out) { rtx_set_error("rtx_multi_render_u8: null argument"); return RTX_E_INVALID; }
    if (order != RTX_ORDER_RGB && order != RTX_ORDER_BGR) { rtx_set_error("rtx_multi_render_u8: unknown channel order"); return RTX_E_INVALID; }
    return render_to_host(m, cam, p, nullptr, rgb8_out, order, prim_id_out, stats);
}

}  // extern "C"
