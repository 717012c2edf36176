// Default pipeline: the persistent frame kernel and the kernels that choose and order the screen tiles it traces.
// Part of the device layer of librtx; included by rtx_device.hip only (one translation unit).
#pragma once

// ---------------------------------------------------------------------------------------------------
// persistent frame kernel: one launch per frame, every wave keeps a pool of 64 paths alive
//
// The level-synchronous kernels above wait at every level for the slowest ray of that level.  Here a wave
// never waits for another wave: each lane carries one pixel's path through its states (nearest-hit ray of
// level d -> one shadow ray per light -> mirror ray of level d+1 ...), all 64 lanes share the candidate tests
// of whatever rays they hold (wave_round), and a lane whose path has ended draws the next pixel from a
// global cursor.  A path's per-level terms L_d live in a private strip of global memory until the path ends
// and is folded from its deepest level up (the return path of trace(), src/main.cpp:99-102).
// Results are bit-identical to the level-synchronous kernels: every ray and every term is computed by the
// same code, only the schedule differs.
// ---------------------------------------------------------------------------------------------------
// Two register budgets are compiled (WAVES = resident waves per SIMD the allocation is held to):
//   4 waves (128 registers, a few spills in the state update, 39.7 KB of LDS per workgroup so that four fit a CU):
//     more waves to hide the latencies of a round behind; the faster one when the frame holds many pixels per lane;
//   3 waves (168 registers, 43.8 KB, record reads issued one batch ahead): shorter time per ray; the faster one when
//     a frame is a few chains of dependent rays per lane (small images, a rank's share of a frame).
// The host picks by the number of pixels of the traced window per resident lane.
#ifndef RT_TAIL_SHARE
#define RT_TAIL_SHARE 1
#endif
#ifndef RT_TAIL_MIN_LANES
#define RT_TAIL_MIN_LANES 32
#endif
template <int WAVES>
struct FrameBudget {
    static constexpr int lds_levels = WAVES >= 4 ? 5 : 6;   // path levels whose terms live in LDS (deeper: global)
};

// Layout of the first arguments of k_frame and k_frame_gen in the kernel's argument block.  Both kernels read what only
// their state updates need through this struct (scalar loads from the argument block), so DevScene, DevGrid, DevSparse
// and DevFrame must stay their four leading parameters, in this order.  By-value kernel arguments are laid out one
// after the other, each at its own alignment -- the rule a struct's members follow -- which the assertions pin.
struct FrameKernelArgs { DevScene sc; DevGrid g; DevSparse sp; DevFrame f; };
constexpr size_t rtx_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static_assert(std::is_trivially_copyable<DevScene>::value && std::is_trivially_copyable<DevGrid>::value &&
              std::is_trivially_copyable<DevSparse>::value && std::is_trivially_copyable<DevFrame>::value,
              "kernel arguments are copied bytewise into the argument block");
static_assert(offsetof(FrameKernelArgs, sc) == 0, "argument block: the scene comes first");
static_assert(offsetof(FrameKernelArgs, g) == rtx_align_up(sizeof(DevScene), alignof(DevGrid)), "argument block: grid after scene");
static_assert(offsetof(FrameKernelArgs, sp) == rtx_align_up(offsetof(FrameKernelArgs, g) + sizeof(DevGrid), alignof(DevSparse)),
              "argument block: sparse grid after grid");
static_assert(offsetof(FrameKernelArgs, f) == rtx_align_up(offsetof(FrameKernelArgs, sp) + sizeof(DevSparse), alignof(DevFrame)),
              "argument block: frame after sparse grid");
static_assert(alignof(DevScene) <= 8 && alignof(DevGrid) <= 8 && alignof(DevSparse) <= 8 && alignof(DevFrame) <= 8,
              "argument block: the segment itself is only guaranteed 8-byte alignment for these");

template <int MODE, bool SINGLE_XF, bool COUNT, int PRIMS, int WAVES>
__global__ __launch_bounds__(RTX_BLOCK, WAVES) void k_frame(DevScene sc, DevGrid g, DevSparse sp, DevFrame f, float4* path_stack, int global_levels) {
    constexpr int RTX_LDS_LEVELS = FrameBudget<WAVES>::lds_levels;
    __shared__ WaveLds lds[RTX_BLOCK / 64];
    __shared__ float4 stack_lds[RTX_BLOCK / 64][RTX_LDS_LEVELS][64];
    __shared__ int hit_count[RTX_BLOCK / 64][RTX_MAX_LEVELS];
    __shared__ int outstanding[RTX_BLOCK / 64][64];   // per path lane: helpers that have not reported yet
    __shared__ unsigned occ_lds[RT_OCC_LDS_WORDS];
    const Occupancy occ = stage_occupancy(g, occ_lds);
    const XfRegs xf = {};                               // (the rounds need none: rays are transformed when they begin)
    const int wave = (int)(threadIdx.x >> 6), lane = wave_lane();
    WaveLds& l = lds[wave];
    if (lane < RTX_MAX_LEVELS) hit_count[wave][lane] = 0;
    // per-level terms of this lane's path: the first RTX_LDS_LEVELS levels (where nearly all hits are) in LDS, deeper
    // ones in a private strip of global memory (global_levels entries per lane, entry 0 = level RTX_LDS_LEVELS)
    float4* stack_g = path_stack + ((size_t)(blockIdx.x * (RTX_BLOCK / 64) + wave) * 64 + lane) * global_levels;
    WorkCounters wc = {};

    // path state of this lane; the ray in flight lives in the walker (w.O, w.D, w.t_limit)
    //   pix >= 0   the lane carries the path of that pixel        pix == -1   the lane is idle
    //   pix <= -2  the lane traces the shadow rays of one hit for the path on lane -2 - pix (see "helpers" below)
    //   phase 0: nearest-hit ray in flight; 1 + i: shadow ray of light i; -1 / -2: the path has ended at level /
    //   level - 1 and waits for its helpers before it folds
    int pix = -1, level = 0, phase = 0, material = 0;
    int tslot = 0;                                      // 8x8 block of the pixel (index into the cost map)
    f3 rayDir = mk(0, 0, 1), P = mk(0, 0, 0), N = mk(0, 0, 1), acc = mk(0, 0, 0);
    Walker<MODE> w;
    LaneSearch r;
    w.init(g, mk(0, 0, 0), mk(0, 0, 1), FLT_MAX, false);
    r.found = false; r.best_t = FLT_MAX; r.best_prim = -1;
    bool exhausted = false;
    const int n_items = f.tile_list ? 64 * __builtin_amdgcn_readfirstlane(f.counters[CNT_TILES]) : f.n_items0;
    // Helpers.  The shadow rays of a hit and the mirror ray that leaves it do not depend on each other, but one lane
    // traces them one after the other.  A lane without a path -- idle, or freed in the same state update -- takes over
    // the shadow rays of a fresh hit (hit record handed over through LDS), the path's own lane goes on with the mirror
    // ray at once, and the chain of dependent rays of a depth-5 path shrinks from 12 to 7.  The helper writes the
    // level's term into the path's slot; a path folds only when all its helpers have reported (outstanding).
    // f.helpers = 1: only once the wave can draw no new pixels (a frame bound by throughput gains nothing from shorter
    // chains before that, and a path that waits for its helper idles a lane); 2: from the start, lanes that become free
    // serve fresh hits before they draw new pixels (a frame that is a few chains of dependent rays per lane).
    const bool helpers = f.helpers != 0;
    outstanding[wave][lane] = 0;
    int path_lanes = f.path_lanes;                      // paths this wave may carry at a time (wave-uniform)
    long long rounds_left = f.round_limit;              // bounds the loop whatever happens (the host derives it from the frame)
    int n_drawn = 0;                                    // pixels this lane drew
    // instrumented build: when the wave started, ran out of new pixels and ended (100 MHz clock), what it did
    unsigned long long log_start = 0, log_exhausted = 0;
    int log_rounds = 0, log_updates = 0, log_pixels = 0, log_deepest = 0, log_tasks = 0;
    if (COUNT) log_start = wall_clock64();
    fill_untraced_blocks(f, (int)blockIdx.x * (RTX_BLOCK / 64) + wave, (int)gridDim.x * (RTX_BLOCK / 64));

    for (;;) {
        const bool alive = pix != -1;
        const bool walking = alive && w.state != 2;
        const bool finished = alive && w.state == 2 && phase >= 0;
        const bool waiting = alive && w.state == 2 && phase < 0;
        // (a wave keeps at most f.path_lanes paths: lanes beyond that stay free for the shadow rays of fresh hits)
        const int n_paths = __popcll(__ballot(pix >= 0));
        const int n_ready = __popcll(__ballot(finished)) + (exhausted ? 0 : max(0, min(__popcll(__ballot(!alive)), path_lanes - n_paths)));
        // once the cursor is exhausted the wave drains: with few paths left, waiting for a full batch of finished
        // rays would chain every path to the slowest ray of each generation
        const int n_alive = __popcll(__ballot(alive));
        const int threshold = exhausted ? min(f.ready_threshold, max(1, (n_alive * f.drain_num) >> 3)) : f.ready_threshold;
        if (!__any(walking) || n_ready >= threshold) {
            // What only the update reads of the frame, the scene and the grid is fetched from the kernel's argument
            // block when an update runs (scalar loads) instead of living in scalar registers through the rounds: the
            // kernel needs more scalar registers than there are, and what does not fit is kept in lanes of vector
            // registers and moved back with vector instructions at every use (150 of them per update before this).
            const __attribute__((address_space(4))) char* arg_block =
                (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(arg_block));         // (keeps the loads inside the update)
            const FrameKernelArgs& args = *(const FrameKernelArgs*)arg_block;
            const DevFrame& fu = args.f;
            const DevScene& su = args.sc;
            const DevGrid& gu = args.g;
            if (COUNT) { wc.w_updates++; log_updates++; }
            const unsigned long long c0 = COUNT ? __builtin_readcyclecounter() : 0ull;
            bool begin = false;
            f3 O = w.O, D = w.D;
            float tl = w.t_limit;
            const bool worker = pix <= -2;
            bool path_end = false, lit = false, want_helper = false;
            int top = -1, next_light = -1;
            if (waiting && atomicAdd(&outstanding[wave][lane], 0) == 0) {       // every helper has reported
                path_end = true;
                top = phase == -1 ? level : level - 1;
            }
            if (finished) {
                if (phase == 0) {
                    if (!w.hit) {                              // src/main.cpp:104-106
                        path_end = true;
                        top = level - 1;
                        if (level == 0 && fu.ids) fu.ids[pix] = -1;
                    } else {
                        if (COUNT) wc.nearest_hits++;
                        const int prim = r.best_prim;
                        finish_hit<SINGLE_XF, PRIMS>(su, prim, O, D, r.best_t, P, N, material);
                        atomicAdd(&hit_count[wave][level], 1);
                        if (level == 0 && fu.ids) fu.ids[pix] = prim;
                        acc = mk(0, 0, 0);
                        // (helpers report through LDS: the few hits of the deep levels keep their shadow rays)
                        if (su.n_lights > 0) { next_light = 0; want_helper = helpers && (exhausted || f.helpers > 1 || path_lanes < 64) && level < RTX_LDS_LEVELS; } else lit = true;
                    }
                } else {                                        // src/main.cpp:57-75
                    const int li = phase - 1;
                    if (!w.hit) {
                        const float* lt = su.lights + 12 * li;
                        const float* m = su.mats + 16 * material;
                        if (COUNT) wc.light_evals++;
                        const f3 c = light_term(rayDir, D, N, mk(m[3], m[4], m[5]), mk(m[6], m[7], m[8]), m[12], mk(lt[4], lt[5], lt[6]));
                        if ((int)lt[0] == RTX_LIGHT_POINT) {
                            const float att = lt[7] + lt[8] * tl + lt[9] * tl * tl;
                            acc = acc + c / att;
                        } else {
                            acc = acc + c;
                        }
                    }
                    if (li + 1 < su.n_lights) next_light = li + 1; else lit = true;
                }
            }
            const auto mirror_ray = [&]() {                     // src/main.cpp:85-94
                const f3 V = normalize3(-rayDir);
                const f3 R = (2.0f * dot3(V, N) * N) - V;
                O = P + fu.bias * R;
                rayDir = R;
                D = normalize3(R);
                tl = FLT_MAX;
                level++;
                phase = 0;
                begin = true;
            };
            // a level whose lights are all evaluated: its term goes into the path's slot (src/main.cpp:97-102)
            if (lit) {
                const float* m = su.mats + 16 * material;
                const f3 Lsum = mk(m[0], m[1], m[2]) + mk(m[9], m[10], m[11]) + clamp_rgb(acc);
                const float4 rec = make_float4(Lsum.x, Lsum.y, Lsum.z, as_float(material));
                if (worker) {                                   // a helper reports to the path it worked for
                    const int owner = -2 - pix;
                    stack_lds[wave][level][owner] = rec;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    atomicSub(&outstanding[wave][owner], 1);
                    pix = -1;
                } else {
                    if (level < RTX_LDS_LEVELS) stack_lds[wave][level][lane] = rec; else stack_g[level - RTX_LDS_LEVELS] = rec;
                    if (level < fu.max_depth) mirror_ray();
                    else { path_end = true; top = level; }
                }
            }
            // a path that ends folds its levels from the deepest up -- once every helper of it has reported
            if (path_end && helpers && atomicAdd(&outstanding[wave][lane], 0) != 0) {
                phase = top == level ? -1 : -2;
                w.state = 2;
                path_end = false;
            }
            if (path_end) {
                f3 C = mk(0, 0, 0);
                const auto fold = [&](const float4 s4) {
                    const float* m = su.mats + 16 * as_int(s4.w);
                    C = clamp_rgb(mk(s4.x, s4.y, s4.z) + mk(m[6], m[7], m[8]) * C);
                };
                // one loop per address space: a pointer selected inside a single loop would turn every read into a
                // flat load
                for (int d = top; d >= RTX_LDS_LEVELS; d--) fold(stack_g[d - RTX_LDS_LEVELS]);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const __attribute__((address_space(3))) float* sl =
                    (const __attribute__((address_space(3))) float*)&stack_lds[wave][0][lane];
                for (int d = min(top, RTX_LDS_LEVELS - 1); d >= 0; d--)
                    fold(make_float4(sl[256 * d], sl[256 * d + 1], sl[256 * d + 2], sl[256 * d + 3]));
                fu.rgb[3 * pix] = C.x;
                fu.rgb[3 * pix + 1] = C.y;
                fu.rgb[3 * pix + 2] = C.z;
                record_tile_depth(fu, tslot, top + 2);
                if (COUNT) log_deepest = max(log_deepest, top + 1);
                pix = -1;
                phase = 0;
            }
            // hand fresh hits to lanes without a path (idle before, or freed just now): the k-th lane with a fresh hit is
            // served by the k-th such lane, which traces the hit's shadow rays while the path's own lane goes on with the
            // mirror ray at once
            bool served = false;
            if (helpers && (exhausted || f.helpers > 1 || path_lanes < 64)) {
                const bool free_lane = pix == -1;
                const unsigned long long req_mask = __ballot(want_helper), idle_mask = __ballot(free_lane);
                const int n_match = min(__popcll(req_mask), __popcll(idle_mask));
                if (n_match > 0) {
                    const unsigned long long below = (1ull << lane) - 1ull;
                    const int req_rank = __popcll(req_mask & below), idle_rank = __popcll(idle_mask & below);
                    if (want_helper && req_rank < n_match) {
                        // the hit record goes through the requester's own ray slots (its ray has ended) and the
                        // candidate ring (idle between rounds); owner_at tells the k-th helper where to look
                        l.ray_a[lane] = make_float4(P.x, P.y, P.z, as_float(material));
                        l.ray_b[lane] = make_float4(N.x, N.y, N.z, as_float(level));
                        l.cand_n[lane] = make_float4(rayDir.x, rayDir.y, rayDir.z, 0.0f);
                        l.owner_at[req_rank] = lane;
                        atomicAdd(&outstanding[wave][lane], 1);
                        next_light = -1;
                        served = true;
                    }
                    wave_lds_sync();
                    if (free_lane && idle_rank < n_match) {
                        const int from = l.owner_at[idle_rank];
                        const float4 a = l.ray_a[from], b = l.ray_b[from], c = l.cand_n[from];
                        P = mk(a.x, a.y, a.z); material = as_int(a.w);
                        N = mk(b.x, b.y, b.z); level = as_int(b.w);
                        rayDir = mk(c.x, c.y, c.z);
                        pix = -2 - from;
                        if (COUNT) log_tasks++;
                        acc = mk(0, 0, 0);
                        next_light = 0;
                    }
                    wave_lds_sync();
                }
            }
            if (served) {
                if (level < fu.max_depth) mirror_ray();
                else { phase = -1; w.state = 2; }                // no mirror ray at the last level: wait for the helper, then fold
            }
            if (next_light >= 0) {                              // src/main.cpp:53-57,68-70
                const float* lt = su.lights + 12 * next_light;
                const f3 lpos = mk(lt[1], lt[2], lt[3]);
                const bool point = (int)lt[0] == RTX_LIGHT_POINT;
                const f3 L = point ? normalize3(lpos - P) : normalize3(lpos);
                O = P + fu.bias * L;
                D = L;
                tl = point ? dot3(lpos - P, L) : FLT_MAX;
                phase = 1 + next_light;
                begin = true;
            }
            // lanes without a path draw the next pixels
            const bool idle = pix == -1;
            const unsigned long long idle_mask = __ballot(idle);
            const int room = path_lanes - __popcll(__ballot(pix >= 0));
            if (!exhausted && idle_mask != 0 && room > 0) {
                const int n_idle = min(__popcll(idle_mask), room);
                int base = 0;
                if (lane == 0) base = atomicAdd(fu.counters + CNT_TRACE_CUR, n_idle);
                base = __builtin_amdgcn_readfirstlane(base);
                const int idle_rank = __popcll(idle_mask & ((1ull << lane) - 1ull));
                if (idle && idle_rank < n_idle) {
                    const int i = base + idle_rank;
                    if (i < n_items) {
                        const int blk = i >> 6, in = i & 63;
                        int lr, x;
                        if (fu.tile_list) {
                            const int t = fu.tile_list[blk];
                            lr = 8 * (t >> 16) + (in >> 3);
                            x = 8 * (t & 0xffff) + (in & 7);
                            tslot = (t >> 16) * fu.tbx + (t & 0xffff);
                        } else {
                            lr = fu.lr0 + 8 * (blk / fu.nbx) + (in >> 3);
                            x = 8 * (fu.bx0 + blk % fu.nbx) + (in & 7);
                        }
                        if (lr < fu.n_local_rows && x < fu.W) {
                            const int tile = lr / fu.tile_rows;
                            const int y = (fu.rank + tile * fu.world) * fu.tile_rows + (lr - tile * fu.tile_rows);
                            pix = lr * fu.W + x;
                            n_drawn++;
                            if (COUNT) log_pixels++;
                            O = fu.cam.eye;
                            rayDir = camera_ray(fu.cam, x, y);
                            D = normalize3(rayDir);             // src/main.cpp:82
                            tl = FLT_MAX;
                            level = 0;
                            phase = 0;
                            begin = true;
                        }
                    }
                }
                if (base + n_idle >= n_items) {
                    if (COUNT && !exhausted) log_exhausted = wall_clock64();
                    exhausted = true;
                }
                // The last pixels of the frame are shared out: once fewer are left than the launch has lanes, a wave
                // only keeps its fair share of them as paths, so that the frame does not end on the few waves that
                // happened to draw last -- and their other lanes are free for the shadow rays of those paths' hits.
                if (RT_TAIL_SHARE) {
                    const int left = max(n_items - (base + n_idle), 0);
                    const int waves = (int)gridDim.x * (RTX_BLOCK / 64);
                    path_lanes = min(f.path_lanes, max(left / waves + 1, RT_TAIL_MIN_LANES));
                }
            }
            if (begin) ray_begin<MODE, SINGLE_XF, COUNT>(su, gu, load_xf_regs(su.xinv, SINGLE_XF), l, w, r, O, D, tl, wc);
            if (COUNT) wc.c_update += __builtin_readcyclecounter() - c0;
            if (exhausted && !__any(pix != -1)) break;
        }
        if (COUNT) log_rounds++;
        if (__any(pix != -1 && w.state != 2)) wave_round<MODE, SINGLE_XF, COUNT, PRIMS, (WAVES < 4)>(sc, g, sp, occ, xf, l, w, r, phase != 0, f.walk_budget, wc);
        if (--rounds_left < 0) {
            // the frame is abandoned: pixels still in flight keep the fill colour, and the host is told
            if (lane == 0) { f.counters[CNT_ABORT] = 1; *f.abort_host = 1; }
            break;
        }
    }
    if (lane < RTX_MAX_LEVELS && hit_count[wave][lane]) atomicAdd(f.counters + CNT_HITS + lane, hit_count[wave][lane]);
    for (int off = 32; off > 0; off >>= 1) n_drawn += __shfl_down(n_drawn, off);
    if (lane == 0 && n_drawn) atomicAdd(f.counters + CNT_PIXELS, n_drawn);
    if (COUNT) {
        flush_work(wc, f.work);
        int pixels = log_pixels, deepest = log_deepest, tasks = log_tasks;
        for (int off = 32; off > 0; off >>= 1) {
            pixels += __shfl_down(pixels, off);
            tasks += __shfl_down(tasks, off);
            deepest = max(deepest, __shfl_down(deepest, off));
        }
        if (lane == 0) {
            unsigned long long* log = f.work + 24 + 8 * (size_t)(blockIdx.x * (RTX_BLOCK / 64) + wave);
            log[0] = log_start; log[1] = log_exhausted; log[2] = wall_clock64(); log[3] = (unsigned long long)log_rounds;
            log[4] = (unsigned long long)log_updates; log[5] = (unsigned long long)pixels; log[6] = (unsigned long long)deepest;
            log[7] = (unsigned long long)tasks;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Start of a frame of the persistent kernels in ONE launch (k_frame_prep): zero the frame's counters, find the
// screen tiles that can see geometry, put them into scheduling classes and write the list the frame kernel draws from.
//
// A camera ray can only hit a primitive inside an occupied fine cell (every point of a primitive lies in a fine cell
// that references it), so an 8x8 pixel tile outside the projection of every occupied fine cell returns the miss colour
// for all its pixels (src/main.cpp:104-106) and is left to the frame kernel's fill.  Three phases:
//   mark      projects the eight corners of every occupied fine cell (grown by the binning margin) and marks the tiles
//             of the bounding rectangle grown by two pixels; a cell with a corner beside or behind the eye, or one
//             covering more than 1024 tiles, raises the "every tile" flag instead.  A marked tile holds the frame's
//             stamp in the tile map; stamps differ from frame to frame, so the map is never cleared.  One chunk of
//             this phase zeroes the frame's counters.
//   classify  one lane per 8x8 block of this rank's traced window (rows are local rows; a block's eight rows may come
//             from different global tiles when the rank split is not in multiples of eight rows): blocks that can see
//             geometry get one of eight classes by the deepest path the block -- or a block within hint_radius of it --
//             produced in the previous frame of the same geometry (0 = unknown), and the classes are counted;
//   list      writes the list with the deepest class first, so that the long dependent chains of a frame start early
//             and the frame does not end on them.
// Phases follow each other inside the launch without a grid-wide barrier of resident workgroups: the work of a phase
// is cut into chunks of 256 items which the launch's workgroups draw from a ticket counter, a workgroup that finds a
// phase's tickets used up waits until that phase's chunks are all reported done and moves on.  Whoever holds a
// ticket is running, so the wait always ends -- also when the GPU is full of other launches' workgroups and only a
// few of this launch's get to run (any one of them can do all the work).  The ticket and done counters are never
// reset either: the host passes the values they had before the launch.
// What one phase hands to the next crosses compute units and their caches inside the launch.  No cache-wide fence is
// used for that (a write-back of a whole L2 per chunk is dear, and dearer still beside frame kernels that are filling
// their frames): every word a later phase reads -- tile-map stamps, the zeroed counters, the class totals -- is
// written by a store or an atomic that goes through to memory and read by a load that bypasses the reader's caches,
// every storing wave waits for its stores before its workgroup reports a chunk, and a workgroup reads only after its
// polling lane has seen every chunk reported.  The class of a block is computed twice, by the phase that counts and
// by the phase that lists (cheap, and the same inputs), instead of being handed over.
// ---------------------------------------------------------------------------------------------------
// word that another compute unit of the same launch reads / wrote: write-through store, cache-bypassing load
__device__ __forceinline__ void store_through(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_through(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned load_through(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int load_through(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct PrepArgs {
    DevFrame f;
    DevGrid g;
    const int* occupied_cells;     // fine cell number of every occupied fine cell
    float pad;                     // margin the fine lists were binned with
    int n_occ;
    int tby;                       // 8-pixel row blocks of the whole image (f.tbx column blocks)
    unsigned* tile_map;            // per 8x8 tile of the image: == stamp if the tile may see geometry this frame
    unsigned stamp;
    int n_row_blocks;              // row blocks of this rank's traced window
    const int* prev_cost;          // the previous frame's per-block depths (entries stamped prev_stamp), or null
    int prev_stamp;                // (already shifted past the depth bits)
    signed char* cls;              // per block of the window: class, -1 = not traced
    int* tile_list;
    unsigned* sync;                // [0..2] tickets drawn per phase, [3..4] chunks done of the first two phases
    unsigned ticket_base[3], done_base[2];
    int n_chunks[3];
    unsigned long long* work;      // instrumented frames: the 24 work counters (zeroed here), else null
};

__device__ __forceinline__ void prep_mark(const PrepArgs& a, int o) {
    if (o >= a.n_occ) return;
    const DevGrid& g = a.g;
    const DevCamera& cam = a.f.cam;
    const int tbx = a.f.tbx, tby = a.tby;
    const int i = a.occupied_cells[o];                    // fine cell number: block * cells per block + local
    const int cell = i / g.cpb, local = i - cell * g.cpb;
    const int k = g.sub;
    const int cx = cell % g.rx, cy = (cell / g.rx) % g.ry, cz = cell / (g.rx * g.ry);
    const int lx = local % k, ly = (local / k) % k, lz = local / (k * k);
    const double pad = a.pad;
    const double lo[3] = {(double)g.lo.x + (double)(cx * k + lx) * g.fcell.x - pad, (double)g.lo.y + (double)(cy * k + ly) * g.fcell.y - pad,
                          (double)g.lo.z + (double)(cz * k + lz) * g.fcell.z - pad};
    const double hi[3] = {lo[0] + g.fcell.x + 2.0 * pad, lo[1] + g.fcell.y + 2.0 * pad, lo[2] + g.fcell.z + 2.0 * pad};
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    bool undefined = false;
    for (int c = 0; c < 8; c++) {
        const double dx = ((c & 1) ? hi[0] : lo[0]) - cam.eye.x, dy = ((c & 2) ? hi[1] : lo[1]) - cam.eye.y,
                     dz = ((c & 4) ? hi[2] : lo[2]) - cam.eye.z;
        const double depth = -(dx * cam.w.x + dy * cam.w.y + dz * cam.w.z);
        const double extent = fabs(dx) + fabs(dy) + fabs(dz);
        if (!(depth > 1e-3 * extent) || !(depth > 0)) { undefined = true; break; }
        const double u = (dx * cam.u.x + dy * cam.u.y + dz * cam.u.z) / depth;
        const double v = (dx * cam.v.x + dy * cam.v.y + dz * cam.v.z) / depth;
        const double sx = u / cam.tanx * cam.halfW + cam.halfW - 0.5;      // inverse of Camera::getRayAt (src/geometries.h:339-351)
        const double sy = cam.halfH - 0.5 - v / cam.tany * cam.halfH;
        if (!(sx == sx) || !(sy == sy)) { undefined = true; break; }
        xmin = fmin(xmin, sx); xmax = fmax(xmax, sx);
        ymin = fmin(ymin, sy); ymax = fmax(ymax, sy);
    }
    if (undefined) { store_through(a.f.counters + CNT_TILES_ALL, (int)a.stamp); return; }
    const double margin = 2.0;
    if (xmax + margin < 0.0 || ymax + margin < 0.0 || xmin - margin > 8.0 * tbx || ymin - margin > 8.0 * tby) return;   // off screen
    const int tx0 = (int)fmax(0.0, floor((xmin - margin) / 8.0)), tx1 = (int)fmin((double)(tbx - 1), floor((xmax + margin) / 8.0));
    const int ty0 = (int)fmax(0.0, floor((ymin - margin) / 8.0)), ty1 = (int)fmin((double)(tby - 1), floor((ymax + margin) / 8.0));
    if (tx1 < tx0 || ty1 < ty0) return;
    if ((long long)(tx1 - tx0 + 1) * (ty1 - ty0 + 1) > 1024) { store_through(a.f.counters + CNT_TILES_ALL, (int)a.stamp); return; }
    for (int ty = ty0; ty <= ty1; ty++)
        for (int tx = tx0; tx <= tx1; tx++) store_through(a.tile_map + ty * tbx + tx, a.stamp);
}

// Class of block i of the window: -1 = sees no geometry (not traced), else one of eight scheduling classes by the
// deepest path the block -- or a block within hint_radius of it: a camera that moved a little still finds its deep
// tiles -- produced in the previous frame (0 = unknown).
__device__ __forceinline__ int prep_class(const PrepArgs& a, int i) {
    const DevFrame& f = a.f;
    if (i >= a.n_row_blocks * f.nbx) return -1;
    const int rb = f.lr0 / 8 + i / f.nbx;                 // row block (of local rows) and column block
    const int bx = f.bx0 + i % f.nbx;
    bool take = load_through(f.counters + CNT_TILES_ALL) == (int)a.stamp;
    for (int r = 0; r < 8 && !take; r++) {
        const int lr = 8 * rb + r;
        if (lr >= f.n_local_rows) break;
        const int tile = lr / f.tile_rows;
        const int y = (f.rank + tile * f.world) * f.tile_rows + (lr - tile * f.tile_rows);
        take = load_through(a.tile_map + (y >> 3) * f.tbx + bx) == a.stamp;
    }
    if (!take) return -1;
    int c = 0;
    if (a.prev_cost) {
        const int nrb = (f.n_local_rows + 7) / 8, R = f.hint_radius;
        for (int dy = -R; dy <= R; dy++)
            for (int dx = -R; dx <= R; dx++) {
                const int y = rb + dy, x = bx + dx;
                if (y >= 0 && y < nrb && x >= 0 && x < f.tbx) {
                    const int v = a.prev_cost[y * f.tbx + x];
                    if ((v & ~((1 << RTX_COST_BITS) - 1)) == a.prev_stamp) c = max(c, v & ((1 << RTX_COST_BITS) - 1));
                }
            }
        c = min(c, 7);
    }
    return c;
}

__global__ __launch_bounds__(256) void k_frame_prep(PrepArgs a) {
    __shared__ unsigned s_ticket;
    __shared__ int cnt[8], base[8], rank[8];
    const int tid = (int)threadIdx.x;
    const DevFrame& f = a.f;
    // the next chunk of a phase, or a number >= the phase's chunks when they are used up (every workgroup draws exactly
    // one such ticket per phase, which is how the host knows where the counters stand after the launch)
    const auto draw = [&](int phase) -> int {
        __syncthreads();                                  // everyone is done with s_ticket and the chunk before
        if (tid == 0) s_ticket = atomicAdd(a.sync + phase, 1u) - a.ticket_base[phase];
        __syncthreads();
        return (int)min(s_ticket, 0x7fffffffu);
    };
    // a chunk is reported done when every wave's stores of it have gone through
    const auto chunk_done = [&](int phase) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) atomicAdd(a.sync + 3 + phase, 1u);
    };
    const auto wait_for = [&](int phase) {
        if (tid == 0) {
            while (load_through(a.sync + 3 + phase) - a.done_base[phase] < (unsigned)a.n_chunks[phase]) __builtin_amdgcn_s_sleep(8);
        }
        __syncthreads();
    };

    // ---- mark (chunk 0 zeroes the frame's counters instead) ----------------------------------------------------------
    for (;;) {
        const int t = draw(0);
        if (t >= a.n_chunks[0]) break;
        if (t == 0) {
            for (int i = tid; i < CNT_TOTAL; i += 256)
                if (i != CNT_TILES_ALL) store_through(f.counters + i, 0);
            if (a.work && tid < 24) a.work[tid] = 0ull;
        } else {
            prep_mark(a, (t - 1) * 256 + tid);
        }
        chunk_done(0);
    }
    wait_for(0);
    // ---- classify ----------------------------------------------------------------------------------------------------
    for (;;) {
        const int t = draw(1);
        if (t >= a.n_chunks[1]) break;
        if (tid < 8) cnt[tid] = 0;
        __syncthreads();
        const int c = prep_class(a, t * 256 + tid);
        if (c >= 0) atomicAdd(&cnt[c], 1);
        __syncthreads();
        if (tid < 8 && cnt[tid]) {
            atomicAdd(f.counters + CNT_CLASS + tid, cnt[tid]);
            atomicAdd(f.counters + CNT_TILES, cnt[tid]);
        }
        chunk_done(1);
    }
    wait_for(1);
    // ---- list --------------------------------------------------------------------------------------------------------
    for (;;) {
        const int t = draw(2);
        if (t >= a.n_chunks[2]) break;
        if (tid < 8) { cnt[tid] = 0; rank[tid] = 0; }
        __syncthreads();
        const int i = t * 256 + tid;
        const int c = prep_class(a, i);
        if (i < a.n_row_blocks * f.nbx) a.cls[i] = (signed char)c;          // (read by the frame kernel's fill)
        if (c >= 0) atomicAdd(&cnt[c], 1);
        __syncthreads();
        if (tid < 8) {
            int offset = 0;                                       // classes above this one come first in the list
            for (int j = tid + 1; j < 8; j++) offset += load_through(f.counters + CNT_CLASS + j);
            base[tid] = offset + (cnt[tid] ? atomicAdd(f.counters + CNT_CLASS + 8 + tid, cnt[tid]) : 0);
        }
        __syncthreads();
        if (c >= 0) a.tile_list[base[c] + atomicAdd(&rank[c], 1)] = ((f.lr0 / 8 + i / f.nbx) << 16) | (f.bx0 + i % f.nbx);
    }
}
