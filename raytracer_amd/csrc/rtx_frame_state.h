// Device-side description of a frame, work cursors, camera rays and the shading terms shared by all kernel families.
// Part of the device layer of librtx; included by rtx_device.hip only (one translation unit).
#pragma once

// ---------------------------------------------------------------------------------------------------
// device-side frame description
// ---------------------------------------------------------------------------------------------------
struct DevCamera {
    f3 eye, u, v, w;               // Camera::updateAxes (src/geometries.h:332-337), computed on the host
    float tanx, tany, halfW, halfH;
};

// counters[] layout (ints): [0,16) hits per level; [16,32) trace work cursor per level;
// [32,48) shade work cursor per level; [48,64) resolve work cursor per level
#define CNT_HITS 0
#define CNT_TRACE_CUR 16
#define CNT_SHADE_CUR 32
#define CNT_RESOLVE_CUR 48
#define CNT_TILES 64        // 8x8 pixel tiles the frame kernel traces (tile list)
#define CNT_TILES_ALL 65    // holds the frame's stamp when a fine cell's screen footprint was too large or undefined: every tile
                            // is traced (never cleared: a value other than this frame's stamp means "not set")
#define CNT_CLASS 66        // 8 counters: tiles per scheduling class; 74..81: fill cursors per class
#define CNT_ABORT 82        // set by a wave of the frame kernel that reached the round limit
#define CNT_PIXELS 83       // pixels whose camera ray the frame kernel walked
#define CNT_TOTAL 96

struct DevFrame {
    int W, H;                      // full image
    int n_local_rows;              // rows this rank renders
    int tile_rows, rank, world;
    int nbx;                       // 8-pixel column blocks per row inside the traced window
    int bx0;                       // first 8-pixel column block of the traced window
    int lr0;                       // first local row of the traced window (a multiple of 8)
    int n_items0;                  // level-0 work items (8x8 pixel blocks of the traced window * 64)
    const int* tile_list;          // persistent kernel, optional: the 8x8 blocks to trace, (row block << 16) | column block,
                                   // their number in counters[CNT_TILES]; null = every block of the window
    int* tile_cost;                // with tile_list: deepest path level seen per 8x8 block this frame (+2), indexed
                                   // row block * tbx + column block; feeds the order of the next frame's list.  An entry
                                   // is cost_stamp | depth: stamps grow from frame to frame, so a frame's entries beat
                                   // whatever older frames left there and the map is never cleared
    int cost_stamp;                // this frame's stamp, shifted past the depth bits (RTX_COST_BITS)
    const signed char* fill_cls;   // non-null: the frame kernel itself writes the miss colour (and id -1) over the 8x8
                                   // blocks it does not trace -- every block outside the traced window and every block
                                   // of the window whose class here is negative
    int fill_window_row_blocks;    // row blocks of the traced window (rows of fill_cls, nbx entries each)
    int tbx;                       // 8-pixel column blocks per image row
    int hint_radius;               // tiles around a tile whose recorded depths count for its class
    int max_depth;
    float bias;
    int walk_budget;               // persistent kernel: walk steps a lane may take per round
    int ready_threshold;           // persistent kernel: lanes with a finished ray that trigger a state update
    int drain_num;                 // ... after the cursor is exhausted: that many eighths of the lanes still alive
    int helpers;                   // persistent kernel: lanes without a path trace the shadow rays of fresh hits (1: once
                                   // the wave can draw no new pixels, 2: from the start)
    int path_lanes;                // persistent kernel: paths a wave carries at a time (64, or fewer to keep helper lanes)
    int endgame_pixels;            // generation pipeline: pixels left in the cursor from which a wave drains (0 = never)
    long long round_limit;         // persistent kernel: rounds after which a wave abandons the frame
    int* abort_host;               // ... and says so here (device address of a word of pinned host memory)
    DevCamera cam;
    float* rgb;                    // [n_local_rows][W][3]
    int* ids;                      // [n_local_rows][W] or null
    int* counters;
    unsigned long long* work;      // instrumented kernels only: 8 x u64 for the trace kernels, then 8 for shade
    // hit records of the level being processed (reused by every level)
    float4* hP;                    // position xyz, material index bits
    float4* hN;                    // normal xyz, primitive index bits
    float4* hD;                    // incoming ray direction as passed to trace() (not normalised)
    int* pix0;                     // level 0: local pixel index per hit record
    // persistent per level; the host fills in the pointers of the level a launch works on
    float4* hL;                    // this level: L_d xyz (later C_d), material index bits
    float4* hL_next;               // level d+1 (resolve only)
    int* hChild;                   // this level: hit record of the mirror ray in level d+1, -1 = none
    int* hChild_prev;              // level d-1 (trace only)
    // mirror rays, ping-pong by level parity
    float4* rO_in;                 // rays of this level (written by level d-1)
    float4* rD_in;
    float4* rO_out;                // rays for level d+1
    float4* rD_out;
};

__device__ __forceinline__ int wave_lane() { return (int)(threadIdx.x & 63); }

#define RTX_COST_BITS 5            // depth + 2 <= 17 fits; the stamp takes the bits above

// deepest level of a finished path, recorded for the 8x8 block of its pixel (scheduling hint of the next frame)
__device__ __forceinline__ void record_tile_depth(const DevFrame& f, int tslot, int depth_plus_2) {
    const int v = f.cost_stamp | depth_plus_2;
    if (f.tile_cost && f.tile_cost[tslot] < v) atomicMax(f.tile_cost + tslot, v);
}

// The miss colour (src/main.cpp:104-106 of the reference tree: no hit, colour 0) and the id -1 over every 8x8 block of
// this rank's rows that the frame kernel does not trace, written by the frame kernel's own waves before they start
// tracing.  The unit of work is one image row of a run of 64 consecutive blocks: 384 consecutive 16-byte chunks of
// colour (six store instructions of 64 lanes) and 128 of ids (two), every lane skipping the chunks of traced blocks and
// of pixels beyond the image width.  Units are numbered along the rows, row after row, and wave w of n takes units w,
// w + n, ...: at any time the waves of the launch write one contiguous band of the image.  (Eight rows of a run per wave
// -- a block run at a time -- had 2000 waves write 6 KB pieces 46 KB apart all over the frame at once: memory pages
// opened and closed for every piece, and a frame that is busy to its end lost 0.23 ms to it.)  One lane looks up
// whether each block of the run is traced.  The host only asks for this when the image width is a multiple of 4 pixels
// and both buffers are 16-byte aligned, which makes every chunk aligned.
#ifndef RT_FILL_NO_STORES
#define RT_FILL_NO_STORES 0        // (development: the fill's bookkeeping without its stores, to price the two apart)
#endif
__device__ __forceinline__ void fill_untraced_blocks(const DevFrame& f, int wave_global, int n_waves) {
    if (!f.fill_cls) return;
    const int lane = wave_lane();
    const int tbx = f.tbx;
    const int runs_per_row = (tbx + 63) >> 6;
    const int n_units = f.n_local_rows * runs_per_row;
    const int wrb0 = f.lr0 >> 3;
    const int row_bytes = 12 * f.W;
    for (int u = wave_global; u < n_units; u += n_waves) {
        const int lr = u / runs_per_row, bx0 = (u - lr * runs_per_row) << 6;    // (wave-uniform) row, first block of the run
        const int rb = lr >> 3;
        bool fill = false;
        {
            const int bx = bx0 + lane;
            if (bx < tbx) {
                const int wr = rb - wrb0, wc = bx - f.bx0;
                const bool in_window = wr >= 0 && wr < f.fill_window_row_blocks && wc >= 0 && wc < f.nbx;
                fill = !in_window || f.fill_cls[wr * f.nbx + wc] < 0;
            }
        }
        const unsigned long long todo = __ballot(fill);
        if (todo == 0) continue;
#pragma unroll
        for (int part = 0; part < 6; part++) {               // colour: chunk c of the run's 384
            const int c = 64 * part + lane;
            const int blk = c / 6, q = c - 6 * blk;
            const int at = 96 * (bx0 + blk) + 16 * q;
            if (!RT_FILL_NO_STORES && ((todo >> blk) & 1ull) && at < row_bytes)
                *(float4*)((char*)f.rgb + (size_t)lr * row_bytes + at) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        if (f.ids) {
#pragma unroll
            for (int part = 0; part < 2; part++) {           // ids: 64 blocks x 32 bytes are 128 chunks
                const int c = 64 * part + lane;
                const int blk = c >> 1, h = c & 1;
                const int px = 8 * (bx0 + blk) + 4 * h;
                if (!RT_FILL_NO_STORES && ((todo >> blk) & 1ull) && px < f.W)
                    *(int4*)(f.ids + (size_t)lr * f.W + px) = make_int4(-1, -1, -1, -1);
            }
        }
    }
}

// One 64-ray chunk per call, handed out through an atomic cursor so that waves that drew cheap rays
// come back for more.  Every wave reaches base >= n eventually because the cursor only grows.
__device__ __forceinline__ int next_chunk(int* cursor) {
    int base = 0;
    if (wave_lane() == 0) base = atomicAdd(cursor, 64);
    return __builtin_amdgcn_readfirstlane(base);
}

// Slot allocation for the lanes with `take` set: one atomic per wave.
__device__ __forceinline__ int wave_alloc(int* counter, bool take) {
    const unsigned long long mask = __ballot(take);
    if (mask == 0) return -1;
    const int lane = wave_lane();
    const int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(mask));
    base = __shfl(base, leader);
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    return take ? base + rank : -1;
}

__device__ __forceinline__ void flush_work(const WorkCounters& wc, unsigned long long* out) {
    const unsigned long long v[8] = {wc.tri_tests, wc.sphere_tests, wc.xform_evals, wc.cell_steps,
                                     wc.fine_steps, wc.light_evals, wc.nearest_hits, wc.rays};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        unsigned long long x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        if (wave_lane() == 0 && x) atomicAdd(out + k, x);
    }
    // wave-level events are incremented by every lane that was active at the time: the maximum over the lanes
    // is the number of times the wave went through that point
    const unsigned long long e[4] = {wc.w_rounds, wc.w_walk_iters, wc.w_test_batches, wc.w_updates};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        unsigned long long x = e[k];
        for (int off = 32; off > 0; off >>= 1) { const unsigned long long y = __shfl_down(x, off); x = y > x ? y : x; }
        if (wave_lane() == 0 && x) atomicAdd(out + 16 + k, x);
    }
    const unsigned long long cy[4] = {wc.c_update, wc.c_walk, wc.c_test, wc.c_fold};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        unsigned long long x = cy[k];
        for (int off = 32; off > 0; off >>= 1) { const unsigned long long y = __shfl_down(x, off); x = y > x ? y : x; }
        if (wave_lane() == 0 && x) atomicAdd(out + 20 + k, x);
    }
}

// Camera::getRayAt (src/geometries.h:339-351); tanx, tany, halfW, halfH are per-frame constants
__device__ __forceinline__ f3 camera_ray(const DevCamera& c, int x, int y) {
    const float j = (float)x + 0.5f;   // exact in fp32 for any image width below 2^22, as the reference's double add is
    const float i = (float)y + 0.5f;
    const float a = c.tanx * ((j - c.halfW) / c.halfW);
    const float b = c.tany * ((c.halfH - i) / c.halfH);
    return normalize3(a * c.u + b * c.v - c.w);
}

// pow(x, y) for the Blinn-Phong term (src/main.cpp:41), x = max(clamp(N.H), 0) in [0, 1] (or NaN), y = shininess.
// The reference's host libm returns the correctly rounded fp32 power in all but vanishingly rare cases; this
// evaluates exp2(y * log2(x)) in fp64 (atanh series for the logarithm, Taylor series for the exponential, both to
// below 1e-16) and rounds once to fp32, which agrees with that except when the exact value lies within ~1e-13
// relative of a rounding boundary.  Written out by hand because the general library pow needs ~40 more VGPRs.
// A double constant held in scalar registers at the point of use.  The fp64 multiply-add takes no 64-bit literal, so
// the compiler keeps every coefficient below in a vector register pair, loads them all ahead of the frame loop and,
// short of registers there, parks them in scratch memory: 26 dependent scratch reads per evaluation.  Laundered
// through an empty asm, a coefficient is two scalar moves next to the instruction that reads it.
__device__ __forceinline__ double dconst(double c) {
    unsigned lo = (unsigned)(unsigned long long)__double_as_longlong(c), hi = (unsigned)((unsigned long long)__double_as_longlong(c) >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ float pow_shine(float x, float y) {
    if (y == 0.0f || x == 1.0f) return 1.0f;
    if (x != x || y != y) return x + y;                        // NaN in, NaN out
    if (x == 0.0f) return y > 0.0f ? 0.0f : __builtin_inff();
    if (__builtin_isinf(y)) return y > 0.0f ? 0.0f : __builtin_inff();   // 0 < x < 1 here
    const long long bits = __double_as_longlong((double)x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);   // [1, 2)
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0), s2 = s * s;        // ln m = 2 atanh(s), |s| <= 0.1716
    double p = dconst(1.0 / 23.0);
    p = fma(p, s2, dconst(1.0 / 21.0)); p = fma(p, s2, dconst(1.0 / 19.0)); p = fma(p, s2, dconst(1.0 / 17.0)); p = fma(p, s2, dconst(1.0 / 15.0));
    p = fma(p, s2, dconst(1.0 / 13.0)); p = fma(p, s2, dconst(1.0 / 11.0)); p = fma(p, s2, dconst(1.0 / 9.0)); p = fma(p, s2, dconst(1.0 / 7.0));
    p = fma(p, s2, dconst(1.0 / 5.0)); p = fma(p, s2, dconst(1.0 / 3.0)); p = fma(p, s2, 1.0);
    const double log2x = fma(2.0 * s * p, dconst(1.4426950408889634), (double)e);
    const double z = (double)y * log2x;
    if (z < -160.0) return 0.0f;
    if (z > 130.0) return __builtin_inff();
    const double n = rint(z), t = (z - n) * dconst(0.6931471805599453);   // |t| <= 0.3466
    double q = dconst(1.0 / 6227020800.0);
    q = fma(q, t, dconst(1.0 / 479001600.0)); q = fma(q, t, dconst(1.0 / 39916800.0)); q = fma(q, t, dconst(1.0 / 3628800.0));
    q = fma(q, t, dconst(1.0 / 362880.0)); q = fma(q, t, dconst(1.0 / 40320.0)); q = fma(q, t, dconst(1.0 / 5040.0)); q = fma(q, t, dconst(1.0 / 720.0));
    q = fma(q, t, dconst(1.0 / 120.0)); q = fma(q, t, dconst(1.0 / 24.0)); q = fma(q, t, dconst(1.0 / 6.0)); q = fma(q, t, 0.5);
    q = fma(q, t, 1.0); q = fma(q, t, 1.0);
    const double scale = __longlong_as_double((long long)((int)n + 1023) << 52);
    return (float)(q * scale);
}

// calc_lighting (src/main.cpp:30-45)
__device__ __forceinline__ f3 light_term(f3 rayDir, f3 L, f3 N, f3 kd, f3 ks, float shininess, f3 lc) {
    const float lambert = clamp01(dot3(normalize3(L), N));
    const f3 diff = kd * lc * lambert;
    const f3 half = normalize3(normalize3(L) + normalize3(-rayDir));
    const float ph = clamp01(dot3(half, N));
    const float shine = pow_shine(sel_max(ph, 0.0f), shininess);
    const f3 spec = ks * lc * shine;
    return diff + spec;
}
