// Device layer of librtx: scene upload, regular-grid build on the GPU (count / prefix-scan / scatter),
// and the per-pixel trace loop as wavefront ray queues for gfx950.
//
// Frame structure (replaces the recursive trace() of src/main.cpp:81-107 in the reference tree):
//   level d = 0 .. max_depth
//     trace kernel   rays of level d (camera rays for d = 0, mirror rays otherwise) -> nearest hit;
//                    each hit appends a hit record and, while d < max_depth, its mirror ray for level d+1
//     shade kernel   per hit record: one shadow ray per light, Blinn-Phong, L_d = (ambient + emission) + shadow
//   resolve kernels  from the deepest level up: C_d = clampRGB(L_d + specular * C_{d+1}); level 0 writes the pixel
// Every queue lives in device memory; counts stay on the device (kernels read them), so a frame needs no
// host synchronisation until its end.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "rtx.h"
#include "rtx_internal.h"
#include "rt_math.h"
#include "rt_xform.h"
#include "rtx_traverse.h"
#include "rtx_wave.h"

using namespace rt;

#define RTX_MAX_LEVELS 16          // max_depth <= 15
#define RTX_BLOCK 256
#ifndef RTX_FRAME_WAVES_PER_SIMD
#define RTX_FRAME_WAVES_PER_SIMD 3      // occupancy the persistent kernel's register allocation is held to
#endif

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            rtx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);  \
            return RTX_E_DEVICE;                                                                       \
        }                                                                                              \
    } while (0)

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
#define CNT_TILES_ALL 65    // set when a fine cell's screen footprint was too large or undefined: every tile is traced
#define CNT_CLASS 66        // 8 counters: tiles per scheduling class; 74..81: fill cursors per class
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
                                   // row block * tbx + column block; feeds the order of the next frame's list
    int tbx;                       // 8-pixel column blocks per image row
    int hint_radius;               // tiles around a tile whose recorded depths count for its class
    int max_depth;
    float bias;
    int walk_budget;               // persistent kernel: walk steps a lane may take per round
    int ready_threshold;           // persistent kernel: lanes with a finished ray that trigger a state update
    int drain_num;                 // ... after the cursor is exhausted: that many eighths of the lanes still alive
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

// ---------------------------------------------------------------------------------------------------
// trace kernel: nearest hit for every ray of one level
// ---------------------------------------------------------------------------------------------------
template <int MODE, bool COUNT>
__global__ __launch_bounds__(RTX_BLOCK) void k_trace(DevScene sc, DevGrid g, DevFrame f, int level) {
    WorkCounters wc = {};
    const int lane = wave_lane();
    const int n = (level == 0) ? f.n_items0 : f.counters[CNT_HITS + level - 1];
    for (;;) {
        const int base = next_chunk(f.counters + CNT_TRACE_CUR + level);
        if (base >= n) break;
        const int i = base + lane;
        bool active = i < n;
        f3 O = mk(0, 0, 0), D = mk(0, 0, 1);
        int pix = -1;
        if (level == 0) {
            // 64 consecutive work items = one 8x8 pixel block, so a wave's rays stay spatially coherent
            const int blk = i >> 6, in = i & 63;
            const int lr = f.lr0 + 8 * (blk / f.nbx) + (in >> 3);
            const int x = 8 * (f.bx0 + blk % f.nbx) + (in & 7);
            active = active && lr < f.n_local_rows && x < f.W;
            if (active) {
                const int tile = lr / f.tile_rows;
                const int y = (f.rank + tile * f.world) * f.tile_rows + (lr - tile * f.tile_rows);
                pix = lr * f.W + x;
                O = f.cam.eye;
                D = camera_ray(f.cam, x, y);
            }
        } else if (active) {
            const float4 o = f.rO_in[i], d = f.rD_in[i];
            O = mk(o.x, o.y, o.z);
            D = mk(d.x, d.y, d.z);
        }
        float t = 0;
        int prim = -1;
        bool hit = false;
        if (active) hit = traverse<MODE, false, COUNT>(sc, g, O, normalize3(D), FLT_MAX, t, prim, wc);   // src/main.cpp:82
        const int slot = wave_alloc(f.counters + CNT_HITS + level, hit);
        if (hit) {
            if (COUNT) wc.nearest_hits++;
            const f3 Dn = normalize3(D);
            f3 P, N;
            int material;
            finish_hit(sc, prim, O, Dn, t, P, N, material);
            f.hP[slot] = make_float4(P.x, P.y, P.z, as_float(material));
            f.hN[slot] = make_float4(N.x, N.y, N.z, as_float(prim));
            f.hD[slot] = make_float4(D.x, D.y, D.z, 0.0f);
            f.hChild[slot] = -1;
            if (level == 0) {
                f.pix0[slot] = pix;
                if (f.ids) f.ids[pix] = prim;
            } else {
                f.hChild_prev[i] = slot;
            }
            if (level < f.max_depth) {   // src/main.cpp:85-94
                const f3 V = normalize3(-D);
                const f3 R = (2.0f * dot3(V, N) * N) - V;
                const f3 Po = P + f.bias * R;
                f.rO_out[slot] = make_float4(Po.x, Po.y, Po.z, 0.0f);
                f.rD_out[slot] = make_float4(R.x, R.y, R.z, 0.0f);
            }
        } else if (level == 0 && active) {   // src/main.cpp:104-106
            f.rgb[3 * pix] = 0.0f;
            f.rgb[3 * pix + 1] = 0.0f;
            f.rgb[3 * pix + 2] = 0.0f;
            if (f.ids) f.ids[pix] = -1;
        }
    }
    if (COUNT) flush_work(wc, f.work);
}

// pow(x, y) for the Blinn-Phong term (src/main.cpp:41), x = max(clamp(N.H), 0) in [0, 1] (or NaN), y = shininess.
// The reference's host libm returns the correctly rounded fp32 power in all but vanishingly rare cases; this
// evaluates exp2(y * log2(x)) in fp64 (atanh series for the logarithm, Taylor series for the exponential, both to
// below 1e-16) and rounds once to fp32, which agrees with that except when the exact value lies within ~1e-13
// relative of a rounding boundary.  Written out by hand because the general library pow needs ~40 more VGPRs.
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
    double p = 1.0 / 23.0;
    p = fma(p, s2, 1.0 / 21.0); p = fma(p, s2, 1.0 / 19.0); p = fma(p, s2, 1.0 / 17.0); p = fma(p, s2, 1.0 / 15.0);
    p = fma(p, s2, 1.0 / 13.0); p = fma(p, s2, 1.0 / 11.0); p = fma(p, s2, 1.0 / 9.0); p = fma(p, s2, 1.0 / 7.0);
    p = fma(p, s2, 1.0 / 5.0); p = fma(p, s2, 1.0 / 3.0); p = fma(p, s2, 1.0);
    const double log2x = fma(2.0 * s * p, 1.4426950408889634, (double)e);
    const double z = (double)y * log2x;
    if (z < -160.0) return 0.0f;
    if (z > 130.0) return __builtin_inff();
    const double n = rint(z), t = (z - n) * 0.6931471805599453;   // |t| <= 0.3466
    double q = 1.0 / 6227020800.0;
    q = fma(q, t, 1.0 / 479001600.0); q = fma(q, t, 1.0 / 39916800.0); q = fma(q, t, 1.0 / 3628800.0);
    q = fma(q, t, 1.0 / 362880.0); q = fma(q, t, 1.0 / 40320.0); q = fma(q, t, 1.0 / 5040.0); q = fma(q, t, 1.0 / 720.0);
    q = fma(q, t, 1.0 / 120.0); q = fma(q, t, 1.0 / 24.0); q = fma(q, t, 1.0 / 6.0); q = fma(q, t, 0.5);
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

// ---------------------------------------------------------------------------------------------------
// shade kernel: shadowRayTest (src/main.cpp:47-79) per hit record, L = (ambient + emission) + shadow
// ---------------------------------------------------------------------------------------------------
template <int MODE, bool COUNT>
__global__ __launch_bounds__(RTX_BLOCK) void k_shade(DevScene sc, DevGrid g, DevFrame f, int level) {
    WorkCounters wc = {};
    const int lane = wave_lane();
    const int n = f.counters[CNT_HITS + level];
    for (;;) {
        const int base = next_chunk(f.counters + CNT_SHADE_CUR + level);
        if (base >= n) break;
        const int i = base + lane;
        if (i < n) {
        const float4 p4 = f.hP[i], n4 = f.hN[i], d4 = f.hD[i];
        const f3 P = mk(p4.x, p4.y, p4.z), N = mk(n4.x, n4.y, n4.z), rayDir = mk(d4.x, d4.y, d4.z);
        const int material = as_int(p4.w);
        const float* m = sc.mats + 16 * material;
        const f3 ka = mk(m[0], m[1], m[2]), kd = mk(m[3], m[4], m[5]), ks = mk(m[6], m[7], m[8]), ke = mk(m[9], m[10], m[11]);
        const float shininess = m[12];
        f3 acc = mk(0, 0, 0);
        for (int l = 0; l < sc.n_lights; l++) {
            const float* lt = sc.lights + 12 * l;
            const f3 lpos = mk(lt[1], lt[2], lt[3]), lcol = mk(lt[4], lt[5], lt[6]);
            float t;
            int prim;
            if ((int)lt[0] == RTX_LIGHT_POINT) {
                const f3 L = normalize3(lpos - P);
                const f3 So = P + f.bias * L;
                const float tL = dot3(lpos - P, L);
                if (!traverse<MODE, true, COUNT>(sc, g, So, L, tL, t, prim, wc)) {
                    const float att = lt[7] + lt[8] * tL + lt[9] * tL * tL;
                    if (COUNT) wc.light_evals++;
                    acc = acc + light_term(rayDir, L, N, kd, ks, shininess, lcol) / att;
                }
            } else {
                const f3 L = normalize3(lpos);
                const f3 So = P + f.bias * L;
                if (!traverse<MODE, true, COUNT>(sc, g, So, L, FLT_MAX, t, prim, wc)) {
                    if (COUNT) wc.light_evals++;
                    acc = acc + light_term(rayDir, L, N, kd, ks, shininess, lcol);
                }
            }
        }
        const f3 sh = clamp_rgb(acc);
        const f3 Lsum = ka + ke + sh;   // src/main.cpp:99-101, left to right
        f.hL[i] = make_float4(Lsum.x, Lsum.y, Lsum.z, as_float(material));
        }
    }
    if (COUNT) flush_work(wc, f.work + 8);
}

// ---------------------------------------------------------------------------------------------------
// resolve kernel: the return path of trace(), C = clampRGB(L + specular * C_child) (src/main.cpp:99-102)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RTX_BLOCK) void k_resolve(DevScene sc, DevFrame f, int level) {
    const int n = f.counters[CNT_HITS + level];
    const int stride = (int)(gridDim.x * blockDim.x);
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < n; i += stride) {
        const float4 l4 = f.hL[i];
        const int material = as_int(l4.w);
        const float* m = sc.mats + 16 * material;
        const f3 ks = mk(m[6], m[7], m[8]);
        f3 refl = mk(0, 0, 0);
        if (level < f.max_depth) {
            const int c = f.hChild[i];
            if (c >= 0) {
                const float4 c4 = f.hL_next[c];
                refl = mk(c4.x, c4.y, c4.z);
            }
        }
        const f3 C = clamp_rgb(mk(l4.x, l4.y, l4.z) + ks * refl);
        if (level == 0) {
            const int pix = f.pix0[i];
            f.rgb[3 * pix] = C.x;
            f.rgb[3 * pix + 1] = C.y;
            f.rgb[3 * pix + 2] = C.z;
        } else {
            f.hL[i] = make_float4(C.x, C.y, C.z, l4.w);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// wave-cooperative variants (rtx_wave.h): same queues, same records, same arithmetic; the 64 lanes of a wave
// share the candidate tests of their 64 rays
// ---------------------------------------------------------------------------------------------------
template <int MODE, bool SINGLE_XF, bool COUNT>
__global__ __launch_bounds__(RTX_BLOCK) void k_trace_w(DevScene sc, DevGrid g, DevSparse sp, DevFrame f, int level) {
    __shared__ WaveLds lds[RTX_BLOCK / 64];
    WaveLds& l = lds[threadIdx.x >> 6];
    __shared__ unsigned occ_lds[RT_OCC_LDS_WORDS];
    const Occupancy occ = stage_occupancy(g, occ_lds);
    const XfRegs xf = load_xf_regs(sc.xinv, SINGLE_XF);
    WorkCounters wc = {};
    const int lane = wave_lane();
    const int n = (level == 0) ? f.n_items0 : f.counters[CNT_HITS + level - 1];
    for (;;) {
        const int base = next_chunk(f.counters + CNT_TRACE_CUR + level);
        if (base >= n) break;
        const int i = base + lane;
        bool active = i < n;
        f3 O = mk(0, 0, 0), D = mk(0, 0, 1);
        int pix = -1;
        if (level == 0) {
            const int blk = i >> 6, in = i & 63;
            const int lr = f.lr0 + 8 * (blk / f.nbx) + (in >> 3);
            const int x = 8 * (f.bx0 + blk % f.nbx) + (in & 7);
            active = active && lr < f.n_local_rows && x < f.W;
            if (active) {
                const int tile = lr / f.tile_rows;
                const int y = (f.rank + tile * f.world) * f.tile_rows + (lr - tile * f.tile_rows);
                pix = lr * f.W + x;
                O = f.cam.eye;
                D = camera_ray(f.cam, x, y);
            }
        } else if (active) {
            const float4 o = f.rO_in[i], d = f.rD_in[i];
            O = mk(o.x, o.y, o.z);
            D = mk(d.x, d.y, d.z);
        }
        const f3 Dn = normalize3(D);
        float t = 0;
        int prim = -1;
        const bool hit = trace_wave<MODE, false, SINGLE_XF, COUNT>(sc, g, sp, occ, xf, l, O, Dn, FLT_MAX, active, t, prim, wc);
        const int slot = wave_alloc(f.counters + CNT_HITS + level, hit);
        if (hit) {
            if (COUNT) wc.nearest_hits++;
            f3 P, N;
            int material;
            finish_hit(sc, prim, O, Dn, t, P, N, material);
            f.hP[slot] = make_float4(P.x, P.y, P.z, as_float(material));
            f.hN[slot] = make_float4(N.x, N.y, N.z, as_float(prim));
            f.hD[slot] = make_float4(D.x, D.y, D.z, 0.0f);
            f.hChild[slot] = -1;
            if (level == 0) {
                f.pix0[slot] = pix;
                if (f.ids) f.ids[pix] = prim;
            } else {
                f.hChild_prev[i] = slot;
            }
            if (level < f.max_depth) {
                const f3 V = normalize3(-D);
                const f3 R = (2.0f * dot3(V, N) * N) - V;
                const f3 Po = P + f.bias * R;
                f.rO_out[slot] = make_float4(Po.x, Po.y, Po.z, 0.0f);
                f.rD_out[slot] = make_float4(R.x, R.y, R.z, 0.0f);
            }
        } else if (level == 0 && active) {
            f.rgb[3 * pix] = 0.0f;
            f.rgb[3 * pix + 1] = 0.0f;
            f.rgb[3 * pix + 2] = 0.0f;
            if (f.ids) f.ids[pix] = -1;
        }
    }
    if (COUNT) flush_work(wc, f.work);
}

template <int MODE, bool SINGLE_XF, bool COUNT>
__global__ __launch_bounds__(RTX_BLOCK) void k_shade_w(DevScene sc, DevGrid g, DevSparse sp, DevFrame f, int level) {
    __shared__ WaveLds lds[RTX_BLOCK / 64];
    WaveLds& l = lds[threadIdx.x >> 6];
    __shared__ unsigned occ_lds[RT_OCC_LDS_WORDS];
    const Occupancy occ = stage_occupancy(g, occ_lds);
    const XfRegs xf = load_xf_regs(sc.xinv, SINGLE_XF);
    WorkCounters wc = {};
    const int lane = wave_lane();
    const int n = f.counters[CNT_HITS + level];
    for (;;) {
        const int base = next_chunk(f.counters + CNT_SHADE_CUR + level);
        if (base >= n) break;
        const int i = base + lane;
        const bool active = i < n;
        const int ii = active ? i : 0;
        const float4 p4 = f.hP[ii], n4 = f.hN[ii], d4 = f.hD[ii];
        const f3 P = mk(p4.x, p4.y, p4.z), N = mk(n4.x, n4.y, n4.z), rayDir = mk(d4.x, d4.y, d4.z);
        const int material = as_int(p4.w);
        const float* m = sc.mats + 16 * material;
        const f3 ka = mk(m[0], m[1], m[2]), kd = mk(m[3], m[4], m[5]), ks = mk(m[6], m[7], m[8]), ke = mk(m[9], m[10], m[11]);
        const float shininess = m[12];
        f3 acc = mk(0, 0, 0);
        for (int li = 0; li < sc.n_lights; li++) {          // wave-uniform loop
            const float* lt = sc.lights + 12 * li;
            const f3 lpos = mk(lt[1], lt[2], lt[3]), lcol = mk(lt[4], lt[5], lt[6]);
            const bool point = (int)lt[0] == RTX_LIGHT_POINT;
            const f3 L = point ? normalize3(lpos - P) : normalize3(lpos);
            const f3 So = P + f.bias * L;
            const float tL = point ? dot3(lpos - P, L) : FLT_MAX;
            float t;
            int prim;
            const bool occluded = trace_wave<MODE, true, SINGLE_XF, COUNT>(sc, g, sp, occ, xf, l, So, L, tL, active, t, prim, wc);
            if (active && !occluded) {
                if (COUNT) wc.light_evals++;
                const f3 c = light_term(rayDir, L, N, kd, ks, shininess, lcol);
                if (point) {
                    const float att = lt[7] + lt[8] * tL + lt[9] * tL * tL;
                    acc = acc + c / att;
                } else {
                    acc = acc + c;
                }
            }
        }
        if (active) {
            const f3 sh = clamp_rgb(acc);
            const f3 Lsum = ka + ke + sh;
            f.hL[i] = make_float4(Lsum.x, Lsum.y, Lsum.z, as_float(material));
        }
    }
    if (COUNT) flush_work(wc, f.work + 8);
}

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
#define RTX_LDS_LEVELS 6              // paths up to this many levels keep their per-level terms in LDS

template <int MODE, bool SINGLE_XF, bool COUNT, int PRIMS>
__global__ __launch_bounds__(RTX_BLOCK, RTX_FRAME_WAVES_PER_SIMD) void k_frame(DevScene sc, DevGrid g, DevSparse sp, DevFrame f, float4* path_stack, int levels) {
    __shared__ WaveLds lds[RTX_BLOCK / 64];
    __shared__ float4 stack_lds[RTX_BLOCK / 64][RTX_LDS_LEVELS][64];
    __shared__ int hit_count[RTX_BLOCK / 64][RTX_MAX_LEVELS];
    __shared__ unsigned occ_lds[RT_OCC_LDS_WORDS];
    const Occupancy occ = stage_occupancy(g, occ_lds);
    const XfRegs xf = load_xf_regs(sc.xinv, SINGLE_XF);
    const int wave = (int)(threadIdx.x >> 6), lane = wave_lane();
    WaveLds& l = lds[wave];
    if (lane < RTX_MAX_LEVELS) hit_count[wave][lane] = 0;
    // per-level terms of this lane's path: LDS when the depth allows, else a private strip of global memory
    const bool stack_in_lds = levels <= RTX_LDS_LEVELS;
    float4* stack_g = path_stack + ((size_t)(blockIdx.x * (RTX_BLOCK / 64) + wave) * 64 + lane) * levels;
    WorkCounters wc = {};

    // path state of this lane; the ray in flight lives in the walker (w.O, w.D, w.t_limit)
    int pix = -1, level = 0, phase = 0, material = 0;   // phase 0: nearest-hit ray in flight; 1 + i: shadow ray of light i
    int tslot = 0;                                      // 8x8 block of the pixel (index into the cost map)
    f3 rayDir = mk(0, 0, 1), P = mk(0, 0, 0), N = mk(0, 0, 1), acc = mk(0, 0, 0);
    Walker<MODE> w;
    LaneSearch r;
    w.init(g, mk(0, 0, 0), mk(0, 0, 1), FLT_MAX, false);
    r.found = false; r.best_t = FLT_MAX; r.best_prim = -1;
    bool exhausted = false;
    const int n_items = f.tile_list ? 64 * __builtin_amdgcn_readfirstlane(f.counters[CNT_TILES]) : f.n_items0;

    for (;;) {
        const bool walking = pix >= 0 && w.state != 2;
        const bool finished = pix >= 0 && w.state == 2;
        const int n_ready = __popcll(__ballot(finished)) + (exhausted ? 0 : __popcll(__ballot(pix < 0)));
        // once the cursor is exhausted the wave drains: with few paths left, waiting for a full batch of finished
        // rays would chain every path to the slowest ray of each generation
        const int n_alive = __popcll(__ballot(pix >= 0));
        const int threshold = exhausted ? min(f.ready_threshold, max(1, (n_alive * f.drain_num) >> 3)) : f.ready_threshold;
        if (!__any(walking) || n_ready >= threshold) {
            if (COUNT) wc.w_updates++;
            const unsigned long long c0 = COUNT ? __builtin_readcyclecounter() : 0ull;
            bool begin = false;
            f3 O = w.O, D = w.D;
            float tl = w.t_limit;
            if (finished) {
                bool path_end = false, lit = false;
                int top = -1, next_light = -1;
                if (phase == 0) {
                    if (!w.hit) {                              // src/main.cpp:104-106
                        path_end = true;
                        top = level - 1;
                        if (level == 0 && f.ids) f.ids[pix] = -1;
                    } else {
                        if (COUNT) wc.nearest_hits++;
                        const int prim = r.best_prim;
                        finish_hit(sc, prim, O, D, r.best_t, P, N, material);
                        atomicAdd(&hit_count[wave][level], 1);
                        if (level == 0 && f.ids) f.ids[pix] = prim;
                        acc = mk(0, 0, 0);
                        if (sc.n_lights > 0) next_light = 0; else lit = true;
                    }
                } else {                                        // src/main.cpp:57-75
                    const int li = phase - 1;
                    if (!w.hit) {
                        const float* lt = sc.lights + 12 * li;
                        const float* m = sc.mats + 16 * material;
                        if (COUNT) wc.light_evals++;
                        const f3 c = light_term(rayDir, D, N, mk(m[3], m[4], m[5]), mk(m[6], m[7], m[8]), m[12], mk(lt[4], lt[5], lt[6]));
                        if ((int)lt[0] == RTX_LIGHT_POINT) {
                            const float att = lt[7] + lt[8] * tl + lt[9] * tl * tl;
                            acc = acc + c / att;
                        } else {
                            acc = acc + c;
                        }
                    }
                    if (li + 1 < sc.n_lights) next_light = li + 1; else lit = true;
                }
                if (lit) {                                      // src/main.cpp:85-102
                    const float* m = sc.mats + 16 * material;
                    const f3 Lsum = mk(m[0], m[1], m[2]) + mk(m[9], m[10], m[11]) + clamp_rgb(acc);
                    const float4 rec = make_float4(Lsum.x, Lsum.y, Lsum.z, as_float(material));
                    if (stack_in_lds) stack_lds[wave][level][lane] = rec; else stack_g[level] = rec;
                    if (level < f.max_depth) {
                        const f3 V = normalize3(-rayDir);
                        const f3 R = (2.0f * dot3(V, N) * N) - V;
                        O = P + f.bias * R;
                        rayDir = R;
                        D = normalize3(R);
                        tl = FLT_MAX;
                        level++;
                        phase = 0;
                        begin = true;
                    } else {
                        path_end = true;
                        top = level;
                    }
                }
                if (next_light >= 0) {
                    const float* lt = sc.lights + 12 * next_light;
                    const f3 lpos = mk(lt[1], lt[2], lt[3]);
                    const bool point = (int)lt[0] == RTX_LIGHT_POINT;
                    const f3 L = point ? normalize3(lpos - P) : normalize3(lpos);
                    O = P + f.bias * L;
                    D = L;
                    tl = point ? dot3(lpos - P, L) : FLT_MAX;
                    phase = 1 + next_light;
                    begin = true;
                }
                if (path_end) {
                    f3 C = mk(0, 0, 0);
                    const auto fold = [&](const float4 s4) {
                        const float* m = sc.mats + 16 * as_int(s4.w);
                        C = clamp_rgb(mk(s4.x, s4.y, s4.z) + mk(m[6], m[7], m[8]) * C);
                    };
                    // one loop per address space (the choice is uniform over the launch): a pointer selected inside a
                    // single loop would turn every read into a flat load
                    if (stack_in_lds) {
                        const __attribute__((address_space(3))) float* sl =
                            (const __attribute__((address_space(3))) float*)&stack_lds[wave][0][lane];
                        for (int d = top; d >= 0; d--) fold(make_float4(sl[256 * d], sl[256 * d + 1], sl[256 * d + 2], sl[256 * d + 3]));
                    } else {
                        for (int d = top; d >= 0; d--) fold(stack_g[d]);
                    }
                    f.rgb[3 * pix] = C.x;
                    f.rgb[3 * pix + 1] = C.y;
                    f.rgb[3 * pix + 2] = C.z;
                    if (f.tile_cost && f.tile_cost[tslot] < top + 2) atomicMax(f.tile_cost + tslot, top + 2);
                    pix = -1;
                }
            }
            // lanes without a path draw the next pixels
            const bool idle = pix < 0;
            const unsigned long long idle_mask = __ballot(idle);
            if (!exhausted && idle_mask != 0) {
                const int n_idle = __popcll(idle_mask);
                int base = 0;
                if (lane == 0) base = atomicAdd(f.counters + CNT_TRACE_CUR, n_idle);
                base = __builtin_amdgcn_readfirstlane(base);
                if (idle) {
                    const int i = base + __popcll(idle_mask & ((1ull << lane) - 1ull));
                    if (i < n_items) {
                        const int blk = i >> 6, in = i & 63;
                        int lr, x;
                        if (f.tile_list) {
                            const int t = f.tile_list[blk];
                            lr = 8 * (t >> 16) + (in >> 3);
                            x = 8 * (t & 0xffff) + (in & 7);
                            tslot = (t >> 16) * f.tbx + (t & 0xffff);
                        } else {
                            lr = f.lr0 + 8 * (blk / f.nbx) + (in >> 3);
                            x = 8 * (f.bx0 + blk % f.nbx) + (in & 7);
                        }
                        if (lr < f.n_local_rows && x < f.W) {
                            const int tile = lr / f.tile_rows;
                            const int y = (f.rank + tile * f.world) * f.tile_rows + (lr - tile * f.tile_rows);
                            pix = lr * f.W + x;
                            O = f.cam.eye;
                            rayDir = camera_ray(f.cam, x, y);
                            D = normalize3(rayDir);             // src/main.cpp:82
                            tl = FLT_MAX;
                            level = 0;
                            phase = 0;
                            begin = true;
                        }
                    }
                }
                if (base + n_idle >= n_items) exhausted = true;
            }
            if (begin) ray_begin<MODE, SINGLE_XF, COUNT>(sc, g, l, w, r, O, D, tl, wc);
            if (COUNT) wc.c_update += __builtin_readcyclecounter() - c0;
            if (exhausted && !__any(pix >= 0)) break;
        }
        if (__any(pix >= 0 && w.state != 2)) wave_round<MODE, SINGLE_XF, COUNT, PRIMS>(sc, g, sp, occ, xf, l, w, r, phase != 0, f.walk_budget, wc);
    }
    if (lane < RTX_MAX_LEVELS && hit_count[wave][lane]) atomicAdd(f.counters + CNT_HITS + lane, hit_count[wave][lane]);
    if (COUNT) flush_work(wc, f.work);
}

// ---------------------------------------------------------------------------------------------------
// Screen tiles that can see geometry (nested search, persistent kernel).  A camera ray can only hit a primitive
// inside an occupied fine cell (every point of a primitive lies in a fine cell that references it), so an 8x8 pixel
// tile outside the projection of every occupied fine cell returns the miss colour for all its pixels
// (src/main.cpp:104-106) and is left to the fills.  k_tile_mark projects the eight corners of every occupied fine
// cell (grown by the binning margin) and marks the tiles of the bounding rectangle grown by two pixels; a cell
// with a corner beside or behind the eye, or one covering more than 1024 tiles, raises the "every tile" flag
// instead.  k_tile_list then collects this rank's marked blocks of the traced window.
// ---------------------------------------------------------------------------------------------------
__global__ void k_tile_mark(DevCamera cam, DevGrid g, DevSparse sp, float pad, int n_fine_cells, int tbx, int tby,
                            unsigned char* tile_map, int* counters) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_fine_cells) return;
    const int cell = i / g.cpb, local = i - cell * g.cpb;
    const int k = g.sub;
    if (local >= k * k * k) return;                       // padding of the block to whole mask words
    const uint2 word = sp.words[cell * sp.words_per_block + (local >> 5)];
    if (!((word.x >> (local & 31)) & 1u)) return;
    const int cx = cell % g.rx, cy = (cell / g.rx) % g.ry, cz = cell / (g.rx * g.ry);
    const int lx = local % k, ly = (local / k) % k, lz = local / (k * k);
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
        const double a = (dx * cam.u.x + dy * cam.u.y + dz * cam.u.z) / depth;
        const double b = (dx * cam.v.x + dy * cam.v.y + dz * cam.v.z) / depth;
        const double sx = a / cam.tanx * cam.halfW + cam.halfW - 0.5;      // inverse of Camera::getRayAt (src/geometries.h:339-351)
        const double sy = cam.halfH - 0.5 - b / cam.tany * cam.halfH;
        if (!(sx == sx) || !(sy == sy)) { undefined = true; break; }
        xmin = fmin(xmin, sx); xmax = fmax(xmax, sx);
        ymin = fmin(ymin, sy); ymax = fmax(ymax, sy);
    }
    if (undefined) { counters[CNT_TILES_ALL] = 1; return; }
    const double margin = 2.0;
    if (xmax + margin < 0.0 || ymax + margin < 0.0 || xmin - margin > 8.0 * tbx || ymin - margin > 8.0 * tby) return;   // off screen
    const int tx0 = (int)fmax(0.0, floor((xmin - margin) / 8.0)), tx1 = (int)fmin((double)(tbx - 1), floor((xmax + margin) / 8.0));
    const int ty0 = (int)fmax(0.0, floor((ymin - margin) / 8.0)), ty1 = (int)fmin((double)(tby - 1), floor((ymax + margin) / 8.0));
    if (tx1 < tx0 || ty1 < ty0) return;
    if ((long long)(tx1 - tx0 + 1) * (ty1 - ty0 + 1) > 1024) { counters[CNT_TILES_ALL] = 1; return; }
    for (int ty = ty0; ty <= ty1; ty++)
        for (int tx = tx0; tx <= tx1; tx++) tile_map[ty * tbx + tx] = 1;
}

// One lane per 8x8 block of this rank's traced window (rows are local rows; a block's eight rows may come from
// different global tiles when the rank split is not in multiples of eight rows).  Blocks that can see geometry are
// put into eight scheduling classes by the deepest path the block produced in the previous frame of the same
// geometry (0 = unknown): k_tile_classify counts the classes, k_tile_fill writes the list with the deepest class
// first, so that the long dependent chains of a frame start early and the frame does not end on them.
__device__ __forceinline__ bool tile_block(const DevFrame& f, int i, int n_row_blocks, int tbx, const unsigned char* tile_map,
                                           int& rb, int& bx) {
    if (i >= n_row_blocks * f.nbx) return false;
    rb = f.lr0 / 8 + i / f.nbx;
    bx = f.bx0 + i % f.nbx;
    bool take = f.counters[CNT_TILES_ALL] != 0;
    for (int r = 0; r < 8 && !take; r++) {
        const int lr = 8 * rb + r;
        if (lr >= f.n_local_rows) break;
        const int tile = lr / f.tile_rows;
        const int y = (f.rank + tile * f.world) * f.tile_rows + (lr - tile * f.tile_rows);
        take = tile_map[(y >> 3) * tbx + bx] != 0;
    }
    return take;
}
__global__ __launch_bounds__(256) void k_tile_classify(DevFrame f, int n_row_blocks, const unsigned char* tile_map, const int* prev_cost,
                                                       signed char* cls) {
    __shared__ int cnt[8];
    if (threadIdx.x < 8) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    int rb = 0, bx = 0;
    const bool take = tile_block(f, i, n_row_blocks, f.tbx, tile_map, rb, bx);
    int c = -1;
    if (take) {
        c = 0;
        if (prev_cost) {
            // the deepest class within `hint_radius` tiles: a camera that moved a little still finds its deep tiles
            const int nrb = (f.n_local_rows + 7) / 8, R = f.hint_radius;
            for (int dy = -R; dy <= R; dy++)
                for (int dx = -R; dx <= R; dx++) {
                    const int y = rb + dy, x = bx + dx;
                    if (y >= 0 && y < nrb && x >= 0 && x < f.tbx) c = max(c, prev_cost[y * f.tbx + x]);
                }
            c = min(c, 7);
        }
    }
    if (i < n_row_blocks * f.nbx) cls[i] = (signed char)c;
    if (c >= 0) atomicAdd(&cnt[c], 1);
    __syncthreads();
    if (threadIdx.x < 8 && cnt[threadIdx.x]) {
        atomicAdd(f.counters + CNT_CLASS + threadIdx.x, cnt[threadIdx.x]);
        atomicAdd(f.counters + CNT_TILES, cnt[threadIdx.x]);
    }
}
__global__ __launch_bounds__(256) void k_tile_fill(DevFrame f, int n_row_blocks, const signed char* cls, int* tile_list) {
    __shared__ int cnt[8], base[8], rank[8];
    if (threadIdx.x < 8) { cnt[threadIdx.x] = 0; rank[threadIdx.x] = 0; }
    __syncthreads();
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int n = n_row_blocks * f.nbx;
    const int c = i < n ? (int)cls[i] : -1;
    if (c >= 0) atomicAdd(&cnt[c], 1);
    __syncthreads();
    if (threadIdx.x < 8) {
        const int k = (int)threadIdx.x;
        int offset = 0;                                       // classes above k come first in the list
        for (int j = k + 1; j < 8; j++) offset += f.counters[CNT_CLASS + j];
        base[k] = offset + (cnt[k] ? atomicAdd(f.counters + CNT_CLASS + 8 + k, cnt[k]) : 0);
    }
    __syncthreads();
    if (c >= 0) tile_list[base[c] + atomicAdd(&rank[c], 1)] = ((f.lr0 / 8 + i / f.nbx) << 16) | (f.bx0 + i % f.nbx);
}

// ---------------------------------------------------------------------------------------------------
// 8-bit conversion of a float framebuffer on the device: (unsigned char)(clamp(c) * 255), truncating, as
// Image3f::write3fPpm and get_3b_bgr_buffer do (src/Image3f.h:63-80,154-170).  Four pixels per lane: three
// 16-byte loads, one 12-byte store.  Pure streaming, 15 bytes of traffic per pixel.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned to_u8_dev(float c) {
    const float hi = (1.0f < c) ? 1.0f : c;          // same selects as clamp() in src/Image3f.h:15-17
    const float v = (hi < 0.0f) ? 0.0f : hi;
    const float s = v * 255.0f;
    return (s == s) ? (unsigned)(int)s : 0u;         // s is in [0, 255] or NaN
}

template <bool BGR>
__global__ __launch_bounds__(256) void k_quantize(const float* __restrict__ rgb, long long n_pixels, unsigned char* __restrict__ out) {
    const long long n_quads = n_pixels >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n_quads; q += stride) {
        const float4* src = reinterpret_cast<const float4*>(rgb) + 3 * q;
        const float4 a = src[0], b = src[1], c = src[2];
        const float v[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
        unsigned w[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            unsigned word = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int e = 4 * k + j;                               // output byte e of this quad
                const int from = BGR ? 3 * (e / 3) + (2 - e % 3) : e;  // channel swap inside each pixel
                word |= to_u8_dev(v[from]) << (8 * j);
            }
            w[k] = word;
        }
        uint3* dst = reinterpret_cast<uint3*>(out) + q;
        *dst = make_uint3(w[0], w[1], w[2]);
    }
    // up to three pixels that do not fill a quad
    const long long tail = n_quads << 2;
    const long long i = tail + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pixels) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) out[3 * i + ch] = (unsigned char)to_u8_dev(rgb[3 * i + (BGR ? 2 - ch : ch)]);
    }
}

// ---------------------------------------------------------------------------------------------------
// probes (tests): single rays through the grid, camera rays
// ---------------------------------------------------------------------------------------------------
template <int MODE>
__global__ void k_probe(DevScene sc, DevGrid g, int n, const float* O, const float* D, const float* tl, float* out_t,
                        int* out_prim, float* out_pos, float* out_n) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    WorkCounters wc = {};
    const f3 o = mk(O[3 * i], O[3 * i + 1], O[3 * i + 2]), d = mk(D[3 * i], D[3 * i + 1], D[3 * i + 2]);
    float t = FLT_MAX;
    int prim = -1;
    const bool hit = traverse<MODE, false, false>(sc, g, o, d, tl ? tl[i] : FLT_MAX, t, prim, wc);
    f3 P = mk(0, 0, 0), N = mk(0, 0, 0);
    int material = -1;
    if (hit) finish_hit(sc, prim, o, d, t, P, N, material);
    out_t[i] = hit ? t : FLT_MAX;
    out_prim[i] = hit ? prim : -1;
    if (out_pos) { out_pos[3 * i] = P.x; out_pos[3 * i + 1] = P.y; out_pos[3 * i + 2] = P.z; }
    if (out_n) { out_n[3 * i] = N.x; out_n[3 * i + 1] = N.y; out_n[3 * i + 2] = N.z; }
}

__global__ void k_camera_rays(DevCamera c, int n, const int* xy, float* dirs) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const f3 d = camera_ray(c, xy[2 * i], xy[2 * i + 1]);
    dirs[3 * i] = d.x; dirs[3 * i + 1] = d.y; dirs[3 * i + 2] = d.z;
}

// ---------------------------------------------------------------------------------------------------
// grid build kernels
// ---------------------------------------------------------------------------------------------------
// World boxes of every primitive: `ref` as the reference computes them (8 transformed corners, minimum
// seeded with FLT_MAX and maximum with FLT_MIN -- the smallest positive float, so every maximum is >= ~0;
// src/geometries.h:122-155,253-259,313-318), `tight` with a true maximum seed.
__global__ void k_boxes(int n, const float* geom, const int* kind, const int* xform, const float* xfwd, float* ref_box,
                        float* tight_box) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const float* gp = geom + 9 * i;
    f3 lo, hi;
    if (kind[i] == RTX_PRIM_TRIANGLE) {
        const f3 A = mk(gp[0], gp[1], gp[2]), B = mk(gp[3], gp[4], gp[5]), C = mk(gp[6], gp[7], gp[8]);
        lo = sel_min(sel_min(A, B), C);
        hi = sel_max(sel_max(A, B), C);
    } else {
        const f3 S = mk(gp[0], gp[1], gp[2]), d = mk(gp[3], gp[3], gp[3]);
        lo = S - d;
        hi = S + d;
    }
    const float* m = xfwd + 16 * xform[i];
    f3 rmin = mk(FLT_MAX, FLT_MAX, FLT_MAX), rmax = mk(FLT_MIN, FLT_MIN, FLT_MIN);
    f3 tmin = rmin, tmax = mk(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int c = 0; c < 8; c++) {
        const f3 corner = mk((c & 1) ? hi.x : lo.x, (c & 2) ? hi.y : lo.y, (c & 4) ? hi.z : lo.z);
        const f3 p = mul_point(m, corner);
        rmin = sel_min(rmin, p);
        rmax = sel_max(rmax, p);
        tmin = sel_min(tmin, p);
        tmax = sel_max(tmax, p);
    }
    float* r = ref_box + 6 * i;
    r[0] = rmin.x; r[1] = rmin.y; r[2] = rmin.z; r[3] = rmax.x; r[4] = rmax.y; r[5] = rmax.z;
    float* t = tight_box + 6 * i;
    t[0] = tmin.x; t[1] = tmin.y; t[2] = tmin.z; t[3] = tmax.x; t[4] = tmax.y; t[5] = tmax.z;
}

// Grid::getSceneBounds (src/grid.h:21-34): min over (start, end), max over (end, start), then -+ 0.001.
// One workgroup; min/max are exact, so the reduction order does not change the result.
__global__ __launch_bounds__(1024) void k_scene_bounds(int n, const float* ref_box, float* bounds) {
    __shared__ float smin[3][1024];
    __shared__ float smax[3][1024];
    const int tid = (int)threadIdx.x;
    f3 mn = mk(FLT_MAX, FLT_MAX, FLT_MAX), mx = mk(FLT_MIN, FLT_MIN, FLT_MIN);
    for (int i = tid; i < n; i += 1024) {
        const float* b = ref_box + 6 * i;
        const f3 s = mk(b[0], b[1], b[2]), e = mk(b[3], b[4], b[5]);
        mn = sel_min(sel_min(mn, s), e);
        mx = sel_max(sel_max(mx, e), s);
    }
    smin[0][tid] = mn.x; smin[1][tid] = mn.y; smin[2][tid] = mn.z;
    smax[0][tid] = mx.x; smax[1][tid] = mx.y; smax[2][tid] = mx.z;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (tid < w) {
            for (int a = 0; a < 3; a++) {
                smin[a][tid] = sel_min(smin[a][tid], smin[a][tid + w]);
                smax[a][tid] = sel_max(smax[a][tid], smax[a][tid + w]);
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float eps = 0.001f;
        for (int a = 0; a < 3; a++) {
            bounds[a] = smin[a][0] - eps;
            bounds[3 + a] = smax[a][0] + eps;
        }
    }
}

struct BinParams {
    f3 lo, size, res;              // getCellIndicesAtPosition operands (src/grid.h:53-59)
    int rx, ry, rz;
    float pad;                     // box growth (0 for the reference lists)
    int sub, cpb;                  // sub > 0: block-major numbering, cell = coarse * cpb + local (nested grid)
};

// Inclusive cell range of one box: clamp(floor((p - lo) * res / size), 0, res - 1) per corner and axis
__device__ __forceinline__ void cell_range(const BinParams& bp, const float* box, int* c0, int* c1) {
    const float lo[3] = {bp.lo.x, bp.lo.y, bp.lo.z}, size[3] = {bp.size.x, bp.size.y, bp.size.z};
    const float res[3] = {bp.res.x, bp.res.y, bp.res.z};
    for (int a = 0; a < 3; a++) {
        c0[a] = (int)clampf(floorf(((box[a] - bp.pad) - lo[a]) * res[a] / size[a]), 0.0f, res[a] - 1.0f);
        c1[a] = (int)clampf(floorf(((box[3 + a] + bp.pad) - lo[a]) * res[a] / size[a]), 0.0f, res[a] - 1.0f);
    }
}

// Conservative triangle / box overlap by separating axes (nine edge-cross-axis directions and the triangle's plane;
// the three box axes are covered by the caller's cell range).  c = box centre, h = half extent already grown by the
// binning margin, which is orders of magnitude above the rounding of this test, so a triangle that touches the
// un-grown cell is never rejected.
__device__ __forceinline__ bool tri_box_overlap(f3 c, f3 h, f3 a, f3 b, f3 cc) {
    const f3 v0 = a - c, v1 = b - c, v2 = cc - c;
    const f3 e[3] = {v1 - v0, v2 - v1, v0 - v2};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const f3 ed = e[k];
        // axis (0, -ez, ey)
        float p0 = ed.y * v0.z - ed.z * v0.y, p1 = ed.y * v1.z - ed.z * v1.y, p2 = ed.y * v2.z - ed.z * v2.y;
        float rad = fabsf(ed.z) * h.y + fabsf(ed.y) * h.z;
        if (fminf(p0, fminf(p1, p2)) > rad || fmaxf(p0, fmaxf(p1, p2)) < -rad) return false;
        // axis (ez, 0, -ex)
        p0 = ed.z * v0.x - ed.x * v0.z; p1 = ed.z * v1.x - ed.x * v1.z; p2 = ed.z * v2.x - ed.x * v2.z;
        rad = fabsf(ed.z) * h.x + fabsf(ed.x) * h.z;
        if (fminf(p0, fminf(p1, p2)) > rad || fmaxf(p0, fmaxf(p1, p2)) < -rad) return false;
        // axis (-ey, ex, 0)
        p0 = ed.x * v0.y - ed.y * v0.x; p1 = ed.x * v1.y - ed.y * v1.x; p2 = ed.x * v2.y - ed.y * v2.x;
        rad = fabsf(ed.y) * h.x + fabsf(ed.x) * h.y;
        if (fminf(p0, fminf(p1, p2)) > rad || fmaxf(p0, fmaxf(p1, p2)) < -rad) return false;
    }
    const f3 nrm = cross3(e[0], e[1]);
    const float d = dot3(nrm, v0);
    const float r = h.x * fabsf(nrm.x) + h.y * fabsf(nrm.y) + h.z * fabsf(nrm.z);
    return fabsf(d) <= r;
}

// pass 1 (FILL = false): count references per cell; pass 2 (FILL = true): scatter primitive ids,
// Grid::placeIntoGrid (src/grid.h:76-88) for every primitive at once.  With geom != nullptr (nested grid only) a
// triangle is referenced only by the cells of its box range that the triangle itself overlaps.
template <bool FILL>
__global__ void k_bin(int n, BinParams bp, const float* boxes, int* count_or_cursor, const int* start, int* ids,
                      const float* geom, const int* kind, const int* xform, const float* xfwd) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    int c0[3], c1[3];
    cell_range(bp, boxes + 6 * i, c0, c1);
    bool exact = false;
    f3 wa = mk(0, 0, 0), wb = wa, wc = wa, cellsz = wa, half = wa;
    if (geom != nullptr && kind[i] == RTX_PRIM_TRIANGLE && (c1[0] > c0[0] || c1[1] > c0[1] || c1[2] > c0[2])) {
        const float* gp = geom + 9 * i;
        const float* m = xfwd + 16 * xform[i];
        wa = mul_point(m, mk(gp[0], gp[1], gp[2]));
        wb = mul_point(m, mk(gp[3], gp[4], gp[5]));
        wc = mul_point(m, mk(gp[6], gp[7], gp[8]));
        cellsz = mk(bp.size.x / bp.res.x, bp.size.y / bp.res.y, bp.size.z / bp.res.z);
        half = mk(0.5f * cellsz.x + bp.pad, 0.5f * cellsz.y + bp.pad, 0.5f * cellsz.z + bp.pad);
        exact = true;
    }
    for (int z = c0[2]; z <= c1[2]; z++)
        for (int y = c0[1]; y <= c1[1]; y++)
            for (int x = c0[0]; x <= c1[0]; x++) {
                if (exact) {
                    const f3 centre = mk(bp.lo.x + ((float)x + 0.5f) * cellsz.x, bp.lo.y + ((float)y + 0.5f) * cellsz.y,
                                         bp.lo.z + ((float)z + 0.5f) * cellsz.z);
                    if (!tri_box_overlap(centre, half, wa, wb, wc)) continue;
                }
                int cell;
                if (bp.sub > 0) {
                    const int cx = x / bp.sub, cy = y / bp.sub, cz = z / bp.sub;
                    const int coarse = cx + (bp.rx / bp.sub) * (cy + (bp.ry / bp.sub) * cz);
                    cell = coarse * bp.cpb + (x - cx * bp.sub) + bp.sub * ((y - cy * bp.sub) + bp.sub * (z - cz * bp.sub));
                } else {
                    cell = x + bp.rx * (y + bp.ry * z);
                }
                const int k = atomicAdd(count_or_cursor + cell, 1);
                if (FILL) ids[start[cell] + k] = i;
            }
}

// Exclusive prefix sum over the cell counts, one workgroup walking the array in 1024-wide strips.
// start has n + 1 entries; start[n] = total.
__global__ __launch_bounds__(1024) void k_scan(int n, const int* count, int* start) {
    __shared__ int buf[1024];
    __shared__ int carry;
    const int tid = (int)threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int v = i < n ? count[i] : 0;
        buf[tid] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int add = tid >= off ? buf[tid - off] : 0;
            __syncthreads();
            buf[tid] += add;
            __syncthreads();
        }
        if (i < n) start[i] = carry + buf[tid] - v;
        __syncthreads();
        if (tid == 1023) carry += buf[1023];
        __syncthreads();
    }
    if (tid == 0) start[n] = carry;
}

// Put every cell's ids in ascending (= file) order.  Short lists: one lane per cell, insertion sort.
__global__ void k_sort_small(int n_cells, const int* start, int* ids, int max_len) {
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (c >= n_cells) return;
    const int b = start[c], e = start[c + 1];
    if (e - b < 2 || e - b > max_len) return;
    for (int i = b + 1; i < e; i++) {
        const int v = ids[i];
        int j = i - 1;
        while (j >= b && ids[j] > v) { ids[j + 1] = ids[j]; j--; }
        ids[j + 1] = v;
    }
}
// Medium lists (up to RTX_SORT_LDS_MAX ids): one workgroup per cell, bitonic network over an LDS copy padded to a
// power of two with INT_MAX.
#define RTX_SORT_LDS_MAX 16384
__global__ __launch_bounds__(256) void k_sort_lds(int n_cells, const int* start, int* ids, int min_len) {
    __shared__ int buf[RTX_SORT_LDS_MAX];
    const int c = (int)blockIdx.x;
    if (c >= n_cells) return;
    const int b = start[c], e = start[c + 1], len = e - b;
    if (len <= min_len || len > RTX_SORT_LDS_MAX) return;      // uniform over the workgroup
    int padded = 64;
    while (padded < len) padded <<= 1;
    for (int i = (int)threadIdx.x; i < padded; i += 256) buf[i] = i < len ? ids[b + i] : 0x7fffffff;
    __syncthreads();
    for (int k = 2; k <= padded; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = (int)threadIdx.x; t < (padded >> 1); t += 256) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // index with bit j clear
                const int hi = lo | j;
                const bool up = (lo & k) == 0;
                const int x = buf[lo], y = buf[hi];
                if ((x > y) == up) { buf[lo] = y; buf[hi] = x; }
            }
            __syncthreads();
        }
    }
    for (int i = (int)threadIdx.x; i < len; i += 256) ids[b + i] = buf[i];
}
// Longer lists: ids within a cell are distinct, so an element's final position is the number of smaller elements
// (rank sort through a scratch copy, quadratic: only for lists that do not fit the LDS sort).
__global__ __launch_bounds__(256) void k_sort_large(int n_cells, const int* start, int* ids, int* scratch, int min_len) {
    const int c = (int)blockIdx.x;
    if (c >= n_cells) return;
    const int b = start[c], e = start[c + 1], len = e - b;
    if (len <= min_len) return;
    for (int i = (int)threadIdx.x; i < len; i += 256) scratch[b + i] = ids[b + i];
    __syncthreads();
    for (int i = (int)threadIdx.x; i < len; i += 256) {
        const int v = scratch[b + i];
        int rank = 0;
        for (int j = 0; j < len; j++) rank += scratch[b + j] < v ? 1 : 0;
        ids[b + rank] = v;
    }
}

// Per-reference copies of the primitive records in list order (nested grid): frecs[3j..3j+2] = record of fids[j]
// with the primitive index in the slot that holds the material in the table.
__global__ void k_expand_records(int n_refs, const int* fids, const float4* prims, float4* frecs) {
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= n_refs) return;
    const int pid = fids[j];
    const float4 q2 = prims[3 * pid + 2];
    frecs[3 * (size_t)j] = prims[3 * pid];
    frecs[3 * (size_t)j + 1] = prims[3 * pid + 1];
    frecs[3 * (size_t)j + 2] = make_float4(q2.x, q2.y, as_float(pid), q2.w);
}

// Sparse view of the nested grid, step 1: occupancy mask of every 32 fine cells and its population count
__global__ void k_words(int n_words, const int* fstart, unsigned* mask, int* pop) {
    const int w = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (w >= n_words) return;
    unsigned m = 0;
    for (int b = 0; b < 32; b++) m |= (fstart[32 * w + b + 1] > fstart[32 * w + b]) ? (1u << b) : 0u;
    mask[w] = m;
    pop[w] = __popc(m);
}
// step 2 (after the prefix sum of pop): (mask, occupied cells before the word) pairs and the compact CSR starts
__global__ void k_compact(int n_words, const int* fstart, const unsigned* mask, const int* wpre, uint2* words, int* ostart) {
    const int w = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (w >= n_words) return;
    const unsigned m = mask[w];
    int o = wpre[w];
    words[w] = make_uint2(m, (unsigned)o);
    for (int b = 0; b < 32; b++)
        if (m & (1u << b)) ostart[o++] = fstart[32 * w + b];
    if (w == n_words - 1) ostart[wpre[n_words]] = fstart[32 * n_words];
}
// coarse-cell occupancy of the nested grid: any fine cell of the block holds a primitive
__global__ void k_occupancy(int n_blocks, int words_per_block, const unsigned* mask, unsigned char* occ) {
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (c >= n_blocks) return;
    unsigned any = 0;
    for (int w = 0; w < words_per_block; w++) any |= mask[c * words_per_block + w];
    occ[c] = any ? 1 : 0;
}
// ... and the same packed 32 coarse cells per word
__global__ void k_occupancy_bits(int n_blocks, const unsigned char* occ, unsigned* bits) {
    const int w = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (w >= (n_blocks + 31) / 32) return;
    unsigned m = 0;
    for (int b = 0; b < 32 && 32 * w + b < n_blocks; b++) m |= occ[32 * w + b] ? (1u << b) : 0u;
    bits[w] = m;
}

// ---------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
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
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

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
    DevBuf<int> cstart, cids, fstart, fids, scratch_ids, cell_tmp;
    DevBuf<unsigned char> cocc;
    DevBuf<unsigned> wmask, cocc_bits;
    DevBuf<int> wpop, wpre, ostart;
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
    DevBuf<unsigned char> tile_map;
    DevBuf<signed char> tile_class;
    DevBuf<int> tile_list, tile_cost[2];
    int cost_key[6] = {0, 0, 0, 0, 0, 0};   // W, H, tile_rows, rank, world, grid generation the cost map belongs to
    int cost_cur = 0;             // tile_cost[cost_cur] was written by the last frame
    bool cost_valid = false;
    int grid_generation = 0;
    float fine_pad = 0.0f;        // margin the nested lists were binned with
    DevBuf<float> rgb_local;
    DevBuf<int> ids_local;
    hipEvent_t ev[4 + 3 * 2 * RTX_MAX_LEVELS];
    bool have_events = false;
};

namespace {

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

template <int MODE, bool SINGLE_XF, int PRIMS>
void launch_frame3(rtx_ctx* c, const DevScene& sc, const DevFrame& f, int blocks, hipStream_t st, bool count, int levels) {
    float4* ps = c->path_stack.p;
    if (count) hipLaunchKernelGGL((k_frame<MODE, SINGLE_XF, true, PRIMS>), dim3(blocks), dim3(RTX_BLOCK), 0, st, sc, c->dgrid, c->dsparse, f, ps, levels);
    else hipLaunchKernelGGL((k_frame<MODE, SINGLE_XF, false, PRIMS>), dim3(blocks), dim3(RTX_BLOCK), 0, st, sc, c->dgrid, c->dsparse, f, ps, levels);
}

// picks the kernel specialisation for the scene: one transform or many, triangles only / spheres only / both
template <int MODE>
void launch_frame(rtx_ctx* c, const DevScene& sc, const DevFrame& f, int blocks, hipStream_t st, bool count, int levels) {
    const bool single = c->n_xforms == 1;
    const int prims = c->n_spheres == 0 ? 1 : (c->n_triangles == 0 ? 2 : 0);
    if (single) {
        if (prims == 1) launch_frame3<MODE, true, 1>(c, sc, f, blocks, st, count, levels);
        else if (prims == 2) launch_frame3<MODE, true, 2>(c, sc, f, blocks, st, count, levels);
        else launch_frame3<MODE, true, 0>(c, sc, f, blocks, st, count, levels);
    } else {
        if (prims == 1) launch_frame3<MODE, false, 1>(c, sc, f, blocks, st, count, levels);
        else if (prims == 2) launch_frame3<MODE, false, 2>(c, sc, f, blocks, st, count, levels);
        else launch_frame3<MODE, false, 0>(c, sc, f, blocks, st, count, levels);
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
    *out = c;
    return RTX_OK;
}

void rtx_destroy(rtx_ctx* c) {
    if (!c) return;
    DeviceGuard guard(c->device);
    (void)hipDeviceSynchronize();
    if (c->have_events)
        for (auto& ev : c->ev) (void)hipEventDestroy(ev);
    c->prims.release(); c->tri_normal.release(); c->xinv.release(); c->xfwd.release(); c->mats.release();
    c->lights.release(); c->geom.release(); c->kind.release(); c->xform.release();
    c->ref_box.release(); c->tight_box.release(); c->bounds.release();
    c->cstart.release(); c->cids.release(); c->fstart.release(); c->fids.release(); c->scratch_ids.release();
    c->cell_tmp.release(); c->cocc.release();
    c->wmask.release(); c->cocc_bits.release(); c->wpop.release(); c->wpre.release(); c->ostart.release(); c->words.release();
    c->counters.release(); c->work.release(); c->hP.release(); c->hN.release(); c->hD.release(); c->pix0.release();
    for (int k = 0; k < 2; k++) { c->rO[k].release(); c->rD[k].release(); }
    for (int l = 0; l < RTX_MAX_LEVELS; l++) { c->hL[l].release(); c->hChild[l].release(); }
    c->path_stack.release();
    c->rgb_local.release(); c->ids_local.release();
    delete c;
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
    if (n > 0)
        hipLaunchKernelGGL((k_bin<false>), dim3(pb), dim3(256), 0, 0, n, bp, boxes, c->cell_tmp.p, (const int*)nullptr, (int*)nullptr, eg,
                           (const int*)c->kind.p, (const int*)c->xform.p, (const float*)c->xfwd.p);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, 0, n_cells, c->cell_tmp.p, start.p);
    int total = 0;
    HIP_TRY(hipMemcpy(&total, start.p + n_cells, sizeof(int), hipMemcpyDeviceToHost));
    if (total < 0) { rtx_set_error("cell reference count overflowed 2^31"); return RTX_E_LIMIT; }
    if ((rc = ids.reserve((size_t)std::max(total, 1))) || (rc = c->scratch_ids.reserve((size_t)std::max(total, 1)))) return rc;
    HIP_TRY(hipMemsetAsync(c->cell_tmp.p, 0, ((size_t)n_cells + 1) * sizeof(int), 0));
    if (n > 0)
        hipLaunchKernelGGL((k_bin<true>), dim3(pb), dim3(256), 0, 0, n, bp, boxes, c->cell_tmp.p, (const int*)start.p, ids.p, eg,
                           (const int*)c->kind.p, (const int*)c->xform.p, (const float*)c->xfwd.p);
    hipLaunchKernelGGL(k_sort_small, dim3((n_cells + 255) / 256), dim3(256), 0, 0, n_cells, (const int*)start.p, ids.p, 24);
    hipLaunchKernelGGL(k_sort_lds, dim3(n_cells), dim3(256), 0, 0, n_cells, (const int*)start.p, ids.p, 24);
    hipLaunchKernelGGL(k_sort_large, dim3(n_cells), dim3(256), 0, 0, n_cells, (const int*)start.p, ids.p, c->scratch_ids.p, RTX_SORT_LDS_MAX);
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
            // aim at a few primitives per occupied fine cell: total fine cells ~ 8 x primitive count
            const double target = std::cbrt(8.0 * std::max(n, 1) / ((double)rx * ry * rz));
            sub = (int)std::min(32.0, std::max(1.0, std::ceil(target)));
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
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, 0, n_words, (const int*)c->wpop.p, c->wpre.p);
        int n_occ = 0;
        HIP_TRY(hipMemcpy(&n_occ, c->wpre.p + n_words, sizeof(int), hipMemcpyDeviceToHost));
        if ((rc = c->ostart.reserve((size_t)n_occ + 1))) return rc;
        hipLaunchKernelGGL(k_compact, dim3((n_words + 255) / 256), dim3(256), 0, 0, n_words, (const int*)c->fstart.p, (const unsigned*)c->wmask.p,
                           (const int*)c->wpre.p, c->words.p, c->ostart.p);
        hipLaunchKernelGGL(k_occupancy, dim3((n_blocks + 255) / 256), dim3(256), 0, 0, n_blocks, wpb, (const unsigned*)c->wmask.p, c->cocc.p);
        if ((rc = c->cocc_bits.reserve((size_t)(n_blocks + 31) / 32))) return rc;
        hipLaunchKernelGGL(k_occupancy_bits, dim3(((n_blocks + 31) / 32 + 255) / 256), dim3(256), 0, 0, n_blocks, (const unsigned char*)c->cocc.p, c->cocc_bits.p);
        g.cocc_bits = c->cocc_bits.p;
        c->dsparse.words = c->words.p;
        c->dsparse.ostart = c->ostart.p;
        c->dsparse.words_per_block = wpb;
        g.sub = sub;
        g.cpb = 32 * wpb;
        g.frx = frx; g.fry = fry; g.frz = frz;
        g.fcell = mk(g.size.x / (float)frx, g.size.y / (float)fry, g.size.z / (float)frz);
        g.inv_fcell = mk((float)frx / g.size.x, (float)fry / g.size.y, (float)frz / g.size.z);
        g.fstart = c->fstart.p;
        g.fids = c->fids.p;
        if ((rc = c->frecs.reserve(3 * (size_t)std::max<int64_t>(gi.n_fine_refs, 1)))) return rc;
        if (gi.n_fine_refs > 0)
            hipLaunchKernelGGL(k_expand_records, dim3((unsigned)((gi.n_fine_refs + 255) / 256)), dim3(256), 0, 0, (int)gi.n_fine_refs,
                               (const int*)c->fids.p, (const float4*)c->prims.p, c->frecs.p);
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

int rtx_render_device(rtx_ctx* c, const rtx_camera* cam, const rtx_render_params* p, float* d_rgb, int32_t* d_ids,
                      void* stream, rtx_stats* stats) {
    if (!c || !cam || !d_rgb) { rtx_set_error("rtx_render_device: null argument"); return RTX_E_INVALID; }
    int tile_rows, rank, world;
    int rc = check_params(p, &tile_rows, &rank, &world);
    if (rc) return rc;
    if (!c->have_scene) { rtx_set_error("render: no scene set"); return RTX_E_STATE; }
    if (p->search == RTX_SEARCH_REFERENCE_LISTS && !c->have_ref) { rtx_set_error("render: reference cell lists not built"); return RTX_E_STATE; }
    if (p->search == RTX_SEARCH_NESTED && !c->have_nested) { rtx_set_error("render: nested grid not built"); return RTX_E_STATE; }
    DeviceGuard guard(c->device);
    if (!guard.ok) { rtx_set_error("hipSetDevice(%d) failed", c->device); return RTX_E_DEVICE; }
    hipStream_t st = (hipStream_t)stream;

    const int nl = count_local_rows(p->height, tile_rows, rank, world);
    const int levels = p->max_depth + 1;
    const size_t pixels = (size_t)nl * p->width;
    const bool serial = (p->flags & RTX_RENDER_SERIAL_LANES) != 0;
    const bool level_queues = serial || (p->flags & RTX_RENDER_LEVEL_QUEUES) != 0;
    int blocks = c->n_cus * 4;   // 3 workgroups of 4 waves fit a CU at a time: the work cursors balance the rest
    if (p->workgroups_per_cu > 0) blocks = c->n_cus * std::min(p->workgroups_per_cu, 16);
    if (const char* e = getenv("RTX_BLOCKS_PER_CU")) blocks = c->n_cus * std::max(1, atoi(e));   // tuning knob
    if (level_queues) {
        if ((rc = ensure_workspace(c, std::max(pixels, (size_t)1), levels))) return rc;
    } else {
        if ((rc = c->counters.reserve(CNT_TOTAL)) || (rc = c->work.reserve(24)) ||
            (rc = c->path_stack.reserve((size_t)blocks * RTX_BLOCK * levels)))
            return rc;
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
    const int tbx = (p->width + 7) / 8, tby = (p->height + 7) / 8;
    const int n_row_blocks = f.n_items0 > 0 ? (lr_end - f.lr0 + 7) / 8 : 0;
    bool tile_cull = !level_queues && p->search == RTX_SEARCH_NESTED && f.n_items0 > 0 && tbx <= 0xffff && (nl + 7) / 8 <= 0x7fff;
    if (const char* e = getenv("RTX_TILE_CULL")) tile_cull = tile_cull && atoi(e) != 0;          // A/B knob
    if (tile_cull) {
        const size_t n_slots = (size_t)tbx * ((nl + 7) / 8);
        if ((rc = c->tile_map.reserve((size_t)tbx * tby)) || (rc = c->tile_list.reserve((size_t)n_row_blocks * f.nbx)) ||
            (rc = c->tile_class.reserve((size_t)n_row_blocks * f.nbx)) || (rc = c->tile_cost[0].reserve(n_slots)) ||
            (rc = c->tile_cost[1].reserve(n_slots)))
            return rc;
        f.tile_list = c->tile_list.p;
        f.tbx = tbx;
        f.hint_radius = 1;
        if (const char* e = getenv("RTX_HINT_RADIUS")) f.hint_radius = std::max(0, std::min(16, atoi(e)));
        whole_frame = false;
    }
    f.max_depth = p->max_depth;
    f.bias = p->eps_bias;
    f.walk_budget = 3;
    f.ready_threshold = 48;
    f.drain_num = 2;
    if (const char* e = getenv("RTX_DRAIN_NUM")) f.drain_num = std::max(0, atoi(e));
    if (const char* e = getenv("RTX_WALK_BUDGET")) f.walk_budget = std::max(1, atoi(e));          // tuning knobs
    if (const char* e = getenv("RTX_READY_THRESHOLD")) f.ready_threshold = std::max(1, atoi(e));
    f.rgb = d_rgb;
    f.ids = d_ids;
    f.counters = c->counters.p;
    f.work = c->work.p;
    if (level_queues) { f.hP = c->hP.p; f.hN = c->hN.p; f.hD = c->hD.p; f.pix0 = c->pix0.p; }

    const DevScene sc = dev_scene(c);
    const bool count = (p->flags & RTX_RENDER_COUNT_WORK) != 0;
    const bool timed = stats != nullptr;

    hipEvent_t* ev = c->ev;
    HIP_TRY(hipEventRecord(ev[0], st));              // frame time includes the fills
    if (!whole_frame) {
        HIP_TRY(hipMemsetAsync(d_rgb, 0, pixels * 3 * sizeof(float), st));            // 0.0f
        if (d_ids) HIP_TRY(hipMemsetAsync(d_ids, 0xff, pixels * sizeof(int32_t), st));   // -1
    }
    HIP_TRY(hipMemsetAsync(c->counters.p, 0, CNT_TOTAL * sizeof(int), st));
    if (count) HIP_TRY(hipMemsetAsync(c->work.p, 0, 24 * sizeof(unsigned long long), st));
    if (tile_cull) {
        HIP_TRY(hipMemsetAsync(c->tile_map.p, 0, (size_t)tbx * tby, st));
        const int n_fine = (int)c->ginfo.n_fine_cells;
        hipLaunchKernelGGL(k_tile_mark, dim3((n_fine + 255) / 256), dim3(256), 0, st, f.cam, c->dgrid, c->dsparse, c->fine_pad, n_fine, tbx, tby,
                           c->tile_map.p, c->counters.p);
        // the previous frame's per-block path depths order this frame's list if they were recorded for the same
        // image geometry, rank split and grid (the camera may have moved: the order is a hint, never a result)
        const int key[6] = {p->width, p->height, tile_rows, rank, world, c->grid_generation};
        bool feedback = c->cost_valid && memcmp(key, c->cost_key, sizeof key) == 0;
        if (const char* e = getenv("RTX_TILE_FEEDBACK")) feedback = feedback && atoi(e) != 0;     // A/B knob
        const int* prev_cost = feedback ? c->tile_cost[c->cost_cur].p : nullptr;
        c->cost_cur ^= 1;
        f.tile_cost = c->tile_cost[c->cost_cur].p;
        memcpy(c->cost_key, key, sizeof key);
        c->cost_valid = true;
        HIP_TRY(hipMemsetAsync(f.tile_cost, 0, (size_t)tbx * ((nl + 7) / 8) * sizeof(int), st));
        const int n_blocks_window = n_row_blocks * f.nbx;
        hipLaunchKernelGGL(k_tile_classify, dim3((n_blocks_window + 255) / 256), dim3(256), 0, st, f, n_row_blocks,
                           (const unsigned char*)c->tile_map.p, prev_cost, c->tile_class.p);
        hipLaunchKernelGGL(k_tile_fill, dim3((n_blocks_window + 255) / 256), dim3(256), 0, st, f, n_row_blocks,
                           (const signed char*)c->tile_class.p, c->tile_list.p);
    }
    HIP_TRY(hipEventRecord(ev[3], st));              // start of the kernels proper
    if (!level_queues) {
        if (p->search == RTX_SEARCH_REFERENCE_LISTS) launch_frame<RT_SEARCH_REF>(c, sc, f, blocks, st, count, levels);
        else launch_frame<RT_SEARCH_NESTED>(c, sc, f, blocks, st, count, levels);
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

    if (stats) {
        HIP_TRY(hipEventSynchronize(ev[2]));
        int cnt[CNT_TOTAL];
        HIP_TRY(hipMemcpyAsync(cnt, c->counters.p, sizeof cnt, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        stats->primary_rays = (int64_t)pixels;
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
            (void)hipEventElapsedTime(&stats->resolve_ms, ev[1], ev[2]);
        } else {
            stats->trace_launches = 1;                 // the persistent frame kernel does everything
            (void)hipEventElapsedTime(&stats->trace_ms, ev[3], ev[1]);
        }
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

int rtx_render(rtx_ctx* c, const rtx_camera* cam, const rtx_render_params* p, float* rgb_out, int32_t* prim_id_out, rtx_stats* stats) {
    if (!c || !cam || !rgb_out) { rtx_set_error("rtx_render: null argument"); return RTX_E_INVALID; }
    int tile_rows, rank, world;
    int rc = check_params(p, &tile_rows, &rank, &world);
    if (rc) return rc;
    DeviceGuard guard(c->device);
    if (!guard.ok) { rtx_set_error("hipSetDevice(%d) failed", c->device); return RTX_E_DEVICE; }
    const int nl = count_local_rows(p->height, tile_rows, rank, world);
    const size_t pixels = (size_t)nl * p->width;
    if ((rc = c->rgb_local.reserve(std::max(pixels, (size_t)1) * 3))) return rc;
    if (prim_id_out && (rc = c->ids_local.reserve(std::max(pixels, (size_t)1)))) return rc;
    rtx_stats local_stats;
    rc = rtx_render_device(c, cam, p, c->rgb_local.p, prim_id_out ? c->ids_local.p : nullptr, nullptr, stats ? stats : &local_stats);
    if (rc) return rc;
    if (nl == 0) return RTX_OK;
    std::vector<int32_t> rows;
    std::vector<float> rgb;
    std::vector<int32_t> ids;
    try {
        rows.resize(nl);
        rgb.resize(pixels * 3);
        if (prim_id_out) ids.resize(pixels);
    } catch (const std::bad_alloc&) {
        return RTX_E_NOMEM;
    }
    int n_rows = 0;
    rtx_local_rows(p, &n_rows, rows.data());
    HIP_TRY(hipMemcpy(rgb.data(), c->rgb_local.p, pixels * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (prim_id_out) HIP_TRY(hipMemcpy(ids.data(), c->ids_local.p, pixels * sizeof(int), hipMemcpyDeviceToHost));
    const size_t W = (size_t)p->width;
    for (int r = 0; r < nl; r++) {
        memcpy(rgb_out + (size_t)rows[r] * W * 3, rgb.data() + (size_t)r * W * 3, W * 3 * sizeof(float));
        if (prim_id_out) memcpy(prim_id_out + (size_t)rows[r] * W, ids.data() + (size_t)r * W, W * sizeof(int32_t));
    }
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
