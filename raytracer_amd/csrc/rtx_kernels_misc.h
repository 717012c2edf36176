// 8-bit conversion of a framebuffer, single-ray probes and camera-ray evaluation (tests).
// Part of the device layer of librtx; included by rtx_device.hip only (one translation unit).
#pragma once

// ---------------------------------------------------------------------------------------------------
// Start of a frame in one launch: the miss colour (0) over the colour buffer, -1 over the id map, zeros over the
// counters and the tile maps.  Up to RTX_CLEAR_RANGES (pointer, 16-byte words, 32-bit pattern) ranges, each a multiple of 16
// bytes long and 16-byte aligned (device allocations are; the host rounds the lengths up inside the allocations).
// ---------------------------------------------------------------------------------------------------
#define RTX_CLEAR_RANGES 8
struct ClearRanges {
    uint4* ptr[RTX_CLEAR_RANGES];
    long long words[RTX_CLEAR_RANGES];
    unsigned pattern[RTX_CLEAR_RANGES];
    int n;
};
__global__ __launch_bounds__(256) void k_frame_clear(ClearRanges r) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    for (int k = 0; k < r.n; k++) {
        const uint4 v = make_uint4(r.pattern[k], r.pattern[k], r.pattern[k], r.pattern[k]);
        uint4* p = r.ptr[k];
        for (long long i = tid; i < r.words[k]; i += stride) p[i] = v;
    }
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
