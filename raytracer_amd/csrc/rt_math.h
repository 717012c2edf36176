// fp32 vector / matrix arithmetic of the renderer, single source for host and gfx950 device code.
//
// Hit/miss decisions and cell walks depend on the rounding of every intermediate, so each helper
// fixes an operation order (the order of the glm calls the reference makes, e.g. normalize at
// src/geometries.h:28,350 of the reference tree, dot at :226-227, mat4*vec4 at :31,35) and the code
// is compiled with -ffp-contract=off: one IEEE rounding per written operation, no fused multiply-add.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt {

struct f3 {
    float x, y, z;
};

RT_HD f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
RT_HD f3 operator*(float s, f3 a) { return mk(s * a.x, s * a.y, s * a.z); }
RT_HD f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
RT_HD f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
RT_HD float comp(f3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// select-style min/max: min(x,y) = (y < x) ? y : x ; max(x,y) = (x < y) ? y : x
RT_HD float sel_min(float x, float y) { return (y < x) ? y : x; }
RT_HD float sel_max(float x, float y) { return (x < y) ? y : x; }
RT_HD f3 sel_min(f3 a, f3 b) { return mk(sel_min(a.x, b.x), sel_min(a.y, b.y), sel_min(a.z, b.z)); }
RT_HD f3 sel_max(f3 a, f3 b) { return mk(sel_max(a.x, b.x), sel_max(a.y, b.y), sel_max(a.z, b.z)); }
// std::max({a,b,c}) / std::min({a,b,c,..}): first extreme element wins
RT_HD float max_of3(float a, float b, float c) { float m = a; if (m < b) m = b; if (m < c) m = c; return m; }
RT_HD float min_of3(float a, float b, float c) { float m = a; if (b < m) m = b; if (c < m) m = c; return m; }
RT_HD float min_of4(float a, float b, float c, float d) { float m = min_of3(a, b, c); if (d < m) m = d; return m; }

// clamp(val, lo, hi) = max(min(val, hi), lo) with the select semantics above (src/Image3f.h:15-17)
RT_HD float clampf(float v, float lo, float hi) { return sel_max(sel_min(v, hi), lo); }
RT_HD float clamp01(float v) { return clampf(v, 0.0f, 1.0f); }
RT_HD f3 clamp_rgb(f3 c) { return mk(clamp01(c.x), clamp01(c.y), clamp01(c.z)); }   // src/Image3f.h:19-21

RT_HD float dot3(f3 a, f3 b) { float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }
RT_HD f3 cross3(f3 a, f3 b) {
    return mk(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
RT_HD f3 normalize3(f3 v) { return v * (1.0f / sqrtf(dot3(v, v))); }

// 4x4 column-major matrix as 16 floats: m[4*col + row]
struct m44 {
    float m[16];
};
RT_HD float at(const m44& a, int col, int row) { return a.m[4 * col + row]; }

// rows 0..2 of M * (v, w): per row (m0*x + m1*y) + (m2*z + m3*w)
RT_HD f3 mul_point(const float* m, f3 v) {   // w = 1
    return mk((m[0] * v.x + m[4] * v.y) + (m[8] * v.z + m[12] * 1.0f),
              (m[1] * v.x + m[5] * v.y) + (m[9] * v.z + m[13] * 1.0f),
              (m[2] * v.x + m[6] * v.y) + (m[10] * v.z + m[14] * 1.0f));
}
RT_HD f3 mul_dir(const float* m, f3 v) {     // w = 0; the 0-products stay (they carry signs of zero, NaN, inf)
    return mk((m[0] * v.x + m[4] * v.y) + (m[8] * v.z + m[12] * 0.0f),
              (m[1] * v.x + m[5] * v.y) + (m[9] * v.z + m[13] * 0.0f),
              (m[2] * v.x + m[6] * v.y) + (m[10] * v.z + m[14] * 0.0f));
}
// rows 0..2 of transpose(inv) * (n, 0), normalised: the world-space normal (src/geometries.h:27-29)
RT_HD f3 mul_normal(const float* inv, f3 n) {
    f3 r = mk((inv[0] * n.x + inv[1] * n.y) + (inv[2] * n.z + inv[3] * 0.0f),
              (inv[4] * n.x + inv[5] * n.y) + (inv[6] * n.z + inv[7] * 0.0f),
              (inv[8] * n.x + inv[9] * n.y) + (inv[10] * n.z + inv[11] * 0.0f));
    return normalize3(r);
}

RT_HD int as_int(float f) {
    union { float f; int i; } u;
    u.f = f;
    return u.i;
}
RT_HD float as_float(int i) {
    union { float f; int i; } u;
    u.i = i;
    return u.f;
}

}  // namespace rt
