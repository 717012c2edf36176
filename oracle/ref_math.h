// TEST INFRASTRUCTURE ONLY -- the CPU oracle's vector/matrix arithmetic.
//
// The reference does all of its low-level arithmetic through the third-party
// header library glm (g-truc/glm; the reference pins no version and does not
// vendor it -- /root/reference/CMakeLists.txt:7-8 lists only OpenCV and OpenMP,
// the includes are /root/reference/src/geometries.h:16-19).  glm is absent from
// this image, so this file restates the published scalar (non-SIMD) algorithms
// of glm 0.9.9.x for exactly the entry points the reference calls, with the
// operation ORDER glm uses, because hit/miss decisions depend on the rounding
// of every intermediate.  PARITY UNPINNED at this level: the reference holds no
// test or golden vector for any of these functions; nothing in the image can
// confirm the restatement bit for bit.
//
// Every function is plain IEEE fp32, one rounding per operation; the oracle is
// built with -ffp-contract=off and without -march so no fused multiply-add is
// ever formed (the reference's own build, CMakeLists.txt:5, has no -march
// either, so x86-64 baseline SSE2 scalar code is what defines its results).
#pragma once
#include <cmath>
#include <cfloat>

namespace refm {

struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}                     // glm value-initialises vec3() to 0
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
struct V4 {
    float x, y, z, w;
    V4() : x(0), y(0), z(0), w(0) {}
    V4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    V4(V3 v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};

// component-wise operators (glm type_vec3.inl / type_vec4.inl)
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(V3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }   // true divides, no reciprocal
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline V3& operator+=(V3& a, V3 b) { a = a + b; return a; }

inline V4 operator+(V4 a, V4 b) { return V4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline V4 operator-(V4 a, V4 b) { return V4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline V4 operator*(V4 a, V4 b) { return V4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline V4 operator*(V4 a, float s) { return V4(a.x * s, a.y * s, a.z * s, a.w * s); }

// glm::min(x,y) = (y < x) ? y : x ; glm::max(x,y) = (x < y) ? y : x  (func_common.inl)
inline float gmin(float x, float y) { return (y < x) ? y : x; }
inline float gmax(float x, float y) { return (x < y) ? y : x; }
inline V3 gmin(V3 a, V3 b) { return V3(gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)); }
inline V3 gmax(V3 a, V3 b) { return V3(gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)); }

// glm::dot(vec3): tmp = a*b; tmp.x + tmp.y + tmp.z   (func_geometric.inl compute_dot)
inline float dot(V3 a, V3 b) { V3 t = a * b; return t.x + t.y + t.z; }
// glm::cross (func_geometric.inl compute_cross)
inline V3 cross(V3 a, V3 b) {
    return V3(a.y * b.z - b.y * a.z,
              a.z * b.x - b.z * a.x,
              a.x * b.y - b.x * a.y);
}
// glm::normalize(v) = v * inversesqrt(dot(v,v)), inversesqrt(x) = 1 / sqrt(x)
inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }

// column-major 4x4, c[j] is column j (glm::mat4)
struct M4 {
    V4 c[4];
    M4() {}
    explicit M4(float d) { c[0] = V4(d, 0, 0, 0); c[1] = V4(0, d, 0, 0); c[2] = V4(0, 0, d, 0); c[3] = V4(0, 0, 0, d); }
    V4& operator[](int j) { return c[j]; }
    const V4& operator[](int j) const { return c[j]; }
};

// mat4 * vec4 (type_mat4x4.inl): (m0*x + m1*y) + (m2*z + m3*w)
inline V4 operator*(const M4& m, V4 v) {
    V4 mul0 = m[0] * v.x, mul1 = m[1] * v.y;
    V4 add0 = mul0 + mul1;
    V4 mul2 = m[2] * v.z, mul3 = m[3] * v.w;
    V4 add1 = mul2 + mul3;
    return add0 + add1;
}
// mat4 * mat4: column j = ((A0*b0 + A1*b1) + A2*b2) + A3*b3
inline M4 operator*(const M4& a, const M4& b) {
    M4 r;
    for (int j = 0; j < 4; j++)
        r[j] = a[0] * b[j].x + a[1] * b[j].y + a[2] * b[j].z + a[3] * b[j].w;
    return r;
}
inline M4 operator*(const M4& m, float s) {
    M4 r;
    for (int j = 0; j < 4; j++) r[j] = m[j] * s;
    return r;
}
inline M4 transpose(const M4& m) {
    M4 r;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) r[j][i] = m[i][j];
    return r;
}

// glm::inverse(mat4) -- cofactor expansion, func_matrix.inl compute_inverse<4,4>
inline M4 inverse(const M4& m) {
    float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];

    float c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];

    float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];

    float c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];

    float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];

    float c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];

    V4 fac0(c00, c00, c02, c03), fac1(c04, c04, c06, c07), fac2(c08, c08, c10, c11);
    V4 fac3(c12, c12, c14, c15), fac4(c16, c16, c18, c19), fac5(c20, c20, c22, c23);

    V4 vec0(m[1][0], m[0][0], m[0][0], m[0][0]);
    V4 vec1(m[1][1], m[0][1], m[0][1], m[0][1]);
    V4 vec2(m[1][2], m[0][2], m[0][2], m[0][2]);
    V4 vec3(m[1][3], m[0][3], m[0][3], m[0][3]);

    V4 inv0 = vec1 * fac0 - vec2 * fac1 + vec3 * fac2;
    V4 inv1 = vec0 * fac0 - vec2 * fac3 + vec3 * fac4;
    V4 inv2 = vec0 * fac1 - vec1 * fac3 + vec3 * fac5;
    V4 inv3 = vec0 * fac2 - vec1 * fac4 + vec2 * fac5;

    V4 signA(+1, -1, +1, -1), signB(-1, +1, -1, +1);
    M4 inv;
    inv[0] = inv0 * signA; inv[1] = inv1 * signB; inv[2] = inv2 * signA; inv[3] = inv3 * signB;

    V4 row0(inv[0][0], inv[1][0], inv[2][0], inv[3][0]);
    V4 dot0 = m[0] * row0;
    float dot1 = (dot0.x + dot0.y) + (dot0.z + dot0.w);
    float oneOverDet = 1.0f / dot1;
    return inv * oneOverDet;
}

// glm::translate(m, v) (ext/matrix_transform.inl): m[3] = m0*v0 + m1*v1 + m2*v2 + m3, left to right
inline M4 translate(const M4& m, V3 v) {
    M4 r = m;
    r[3] = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3];
    return r;
}
// glm::rotate(m, angle, axis) (ext/matrix_transform.inl); gtx/transform.hpp's
// rotate(angle, axis) is this applied to mat4(1)
inline M4 rotate(const M4& m, float angle, V3 v) {
    float a = angle, c = std::cos(a), s = std::sin(a);
    V3 axis = normalize(v);
    V3 temp = (1.0f - c) * axis;
    float r00 = c + temp[0] * axis[0];
    float r01 = temp[0] * axis[1] + s * axis[2];
    float r02 = temp[0] * axis[2] - s * axis[1];
    float r10 = temp[1] * axis[0] - s * axis[2];
    float r11 = c + temp[1] * axis[1];
    float r12 = temp[1] * axis[2] + s * axis[0];
    float r20 = temp[2] * axis[0] + s * axis[1];
    float r21 = temp[2] * axis[1] - s * axis[0];
    float r22 = c + temp[2] * axis[2];
    M4 r;
    r[0] = m[0] * r00 + m[1] * r01 + m[2] * r02;
    r[1] = m[0] * r10 + m[1] * r11 + m[2] * r12;
    r[2] = m[0] * r20 + m[1] * r21 + m[2] * r22;
    r[3] = m[3];
    return r;
}
// glm::scale(m, v): columns 0..2 scaled, column 3 kept; gtx scale(v) applies it to mat4(1)
inline M4 scale(const M4& m, V3 v) {
    M4 r;
    r[0] = m[0] * v.x; r[1] = m[1] * v.y; r[2] = m[2] * v.z; r[3] = m[3];
    return r;
}

static const float PI_F = 3.14159265358979323846264338327950288f;   // glm::pi<float>()

}  // namespace refm
