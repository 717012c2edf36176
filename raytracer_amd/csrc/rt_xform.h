// Host-side 4x4 transform algebra for the scene layer: the transform stack of the .test format
// (translate / rotate / scale right-multiplied onto the stack top, src/readScene.h:153-168 of the
// reference tree) and the general 4x4 inverse the renderer needs once per distinct transform
// (the reference recomputes it per primitive test, src/geometries.h:179-180).
//
// fp32 throughout with a fixed operation order (that of the glm routines the reference calls:
// translate(m,v), gtx rotate(angle,axis), gtx scale(v), mat4*mat4, inverse(mat4)), because the
// resulting matrix bits feed every intersection test.  Column-major: m[4*col + row].
#pragma once
#include <cmath>
#include <cstring>

namespace rt {

struct Mat4 {
    float m[16];
    float& operator()(int col, int row) { return m[4 * col + row]; }
    float operator()(int col, int row) const { return m[4 * col + row]; }
};

inline Mat4 mat_identity() {
    Mat4 r;
    std::memset(r.m, 0, sizeof r.m);
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
    return r;
}

// column j of A*B = ((A0*b0 + A1*b1) + A2*b2) + A3*b3
inline Mat4 mat_mul(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++)
            r(j, i) = ((a(0, i) * b(j, 0) + a(1, i) * b(j, 1)) + a(2, i) * b(j, 2)) + a(3, i) * b(j, 3);
    return r;
}

// m with its last column replaced by ((m0*vx + m1*vy) + m2*vz) + m3
inline Mat4 mat_translate(const Mat4& m, float vx, float vy, float vz) {
    Mat4 r = m;
    for (int i = 0; i < 4; i++) r(3, i) = ((m(0, i) * vx + m(1, i) * vy) + m(2, i) * vz) + m(3, i);
    return r;
}

// axis-angle rotation applied to m (Rodrigues form with temp = (1-c)*axis), angle in radians
inline Mat4 mat_rotate(const Mat4& m, float angle, float ax, float ay, float az) {
    const float c = std::cos(angle), s = std::sin(angle);
    const float inv_len = 1.0f / std::sqrt((ax * ax + ay * ay) + az * az);
    const float a[3] = {ax * inv_len, ay * inv_len, az * inv_len};
    const float t[3] = {(1.0f - c) * a[0], (1.0f - c) * a[1], (1.0f - c) * a[2]};
    float R[3][3];
    R[0][0] = c + t[0] * a[0];
    R[0][1] = t[0] * a[1] + s * a[2];
    R[0][2] = t[0] * a[2] - s * a[1];
    R[1][0] = t[1] * a[0] - s * a[2];
    R[1][1] = c + t[1] * a[1];
    R[1][2] = t[1] * a[2] + s * a[0];
    R[2][0] = t[2] * a[0] + s * a[1];
    R[2][1] = t[2] * a[1] - s * a[0];
    R[2][2] = c + t[2] * a[2];
    Mat4 r;
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 4; i++) r(j, i) = (m(0, i) * R[j][0] + m(1, i) * R[j][1]) + m(2, i) * R[j][2];
    for (int i = 0; i < 4; i++) r(3, i) = m(3, i);
    return r;
}

inline Mat4 mat_scale(const Mat4& m, float sx, float sy, float sz) {
    Mat4 r;
    for (int i = 0; i < 4; i++) {
        r(0, i) = m(0, i) * sx;
        r(1, i) = m(1, i) * sy;
        r(2, i) = m(2, i) * sz;
        r(3, i) = m(3, i);
    }
    return r;
}

// General inverse by cofactors: 18 2x2 sub-determinants, four signed cofactor columns, determinant as
// (d0 + d1) + (d2 + d3) over column 0 times the first row of the cofactor matrix, one reciprocal.
inline Mat4 mat_inverse(const Mat4& a) {
    auto M = [&](int c, int r) { return a(c, r); };
    const float s00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
    const float s02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
    const float s03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
    const float s04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
    const float s06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
    const float s07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
    const float s08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
    const float s10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
    const float s11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
    const float s12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
    const float s14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
    const float s15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
    const float s16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
    const float s18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
    const float s19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
    const float s20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
    const float s22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
    const float s23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);

    const float f0[4] = {s00, s00, s02, s03}, f1[4] = {s04, s04, s06, s07}, f2[4] = {s08, s08, s10, s11};
    const float f3[4] = {s12, s12, s14, s15}, f4[4] = {s16, s16, s18, s19}, f5[4] = {s20, s20, s22, s23};
    const float v0[4] = {M(1, 0), M(0, 0), M(0, 0), M(0, 0)};
    const float v1[4] = {M(1, 1), M(0, 1), M(0, 1), M(0, 1)};
    const float v2[4] = {M(1, 2), M(0, 2), M(0, 2), M(0, 2)};
    const float v3[4] = {M(1, 3), M(0, 3), M(0, 3), M(0, 3)};
    const float sa[4] = {+1, -1, +1, -1}, sb[4] = {-1, +1, -1, +1};

    Mat4 inv;
    for (int i = 0; i < 4; i++) {
        inv(0, i) = ((v1[i] * f0[i] - v2[i] * f1[i]) + v3[i] * f2[i]) * sa[i];
        inv(1, i) = ((v0[i] * f0[i] - v2[i] * f3[i]) + v3[i] * f4[i]) * sb[i];
        inv(2, i) = ((v0[i] * f1[i] - v1[i] * f3[i]) + v3[i] * f5[i]) * sa[i];
        inv(3, i) = ((v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i]) * sb[i];
    }
    const float d0 = M(0, 0) * inv(0, 0), d1 = M(0, 1) * inv(1, 0), d2 = M(0, 2) * inv(2, 0), d3 = M(0, 3) * inv(3, 0);
    const float det = (d0 + d1) + (d2 + d3);
    const float one_over = 1.0f / det;
    for (int k = 0; k < 16; k++) inv.m[k] = inv.m[k] * one_over;
    return inv;
}

}  // namespace rt
