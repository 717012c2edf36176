// Device-side ray/primitive tests and the grid walk (gfx950, wave64).
//
// One lane = one ray.  Arithmetic that decides hits or cell steps is written operation by operation
// in the order of the reference renderer (files cited below are relative to the reference tree) and
// compiled with -ffp-contract=off, so a lane computes the same bits as the reference's scalar CPU code.
#pragma once
#include "rt_math.h"

namespace rt {

// ---- flat scene on the device ---------------------------------------------------------------------
// Primitive record: 3 x float4 (48 B, one aligned gather per lane, or scalar loads when the index is
// wave-uniform):
//   triangle: q0 = (A.x, A.y, A.z, e1.x)  q1 = (e1.y, e1.z, e2.x, e2.y)  q2 = (e2.z, xform, material, kind)
//             with e1 = A - B, e2 = A - C, the first two columns of the reference's 3x3 system
//             (src/geometries.h:306)
//   sphere:   q0 = (S.x, S.y, S.z, r)      q1 = unused                    q2 = (0, xform, material, kind)
struct DevScene {
    const float4* prims;
    const float4* tri_normal;   // object-space normal normalize(cross(B-A, C-A)) per triangle (src/geometries.h:308)
    const float* xinv;          // 16 floats per transform: inverse, column-major
    const float* mats;          // 16 floats per material: ambient, diffuse, specular, emission, shininess, pad
    const float* lights;        // 12 floats per light: kind, pos xyz, colour rgb, attenuation c0 c1 c2, pad
    int n_prims, n_lights;
};

struct DevGrid {
    f3 lo, hi, size;            // Grid::start_pos, end_pos, size (src/grid.h:12-14)
    f3 res;                     // Grid::resolution as floats (src/grid.h:15)
    int rx, ry, rz;
    const int* cstart;          // reference cell lists, CSR over rx*ry*rz cells
    const int* cids;
    int sub;                    // fine cells per coarse cell and axis
    int cpb;                    // fine cells per coarse cell in the block-major numbering (sub^3 padded to 32)
    int frx, fry, frz;          // fine resolution = r * sub
    f3 fcell, inv_fcell;        // fine cell edge and its reciprocal
    f3 fine_off;                // -lo * inv_fcell: fine coordinate of a point p = p * inv_fcell + fine_off
    const int* fstart;          // fine cell lists (tight padded boxes), CSR over cell = block * cpb + local,
                                // block = coarse cell index, local = lx + sub * (ly + sub * lz)
    const int* fids;
    const float4* frecs;        // the primitive records of fids, copied into list order (3 x float4 per reference -- 6, with the inverse transform's rows, when the scene has several transforms --, the
                                // primitive index in place of the material): a cell's candidates are one contiguous read
    const unsigned char* cocc;  // per coarse cell: 1 if any nested fine cell holds a primitive
    const unsigned* cocc_bits;  // the same as a bit mask, 32 coarse cells per word
};

struct WorkCounters {
    unsigned long long tri_tests, sphere_tests, xform_evals, cell_steps, fine_steps, light_evals, nearest_hits, rays;
    // wave-level events (counted once per wave): rounds, walk-loop iterations, 64-pair test batches, state updates
    unsigned long long w_rounds, w_walk_iters, w_test_batches, w_updates;
    // shader-clock cycles a wave spent per section (counting builds only): state update, walk, pair tests, fold
    unsigned long long c_update, c_walk, c_test, c_fold;
};

#define RT_KIND_TRI 0
#define RT_KIND_SPHERE 1
#define RT_SEARCH_REF 0
#define RT_SEARCH_NESTED 1

// (a[k]*b[k+1])*c[k+2] - (c[k]*b[k+1])*a[k+2] summed as (t0 + t1) + t2: rule of Sarrus exactly as
// detTerm3x3 / det3x3 evaluate it (src/geometries.h:267-280)
__device__ __forceinline__ float det3(f3 a, f3 b, f3 c) {
    const float t0 = a.x * b.y * c.z - c.x * b.y * a.z;
    const float t1 = a.y * b.z * c.x - c.y * b.z * a.x;
    const float t2 = a.z * b.x * c.y - c.z * b.x * a.y;
    return t0 + t1 + t2;
}

// True when the IEEE quotient num/den is certainly a nonzero negative number, decided without dividing:
// opposite signs, and magnitudes that keep the quotient above the underflow threshold (2^-100 / 2^40 = 2^-140).
__device__ __forceinline__ bool quotient_negative(float num, float den) {
    return ((as_int(num) ^ as_int(den)) < 0) && (fabsf(num) >= 0x1p-100f) && (fabsf(den) <= 0x1p40f);
}

// Triangle::intersect (src/geometries.h:305-311): Cramer solve of [e1 e2 D](beta,gamma,t)^T = A - O,
// hit iff beta >= 0 && gamma >= 0 && beta + gamma <= 1 && t > 0.  The three IEEE divides are skipped when
// a numerator's sign already decides a failing comparison (the common case); the accepted set is unchanged.
// Split in two so that the wave-cooperative search can run the divisions of several batches together:
// numerators + sign tests, then quotients + comparisons.  hit_triangle is the two in sequence.
#ifndef RT_PACKED_DETS
#define RT_PACKED_DETS 1
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool triangle_numerators(f3 A, f3 e1, f3 e2, f3 O, f3 D, float& nb, float& ng, float& nt, float& detA) {
    const f3 d = A - O;
#if RT_PACKED_DETS
    // The four determinants two at a time, each pair in the two halves of packed fp32 operations.  Every component is
    // the same sequence of separately rounded multiplies, subtractions and additions as det3 -- term k is
    // (a[k] * b[k+1]) * c[k+2] - (c[k] * b[k+1]) * a[k+2], the sum (t0 + t1) + t2 -- so the values are det3's bit for bit.
    //   (detA, nt) = det3(e1, e2, c) for c = D and c = d;   (nb, ng) = (det3(d, e2, D), det3(e1, d, D))
    const float a1[3] = {e1.x, e1.y, e1.z}, b2[3] = {e2.x, e2.y, e2.z}, dd[3] = {d.x, d.y, d.z}, DD[3] = {D.x, D.y, D.z};
    v2f t1[3], t2[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        const v2f cD_d_k = {DD[k], dd[k]}, cD_d_k2 = {DD[k2], dd[k2]};
        const float m = a1[k] * b2[k1];
        const v2f p = (v2f){m, m} * cD_d_k2;
        const v2f q = (cD_d_k * (v2f){b2[k1], b2[k1]}) * (v2f){a1[k2], a1[k2]};
        t1[k] = p - q;
        const v2f a_k = {dd[k], a1[k]}, b_k1 = {b2[k1], dd[k1]}, a_k2 = {dd[k2], a1[k2]};
        const v2f p2 = (a_k * b_k1) * (v2f){DD[k2], DD[k2]};
        const v2f q2 = ((v2f){DD[k], DD[k]} * b_k1) * a_k2;
        t2[k] = p2 - q2;
    }
    const v2f s1 = (t1[0] + t1[1]) + t1[2], s2 = (t2[0] + t2[1]) + t2[2];
    detA = s1.x; nt = s1.y;
    nb = s2.x; ng = s2.y;
#else
    detA = det3(e1, e2, D);
    nb = det3(d, e2, D);
    ng = det3(e1, d, D);
    nt = det3(e1, e2, d);
#endif
    // t > 0 fails for every quotient that is negative, zero (either sign) or NaN
    if (((as_int(nt) ^ as_int(detA)) < 0) || nt == 0.0f) return false;
    if (quotient_negative(nb, detA) || quotient_negative(ng, detA)) return false;
    return true;
}
__device__ __forceinline__ bool triangle_quotients(float nb, float ng, float nt, float detA, float& t_out) {
    const float beta = nb / detA, gamma = ng / detA, t = nt / detA;
    if (beta >= 0 && gamma >= 0 && beta + gamma <= 1 && t > 0) {
        t_out = t;
        return true;
    }
    return false;
}
__device__ __forceinline__ bool hit_triangle(f3 A, f3 e1, f3 e2, f3 O, f3 D, float& t_out) {
    float nb, ng, nt, detA;
    return triangle_numerators(A, e1, e2, O, D, nb, ng, nt, detA) && triangle_quotients(nb, ng, nt, detA, t_out);
}

// Sphere::intersect (src/geometries.h:221-251) followed by the sanitising HitInfo constructor (:72-77)
__device__ __forceinline__ bool hit_sphere(f3 S, float r, f3 O, f3 D, float& t_out) {
    const f3 Q = O - S;
    const float dd = dot3(D, D);
    const float p = 2.0f * dot3(Q, D) / dd;
    const float q = (dot3(Q, Q) - r * r) / dd;
    const float disc = p * p / 4.0f - q;
    float t;
    if (disc == 0) {
        t = -p / 2.0f;
    } else if (disc > 0) {
        const float root = sqrtf(disc);
        const float x1 = -p / 2.0f + root, x2 = -p / 2.0f - root;
        t = sel_min(x1, x2);
        if (x1 <= 0) t = x2;
        if (x2 <= 0) t = x1;
    } else {
        return false;
    }
    if (t > 0) {
        t_out = t;
        return true;
    }
    return false;
}

// Per-lane search state: the nearest accepted hit so far and the object-space ray of the transform
// used last (the reference transforms the ray anew for every primitive, src/geometries.h:179-180;
// the result only depends on the transform, so it is kept while consecutive primitives share one).
struct Search {
    float best_t;
    int best_prim;
    int cur_xf;
    f3 Oo, Do;
};

template <bool COUNT>
__device__ __forceinline__ void test_prim(const DevScene& sc, int pid, f3 O, f3 D, float t_bound, Search& s,
                                          WorkCounters& wc) {
    const float4 q0 = sc.prims[3 * pid], q1 = sc.prims[3 * pid + 1], q2 = sc.prims[3 * pid + 2];
    const int xf = as_int(q2.y);
    if (xf != s.cur_xf) {
        const float* m = sc.xinv + 16 * xf;
        s.Oo = mul_point(m, O);
        s.Do = mul_dir(m, D);
        s.cur_xf = xf;
        if (COUNT) wc.xform_evals++;
    }
    float t;
    bool ok;
    if (as_int(q2.w) == RT_KIND_TRI) {
        if (COUNT) wc.tri_tests++;
        ok = hit_triangle(mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x), s.Oo, s.Do, t);
    } else {
        if (COUNT) wc.sphere_tests++;
        ok = hit_sphere(mk(q0.x, q0.y, q0.z), q0.w, s.Oo, s.Do, t);
    }
    // Container::intersect keeps a candidate iff t < best && t < bound, first in file order wins ties
    // (src/geometries.h:184-187); written with an explicit index tie-break so list order does not matter.
    if (ok && t < t_bound && (t < s.best_t || (t == s.best_t && pid < s.best_prim))) {
        s.best_t = t;
        s.best_prim = pid;
    }
}

// Grid::intersect (src/grid.h:259-261) -> traverseGrid (:149-257) for one ray.  ANY stops at the first
// accepted candidate (enough for shadow rays, which only read validHit, src/main.cpp:59,72).
// Returns true on a hit and fills t / prim.
template <int MODE, bool ANY, bool COUNT>
__device__ bool traverse(const DevScene& sc, const DevGrid& g, f3 O, f3 D, float t_limit, float& t_hit, int& prim_hit,
                         WorkCounters& wc) {
    if (COUNT) wc.rays++;
    // collidesWithBox (:91-138): six IEEE divides, +-inf for zero direction components
    const float txs = (g.lo.x - O.x) / D.x, txe = (g.hi.x - O.x) / D.x;
    const float tys = (g.lo.y - O.y) / D.y, tye = (g.hi.y - O.y) / D.y;
    const float tzs = (g.lo.z - O.z) / D.z, tze = (g.hi.z - O.z) / D.z;
    float tx0 = txs, tx1 = txe; if (D.x < 0) { tx0 = txe; tx1 = txs; }
    float ty0 = tys, ty1 = tye; if (D.y < 0) { ty0 = tye; ty1 = tys; }
    float tz0 = tzs, tz1 = tze; if (D.z < 0) { tz0 = tze; tz1 = tzs; }
    const float t0 = max_of3(tx0, ty0, tz0), t1 = min_of3(tx1, ty1, tz1);
    const float dtx = (tx1 - tx0) / g.res.x, dty = (ty1 - ty0) / g.res.y, dtz = (tz1 - tz0) / g.res.z;
    if (!(t0 < t1 && t1 > 0.0001f)) return false;

    // isInsideGrid (:140-147) and the start cell (:156-164, :53-59)
    const bool inside = O.x > g.lo.x && O.x < g.hi.x && O.y > g.lo.y && O.y < g.hi.y && O.z > g.lo.z && O.z < g.hi.z;
    const f3 pos = inside ? O : O + t0 * D;
    int ix = (int)clampf(floorf((pos.x - g.lo.x) * g.res.x / g.size.x), 0.0f, g.res.x - 1.0f);
    int iy = (int)clampf(floorf((pos.y - g.lo.y) * g.res.y / g.size.y), 0.0f, g.res.y - 1.0f);
    int iz = (int)clampf(floorf((pos.z - g.lo.z) * g.res.z / g.size.z), 0.0f, g.res.z - 1.0f);

    // per-axis step, stop index and first crossing (:166-227)
    int sx, ex, sy, ey, sz, ez;
    float nx, ny, nz;
    if (D.x > 0) { sx = 1; ex = g.rx; nx = tx0 + (float)(ix + 1) * dtx; }
    else if (D.x < 0) { sx = -1; ex = -1; nx = tx0 + (float)(g.rx - ix) * dtx; }
    else { sx = -1; ex = -1; nx = FLT_MAX; }
    if (D.y > 0) { sy = 1; ey = g.ry; ny = ty0 + (float)(iy + 1) * dty; }
    else if (D.y < 0) { sy = -1; ey = -1; ny = ty0 + (float)(g.ry - iy) * dty; }
    else { sy = -1; ey = -1; ny = FLT_MAX; }
    if (D.z > 0) { sz = 1; ez = g.rz; nz = tz0 + (float)(iz + 1) * dtz; }
    else if (D.z < 0) { sz = -1; ez = -1; nz = tz0 + (float)(g.rz - iz) * dtz; }
    else { sz = -1; ez = -1; nz = FLT_MAX; }

    Search s;
    s.cur_xf = -1;
    s.Oo = O;
    s.Do = D;

    // nested walk only: reciprocal direction and the ray parameter at which the current coarse cell is entered
    f3 inv_d = mk(0, 0, 0);
    float t_in = 0.0f;
    if (MODE == RT_SEARCH_NESTED) {
        inv_d = mk(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
        t_in = inside ? 0.0f : t0;
    }

    while (ix != ex && iy != ey && iz != ez) {   // :229-254
        if (COUNT) wc.cell_steps++;
        const float t_exit = min_of4(nx, ny, nz, t_limit);
        const int cell = ix + g.rx * iy + g.rx * g.ry * iz;
        s.best_t = FLT_MAX;
        s.best_prim = -1;

        if (MODE == RT_SEARCH_REF) {
            const int b = g.cstart[cell], e = g.cstart[cell + 1];
            for (int j = b; j < e; j++) {
                test_prim<COUNT>(sc, g.cids[j], O, D, t_exit, s, wc);
                if (ANY && s.best_prim >= 0) break;
            }
        } else if (g.cocc[cell]) {
            // 3D-DDA over the sub x sub x sub fine cells of this coarse cell.  Fine lists hold primitives by
            // tight boxes grown by a margin far above the rounding of this walk, so every primitive that
            // can be hit at t < t_exit inside this coarse cell is met; acceptance is the same t < t_exit.
            const int k = g.sub;
            const int bx = ix * k, by = iy * k, bz = iz * k;
            const float ts = fmaxf(t_in, 0.0f);
            const f3 p = O + ts * D;
            int fx = min(max((int)floorf((p.x - g.lo.x) * g.inv_fcell.x), bx), bx + k - 1);
            int fy = min(max((int)floorf((p.y - g.lo.y) * g.inv_fcell.y), by), by + k - 1);
            int fz = min(max((int)floorf((p.z - g.lo.z) * g.inv_fcell.z), bz), bz + k - 1);
            const float inf = __builtin_inff();
            float fnx = D.x > 0 ? (g.lo.x + (float)(fx + 1) * g.fcell.x - O.x) * inv_d.x
                                : (D.x < 0 ? (g.lo.x + (float)fx * g.fcell.x - O.x) * inv_d.x : inf);
            float fny = D.y > 0 ? (g.lo.y + (float)(fy + 1) * g.fcell.y - O.y) * inv_d.y
                                : (D.y < 0 ? (g.lo.y + (float)fy * g.fcell.y - O.y) * inv_d.y : inf);
            float fnz = D.z > 0 ? (g.lo.z + (float)(fz + 1) * g.fcell.z - O.z) * inv_d.z
                                : (D.z < 0 ? (g.lo.z + (float)fz * g.fcell.z - O.z) * inv_d.z : inf);
            const float fdx = fabsf(g.fcell.x * inv_d.x), fdy = fabsf(g.fcell.y * inv_d.y), fdz = fabsf(g.fcell.z * inv_d.z);
            // every lane leaves this loop: each pass either breaks or moves one fine index toward the block edge
            for (;;) {
                if (COUNT) wc.fine_steps++;
                const int fc = cell * g.cpb + (fx - bx) + k * ((fy - by) + k * (fz - bz));
                const int b = g.fstart[fc], e = g.fstart[fc + 1];
                for (int j = b; j < e; j++) {
                    test_prim<COUNT>(sc, g.fids[j], O, D, t_exit, s, wc);
                    if (ANY && s.best_prim >= 0) break;
                }
                if (ANY && s.best_prim >= 0) break;
                const float tc = fminf(fnx, fminf(fny, fnz));
                if (s.best_t < tc) break;       // nearest hit lies before the next fine cell starts
                if (!(tc < t_exit)) break;      // the remaining fine cells begin beyond the coarse exit
                if (tc == fnx) { fx += sx; fnx += fdx; if (fx < bx || fx >= bx + k) break; }
                else if (tc == fny) { fy += sy; fny += fdy; if (fy < by || fy >= by + k) break; }
                else { fz += sz; fnz += fdz; if (fz < bz || fz >= bz + k) break; }
            }
        }

        if (s.best_prim >= 0) {
            t_hit = s.best_t;
            prim_hit = s.best_prim;
            return true;
        }
        // advance the axis whose crossing equals the exit parameter, x before y before z (:239-253)
        if (t_exit == nx) { ix += sx; nx += dtx; }
        else if (t_exit == ny) { iy += sy; ny += dty; }
        else if (t_exit == nz) { iz += sz; nz += dtz; }
        else return false;   // t_exit == t_limit (:251-253); also taken when nothing compares equal (NaN), where the
                             // reference would spin forever
        if (MODE == RT_SEARCH_NESTED) t_in = t_exit;
    }
    return false;
}

// World-space position and normal of an accepted hit (Container::intersect, src/geometries.h:195-201):
// P = O + t*D, N = normalize(transpose(inverse(M)) * (n, 0)) with n the object-space normal
// (sphere: (O' + t*D') - S, :234,247; triangle: constant per triangle, :308).
// SINGLE_XF: the scene has one transform (its matrix is then read through a wave-uniform address: scalar loads);
// PRIMS: 1 = only triangles, 2 = only spheres in the scene (the other branch is compiled out), 0 = both
template <bool SINGLE_XF = false, int PRIMS = 0>
__device__ __forceinline__ void finish_hit(const DevScene& sc, int pid, f3 O, f3 D, float t, f3& P, f3& N, int& material) {
    const float4 q0 = sc.prims[3 * pid], q2 = sc.prims[3 * pid + 2];
    const int xf = SINGLE_XF ? 0 : as_int(q2.y);
    const float* m = sc.xinv + 16 * xf;
    f3 n;
    if (PRIMS == 1 || (PRIMS == 0 && as_int(q2.w) == RT_KIND_TRI)) {
        const float4 tn = sc.tri_normal[pid];
        n = mk(tn.x, tn.y, tn.z);
    } else {
        const f3 Oo = mul_point(m, O), Do = mul_dir(m, D);
        n = (Oo + t * Do) - mk(q0.x, q0.y, q0.z);
    }
    P = O + t * D;
    N = mul_normal(m, n);
    material = as_int(q2.z);
}

}  // namespace rt
