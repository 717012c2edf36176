// Level-synchronous pipelines: trace / shade kernels per recursion level over global queues (one lane per ray, and wave-cooperative), resolve kernels.
// Part of the device layer of librtx; included by rtx_device.hip only (one translation unit).
#pragma once

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
