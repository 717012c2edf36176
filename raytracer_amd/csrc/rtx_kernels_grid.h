// Grid build on the device: primitive boxes, scene bounds, count / prefix-scan / scatter of the cell lists, list ordering, sparse view of the nested grid.
// Part of the device layer of librtx; included by rtx_device.hip only (one translation unit).
#pragma once

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
// One lane per primitive walks the primitive's cells when they are few; a primitive whose box spans more than
// RTX_BIN_WIDE cells (a wall of a room spans the whole grid) is put on the `wide` list instead and k_bin_wide gives
// it a whole workgroup, so that the build does not wait for the one lane that holds the largest primitive.
#define RTX_BIN_WIDE 256
struct BinPrim {
    int c0[3], c1[3];
    bool exact;
    f3 wa, wb, wc, cellsz, half;
};
__device__ __forceinline__ BinPrim bin_prim(int i, const BinParams& bp, const float* boxes, const float* geom, const int* kind, const int* xform,
                                           const float* xfwd) {
    BinPrim q;
    cell_range(bp, boxes + 6 * i, q.c0, q.c1);
    q.exact = false;
    q.wa = q.wb = q.wc = q.cellsz = q.half = mk(0, 0, 0);
    if (geom != nullptr && kind[i] == RTX_PRIM_TRIANGLE && (q.c1[0] > q.c0[0] || q.c1[1] > q.c0[1] || q.c1[2] > q.c0[2])) {
        const float* gp = geom + 9 * i;
        const float* m = xfwd + 16 * xform[i];
        q.wa = mul_point(m, mk(gp[0], gp[1], gp[2]));
        q.wb = mul_point(m, mk(gp[3], gp[4], gp[5]));
        q.wc = mul_point(m, mk(gp[6], gp[7], gp[8]));
        q.cellsz = mk(bp.size.x / bp.res.x, bp.size.y / bp.res.y, bp.size.z / bp.res.z);
        q.half = mk(0.5f * q.cellsz.x + bp.pad, 0.5f * q.cellsz.y + bp.pad, 0.5f * q.cellsz.z + bp.pad);
        q.exact = true;
    }
    return q;
}
template <bool FILL>
__device__ __forceinline__ void bin_cell(int i, const BinParams& bp, const BinPrim& q, int x, int y, int z, int* count_or_cursor, const int* start,
                                         int* ids) {
    if (q.exact) {
        const f3 centre = mk(bp.lo.x + ((float)x + 0.5f) * q.cellsz.x, bp.lo.y + ((float)y + 0.5f) * q.cellsz.y,
                             bp.lo.z + ((float)z + 0.5f) * q.cellsz.z);
        if (!tri_box_overlap(centre, q.half, q.wa, q.wb, q.wc)) return;
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
template <bool FILL>
__global__ void k_bin(int n, BinParams bp, const float* boxes, int* count_or_cursor, const int* start, int* ids,
                      const float* geom, const int* kind, const int* xform, const float* xfwd, int* wide, int* n_wide) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const BinPrim q = bin_prim(i, bp, boxes, geom, kind, xform, xfwd);
    const long long cells = (long long)(q.c1[0] - q.c0[0] + 1) * (q.c1[1] - q.c0[1] + 1) * (q.c1[2] - q.c0[2] + 1);
    if (wide != nullptr && cells > RTX_BIN_WIDE) { wide[atomicAdd(n_wide, 1)] = i; return; }
    for (int z = q.c0[2]; z <= q.c1[2]; z++)
        for (int y = q.c0[1]; y <= q.c1[1]; y++)
            for (int x = q.c0[0]; x <= q.c1[0]; x++) bin_cell<FILL>(i, bp, q, x, y, z, count_or_cursor, start, ids);
}
// the primitives k_bin put aside: a workgroup per primitive, its lanes striding over the primitive's cells
template <bool FILL>
__global__ __launch_bounds__(256) void k_bin_wide(const int* wide, const int* n_wide, BinParams bp, const float* boxes, int* count_or_cursor,
                                                  const int* start, int* ids, const float* geom, const int* kind, const int* xform,
                                                  const float* xfwd) {
    for (int at = (int)blockIdx.x; at < *n_wide; at += (int)gridDim.x) {
        const int i = wide[at];
        const BinPrim q = bin_prim(i, bp, boxes, geom, kind, xform, xfwd);
        const int nx = q.c1[0] - q.c0[0] + 1, ny = q.c1[1] - q.c0[1] + 1, nz = q.c1[2] - q.c0[2] + 1;
        const long long cells = (long long)nx * ny * nz;
        for (long long j = (long long)threadIdx.x; j < cells; j += 256) {
            const int x = q.c0[0] + (int)(j % nx), y = q.c0[1] + (int)((j / nx) % ny), z = q.c0[2] + (int)(j / ((long long)nx * ny));
            bin_cell<FILL>(i, bp, q, x, y, z, count_or_cursor, start, ids);
        }
    }
}

// Exclusive prefix sum over the cell counts in three launches: every workgroup scans its own strip of RTX_SCAN_STRIP
// counts and leaves the strip's total; one workgroup turns the totals into strip offsets; every workgroup adds its
// offset.  start has n + 1 entries; start[n] = total.  n <= 2^27 cells gives at most 2^15 strips, which the one
// workgroup of the middle launch walks in 32 passes.
#define RTX_SCAN_STRIP 4096                      // counts per workgroup: 1024 lanes x 4
__device__ __forceinline__ int block_exclusive_scan_1024(int v, int* buf, int* total) {
    // inclusive scan of one value per lane over 1024 lanes: per wave with lane shuffles, then the 16 wave totals
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int x = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    if (lane == 63) buf[wave] = x;
    __syncthreads();
    if (wave == 0) {
        int w = lane < 16 ? buf[lane] : 0;
        for (int off = 1; off < 16; off <<= 1) {
            const int y = __shfl_up(w, off);
            if (lane >= off) w += y;
        }
        if (lane < 16) buf[16 + lane] = w;           // inclusive totals of waves 0..lane
    }
    __syncthreads();
    const int before = wave > 0 ? buf[16 + wave - 1] : 0;
    *total = buf[31];
    __syncthreads();
    return before + x - v;
}
__global__ __launch_bounds__(1024) void k_scan_strips(int n, const int* count, int* start, int* strip_total) {
    __shared__ int buf[32];
    const int base = (int)blockIdx.x * RTX_SCAN_STRIP + 4 * (int)threadIdx.x;
    int v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = base + k < n ? count[base + k] : 0; sum += v[k]; }
    int total;
    int run = block_exclusive_scan_1024(sum, buf, &total);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (base + k < n) start[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 0) strip_total[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void k_scan_totals(int n_strips, int* strip_total, int n, int* start) {
    __shared__ int buf[32];
    int carry = 0;
    for (int base = 0; base < n_strips; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const int v = i < n_strips ? strip_total[i] : 0;
        int total;
        const int ex = block_exclusive_scan_1024(v, buf, &total);
        if (i < n_strips) strip_total[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) start[n] = carry;
}
__global__ __launch_bounds__(1024) void k_scan_offsets(int n, const int* strip_total, int* start) {
    const int off = strip_total[blockIdx.x];
    const int base = (int)blockIdx.x * RTX_SCAN_STRIP + 4 * (int)threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (base + k < n) start[base + k] += off;
}

// Put every cell's ids in ascending (= file) order.  Short lists: one lane per cell, insertion sort; the cells with
// longer lists are collected in `long_cells` (their number in *n_long) for the workgroup-per-cell kernels below, which
// walk that list with a fixed number of workgroups instead of being launched once per cell of the grid.
__global__ void k_sort_small(int n_cells, const int* start, int* ids, int max_len, int* long_cells, int* n_long) {
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (c >= n_cells) return;
    const int b = start[c], e = start[c + 1];
    if (e - b > max_len) { long_cells[atomicAdd(n_long, 1)] = c; return; }
    if (e - b < 2) return;
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
__global__ __launch_bounds__(256) void k_sort_lds(const int* long_cells, const int* n_long, const int* start, int* ids) {
    __shared__ int buf[RTX_SORT_LDS_MAX];
  for (int at = (int)blockIdx.x; at < *n_long; at += (int)gridDim.x) {
    const int c = long_cells[at];
    const int b = start[c], e = start[c + 1], len = e - b;
    if (len > RTX_SORT_LDS_MAX) continue;                      // uniform over the workgroup
    __syncthreads();                                           // the previous cell's copy-out is complete
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
}
// Longer lists: ids within a cell are distinct, so an element's final position is the number of smaller elements
// (rank sort through a scratch copy, quadratic: only for lists that do not fit the LDS sort).
__global__ __launch_bounds__(256) void k_sort_large(const int* long_cells, const int* n_long, const int* start, int* ids, int* scratch) {
  for (int at = (int)blockIdx.x; at < *n_long; at += (int)gridDim.x) {
    const int c = long_cells[at];
    const int b = start[c], e = start[c + 1], len = e - b;
    if (len <= RTX_SORT_LDS_MAX) continue;
    for (int i = (int)threadIdx.x; i < len; i += 256) scratch[b + i] = ids[b + i];
    __syncthreads();
    for (int i = (int)threadIdx.x; i < len; i += 256) {
        const int v = scratch[b + i];
        int rank = 0;
        for (int j = 0; j < len; j++) rank += scratch[b + j] < v ? 1 : 0;
        ids[b + rank] = v;
    }
  }
}

// Per-reference copies of the primitive records in list order (nested grid): frecs[S j .. S j + 2] = record of fids[j]
// with the primitive index in the slot that holds the material in the table.  S = 3 for a scene with one transform
// (its inverse lives in scalar registers); otherwise S = 6 and rows 0..2 of the primitive's inverse transform follow
// the record -- (m0 m1 m2 m4) (m5 m6 m8 m9) (m10 m12 m13 m14) of the column-major 4x4 -- so that a pair's test does not
// wait for a second load whose address it only learns from the first.
__global__ void k_expand_records(int n_refs, const int* fids, const float4* prims, const float* xinv, int stride, float4* frecs) {
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= n_refs) return;
    const int pid = fids[j];
    const float4 q2 = prims[3 * pid + 2];
    float4* out = frecs + (size_t)stride * j;
    out[0] = prims[3 * pid];
    out[1] = prims[3 * pid + 1];
    out[2] = make_float4(q2.x, q2.y, as_float(pid), q2.w);
    if (stride == 6) {
        const float* m = xinv + 16 * as_int(q2.y);
        out[3] = make_float4(m[0], m[1], m[2], m[4]);
        out[4] = make_float4(m[5], m[6], m[8], m[9]);
        out[5] = make_float4(m[10], m[12], m[13], m[14]);
    }
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
// and, per occupied cell, its fine cell number (block * cells per block + local)
__global__ void k_compact(int n_words, const int* fstart, const unsigned* mask, const int* wpre, uint2* words, int* ostart, int* ocell) {
    const int w = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (w >= n_words) return;
    const unsigned m = mask[w];
    int o = wpre[w];
    words[w] = make_uint2(m, (unsigned)o);
    for (int b = 0; b < 32; b++)
        if (m & (1u << b)) { ocell[o] = 32 * w + b; ostart[o++] = fstart[32 * w + b]; }
    if (w == n_words - 1) ostart[wpre[n_words]] = fstart[32 * n_words];
}
// Classes of equal lists.  Neighbouring fine cells often hold the same list (a sphere or a large triangle spans many of
// them): a ray that has just tested a list need not test it again in the next cell that holds the same one.  Every
// occupied fine cell gets the number of the first occupied cell (compact numbering) whose list is identical -- same
// primitives, same order -- so that two cells' lists are equal if (not only if) their classes are: cells are
// keyed by a 64-bit hash of their list in an open-addressing table, the smallest cell number per key is the key's
// representative, and a cell whose list differs from its representative's (two different lists with one hash) keeps
// its own number.  info[o] = (list start, list length, class, 0).
__device__ __forceinline__ unsigned long long list_hash(const int* ids, int n) {
    unsigned long long h = 0x9e3779b97f4a7c15ull ^ (unsigned long long)(unsigned)n;
    for (int i = 0; i < n; i++) {
        h ^= (unsigned long long)(unsigned)ids[i] + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
        h *= 0xff51afd7ed558ccdull;
        h ^= h >> 33;
    }
    return h | 1ull;                                      // (0 marks a free slot of the table)
}
__global__ void k_list_keys(int n_occ, const int* ostart, const int* fids, unsigned long long* key, unsigned long long* table_key,
                            int* table_rep, int table_mask) {
    const int o = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (o >= n_occ) return;
    const int b = ostart[o], e = ostart[o + 1];
    const unsigned long long h = list_hash(fids + b, e - b);
    key[o] = h;
    for (int slot = (int)(h >> 17) & table_mask;; slot = (slot + 1) & table_mask) {       // (the table has twice as many slots as cells)
        const unsigned long long seen = atomicCAS(table_key + slot, 0ull, h);
        if (seen == 0ull || seen == h) { atomicMin(table_rep + slot, o); return; }
    }
}
__global__ void k_list_classes(int n_occ, const int* ostart, const int* fids, const unsigned long long* key,
                               const unsigned long long* table_key, const int* table_rep, int table_mask, int4* info, int* n_shared) {
    const int o = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (o >= n_occ) return;
    const int b = ostart[o], n = ostart[o + 1] - b;
    const unsigned long long h = key[o];
    int slot = (int)(h >> 17) & table_mask;
    while (table_key[slot] != h) slot = (slot + 1) & table_mask;
    int cls = table_rep[slot];
    const int rb = ostart[cls];
    bool same = ostart[cls + 1] - rb == n;
    for (int i = 0; same && i < n; i++) same = fids[rb + i] == fids[b + i];
    if (!same) cls = o;
    if (cls != o) atomicAdd(n_shared, 1);                 // (cells whose list another cell holds as well)
    info[o] = make_int4(b, n, cls, 0);
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
