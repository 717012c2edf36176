// Wave-cooperative grid traversal for gfx950 (wave64).
//
// A wave owns 64 rays.  Each round has two phases:
//   walk   every lane advances its own ray through the grid (the reference's coarse 3D-DDA,
//          src/grid.h:149-257 of the reference tree, and for the nested search a fine 3D-DDA over an
//          occupancy bitmask) until it stands on a cell that holds primitives;
//   test   the (ray, primitive) candidate pairs of ALL lanes are spread over the 64 lanes: lane p takes
//          pair p, finds its owner ray from the per-lane prefix sums staged in LDS, reads the owner's
//          ray from LDS, gathers the primitive and runs the test; hits are folded into the owner's slot
//          with a 64-bit LDS atomic min over (t bits, primitive index) -- the nearest hit of the list,
//          the first in file order on ties, whatever its distance; the owner then accepts it if it lies
//          before the exit of its coarse cell, which is the reference's "nearest with t < t_exit, first
//          in file order on ties" (src/geometries.h:184-187).  The slot keeps the list's result, so a
//          ray that meets the same list again in its next cell (a sphere or a large triangle spans many
//          cells) applies that cell's bound to it instead of testing again (Walker::walk).
// A ray's dependent-load chain is therefore one round per occupied cell it visits, not one per candidate,
// and a lane whose ray has finished or has few candidates still runs tests for its neighbours.
//
// The arithmetic of the coarse walk and of the primitive tests is the per-lane code of rtx_traverse.h,
// operation for operation; only who executes a test changes, so results are bit-identical to it.
#pragma once
#include "rtx_traverse.h"


namespace rt {

// Sparse view of the fine grid.  Fine cells are numbered block-major: cell = block * cells_per_block + local,
// block = coarse cell index, local = lx + sub * (ly + sub * lz), cells_per_block = 32 * words_per_block
// (padded).  Per 32 cells one uint2: x = occupancy mask, y = number of occupied cells before this word (over
// the whole grid).  The lists of the occupied cells are CSR over that compact numbering (ids in fids, records in frecs).
// Where many cells share their list (skip_repeats, set by the host) the walk looks up, per occupied cell it reaches, the
// class of equal lists the cell belongs to (cells of one class hold identical lists, see k_list_classes) and skips the
// list it tested last -- one more load inside the walk, so not where lists are all but unique.  `lists` is then an int4
// per occupied cell (start, length, class, 0), otherwise the plain CSR starts (int per cell, one more at the end: a
// quarter of the cache footprint).
struct DevSparse {
    const uint2* words;
    const int* lists;
    int words_per_block;
    int skip_repeats;
};

struct __attribute__((aligned(16))) WaveLds {
    float4 ray_a[64];              // ray origin xyz (in the space the tests run in), w = class (of equal lists) of the list the
                                   // ray tested last, -1 = none (kept here, not in a register: the walk only needs it where
                                   // it skips repeated lists)
    float4 ray_b[64];              // ray direction xyz, w = bits of (list start - exclusive prefix)
    unsigned long long best[64];   // nearest hit over the list the lane's ray tested last, whatever its distance:
                                   // (t bits << 32) | primitive, all ones = none
    int owner_at[64];              // per pair slot of the current 64-pair batch: lane whose list starts there, else 0
    // triangle candidates that passed the sign tests and await their divisions (ring of RT_CAND_RING entries)
    float4 cand_n[64];             // numerators of beta, gamma, t and the common denominator
    int2 cand_who[64];             // owner lane, primitive index
};
#define RT_CAND_RING 64

#define RT_BEST_NONE 0xffffffffffffffffull

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Orders LDS traffic between the lanes of one wave.  The lanes run in lockstep, but to the compiler every lane
// is an independent thread: without a fence it may forward a lane's own store to its later load of the same
// address even though another lane wrote there in between.  Wavefront scope costs no cross-wave traffic.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Inclusive prefix sum over the 64 lanes with data-parallel-primitive operands (no LDS round trips): four shifted
// adds inside each row of 16 lanes, then the totals of rows 0 and 2 are broadcast into rows 1 and 3 (row_bcast:15,
// row mask 0b1010) and the total of rows 0-1 into rows 2-3 (row_bcast:31, row mask 0b1100).  A lane without a source
// lane, or outside the row mask, keeps its value.  Every lane must be active.
// Written as assembly: from the intrinsic form the compiler makes a zero, a moved copy and an add of every step (18
// vector instructions), here a step is the one add that reads its first operand through the shift.  A vector
// register written by a vector instruction may be read through a shift two wait states later at the earliest;
// the compiler does not look into the block, so the block itself keeps that distance, also to whatever stands
// before and behind it.
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}

// Coarse-cell occupancy bits of the nested grid: staged in LDS by the whole workgroup when the grid is small enough
// (the reference's 15^3 grid is 106 words), else read from global memory.  Returns the pointer the walkers use.
#define RT_OCC_LDS_WORDS 128
struct Occupancy {
    // The staged copy is addressed as LDS (address space 3) explicitly: a generic pointer selected between the two
    // copies would make every test a flat load, which goes through the vector-memory path even when it lands in LDS.
    unsigned lds_base;        // byte offset of the staged copy in LDS, used when global is null
    const unsigned* global;   // set only when the grid is too large for the LDS copy
    __device__ __forceinline__ bool test(int cell) const {
        unsigned word;
        if (global) word = global[cell >> 5];
        else word = *(const __attribute__((address_space(3))) unsigned*)(size_t)(lds_base + 4u * (unsigned)(cell >> 5));
        return (word >> (cell & 31)) & 1u;
    }
};
__device__ __forceinline__ Occupancy stage_occupancy(const DevGrid& g, unsigned* lds_words) {
    Occupancy o;
    o.lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds_words;
    o.global = nullptr;
    const int n_words = (g.rx * g.ry * g.rz + 31) >> 5;
    if (g.cocc_bits == nullptr || n_words > RT_OCC_LDS_WORDS) { o.global = g.cocc_bits; return o; }
    for (int i = (int)threadIdx.x; i < n_words; i += (int)blockDim.x) lds_words[i] = g.cocc_bits[i];
    __syncthreads();
    return o;
}

// Per-lane search state that lives beside the Walker: the ray in the space the tests run in and the best hit.
struct LaneSearch {
    bool found;
    float best_t;
    int best_prim;
};

// What the last tested list says under the acceptance bound of the coarse cell the ray is in (src/geometries.h:184-187:
// nearest with t < bound, the first in file order on ties -- the key orders by t, then by primitive index): folded into
// the ray's accepted hit.  Called by the ray's own lane, after the tests of a round or when the walk skips a cell that
// holds the list tested last.
__device__ __forceinline__ void accept_under(WaveLds& l, LaneSearch& r, float bound) {
    const unsigned long long b = l.best[lane_id()];
    const float t = as_float((int)(b >> 32));
    const unsigned long long have = r.found ? (((unsigned long long)(unsigned)as_int(r.best_t) << 32) | (unsigned)r.best_prim) : RT_BEST_NONE;
    if (b != RT_BEST_NONE && t < bound && b < have) {
        r.found = true;
        r.best_t = t;
        r.best_prim = (int)(b & 0xffffffffull);
    }
}

// Per-lane walk state.  MODE = RT_SEARCH_REF: every coarse cell's reference list is a candidate list.
// MODE = RT_SEARCH_NESTED: fine 3D-DDA inside occupied coarse cells, candidates = lists of occupied fine cells.
// The advance steps are written with selects instead of branches: the three axes of a divergent wave would
// otherwise run one after the other.
template <int MODE>
struct Walker {
    f3 O, D;
    float t_limit;
    int ix, iy, iz;
    float nx, ny, nz, dtx, dty, dtz;
    float t_exit;
    // nested
    f3 inv_d;
    float t_in;
    int lx, ly, lz;      // fine cell inside the current coarse cell, each in [0, sub)
    float fnx, fny, fnz; // ray parameter of the next fine crossing per axis
    int word_base;       // index of the current coarse cell's first occupancy word
    int cur_w;
    uint2 cur;
    int state;           // 0: about to enter coarse cell (ix,iy,iz); 1: standing on fine cell (lx,ly,lz), not looked
                         // at yet; 3: done with that fine cell, leaves it at the next step; 4: done with the coarse
                         // cell, leaves it at the next step; 2: finished
    bool hit;

    // collidesWithBox + start cell + first crossings (src/grid.h:91-138,156-227); same arithmetic as traverse()
    __device__ __forceinline__ void init(const DevGrid& g, f3 o, f3 d, float tl, bool active) {
        O = o; D = d; t_limit = tl;
        hit = false;
        state = 2;
        cur_w = -1;
        cur = make_uint2(0u, 0u);
        ix = iy = iz = 0; nx = ny = nz = 0; dtx = dty = dtz = 0; t_exit = 0; t_in = 0;
        lx = ly = lz = 0; fnx = fny = fnz = 0; word_base = 0;
        inv_d = mk(0, 0, 0);
        if (!active) return;
        const float txs = (g.lo.x - O.x) / D.x, txe = (g.hi.x - O.x) / D.x;
        const float tys = (g.lo.y - O.y) / D.y, tye = (g.hi.y - O.y) / D.y;
        const float tzs = (g.lo.z - O.z) / D.z, tze = (g.hi.z - O.z) / D.z;
        float tx0 = txs, tx1 = txe; if (D.x < 0) { tx0 = txe; tx1 = txs; }
        float ty0 = tys, ty1 = tye; if (D.y < 0) { ty0 = tye; ty1 = tys; }
        float tz0 = tzs, tz1 = tze; if (D.z < 0) { tz0 = tze; tz1 = tzs; }
        const float t0 = max_of3(tx0, ty0, tz0), t1 = min_of3(tx1, ty1, tz1);
        dtx = (tx1 - tx0) / g.res.x; dty = (ty1 - ty0) / g.res.y; dtz = (tz1 - tz0) / g.res.z;
        if (!(t0 < t1 && t1 > 0.0001f)) return;
        const bool inside = O.x > g.lo.x && O.x < g.hi.x && O.y > g.lo.y && O.y < g.hi.y && O.z > g.lo.z && O.z < g.hi.z;
        const f3 pos = inside ? O : O + t0 * D;
        ix = (int)clampf(floorf((pos.x - g.lo.x) * g.res.x / g.size.x), 0.0f, g.res.x - 1.0f);
        iy = (int)clampf(floorf((pos.y - g.lo.y) * g.res.y / g.size.y), 0.0f, g.res.y - 1.0f);
        iz = (int)clampf(floorf((pos.z - g.lo.z) * g.res.z / g.size.z), 0.0f, g.res.z - 1.0f);
        nx = D.x > 0 ? tx0 + (float)(ix + 1) * dtx : (D.x < 0 ? tx0 + (float)(g.rx - ix) * dtx : FLT_MAX);
        ny = D.y > 0 ? ty0 + (float)(iy + 1) * dty : (D.y < 0 ? ty0 + (float)(g.ry - iy) * dty : FLT_MAX);
        nz = D.z > 0 ? tz0 + (float)(iz + 1) * dtz : (D.z < 0 ? tz0 + (float)(g.rz - iz) * dtz : FLT_MAX);
        if (MODE == RT_SEARCH_NESTED) {
            // the fine walk is acceleration only (its rounding is covered by the margin of the fine lists)
            inv_d = mk(__frcp_rn(D.x), __frcp_rn(D.y), __frcp_rn(D.z));
            t_in = inside ? 0.0f : t0;
        }
        state = 0;
    }

    // leave the current coarse cell without a hit: src/grid.h:239-253, then the loop condition :229
    __device__ __forceinline__ void coarse_advance(const DevGrid& g) {
        const bool ax = t_exit == nx;
        const bool ay = !ax && t_exit == ny;
        const bool az = !ax && !ay && t_exit == nz;
        // no crossing equals the exit parameter: the ray ends at t_limit (src/grid.h:251-253).  The same exit is
        // taken when nothing compares equal at all (a NaN parameter), where the reference would spin forever.
        const bool stop = !ax && !ay && !az;
        ix += ax ? (D.x > 0 ? 1 : -1) : 0;
        iy += ay ? (D.y > 0 ? 1 : -1) : 0;
        iz += az ? (D.z > 0 ? 1 : -1) : 0;
        nx += ax ? dtx : 0.0f;   // adding +0 leaves every value, including infinities and NaN payloads, as it is
        ny += ay ? dty : 0.0f;
        nz += az ? dtz : 0.0f;
        t_in = t_exit;
        const bool in_grid = ix != (D.x > 0 ? g.rx : -1) && iy != (D.y > 0 ? g.ry : -1) && iz != (D.z > 0 ? g.rz : -1);
        state = (stop || !in_grid) ? 2 : 0;
    }

    // Leave the current fine cell (state 3).  With an accepted hit in hand (found) the ray finishes as soon as the
    // hit lies before the next fine cell, or when the walk leaves the block; without one the walk goes on, into
    // the next coarse cell when it leaves the block.
    __device__ __forceinline__ void fine_advance(const DevGrid& g, bool found, float best_t) {
        const int k = g.sub;
        const float tc = fminf(fnx, fminf(fny, fnz));
        const bool done = found && best_t < tc;
        const bool beyond = !(tc < t_exit);               // the remaining fine cells begin beyond the coarse exit
        const bool ax = tc == fnx;
        const bool ay = !ax && tc == fny;
        const bool az = !ax && !ay;
        lx += ax ? (D.x > 0 ? 1 : -1) : 0;
        ly += ay ? (D.y > 0 ? 1 : -1) : 0;
        lz += az ? (D.z > 0 ? 1 : -1) : 0;
        fnx = ax ? __builtin_fmaf(g.fcell.x, fabsf(inv_d.x), fnx) : fnx;
        fny = ay ? __builtin_fmaf(g.fcell.y, fabsf(inv_d.y), fny) : fny;
        fnz = az ? __builtin_fmaf(g.fcell.z, fabsf(inv_d.z), fnz) : fnz;
        const bool outside = (unsigned)lx >= (unsigned)k || (unsigned)ly >= (unsigned)k || (unsigned)lz >= (unsigned)k;
        state = 1;
        if (done) { state = 2; hit = true; }
        else if (beyond || outside) {
            if (found) { state = 2; hit = true; }
            else state = 4;
        }
    }

    // Walks until the lane stands on a candidate list, finishes, or has used `budget` steps.  Returns true on a
    // list: start/count are the list's range.  An occupied fine cell that holds the very list the ray tested last is
    // not a candidate: the same ray against the same primitives gives the same distances, so the result of those tests
    // -- kept without a bound in the lane's slot -- is folded in under this cell's bound (accept_under) and the walk goes
    // on.  Every step either finishes the ray or moves a cell index monotonically toward the grid boundary, so the
    // walk terminates.
    // SKIP: the skipping is compiled in.  (Its mere presence in the loop costs the kernels that never use it 4 - 5 % -- a
    // 4K frame of the dragon, 1.2 % repeats: 1.61 -> 1.68 ms with three frames in flight --, so only the kernels for
    // scenes with several transforms, which are scenes of several objects, carry it.)
    template <bool COUNT, bool SKIP>
    __device__ __forceinline__ bool walk(const DevGrid& g, const DevSparse& sp, const Occupancy& occ, int budget, int& start,
                                         int& count, WaveLds& l, LaneSearch& r, bool any_hit, WorkCounters& wc) {
        while (state != 2 && budget-- > 0) {
            if (COUNT) wc.w_walk_iters++;
            // one step = leave the fine cell that is done with, enter a coarse cell if the walk stands before one,
            // look at the fine cell the walk stands on: the only copy of each piece of the walk in the kernel
            if (MODE == RT_SEARCH_NESTED && state == 3) fine_advance(g, r.found, r.best_t);
            if (state == 4) coarse_advance(g);
            if (state == 0) {
                if (COUNT) wc.cell_steps++;
                t_exit = min_of4(nx, ny, nz, t_limit);
                // 24-bit multiplies (full rate; the 32-bit ones take four times as long): indices stay below 2^10 per
                // axis and below 2^22 per coarse cell number, see the limits in rtx_build_grid
                const int cell = ix + __mul24(g.rx, iy) + __mul24(g.rx * g.ry, iz);
                if (MODE == RT_SEARCH_REF) {
                    const int b = g.cstart[cell], e = g.cstart[cell + 1];
                    if (e > b) { start = b; count = e - b; return true; }
                    state = 4;
                } else if (occ.test(cell)) {
                    const int k = g.sub;
                    // (fused multiply-adds here: the fine walk only chooses which lists are looked at, and the margin the
                    // lists were binned with is four orders of magnitude above its rounding)
                    const float te = fmaxf(t_in, 0.0f);
                    const f3 p = mk(__builtin_fmaf(te, D.x, O.x), __builtin_fmaf(te, D.y, O.y), __builtin_fmaf(te, D.z, O.z));
                    const int bx = __mul24(ix, k), by = __mul24(iy, k), bz = __mul24(iz, k);
                    const int fx = min(max((int)floorf(__builtin_fmaf(p.x, g.inv_fcell.x, g.fine_off.x)) - bx, 0), k - 1);
                    const int fy = min(max((int)floorf(__builtin_fmaf(p.y, g.inv_fcell.y, g.fine_off.y)) - by, 0), k - 1);
                    const int fz = min(max((int)floorf(__builtin_fmaf(p.z, g.inv_fcell.z, g.fine_off.z)) - bz, 0), k - 1);
                    lx = fx; ly = fy; lz = fz;
                    const float inf = __builtin_inff();
                    const float px = __builtin_fmaf((float)(bx + fx + (D.x > 0 ? 1 : 0)), g.fcell.x, g.lo.x);
                    const float py = __builtin_fmaf((float)(by + fy + (D.y > 0 ? 1 : 0)), g.fcell.y, g.lo.y);
                    const float pz = __builtin_fmaf((float)(bz + fz + (D.z > 0 ? 1 : 0)), g.fcell.z, g.lo.z);
                    fnx = D.x != 0 ? (px - O.x) * inv_d.x : inf;
                    fny = D.y != 0 ? (py - O.y) * inv_d.y : inf;
                    fnz = D.z != 0 ? (pz - O.z) * inv_d.z : inf;
                    word_base = __mul24(cell, sp.words_per_block);
                    state = 1;
                } else {
                    state = 4;
                }
            }
            if (MODE == RT_SEARCH_NESTED && state == 1) {
                if (COUNT) wc.fine_steps++;
                const int k = g.sub;
                const int local = lx + __mul24(k, ly + __mul24(k, lz));
                const int w = word_base + (local >> 5);
                if (w != cur_w) { cur = sp.words[w]; cur_w = w; }
                const unsigned bit = 1u << (local & 31);
                if (cur.x & bit) {
                    const int cell = (int)cur.y + __popc(cur.x & (bit - 1u));
                    if (!SKIP || !sp.skip_repeats) { start = cell; count = -1; return true; }      // (the caller reads the list's range)
                    const int4 info = ((const int4*)sp.lists)[cell];
                    float& last_class = l.ray_a[lane_id()].w;
                    if (info.z != as_int(last_class)) {
                        start = info.x;
                        count = info.y;
                        last_class = as_float(info.z);
                        return true;
                    }
                    accept_under(l, r, t_exit);
                    if (r.found && any_hit) { state = 2; hit = true; continue; }
                }
                state = 3;
            }
        }
        return false;
    }

    // after the candidates of the current cell were tested
    __device__ __forceinline__ void after_tests(const DevGrid& g, bool found, float best_t, bool any_hit) {
        if (MODE == RT_SEARCH_REF) {
            if (found) { state = 2; hit = true; }
            else state = 4;
        } else {
            if (found && any_hit) { state = 2; hit = true; }
            else state = 3;                 // the next step leaves the cell, knowing found / best_t
        }
    }
};

// One test of pair (owner ray, primitive record q0..q2).  SINGLE_XF: the scene has exactly one transform and the LDS
// holds the ray already in object space; otherwise the LDS holds the world-space ray and the pair is transformed
// by its primitive's inverse here (src/geometries.h:179-180).  pid < 0: the record is one of the nested grid's
// per-reference copies, which carry the primitive index where the primitive table carries the material.
// Returns true for a triangle whose divisions are still to be done (numerators in cand, index in cand_pid).
template <bool SINGLE_XF, bool COUNT, int PRIMS>
__device__ __forceinline__ bool test_pair(const DevScene& sc, WaveLds& l, int owner, f3 Oo, f3 Do, float4 q0, float4 q1,
                                          float4 q2, float4 x0, float4 x1, float4 x2, int pid, float4& cand, int& cand_pid, WorkCounters& wc) {
    const bool inverse_in_record = pid < 0;              // (the nested grid's copies carry it, see k_expand_records)
    if (pid < 0) pid = as_int(q2.z);
    if (!SINGLE_XF) {
        const f3 Ow = Oo, Dw = Do;
        if (inverse_in_record) {
            const float m[16] = {x0.x, x0.y, x0.z, 0.0f, x0.w, x1.x, x1.y, 0.0f, x1.z, x1.w, x2.x, 0.0f, x2.y, x2.z, x2.w, 1.0f};
            Oo = mul_point(m, Ow);
            Do = mul_dir(m, Dw);
        } else {
            const float* m = sc.xinv + 16 * as_int(q2.y);
            Oo = mul_point(m, Ow);
            Do = mul_dir(m, Dw);
        }
        if (COUNT) wc.xform_evals++;
    }
    // PRIMS: 1 = the scene holds only triangles, 2 = only spheres (the other test is compiled out), 0 = both
    if (PRIMS == 1 || (PRIMS == 0 && as_int(q2.w) == RT_KIND_TRI)) {
        if (COUNT) wc.tri_tests++;
        // a triangle that survives the sign tests is handed back: its divisions run later, many at a time
        cand_pid = pid;
        return triangle_numerators(mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x), Oo, Do, cand.x, cand.y, cand.z, cand.w);
    }
    if (COUNT) wc.sphere_tests++;
    float t;
    if (hit_sphere(mk(q0.x, q0.y, q0.z), q0.w, Oo, Do, t)) {      // (kept whatever its distance: the ray's own lane applies the bound)
        // t > 0 here, so its bit pattern orders like its value; the low word breaks ties toward the lower index
        const unsigned long long key = ((unsigned long long)(unsigned)as_int(t) << 32) | (unsigned)pid;
        atomicMin(&l.best[owner], key);
    }
    return false;
}

// Divisions and acceptance of `n` queued triangle candidates starting at ring position `head` (n <= 64)
__device__ __forceinline__ void finish_candidates(WaveLds& l, int head, int n) {
    const int lane = lane_id();
    if (lane < n) {
        const int at = (head + lane) & (RT_CAND_RING - 1);
        const float4 c = l.cand_n[at];
        const int2 who = l.cand_who[at];
        float t;
        if (triangle_quotients(c.x, c.y, c.z, c.w, t)) {
            const unsigned long long key = ((unsigned long long)(unsigned)as_int(t) << 32) | (unsigned)who.y;
            atomicMin(&l.best[who.x], key);
        }
    }
}

// Rows 0..2 of the scene's only inverse transform, read once per kernel and held in scalar registers
// (SINGLE_XF kernels): the object-space ray of a lane is then 33 VALU operations on registers whenever needed.
struct XfRegs {
    float m[12];   // m[4 * col + row], row < 3
};
__device__ __forceinline__ XfRegs load_xf_regs(const float* xinv, bool use) {
    XfRegs x;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int r = 0; r < 3; r++)
            x.m[3 * c + r] = use ? as_float(__builtin_amdgcn_readfirstlane(as_int(xinv[4 * c + r]))) : 0.0f;
    return x;
}
// same operation order as mul_point / mul_dir
__device__ __forceinline__ f3 xf_point(const XfRegs& x, f3 v) {
    return mk((x.m[0] * v.x + x.m[3] * v.y) + (x.m[6] * v.z + x.m[9] * 1.0f),
              (x.m[1] * v.x + x.m[4] * v.y) + (x.m[7] * v.z + x.m[10] * 1.0f),
              (x.m[2] * v.x + x.m[5] * v.y) + (x.m[8] * v.z + x.m[11] * 1.0f));
}
__device__ __forceinline__ f3 xf_dir(const XfRegs& x, f3 v) {
    return mk((x.m[0] * v.x + x.m[3] * v.y) + (x.m[6] * v.z + x.m[9] * 0.0f),
              (x.m[1] * v.x + x.m[4] * v.y) + (x.m[7] * v.z + x.m[10] * 0.0f),
              (x.m[2] * v.x + x.m[5] * v.y) + (x.m[8] * v.z + x.m[11] * 0.0f));
}

// Inclusive running maximum over the 64 lanes (the prefix sum's pattern, same form; any values: a lane without a
// source lane keeps its own)
__device__ __forceinline__ int wave_inclusive_max(int v) {
    asm volatile(
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}

// Starts a new ray on the calling lanes (others keep theirs).  May be called under divergence.  The ray in the
// space the tests run in goes to the lane's LDS slot here, once: with a single transform in the scene it is taken to
// object space from the matrix held in scalar registers (src/geometries.h:179-180, same arithmetic for every
// primitive); the rounds only add the acceptance bound and the list position (the w components).
template <int MODE, bool SINGLE_XF, bool COUNT>
__device__ __forceinline__ void ray_begin(const DevScene& sc, const DevGrid& g, const XfRegs& xf, WaveLds& l, Walker<MODE>& w, LaneSearch& r,
                                          f3 O, f3 D, float t_limit, WorkCounters& wc) {
    w.init(g, O, D, t_limit, true);
    if (COUNT) { wc.rays++; if (SINGLE_XF) wc.xform_evals++; }
    r.found = false;
    r.best_t = FLT_MAX;
    r.best_prim = -1;
    const int lane = lane_id();
    l.best[lane] = RT_BEST_NONE;
    f3 Or = O, Dr = D;
    if (SINGLE_XF) {
        Or = xf_point(xf, O);
        Dr = xf_dir(xf, D);
    }
    l.ray_a[lane] = make_float4(Or.x, Or.y, Or.z, as_float(-1));
    l.ray_b[lane] = make_float4(Dr.x, Dr.y, Dr.z, 0.0f);
}

// One round for the whole wave: every lane with an unfinished ray walks to its next candidate list, the
// candidate pairs of all lanes are tested by all lanes, and each lane folds its result into its walk.
// any_hit is per lane.  Must be called by all 64 lanes in convergent control flow.
//
// Pair p of the round belongs to the lane with the largest exclusive prefix <= p.  Per batch of 64 pairs each
// lane whose list starts inside the batch writes its lane number at the slot of its first pair; a running
// maximum over the slots (owners ascend with the slots), seeded with the owner carried over from the previous
// batch, gives every slot its owner.
template <int MODE, bool SINGLE_XF, bool COUNT, int PRIMS = 0, bool AHEAD = false>
__device__ __forceinline__ void wave_round(const DevScene& sc, const DevGrid& g, const DevSparse& sp, const Occupancy& occ, const XfRegs& xf,
                                           WaveLds& l, Walker<MODE>& w, LaneSearch& r, bool any_hit, int budget, WorkCounters& wc) {
    const int lane = lane_id();
    int start = 0, count = 0;
    if (COUNT) wc.w_rounds++;
    const unsigned long long c0 = COUNT ? __builtin_readcyclecounter() : 0ull;
    const bool pending = w.template walk<COUNT, !SINGLE_XF>(g, sp, occ, budget, start, count, l, r, any_hit, wc);
    if (!pending) count = 0;
    else {
        if (MODE == RT_SEARCH_NESTED && count < 0) {
            const int b = sp.lists[start], e = sp.lists[start + 1];
            start = b; count = e - b;
        }
        l.best[lane] = RT_BEST_NONE;                 // (the slot takes the result of this list alone)
    }
    const int incl = wave_inclusive_scan(count);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    const unsigned long long c1 = COUNT ? __builtin_readcyclecounter() : 0ull;
    if (total > 0) {
        const int excl = incl - count;
        // the ray itself has been in the lane's slot since ray_begin
        l.ray_b[lane].w = as_float(start - excl);
        int carry = 0;                               // owner of the last pair of the previous batch (wave-uniform)
        int q_head = 0, q_tail = 0;                  // ring positions of the queued triangle candidates (wave-uniform)
        // Owner lookup and record reads of the batch that starts at pair `base`.  With AHEAD the loop below issues
        // them one batch ahead, so that the reads of the next batch travel while the current one is computed: worth
        // 3 to 8 % where a frame is bound by the latency of its rays (small images, a rank's share), a loss of 5 %
        // where it is bound by throughput (the registers that hold the second batch).
        const auto fetch = [&](int base, int& owner, int& pid, float4& q0, float4& q1, float4& q2, float4& x0, float4& x1, float4& x2) {
            l.owner_at[lane] = 0;
            wave_lds_sync();
            if (count > 0 && excl >= base && excl < base + 64) l.owner_at[excl - base] = lane;
            wave_lds_sync();                         // also publishes ray_a / ray_b before the first batch reads them
            owner = max(wave_inclusive_max(l.owner_at[lane]), carry);
            carry = __builtin_amdgcn_readlane(owner, 63);
            const int p = base + lane;
            pid = -1;
            if (p < total) {
                int rec = as_int(l.ray_b[owner].w) + p;       // position in the concatenated cell lists
                const float4* recs = g.frecs;                // the nested grid's copies of the records, in list order
                if (MODE == RT_SEARCH_REF) { pid = rec = g.cids[rec]; recs = sc.prims; }
                constexpr int stride = (MODE == RT_SEARCH_REF || SINGLE_XF) ? 3 : 6;
                q0 = recs[stride * rec]; q1 = recs[stride * rec + 1]; q2 = recs[stride * rec + 2];
                if (stride == 6) { x0 = recs[stride * rec + 3]; x1 = recs[stride * rec + 4]; x2 = recs[stride * rec + 5]; }
            }
        };
        int owner = 0, pid = -1;
        float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0, x0 = q0, x1 = q0, x2 = q0;
        if (AHEAD) fetch(0, owner, pid, q0, q1, q2, x0, x1, x2);
        for (int base = 0; base < total; base += 64) {
            if (COUNT) wc.w_test_batches++;
            int owner_n = 0, pid_n = -1;
            float4 n0 = make_float4(0, 0, 0, 0), n1 = n0, n2 = n0, y0 = n0, y1 = n0, y2 = n0;
            if (!AHEAD) fetch(base, owner, pid, q0, q1, q2, x0, x1, x2);
            else if (base + 64 < total) fetch(base + 64, owner_n, pid_n, n0, n1, n2, y0, y1, y2);
            bool queued = false;
            float4 cand = make_float4(0, 0, 0, 0);
            int cand_pid = -1;
            if (base + lane < total) {
                const float4 ra = l.ray_a[owner], rb = l.ray_b[owner];
                queued = test_pair<SINGLE_XF, COUNT, PRIMS>(sc, l, owner, mk(ra.x, ra.y, ra.z), mk(rb.x, rb.y, rb.z), q0, q1, q2, x0, x1, x2, pid,
                                                            cand, cand_pid, wc);
            }
            // The divisions are the dearest part of a triangle test and one pair in five gets that far: the
            // survivors are queued and divided when the queue cannot take the next batch's, not a few per batch.
            const unsigned long long qm = __ballot(queued);
            if (qm != 0) {
                const int n_new = __popcll(qm);
                if (q_tail - q_head + n_new > RT_CAND_RING) {
                    wave_lds_sync();
                    finish_candidates(l, q_head, q_tail - q_head);
                    q_head = q_tail;
                    wave_lds_sync();
                }
                if (queued) {
                    const int at = (q_tail + __popcll(qm & ((1ull << lane) - 1ull))) & (RT_CAND_RING - 1);
                    l.cand_n[at] = cand;
                    l.cand_who[at] = make_int2(owner, cand_pid);
                }
                q_tail += n_new;
            }
            if (AHEAD) {
                owner = owner_n; pid = pid_n;
                q0 = n0; q1 = n1; q2 = n2;
                x0 = y0; x1 = y1; x2 = y2;
            }
        }
        if (q_tail > q_head) {                       // at most 64 left
            wave_lds_sync();
            finish_candidates(l, q_head, q_tail - q_head);
        }
    }
    wave_lds_sync();                                 // the atomic minima of all lanes land before the owners read them
    const unsigned long long c2 = COUNT ? __builtin_readcyclecounter() : 0ull;
    if (pending) {
        accept_under(l, r, w.t_exit);
        w.after_tests(g, r.found, r.best_t, any_hit);
    }
    if (COUNT) {
        const unsigned long long c3 = __builtin_readcyclecounter();
        wc.c_walk += c1 - c0; wc.c_test += c2 - c1; wc.c_fold += c3 - c2;
    }
}

// Traces the wave's 64 rays to completion (a lane without a ray passes active = false).  Returns per lane
// whether its ray hit; t_hit / prim_hit are the accepted hit.  Must be called by all 64 lanes together.
template <int MODE, bool ANY, bool SINGLE_XF, bool COUNT>
__device__ bool trace_wave(const DevScene& sc, const DevGrid& g, const DevSparse& sp, const Occupancy& occ, const XfRegs& xf, WaveLds& l,
                           f3 O, f3 D, float t_limit, bool active, float& t_hit, int& prim_hit, WorkCounters& wc) {
    Walker<MODE> w;
    LaneSearch r;
    w.init(g, O, D, t_limit, false);
    r.found = false; r.best_t = FLT_MAX; r.best_prim = -1;
    if (active) ray_begin<MODE, SINGLE_XF, COUNT>(sc, g, xf, l, w, r, O, D, t_limit, wc);
    while (__any(w.state != 2)) wave_round<MODE, SINGLE_XF, COUNT>(sc, g, sp, occ, xf, l, w, r, ANY, MODE == RT_SEARCH_REF ? 4 : 8, wc);
    t_hit = r.best_t;
    prim_hit = r.best_prim;
    return w.hit;
}

}  // namespace rt
