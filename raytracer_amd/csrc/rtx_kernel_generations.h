// Generation pipeline: the persistent frame kernel with its state updates run on full waves.
// Part of the device layer of librtx; included by rtx_device.hip only (one translation unit).
#pragma once

// ---------------------------------------------------------------------------------------------------
// In k_frame every lane carries one path through its states, so a state update finds the wave's lanes in every
// state at once: each of its sections (finish a hit, evaluate a light, mirror ray, fold, next pixel) runs for
// the handful of lanes that need it and costs the wave its full length.  Here a wave works in generations of
// 64 rays of ONE kind:
//   N generation  nearest-hit rays: mirror rays popped from the wave's M stack, the remaining lanes draw new
//                 pixels (camera rays).  All 64 are traced to their end; then every hit pushes BOTH requests that
//                 leave it -- the shadow request of its level (src/main.cpp:97) onto the S stack and, above the
//                 last level, the mirror request (:85-94) onto the M stack: the two do not depend on each other
//                 (trace() merely evaluates them one after the other), so the path's next nearest-hit ray is in
//                 the very next N generation; every miss ends its path's chain (:104-106).
//   S generation  64 shadow requests popped from the S stack: one pass of shadow rays per light (:53-75), then the
//                 level's term (:97-102) goes into the path's slot.  An S generation pushes nothing.
// A path is not bound to a lane.  While it lives it owns a slot of the wave's path pool: its per-level terms in
// device memory, its pixel, and one word of LDS that counts what is still outstanding for it (shadow requests not
// yet evaluated, plus one while its chain of nearest-hit rays goes on) and, once the chain has ended, holds the
// deepest level that has a term.  Whichever lane brings that count to zero -- the lane that ends the chain or the
// lane that evaluates the last outstanding shadow request -- folds the terms from the deepest level up
// (src/main.cpp:99-102), writes the pixel and frees the slot.  Stacks and pool are private to the wave (no
// cross-wave traffic), stacks are last in first out (deep paths first: the live records stay few and cached).
// Every ray, hit, term and fold is the arithmetic of k_frame, which is the reference's; only the schedule differs,
// so frames are bit-identical.
//
// Per wave in device memory (float4 units, see rtx_gen_wave_f4): S stack, 3 fields x RTX_GEN_CAP slots, structure of
// arrays (field k of slot s at k * RTX_GEN_CAP + s, so a generation's pops and pushes are contiguous 1 KB accesses per
// field); M stack, 2 fields; pixel of every path slot; terms, n_levels per path slot.
//   S request: field 0 = (P, material << 4 | level)   1 = (N, path slot)    2 = (incoming direction, -)
//   M request: field 0 = (O, path slot << 4 | level)  1 = (R, -)
//   term:      (L xyz of the level, material)
// (level < 16, path slot < 2^9 and material < 2^27 are the library's limits.)
// Capacities.  An S generation runs as soon as 64 shadow requests wait, a generation of mirror rays as soon as 64 mirror
// requests do, and a generation of either kind of nearest-hit ray adds at most 64 to each stack: fewer than 128 on
// either.  Live paths: at most the requests on both stacks plus the 64 rays of the generation in flight, fewer than
// RTX_GEN_PATHS.
// ---------------------------------------------------------------------------------------------------
// Lanes per workgroup of the generation kernel (a multiple of 64 that divides RTX_BLOCK).  Its waves share nothing but
// the 0.5 KB copy of the coarse occupancy bits, and a workgroup gives its place on the CU back only when its last wave
// has ended: with one wave per workgroup a wave that has run dry makes room at once for a wave of another frame in
// flight (three 4K frames in flight 1.587 -> 1.571 ms per frame, four halves 0.856 -> 0.842; two waves per workgroup 1.593).
#ifndef RTX_GEN_BLOCK
#define RTX_GEN_BLOCK 64
#endif
#define RTX_GEN_CAP 128
#define RTX_GEN_S_FIELDS 3
#define RTX_GEN_M_FIELDS 2
#define RTX_GEN_PATHS 320
// float4s of device memory per wave for frames of n_levels path levels
__host__ __device__ constexpr size_t rtx_gen_wave_f4(int n_levels) {
    return (size_t)RTX_GEN_CAP * (RTX_GEN_S_FIELDS + RTX_GEN_M_FIELDS) + (size_t)RTX_GEN_PATHS / 2 + (size_t)RTX_GEN_PATHS * (size_t)n_levels;
}

#ifndef RT_GEN_WAVES_MIN
#define RT_GEN_WAVES_MIN 0         // (development: hold the generation kernel's allocation to at least this many waves per SIMD)
#endif
template <int MODE, bool SINGLE_XF, bool COUNT, int PRIMS, int WAVES>
__global__ __launch_bounds__(RTX_GEN_BLOCK, (WAVES > RT_GEN_WAVES_MIN ? WAVES : RT_GEN_WAVES_MIN)) void k_frame_gen(DevScene sc, DevGrid g, DevSparse sp, DevFrame f, float4* queues, int n_levels) {
    __shared__ WaveLds lds[RTX_GEN_BLOCK / 64];
    __shared__ int hit_count[RTX_GEN_BLOCK / 64][RTX_MAX_LEVELS];
    // per path slot: low half = what is outstanding for the path, high half = 1 + its deepest level with a term (once
    // its chain has ended, 0 before)
    __shared__ int path_state[RTX_GEN_BLOCK / 64][RTX_GEN_PATHS];
    __shared__ unsigned short free_slot[RTX_GEN_BLOCK / 64][RTX_GEN_PATHS];
    __shared__ unsigned occ_lds[RT_OCC_LDS_WORDS];
    const Occupancy occ = stage_occupancy(g, occ_lds);
    const XfRegs xf = {};
    const int wave = (int)(threadIdx.x >> 6), lane = wave_lane();
    WaveLds& l = lds[wave];
    if (lane < RTX_MAX_LEVELS) hit_count[wave][lane] = 0;
    for (int i = lane; i < RTX_GEN_PATHS; i += 64) free_slot[wave][i] = (unsigned short)i;
    int n_free = RTX_GEN_PATHS;                         // free path slots (wave-uniform): free_slot[0 .. n_free)
    float4* const qS = queues + (size_t)(blockIdx.x * (RTX_GEN_BLOCK / 64) + wave) * rtx_gen_wave_f4(n_levels);
    float4* const qM = qS + RTX_GEN_CAP * RTX_GEN_S_FIELDS;
    int2* const path_pixel = (int2*)(qM + RTX_GEN_CAP * RTX_GEN_M_FIELDS);   // per path slot: local pixel index, 8x8 block of the cost map
    float4* const terms = qM + RTX_GEN_CAP * RTX_GEN_M_FIELDS + RTX_GEN_PATHS / 2;
    WorkCounters wc = {};

    Walker<MODE> w;
    LaneSearch r;
    w.init(g, mk(0, 0, 0), mk(0, 0, 1), FLT_MAX, false);
    r.found = false; r.best_t = FLT_MAX; r.best_prim = -1;
    int S_n = 0, M_n = 0;                               // requests on the wave's stacks (wave-uniform)
    bool exhausted = false;
    const int n_items = f.tile_list ? 64 * __builtin_amdgcn_readfirstlane(f.counters[CNT_TILES]) : f.n_items0;
    long long rounds_left = f.round_limit;
    int n_drawn = 0;
    int new_paths = 64;                                 // pixels an N generation may draw (wave-uniform)
    bool abandoned = false;
    bool endgame = false;                               // few pixels are left: drain instead of stocking up on paths
    unsigned long long log_start = 0, log_exhausted = 0;
    int log_rounds = 0, log_updates = 0, log_pixels = 0, log_deepest = 0;
    if (COUNT) log_start = wall_clock64();
    const unsigned long long below = (1ull << lane) - 1ull;
    fill_untraced_blocks(f, (int)blockIdx.x * (RTX_GEN_BLOCK / 64) + wave, (int)gridDim.x * (RTX_GEN_BLOCK / 64));

    for (;;) {
        // ---- which generation ----------------------------------------------------------------------------------
        int nS = 0, nM = 0, nC = 0;
        if (S_n >= 64) nS = 64;
        else if (!exhausted && !endgame) {
            // A generation of one kind while the frame lasts: 64 mirror rays once that many wait, else 64 new pixels.  (Camera
            // rays and mirror rays in one generation cost both kinds' set-up every time, and the camera rays of a tile, which
            // walk the same cells and end together, would wait for the mirror rays' longer walks.)
            if (M_n >= 64) nM = 64;
            else nC = new_paths;
        } else {
            // Toward the end of the frame the wave drains: what is left of both stacks goes into the same generation (no
            // request waits for a generation of its kind to fill) and new pixels only take the lanes that are left, so that
            // the last pixels go to the waves that have run out of work, not onto the stacks of those that have not.  A
            // path then advances one level per generation: its mirror ray and the shadow rays of the level before it
            // travel together.
            nM = min(M_n, 64);
            nS = min(S_n, 64 - nM);
            nC = exhausted ? 0 : min(new_paths, 64 - nM - nS);
            if (nS + nM + nC == 0) break;               // nothing left on either stack, no pixels left to draw
        }
        S_n -= nS; M_n -= nM;
        // (a generation of nearest-hit rays pushes at most 64 requests onto either stack and starts at most nC paths, and the
        // bounds above keep that much room.  The instrumented build, which the tests run, checks it: a change of the policy that broke the
        // bounds gives the frame up there.  In the plain build the same check costs 3 % of the frame -- the extra exit from
        // the loop moves the register allocation of the rounds.)
        if (COUNT && nM + nC > 0 && (S_n > RTX_GEN_CAP - 64 || M_n > RTX_GEN_CAP - 64 || n_free < nC)) {
            if (lane == 0) { f.counters[CNT_ABORT] = 1; *f.abort_host = 1; }
            break;
        }
        // What only the generation boundaries read of the frame, the scene and the grid comes from the kernel's
        // argument block by scalar loads here, not out of scalar registers held through the rounds.
        const __attribute__((address_space(4))) char* arg_block =
            (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(arg_block));
        const FrameKernelArgs& args = *(const FrameKernelArgs*)arg_block;
        const DevFrame& fu = args.f;
        const DevScene& su = args.sc;
        const DevGrid& gu = args.g;
        if (COUNT) { wc.w_updates++; log_updates++; }
        unsigned long long c0 = COUNT ? __builtin_readcyclecounter() : 0ull;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // the records other lanes of this wave pushed

        // ---- prologue ------------------------------------------------------------------------------------------
        // kind 0: no request this generation; 1: nearest-hit ray (mirror or camera); 2: shadow request
        int kind = 0, src = -1, level = 0, pix = -1, tslot = 0;
        int slot = -1;                                    // the path's slot (nearest-hit rays above level 0, shadow requests)
        f3 rayDir = mk(0, 0, 1), O = mk(0, 0, 0), D = mk(0, 0, 1);
        float tl = FLT_MAX;
        if (lane < nS) {
            kind = 2; src = S_n + lane;
            const float4 h2 = qS[2 * RTX_GEN_CAP + src];
            rayDir = mk(h2.x, h2.y, h2.z);
        } else if (lane < nS + nM) {
            kind = 1;
            const int from = M_n + lane - nS;
            const float4 h0 = qM[from], h1 = qM[RTX_GEN_CAP + from];
            O = mk(h0.x, h0.y, h0.z);
            level = as_int(h0.w) & 15; slot = (int)((unsigned)as_int(h0.w) >> 4);
            rayDir = mk(h1.x, h1.y, h1.z);
            D = normalize3(rayDir);                       // src/main.cpp:82
        }
        if (nC > 0) {
            int base = 0;
            if (lane == 0) base = atomicAdd(fu.counters + CNT_TRACE_CUR, nC);
            base = __builtin_amdgcn_readfirstlane(base);
            const int j = lane - nS - nM;
            if (j >= 0 && j < nC) {
                const int i = base + j;
                if (i < n_items) {
                    const int blk = i >> 6, in = i & 63;
                    int lr, x;
                    if (fu.tile_list) {
                        const int t = fu.tile_list[blk];
                        lr = 8 * (t >> 16) + (in >> 3);
                        x = 8 * (t & 0xffff) + (in & 7);
                        tslot = (t >> 16) * fu.tbx + (t & 0xffff);
                    } else {
                        lr = fu.lr0 + 8 * (blk / fu.nbx) + (in >> 3);
                        x = 8 * (fu.bx0 + blk % fu.nbx) + (in & 7);
                    }
                    if (lr < fu.n_local_rows && x < fu.W) {
                        const int tile = lr / fu.tile_rows;
                        const int y = (fu.rank + tile * fu.world) * fu.tile_rows + (lr - tile * fu.tile_rows);
                        pix = lr * fu.W + x;
                        n_drawn++;
                        if (COUNT) log_pixels++;
                        O = fu.cam.eye;
                        rayDir = camera_ray(fu.cam, x, y);
                        D = normalize3(rayDir);           // src/main.cpp:82
                        kind = 1; level = 0;
                    }
                }
            }
            if (base + nC >= n_items) {
                if (COUNT && !exhausted) log_exhausted = wall_clock64();
                exhausted = true;
            }
            // the last pixels of the frame are shared out among the waves (at least 32 a generation)
            const int left = max(n_items - (base + nC), 0);
            const int waves = (int)gridDim.x * (RTX_GEN_BLOCK / 64);
            new_paths = min(64, max(left / waves + 1, 32));
            if (left < fu.endgame_pixels) endgame = true;
        }

        // ---- the rays: one pass for the nearest-hit rays and the first light, one more per further light -----
        f3 acc = mk(0, 0, 0);
        const int n_pass = nS > 0 ? max(su.n_lights, 1) : 1;
        for (int pass = 0; pass < n_pass; pass++) {
            bool go = kind == 1 && pass == 0;
            if (kind == 2 && pass < su.n_lights) {          // src/main.cpp:53-57,68-70
                const float4 h0 = qS[src];
                const f3 P = mk(h0.x, h0.y, h0.z);
                const float* lt = su.lights + 12 * pass;
                const f3 lpos = mk(lt[1], lt[2], lt[3]);
                const bool point = (int)lt[0] == RTX_LIGHT_POINT;
                const f3 L = point ? normalize3(lpos - P) : normalize3(lpos);
                O = P + fu.bias * L;
                D = L;
                tl = point ? dot3(lpos - P, L) : FLT_MAX;
                go = true;
            }
            if (go) ray_begin<MODE, SINGLE_XF, COUNT>(su, gu, load_xf_regs(su.xinv, SINGLE_XF), l, w, r, O, D, tl, wc);
            if (COUNT) wc.c_update += __builtin_readcyclecounter() - c0;
            // (The round limit is a second way out of this loop on purpose.  Folding it into the loop's condition, or counting
            // the rounds of a generation and checking afterwards, looks tidier and makes the frame 14 - 20 % slower: the
            // register allocator then spills inside the walk and test loops.)
            while (__any(w.state != 2)) {
                if (COUNT) log_rounds++;
                wave_round<MODE, SINGLE_XF, COUNT, PRIMS, (WAVES < 4)>(sc, g, sp, occ, xf, l, w, r, kind == 2, f.walk_budget, wc);
                if (--rounds_left < 0) { abandoned = true; break; }
            }
            if (abandoned) break;
            if (COUNT) c0 = __builtin_readcyclecounter();
            if (kind == 2 && pass < su.n_lights && !w.hit) {    // src/main.cpp:57-75
                const float4 h0 = qS[src], h1 = qS[RTX_GEN_CAP + src];
                const f3 N = mk(h1.x, h1.y, h1.z);
                const int material = (int)((unsigned)as_int(h0.w) >> 4);
                const float* lt = su.lights + 12 * pass;
                const float* m = su.mats + 16 * material;
                if (COUNT) wc.light_evals++;
                const float tlim = w.t_limit;
                const f3 c = light_term(rayDir, w.D, N, mk(m[3], m[4], m[5]), mk(m[6], m[7], m[8]), m[12], mk(lt[4], lt[5], lt[6]));
                if ((int)lt[0] == RTX_LIGHT_POINT) {
                    const float att = lt[7] + lt[8] * tlim + lt[9] * tlim * tlim;
                    acc = acc + c / att;
                } else {
                    acc = acc + c;
                }
            }
        }
        if (abandoned) {
            // the frame is abandoned: pixels still in flight keep the fill colour, and the host is told
            if (lane == 0) { f.counters[CNT_ABORT] = 1; *f.abort_host = 1; }
            break;
        }

        // ---- epilogue ------------------------------------------------------------------------------------------
        // What each lane's ray means for its path: `delta` is what it adds to the path's state word -- +1 for the shadow
        // request of a hit, -1 for an evaluated shadow request, and for the lane that ends the chain -1 together with the
        // deepest level that has a term (+1, in the high half).
        bool push_s = false, push_m = false;
        int delta = 0;
        f3 P = mk(0, 0, 0), N = mk(0, 0, 1);
        int material = 0;
        if (kind == 1) {
            if (!w.hit) {                                   // src/main.cpp:104-106
                if (level == 0) {                           // a camera ray that meets nothing: the pixel is the miss colour
                    if (fu.ids) fu.ids[pix] = -1;
                    fu.rgb[3 * pix] = 0.0f;
                    fu.rgb[3 * pix + 1] = 0.0f;
                    fu.rgb[3 * pix + 2] = 0.0f;
                    record_tile_depth(fu, tslot, 1);
                } else {
                    delta = (level << 16) - 1;              // terms of levels 0 .. level - 1
                }
            } else {
                if (COUNT) wc.nearest_hits++;
                const int prim = r.best_prim;
                finish_hit<SINGLE_XF, PRIMS>(su, prim, w.O, w.D, r.best_t, P, N, material);
                atomicAdd(&hit_count[wave][level], 1);
                if (level == 0 && fu.ids) fu.ids[pix] = prim;
                push_s = true;
                push_m = level < fu.max_depth;
                delta = push_m ? 1 : ((level + 1) << 16);   // (+1 for the request, and at the last level -1 with the top level)
            }
        } else if (kind == 2) {
            const float4 h0 = qS[src], h1 = qS[RTX_GEN_CAP + src];
            material = (int)((unsigned)as_int(h0.w) >> 4); level = as_int(h0.w) & 15;
            slot = as_int(h1.w);
            // the level's term (src/main.cpp:97-102)
            const float* m = su.mats + 16 * material;
            const f3 Lsum = mk(m[0], m[1], m[2]) + mk(m[9], m[10], m[11]) + clamp_rgb(acc);
            terms[slot * n_levels + level] = make_float4(Lsum.x, Lsum.y, Lsum.z, as_float(material));
            delta = -1;
        }
        // every lane has read what it needs of the stacks' records (a generation of both kinds pushes over the very
        // records its shadow lanes popped), and the terms stored above have landed before a fold reads them
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        // paths that begin: a camera ray that hit takes a free slot; its state starts with the chain and one request
        const bool begins = push_s && level == 0;
        const unsigned long long mask_b = __ballot(begins);
        if (begins) {
            slot = (int)free_slot[wave][n_free - 1 - __popcll(mask_b & below)];
            path_state[wave][slot] = 1;
            path_pixel[slot] = make_int2(pix, tslot);
        }
        n_free -= __popcll(mask_b);
        wave_lds_sync();
        int state = 1;
        if (delta != 0) state = atomicAdd(&path_state[wave][slot], delta) + delta;
        const bool folds = delta != 0 && (state & 0xffff) == 0;
        // Pushes: the k-th pushing lane takes the k-th free slot of its stack.
        const unsigned long long mask_s = __ballot(push_s), mask_m = __ballot(push_m);
        if (push_s) {
            const int at = S_n + __popcll(mask_s & below);
            qS[at] = make_float4(P.x, P.y, P.z, as_float((material << 4) | level));
            qS[RTX_GEN_CAP + at] = make_float4(N.x, N.y, N.z, as_float(slot));
            qS[2 * RTX_GEN_CAP + at] = make_float4(rayDir.x, rayDir.y, rayDir.z, 0.0f);
        }
        if (push_m) {                                        // src/main.cpp:85-94
            const f3 V = normalize3(-rayDir);
            const f3 R = (2.0f * dot3(V, N) * N) - V;
            const f3 On = P + fu.bias * R;
            const int at = M_n + __popcll(mask_m & below);
            qM[at] = make_float4(On.x, On.y, On.z, as_float((slot << 4) | (level + 1)));
            qM[RTX_GEN_CAP + at] = make_float4(R.x, R.y, R.z, 0.0f);
        }
        S_n += __popcll(mask_s);
        M_n += __popcll(mask_m);
        // a path with nothing outstanding folds its terms from the deepest level up (src/main.cpp:99-102), writes its
        // pixel and gives its slot back
        const unsigned long long mask_f = __ballot(folds);
        if (folds) {
            const int top = (state >> 16) - 1;
            const int2 where = path_pixel[slot];
            f3 C = mk(0, 0, 0);
            for (int d = top; d >= 0; d--) {
                const float4 s4 = terms[slot * n_levels + d];
                const float* m = su.mats + 16 * as_int(s4.w);
                C = clamp_rgb(mk(s4.x, s4.y, s4.z) + mk(m[6], m[7], m[8]) * C);
            }
            fu.rgb[3 * where.x] = C.x;
            fu.rgb[3 * where.x + 1] = C.y;
            fu.rgb[3 * where.x + 2] = C.z;
            record_tile_depth(fu, where.y, top + 2);
            if (COUNT) log_deepest = max(log_deepest, top + 1);
            free_slot[wave][n_free + __popcll(mask_f & below)] = (unsigned short)slot;
        }
        n_free += __popcll(mask_f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (COUNT) wc.c_update += __builtin_readcyclecounter() - c0;
    }
    if (lane < RTX_MAX_LEVELS && hit_count[wave][lane]) atomicAdd(f.counters + CNT_HITS + lane, hit_count[wave][lane]);
    for (int off = 32; off > 0; off >>= 1) n_drawn += __shfl_down(n_drawn, off);
    if (lane == 0 && n_drawn) atomicAdd(f.counters + CNT_PIXELS, n_drawn);
    if (COUNT) {
        flush_work(wc, f.work);
        int pixels = log_pixels, deepest = log_deepest;
        for (int off = 32; off > 0; off >>= 1) {
            pixels += __shfl_down(pixels, off);
            deepest = max(deepest, __shfl_down(deepest, off));
        }
        if (lane == 0) {
            unsigned long long* log = f.work + 24 + 8 * (size_t)(blockIdx.x * (RTX_GEN_BLOCK / 64) + wave);
            log[0] = log_start; log[1] = log_exhausted; log[2] = wall_clock64(); log[3] = (unsigned long long)log_rounds;
            log[4] = (unsigned long long)log_updates; log[5] = (unsigned long long)pixels; log[6] = (unsigned long long)deepest;
            log[7] = 0ull;
        }
    }
}
