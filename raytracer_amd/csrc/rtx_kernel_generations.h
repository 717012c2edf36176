// Generation pipeline: the persistent frame kernel with its state updates run on full waves.
// Part of the device layer of librtx; included by rtx_device.hip only (one translation unit).
#pragma once

// ---------------------------------------------------------------------------------------------------
// In k_frame every lane carries one path through its states, so a state update finds the wave's lanes in every
// state at once: each of its sections (finish a hit, evaluate a light, mirror ray, fold, next pixel) runs for
// the handful of lanes that need it and costs the wave its full length.  Here a wave works in generations of
// 64 rays of ONE kind:
//   N generation  nearest-hit rays: mirror rays popped from the wave's M stack, the remaining lanes draw new
//                 pixels (camera rays).  All 64 are traced to their end; then every hit becomes a shadow request
//                 on the wave's S stack (position, normal, incoming direction, material, the path's terms so
//                 far), every miss ends its path (src/main.cpp:104-106): the path's terms are folded, the pixel
//                 is written.
//   S generation  64 shadow requests popped from the S stack: one pass of shadow rays per light
//                 (src/main.cpp:53-75), then the level's term (:97) and the mirror ray (:85-94) as a request on
//                 the M stack, or the fold when the path is at its last level.
// A path is not bound to a lane: between generations it lives in its request -- a record in device memory that
// carries the terms of the levels above it -- so each section of a generation's prologue and epilogue runs on
// (nearly) all 64 lanes.  The stacks are private to the wave (no cross-wave traffic, no atomics), last in first
// out (deep paths first: the live records of a wave stay few and cached).  Every ray, hit, term and fold is the
// arithmetic of k_frame, which is the reference's; only the schedule differs, so frames are bit-identical.
//
// Record layout (structure of arrays per stack: field k of slot s is q[k * RTX_GEN_CAP + s], so a generation's
// pops and pushes are contiguous 1 KB reads and writes per field):
//   S request: field 0 = (P, material << 4 | level)   1 = (N, tile slot)      2 = (incoming direction, pixel)
//   M request: field 0 = (O, tile slot << 4 | level)  1 = (R, pixel)          2 unused
//   fields 3 .. 3 + n_terms - 1: the terms (L xyz, material) of levels 0 .. level - 1
// (level < 16, tile slot < 2^22 and material < 2^27 are the library's limits on max_depth, pixels and materials.)
// Capacity: an S generation runs as soon as 64 requests wait, an N generation adds at most 64: fewer than 128
// S requests at any time; an S generation adds at most 64 M requests and the next N generation takes them all.
// ---------------------------------------------------------------------------------------------------
#define RTX_GEN_CAP 128
#define RTX_GEN_HEADER 3
#define RTX_GEN_REG_TERMS 6         // terms of a path that travel through registers when its record is rewritten

template <int MODE, bool SINGLE_XF, bool COUNT, int PRIMS, int WAVES>
__global__ __launch_bounds__(RTX_BLOCK, WAVES) void k_frame_gen(DevScene sc, DevGrid g, DevSparse sp, DevFrame f, float4* queues, int n_terms) {
    __shared__ WaveLds lds[RTX_BLOCK / 64];
    __shared__ int hit_count[RTX_BLOCK / 64][RTX_MAX_LEVELS];
    __shared__ unsigned occ_lds[RT_OCC_LDS_WORDS];
    const Occupancy occ = stage_occupancy(g, occ_lds);
    const XfRegs xf = {};
    const int wave = (int)(threadIdx.x >> 6), lane = wave_lane();
    WaveLds& l = lds[wave];
    if (lane < RTX_MAX_LEVELS) hit_count[wave][lane] = 0;
    const size_t stack_f4 = (size_t)RTX_GEN_CAP * (RTX_GEN_HEADER + n_terms);
    float4* const qS = queues + (size_t)(blockIdx.x * (RTX_BLOCK / 64) + wave) * 2 * stack_f4;
    float4* const qM = qS + stack_f4;
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
    const bool mixing = n_terms <= RTX_GEN_REG_TERMS;   // (deeper terms are copied record to record, pure generations only)

    for (;;) {
        // ---- which generation ----------------------------------------------------------------------------------
        int nS = 0, nM = 0, nC = 0;
        if (S_n >= 64) nS = 64;
        else {
            nM = min(M_n, 64);
            nC = exhausted ? 0 : min(64 - nM, new_paths);
            // Toward the end of the frame the wave drains: what is left of both stacks goes into the same generation (no
            // path waits for a generation of the other kind) and new pixels only fill the lanes that are left, so that
            // the last pixels go to the waves that have run out of work, not onto the stacks of those that have not.
            if ((exhausted || endgame) && mixing) { nS = min(S_n, 64 - nM); nC = min(nC, 64 - nM - nS); }
            else if (exhausted && S_n > 0 && S_n >= M_n) { nS = min(S_n, 64); nM = 0; }
            if (nS + nM + nC == 0) break;               // nothing left on either stack, no pixels left to draw
        }
        S_n -= nS; M_n -= nM;
        // (a generation pushes at most 64 requests onto either stack and the bounds above keep that much room.  The
        // instrumented build, which the tests run, checks it: a change of the policy that broke the bounds gives the
        // frame up there.  In the plain build the same check costs 3 % of the frame -- the extra exit from the loop
        // moves the register allocation of the rounds.)
        if (COUNT && (S_n > RTX_GEN_CAP - 64 || M_n > RTX_GEN_CAP - 64)) {
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
        f3 rayDir = mk(0, 0, 1), O = mk(0, 0, 0), D = mk(0, 0, 1);
        float tl = FLT_MAX;
        if (lane < nS) {
            kind = 2; src = S_n + lane;
            const float4 h2 = qS[2 * RTX_GEN_CAP + src];
            rayDir = mk(h2.x, h2.y, h2.z); pix = as_int(h2.w);
        } else if (lane < nS + nM) {
            kind = 1; src = M_n + lane - nS;
            const float4 h0 = qM[src], h1 = qM[RTX_GEN_CAP + src];
            O = mk(h0.x, h0.y, h0.z);
            level = as_int(h0.w) & 15; tslot = (int)((unsigned)as_int(h0.w) >> 4);
            rayDir = mk(h1.x, h1.y, h1.z); pix = as_int(h1.w);
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
                        kind = 1; src = -1; level = 0;
                    }
                }
            }
            if (base + nC >= n_items) {
                if (COUNT && !exhausted) log_exhausted = wall_clock64();
                exhausted = true;
            }
            // the last pixels of the frame are shared out among the waves (at least 32 a generation)
            const int left = max(n_items - (base + nC), 0);
            const int waves = (int)gridDim.x * (RTX_BLOCK / 64);
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
            // register allocator then spills inside the walk and test loops -- 138 scratch accesses in the kernel against
            // 29, all of which now sit in the generation boundaries.)
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
        // a path that ends folds its terms from the deepest level up (src/main.cpp:99-102) and writes its pixel
        const float4* const q_src = kind == 2 ? qS : qM;
        const auto end_path = [&](f3 C, int top_below) {      // C: the fold so far; terms top_below .. 0 come from the record
            for (int d = top_below; d >= 0; d--) {
                const float4 s4 = q_src[(RTX_GEN_HEADER + d) * RTX_GEN_CAP + src];
                const float* m = su.mats + 16 * as_int(s4.w);
                C = clamp_rgb(mk(s4.x, s4.y, s4.z) + mk(m[6], m[7], m[8]) * C);
            }
            fu.rgb[3 * pix] = C.x;
            fu.rgb[3 * pix + 1] = C.y;
            fu.rgb[3 * pix + 2] = C.z;
        };
        bool push_s = false, push_m = false;
        f3 P = mk(0, 0, 0), N = mk(0, 0, 1);
        int material = 0;
        float4 rec = make_float4(0, 0, 0, 0);
        if (kind == 1) {
            if (!w.hit) {                                   // src/main.cpp:104-106
                if (level == 0 && fu.ids) fu.ids[pix] = -1;
                end_path(mk(0, 0, 0), level - 1);
                if (fu.tile_cost && fu.tile_cost[tslot] < level + 1) atomicMax(fu.tile_cost + tslot, level + 1);
                if (COUNT) log_deepest = max(log_deepest, level);
            } else {
                if (COUNT) wc.nearest_hits++;
                const int prim = r.best_prim;
                finish_hit<SINGLE_XF, PRIMS>(su, prim, w.O, w.D, r.best_t, P, N, material);
                atomicAdd(&hit_count[wave][level], 1);
                if (level == 0 && fu.ids) fu.ids[pix] = prim;
                push_s = true;
            }
        } else if (kind == 2) {
            const float4 h0 = qS[src], h1 = qS[RTX_GEN_CAP + src];
            P = mk(h0.x, h0.y, h0.z); N = mk(h1.x, h1.y, h1.z);
            material = (int)((unsigned)as_int(h0.w) >> 4); level = as_int(h0.w) & 15;
            tslot = as_int(h1.w);
            // the level's term (src/main.cpp:97-102)
            const float* m = su.mats + 16 * material;
            const f3 Lsum = mk(m[0], m[1], m[2]) + mk(m[9], m[10], m[11]) + clamp_rgb(acc);
            rec = make_float4(Lsum.x, Lsum.y, Lsum.z, as_float(material));
            if (level < fu.max_depth) {
                push_m = true;
            } else {
                end_path(clamp_rgb(Lsum + mk(m[6], m[7], m[8]) * mk(0, 0, 0)), level - 1);
                if (fu.tile_cost && fu.tile_cost[tslot] < level + 2) atomicMax(fu.tile_cost + tslot, level + 2);
                if (COUNT) log_deepest = max(log_deepest, level + 1);
            }
        }
        // Pushes: the k-th pushing lane takes the k-th free slot of the other kind's stack.  In a mixed generation those
        // slots may be the very records lanes of the other kind are still reading, so every lane first reads what it
        // carries over (the terms of the levels above, through registers), then all lanes write.
        const unsigned long long mask_s = __ballot(push_s), mask_m = __ballot(push_m);
        float4 carried[RTX_GEN_REG_TERMS];
#pragma unroll
        for (int d = 0; d < RTX_GEN_REG_TERMS; d++) {
            carried[d] = make_float4(0, 0, 0, 0);
            if ((push_s || push_m) && d < level) carried[d] = q_src[(RTX_GEN_HEADER + d) * RTX_GEN_CAP + src];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        float4* const q_dst = push_s ? qS : qM;
        int at = 0;
        if (push_s) {
            at = S_n + __popcll(mask_s & below);
            qS[at] = make_float4(P.x, P.y, P.z, as_float((material << 4) | level));
            qS[RTX_GEN_CAP + at] = make_float4(N.x, N.y, N.z, as_float(tslot));
            qS[2 * RTX_GEN_CAP + at] = make_float4(rayDir.x, rayDir.y, rayDir.z, as_float(pix));
        }
        if (push_m) {                                        // src/main.cpp:85-94
            const f3 V = normalize3(-rayDir);
            const f3 R = (2.0f * dot3(V, N) * N) - V;
            const f3 On = P + fu.bias * R;
            at = M_n + __popcll(mask_m & below);
            qM[at] = make_float4(On.x, On.y, On.z, as_float((tslot << 4) | (level + 1)));
            qM[RTX_GEN_CAP + at] = make_float4(R.x, R.y, R.z, as_float(pix));
            qM[(RTX_GEN_HEADER + level) * RTX_GEN_CAP + at] = rec;
        }
        if (push_s || push_m) {
#pragma unroll
            for (int d = 0; d < RTX_GEN_REG_TERMS; d++)
                if (d < level) q_dst[(RTX_GEN_HEADER + d) * RTX_GEN_CAP + at] = carried[d];
            for (int d = RTX_GEN_REG_TERMS; d < level; d++)
                q_dst[(RTX_GEN_HEADER + d) * RTX_GEN_CAP + at] = q_src[(RTX_GEN_HEADER + d) * RTX_GEN_CAP + src];
        }
        S_n += __popcll(mask_s);
        M_n += __popcll(mask_m);
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
            unsigned long long* log = f.work + 24 + 8 * (size_t)(blockIdx.x * (RTX_BLOCK / 64) + wave);
            log[0] = log_start; log[1] = log_exhausted; log[2] = wall_clock64(); log[3] = (unsigned long long)log_rounds;
            log[4] = (unsigned long long)log_updates; log[5] = (unsigned long long)pixels; log[6] = (unsigned long long)deepest;
            log[7] = 0ull;
        }
    }
}
