// rt -- command-line renderer: reads a CSE167x-style .test scene and writes the image it names.
//
// Stands where the reference's main()/raytrace() stand (src/main.cpp:109-163 of the reference tree):
// parse, render, save to the scene's `output` name or "raytrace.png".  Differences on purpose: the scene
// is an argument (the reference hard-codes res/scene7.test and needs a recompile to change it, :148-160),
// nothing is displayed (Image3f::display blocks on a key press, src/Image3f.h:121-126), the constants the
// reference compiles in are options with the reference's values as defaults (recursion depth 5, :131; grid
// resolution 15, src/grid.h:37; epsilonBias 0.001, src/readScene.h:37), and the pixel loop runs on one or
// several GPUs through the C ABI of rtx.h.
#include "rtx.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

static int fail(const char* what, int rc) {
    fprintf(stderr, "rt: %s: %s (%s)\n", what, rtx_strerror(rc), rtx_last_error());
    return 1;
}

static void usage() {
    printf("usage: rt [scene.test] [--size W H] [--maxdepth N] [--grid R] [--sub K] [--bias E]\n"
           "          [--search nested|reference-lists] [--gpus N | --devices a,b,...] [--device D]\n"
           "          [--out image.png|image.ppm] [--frames K] [--headless] [--timing]\n"
           "  --maxdepth N   recursion budget of trace() (reference: 5)\n"
           "  --grid R       cells per axis of the regular grid (reference: 15)\n"
           "  --sub K        fine cells per grid cell and axis of the nested search (0 = automatic)\n"
           "  --bias E       offset of shadow and mirror ray origins (reference: 0.001)\n"
           "  --gpus N       render on GPUs 0..N-1 (interleaved row tiles, gathered on GPU 0)\n"
           "  --devices ...  the same with an explicit device list (an id may repeat)\n"
           "  --frames K     render the frame K times and report the last one's time (warm-up for timing)\n"
           "  --headless     accepted for compatibility: nothing is ever displayed\n"
           "  --timing       print the wall time of every phase of the run\n");
}

int main(int argc, char** argv) {
    std::string scene_path = "res/scene7.test", out_path;
    int width = 0, height = 0, max_depth = 5, device = 0, sub = 0, grid = 15, frames = 1;
    float bias = 0.001f;
    int search = RTX_SEARCH_NESTED;
    bool timing = false;
    std::vector<int32_t> devices;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--size" && i + 2 < argc) { width = atoi(argv[++i]); height = atoi(argv[++i]); }
        else if (a == "--maxdepth" && i + 1 < argc) max_depth = atoi(argv[++i]);
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--sub" && i + 1 < argc) sub = atoi(argv[++i]);
        else if (a == "--grid" && i + 1 < argc) grid = atoi(argv[++i]);
        else if (a == "--bias" && i + 1 < argc) bias = (float)atof(argv[++i]);
        else if (a == "--frames" && i + 1 < argc) frames = atoi(argv[++i]);
        else if (a == "--out" && i + 1 < argc) out_path = argv[++i];
        else if (a == "--headless") {}
        else if (a == "--timing") timing = true;
        else if (a == "--gpus" && i + 1 < argc) {
            const int n = atoi(argv[++i]);
            devices.clear();
            for (int k = 0; k < n; k++) devices.push_back(k);
        } else if (a == "--devices" && i + 1 < argc) {
            devices.clear();
            for (char* tok = strtok(argv[++i], ","); tok; tok = strtok(nullptr, ",")) devices.push_back(atoi(tok));
        } else if (a == "--search" && i + 1 < argc) {
            std::string s = argv[++i];
            search = (s == "reference-lists") ? RTX_SEARCH_REFERENCE_LISTS : RTX_SEARCH_NESTED;
        } else if (a == "-h" || a == "--help") {
            usage();
            return 0;
        } else if (!a.empty() && a[0] != '-') scene_path = a;
        else { fprintf(stderr, "rt: unknown option %s\n", a.c_str()); return 2; }
    }
    if (frames < 1) frames = 1;
    // --timing: milliseconds of wall time per phase of the process, printed at the end
    using clock = std::chrono::steady_clock;
    const clock::time_point t_start = clock::now();
    std::vector<std::pair<const char*, double>> phases;
    clock::time_point t_phase = t_start;
    const auto phase_done = [&](const char* name) {
        const clock::time_point now = clock::now();
        phases.emplace_back(name, std::chrono::duration<double, std::milli>(now - t_phase).count());
        t_phase = now;
    };

    rtx_scene* scene = nullptr;
    int rc = rtx_scene_load(&scene, scene_path.c_str());
    if (rc) return fail("reading scene", rc);
    printf("reading in %s: \n", scene_path.c_str());
    phase_done("parse");
    rtx_scene_view view;
    rtx_camera cam;
    int32_t w = 0, h = 0;
    rtx_scene_get_view(scene, &view);
    rtx_scene_get_camera(scene, &cam);
    rtx_scene_get_size(scene, &w, &h);
    if (width > 0 && height > 0) { w = width; h = height; }
    if (w <= 0 || h <= 0) { fprintf(stderr, "rt: scene has no `size` and none was given\n"); return 2; }
    printf("Image %d %d\n", w, h);

    const bool multi = !devices.empty();
    // each search needs its own structure only: the nested grid, or the reference's cell lists
    const uint32_t grid_flags = search == RTX_SEARCH_NESTED ? RTX_GRID_NESTED : RTX_GRID_REFERENCE;
    rtx_ctx* ctx = nullptr;
    rtx_multi* many = nullptr;
    rtx_grid_info gi;
    if (multi) {
        if ((rc = rtx_multi_create(&many, devices.data(), (int32_t)devices.size()))) return fail("creating device contexts", rc);
        phase_done("device contexts (HIP start-up)");
        if ((rc = rtx_multi_set_scene(many, &view))) return fail("uploading scene", rc);
        phase_done("scene upload");
        if ((rc = rtx_multi_build_grid(many, grid, grid, grid, sub, grid_flags))) return fail("building grid", rc);
        phase_done("grid build");
        rtx_multi_get_grid_info(many, &gi);
    } else {
        if ((rc = rtx_create(&ctx, device))) return fail("creating device context", rc);
        phase_done("device context (HIP start-up)");
        if ((rc = rtx_set_scene(ctx, &view))) return fail("uploading scene", rc);
        phase_done("scene upload");
        if ((rc = rtx_build_grid(ctx, grid, grid, grid, sub, grid_flags))) return fail("building grid", rc);
        phase_done("grid build");
        rtx_get_grid_info(ctx, &gi);
    }
    if (search == RTX_SEARCH_NESTED)
        printf("grid %d^3 with %d^3 nested cells each: %lld references, built in %.2f ms%s\n", grid, gi.sub, (long long)gi.n_fine_refs,
               gi.build_ms, multi ? " on every GPU" : "");
    else
        printf("grid %d^3: %lld references, built in %.2f ms%s\n", grid, (long long)gi.n_refs, gi.build_ms, multi ? " on every GPU" : "");

    // the saved image is 8 bits per channel (Image3f::save, src/Image3f.h:128-133): it is converted on the GPU and
    // only those bytes cross to the host
    std::vector<uint8_t> rgb8((size_t)w * h * 3);
    rtx_render_params p;
    memset(&p, 0, sizeof p);
    p.width = w; p.height = h;
    p.max_depth = max_depth;
    p.eps_bias = bias;
    p.search = search;
    p.tile_rows = 8; p.rank = 0; p.world = 1;
    rtx_stats st;
    printf("start raytrace\n");
    double sec = 0;
    for (int k = 0; k < frames; k++) {
        auto t0 = std::chrono::high_resolution_clock::now();
        rc = multi ? rtx_multi_render_u8(many, &cam, &p, rgb8.data(), RTX_ORDER_RGB, nullptr, &st)
                   : rtx_render_u8(ctx, &cam, &p, rgb8.data(), RTX_ORDER_RGB, nullptr, &st);
        if (rc) return fail("rendering", rc);
        auto t1 = std::chrono::high_resolution_clock::now();
        sec = std::chrono::duration<double>(t1 - t0).count();
    }
    phase_done(frames > 1 ? "render + fetch (all frames)" : "render + fetch");
    const long long rays = (long long)(st.primary_rays + st.shadow_rays + st.reflection_rays);
    printf("finished raytracing after %g seconds (%s %.3f ms, %lld rays, %.1f Mrays/s%s)\n", sec,
           multi ? "slowest GPU's kernel" : "GPU frame", multi ? st.trace_ms : st.frame_ms, rays,
           rays / ((multi ? st.trace_ms : st.frame_ms) * 1e-3) / 1e6, multi ? ", gather included in the seconds" : "");

    std::string out = !out_path.empty() ? out_path : rtx_scene_get_output(scene);
    if (out.empty()) out = "raytrace.png";
    if ((rc = rtx_save_image_u8(out.c_str(), rgb8.data(), w, h))) return fail("saving image", rc);
    printf("Written to: %s\n", out.c_str());
    phase_done("image file");
    if (many) rtx_multi_destroy(many);
    if (ctx) rtx_destroy(ctx);
    rtx_scene_free(scene);
    phase_done("teardown");
    if (timing) {
        printf("wall time per phase (ms):");
        for (const auto& ph : phases) printf("  %s %.1f", ph.first, ph.second);
        printf("  | total %.1f\n", std::chrono::duration<double, std::milli>(clock::now() - t_start).count());
    }
    return 0;
}
