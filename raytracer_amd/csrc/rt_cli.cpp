// rt -- command-line renderer: reads a CSE167x-style .test scene and writes the image it names.
//
// Stands where the reference's main()/raytrace() stand (src/main.cpp:109-163 of the reference tree):
// parse, render, save to the scene's `output` name or "raytrace.png".  Differences on purpose: the scene
// is an argument (the reference hard-codes res/scene7.test and needs a recompile to change it, :148-160),
// nothing is displayed (Image3f::display blocks on a key press, src/Image3f.h:121-126), and the pixel
// loop runs on the GPU through the C ABI of rtx.h.
#include "rtx.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static int fail(const char* what, int rc) {
    fprintf(stderr, "rt: %s: %s (%s)\n", what, rtx_strerror(rc), rtx_last_error());
    return 1;
}

int main(int argc, char** argv) {
    std::string scene_path = "res/scene7.test", out_path;
    int width = 0, height = 0, max_depth = 5, device = 0, sub = 0;
    int search = RTX_SEARCH_NESTED;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--size" && i + 2 < argc) { width = atoi(argv[++i]); height = atoi(argv[++i]); }
        else if (a == "--maxdepth" && i + 1 < argc) max_depth = atoi(argv[++i]);
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--sub" && i + 1 < argc) sub = atoi(argv[++i]);
        else if (a == "--out" && i + 1 < argc) out_path = argv[++i];
        else if (a == "--search" && i + 1 < argc) {
            std::string s = argv[++i];
            search = (s == "reference-lists") ? RTX_SEARCH_REFERENCE_LISTS : RTX_SEARCH_NESTED;
        } else if (a == "-h" || a == "--help") {
            printf("usage: rt [scene.test] [--size W H] [--maxdepth N] [--device D] [--search nested|reference-lists]\n"
                   "          [--sub K] [--out image.png|image.ppm]\n");
            return 0;
        } else if (!a.empty() && a[0] != '-') scene_path = a;
        else { fprintf(stderr, "rt: unknown option %s\n", a.c_str()); return 2; }
    }

    rtx_scene* scene = nullptr;
    int rc = rtx_scene_load(&scene, scene_path.c_str());
    if (rc) return fail("reading scene", rc);
    printf("reading in %s: \n", scene_path.c_str());
    rtx_scene_view view;
    rtx_camera cam;
    int32_t w = 0, h = 0;
    rtx_scene_get_view(scene, &view);
    rtx_scene_get_camera(scene, &cam);
    rtx_scene_get_size(scene, &w, &h);
    if (width > 0 && height > 0) { w = width; h = height; }
    if (w <= 0 || h <= 0) { fprintf(stderr, "rt: scene has no `size` and none was given\n"); return 2; }
    printf("Image %d %d\n", w, h);

    rtx_ctx* ctx = nullptr;
    if ((rc = rtx_create(&ctx, device))) return fail("creating device context", rc);
    if ((rc = rtx_set_scene(ctx, &view))) return fail("uploading scene", rc);
    if ((rc = rtx_build_grid(ctx, 15, 15, 15, sub, RTX_GRID_REFERENCE | RTX_GRID_NESTED))) return fail("building grid", rc);
    rtx_grid_info gi;
    rtx_get_grid_info(ctx, &gi);
    printf("grid 15^3 (+ nested x%d): %lld / %lld references, built in %.2f ms\n", gi.sub, (long long)gi.n_refs,
           (long long)gi.n_fine_refs, gi.build_ms);

    std::vector<float> rgb((size_t)w * h * 3);
    rtx_render_params p;
    memset(&p, 0, sizeof p);
    p.width = w; p.height = h;
    p.max_depth = max_depth;
    p.eps_bias = 0.001f;
    p.search = search;
    p.tile_rows = 8; p.rank = 0; p.world = 1;
    rtx_stats st;
    printf("start raytrace\n");
    auto t0 = std::chrono::high_resolution_clock::now();
    if ((rc = rtx_render(ctx, &cam, &p, rgb.data(), nullptr, &st))) return fail("rendering", rc);
    auto t1 = std::chrono::high_resolution_clock::now();
    const double sec = std::chrono::duration<double>(t1 - t0).count();
    const long long rays = (long long)(st.primary_rays + st.shadow_rays + st.reflection_rays);
    printf("finished raytracing after %g seconds (GPU frame %.3f ms, %lld rays, %.1f Mrays/s)\n", sec, st.frame_ms, rays,
           rays / (st.frame_ms * 1e-3) / 1e6);

    std::string out = !out_path.empty() ? out_path : rtx_scene_get_output(scene);
    if (out.empty()) out = "raytrace.png";
    if ((rc = rtx_save_image(out.c_str(), rgb.data(), w, h))) return fail("saving image", rc);
    printf("Written to: %s\n", out.c_str());
    rtx_destroy(ctx);
    rtx_scene_free(scene);
    return 0;
}
