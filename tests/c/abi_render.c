/* A plain C11 caller of include/rtx.h, the way the reference's raytrace() (src/main.cpp:109-146 of the reference tree)
 * would drive the library: load the scene file, create a context, upload, build the grid, render the pixel loop
 * (src/main.cpp:126-133) and write the float framebuffer and the primary-hit id map as raw files.
 *
 *   abi_render <scene.test> <width> <height> <max_depth> <search: 0 reference lists | 1 nested> <out.rgb> <out.ids>
 *
 * Exit status 0 on success; on a failing call the code and rtx_last_error() go to stderr and the status is 1. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtx.h"

#define TRY(call)                                                                                              \
    do {                                                                                                       \
        int rc_ = (call);                                                                                      \
        if (rc_ != RTX_OK) {                                                                                   \
            fprintf(stderr, "%s -> %d (%s): %s\n", #call, rc_, rtx_strerror(rc_), rtx_last_error());           \
            return 1;                                                                                          \
        }                                                                                                      \
    } while (0)

static int dump(const char* path, const void* data, size_t bytes) {
    FILE* f = fopen(path, "wb");
    if (!f) return 1;
    const size_t n = fwrite(data, 1, bytes, f);
    return fclose(f) != 0 || n != bytes;
}

int main(int argc, char** argv) {
    if (argc != 8) {
        fprintf(stderr, "usage: %s scene.test width height max_depth search out.rgb out.ids\n", argv[0]);
        return 2;
    }
    const int width = atoi(argv[2]), height = atoi(argv[3]), max_depth = atoi(argv[4]), search = atoi(argv[5]);
    rtx_scene* scene = NULL;
    rtx_ctx* ctx = NULL;
    rtx_scene_view view;
    rtx_camera cam;
    rtx_render_params p;
    rtx_stats st;
    TRY(rtx_scene_load(&scene, argv[1]));
    TRY(rtx_scene_get_view(scene, &view));
    TRY(rtx_scene_get_camera(scene, &cam));
    TRY(rtx_create(&ctx, 0));
    TRY(rtx_set_scene(ctx, &view));
    TRY(rtx_build_grid(ctx, 15, 15, 15, 0, search == RTX_SEARCH_NESTED ? RTX_GRID_NESTED : RTX_GRID_REFERENCE));   /* src/grid.h:37 */
    memset(&p, 0, sizeof p);
    p.width = width;
    p.height = height;
    p.max_depth = max_depth;          /* src/main.cpp:131 */
    p.eps_bias = 0.001f;              /* src/readScene.h:37 */
    p.search = search;
    p.round_limit = 1 << 23;          /* a kernel that would never end gives the frame up instead of hanging the machine */
    const size_t pixels = (size_t)width * (size_t)height;
    float* rgb = (float*)malloc(pixels * 3 * sizeof(float));
    int32_t* ids = (int32_t*)malloc(pixels * sizeof(int32_t));
    if (!rgb || !ids) return 1;
    TRY(rtx_render(ctx, &cam, &p, rgb, ids, &st));
    printf("rays %lld %lld %lld\n", (long long)st.primary_rays, (long long)st.shadow_rays, (long long)st.reflection_rays);
    if (dump(argv[6], rgb, pixels * 3 * sizeof(float)) || dump(argv[7], ids, pixels * sizeof(int32_t))) {
        fprintf(stderr, "writing the output files failed\n");
        return 1;
    }
    free(rgb);
    free(ids);
    rtx_destroy(ctx);
    rtx_scene_free(scene);
    return 0;
}
