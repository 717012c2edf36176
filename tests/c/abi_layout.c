/* Prints the size of every struct of include/rtx.h and the offset of each of its fields, as a C11 compiler lays them
 * out: "struct <name> <size>" then "<struct>.<field> <offset> <size>" per field.  tests/test_abi.py compares the lines
 * with the ctypes mirrors of raytracer_amd/api.py.  Compiling this file with a C compiler also proves the header is C. */
#include <stddef.h>
#include <stdio.h>

#include "rtx.h"

#define S(T) printf("struct %s %zu\n", #T, sizeof(T))
#define F(T, f) printf("%s.%s %zu %zu\n", #T, #f, offsetof(T, f), sizeof(((T*)0)->f))

int main(void) {
    S(rtx_scene_view);
    F(rtx_scene_view, n_prims); F(rtx_scene_view, n_xforms); F(rtx_scene_view, n_materials); F(rtx_scene_view, n_lights);
    F(rtx_scene_view, prim_kind); F(rtx_scene_view, prim_geom); F(rtx_scene_view, prim_xform); F(rtx_scene_view, prim_material);
    F(rtx_scene_view, xforms); F(rtx_scene_view, materials); F(rtx_scene_view, lights);
    S(rtx_camera);
    F(rtx_camera, eye); F(rtx_camera, center); F(rtx_camera, up); F(rtx_camera, fov_deg);
    S(rtx_grid_info);
    F(rtx_grid_info, lo); F(rtx_grid_info, hi); F(rtx_grid_info, rx); F(rtx_grid_info, ry); F(rtx_grid_info, rz); F(rtx_grid_info, sub);
    F(rtx_grid_info, n_cells); F(rtx_grid_info, n_refs); F(rtx_grid_info, n_fine_cells); F(rtx_grid_info, n_fine_refs);
    F(rtx_grid_info, n_occupied_fine_cells); F(rtx_grid_info, build_ms); F(rtx_grid_info, skips_repeated_lists);
    S(rtx_render_params);
    F(rtx_render_params, width); F(rtx_render_params, height); F(rtx_render_params, max_depth); F(rtx_render_params, eps_bias);
    F(rtx_render_params, search); F(rtx_render_params, tile_rows); F(rtx_render_params, rank); F(rtx_render_params, world);
    F(rtx_render_params, flags); F(rtx_render_params, workgroups_per_cu); F(rtx_render_params, walk_budget);
    F(rtx_render_params, ready_threshold); F(rtx_render_params, drain_eighths); F(rtx_render_params, path_lanes);
    F(rtx_render_params, round_limit); F(rtx_render_params, endgame_eighths); F(rtx_render_params, reserved1);
    S(rtx_stats);
    F(rtx_stats, primary_rays); F(rtx_stats, shadow_rays); F(rtx_stats, reflection_rays); F(rtx_stats, hits_per_level);
    F(rtx_stats, tri_tests); F(rtx_stats, sphere_tests); F(rtx_stats, xform_evals); F(rtx_stats, cell_steps); F(rtx_stats, fine_steps);
    F(rtx_stats, light_evals); F(rtx_stats, nearest_hits); F(rtx_stats, trace_work); F(rtx_stats, shade_work);
    F(rtx_stats, wave_events); F(rtx_stats, wave_cycles); F(rtx_stats, frame_ms); F(rtx_stats, trace_ms); F(rtx_stats, shade_ms);
    F(rtx_stats, resolve_ms); F(rtx_stats, level_trace_ms); F(rtx_stats, level_shade_ms); F(rtx_stats, trace_launches);
    F(rtx_stats, shade_launches); F(rtx_stats, resolve_launches); F(rtx_stats, pipeline); F(rtx_stats, traced_primary_rays);
    F(rtx_stats, culled_pixels);
    return 0;
}
