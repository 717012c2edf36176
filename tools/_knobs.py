"""Development only: environment knobs of the tools, copied into the render-call defaults of the python binding.

The library itself reads no environment variables; scheduling switches and tuning values travel in
rtx_render_params.  The A/B scripts in this directory (ab.sh, sweep.sh, ...) set RTX_* variables around a tool run;
every tool imports this module, which translates them once at start-up:

    RTX_TILE_FEEDBACK=0   no previous-frame scheduling hint        RTX_TILE_CULL=0        trace every tile
    RTX_HELPERS=0|2|3     no / early / late helpers                         RTX_FRAME_WAVES=3|4    force a register budget
    RTX_WALK_BUDGET=n     RTX_READY_THRESHOLD=n   RTX_DRAIN_NUM=n  RTX_ROUND_LIMIT=n
    RTX_GENERATIONS=0|1   never / always the generation pipeline (unset: the library chooses)
"""
import os

from raytracer_amd import api


def apply(env=os.environ):
    d = api.DEFAULTS
    for var, key in (("RTX_TILE_FEEDBACK", "tile_hint"), ("RTX_TILE_CULL", "tile_cull"), ("RTX_GENERATIONS", "generations")):   # (unset: by frame size)
        if var in env:
            d[key] = int(env[var]) != 0
    if "RTX_HELPERS" in env:                                  # 0 none, 1 by frame size, 2 early, 3 late
        d["helpers"] = {0: False, 1: True, 2: "early", 3: "late"}[int(env["RTX_HELPERS"])]
    for var, key in (("RTX_FRAME_WAVES", "frame_waves"), ("RTX_WALK_BUDGET", "walk_budget"),
                     ("RTX_READY_THRESHOLD", "ready_threshold"), ("RTX_DRAIN_NUM", "drain_eighths"),
                     ("RTX_ROUND_LIMIT", "round_limit"), ("RTX_PATH_LANES", "path_lanes"), ("RTX_ENDGAME", "endgame_eighths")):
        if var in env:
            d[key] = int(env[var])
    return d


apply()
