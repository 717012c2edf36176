#!/bin/bash
# development A/B on the GPU box over library variants (tools/variant.sh), generation pipeline, scene7 4K:
#   tools/ab_genvar.sh "<name> ..."   -- one hinted frame at a time, one first frame at a time, three first frames in flight
for v in $1; do
  lib=$PWD/raytracer_amd/variants/$v.so
  for hint in 1 0; do
    echo "$v hint=$hint: $(RTX_GENERATIONS=1 RTX_TILE_FEEDBACK=$hint RTX_LIB_PATH=$lib timeout -k 5 120 python -u tools/gpu_perf.py scene7 3840 2160 6 noref noserial nolevels 2>/dev/null | grep 'frame :' | cut -c1-60,150-330)"
  done
  echo "$v in flight: $(RTX_GENERATIONS=1 RTX_TILE_FEEDBACK=0 RTX_LIB_PATH=$lib timeout -k 5 120 python -u tools/gpu_inflight.py 1 3 2 6 2>/dev/null | tail -1)"
done
