#!/bin/bash
# development A/B on the GPU box over library variants built with tools/variant.sh:
#   tools/ab_variants.sh "<name> ..." [sub]  -- scene7 4K frame time per variant, with and without the previous-frame hint
for v in $1; do
  for hint in 1 0; do
    echo "$v hint=$hint: $(RTX_TILE_FEEDBACK=$hint RTX_LIB_PATH=$PWD/raytracer_amd/variants/$v.so timeout -k 5 120 python -u tools/gpu_perf.py scene7 3840 2160 ${2:-6} noref noserial nolevels 2>/dev/null | grep 'frame :' | cut -c1-175)"
  done
done
