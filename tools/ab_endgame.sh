#!/bin/bash
# development A/B on the GPU box: scene7 4K, generation pipeline, start of the endgame swept (eighths of a pixel left per lane)
for e in -1 2 4 8 16 24 32; do
  for hint in 1 0; do
    echo "endgame=$e hint=$hint: $(RTX_GENERATIONS=1 RTX_ENDGAME=$e RTX_TILE_FEEDBACK=$hint timeout -k 5 120 python -u tools/gpu_perf.py scene7 3840 2160 ${1:-6} noref noserial nolevels 2>/dev/null | grep 'frame :' | cut -c1-60,150-270)"
  done
done
