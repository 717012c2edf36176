#!/bin/bash
# development A/B on the GPU box: scene7 4K with the default pipeline and with the generation pipeline
# (one frame at a time with / without the previous-frame hint; three first frames in flight)
for gen in 0 1; do
  for hint in 1 0; do
    echo "generations=$gen hint=$hint: $(RTX_GENERATIONS=$gen RTX_TILE_FEEDBACK=$hint timeout -k 5 120 python -u tools/gpu_perf.py scene7 3840 2160 ${1:-6} noref noserial nolevels 2>/dev/null | grep 'frame :' | cut -c1-60,150-330)"
  done
  echo "generations=$gen in flight: $(RTX_GENERATIONS=$gen RTX_TILE_FEEDBACK=0 timeout -k 5 120 python -u tools/gpu_inflight.py 1 3 2 ${1:-6} 2>/dev/null | tail -1)"
done
