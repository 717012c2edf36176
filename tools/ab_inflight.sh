#!/bin/bash
# development sweep on the GPU box: frames in flight x workgroups per CU, default and generation pipeline (scene7 4K, first frames)
for gen in 0 1; do
  for cfg in "1 4" "2 2" "2 4" "3 2" "3 4" "4 1" "4 2" "6 1"; do
    set -- $cfg
    echo "generations=$gen: $(RTX_GENERATIONS=$gen RTX_TILE_FEEDBACK=0 timeout -k 5 120 python -u tools/gpu_inflight.py 1 $1 $2 6 2>/dev/null | tail -1)"
  done
done
