#!/bin/bash
# development sweep on the GPU box: a rank's share of the 4K frame (world 2 / 4 / 8) with frames in flight, both pipelines
for gen in 0 1; do
  for cfg in "2 3 2" "4 3 2" "8 3 2" "8 6 1" "8 4 2"; do
    set -- $cfg
    echo "generations=$gen: $(RTX_GENERATIONS=$gen RTX_TILE_FEEDBACK=0 GPU_MAX_HW_QUEUES=8 timeout -k 5 120 python -u tools/gpu_inflight.py $1 $2 $3 6 2>/dev/null | tail -1)"
  done
done
