#!/bin/bash
# development sweep on the GPU box: a rank's share of the 4K frame, frames in flight x workgroups per CU (library's own pipeline choice)
for cfg in "2 3 2" "2 4 2" "2 4 1" "2 6 1" "4 3 2" "4 4 2" "4 6 1" "4 8 1" "8 6 1" "8 8 1" "8 4 1" "8 8 2" "8 6 2"; do
  set -- $cfg
  echo "$(RTX_TILE_FEEDBACK=0 GPU_MAX_HW_QUEUES=8 timeout -k 5 120 python -u tools/gpu_inflight.py $1 $2 $3 6 2>/dev/null | tail -1)"
done
