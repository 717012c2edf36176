#!/bin/bash
# development sweep of the persistent kernel's scheduling knobs (GPU box)
for b in 1 2 4 8 16; do for t in 8 16 32 48; do
  echo "budget=$b thr=$t: $(RTX_WALK_BUDGET=$b RTX_READY_THRESHOLD=$t timeout -k 5 120 python -u tools/gpu_perf.py scene7 3840 2160 ${1:-6} noref noserial 2>/dev/null | grep 'frame :' | cut -c1-75)"
done; done
