#!/bin/bash
# development sweep of the persistent kernel's scheduling knobs (GPU box): tools/sweep.sh <sub> "<budgets>" "<thresholds>"
for b in ${2:-1 2 3 4}; do for t in ${3:-16 24 32 40 48}; do
  echo "budget=$b thr=$t: $(RTX_WALK_BUDGET=$b RTX_READY_THRESHOLD=$t timeout -k 5 120 python -u tools/gpu_perf.py scene7 3840 2160 ${1:-6} noref noserial 2>/dev/null | grep 'frame :' | cut -c1-75)"
done; done
