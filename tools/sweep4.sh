#!/bin/bash
tools/quick.sh 6 2>&1 | grep -E "nest-frame|ref-frame" | cut -c1-150
for b in 1 2 3 4 6 8; do
  echo "budget=$b: $(RTX_WALK_BUDGET=$b timeout -k 5 120 python -u tools/gpu_perf.py scene7 3840 2160 6 noref noserial 2>/dev/null | grep 'frame :' | cut -c22-60)"
done
