#!/bin/bash
for sub in 5 6 7 8; do for b in 1 2 3; do
  echo "sub=$sub budget=$b: $(RTX_WALK_BUDGET=$b timeout -k 5 120 python -u tools/gpu_perf.py scene7 3840 2160 $sub noref noserial 2>/dev/null | grep 'frame :' | cut -c22-60)"
done; done
for bpc in 3 4 6 8; do for t in 24 32 40; do
  echo "blocks/cu=$bpc thr=$t: $(RTX_BLOCKS_PER_CU=$bpc RTX_READY_THRESHOLD=$t timeout -k 5 120 python -u tools/gpu_perf.py scene7 3840 2160 6 noref noserial 2>/dev/null | grep 'frame :' | cut -c22-60)"
done; done
