#!/bin/bash
# development A/B on the GPU box over image sizes: tools/ab_sizes.sh "<ENV=val ...>" ... -- scene7 frame time per size and setting
for size in "640 480" "1280 720" "1920 1080" "2560 1440" "3840 2160"; do
  for setting in "$@"; do
    echo "$size $setting: $(env $setting timeout -k 5 120 python -u tools/gpu_perf.py scene7 $size 6 noref noserial 2>/dev/null | grep 'frame :' | cut -c1-50)"
  done
done
