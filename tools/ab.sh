#!/bin/bash
# development A/B on the GPU box: tools/ab.sh "<ENV=val ...>" ["<ENV=val ...>" ...] -- one timing line per setting
for setting in "$@"; do
  echo "$setting: $(env $setting timeout -k 5 120 python -u tools/gpu_perf.py scene7 3840 2160 6 noref noserial 2>/dev/null | grep 'frame :' | cut -c1-75)"
done
