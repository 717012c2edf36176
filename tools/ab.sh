#!/bin/bash
# development A/B of two builds of the library on the 4K frame (GPU box): ab.sh <libA> <libB> [sub]
for i in 1 2; do for lib in "$1" "$2"; do
  echo "$lib: $(RTX_LIB_PATH=$PWD/$lib timeout -k 5 120 python -u tools/gpu_perf.py scene7 3840 2160 ${3:-6} noref noserial 2>/dev/null | grep 'frame :' | cut -c22-60)"
done; done
