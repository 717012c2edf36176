#!/bin/bash
# development A/B on the GPU box at the small configurations: tools/ab_small.sh "<ENV=val ...>" ... -- frame times of the
# default pipeline on scene4-specular, scene5, scene7 at 640x480 (checked against the oracle) per setting
for setting in "$@"; do
  echo "$setting: $(env $setting timeout -k 5 200 python -u tools/gpu_check.py scene4-specular scene5 scene6 scene7 2>&1 | grep -E 'nest-frame' | sed -E 's/.*ids_identical=(\w+).*frame ([0-9.]+) ms.*/\1 \2/' | tr '\n' ' ')"
done
