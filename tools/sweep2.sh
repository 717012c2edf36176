#!/bin/bash
# development: compare register-allocation variants of the persistent kernel (GPU box)
python -u tools/gpu_check.py scene4-specular scene5 scene7 2>&1 | grep -E "^scene|frame " | cut -c1-150
for lib in raytracer_amd/librtx.so tmp_libs/librtx_w3.so tmp_libs/librtx_w4.so; do for b in 2 4; do for t in 32; do
  echo "$lib budget=$b thr=$t: $(RTX_LIB_PATH=$PWD/$lib RTX_WALK_BUDGET=$b RTX_READY_THRESHOLD=$t timeout -k 5 120 python -u tools/gpu_perf.py scene7 3840 2160 ${1:-6} noref noserial 2>/dev/null | grep 'frame :' | cut -c1-75)"
done; done; done
