#!/bin/bash
# development A/B on the GPU box over library variants ("tree" = in-tree library): bit-exactness of both persistent
# pipelines against each other on every scene, three 4K first frames in flight (twice), a lone 4K first frame in either
# pipeline, the four 640x480 configurations and an eighth of the 4K frame as lone frames.
#   tools/ab_all3.sh "<name> ..."
lib() { if [ "$1" = tree ]; then echo $PWD/raytracer_amd/librtx.so; else echo $PWD/raytracer_amd/variants/$1.so; fi; }
for v in $1; do
  export RTX_LIB_PATH=$(lib $v)
  echo "== $v: $(timeout -k 10 300 python -u tools/gpu_gen_check.py 2>&1 | tail -1)"
  for rep in 1 2; do echo "$v in flight 3x2: $(RTX_TILE_FEEDBACK=0 timeout -k 5 60 python -u tools/gpu_inflight.py 1 3 2 6 2>/dev/null | tail -1 | sed 's/.*: //')"; done
  for gen in 0 1; do echo "$v lone 4K generations=$gen: $(RTX_GENERATIONS=$gen RTX_TILE_FEEDBACK=0 timeout -k 5 60 python -u tools/gpu_perf.py scene7 3840 2160 6 noref noserial nolevels 2>/dev/null | grep 'frame :' | cut -c20-60)"; done
  echo "$v small: $(timeout -k 5 200 python -u tools/ab_small_gen.py 2>/dev/null | cut -c1-40 | tr '\n' '|')"
  echo "$v eighth: $(RTX_TILE_FEEDBACK=0 timeout -k 5 100 python -u tools/gpu_ranks.py 8 2>/dev/null | tail -1 | cut -c1-60)"
done
