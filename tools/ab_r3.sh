#!/bin/bash
# development A/B on the GPU box over library variants (raytracer_amd/variants/<name>.so, or "tree" for the in-tree
# library): scene7 @ 4K, three first frames in flight (the benchmark's headline loop without its gather), twice per
# variant in alternation so that drift of the box shows; then a lone first frame in either pipeline.
#   tools/ab_r3.sh "<name> ..." [lone]
lib() { if [ "$1" = tree ]; then echo $PWD/raytracer_amd/librtx.so; else echo $PWD/raytracer_amd/variants/$1.so; fi; }
for rep in 1 2; do
  for v in $1; do
    echo "$v in flight 3x2: $(RTX_LIB_PATH=$(lib $v) RTX_TILE_FEEDBACK=0 timeout -k 5 120 python -u tools/gpu_inflight.py 1 3 2 6 2>/dev/null | tail -1 | sed 's/.*: //')"
  done
done
[ -z "$2" ] && exit 0
for v in $1; do
  for gen in 0 1; do
    echo "$v lone first frame generations=$gen: $(RTX_LIB_PATH=$(lib $v) RTX_GENERATIONS=$gen RTX_TILE_FEEDBACK=0 timeout -k 5 120 python -u tools/gpu_perf.py scene7 3840 2160 6 noref noserial nolevels 2>/dev/null | grep 'frame :' | cut -c1-60,150-330)"
  done
done
