#!/bin/bash
# development A/B on the GPU box over library variants ("tree" = in-tree library): golden parity, the 640x480
# configurations, an eighth of the 4K frame, lone 4K frames without and with the hint, three 4K frames in flight.
#   tools/ab_lat.sh "<name> ..."
lib() { if [ "$1" = tree ]; then echo $PWD/raytracer_amd/librtx.so; else echo $PWD/raytracer_amd/variants/$1.so; fi; }
for v in $1; do
  export RTX_LIB_PATH=$(lib $v)
  echo "== $v: $(timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k 'golden or single_rays' 2>&1 | tail -1)"
  for rep in 1 2; do echo "$v small: $(timeout -k 5 200 python -u tools/ab_small_gen.py 2>/dev/null | cut -c1-34 | tr '\n' '|')"; done
  for rep in 1 2; do echo "$v eighth: $(RTX_TILE_FEEDBACK=0 timeout -k 5 100 python -u tools/gpu_ranks.py 8 2>/dev/null | tail -1 | cut -c1-60)"; done
  echo "$v lone: $(timeout -k 5 100 python -u tools/dbg_hint.py 2>&1 | tail -2 | cut -c1-90 | tr '\n' ' ')"
  a=""; for rep in 1 2 3; do a="$a $(RTX_TILE_FEEDBACK=0 timeout -k 5 60 python -u tools/gpu_inflight.py 1 3 2 6 2>/dev/null | tail -1 | sed 's/.*: //;s/ ms per frame//')"; done; echo "$v in flight 3x2:$a"
done
