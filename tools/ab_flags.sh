#!/bin/bash
# development A/B on the GPU box over library variants built with other compiler options (tools/variant.sh): scene7 4K,
# three first frames in flight (generation pipeline) and one first frame at a time (per-lane paths); every variant's frame
# is compared with the first variant's bit for bit by tools/gpu_perf.py's "same" columns only within a run, so run the
# parity tests on whatever is kept
for v in $1; do
  lib=$PWD/raytracer_amd/variants/$v.so
  a=$(RTX_LIB_PATH=$lib RTX_TILE_FEEDBACK=0 timeout -k 5 100 python -u tools/gpu_inflight.py 1 3 2 6 2>/dev/null | tail -1 | sed "s/.*sub 6: //")
  b=$(RTX_LIB_PATH=$lib RTX_TILE_FEEDBACK=0 timeout -k 5 100 python -u tools/gpu_inflight.py 1 1 4 6 2>/dev/null | tail -1 | sed "s/.*sub 6: //")
  c=$(RTX_LIB_PATH=$lib RTX_TILE_FEEDBACK=0 GPU_MAX_HW_QUEUES=8 timeout -k 5 100 python -u tools/gpu_inflight.py 8 6 1 6 2>/dev/null | tail -1 | sed "s/.*sub 6: //")
  echo "$v: three in flight $a; one at a time (4 workgroups per CU given) $b; eighth, six in flight $c"
done
