#!/bin/bash
# development: builds raytracer_amd/variants/<name>.so with extra compiler flags for the device layer, for A/B runs
# on the GPU box through RTX_LIB_PATH:   tools/variant.sh skip8 -DRT_EMPTY_RUN=8
set -e
cd "$(dirname "$0")/../raytracer_amd"
name=$1; shift
make -s build/scene_host.o build/image_io.o build/rtx_multi.o
mkdir -p variants build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function \
  -I../include -Icsrc "$@" -c csrc/rtx_device.hip -o build/var_$name.o
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o variants/$name.so build/var_$name.o build/rtx_multi.o build/scene_host.o build/image_io.o -lz -lpthread
echo "built variants/$name.so"
