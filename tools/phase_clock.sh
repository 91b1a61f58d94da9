#!/bin/bash
# Builds the -DTRT_PHASE_CLOCK diagnostic library (s_memtime at the phase boundaries of the streamed kernels) and prints the
# wave-time split per scene: tools/phase_clock.sh [cornell] [random_spheres] [sphere_grid100k]
set -e
mkdir -p build
( cd tiny-raytracer_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
    -DTRT_PHASE_CLOCK -shared -o ../../build/libtinyrt_clock.so capi.hip kernels.hip wavefront.hip streamed.hip scene_host.cpp )
TRT_LIB_PATH=$PWD/build/libtinyrt_clock.so python3 tools/phase_clock.py "$@"
