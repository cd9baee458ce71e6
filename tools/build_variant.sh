#!/bin/bash
# Builds an A/B variant of the device library next to the product one: bifrost3d_amd/csrc/libhiprenderer_<suffix>.so, with extra preprocessor /
# compiler flags for the two translation units that hold the path tracing kernels (tools/gpu_ab.sh picks variants up by suffix through HIPR_LIBRARY).
# usage: tools/build_variant.sh <suffix> "<extra flags>"       e.g. tools/build_variant.sh stack24 "-DHIPR_STACK_MID=24"
set -eu
suffix=$1; extra=${2:-}
cd "$(dirname "$0")/../bifrost3d_amd"
make -s csrc/libhiprenderer.so
tmp=$(mktemp -d)
HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function -Wno-pass-failed"
SHADEFLAGS="$HIPFLAGS -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -ffp-contract=fast ${SHADE_MATH--freciprocal-math -fapprox-func} -DHIPR_FAST_MATH=1"     # SHADE_MATH="" builds the shade kernel with round 3's division
hipcc $HIPFLAGS -DHIPR_VERIFY_MATH=1 $extra -c -o $tmp/hiprenderer.o csrc/hiprenderer.hip &
hipcc ${SHADE_ALL_FLAGS:-$SHADEFLAGS} $extra -c -o $tmp/shade.o csrc/shade.hip &      # SHADE_ALL_FLAGS: the whole flag set of the fast shade unit
hipcc $HIPFLAGS -DHIPR_SHADE_EXACT=1 ${EXACT_EXTRA--DHIPR_SHADE_WAVES=2} $extra -c -o $tmp/shade_exact.o csrc/shade.hip &      # EXACT_EXTRA: flags for the exact shade unit only (e.g. -DHIPR_SINCOS_KIND=2)
sort_object=
case "$extra" in *HIPR_RAY_SORT=1*) hipcc $HIPFLAGS $extra -Icsrc -c -o $tmp/ray_sort.o ../tools/experiments/ray_sort.hip; sort_object=$tmp/ray_sort.o;; esac
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o csrc/libhiprenderer_$suffix.so $tmp/hiprenderer.o $tmp/shade.o $tmp/shade_exact.o $sort_object csrc/camera_effects.o csrc/denoiser.o csrc/group.o -ldl -lpthread
rm -rf $tmp
echo built csrc/libhiprenderer_$suffix.so
