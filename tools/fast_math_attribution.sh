#!/bin/bash
# Which of the shade unit's approximations moves the image how far from the verification build? (round 5)
# Builds libhiprenderer_<tag>.so variants whose FAST shade unit has one approximation made exact (P_*) or whose EXACT shade unit has one approximation made fast (V_*);
# only that unit is recompiled, the other one and the rest of the library are the product's. tools/fast_math_attribution.py then renders with each on one box
# (P_* in the fast arithmetic mode, V_* in the exact one). Round 6: the "verification build" of round 5 is the exact mode of the one library.
set -eu
cd "$(dirname "$0")/../bifrost3d_amd"
make -s csrc/libhiprenderer.so
H="-O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -fPIC -Wall -Wno-unused-function -Wno-pass-failed"
FAST_DIV="-fno-hip-fp32-correctly-rounded-divide-sqrt"; RCP="-freciprocal-math -fapprox-func"; FTZ="-fgpu-flush-denormals-to-zero"
declare -A V
V[P_sincos_exact]="$H $FAST_DIV $FTZ -ffp-contract=fast $RCP -DHIPR_FAST_MATH=1 -DHIPR_SINCOS_KIND=3"
V[P_pow_exact]="$H $FAST_DIV $FTZ -ffp-contract=fast $RCP -DHIPR_FAST_MATH=1 -DHIPR_POW_KIND=3"
V[P_no_reciprocal_math]="$H $FAST_DIV $FTZ -ffp-contract=fast -DHIPR_FAST_MATH=1 -DHIPR_RECIPROCAL_DIVISION=0"
V[P_ieee_div_sqrt]="$H $FTZ -ffp-contract=fast -DHIPR_FAST_MATH=1 -DHIPR_RECIPROCAL_DIVISION=0"
V[P_no_contraction]="$H $FAST_DIV $FTZ -ffp-contract=off $RCP -DHIPR_FAST_MATH=1"
V[P_denormals_kept]="$H $FAST_DIV -ffp-contract=fast $RCP -DHIPR_FAST_MATH=1"
V[V_sincos_fast]="$H -ffp-contract=off -DHIPR_SHADE_EXACT=1 -DHIPR_SHADE_WAVES=2 -DHIPR_SINCOS_KIND=1"
V[V_pow_fast]="$H -ffp-contract=off -DHIPR_SHADE_EXACT=1 -DHIPR_SHADE_WAVES=2 -DHIPR_POW_KIND=1"
V[V_reciprocal_div_sqrt_fast]="$H $FAST_DIV $RCP -ffp-contract=off -DHIPR_SHADE_EXACT=1 -DHIPR_SHADE_WAVES=2 -DHIPR_RECIPROCAL_DIVISION=1"
V[V_approx_div_sqrt_only]="$H $FAST_DIV -ffp-contract=off -DHIPR_SHADE_EXACT=1 -DHIPR_SHADE_WAVES=2"
V[V_contraction]="$H -ffp-contract=fast -DHIPR_SHADE_EXACT=1 -DHIPR_SHADE_WAVES=2"
V[V_flush_denormals]="$H $FTZ -ffp-contract=off -DHIPR_SHADE_EXACT=1 -DHIPR_SHADE_WAVES=2"
tmp=$(mktemp -d)
running=0
for tag in "${!V[@]}"; do
    hipcc ${V[$tag]} -c -o $tmp/$tag.o csrc/shade.hip &
    running=$((running + 1))
    if [ $running -ge 6 ]; then wait; running=0; fi
done
wait
for tag in "${!V[@]}"; do
    fast=$tmp/$tag.o; exact=csrc/shade_exact.o; case $tag in V_*) fast=csrc/shade.o; exact=$tmp/$tag.o;; esac
    hipcc --offload-arch=gfx950 -shared -fPIC -o csrc/libhiprenderer_$tag.so csrc/hiprenderer.o $fast $exact csrc/camera_effects.o csrc/denoiser.o csrc/group.o -ldl -lpthread
    echo built csrc/libhiprenderer_$tag.so
done
rm -rf $tmp
