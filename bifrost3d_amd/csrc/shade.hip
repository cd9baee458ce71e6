// shade.hip -- translation unit of the shade kernel (see shade_kernel.h for why it is separate).
#define HIPR_SHADE_TU 1
#ifndef HIPR_FAST_MATH
#define HIPR_FAST_MATH 1
#endif
#include "shade_kernel.h"
#include "launch.h"

namespace hipr {

template <int MODELS, bool AOV>
static void launch_models(const ShadeLaunch& a) {
    hipLaunchKernelGGL((k_shade<MODELS, AOV>), dim3(a.grid), dim3(SHADE_BLOCK), 0, a.stream, a.scene, a.camera, a.entry, a.in, a.hits, a.out, a.shadows, a.radiance,
                       a.in_count, a.out_counts, a.counters);
}

// The kernel is instantiated per set of shading models the uploaded scene uses (bit 0 Default, 1 Diffuse, 2 Transmissive).
void launch_shade(int shading_models, const ShadeLaunch& a) {
    if (a.entry != HIPR_ENTRY_PATH_TRACING) { launch_models<7, true>(a); return; }   // AOV entries: one generic instantiation
    switch (shading_models) {
    case 1: launch_models<1, false>(a); break;
    case 2: launch_models<2, false>(a); break;
    case 4: launch_models<4, false>(a); break;
    case 3: launch_models<3, false>(a); break;
    case 5: launch_models<5, false>(a); break;
    case 6: launch_models<6, false>(a); break;
    default: launch_models<7, false>(a); break;
    }
}

} // namespace hipr
