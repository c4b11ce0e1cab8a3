// kernels_shade_rcw_env_small.hip -- k_shade<RC = true, ENV = true, SMALL = true, WRAP = true>; called from kernels_shade_rcw_env.hip; see shade.h
#include "shade.h"
extern "C" void mi_launch_shade_rcw_env_small(const DScene &sc, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, size_t lds, hipStream_t st) {
    launchShadeVariantSM<true, true, true, true>(sc, rc, q, buf, grid, lds, st);
}
