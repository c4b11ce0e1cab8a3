// kernels_shade_rcw_small.hip -- k_shade<RC = true, ENV = false, SMALL = true, WRAP = true>; called from kernels_shade_rcw.hip; see shade.h
#include "shade.h"
extern "C" void mi_launch_shade_rcw_small(const DScene &sc, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, size_t lds, hipStream_t st) {
    launchShadeVariantSM<true, false, true, true>(sc, rc, q, buf, grid, lds, st);
}
