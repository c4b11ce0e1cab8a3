// kernels_shade_d_env.hip -- k_shade<RC = false, ENV = true, WRAP = false>; see shade.h
#include "shade.h"
extern "C" void mi_launch_shade_d_env(const DScene &sc, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, size_t lds, hipStream_t st) {
    launchShadeVariant<false, true, false>(sc, rc, q, buf, grid, lds, st);
}
