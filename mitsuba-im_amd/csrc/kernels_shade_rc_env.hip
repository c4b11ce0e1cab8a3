// kernels_shade_rc_env.hip -- k_shade<RC = true, ENV = true, WRAP = false>; see shade.h
#include "shade.h"
extern "C" void mi_launch_shade_rc_env(const DScene &sc, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, size_t lds, hipStream_t st) {
    launchShadeVariant<true, true, false>(sc, rc, q, buf, grid, lds, st);
}
