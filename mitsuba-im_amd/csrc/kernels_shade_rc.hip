// kernels_shade_rc.hip -- k_shade<RC = true, ENV = false, WRAP = false>; see shade.h
#include "shade.h"
extern "C" void mi_launch_shade_rc(const DScene &sc, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, size_t lds, hipStream_t st) {
    launchShadeVariant<true, false, false>(sc, rc, q, buf, grid, lds, st);
}
