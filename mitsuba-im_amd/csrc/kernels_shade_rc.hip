// kernels_shade_rc.hip -- k_shade<RC = true, ENV = false>: path classes 0 and 2 (every BSDF); see shade.h
#include "shade.h"
extern "C" void mi_launch_shade_rc(const DScene &sc, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, size_t lds, int cls, hipStream_t st) {
    if (cls == 0) launchShadeVariant<true, false, 0>(sc, rc, q, buf, grid, lds, st); else launchShadeVariant<true, false, 2>(sc, rc, q, buf, grid, lds, st);
}
