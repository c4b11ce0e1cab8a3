// kernels_shade_d.hip -- k_shade<RC = false, ENV = false>: path classes 0 and 1 (diffuse-only code); see shade.h
#include "shade.h"
extern "C" void mi_launch_shade_d(const DScene &sc, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, size_t lds, int cls, hipStream_t st) {
    if (cls == 0) launchShadeVariant<false, false, 0>(sc, rc, q, buf, grid, lds, st); else launchShadeVariant<false, false, 1>(sc, rc, q, buf, grid, lds, st);
}
