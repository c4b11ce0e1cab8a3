// kernels_shade_rcw_env.hip -- k_shade<RC = true, ENV = true, WRAP = true>, the variants for scenes with large emitter tables; the small-table half lives in
// kernels_shade_rcw_env_small.hip (the WRAP variants are the slowest kernels to compile: two translation units build side by side); see shade.h
#include "shade.h"
extern "C" void mi_launch_shade_rcw_env_small(const DScene &sc, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, size_t lds, hipStream_t st);
extern "C" void mi_launch_shade_rcw_env(const DScene &sc, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, size_t lds, hipStream_t st) {
    if (sc.small_tables != 0) mi_launch_shade_rcw_env_small(sc, rc, q, buf, grid, lds, st);
    else launchShadeVariantSM<true, true, true, false>(sc, rc, q, buf, grid, lds, st);
}
