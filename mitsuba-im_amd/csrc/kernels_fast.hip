// kernels_fast.hip -- the fast-arithmetic translation unit: the same kernels as kernels.hip, compiled with fused multiply-adds and
// approximate division / square root (Makefile: FASTFLAGS), selected by mi_render_params.fast_math.
#define MI_FAST_MATH 1
#include "kernels.hip"
