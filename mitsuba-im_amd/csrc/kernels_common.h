// kernels_common.h -- shared by the kernel translation units of libmi355pt.so (kernels_trace.hip, kernels_shade.hip, kernels_misc.hip).
//
// gfx950 (MI355X) wavefront path tracer.  Replaces the per-sample loop SamplingIntegrator::renderBlock -> MIPathTracer::Li -> Scene::rayIntersect /
// BSDF / emitter / sampler plugins (reference src/librender/integrator.cpp:141-189, src/integrators/path/path.cpp:119-294) by stages over SoA queues in HBM:
//     generate -> [ extend (closest hit) -> shade (MIS bookkeeping, RR, NEE sample, BSDF sample) -> shadow (any hit) ] x depth -> film
// Work ownership: the path pool of a batch is cut into `n_seg` contiguous SEGMENTS; a workgroup owns whole segments (segments b, b + gridDim.x, ...) in the
// traversal stages, a WAVE owns them in the shading stage, so stream compaction never leaves the wave: wave64 ballots, no global atomics, fully coalesced
// queue reads / writes.  Compiled with -ffp-contract=off and IEEE divide / sqrt: radiance is bit-identical to the strict-IEEE oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "pt_device.h"
#include "queues.h"

#define WG 256
#define STACK_DEPTH 32   // >= BVH depth (scene_build.cpp caps it; mi_scene_commit refuses deeper trees)

// Launch with `lds` bytes of dynamic LDS.  gfx950 has 160 KB per CU; beyond the 64 KB a launch may request by default the kernel is told so first
// (deep maxDepth x many Sobol index bits: the lookup tables alone can pass 64 KB).
template <typename K, typename... A>
static inline void launchWithLds(K kernel, uint32_t grid, size_t lds, hipStream_t st, A... args) {
    if (lds > 64 * 1024) (void) hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(WG), lds, st, args...);
}
