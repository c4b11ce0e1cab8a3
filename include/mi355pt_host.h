/* include/mi355pt_host.h -- C shim over the C++ host mirror mi355::MIPathTracerHIP (mitsuba-im_amd/csrc/integrator_host.h), exported by
 * libmi355pt.so so that non-C++ hosts and the ctypes tests can drive the reference-shaped interface:
 * ResponsiveIntegrator::preprocess / render(..., Controls, threadIdx, threadCount) (reference include/mitsuba/render/integrator2.h:49-100),
 * Integrator::cancel (include/mitsuba/render/integrator.h:89-94), MonteCarloIntegrator's properties (src/librender/integrator.cpp:191-226). */
#ifndef MI355PT_HOST_H
#define MI355PT_HOST_H
#include "mi355pt.h"
#ifdef __cplusplus
extern "C" {
#endif
const char *mi_host_last_error(void);
/* returns NULL (message in mi_host_last_error) when rrDepth <= 0 or maxDepth is neither -1 nor > 0 -- the reference's Log(EError) texts */
void *mi_host_create(int maxDepth, int rrDepth, int strictNormals, int hideEmitters, int sampler, uint32_t spp, uint64_t seed, uint32_t device, uint32_t planes_per_batch);
/* the same with the build-specific `devices` property: the film rows are spread over these HIP devices (an entry may repeat); the scene given to
 * mi_host_preprocess must live on devices[0], the other devices receive replicas (mi_scene_clone) and their films are merged (mi_render_merge_film) */
void *mi_host_create_devices(int maxDepth, int rrDepth, int strictNormals, int hideEmitters, int sampler, uint32_t spp, uint64_t seed, const uint32_t *devices, uint32_t n_devices, uint32_t planes_per_batch);
/* everything at once: `integrator` = MI_INTEGRATOR_PATH / _VOLPATH_SIMPLE / _VOLPATH (the plugin's `integrator` property), preview_interval_ms < 0 = the default (100 ms) */
void *mi_host_create_ex(int maxDepth, int rrDepth, int strictNormals, int hideEmitters, int sampler, uint32_t spp, uint64_t seed, const uint32_t *devices, uint32_t n_devices, uint32_t planes_per_batch, int integrator, double preview_interval_ms);
void mi_host_destroy(void *integrator);
int mi_host_preprocess(void *integrator, mi_scene *scene);
/* Controls = {continu, abort, interrupt}: returns 0 done, -1 *abort set, -2 *continu cleared, otherwise progress()'s non-zero value;
 * targetRGBA: (H+2b)x(W+2b)x4 un-normalised sums; all work on threadIdx 0 */
int mi_host_render(void *integrator, float *targetRGBA, const int *continu, const int *abort_flag, int (*progress)(double spp, void *user), void *user, int threadIdx, int threadCount);
void mi_host_cancel(void *integrator);
const char *mi_host_statistics(void *integrator);
#ifdef __cplusplus
}
#endif
#endif
