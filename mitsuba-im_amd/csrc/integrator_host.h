// integrator_host.h -- C++ host mirror of the reference's integrator interface for the one path this build accelerates.
//
// Mirrors, with the same names, argument meaning and error behaviour:
//   * MonteCarloIntegrator's properties (reference src/librender/integrator.cpp:191-226): maxDepth (-1 = infinite), rrDepth (5),
//     strictNormals, hideEmitters; the constructor raises std::runtime_error with the reference's messages (Log(EError) throws there too);
//   * ResponsiveIntegrator (include/mitsuba/render/integrator2.h:49-100): preprocess / allocate / render(..., Controls, threadIdx, threadCount)
//     / getLowerSampleBound / getRealtimeStatistics, `Controls {continu, abort, interrupt}` and `Interrupt::progress`, return codes
//     0 = all sample planes done, -1 = *abort set, -2 = *continu cleared, other = value returned by progress (integrator.cpp:349-401);
//   * Integrator::cancel (integrator.h:89-94): asynchronous, any thread.
// It sits directly on the C-ABI (include/mi355pt.h) and carries no reference types, so it builds without the reference; the adapter plugin
// (adapter/path_hip.cpp) wraps it in the real mitsuba::Integrator / ResponsiveIntegrator classes.
#pragma once
#include <atomic>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/mi355pt.h"

namespace mi355 {

class MIPathTracerHIP;

struct Properties {                 // the subset of mitsuba::Properties this integrator queries
    int maxDepth = -1, rrDepth = 5; bool strictNormals = false, hideEmitters = false;
    int sampler = MI_SAMPLER_SOBOL; uint32_t sampleCount = 4; uint64_t seed = 0;   // the scene's sampler (Sampler::getSampleCount, getProperties)
    uint32_t device = 0, planesPerBatch = 0;                                       // build-specific
    std::vector<uint32_t> devices;   // build-specific (SURVEY §8b `devices`): HIP devices to spread the film rows over; empty = {device}.  An entry may repeat (two
                                     // replicas on one GPU).  The scene handed to preprocess() lives on devices[0]; the others get clones (mi_scene_clone)
    int integrator = MI_INTEGRATOR_PATH;   // build-specific: MI_INTEGRATOR_VOLPATH_SIMPLE / MI_INTEGRATOR_VOLPATH = the loops of volpath_simple / volpath over the scene's participating media
    double previewIntervalMs = 100.0;   // build-specific: the responsive face copies the film into its target at most this often between submissions (always after the last one); 0 = after every submission
    bool opacity = true;   // RadianceQueryRecord::EOpacity: the responsive drivers always request it (integrator.cpp:474)
};

class Interrupt;
struct Controls { int volatile const *continu; int volatile const *abort; Interrupt *interrupt; };
class Interrupt {
public:
    // called at the start of every sample-plane batch (always on a new plane, integrator.cpp:376-378); spp = completed planes
    virtual int progress(MIPathTracerHIP *integrator, const float *targetRGBA, double spp, Controls controls, int threadIdx, int threadCount) = 0;
    virtual ~Interrupt() {}
};

class MIPathTracerHIP {
public:
    explicit MIPathTracerHIP(const Properties &props);
    ~MIPathTracerHIP();
    // scene must be committed (mi_scene_commit); the integrator borrows it for its lifetime
    bool preprocess(mi_scene *scene);
    bool allocate(int threadCount);
    // targetRGBA: (H+2b) x (W+2b) x 4 un-normalised sums (the responsive ImageBlock layout, src/im-mts/scene.cpp:317-321), written with
    // plain stores between progress() calls.  All work happens on threadIdx 0; other threads return 0 immediately.
    int render(float *targetRGBA, Controls controls, int threadIdx, int threadCount);
    void cancel();
    float getLowerSampleBound() const { return 1.0f; }
    const char *getRealtimeStatistics();
    const Properties &getProperties() const { return m_props; }
    mi_render *handle() const { return m_render; }
    std::string toString() const;
private:
    Properties m_props; mi_scene *m_scene = nullptr; mi_render *m_render = nullptr; std::string m_stats; int m_threads = 1;
    std::vector<mi_scene *> m_replicaScenes; std::vector<mi_render *> m_replicaRenders;   // devices[1..]: one scene clone + one render handle each, driven by one host thread each
    void releaseReplicas();
    struct Workers; std::unique_ptr<Workers> m_workers;   // one persistent host thread per replica (multi-device renders)
    std::atomic<int> m_cancel{0};   // set by cancel(), reset at the start of render(): a cancel between two batches or two mi_render_run calls is never lost
};

}  // namespace mi355
