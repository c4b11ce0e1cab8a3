// integrator_host.cpp -- see integrator_host.h
#include "integrator_host.h"
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <mutex>
#include <thread>

namespace mi355 {

static void check(int rc, const char *what) {
    if (rc != MI_OK) throw std::runtime_error(std::string(what) + ": " + mi_last_error());
}

// One worker thread per replica, alive from preprocess() to the destructor: a submission hands every worker one job (its rows of the sample planes [a, b), or one
// film merge of the reduction tree) and waits for all of them -- no thread is created per chunk.
struct MIPathTracerHIP::Workers {
    struct Slot { std::function<void()> job; bool pending = false; };
    std::vector<std::thread> threads; std::vector<Slot> slots; std::mutex m; std::condition_variable cvJob, cvDone; int outstanding = 0; bool stop = false;
    explicit Workers(size_t n) : slots(n) { for (size_t k = 0; k < n; ++k) threads.emplace_back([this, k] { loop(k); }); }
    ~Workers() { { std::lock_guard<std::mutex> l(m); stop = true; } cvJob.notify_all(); for (std::thread &t : threads) t.join(); }
    void loop(size_t k) {
        for (;;) {
            std::function<void()> job;
            { std::unique_lock<std::mutex> l(m); cvJob.wait(l, [&] { return stop || slots[k].pending; }); if (stop) return; job = std::move(slots[k].job); slots[k].pending = false; }
            job();
            { std::lock_guard<std::mutex> l(m); --outstanding; } cvDone.notify_all();
        }
    }
    void post(size_t k, std::function<void()> job) { { std::lock_guard<std::mutex> l(m); slots[k].job = std::move(job); slots[k].pending = true; ++outstanding; } cvJob.notify_all(); }
    void wait() { std::unique_lock<std::mutex> l(m); cvDone.wait(l, [&] { return outstanding == 0; }); }
};

MIPathTracerHIP::MIPathTracerHIP(const Properties &props) : m_props(props) {
    if (m_props.rrDepth <= 0) throw std::runtime_error("'rrDepth' must be set to a value greater than zero!");                          // integrator.cpp:221-222
    if (m_props.maxDepth <= 0 && m_props.maxDepth != -1)
        throw std::runtime_error("'maxDepth' must be set to -1 (infinite) or a value greater than zero!");                            // integrator.cpp:224-225
}
MIPathTracerHIP::~MIPathTracerHIP() { releaseReplicas(); if (m_render) mi_render_destroy(m_render); }
void MIPathTracerHIP::releaseReplicas() {
    m_workers.reset();
    for (mi_render *r : m_replicaRenders) mi_render_destroy(r);
    for (mi_scene *s : m_replicaScenes) mi_scene_destroy(s);
    m_replicaRenders.clear(); m_replicaScenes.clear();
}

bool MIPathTracerHIP::preprocess(mi_scene *scene) {
    releaseReplicas();
    if (m_render) { mi_render_destroy(m_render); m_render = nullptr; }
    m_scene = scene;
    mi_render_params p{};
    p.max_depth = m_props.maxDepth; p.rr_depth = m_props.rrDepth; p.strict_normals = m_props.strictNormals; p.hide_emitters = m_props.hideEmitters;
    p.sampler = (uint32_t) m_props.sampler; p.spp = m_props.sampleCount; p.seed = m_props.seed; p.device = m_props.device; p.planes_per_batch = m_props.planesPerBatch; p.opacity = m_props.opacity ? 1 : 0;
    p.integrator = (uint32_t) m_props.integrator;
    check(mi_render_create(scene, &p, &m_render), "MIPathTracerHIP::preprocess");
    // `devices`: every further entry gets a replica of the scene and a render handle of its own; film rows are interleaved over the replicas
    for (size_t i = 1; i < m_props.devices.size(); ++i) {
        mi_scene *rs = nullptr; check(mi_scene_clone(scene, m_props.devices[i], &rs), "MIPathTracerHIP::preprocess"); m_replicaScenes.push_back(rs);
        mi_render *rr = nullptr; p.device = m_props.devices[i]; check(mi_render_create(rs, &p, &rr), "MIPathTracerHIP::preprocess"); m_replicaRenders.push_back(rr);
    }
    if (!m_replicaRenders.empty()) m_workers.reset(new Workers(1 + m_replicaRenders.size()));
    return true;
}
bool MIPathTracerHIP::allocate(int threadCount) { m_threads = threadCount; return m_render != nullptr; }

int MIPathTracerHIP::render(float *target, Controls controls, int threadIdx, int threadCount) {
    if (threadIdx != 0) return 0;
    if (!m_render) throw std::runtime_error("MIPathTracerHIP::render: preprocess() was not called");
    m_cancel.store(0);
    check(mi_render_clear(m_render), "MIPathTracerHIP::render");
    uint32_t h, w, c, b; check(mi_render_film_size(m_render, 1, &h, &w, &c, &b), "MIPathTracerHIP::render");
    const uint32_t spp = m_props.sampleCount;
    const uint32_t nRep = 1 + (uint32_t) m_replicaRenders.size();
    mi_tile tile{0, 0, w - 2 * b, h - 2 * b};
    // Classic face (no target, no controls: Scene::render has neither a preview nor an interrupt): the whole job is ONE submission, the library cuts it into its own
    // batches (64 M paths per pool, alternating between the two path pools / HIP streams inside mi_render_run).  Responsive face: progress() is called between
    // submissions, always on a new sample plane (integrator.cpp:376-378); a submission is two of the library's batches (both streams busy) -- or, with an explicit
    // planesPerBatch, two such batches.
    const bool interactive = target || controls.interrupt || controls.abort || controls.continu;
    const uint64_t rowsPerRep = (tile.y1 + nRep - 1) / nRep, pixPerRep = std::max<uint64_t>(1, (uint64_t) tile.x1 * rowsPerRep);
    const uint32_t autoPlanes = (uint32_t) std::max<uint64_t>(1, (uint64_t) (64u << 20) / pixPerRep);
    const uint32_t planes = m_props.planesPerBatch ? m_props.planesPerBatch : autoPlanes;
    const uint32_t chunk = interactive ? (planes >= spp ? std::max(planes, 1u) : 2 * planes) : std::max(spp, 1u);
    for (mi_render *rr : m_replicaRenders) check(mi_render_clear(rr), "MIPathTracerHIP::render");
    auto renderOf = [&](uint32_t k) { return k ? m_replicaRenders[k - 1] : m_render; };
    // one submission of sample planes [a, b): every replica traces its rows (k, k + nRep, ...) on its own worker thread
    auto submit = [&](uint32_t a, uint32_t b2) -> int {
        if (nRep == 1) return mi_render_run(m_render, tile, a, b2);
        std::vector<int> rcs(nRep, MI_OK); std::vector<std::string> errs(nRep);
        for (uint32_t k = 0; k < nRep; ++k)
            m_workers->post(k, [&, k] { mi_tile t{0, k, tile.x1, tile.y1};
                                        if (k < tile.y1) { rcs[k] = mi_render_run_rows(renderOf(k), t, nRep, a, b2); if (rcs[k] != MI_OK) errs[k] = mi_last_error(); } });
        m_workers->wait();
        for (uint32_t k = 0; k < nRep; ++k) if (rcs[k] != MI_OK) { if (rcs[k] == MI_CANCELLED) return MI_CANCELLED; throw std::runtime_error("MIPathTracerHIP::render: " + errs[k]); }
        return MI_OK;
    };
    // Every replica keeps accumulating into its OWN film over the whole render, so each pixel's samples are summed in sample order by one device, exactly as a
    // single device would.  Previews (the responsive face's target between submissions) add the replicas' films on the host without touching them; the final
    // film is produced once, at the end, by a reduction TREE over the replicas (renderproc.cpp:142-149 merges worker blocks by addition too): at level s replica k
    // (k a multiple of 2^(s+1)) takes in replica k + 2^s, the pairs of a level run side by side on the destination replicas' worker threads (peer copies over
    // xGMI between different devices) -- log2(N) steps instead of N - 1 on device 0.
    auto mergeAll = [&]() {
        for (uint32_t stride = 1; stride < nRep; stride *= 2) {
            std::vector<int> rcs(nRep, MI_OK); std::vector<std::string> errs(nRep);
            for (uint32_t k = 0; k + stride < nRep; k += 2 * stride)
                m_workers->post(k, [&, k, stride] { rcs[k] = mi_render_merge_film(renderOf(k), renderOf(k + stride)); if (rcs[k] != MI_OK) errs[k] = mi_last_error(); });
            m_workers->wait();
            for (uint32_t k = 0; k < nRep; ++k) if (rcs[k] != MI_OK) throw std::runtime_error("MIPathTracerHIP::render: " + errs[k]);
        }
    };
    std::vector<float> scratch;
    auto preview = [&]() {
        check(mi_render_read_film(m_render, 1, target), "MIPathTracerHIP::render");
        const size_t nf = (size_t) h * w * c; scratch.resize(nf);
        for (mi_render *rr : m_replicaRenders) { check(mi_render_read_film(rr, 1, scratch.data()), "MIPathTracerHIP::render"); for (size_t i = 0; i < nf; ++i) target[i] += scratch[i]; }
    };
    // the film crosses PCIe only when a preview is due: after the last submission, and in between at most every previewIntervalMs (the GUI repaints at its own
    // rate; the reference's per-block puts have no counterpart here, im_render.cpp:103-222 only reads the target between progress() calls)
    auto lastPreview = std::chrono::steady_clock::now(); bool previewed = false;
    for (uint32_t s = 0; s < spp; s += chunk) {
        if (m_cancel.load()) return -1;                                        // Integrator::cancel (any thread, any time)
        if (controls.abort && *controls.abort) return -1;
        if (controls.continu && !*controls.continu) return -2;
        if (controls.interrupt) { int rc = controls.interrupt->progress(this, target, (double) s, controls, threadIdx, threadCount); if (rc != 0) return rc; }
        if (m_cancel.load()) return -1;
        int rc = submit(s, std::min(spp, s + chunk));
        if (rc == MI_CANCELLED) return -1;
        check(rc, "MIPathTracerHIP::render");
        const bool last = s + chunk >= spp;
        if (last) mergeAll();
        if (target) {
            const auto now = std::chrono::steady_clock::now();
            const bool due = !previewed || std::chrono::duration<double, std::milli>(now - lastPreview).count() >= m_props.previewIntervalMs;
            if (last) check(mi_render_read_film(m_render, 1, target), "MIPathTracerHIP::render");
            else if (due) { if (nRep == 1) check(mi_render_read_film(m_render, 1, target), "MIPathTracerHIP::render"); else preview(); lastPreview = now; previewed = true; }
        }
    }
    return m_cancel.load() ? -1 : 0;
}
void MIPathTracerHIP::cancel() { m_cancel.store(1); if (m_render) mi_render_cancel(m_render); for (mi_render *rr : m_replicaRenders) mi_render_cancel(rr); }

const char *MIPathTracerHIP::getRealtimeStatistics() {
    if (!m_render) return nullptr;
    mi_stats st{}; if (mi_render_stats(m_render, &st) != MI_OK) return nullptr;
    char buf[256];
    snprintf(buf, sizeof(buf), "%.1f Msamples/s | %.2f rays/sample | %.2f shadow rays/sample | avg path length %.2f",
             st.render_ms > 0 ? st.samples / st.render_ms * 1e-3 : 0.0, st.samples ? (double) st.rays / st.samples : 0.0,
             st.samples ? (double) st.shadow_rays / st.samples : 0.0, st.samples ? (double) st.path_length_sum / st.samples : 0.0);
    m_stats = buf; return m_stats.c_str();
}
std::string MIPathTracerHIP::toString() const {
    char buf[256];
    snprintf(buf, sizeof(buf), "MIPathTracerHIP[\n  maxDepth = %d,\n  rrDepth = %d,\n  strictNormals = %d\n]", m_props.maxDepth, m_props.rrDepth, (int) m_props.strictNormals);
    return buf;
}

}  // namespace mi355

// C shim so the host mirror can be driven from the ctypes tests (no new functionality: thin calls into the class above)
extern "C" {
struct mi_host_integrator { mi355::MIPathTracerHIP *p; std::string err; };
static thread_local std::string g_hostErr;
const char *mi_host_last_error(void) { return g_hostErr.c_str(); }
void *mi_host_create_ex(int maxDepth, int rrDepth, int strictNormals, int hideEmitters, int sampler, uint32_t spp, uint64_t seed, const uint32_t *devices, uint32_t nDevices, uint32_t planes, int integrator, double previewIntervalMs) {
    try {
        mi355::Properties pr; pr.maxDepth = maxDepth; pr.rrDepth = rrDepth; pr.strictNormals = strictNormals != 0; pr.hideEmitters = hideEmitters != 0;
        pr.sampler = sampler; pr.sampleCount = spp; pr.seed = seed; pr.planesPerBatch = planes; pr.integrator = integrator; if (previewIntervalMs >= 0) pr.previewIntervalMs = previewIntervalMs;
        if (devices && nDevices) { pr.devices.assign(devices, devices + nDevices); pr.device = devices[0]; }
        return new mi355::MIPathTracerHIP(pr);
    } catch (const std::exception &e) { g_hostErr = e.what(); return nullptr; }
}
void *mi_host_create_devices(int maxDepth, int rrDepth, int strictNormals, int hideEmitters, int sampler, uint32_t spp, uint64_t seed, const uint32_t *devices, uint32_t nDevices, uint32_t planes) {
    try {
        mi355::Properties pr; pr.maxDepth = maxDepth; pr.rrDepth = rrDepth; pr.strictNormals = strictNormals != 0; pr.hideEmitters = hideEmitters != 0;
        pr.sampler = sampler; pr.sampleCount = spp; pr.seed = seed; pr.planesPerBatch = planes;
        if (devices && nDevices) { pr.devices.assign(devices, devices + nDevices); pr.device = devices[0]; }
        return new mi355::MIPathTracerHIP(pr);
    } catch (const std::exception &e) { g_hostErr = e.what(); return nullptr; }
}
void *mi_host_create(int maxDepth, int rrDepth, int strictNormals, int hideEmitters, int sampler, uint32_t spp, uint64_t seed, uint32_t device, uint32_t planes) {
    try {
        mi355::Properties pr; pr.maxDepth = maxDepth; pr.rrDepth = rrDepth; pr.strictNormals = strictNormals != 0; pr.hideEmitters = hideEmitters != 0;
        pr.sampler = sampler; pr.sampleCount = spp; pr.seed = seed; pr.device = device; pr.planesPerBatch = planes;
        return new mi355::MIPathTracerHIP(pr);
    } catch (const std::exception &e) { g_hostErr = e.what(); return nullptr; }
}
void mi_host_destroy(void *h) { delete (mi355::MIPathTracerHIP *) h; }
int mi_host_preprocess(void *h, mi_scene *scene) { try { return ((mi355::MIPathTracerHIP *) h)->preprocess(scene) ? 0 : 1; } catch (const std::exception &e) { g_hostErr = e.what(); return 2; } }
namespace { struct CbInterrupt : mi355::Interrupt { int (*cb)(double, void *); void *user; int progress(mi355::MIPathTracerHIP *, const float *, double spp, mi355::Controls, int, int) override { return cb ? cb(spp, user) : 0; } }; }
int mi_host_render(void *h, float *target, const int *continu, const int *abortFlag, int (*progress)(double, void *), void *user, int threadIdx, int threadCount) {
    try {
        CbInterrupt in; in.cb = progress; in.user = user;
        mi355::Controls c{continu, abortFlag, progress ? &in : nullptr};
        return ((mi355::MIPathTracerHIP *) h)->render(target, c, threadIdx, threadCount);
    } catch (const std::exception &e) { g_hostErr = e.what(); return 1000; }
}
void mi_host_cancel(void *h) { ((mi355::MIPathTracerHIP *) h)->cancel(); }
const char *mi_host_statistics(void *h) { return ((mi355::MIPathTracerHIP *) h)->getRealtimeStatistics(); }
}
