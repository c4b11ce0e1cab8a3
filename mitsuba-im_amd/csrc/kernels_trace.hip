// kernels_trace.hip -- extend (closest hit) and shadow (any hit) stages, unit-level intersection entry point (see kernels_common.h)
#include "kernels_common.h"
#include <algorithm>
#include "trace.h"
#include "trace_fused.h"

// ---------------------------------------------------------------------------------------------- extend
// Scene::rayIntersect -> ShapeKDTree::rayIntersect (src/librender/skdtree.cpp:112-142): closest hit (t, u, v, prim)
template <int STACK, int AN, bool WIDE>   // STACK = 0: packet mode; else LDS stack entries per lane (>= the tree's stack need); AN: bit 0 analytic shapes, bit 1 instances; WIDE: 4-wide quantised nodes
__global__ __launch_bounds__(WG) void k_extend(DScene sc, Queues q, int buf) {
    __shared__ int s_stk[(STACK > 0 ? STACK : 1) * WG];
    __shared__ f4 s_exact[STACK > 0 ? 1 : MI_PACKET_MAX * 3];
    const uint32_t tid = threadIdx.x;
    unsigned long long rays = 0;
    if (STACK == 0) packetStage(sc, s_exact);
    for (uint32_t seg = blockIdx.x; seg < q.n_seg; seg += gridDim.x) {
    const uint32_t n = q.count[buf][seg];
    const uint64_t segBase = (uint64_t) seg * q.cap;
    rays += n;
    for (uint32_t i = tid; i < n; i += WG) {
        float4 ro = q.rayO[buf][segBase + i], rd = q.rayD[buf][segBase + i];
        v3 o = V(ro.x, ro.y, ro.z), d = V(rd.x, rd.y, rd.z);
        float mint, maxt, t = 0, u = 0, v = 0; uint32_t prim = 0xFFFFFFFFu; bool hit = false; int inst = -1;
        if (clipInterval(sc, o, d, ro.w, rd.w, false, mint, maxt)) {
            if (STACK == 0) hit = packetIntersect<false, AN>(sc, (AS<true>::p4) s_exact, o, d, mint, maxt, t, prim, u, v);
            else hit = traverse<false, AN, WIDE>(sc, o, d, mint, maxt, s_stk + tid, t, prim, u, v, inst);
        }
        q.hit[segBase + i] = make_float4(t, u, v, __uint_as_float(hit ? prim : 0xFFFFFFFFu));
        if ((AN & 2) && q.hitInst) q.hitInst[segBase + i] = inst;
    }
    }
    if (tid == 0 && rays) atomicAdd(&q.counters[0], rays);
}

// ---------------------------------------------------------------------------------------------- shadow
// Visibility test of Scene::sampleEmitterDirect (src/librender/scene.cpp:871-875 -> skdtree.cpp:207-226, any hit) and the
// deferred `Li += throughput * value * bsdfVal * weight` (path.cpp:196)
template <int STACK, int AN, bool WIDE>
__global__ __launch_bounds__(WG) void k_shadow(DScene sc, Queues q) {
    __shared__ int s_stk[(STACK > 0 ? STACK : 1) * WG];
    __shared__ f4 s_exact[STACK > 0 ? 1 : MI_PACKET_MAX * 3];
    const uint32_t tid = threadIdx.x;
    if (STACK == 0) packetStage(sc, s_exact);
    for (uint32_t seg = blockIdx.x; seg < q.n_seg; seg += gridDim.x) {
    const uint32_t n = q.shCount[seg];
    const uint64_t segBase = (uint64_t) seg * q.cap;
    for (uint32_t i = tid; i < n; i += WG) {
        float4 so = q.shO[segBase + i], sd = q.shD[segBase + i];
        v3 o = V(so.x, so.y, so.z), d = V(sd.x, sd.y, sd.z);
        float mint, maxt, t, u, v; uint32_t prim; bool occluded = false; int inst;
        if (clipInterval(sc, o, d, MI_EPSILON, so.w, true, mint, maxt)) {
            if (STACK == 0) occluded = packetIntersect<true, AN>(sc, (AS<true>::p4) s_exact, o, d, mint, maxt, t, prim, u, v);
            else occluded = traverse<true, AN, WIDE>(sc, o, d, mint, maxt, s_stk + tid, t, prim, u, v, inst);
        }
        if (!occluded) {
            float4 c = q.shC[segBase + i]; const uint32_t pid = __float_as_uint(sd.w);
            float4 a = q.acc[pid]; a.x += c.x; a.y += c.y; a.z += c.z; q.acc[pid] = a;
        }
    }
    }
}



// ---------------------------------------------------------------------------------------------- fused walk (trace_fused.h): triangle-only trees
template <bool WIDE>
__global__ __launch_bounds__(WG) void k_extend_f(DScene sc, Queues q, int buf, uint32_t *ticket, uint32_t thr) {
    __shared__ int s_lds[FZ_LDS_WORDS];
    fusedStage<false, WIDE>(sc, q, buf, ticket, thr, s_lds);
}
template <bool WIDE>
__global__ __launch_bounds__(WG) void k_shadow_f(DScene sc, Queues q, uint32_t *ticket, uint32_t thr) {
    __shared__ int s_lds[FZ_LDS_WORDS];
    fusedStage<true, WIDE>(sc, q, 0, ticket, thr, s_lds);
}

// ---------------------------------------------------------------------------------------------- unit-level entry point (parity tests)
__global__ __launch_bounds__(WG) void k_debug_intersect(DScene sc, const float *rays, uint64_t n, int anyHit, float *out, int *outInst) {
    __shared__ int s_stk[STACK_DEPTH * WG];
    __shared__ f4 s_exact[MI_PACKET_MAX * 3];
    if (sc.packet_n) packetStage(sc, s_exact);
    const uint64_t i = (uint64_t) blockIdx.x * WG + threadIdx.x;
    if (i >= n) return;
    const float *r = rays + i * 8;
    v3 o = V(r[0], r[1], r[2]), d = V(r[4], r[5], r[6]);
    float mint, maxt, t = 0, u = 0, v = 0; uint32_t prim = 0xFFFFFFFFu; bool hit = false; int inst = -1;
    if (clipInterval(sc, o, d, r[3], r[7], anyHit != 0, mint, maxt)) {
        if (sc.packet_n) { if (anyHit) hit = packetIntersect<true, 1>(sc, (AS<true>::p4) s_exact, o, d, mint, maxt, t, prim, u, v); else hit = packetIntersect<false, 1>(sc, (AS<true>::p4) s_exact, o, d, mint, maxt, t, prim, u, v); }
        else if (sc.bvh_wide) { if (anyHit) hit = traverse<true, 3, true>(sc, o, d, mint, maxt, s_stk + threadIdx.x, t, prim, u, v, inst); else hit = traverse<false, 3, true>(sc, o, d, mint, maxt, s_stk + threadIdx.x, t, prim, u, v, inst); }
        else if (anyHit) hit = traverse<true, 3, false>(sc, o, d, mint, maxt, s_stk + threadIdx.x, t, prim, u, v, inst);
        else hit = traverse<false, 3, false>(sc, o, d, mint, maxt, s_stk + threadIdx.x, t, prim, u, v, inst);
    }
    if (outInst) outInst[i] = hit ? inst : -1;
    out[i * 4] = t; out[i * 4 + 1] = u; out[i * 4 + 2] = v; out[i * 4 + 3] = hit ? (anyHit ? 1.0f : (float) prim) : -1.0f;
}

// ---------------------------------------------------------------------------------------------- Scene::rayIntersect, full records
// bool Scene::rayIntersect(const Ray &, Intersection &) (include/mitsuba/render/scene.h:187-243 -> ShapeKDTree::rayIntersect + fillIntersectionRecord,
// src/librender/skdtree.cpp:112-142, include/mitsuba/render/skdtree.h:343-428) for a batch of rays: the record of mi_intersection (include/mi355pt.h)
__global__ __launch_bounds__(WG) void k_ray_intersect(DScene sc, const float *rays, uint64_t n, mi_intersection *out) {
    __shared__ int s_stk[STACK_DEPTH * WG];
    __shared__ f4 s_exact[MI_PACKET_MAX * 3];
    if (sc.packet_n) packetStage(sc, s_exact);
    const uint64_t i = (uint64_t) blockIdx.x * WG + threadIdx.x;
    if (i >= n) return;
    const float *r = rays + i * 8;
    v3 o = V(r[0], r[1], r[2]), d = V(r[4], r[5], r[6]);
    float mint, maxt, t = 0, u = 0, v = 0; uint32_t prim = 0xFFFFFFFFu; bool hit = false; int inst = -1;
    if (clipInterval(sc, o, d, r[3], r[7], false, mint, maxt)) {
        if (sc.packet_n) hit = packetIntersect<false, 1>(sc, (AS<true>::p4) s_exact, o, d, mint, maxt, t, prim, u, v);
        else if (sc.bvh_wide) hit = traverse<false, 3, true>(sc, o, d, mint, maxt, s_stk + threadIdx.x, t, prim, u, v, inst);
        else hit = traverse<false, 3, false>(sc, o, d, mint, maxt, s_stk + threadIdx.x, t, prim, u, v, inst);
    }
    mi_intersection rec; memset(&rec, 0, sizeof(rec)); rec.valid = hit ? 1u : 0u; rec.prim = 0xFFFFFFFFu; rec.instance = -1; rec.material = -1; rec.emitter = -1;
    if (hit) {
        Tabs<false> tb; tb.shade4 = (AS<false>::p4) sc.shade; tb.materials4 = (AS<false>::p4) sc.materials; tb.emitters4 = (AS<false>::p4) sc.emitters; tb.emitter_cdf = sc.emitter_cdf; tb.area_cdf = sc.area_cdf;
        Hit h;
        if (inst >= 0) fillHitInstanced(sc, tb, sc.instances[inst], o, d, t, prim, u, v, h);
        else if (prim >= sc.n_tris) fillHitAnalytic(sc.analytic[prim - sc.n_tris], o, d, t, u, v, h);
        else fillHit<false, true>(sc, tb, d, t, prim, u, v, h);
        if (inst < 0 && prim >= sc.n_tris) { v3 du, dv; analyticUV(sc.analytic[prim - sc.n_tris], u, v, o + d * t, h.uvx, h.uvy, du, dv); }
        rec.t = t; rec.p[0] = h.p.x; rec.p[1] = h.p.y; rec.p[2] = h.p.z; rec.ng[0] = h.ng.x; rec.ng[1] = h.ng.y; rec.ng[2] = h.ng.z;
        rec.ns[0] = h.ns.x; rec.ns[1] = h.ns.y; rec.ns[2] = h.ns.z; rec.s[0] = h.s.x; rec.s[1] = h.s.y; rec.s[2] = h.s.z; rec.tt[0] = h.t.x; rec.tt[1] = h.t.y; rec.tt[2] = h.t.z;
        rec.uv[0] = h.uvx; rec.uv[1] = h.uvy; rec.wi[0] = h.wi.x; rec.wi[1] = h.wi.y; rec.wi[2] = h.wi.z; rec.bary[0] = u; rec.bary[1] = v;
        rec.prim = prim; rec.instance = inst; rec.material = h.material; rec.emitter = h.emitter;
    }
    out[i] = rec;
}

// ---------------------------------------------------------------------------------------------- launch wrappers (used by api.cpp)
extern "C" {
static const bool kForceStack24 = getenv("MI355PT_STACK24") != nullptr;      // A/B switch, read once
#define MI_BY_STACK_W(KERNEL, AN, W, ...) do { \
    if (sc.bvh_depth <= 8) hipLaunchKernelGGL((KERNEL<8, AN, W>), dim3(grid), dim3(WG), 0, st, __VA_ARGS__); \
    else if (sc.bvh_depth <= 12) hipLaunchKernelGGL((KERNEL<12, AN, W>), dim3(grid), dim3(WG), 0, st, __VA_ARGS__); \
    else if (sc.bvh_depth <= 16) hipLaunchKernelGGL((KERNEL<16, AN, W>), dim3(grid), dim3(WG), 0, st, __VA_ARGS__); \
    else if (sc.bvh_depth <= 20 && !kForceStack24) hipLaunchKernelGGL((KERNEL<20, AN, W>), dim3(grid), dim3(WG), 0, st, __VA_ARGS__); \
    else if (sc.bvh_depth <= 24) hipLaunchKernelGGL((KERNEL<24, AN, W>), dim3(grid), dim3(WG), 0, st, __VA_ARGS__); \
    else if (sc.bvh_depth <= 28) hipLaunchKernelGGL((KERNEL<28, AN, W>), dim3(grid), dim3(WG), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<STACK_DEPTH, AN, W>), dim3(grid), dim3(WG), 0, st, __VA_ARGS__); } while (0)
#define MI_BY_STACK(KERNEL, AN, ...) do { \
    if (sc.packet_n) hipLaunchKernelGGL((KERNEL<0, (AN) & 1, false>), dim3(grid), dim3(WG), 0, st, __VA_ARGS__); \
    else if (sc.bvh_wide) MI_BY_STACK_W(KERNEL, AN, true, __VA_ARGS__); \
    else MI_BY_STACK_W(KERNEL, AN, false, __VA_ARGS__); } while (0)
void mi_launch_extend(const DScene &sc, const Queues &q, int buf, uint32_t grid, hipStream_t st) {
    const int mode = (sc.n_analytic ? 1 : 0) | (sc.n_instances ? 2 : 0);
    if (mode == 3) MI_BY_STACK(k_extend, 3, sc, q, buf); else if (mode == 2) MI_BY_STACK(k_extend, 2, sc, q, buf); else if (mode == 1) MI_BY_STACK(k_extend, 1, sc, q, buf); else MI_BY_STACK(k_extend, 0, sc, q, buf);
}
void mi_launch_shadow(const DScene &sc, const Queues &q, uint32_t grid, hipStream_t st) {
    const int mode = (sc.n_analytic ? 1 : 0) | (sc.n_instances ? 2 : 0);
    if (mode == 3) MI_BY_STACK(k_shadow, 3, sc, q); else if (mode == 2) MI_BY_STACK(k_shadow, 2, sc, q); else if (mode == 1) MI_BY_STACK(k_shadow, 1, sc, q); else MI_BY_STACK(k_shadow, 0, sc, q);
}
#undef MI_BY_STACK
#undef MI_BY_STACK_W
// The fused walk serves triangle-only trees (no analytic shapes, no instances, not the packet); MI355PT_FUSED=0 restores the while-while kernels for A/B runs.
// `ticket`: a zeroed word of Queues::ticket (one per traversal launch of a batch); grid = persistent workgroups (waves fetch segments through the ticket).
static const int kFused = [] { const char *e = getenv("MI355PT_FUSED"); return e && e[0] ? atoi(e) : 1; }();
static const uint32_t kFusedThr = [] { const char *e = getenv("MI355PT_FUSED_THR"); const int v = e && e[0] ? atoi(e) : 48; return (uint32_t) (v < 1 ? 1 : (v > 64 ? 64 : v)); }();
static const uint32_t kFusedGrid = [] { const char *e = getenv("MI355PT_FUSED_GRID"); const int v = e && e[0] ? atoi(e) : 1792; return (uint32_t) (v < 1 ? 1 : (v > 16384 ? 16384 : v)); }();
// wide (4-way) trees only: on the small binary trees the while-while kernels win (Veach-MIS 1080p: 1931 vs 1454 Msamples/s), MI355PT_FUSED=2 forces it there too
bool mi_fused_walk(const DScene &sc) { return kFused && (sc.bvh_wide || kFused == 2) && !sc.packet_n && !sc.n_analytic && !sc.n_instances && sc.geo_bytes < 0xFFFFFF00ull; }
uint32_t mi_fused_grid(void) { return kFusedGrid; }
void mi_launch_extend_fused(const DScene &sc, const Queues &q, int buf, uint32_t *ticket, hipStream_t st) {
    const uint32_t g = kFusedGrid;
    if (sc.bvh_wide) hipLaunchKernelGGL((k_extend_f<true>), dim3(g), dim3(WG), 0, st, sc, q, buf, ticket, kFusedThr); else hipLaunchKernelGGL((k_extend_f<false>), dim3(g), dim3(WG), 0, st, sc, q, buf, ticket, kFusedThr);
}
void mi_launch_shadow_fused(const DScene &sc, const Queues &q, uint32_t *ticket, hipStream_t st) {
    const uint32_t g = kFusedGrid;
    if (sc.bvh_wide) hipLaunchKernelGGL((k_shadow_f<true>), dim3(g), dim3(WG), 0, st, sc, q, ticket, kFusedThr); else hipLaunchKernelGGL((k_shadow_f<false>), dim3(g), dim3(WG), 0, st, sc, q, ticket, kFusedThr);
}
void mi_launch_ray_intersect(const DScene &sc, const float *rays, uint64_t n, mi_intersection *out, hipStream_t st) { hipLaunchKernelGGL(k_ray_intersect, dim3((unsigned) ((n + WG - 1) / WG)), dim3(WG), 0, st, sc, rays, n, out); }
void mi_launch_debug_intersect(const DScene &sc, const float *rays, uint64_t n, int anyHit, float *out, int *outInst, hipStream_t st) { hipLaunchKernelGGL(k_debug_intersect, dim3((unsigned) ((n + WG - 1) / WG)), dim3(WG), 0, st, sc, rays, n, anyHit, out, outInst); }
}
