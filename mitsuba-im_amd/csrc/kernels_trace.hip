// kernels_trace.hip -- extend (closest hit) and shadow (any hit) stages, unit-level intersection entry point (see kernels_common.h)
#include "kernels_common.h"
#include "trace.h"

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


// ---------------------------------------------------------------------------------------------- tree traversal with dynamic ray fetch
// The BVH stages (STACK > 0).  In k_extend / k_shadow above a lane takes its next ray only when ALL 64 lanes of its wave have finished theirs, so a wave idles
// behind its longest traversal.  Here the rays of a segment are handed out from a workgroup counter in LDS: whenever enough lanes of a wave are idle
// (MI_REFILL of 64, or all of them) the idle lanes fetch the next rays together (one LDS atomic per wave, ranks by ballot) while the other lanes keep their
// traversal state -- the persistent "while-while" loop of Aila & Laine 2009, confined to the workgroup's own segment, so queue ownership and the
// 16-B-per-lane record layout stay as they are.  Same arithmetic as traverse() (trace.h), same results: traversal order never changes the closest hit.
#define MI_REFILL 20
template <int STACK, int AN, bool WIDE, bool ANY>
__global__ __launch_bounds__(WG) void k_trace(DScene sc, Queues q, int buf) {
    __shared__ int s_stk[STACK * WG];
    __shared__ uint32_t s_next;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const f4 *nodes4 = reinterpret_cast<const f4 *>(sc.nodes);
    const f4 *tris4 = reinterpret_cast<const f4 *>(sc.tris);
    int *stk = s_stk + tid;
    unsigned long long rays = 0;
    for (uint32_t seg = blockIdx.x; seg < q.n_seg; seg += gridDim.x) {
        const uint32_t n = ANY ? q.shCount[seg] : q.count[buf][seg];
        const uint64_t segBase = (uint64_t) seg * q.cap;
        if (!ANY) rays += n;
        __syncthreads();                                   // every wave is done with the previous segment's counter
        if (tid == 0) s_next = 0;
        __syncthreads();
        // per-lane traversal state (see traverse() in trace.h for the meaning of each)
        bool active = false, drained = false; uint32_t myIdx = 0;
        v3 o = V(0, 0, 0), d = o, o0 = o, d0 = o, inv = o, oi = o; float mint = 0, mint0 = 0, cap = INFINITY;
        int cur = BVH_DONE, sp = 0, curInst = -1, binst = -1;
        float best = 0, bu = 0, bv = 0; uint32_t bprim = 0xFFFFFFFFu; bool found = false; uint32_t pidS = 0;
#define BVH_POP() do { \
        if (sp > 0) { --sp; cur = stk[sp * WG]; \
            if ((AN & 2) && cur == BVH_RET) { o = o0; d = d0; mint = mint0; cap = INFINITY; curInst = -1; \
                inv = V(safeInv(d.x), safeInv(d.y), safeInv(d.z)); oi = V(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z); \
                if (sp > 0) { --sp; cur = stk[sp * WG]; } else cur = BVH_DONE; } \
            if (WIDE && cur != BVH_DONE) { \
                const uint32_t e_ = (uint32_t) cur, node_ = e_ & 0x7FFFFFu, slots_ = (e_ >> 23) & 0x3Fu, more_ = (e_ >> 29) & 3u; \
                if (more_) { stk[sp * WG] = (int) (node_ | ((slots_ >> 2) << 23) | ((more_ - 1u) << 29)); ++sp; } \
                const f4 ch_ = nodes4[node_ * 4u + 3u]; const uint32_t sl_ = slots_ & 3u; \
                cur = __float_as_int(sl_ == 0u ? ch_.x : (sl_ == 1u ? ch_.y : (sl_ == 2u ? ch_.z : ch_.w))); } \
        } else cur = BVH_DONE; } while (0)
        while (true) {
            const unsigned long long idle = __ballot(!active);
            const uint32_t nIdle = (uint32_t) __popcll(idle);
            if (!drained && (nIdle >= MI_REFILL || nIdle == 64u)) {      // (wave-uniform) hand the idle lanes their next rays
                uint32_t base = 0;
                const int leader = __builtin_ctzll(idle);
                if ((int) lane == leader) base = atomicAdd(&s_next, nIdle);
                base = (uint32_t) __shfl((int) base, leader);
                drained = base + nIdle >= n;                // the counter has passed the end of the segment: no further fetches by this wave
                if (!active) {
                    const uint32_t idx = base + (uint32_t) __popcll(idle & lt);
                    if (idx < n) {
                        float4 ro = ANY ? q.shO[segBase + idx] : q.rayO[buf][segBase + idx], rd = ANY ? q.shD[segBase + idx] : q.rayD[buf][segBase + idx];
                        o = V(ro.x, ro.y, ro.z); d = V(rd.x, rd.y, rd.z); myIdx = idx; if (ANY) pidS = __float_as_uint(rd.w);
                        float maxt;
                        found = false; bprim = 0xFFFFFFFFu; bu = bv = 0; binst = -1; curInst = -1; cap = INFINITY; sp = 0;
                        if (clipInterval(sc, o, d, ANY ? MI_EPSILON : ro.w, ANY ? ro.w : rd.w, ANY, mint, maxt)) {
                            best = maxt; o0 = o; d0 = d; mint0 = mint; cur = 0; active = true;
                            inv = V(safeInv(d.x), safeInv(d.y), safeInv(d.z)); oi = V(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
                        } else {                             // the ray misses the scene box: its result right away
                            if (ANY) { float4 c = q.shC[segBase + idx]; float4 a = q.acc[pidS]; a.x += c.x; a.y += c.y; a.z += c.z; q.acc[pidS] = a; }
                            else { q.hit[segBase + idx] = make_float4(0, 0, 0, __uint_as_float(0xFFFFFFFFu)); if ((AN & 2) && q.hitInst) q.hitInst[segBase + idx] = -1; }
                        }
                    }
                }
            }
            if (__ballot(active) == 0ull) { if (drained) break; continue; }
            if (active) {
                // ---- inner nodes until this lane holds a leaf (or is done)
                if (WIDE) {
                    while (cur >= 0 && cur != BVH_DONE) {
                        const f4 n0 = nodes4[cur * 4 + 0], n1 = nodes4[cur * 4 + 1], n2 = nodes4[cur * 4 + 2], n3 = nodes4[cur * 4 + 3];
                        const uint32_t ex = __float_as_uint(n0.w);
                        const float sx = __uint_as_float((ex & 0xFFu) << 23), sy = __uint_as_float(((ex >> 8) & 0xFFu) << 23), sz = __uint_as_float(((ex >> 16) & 0xFFu) << 23);
                        const float bx = sx * inv.x, by = sy * inv.y, bz = sz * inv.z;
                        const float ax = __builtin_fmaf(n0.x, inv.x, oi.x), ay = __builtin_fmaf(n0.y, inv.y, oi.y), az = __builtin_fmaf(n0.z, inv.z, oi.z);
                        const uint32_t lx = __float_as_uint(n1.x), ly = __float_as_uint(n1.y), lz = __float_as_uint(n1.z), hx = __float_as_uint(n1.w), hy = __float_as_uint(n2.x), hz = __float_as_uint(n2.y);
                        const uint32_t nxq = inv.x >= 0 ? lx : hx, fxq = inv.x >= 0 ? hx : lx, nyq = inv.y >= 0 ? ly : hy, fyq = inv.y >= 0 ? hy : ly, nzq = inv.z >= 0 ? lz : hz, fzq = inv.z >= 0 ? hz : lz;
                        const float far = (AN & 2) ? fminf(best, cap) : best;
                        uint32_t key[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float tn = fmaxf(fmaxf(__builtin_fmaf((float) ((nxq >> (8 * c)) & 0xFFu), bx, ax), __builtin_fmaf((float) ((nyq >> (8 * c)) & 0xFFu), by, ay)),
                                                   fmaxf(__builtin_fmaf((float) ((nzq >> (8 * c)) & 0xFFu), bz, az), mint));
                            const float tf = fminf(fminf(__builtin_fmaf((float) ((fxq >> (8 * c)) & 0xFFu), bx, ax), __builtin_fmaf((float) ((fyq >> (8 * c)) & 0xFFu), by, ay)),
                                                   fminf(__builtin_fmaf((float) ((fzq >> (8 * c)) & 0xFFu), bz, az), far));
                            key[c] = (tn <= tf * 1.000002f + 1e-30f) ? ((__float_as_uint(tn) & ~3u) | (uint32_t) c) : 0xFFFFFFFFu;
                        }
                        {   uint32_t a = min(key[0], key[1]), b = max(key[0], key[1]), c = min(key[2], key[3]), e = max(key[2], key[3]);
                            key[0] = min(a, c); const uint32_t m1 = max(a, c), m2 = min(b, e); key[3] = max(b, e); key[1] = min(m1, m2); key[2] = max(m1, m2); }
                        const int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y), c2 = __float_as_int(n3.z), c3 = __float_as_int(n3.w);
                        if (key[0] == 0xFFFFFFFFu) BVH_POP();
                        else {
                            const uint32_t more = (key[1] != 0xFFFFFFFFu) + (key[2] != 0xFFFFFFFFu) + (key[3] != 0xFFFFFFFFu);
                            if (more) { stk[sp * WG] = (int) ((uint32_t) cur | ((key[1] & 3u) << 23) | ((key[2] & 3u) << 25) | ((key[3] & 3u) << 27) | ((more - 1u) << 29)); ++sp; }
                            const uint32_t sl = key[0] & 3u; cur = sl == 0u ? c0 : (sl == 1u ? c1 : (sl == 2u ? c2 : c3));
                        }
                    }
                } else {
                    while (cur >= 0 && cur != BVH_DONE) {
                        f4 n0 = nodes4[cur * 4 + 0], n1 = nodes4[cur * 4 + 1], n2 = nodes4[cur * 4 + 2], n3 = nodes4[cur * 4 + 3];
                        int c0 = __float_as_int(n0.w), c1 = __float_as_int(n1.w);
                        float t0, t1;
                        const float far = (AN & 2) ? fminf(best, cap) : best;
                        bool h0 = slab(n0, n1, inv, oi, mint, far, t0), h1 = slab(n2, n3, inv, oi, mint, far, t1);
                        if (h0 && h1) { bool swap = t1 < t0; stk[sp * WG] = swap ? c0 : c1; ++sp; cur = swap ? c1 : c0; }
                        else if (h0) cur = c0;
                        else if (h1) cur = c1;
                        else BVH_POP();
                    }
                }
                bool finished = cur == BVH_DONE;
                if (!finished) {
                    // ---- the leaf
                    uint32_t code = (uint32_t) ~cur; uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
                    bool entered = false;
                    for (uint32_t i = 0; i < cnt; ++i) {
                        f4 a = tris4[(first + i) * 3 + 0], b = tris4[(first + i) * 3 + 1], c = tris4[(first + i) * 3 + 2];
                        TriAccelD ta; ta.k = __float_as_uint(a.x); ta.n_u = a.y; ta.n_v = a.z; ta.n_d = a.w;
                        ta.a_u = b.x; ta.a_v = b.y; ta.b_nu = b.z; ta.b_nv = b.w; ta.c_nu = c.x; ta.c_nv = c.y; ta.prim = __float_as_uint(c.z);
                        float u, v, t; bool ok;
                        if ((AN & 2) && ta.k == MI_K_INSTANCE) {
                            const InstanceD &in = sc.instances[ta.prim];
                            v3 o2 = xfPoint(in.to_object, o0), d2 = xfVector(in.to_object, d0);
                            float nearT, farT;
                            if (aabbRay(in.glo, in.ghi, o2, d2, nearT, farT)) {
                                const float mi = mint0 > nearT ? mint0 : nearT, ma = best < farT ? best : farT;
                                if (ma > mi) {
                                    stk[sp * WG] = BVH_RET; ++sp;
                                    o = o2; d = d2; mint = mi; cap = farT; curInst = (int) ta.prim;
                                    inv = V(safeInv(d.x), safeInv(d.y), safeInv(d.z)); oi = V(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
                                    cur = in.root; entered = true;
                                }
                            }
                            break;
                        }
                        const float far = (AN & 2) ? fminf(best, cap) : best;
                        if ((AN & 1) && ta.k == MI_K_ANALYTIC) ok = analyticIntersect<ANY>(sc.analytic[ta.prim - sc.n_tris], o, d, mint, far, t, u, v);
                        else ok = triIntersect(ta, o, d, mint, far, u, v, t);
                        if (ok) {
                            if (ANY) { found = true; break; }
                            if (!found || t < best || (t == best && (ta.prim < bprim || (ta.prim == bprim && curInst < binst)))) { best = t; bprim = ta.prim; binst = curInst; bu = u; bv = v; found = true; }
                        }
                    }
                    if (ANY && found) finished = true;
                    else if (!entered) BVH_POP();
                }
                if (finished) {
                    if (ANY) { if (!found) { float4 c = q.shC[segBase + myIdx]; float4 a = q.acc[pidS]; a.x += c.x; a.y += c.y; a.z += c.z; q.acc[pidS] = a; } }
                    else { q.hit[segBase + myIdx] = make_float4(best, bu, bv, __uint_as_float(found ? bprim : 0xFFFFFFFFu)); if ((AN & 2) && q.hitInst) q.hitInst[segBase + myIdx] = binst; }
                    active = false;
                }
            }
        }
#undef BVH_POP
    }
    if (!ANY && tid == 0 && rays) atomicAdd(&q.counters[0], rays);
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
static const bool kNoDynamicFetch = getenv("MI355PT_NO_DYNAMIC_FETCH") != nullptr;      // A/B switch: the wave-synchronous k_extend / k_shadow for tree scenes too
#define MI_TRACE_W(AN, W, ANYHIT) do { \
    if (sc.bvh_depth <= 8) hipLaunchKernelGGL((k_trace<8, AN, W, ANYHIT>), dim3(grid), dim3(WG), 0, st, sc, q, buf); \
    else if (sc.bvh_depth <= 12) hipLaunchKernelGGL((k_trace<12, AN, W, ANYHIT>), dim3(grid), dim3(WG), 0, st, sc, q, buf); \
    else if (sc.bvh_depth <= 16) hipLaunchKernelGGL((k_trace<16, AN, W, ANYHIT>), dim3(grid), dim3(WG), 0, st, sc, q, buf); \
    else if (sc.bvh_depth <= 24) hipLaunchKernelGGL((k_trace<24, AN, W, ANYHIT>), dim3(grid), dim3(WG), 0, st, sc, q, buf); \
    else hipLaunchKernelGGL((k_trace<STACK_DEPTH, AN, W, ANYHIT>), dim3(grid), dim3(WG), 0, st, sc, q, buf); } while (0)
#define MI_TRACE(AN, ANYHIT) do { if (sc.bvh_wide) MI_TRACE_W(AN, true, ANYHIT); else MI_TRACE_W(AN, false, ANYHIT); } while (0)
void mi_launch_extend(const DScene &sc, const Queues &q, int buf, uint32_t grid, hipStream_t st) {
    const int mode = (sc.n_analytic ? 1 : 0) | (sc.n_instances ? 2 : 0);
    if (!sc.packet_n && !kNoDynamicFetch) { if (mode == 3) MI_TRACE(3, false); else if (mode == 2) MI_TRACE(2, false); else if (mode == 1) MI_TRACE(1, false); else MI_TRACE(0, false); return; }
    if (mode == 3) MI_BY_STACK(k_extend, 3, sc, q, buf); else if (mode == 2) MI_BY_STACK(k_extend, 2, sc, q, buf); else if (mode == 1) MI_BY_STACK(k_extend, 1, sc, q, buf); else MI_BY_STACK(k_extend, 0, sc, q, buf);
}
void mi_launch_shadow(const DScene &sc, const Queues &q, uint32_t grid, hipStream_t st) {
    const int mode = (sc.n_analytic ? 1 : 0) | (sc.n_instances ? 2 : 0); const int buf = 0;
    if (!sc.packet_n && !kNoDynamicFetch) { if (mode == 3) MI_TRACE(3, true); else if (mode == 2) MI_TRACE(2, true); else if (mode == 1) MI_TRACE(1, true); else MI_TRACE(0, true); return; }
    if (mode == 3) MI_BY_STACK(k_shadow, 3, sc, q); else if (mode == 2) MI_BY_STACK(k_shadow, 2, sc, q); else if (mode == 1) MI_BY_STACK(k_shadow, 1, sc, q); else MI_BY_STACK(k_shadow, 0, sc, q);
}
#undef MI_BY_STACK
#undef MI_BY_STACK_W
#undef MI_TRACE
#undef MI_TRACE_W
void mi_launch_ray_intersect(const DScene &sc, const float *rays, uint64_t n, mi_intersection *out, hipStream_t st) { hipLaunchKernelGGL(k_ray_intersect, dim3((unsigned) ((n + WG - 1) / WG)), dim3(WG), 0, st, sc, rays, n, out); }
void mi_launch_debug_intersect(const DScene &sc, const float *rays, uint64_t n, int anyHit, float *out, int *outInst, hipStream_t st) { hipLaunchKernelGGL(k_debug_intersect, dim3((unsigned) ((n + WG - 1) / WG)), dim3(WG), 0, st, sc, rays, n, anyHit, out, outInst); }
}
