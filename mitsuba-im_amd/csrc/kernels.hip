// kernels.hip -- gfx950 (MI355X) wavefront path tracer: the stage kernels.
//
// Replaces the per-sample loop SamplingIntegrator::renderBlock -> MIPathTracer::Li -> Scene::rayIntersect / BSDF / emitter /
// sampler plugins (reference src/librender/integrator.cpp:141-189, src/integrators/path/path.cpp:119-294) by stages over
// SoA queues in HBM:
//     generate -> [ extend (closest hit) -> shade (MIS bookkeeping, RR, NEE sample, BSDF sample) -> shadow (any hit) ] x depth -> film
// Work ownership: the path pool of a batch is cut into `n_seg` contiguous SEGMENTS; a workgroup owns whole segments
// (segments b, b + gridDim.x, ...) in every stage, so a segment is produced and consumed by one workgroup at a time.  Stream compaction therefore never leaves the workgroup: wave64 ballots +
// one LDS exchange per 256-path chunk, no global atomics, fully coalesced queue reads/writes, and a workgroup re-reads what
// it (= the same XCD's L2 under round-robin dispatch) wrote in the previous stage.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "pt_device.h"
#include "queues.h"

// This file is compiled twice into libmi355pt.so:
//   precise TU (kernels.hip):       -ffp-contract=off, IEEE divide/sqrt  -> radiance bit-identical to the strict-IEEE oracle;
//   fast TU (kernels_fast.hip):     -ffp-contract=fast, approximate divide/sqrt (the reference itself is built with -ffast-math)
//                                   -> same code, results within float rounding (tests: <= 1e-4 relative, forks aside).
// Everything lives in a per-TU namespace so the two sets of kernel / template instantiations cannot be merged by the linker.
#ifdef MI_FAST_MATH
#define MI_NS mi_fast
#define MI_FN(x) x##_fast
#else
#define MI_NS mi_precise
#define MI_FN(x) x
#endif
namespace MI_NS {

#define WG 256
#define STACK_DEPTH 32   // >= BVH depth (scene_build.cpp caps it; mi_scene_commit refuses deeper trees)

// ---------------------------------------------------------------------------------------------- BVH traversal
// Conservative slab test against a (padded) child box; returns the entry distance.  Pure culling, so it is free to use fused
// multiply-adds: t = lo * inv + (-o * inv).  NaN-free: zero direction components are replaced by +-1e-30 before the reciprocal.
DEV bool slab(f4 lo, f4 hi, v3 inv, v3 oi, float tmin, float tmax, float &tnear) {
    float ax = __builtin_fmaf(lo.x, inv.x, oi.x), bx = __builtin_fmaf(hi.x, inv.x, oi.x);
    float ay = __builtin_fmaf(lo.y, inv.y, oi.y), by = __builtin_fmaf(hi.y, inv.y, oi.y);
    float az = __builtin_fmaf(lo.z, inv.z, oi.z), bz = __builtin_fmaf(hi.z, inv.z, oi.z);
    float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
    float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    tnear = t0;
    return t0 <= t1 * 1.000002f + 1e-30f;
}
// include/mitsuba/core/aabb.h:308-339 TAABB::rayIntersect(ray, nearT, farT), exact arithmetic (used for the group box of an instance)
DEV bool aabbRay(const float *lo, const float *hi, v3 o, v3 d, float &nearT, float &farT) {
    float nt = -INFINITY, ft = INFINITY;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float origin = oo[i], minv = lo[i], maxv = hi[i], di = dd[i];
        if (di == 0) { if (origin < minv || origin > maxv) return false; }
        else {
            float rcp = 1.0f / di;
            float t1 = (minv - origin) * rcp, t2 = (maxv - origin) * rcp;
            if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; }
            nt = maxf(t1, nt); ft = minf(t2, ft);
            if (!(nt <= ft)) return false;
        }
    }
    nearT = nt; farT = ft; return true;
}
DEV float safeInv(float d) { float a = fabsf(d) < 1e-30f ? copysignf(1e-30f, d) : d; return 1.0f / a; }

// Closest hit: minimum t, ties towards the lower original triangle index (order independent).  `stk` points at this lane's column
// of the workgroup's LDS stack (stride WG).  "while-while" shape: all lanes of a wave first descend through inner nodes until each
// holds a leaf (or is done), then all test their leaf triangles -- the two code paths are not interleaved lane by lane.
#define BVH_DONE 0x7FFFFFFF
#define BVH_RET 0x7FFFFFFE                  // stack marker: leave the current instance, back to the scene-level ray
// AN ("extended" geometry): leaf records may be analytic shapes (k = MI_K_ANALYTIC) or instances of shape groups (k = MI_K_INSTANCE).
// Instance::rayIntersect (src/shapes/instance.cpp:91-108): the ray is taken to the group's object space (direction NOT renormalised, so t keeps
// its meaning), [mint, maxt] is clipped against the group's kd-tree box (skdtree.h:431-452), then the group's own BVH -- stored in the same
// node array -- is walked with the same stack; a marker entry brings the walk back to the scene level.  Instances sit alone in their leaves.
template <bool ANY, int AN>        // AN bit 0: analytic shapes present, bit 1: instances present (each kernel variant carries only the code it needs)
DEV bool traverse(const DScene &sc, v3 o, v3 d, float mint, float maxt, int *stk,
                  float &bestT, uint32_t &bestPrim, float &bestU, float &bestV, int &bestInst) {
    v3 inv = V(safeInv(d.x), safeInv(d.y), safeInv(d.z));
    v3 oi = V(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
    const v3 o0 = o, d0 = d; const float mint0 = mint;
    float cap = INFINITY;                       // inside an instance: far end of the group-box interval
    int curInst = -1, binst = -1;
    const f4 *nodes4 = reinterpret_cast<const f4 *>(sc.nodes);
    const f4 *tris4 = reinterpret_cast<const f4 *>(sc.tris);
    float best = maxt; uint32_t bprim = 0xFFFFFFFFu; bool found = false; float bu = 0, bv = 0;
    int sp = 0; int cur = 0;
#define BVH_POP() do { \
        if (sp > 0) { --sp; cur = stk[sp * WG]; \
            if ((AN & 2) && cur == BVH_RET) { o = o0; d = d0; mint = mint0; cap = INFINITY; curInst = -1; \
                inv = V(safeInv(d.x), safeInv(d.y), safeInv(d.z)); oi = V(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z); \
                if (sp > 0) { --sp; cur = stk[sp * WG]; } else cur = BVH_DONE; } \
        } else cur = BVH_DONE; } while (0)
    while (true) {
        while (cur >= 0 && cur != BVH_DONE) {
            f4 n0 = nodes4[cur * 4 + 0], n1 = nodes4[cur * 4 + 1], n2 = nodes4[cur * 4 + 2], n3 = nodes4[cur * 4 + 3];
            int c0 = __float_as_int(n0.w), c1 = __float_as_int(n1.w);
            float t0, t1;
            const float far = (AN & 2) ? fminf(best, cap) : best;
            bool h0 = slab(n0, n1, inv, oi, mint, far, t0), h1 = slab(n2, n3, inv, oi, mint, far, t1);
            if (h0 && h1) {
                bool swap = t1 < t0;
                stk[sp * WG] = swap ? c0 : c1; ++sp;
                cur = swap ? c1 : c0;
            } else if (h0) cur = c0;
            else if (h1) cur = c1;
            else BVH_POP();
        }
        if (cur == BVH_DONE) break;
        {
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
                    break;                                                             // an instance is alone in its leaf
                }
                const float far = (AN & 2) ? fminf(best, cap) : best;
                if ((AN & 1) && ta.k == MI_K_ANALYTIC) ok = analyticIntersect<ANY>(sc.analytic[ta.prim - sc.n_tris], o, d, mint, far, t, u, v);   // skdtree.h:292-301
                else ok = triIntersect(ta, o, d, mint, far, u, v, t);
                if (ok) {
                    if (ANY) return true;
                    if (!found || t < best || (t == best && (ta.prim < bprim || (ta.prim == bprim && curInst < binst)))) { best = t; bprim = ta.prim; binst = curInst; bu = u; bv = v; found = true; }
                }
            }
            if (!entered) BVH_POP();
        }
    }
#undef BVH_POP
    bestT = best; bestPrim = bprim; bestU = bu; bestV = bv; bestInst = binst;
    return found;
}

// Packet mode (scenes of <= MI_PACKET_MAX triangles): the scene is one triangle packet in constant memory.  The loop index is
// wave-uniform, so the Wald records arrive through the scalar cache as SGPR operands: no per-lane loads, no stack, no divergence besides
// lanes that have left the loop.  The one scalar unit of a CU serves all 32 resident waves, so scalar work per triangle is kept minimal:
// the records are sorted by projection axis k on the host and each axis gets its own loop (compile-time component selection, no
// per-triangle branch); ties in t are broken towards the lower ORIGINAL triangle index, which keeps the result order independent.
__constant__ TriAccelD c_packet[MI_PACKET_MAX];
__constant__ AnalyticD c_analytic[MI_ANALYTIC_PACKET_MAX];   // analytic shapes of a packet-mode scene (wave-uniform records as well)
template <int K> DEV bool triIntersectK(const TriAccelD &ta, v3 o, v3 d, float mint, float maxt, float &u, float &v, float &t) {
    const float o_u = K == 0 ? o.y : (K == 1 ? o.z : o.x), o_v = K == 0 ? o.z : (K == 1 ? o.x : o.y), o_k = K == 0 ? o.x : (K == 1 ? o.y : o.z);
    const float d_u = K == 0 ? d.y : (K == 1 ? d.z : d.x), d_v = K == 0 ? d.z : (K == 1 ? d.x : d.y), d_k = K == 0 ? d.x : (K == 1 ? d.y : d.z);
    float tt = (ta.n_d - o_u * ta.n_u - o_v * ta.n_v - o_k) / (d_u * ta.n_u + d_v * ta.n_v + d_k);
    if (tt < mint || tt > maxt) return false;
    float hu = o_u + tt * d_u - ta.a_u, hv = o_v + tt * d_v - ta.a_v;
    float uu = hv * ta.b_nu + hu * ta.b_nv, vv = hu * ta.c_nu + hv * ta.c_nv;
    u = uu; v = vv; t = tt;
    return uu >= 0 && vv >= 0 && uu + vv <= 1.0f;
}
template <bool ANY, int K>
DEV bool packetLoop(uint32_t first, uint32_t last, v3 o, v3 d, float mint, float &best, uint32_t &bprim, float &bu, float &bv, bool &found) {
    for (uint32_t i = first; i < last; ++i) {
        const TriAccelD ta = c_packet[i];
        float u, v, t;
        if (triIntersectK<K>(ta, o, d, mint, best, u, v, t)) {
            if (ANY) return true;
            if (!found || t < best || ta.prim < bprim) { best = t; bprim = ta.prim; bu = u; bv = v; found = true; }   // t <= best here: equal t -> lower prim wins
        }
    }
    return false;
}
template <bool ANY, int AN>
DEV bool packetIntersect(const DScene &sc, v3 o, v3 d, float mint, float maxt, float &bestT, uint32_t &bestPrim, float &bestU, float &bestV) {
    float best = maxt; uint32_t bprim = 0xFFFFFFFFu; bool found = false; float bu = 0, bv = 0;
    if (packetLoop<ANY, 0>(0, sc.packet_k[0], o, d, mint, best, bprim, bu, bv, found)) return true;
    if (packetLoop<ANY, 1>(sc.packet_k[0], sc.packet_k[1], o, d, mint, best, bprim, bu, bv, found)) return true;
    if (packetLoop<ANY, 2>(sc.packet_k[1], sc.packet_k[2], o, d, mint, best, bprim, bu, bv, found)) return true;
    if (AN & 1) {
        for (uint32_t i = 0; i < sc.n_analytic; ++i) {
            float u, v, t; const uint32_t prim = sc.n_tris + i;
            if (analyticIntersect<ANY>(c_analytic[i], o, d, mint, best, t, u, v)) {
                if (ANY) return true;
                if (!found || t < best || prim < bprim) { best = t; bprim = prim; bu = u; bv = v; found = true; }
            }
        }
    }
    bestT = best; bestPrim = bprim; bestU = bu; bestV = bv;
    return found;
}

// ---------------------------------------------------------------------------------------------- generate
// One camera sample per path: src/librender/integrator.cpp:166-181 (pixel offset + sensor ray), sampler set-up
// src/samplers/sobol.cpp:171-216.  Path q of the batch = (plane q / npix, tile pixel q % npix).
__global__ __launch_bounds__(WG) void k_generate(DScene sc, RenderConst rc, Queues q, BatchDesc bd) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t seg = blockIdx.x; seg < q.n_seg; seg += gridDim.x) {
    const uint64_t segBase = (uint64_t) seg * q.cap;
    uint64_t remaining = bd.n_paths > segBase ? bd.n_paths - segBase : 0;
    const uint32_t n = remaining > q.cap ? q.cap : (uint32_t) remaining;
    const uint32_t tw = bd.tile.x1 - bd.tile.x0;
    for (uint32_t i = tid; i < n; i += WG) {
        const uint64_t pid = segBase + i;
        uint32_t plane = (uint32_t) (pid / bd.n_pix), pl = (uint32_t) (pid % bd.n_pix);
        uint32_t px = bd.tile.x0 + pl % tw, py = bd.tile.y0 + (pl / tw) * bd.row_stride, sidx = bd.sample_begin + plane;
        if (bd.list) { px = bd.list[pid * 3]; py = bd.list[pid * 3 + 1]; sidx = bd.list[pid * 3 + 2]; }
        SamplerState ss; float jx, jy;
        if (rc.sampler == 1) {
            uint64_t idx = sc.log_res > 1 ? sobolLookUp(sc.sobol_vdc, sc.sobol_vdc_inv, sc.log_res, sidx, px, py, rc.sobol_scramble) : (uint64_t) sidx;
            ss.a = (uint32_t) idx; ss.b = (uint32_t) (idx >> 32); ss.dim = 0;
            const SobolTab gt{rc.sobol_nib, rc.nib_count, rc.sobol_scramble};
            jx = sobolSampleNib(gt, ss.a, ss.b, 0); jy = sobolSampleNib(gt, ss.a, ss.b, 1); ss.dim = 2;
            if (idx != (uint64_t) sidx) {      // sobol.cpp:239-245: rescale the first two dimensions to a pixel-relative offset
                jx = jx * sc.resolution - (float) (int) px; jy = jy * sc.resolution - (float) (int) py;
            }
        } else {
            ss.a = (py * sc.width + px) ^ rc.seed_mix; ss.b = sidx; ss.dim = 0;
            next2D(ss, 0, SobolTab{nullptr, 0, 0}, jx, jy);
        }
        float sx = (float) (int) px + jx, sy = (float) (int) py + jy;
        v3 o, d; float mint, maxt; cameraRay(sc, sx, sy, o, d, mint, maxt);
        const uint64_t slot = segBase + i;
        q.rayO[0][slot] = make_float4(o.x, o.y, o.z, mint);
        q.rayD[0][slot] = make_float4(d.x, d.y, d.z, maxt);
        // packed: dim | depth << 8 | flags << 16   (flags bit0: facingRef of the previous vertex)
        q.st0[0][slot] = make_uint4((uint32_t) pid, ss.a, ss.b, ss.dim | (1u << 8));
        q.st1[0][slot] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);      // throughput rgb, eta
        q.st2[0][slot] = 0.0f;                                      // bsdfPdf of the segment that produced this ray
        q.pos[pid] = make_float2(sx, sy);
        q.acc[pid] = make_float4(0, 0, 0, 1.0f);        // Li rgb, alpha (newQuery: alpha = 1, integrator.h:223-229)
    }
    if (tid == 0) q.count[0][seg] = n;
    }
}

// ---------------------------------------------------------------------------------------------- extend
// Scene::rayIntersect -> ShapeKDTree::rayIntersect (src/librender/skdtree.cpp:112-142): closest hit (t, u, v, prim)
template <int STACK, int AN>   // STACK = 0: packet mode; else LDS stack entries per lane (>= BVH depth); AN: bit 0 analytic shapes, bit 1 instances
__global__ __launch_bounds__(WG) void k_extend(DScene sc, Queues q, int buf) {
    __shared__ int s_stk[(STACK > 0 ? STACK : 1) * WG];
    const uint32_t tid = threadIdx.x;
    unsigned long long rays = 0;
    for (uint32_t seg = blockIdx.x; seg < q.n_seg; seg += gridDim.x) {
    const uint32_t n = q.count[buf][seg];
    const uint64_t segBase = (uint64_t) seg * q.cap;
    rays += n;
    for (uint32_t i = tid; i < n; i += WG) {
        float4 ro = q.rayO[buf][segBase + i], rd = q.rayD[buf][segBase + i];
        v3 o = V(ro.x, ro.y, ro.z), d = V(rd.x, rd.y, rd.z);
        float mint, maxt, t = 0, u = 0, v = 0; uint32_t prim = 0xFFFFFFFFu; bool hit = false; int inst = -1;
        if (clipInterval(sc, o, d, ro.w, rd.w, false, mint, maxt)) {
            if (STACK == 0) hit = packetIntersect<false, AN>(sc, o, d, mint, maxt, t, prim, u, v);
            else hit = traverse<false, AN>(sc, o, d, mint, maxt, s_stk + tid, t, prim, u, v, inst);
        }
        q.hit[segBase + i] = make_float4(t, u, v, __uint_as_float(hit ? prim : 0xFFFFFFFFu));
        if ((AN & 2) && q.hitInst) q.hitInst[segBase + i] = inst;
    }
    }
    if (tid == 0 && rays) atomicAdd(&q.counters[0], rays);
}

// Camera rays that leave the scene: EnvironmentMap::evalEnvironment WITH ray differentials (src/emitters/envmap.cpp:384-416) -- texture-space partials
// of the sensor ray's rx / ry directions, then TMIPMap::eval (EWA, anisotropy <= 10; u repeats, v clamps) over the map's MIP pyramid (input data).
// Runs once per batch between the first extend and the first shade; throughput is 1 at depth 1.
__global__ __launch_bounds__(WG) void k_env_primary(DScene sc, RenderConst rc, Queues q, int buf) {
    const TextureD tx = sc.textures[sc.env_texture - 1u];
    for (uint32_t seg = blockIdx.x; seg < q.n_seg; seg += gridDim.x) {
        const uint32_t n = q.count[buf][seg]; const uint64_t segBase = (uint64_t) seg * q.cap;
        for (uint32_t i = threadIdx.x; i < n; i += WG) {
            if (__float_as_uint(q.hit[segBase + i].w) != 0xFFFFFFFFu) continue;
            const float4 rd = q.rayD[buf][segBase + i]; const uint32_t pid = q.st0[buf][segBase + i].x;
            const v3 d = V(rd.x, rd.y, rd.z); const float2 sp = q.pos[pid];
            v3 rxd, ryd; cameraDifferentials(sc, rc.inv_sqrt_spp, sp.x, sp.y, d, rxd, ryd);
            const v3 v = mat3(sc.env_to_local, d);
            const float uvx = atan2f(v.x, -v.z) * MI_INV_TWOPI, uvy = acosf(minf(1.0f, maxf(-1.0f, v.y))) * MI_INV_PI;
            const v3 dvdx = mat3(sc.env_to_local, rxd) - v, dvdy = mat3(sc.env_to_local, ryd) - v;
            const float t1 = MI_INV_TWOPI / (v.x * v.x + v.z * v.z), t2 = -MI_INV_PI / maxf(sqrtf(maxf(0.0f, 1.0f - v.y * v.y)), MI_EPSILON);
            const v3 value = mipEval(sc, tx, uvx, uvy, t1 * (dvdx.z * v.x - dvdx.x * v.z), t2 * dvdx.y, t1 * (dvdy.z * v.x - dvdy.x * v.z), t2 * dvdy.y) * sc.env_scale;
            float4 a = q.acc[pid]; a.x += value.x; a.y += value.y; a.z += value.z; q.acc[pid] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------- shade
// One bounce of MIPathTracer::Li (src/integrators/path/path.cpp:135-287) for every live path of the segment:
//   tail of the previous iteration (emitter hit by the BSDF ray -> MIS term :257-264, Russian roulette :276-286), then
//   emitted radiance :148-150, depth test :156-165, emitter sampling :172-200 (visibility deferred to the shadow queue),
//   BSDF sampling :207-226.  Survivors are compacted into the other ray/state buffer, shadow rays into the shadow queue.
template <bool RC, bool ENV, bool SMALL, bool AN, bool TEX>   // TEX: textures bound to materials (implies AN); RC: rough conductors present; ENV: environment emitter present; SMALL: scene tables staged in LDS; AN ("extended"): analytic shapes or delta emitters (point / spot / directional) present
__global__ __launch_bounds__(WG) void k_shade(DScene sc, RenderConst rc, Queues q, int buf) {
    extern __shared__ uint32_t s_dyn[];
    uint32_t *s_nib = s_dyn;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = buf ^ 1;
    const SobolTabLds m32{(lds_u32_ptr) s_nib, rc.nib_count, rc.sobol_scramble};   // always the LDS copy (ds_read lookups); unused by the independent stream
    const uint32_t nibWords = rc.sampler == 1 ? rc.nib_dims * rc.nib_count * 16u : 4u;
    if (rc.sampler == 1) {   // stage the Sobol' nibble tables in LDS (nib_dims x nib_count x 16 words)
        for (uint32_t i = tid; i < nibWords; i += WG) s_nib[i] = rc.sobol_nib[i];
    }
    // Scene tables: shading records, materials, emitters, CDFs.  Small scenes (<= 128 triangles ...) are staged in LDS: the shading
    // stage chases hit -> triangle record -> material -> emitter CDF -> light triangle, and each hop is an L2 round trip otherwise.
    Tabs<SMALL> tb;
    if (SMALL) {
        uint32_t *base = s_dyn + ((nibWords + 3u) & ~3u);
        const uint32_t wShade = sc.n_tris * 24u, wMat = sc.n_materials * 16u, wEm = sc.n_emitters * 12u, wEc = (sc.n_emitters + 1u + 3u) & ~3u, wAc = sc.area_cdf_len;
        uint32_t *pS = base, *pM = pS + wShade, *pE = pM + wMat, *pEc = pE + wEm, *pAc = pEc + wEc;
        const uint32_t *gS = (const uint32_t *) sc.shade, *gM = (const uint32_t *) sc.materials, *gE = (const uint32_t *) sc.emitters, *gEc = (const uint32_t *) sc.emitter_cdf, *gAc = (const uint32_t *) sc.area_cdf;
        for (uint32_t i = tid; i < wShade; i += WG) pS[i] = gS[i];
        for (uint32_t i = tid; i < wMat; i += WG) pM[i] = gM[i];
        for (uint32_t i = tid; i < wEm; i += WG) pE[i] = gE[i];
        for (uint32_t i = tid; i < sc.n_emitters + 1u; i += WG) pEc[i] = gEc[i];
        for (uint32_t i = tid; i < wAc; i += WG) pAc[i] = gAc[i];
        tb.shade4 = (typename AS<SMALL>::p4) pS; tb.materials4 = (typename AS<SMALL>::p4) pM; tb.emitters4 = (typename AS<SMALL>::p4) pE;
        tb.emitter_cdf = (typename AS<SMALL>::pf) pEc; tb.area_cdf = (typename AS<SMALL>::pf) pAc;
    } else {
        tb.shade4 = (typename AS<SMALL>::p4) sc.shade; tb.materials4 = (typename AS<SMALL>::p4) sc.materials; tb.emitters4 = (typename AS<SMALL>::p4) sc.emitters;
        tb.emitter_cdf = (typename AS<SMALL>::pf) sc.emitter_cdf; tb.area_cdf = (typename AS<SMALL>::pf) sc.area_cdf;
    }
    // Material-sorted shading (scenes that mix BSDF classes): the paths of a segment are first ordered by the class of the surface
    // they hit -- diffuse-like from the front, rough conductors from the back of an LDS index list (wave64 ballots, order preserving) --
    // so a wave runs either the cheap diffuse code or the microfacet code, not both.  The queues are then read through that list.
    // In this stage a SEGMENT IS OWNED BY ONE WAVE: compaction is a pair of wave64 ballots with the running output offsets kept in
    // (uniform) registers -- no LDS exchange and no workgroup barrier anywhere in the loop; the four waves of a workgroup only share the
    // LDS copies of the tables above.
    uint16_t *s_order = reinterpret_cast<uint16_t *>(s_dyn + rc.order_offset_words) + (size_t) wave * q.cap;
    const bool doSort = RC && rc.order_offset_words != 0 && q.cap <= 0xFFFFu;
    unsigned long long pathLen = 0, shadowRays = 0;
    __syncthreads();                                         // tables staged
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t seg = blockIdx.x * (WG / 64) + wave; seg < q.n_seg; seg += gridDim.x * (WG / 64)) {
    const uint32_t n = q.count[buf][seg];
    const uint64_t segBase = (uint64_t) seg * q.cap;
    uint32_t outA = 0, outS = 0;                             // survivors / shadow records written so far (uniform)
    if (doSort) {
        uint32_t done0 = 0, done1 = 0;                       // uniform running counts (front / back)
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t i = base + lane; int cls = 2;
            if (i < n) {
                const uint32_t prim = __float_as_uint(q.hit[segBase + i].w);
                if (prim == 0xFFFFFFFFu) cls = 0;
                else if (AN && prim >= sc.n_tris) cls = (sc.analytic[prim - sc.n_tris].flags & 8u) ? 1 : 0;
                else cls = (__float_as_uint(tb.shade4[prim * 6u + 2u].w) & 8u) ? 1 : 0;
            }
            const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1);
            if (cls == 0) s_order[done0 + (uint32_t) __popcll(m0 & lt)] = (uint16_t) i;
            else if (cls == 1) s_order[n - 1u - (done1 + (uint32_t) __popcll(m1 & lt))] = (uint16_t) i;
            done0 += (uint32_t) __popcll(m0); done1 += (uint32_t) __popcll(m1);
        }
    }
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        // Two phases per chunk, each closed by its own wave64 ballot, so that the 12 registers of a shadow record are written out before
        // the BSDF-sampling code runs (register budget -> one more resident wave per SIMD):
        //   A: tail of the previous bounce, emitted radiance, emitter sampling  -> shadow queue
        //   B: BSDF sampling                                                     -> next ray / state
        bool alive = false, wantShadow = false, toSample = false;
        float4 shO, shD, shC;
        Hit h; MaterialD bsdf; SamplerState ss; v3 T = V(0, 0, 0); float eta = 1.0f; uint32_t pid = 0; int depth = 0; bool unscattered = false; v3 opac = V(1, 1, 1); bool masked = false;   // mask wrapper (mask.cpp): opacity in front of `bsdf` (RC variants)
        if (i < n) {
            const uint64_t slot = segBase + (doSort ? (uint32_t) s_order[i] : i);
            float4 rd = q.rayD[buf][slot], hr = q.hit[slot]; uint4 s0 = q.st0[buf][slot]; float4 s1 = q.st1[buf][slot];
            float prevPdf = q.st2[buf][slot];
            pid = s0.x; ss.a = s0.y; ss.b = s0.z; ss.dim = s0.w & 0xFFu;
            depth = (int) ((s0.w >> 8) & 0xFFu); const bool facingRef = ((s0.w >> 16) & 1u) != 0;
            const bool prevDelta = RC && ((s0.w >> 17) & 1u) != 0;      // the BSDF sample that spawned this ray was a delta component -> lumPdf = 0 (path.cpp:259-260)
            unscattered = RC && ((s0.w >> 18) & 1u) != 0;    // every component sampled so far was ENull (thin dielectric panes): `scattered` is still false (path.cpp:213)
            v3 d = V(rd.x, rd.y, rd.z); T = V(s1.x, s1.y, s1.z); eta = s1.w;
            const uint32_t prim = __float_as_uint(hr.w);
            v3 add = V(0, 0, 0); bool haveAdd = false;
            do {
                if (prim == 0xFFFFFFFFu) {                                     // miss: path.cpp:136-143 / :234-248
                    pathLen += (unsigned) (depth > 1 ? depth - 1 : 1);
                    if (depth == 1 && rc.opacity) { float4 a = q.acc[pid]; a.w = 0.0f; q.acc[pid] = a; }   // records.inl:121-137: alpha = 0 on a camera-ray miss
                    if (ENV) {
                        if (depth == 1) { if (!rc.hide_emitters && !sc.env_texture) { add = T * envEval(sc, d); haveAdd = true; } }     // path.cpp:139-141; with a MIP pyramid k_env_primary has added the filtered lookup
                        else {
                            // BSDF ray left the scene: env->evalEnvironment + fillDirectSamplingRecord (envmap.cpp:362-378), MIS term path.cpp:257-264
                            float4 ro = q.rayO[buf][slot]; float nearT, farT;
                            if (!(rc.hide_emitters && unscattered) && bsphereIntersect(sc, V(ro.x, ro.y, ro.z), d, nearT, farT) && !(nearT > 0) && !(farT < 0)) {   // path.cpp:238-239: hideEmitters && !scattered
                                v3 value = envEval(sc, d);
                                float pdfSA;
                                if (sc.env_constant) {      // ConstantBackgroundEmitter::pdfDirect (constant.cpp:219-233): needs the reference normal of the previous vertex
                                    const float c = q.st3[buf][slot]; pdfSA = c != 2.0f ? MI_INV_PI * maxf(0.0f, c) : MI_INV_FOURPI;
                                } else pdfSA = envPdfDirection(sc, mat3(sc.env_to_local, d));
                                float lumPdf = prevDelta ? 0.0f : pdfSA * (loadEmitter(tb, sc.env_index).weight * sc.emitter_norm);
                                add = (T * value) * miWeight(prevPdf, lumPdf); haveAdd = true;
                            }
                        }
                    }
                    break;
                }
                v3 ro3 = V(0, 0, 0);
                if (AN) { float4 ro = q.rayO[buf][slot]; ro3 = V(ro.x, ro.y, ro.z); }      // analytic shapes: hit point = o + t d; sphere lights: reference point of pdfDirect
                const int inst = (AN && q.hitInst) ? q.hitInst[slot] : -1;
                if (AN && inst >= 0) fillHitInstanced(sc, tb, sc.instances[inst], ro3, d, hr.x, prim, hr.y, hr.z, h);
                else if (AN && prim >= sc.n_tris) fillHitAnalytic(sc.analytic[prim - sc.n_tris], ro3, d, hr.x, hr.y, hr.z, h);
                else fillHit<SMALL, AN>(sc, tb, d, hr.x, prim, hr.y, hr.z, h);
                if (depth > 1) {
                    if (h.emitter >= 0) {                                    // path.cpp:229-233, 257-264
                        v3 value = emitterEval(tb, h.emitter, h.ns, -d);
                        float lumPdf = prevDelta ? 0.0f : pdfEmitterDirect<AN>(sc, tb, h.emitter, ro3, d, h.ns, h.dist, facingRef);
                        add = (T * value) * miWeight(prevPdf, lumPdf); haveAdd = true;
                    }
                    const int prevDepth = depth - 1;                         // rRec.depth++ >= m_rrDepth (path.cpp:276)
                    if (prevDepth >= rc.rr_depth) {
                        float qq = minf(maxf(maxf(T.x, T.y), T.z) * eta * eta, 0.95f);
                        if (next1D(ss, rc.sampler, m32) >= qq) { pathLen += (unsigned) depth; break; }
                        float r = 1.0f / qq; T = T * r;
                    }
                }
                if (!(depth <= rc.max_depth || rc.max_depth < 0)) { pathLen += (unsigned) depth; break; }   // loop guard path.cpp:135
                bsdf = loadMaterial(tb, h.material);
                auto applyTexture = [&](MaterialD &mm) {
                if (TEX) {                                                   // a textured parameter: m_reflectance->eval(bRec.its) (diffuse.cpp:112-121) and its siblings
                    const uint32_t tex = (mm.flags >> 8) & 0xFFFFu;
                    if (tex) {
                        const TextureD &tx = sc.textures[tex - 1]; v3 c;
                        float huvx = h.uvx, huvy = h.uvy;
                        const bool onAnalytic = AN && inst < 0 && prim >= sc.n_tris;         // an analytic shape: its own parameterisation, evaluated on demand
                        if (onAnalytic) { v3 du, dv; analyticUV(sc.analytic[prim - sc.n_tris], hr.y, hr.z, ro3 + d * hr.x, huvx, huvy, du, dv); }   // (tangents: recomputed below where a filtered lookup needs them)
                        if (tx.type == 2u) {                                 // BitmapTexture::eval (src/textures/bitmap.cpp:434-502) under Texture2D::eval (texture.cpp:112-121)
                            const float uvx = huvx * tx.uscale + tx.uoffset, uvy = huvy * tx.vscale + tx.voffset;
                            if (depth == 1) {                                // its.getBSDF(ray) -> computePartials: only the camera ray carries differentials (records.inl:68-75)
                                v3 dpdu, dpdv;
                                if (onAnalytic) { float tu_, tv_; analyticUV(sc.analytic[prim - sc.n_tris], hr.y, hr.z, ro3 + d * hr.x, tu_, tv_, dpdu, dpdv); }
                                else if (h.flags & 16u) { const TriUV &tu = sc.triuv[prim]; dpdu = ld3(tu.dpdu); dpdv = ld3(tu.dpdv); }
                                else { typename AS<SMALL>::p4 rec = tb.shade4 + prim * 6u; f4 r0 = rec[0], r1 = rec[1], r2 = rec[2]; dpdu = V(r1.x - r0.x, r1.y - r0.y, r1.z - r0.z); dpdv = V(r2.x - r0.x, r2.y - r0.y, r2.z - r0.z); }
                                if (inst >= 0) { dpdu = xfVector(sc.instances[inst].to_world, dpdu); dpdv = xfVector(sc.instances[inst].to_world, dpdv); }
                                const float2 sp = q.pos[pid]; v3 rxd, ryd; cameraDifferentials(sc, rc.inv_sqrt_spp, sp.x, sp.y, d, rxd, ryd);
                                float pa[4]; computePartials(h.p, h.ng, dpdu, dpdv, ro3, rxd, ryd, pa);
                                c = mipEval(sc, tx, uvx, uvy, pa[0] * tx.uscale, pa[1] * tx.vscale, pa[2] * tx.uscale, pa[3] * tx.vscale);
                            } else c = tx.filter != 0u ? mipBilinear(sc, tx, 0, uvx, uvy) : mipBox(sc, tx, 0, uvx, uvy);
                        } else c = textureEval(tx, huvx, huvy);
                        mm.reflectance[0] = c.x; mm.reflectance[1] = c.y; mm.reflectance[2] = c.z;
                    }
                }
                };
                applyTexture(bsdf);
                if (RC && bsdf.type == MI_BSDF_T_MASK) {                     // mask.cpp: this record's (textured) `reflectance` is the opacity in front of the nested record `distr`
                    opac = ld3(bsdf.reflectance); masked = true; bsdf = loadMaterial(tb, (int) bsdf.distr); applyTexture(bsdf);
                }
                if (depth == 1 && h.emitter >= 0 && !rc.hide_emitters) {    // path.cpp:148-150 (EEmittedRadiance only on the camera segment)
                    add = T * emitterEval(tb, h.emitter, h.ns, -d); haveAdd = true;
                }
                if ((depth >= rc.max_depth && rc.max_depth > 0) || (rc.strict_normals && dot(d, h.ng) * h.wi.z >= 0)) { pathLen += (unsigned) depth; break; }
                // emitter sampling (path.cpp:172-200)
                v3 refN = (h.flags & 2u) ? V(0, 0, 0) : h.ns;               // records.inl:160-164
                if (!(h.flags & 4u)) {                                       // bsdf->getType() & BSDF::ESmooth
                    float sx, sy; next2D(ss, rc.sampler, m32, sx, sy);
                    Direct dr; v3 value = sampleEmitterDirect<ENV, AN>(sc, tb, h.p, refN, sx, sy, dr);
                    if (dr.pdf != 0) {
                        ++shadowRays;                                        // scene.cpp:871-875: a shadow ray is cast whenever pdf != 0
                        v3 wo = toLocal(h, dr.d);
                        v3 bsdfVal = bsdfEval<RC>(sc, bsdf, h.wi, wo);
                        if (RC && masked) bsdfVal = bsdfVal * opac;                          // mask.cpp:124-127
                        if (!isZero(value) && !isZero(bsdfVal) && (!rc.strict_normals || dot(h.ng, dr.d) * wo.z > 0)) {
                            float bp = dr.delta ? 0.0f : bsdfPdf<RC>(sc, bsdf, h.wi, wo);     // emitter->isOnSurface() && measure == ESolidAngle (path.cpp:191-192)
                            if (RC && masked) bp *= luminance3(opac);                          // mask.cpp:141-146
                            float weight = miWeight(dr.pdf, bp);
                            v3 c = ((T * value) * bsdfVal) * weight;
                            wantShadow = true;
                            shO = make_float4(h.p.x, h.p.y, h.p.z, dr.dist * (1 - MI_SHADOW_EPSILON));
                            shD = make_float4(dr.d.x, dr.d.y, dr.d.z, __uint_as_float(pid));
                            shC = make_float4(c.x, c.y, c.z, 0.0f);
                        }
                    }
                }
                toSample = true;
            } while (false);
            if (haveAdd) { float4 a = q.acc[pid]; a.x += add.x; a.y += add.y; a.z += add.z; q.acc[pid] = a; }
        }
        // wave64 ballots: order-preserving compaction inside the (wave-owned) segment
        const unsigned long long mS = __ballot(wantShadow);
        if (wantShadow) {
            const uint64_t o = segBase + outS + (uint32_t) __popcll(mS & lt);
            q.shO[o] = shO; q.shD[o] = shD; q.shC[o] = shC;
        }
        outS += (uint32_t) __popcll(mS);
        float4 nrO, nrD, nS1; uint4 nS0; float nS2 = 0, nS3 = 0;
        if (toSample) {
            // BSDF sampling (path.cpp:207-226)
            float bPdf = 0, bEta = 1; v3 woL = V(0, 0, 0);
            float sx, sy; next2D(ss, rc.sampler, m32, sx, sy);
            bool sampledDelta, sampledNull; float extra = 0.0f; v3 bw;
            bool passThrough = false;                                                   // mask.cpp:196-208: the nested BSDF with probability luminance(opacity), else straight through
            if (RC && masked) { const float prob = luminance3(opac); if (sx < prob) sx /= prob; else passThrough = true; }
            if (RC && passThrough) {
                const float p = 1 - luminance3(opac);
                woL = V(-h.wi.x, -h.wi.y, -h.wi.z); bEta = 1.0f; bPdf = p; sampledDelta = true; sampledNull = true;
                bw = V((1.0f - opac.x) / p, (1.0f - opac.y) / p, (1.0f - opac.z) / p);
            } else {
                if (RC && bsdfUsesSampler(bsdf)) extra = next1D(ss, rc.sampler, m32);  // bRec.sampler->next1D() inside BSDF::sample (EUsesSampler)
                bw = bsdfSample<RC>(sc, bsdf, h.wi, sx, sy, extra, woL, bPdf, bEta, sampledDelta, sampledNull);
                if (RC && masked) { const float prob = luminance3(opac); bw = V(bw.x * opac.x / prob, bw.y * opac.y / prob, bw.z * opac.z / prob); bPdf *= prob; }
            }
            v3 wo = toWorld(h, woL);
            if (isZero(bw) || (rc.strict_normals && dot(h.ng, wo) * woL.z <= 0)) pathLen += (unsigned) depth;
            else {
                T = T * bw; eta *= bEta;
                alive = true;
                nrO = make_float4(h.p.x, h.p.y, h.p.z, MI_EPSILON);
                nrD = make_float4(wo.x, wo.y, wo.z, INFINITY);
                v3 refN = (h.flags & 2u) ? V(0, 0, 0) : h.ns;           // records.inl:160-164
                const float cosRef = dot(wo, refN);
                uint32_t fl = cosRef >= 0 ? 1u : 0u;
                nS3 = (h.flags & 2u) ? 2.0f : cosRef;
                nS0 = make_uint4(pid, ss.a, ss.b, (ss.dim & 0xFFu) | ((uint32_t) (depth + 1) << 8) | (fl << 16) | ((RC && sampledDelta) ? (1u << 17) : 0u) | ((RC && sampledNull && (depth == 1 || unscattered)) ? (1u << 18) : 0u));
                nS1 = make_float4(T.x, T.y, T.z, eta); nS2 = bPdf;
            }
        }
        const unsigned long long mA = __ballot(alive);
        if (alive) {
            const uint64_t o = segBase + outA + (uint32_t) __popcll(mA & lt);
            q.rayO[nb][o] = nrO; q.rayD[nb][o] = nrD; q.st0[nb][o] = nS0; q.st1[nb][o] = nS1; q.st2[nb][o] = nS2;
            if (ENV && sc.env_constant) q.st3[nb][o] = nS3;
        }
        outA += (uint32_t) __popcll(mA);
    }
    if (lane == 0) { q.count[nb][seg] = outA; q.shCount[seg] = outS; }
    }
    // counters: wave reduction, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) { pathLen += __shfl_down(pathLen, off); shadowRays += __shfl_down(shadowRays, off); }
    if (lane == 0) { if (pathLen) atomicAdd(&q.counters[2], pathLen); if (shadowRays) atomicAdd(&q.counters[1], shadowRays); }
}

// ---------------------------------------------------------------------------------------------- shadow
// Visibility test of Scene::sampleEmitterDirect (src/librender/scene.cpp:871-875 -> skdtree.cpp:207-226, any hit) and the
// deferred `Li += throughput * value * bsdfVal * weight` (path.cpp:196)
template <int STACK, int AN>
__global__ __launch_bounds__(WG) void k_shadow(DScene sc, Queues q) {
    __shared__ int s_stk[(STACK > 0 ? STACK : 1) * WG];
    const uint32_t tid = threadIdx.x;
    for (uint32_t seg = blockIdx.x; seg < q.n_seg; seg += gridDim.x) {
    const uint32_t n = q.shCount[seg];
    const uint64_t segBase = (uint64_t) seg * q.cap;
    for (uint32_t i = tid; i < n; i += WG) {
        float4 so = q.shO[segBase + i], sd = q.shD[segBase + i];
        v3 o = V(so.x, so.y, so.z), d = V(sd.x, sd.y, sd.z);
        float mint, maxt, t, u, v; uint32_t prim; bool occluded = false; int inst;
        if (clipInterval(sc, o, d, MI_EPSILON, so.w, true, mint, maxt)) {
            if (STACK == 0) occluded = packetIntersect<true, AN>(sc, o, d, mint, maxt, t, prim, u, v);
            else occluded = traverse<true, AN>(sc, o, d, mint, maxt, s_stk + tid, t, prim, u, v, inst);
        }
        if (!occluded) {
            float4 c = q.shC[segBase + i]; const uint32_t pid = __float_as_uint(sd.w);
            float4 a = q.acc[pid]; a.x += c.x; a.y += c.y; a.z += c.z; q.acc[pid] = a;
        }
    }
    }
}

// ---------------------------------------------------------------------------------------------- film
// ImageBlock::put (include/mitsuba/render/imageblock.h:161-221) of every sample of the batch, 5 channels (R,G,B,alpha,weight),
// film planes are SoA.  One thread per tile pixel walks its planes in sample order.  The part of a footprint that lands on
// the thread's own pixel is added to `film` with plain loads/stores, one sample after the other -- the same order of float
// additions as the reference's per-pixel loop, independent of batch size and tiling.  Anything that spills into another pixel
// (box filter: only samples within 1e-5 of a pixel edge; wider filters: most of the footprint) goes to the separate `spill`
// planes with float atomics; read-back returns film + spill.
__global__ __launch_bounds__(WG) void k_film(DScene sc, Queues q, BatchDesc bd, float *film, float *spill) {
    const uint32_t pl = blockIdx.x * WG + threadIdx.x;
    if (pl >= bd.n_pix) return;
    const uint32_t tw = bd.tile.x1 - bd.tile.x0;
    const int px = (int) (bd.tile.x0 + pl % tw), py = (int) (bd.tile.y0 + (pl / tw) * bd.row_stride);
    const int W = (int) sc.width + 2 * sc.border, H = (int) sc.height + 2 * sc.border;
    const size_t plane = (size_t) W * H;
    const int ownX = px + sc.border, ownY = py + sc.border;
    const size_t ownIdx = (size_t) ownY * W + ownX;
    float own[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) own[k] = film[k * plane + ownIdx];
    const float r = sc.filter_radius;
    for (uint32_t s = 0; s < bd.n_planes; ++s) {
        const uint64_t pid = (uint64_t) s * bd.n_pix + pl;
        float4 li = q.acc[pid]; float2 sp = q.pos[pid];
        float vals[5] = {li.x, li.y, li.z, li.w, 1.0f};
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 5; ++k) bad |= (!isfinite(vals[k]) || vals[k] < 0);
        if (bad) continue;
        float posx = sp.x - 0.5f - (float) (0 - sc.border), posy = sp.y - 0.5f - (float) (0 - sc.border);
        int minx = (int) ceilf(posx - r), miny = (int) ceilf(posy - r), maxx = (int) floorf(posx + r), maxy = (int) floorf(posy + r);
        minx = max(minx, 0); miny = max(miny, 0); maxx = min(maxx, W - 1); maxy = min(maxy, H - 1);
        for (int y = miny; y <= maxy; ++y) {
            float wy = filterEvalDiscretized(sc, (float) y - posy);
            for (int x = minx; x <= maxx; ++x) {
                float w = filterEvalDiscretized(sc, (float) x - posx) * wy;
                if (x == ownX && y == ownY) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) own[k] += w * vals[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 5; ++k) atomicAdd(&spill[k * plane + (size_t) y * W + x], w * vals[k]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) film[k * plane + ownIdx] = own[k];
}

// film read-back helpers: SoA planes -> interleaved layouts of mi_render_read_film
__global__ void k_film_layout(const float *film, const float *spill, float *out, int W, int H, int border, int layout) {
    const size_t plane = (size_t) W * H;
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (layout == 0) { if (i < plane) for (int k = 0; k < 5; ++k) out[i * 5 + k] = film[k * plane + i] + spill[k * plane + i]; }
    else if (layout == 1) { if (i < plane) for (int k = 0; k < 4; ++k) out[i * 4 + k] = film[k * plane + i] + spill[k * plane + i]; }
    else {
        const int w = W - 2 * border, h = H - 2 * border;
        if (i < (size_t) w * h) {
            const int x = (int) (i % w), y = (int) (i / w); const size_t src = (size_t) (y + border) * W + (x + border);
            const float wgt = film[4 * plane + src] + spill[4 * plane + src], inv = wgt != 0 ? 1.0f / wgt : 0.0f;
            for (int k = 0; k < 3; ++k) out[i * 3 + k] = (film[k * plane + src] + spill[k * plane + src]) * inv;
        }
    }
}

// ---------------------------------------------------------------------------------------------- parity / unit kernels
__global__ void k_gather_samples(Queues q, const uint32_t *slots, uint64_t n, float *out) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float4 a = q.acc[slots[i]]; out[i * 3] = a.x; out[i * 3 + 1] = a.y; out[i * 3 + 2] = a.z; }
}
__global__ __launch_bounds__(WG) void k_debug_intersect(DScene sc, const float *rays, uint64_t n, int anyHit, float *out, int *outInst) {
    __shared__ int s_stk[STACK_DEPTH * WG];
    const uint64_t i = (uint64_t) blockIdx.x * WG + threadIdx.x;
    if (i >= n) return;
    const float *r = rays + i * 8;
    v3 o = V(r[0], r[1], r[2]), d = V(r[4], r[5], r[6]);
    float mint, maxt, t = 0, u = 0, v = 0; uint32_t prim = 0xFFFFFFFFu; bool hit = false; int inst = -1;
    if (clipInterval(sc, o, d, r[3], r[7], anyHit != 0, mint, maxt)) {
        if (sc.packet_n) { if (anyHit) hit = packetIntersect<true, 3>(sc, o, d, mint, maxt, t, prim, u, v); else hit = packetIntersect<false, 3>(sc, o, d, mint, maxt, t, prim, u, v); }
        else if (anyHit) hit = traverse<true, 3>(sc, o, d, mint, maxt, s_stk + threadIdx.x, t, prim, u, v, inst);
        else hit = traverse<false, 3>(sc, o, d, mint, maxt, s_stk + threadIdx.x, t, prim, u, v, inst);
    }
    if (outInst) outInst[i] = hit ? inst : -1;
    out[i * 4] = t; out[i * 4 + 1] = u; out[i * 4 + 2] = v; out[i * 4 + 3] = hit ? (anyHit ? 1.0f : (float) prim) : -1.0f;
}
__global__ void k_debug_sobol(DScene sc, const uint32_t *in, uint64_t n, uint32_t ndims, unsigned long long *outIdx, float *outVals) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t idx = sc.log_res > 1 ? sobolLookUp(sc.sobol_vdc, sc.sobol_vdc_inv, sc.log_res, in[i * 3 + 2], in[i * 3], in[i * 3 + 1]) : in[i * 3 + 2];
    outIdx[i] = idx;
    for (uint32_t dmn = 0; dmn < ndims; ++dmn) outVals[i * ndims + dmn] = sobolSample(sc.sobol_m32, idx, dmn);
}
__global__ void k_debug_sincosf(const float *in, uint64_t n, float *out) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float s, c; glibcSincosf(in[i], s, c); out[i * 2] = s; out[i * 2 + 1] = c; }
}
__global__ void k_debug_camera(DScene sc, const float *pos, uint64_t n, float *out) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    v3 o, d; float mint, maxt; cameraRay(sc, pos[i * 2], pos[i * 2 + 1], o, d, mint, maxt);
    float *r = out + i * 8; r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = mint; r[4] = d.x; r[5] = d.y; r[6] = d.z; r[7] = maxt;
}

}  // namespace MI_NS
using namespace MI_NS;

// ---------------------------------------------------------------------------------------------- launch wrappers (used by api.cpp)
extern "C" {
void MI_FN(mi_launch_generate)(const DScene &sc, const RenderConst &rc, const Queues &q, const BatchDesc &bd, uint32_t grid, hipStream_t st) { hipLaunchKernelGGL(k_generate, dim3(grid), dim3(WG), 0, st, sc, rc, q, bd); }
void MI_FN(mi_upload_packet)(const TriAccelD *tris, uint32_t n, const AnalyticD *an, uint32_t na, hipStream_t st) {
    if (n) (void) hipMemcpyToSymbolAsync(HIP_SYMBOL(c_packet), tris, n * sizeof(TriAccelD), 0, hipMemcpyHostToDevice, st);
    if (na) (void) hipMemcpyToSymbolAsync(HIP_SYMBOL(c_analytic), an, na * sizeof(AnalyticD), 0, hipMemcpyHostToDevice, st);
}
static const bool kForceStack24 = getenv("MI355PT_STACK24") != nullptr;      // A/B switch, read once
#define MI_BY_STACK(KERNEL, AN, ...) do { \
    if (sc.packet_n) hipLaunchKernelGGL((KERNEL<0, AN>), dim3(grid), dim3(WG), 0, st, __VA_ARGS__); \
    else if (sc.bvh_depth <= 8) hipLaunchKernelGGL((KERNEL<8, AN>), dim3(grid), dim3(WG), 0, st, __VA_ARGS__); \
    else if (sc.bvh_depth <= 12) hipLaunchKernelGGL((KERNEL<12, AN>), dim3(grid), dim3(WG), 0, st, __VA_ARGS__); \
    else if (sc.bvh_depth <= 16) hipLaunchKernelGGL((KERNEL<16, AN>), dim3(grid), dim3(WG), 0, st, __VA_ARGS__); \
    else if (sc.bvh_depth <= 20 && !kForceStack24) hipLaunchKernelGGL((KERNEL<20, AN>), dim3(grid), dim3(WG), 0, st, __VA_ARGS__); \
    else if (sc.bvh_depth <= 24) hipLaunchKernelGGL((KERNEL<24, AN>), dim3(grid), dim3(WG), 0, st, __VA_ARGS__); \
    else if (sc.bvh_depth <= 28) hipLaunchKernelGGL((KERNEL<28, AN>), dim3(grid), dim3(WG), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<STACK_DEPTH, AN>), dim3(grid), dim3(WG), 0, st, __VA_ARGS__); } while (0)
void MI_FN(mi_launch_extend)(const DScene &sc, const Queues &q, int buf, uint32_t grid, hipStream_t st) {
    const int mode = (sc.n_analytic ? 1 : 0) | (sc.n_instances ? 2 : 0);
    if (mode == 3) MI_BY_STACK(k_extend, 3, sc, q, buf); else if (mode == 2) MI_BY_STACK(k_extend, 2, sc, q, buf); else if (mode == 1) MI_BY_STACK(k_extend, 1, sc, q, buf); else MI_BY_STACK(k_extend, 0, sc, q, buf);
}
void MI_FN(mi_launch_shade)(const DScene &sc, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, hipStream_t st) {
    size_t lds = rc.sampler == 1 ? (size_t) rc.nib_dims * rc.nib_count * 64 : 16;
    const bool env = sc.env_index >= 0, small = sc.small_tables != 0;
    if (small) lds += 16 + 4 * ((size_t) sc.n_tris * 24 + sc.n_materials * 16 + sc.n_emitters * 12 + ((sc.n_emitters + 4) & ~3u) + sc.area_cdf_len);
    RenderConst rcl = rc; rcl.order_offset_words = 0;
    if (sc.has_roughconductor && q.cap <= 8192u) {      // material-order list: only where it still fits the 64 KB a launch may request (else unsorted shading)
        const uint32_t off = (uint32_t) ((lds + 15) / 16 * 4); const size_t total = (size_t) off * 4 + (size_t) q.cap * 2 * (WG / 64) + 16;
        if (total <= 64 * 1024) { rcl.order_offset_words = off; lds = total; }
    }
#define MI_SHADE(RC, ENV, SM) do { if (sc.ext && sc.n_textures) hipLaunchKernelGGL((k_shade<RC, ENV, SM, true, true>), dim3(grid), dim3(WG), lds, st, sc, rcl, q, buf); \
                                   else if (sc.ext) hipLaunchKernelGGL((k_shade<RC, ENV, SM, true, false>), dim3(grid), dim3(WG), lds, st, sc, rcl, q, buf); \
                                   else hipLaunchKernelGGL((k_shade<RC, ENV, SM, false, false>), dim3(grid), dim3(WG), lds, st, sc, rcl, q, buf); } while (0)
    if (small) { if (sc.has_roughconductor) { if (env) MI_SHADE(true, true, true); else MI_SHADE(true, false, true); } else { if (env) MI_SHADE(false, true, true); else MI_SHADE(false, false, true); } }
    else { if (sc.has_roughconductor) { if (env) MI_SHADE(true, true, false); else MI_SHADE(true, false, false); } else { if (env) MI_SHADE(false, true, false); else MI_SHADE(false, false, false); } }
#undef MI_SHADE
}
void MI_FN(mi_launch_shadow)(const DScene &sc, const Queues &q, uint32_t grid, hipStream_t st) {
    const int mode = (sc.n_analytic ? 1 : 0) | (sc.n_instances ? 2 : 0);
    if (mode == 3) MI_BY_STACK(k_shadow, 3, sc, q); else if (mode == 2) MI_BY_STACK(k_shadow, 2, sc, q); else if (mode == 1) MI_BY_STACK(k_shadow, 1, sc, q); else MI_BY_STACK(k_shadow, 0, sc, q);
}
#undef MI_BY_STACK
void MI_FN(mi_launch_env_primary)(const DScene &sc, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, hipStream_t st) { hipLaunchKernelGGL(k_env_primary, dim3(grid), dim3(WG), 0, st, sc, rc, q, buf); }
void MI_FN(mi_launch_film)(const DScene &sc, const Queues &q, const BatchDesc &bd, float *film, float *spill, hipStream_t st) { hipLaunchKernelGGL(k_film, dim3((bd.n_pix + WG - 1) / WG), dim3(WG), 0, st, sc, q, bd, film, spill); }
#ifndef MI_FAST_MATH
void mi_launch_film_layout(const float *film, const float *spill, float *out, int W, int H, int border, int layout, hipStream_t st) {
    size_t n = (size_t) W * H; hipLaunchKernelGGL(k_film_layout, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, film, spill, out, W, H, border, layout);
}
void mi_launch_gather_samples(const Queues &q, const uint32_t *slots, uint64_t n, float *out, hipStream_t st) { hipLaunchKernelGGL(k_gather_samples, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, q, slots, n, out); }
void mi_launch_debug_intersect(const DScene &sc, const float *rays, uint64_t n, int anyHit, float *out, int *outInst, hipStream_t st) { hipLaunchKernelGGL(k_debug_intersect, dim3((unsigned) ((n + WG - 1) / WG)), dim3(WG), 0, st, sc, rays, n, anyHit, out, outInst); }
void mi_launch_debug_sobol(const DScene &sc, const uint32_t *in, uint64_t n, uint32_t ndims, unsigned long long *oi, float *ov, hipStream_t st) { hipLaunchKernelGGL(k_debug_sobol, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, sc, in, n, ndims, oi, ov); }
void mi_launch_debug_sincosf(const float *in, uint64_t n, float *out, hipStream_t st) { hipLaunchKernelGGL(k_debug_sincosf, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, in, n, out); }
void mi_launch_debug_camera(const DScene &sc, const float *pos, uint64_t n, float *out, hipStream_t st) { hipLaunchKernelGGL(k_debug_camera, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, sc, pos, n, out); }
#endif
}
