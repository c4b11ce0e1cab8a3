// trace.h -- ray casting behind Scene::rayIntersect: BVH2 traversal with a per-lane LDS stack, and the triangle packet of small scenes
#pragma once
#include "kernels_common.h"

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
// WIDE: 4-wide nodes with child boxes quantised to 8 bits per coordinate (pt_types.h Bvh4Node, 64 B like a BVH2 node): one node fetch decides four descents,
// the tree is half as deep -- the incoherent traversal of a large scene is bound by node traffic from L2 / Infinity Cache, and this halves it.  Children that
// are hit are visited nearest first (a 5-comparator sorting network on (entry distance | slot) keys); the quantised boxes only ever grow, so the exact
// triangle tests below see every triangle they saw before.
template <bool ANY, int AN, bool WIDE>        // AN bit 0: analytic shapes present, bit 1: instances present (each kernel variant carries only the code it needs)
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
    // BVH2: a stack entry is a child code.  WIDE: an entry stands for the children of ONE node that are still to be visited -- node index in bits 0..22, their
    // slot numbers (nearest first, 2 bits each) in bits 23..28, how many beyond the first in bits 29..30 -- so the stack never holds more than one entry per
    // tree level; popping re-reads the node's child pointers (16 B).  The markers BVH_DONE / BVH_RET carry 3 in bits 29..30, which no entry does.
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
        if (WIDE) {
            while (cur >= 0 && cur != BVH_DONE) {
                const f4 n0 = nodes4[cur * 4 + 0], n1 = nodes4[cur * 4 + 1], n2 = nodes4[cur * 4 + 2], n3 = nodes4[cur * 4 + 3];
                const float sx = n0.w, sy = n2.z, sz = n2.w;      // quantisation steps (powers of two)
                // plane distance t = (org + q s - o) inv = q (s inv) + (org inv - o inv)
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
                    key[c] = (tn <= tf * 1.000002f + 1e-30f) ? ((__float_as_uint(tn) & ~3u) | (uint32_t) c) : 0xFFFFFFFFu;      // tn >= mint > 0: its bit pattern orders like the value
                }
                {   // ascending (misses last): (0,1) (2,3) (0,2) (1,3) (1,2)
                    uint32_t a = min(key[0], key[1]), b = max(key[0], key[1]), c = min(key[2], key[3]), e = max(key[2], key[3]);
                    key[0] = min(a, c); const uint32_t m1 = max(a, c), m2 = min(b, e); key[3] = max(b, e); key[1] = min(m1, m2); key[2] = max(m1, m2);
                }
                const int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y), c2 = __float_as_int(n3.z), c3 = __float_as_int(n3.w);
                auto childOf = [&](uint32_t k) { const uint32_t sl = k & 3u; return sl == 0u ? c0 : (sl == 1u ? c1 : (sl == 2u ? c2 : c3)); };
                if (key[0] == 0xFFFFFFFFu) BVH_POP();
                else {
                    const uint32_t more = (key[1] != 0xFFFFFFFFu) + (key[2] != 0xFFFFFFFFu) + (key[3] != 0xFFFFFFFFu);
                    if (more) { stk[sp * WG] = (int) ((uint32_t) cur | ((key[1] & 3u) << 23) | ((key[2] & 3u) << 25) | ((key[3] & 3u) << 27) | ((more - 1u) << 29)); ++sp; }
                    cur = childOf(key[0]);
                }
            }
        } else
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

// Packet mode (scenes of <= MI_PACKET_MAX = 64 triangles): no tree.  Two passes per ray:
//   pass 1 (conservative, wave-uniform operands): one record per coplanar PAIR of triangles that form a parallelogram (or per single triangle) arrives through
//          the scalar cache as SGPR operands; an APPROXIMATE Wald test (fused multiply-adds, v_rcp_f32 instead of the IEEE division) with error margins that
//          cover every rounding difference to the exact test marks the triangles the ray may hit in a per-lane 64-bit candidate mask;
//   pass 2 (exact, per lane): each lane walks ITS candidates (typically 1 - 3), reads their exact Wald records from LDS and runs the reference's test
//          (TriAccel::rayIntersect, IEEE division and all) -- so (t, u, v, prim) are bit-identical to testing all triangles, at about half the vector
//          and scalar instructions (round 1: 32 exact tests with 32 IEEE divisions per ray; now ~16 approximate plane tests + ~2 exact ones).
// Margins (pass 1): with So = max(|scene box|, |ray origin|), r = 1 / D', q = |r| + 1 the distance t' is off by at most mt = q (1e-5 So + 2e-6 |t'|) (numerator
// and denominator carry <= 2e-6 So / 1e-6 absolute rounding error, the reciprocal 2e-7 relative), barycentrics by at most m = margin * mt with margin =
// 1.1 (|b_nu| + |b_nv| + |c_nu| + |c_nv|) of the record; rays within 1e-4 of parallel to a plane (|D'| < 1e-4, where the bound degenerates) make both triangles
// candidates.  Ties in t go to the lower ORIGINAL triangle index (candidates are walked in ascending order), as in round 1.
typedef const __attribute__((address_space(4))) f4 *cf4p;     // constant address space: a wave-uniform index makes these scalar loads (s_load_dwordx4)
struct PacketGroupS { f4 g0, g1, g2; };      // PacketGroupD as three 16-B words: n_u n_v n_d a_u | a_v b_nu b_nv c_nu | c_nv margin prim0 prim1
typedef const __attribute__((address_space(4))) PacketGroupS *cgrp;
template <typename MaskT> DEV MaskT maskBit(uint32_t p) { return p < sizeof(MaskT) * 8u ? (MaskT) 1 << p : (MaskT) 0; }      // uniform (scalar) value
template <int K, typename MaskT>
DEV void packetPass1(cgrp g, uint32_t count, v3 o, v3 d, float mint, float maxt, float so1e5, MaskT &mask) {
    const float o_u = K == 0 ? o.y : (K == 1 ? o.z : o.x), o_v = K == 0 ? o.z : (K == 1 ? o.x : o.y), o_k = K == 0 ? o.x : (K == 1 ? o.y : o.z);
    const float d_u = K == 0 ? d.y : (K == 1 ? d.z : d.x), d_v = K == 0 ? d.z : (K == 1 ? d.x : d.y), d_k = K == 0 ? d.x : (K == 1 ? d.y : d.z);
    for (; count; --count, ++g) {
        const f4 g0 = g->g0, g1 = g->g1, g2 = g->g2;
        const MaskT bit0 = maskBit<MaskT>(__float_as_uint(g2.z)), bit1 = maskBit<MaskT>(__float_as_uint(g2.w));
        const float D = __builtin_fmaf(d_u, g0.x, __builtin_fmaf(d_v, g0.y, d_k));
        const float N = __builtin_fmaf(-o_v, g0.y, __builtin_fmaf(-o_u, g0.x, g0.z - o_k));
        const float r = __builtin_amdgcn_rcpf(D), t = N * r, q = fabsf(r) + 1.0f;
        const float mt = q * __builtin_fmaf(fabsf(t), 2e-6f, so1e5);
        const float hu = __builtin_fmaf(t, d_u, o_u - g0.w), hv = __builtin_fmaf(t, d_v, o_v - g1.x);
        const float u = __builtin_fmaf(hv, g1.y, hu * g1.z), v = __builtin_fmaf(hu, g1.w, hv * g2.x);
        const float m = g2.y * mt, s = u + v, one = 1.0f + m;
        // reject = definitely outside (comparisons are false on NaN, so a NaN anywhere keeps the candidate); |D'| < 1e-4 (nearly parallel) keeps both
        const bool reject = (fabsf(D) >= 1e-4f) & ((t + mt < mint) | (t - mt > maxt) | (s < -m) | (s > one) | (v < -m) | (v > one));      // '&', '|': no short-circuit branches
        mask |= (reject | (u < -m)) ? (MaskT) 0 : bit0;
        mask |= (reject | (u > m)) ? (MaskT) 0 : bit1;
    }
}
// s_exact: the workgroup's LDS copy of the exact Wald records in ORIGINAL triangle order (3 x 16 B each)
template <bool ANY, int AN, typename MaskT>
DEV bool packetIntersectM(const DScene &sc, AS<true>::p4 s_exact, v3 o, v3 d, float mint, float maxt, float &bestT, uint32_t &bestPrim, float &bestU, float &bestV) {
    float best = maxt; uint32_t bprim = 0xFFFFFFFFu; bool found = false; float bu = 0, bv = 0;
    MaskT mask = 0;
    const float so1e5 = maxf(maxf(sc.packet_scale, fabsf(o.x)), maxf(fabsf(o.y), fabsf(o.z))) * 1e-5f;
    cgrp groups = (cgrp) sc.packet_groups;
    packetPass1<0, MaskT>(groups, sc.packet_gk[0], o, d, mint, maxt, so1e5, mask);
    packetPass1<1, MaskT>(groups + sc.packet_gk[0], sc.packet_gk[1] - sc.packet_gk[0], o, d, mint, maxt, so1e5, mask);
    packetPass1<2, MaskT>(groups + sc.packet_gk[1], sc.packet_gk[2] - sc.packet_gk[1], o, d, mint, maxt, so1e5, mask);
    while (mask) {      // pass 2, per lane: the reference's test (triaccel.h:96-158) on this lane's next candidate, written without branches (selects only)
        const uint32_t i = sizeof(MaskT) == 8 ? (uint32_t) __builtin_ctzll((unsigned long long) mask) : (uint32_t) __builtin_ctz((uint32_t) mask); mask &= mask - 1;
        const f4 a = s_exact[i * 3u], b = s_exact[i * 3u + 1u], c = s_exact[i * 3u + 2u];      // k n_u n_v n_d | a_u a_v b_nu b_nv | c_nu c_nv prim pad
        const bool k0 = __float_as_uint(a.x) == 0u, k1 = __float_as_uint(a.x) == 1u;
        const float o_u = k0 ? o.y : (k1 ? o.z : o.x), o_v = k0 ? o.z : (k1 ? o.x : o.y), o_k = k0 ? o.x : (k1 ? o.y : o.z);
        const float d_u = k0 ? d.y : (k1 ? d.z : d.x), d_v = k0 ? d.z : (k1 ? d.x : d.y), d_k = k0 ? d.x : (k1 ? d.y : d.z);
        const float tt = (a.w - o_u * a.y - o_v * a.z - o_k) / (d_u * a.y + d_v * a.z + d_k);
        const float hu = o_u + tt * d_u - b.x, hv = o_v + tt * d_v - b.y;
        const float uu = hv * b.z + hu * b.w, vv = hu * c.x + hv * c.y;
        const bool ok = !(tt < mint) & !(tt > best) & (uu >= 0) & (vv >= 0) & (uu + vv <= 1.0f);
        if (ANY) { found |= ok; mask = ok ? (MaskT) 0 : mask; }
        else {
            const bool better = ok & (!found | (tt < best));      // tt <= best here; equal t: the earlier (lower) index stays
            best = better ? tt : best; bprim = better ? i : bprim; bu = better ? uu : bu; bv = better ? vv : bv; found |= ok;
        }
    }
    if (ANY && found) return true;
    if (AN & 1) {
        for (uint32_t i = 0; i < sc.n_analytic; ++i) {
            AnalyticD sh;                                      // wave-uniform record: ten 16-B scalar loads
            { cf4p src = (cf4p) (sc.analytic + i); f4 *dst = reinterpret_cast<f4 *>(&sh);
#pragma unroll
              for (int j = 0; j < 10; ++j) dst[j] = src[j]; }
            float u, v, t; const uint32_t prim = sc.n_tris + i;
            if (analyticIntersect<ANY>(sh, o, d, mint, best, t, u, v)) {
                if (ANY) return true;
                if (!found || t < best || prim < bprim) { best = t; bprim = prim; bu = u; bv = v; found = true; }
            }
        }
    }
    bestT = best; bestPrim = bprim; bestU = bu; bestV = bv;
    return found;
}
template <bool ANY, int AN>
DEV bool packetIntersect(const DScene &sc, AS<true>::p4 s_exact, v3 o, v3 d, float mint, float maxt, float &bestT, uint32_t &bestPrim, float &bestU, float &bestV) {
    if (sc.n_tris <= 32u) return packetIntersectM<ANY, AN, uint32_t>(sc, s_exact, o, d, mint, maxt, bestT, bestPrim, bestU, bestV);      // uniform branch: 32-bit candidate masks
    return packetIntersectM<ANY, AN, unsigned long long>(sc, s_exact, o, d, mint, maxt, bestT, bestPrim, bestU, bestV);
}
// stage the exact Wald records of a packet-mode scene in LDS (called by every kernel that uses packetIntersect; ends with a barrier)
DEV void packetStage(const DScene &sc, f4 *s_exact) {
    const f4 *src = reinterpret_cast<const f4 *>(sc.packet_exact);
    for (uint32_t i = threadIdx.x; i < sc.n_tris * 3u; i += blockDim.x) s_exact[i] = src[i];
    __syncthreads();
}
