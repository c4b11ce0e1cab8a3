// scene_build.cpp -- host side of mi_scene_commit: Wald triangle table, per-triangle shading records, emitter CDFs,
// reconstruction-filter table and a binned-SAH BVH2 laid out for the gfx950 traversal kernel.
//
// Replaces (reference): ShapeKDTree::build + TriAccel::load (src/librender/skdtree.cpp:68-105,
// include/mitsuba/render/triaccel.h:61-94), Scene::initialize's emitter PDF (src/librender/scene.cpp:383-388),
// TriMesh::prepareSamplingTable (src/librender/trimesh.cpp:389-402), ReconstructionFilter::configure
// (src/libcore/rfilter.cpp:37-56).  The kd-tree itself is NOT reproduced: the contract is the nearest hit
// (t, prim, u, v), not the tree layout.  Compiled with -ffp-contract=off: the values computed
// here feed the bit-exact parity tests.
#include "scene_host.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace mi {

struct V3 { float x, y, z; };
static inline V3 mk(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline V3 normalize(V3 a) { float inv = 1.0f / std::sqrt(dot(a, a)); return a * inv; }
static inline V3 vmin(V3 a, V3 b) { return mk(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)); }
static inline V3 vmax(V3 a, V3 b) { return mk(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)); }

// reference include/mitsuba/render/triaccel.h:61-94
static void triaccelLoad(TriAccelD &ta, V3 A, V3 B, V3 C) {
    static const int wald[4] = {1, 2, 0, 1};
    V3 b = C - A, c = B - A, N = cross(c, b);
    int k = 0;
    for (int j = 0; j < 3; ++j) if (std::fabs(comp(N, j)) > std::fabs(comp(N, k))) k = j;
    int u = wald[k], v = wald[k + 1];
    float n_k = comp(N, k), denom = comp(b, u) * comp(c, v) - comp(b, v) * comp(c, u);
    std::memset(&ta, 0, sizeof(ta));
    if (denom == 0) { ta.k = 3; return; }
    ta.k = (uint32_t) k;
    ta.n_u = comp(N, u) / n_k; ta.n_v = comp(N, v) / n_k; ta.n_d = dot(A, N) / n_k;
    ta.b_nu = comp(b, u) / denom; ta.b_nv = -comp(b, v) / denom;
    ta.a_u = comp(A, u); ta.a_v = comp(A, v);
    ta.c_nu = comp(c, v) / denom; ta.c_nv = -comp(c, u) / denom;
}

// Analytic shapes: derived constants + Shape::getAABB (rectangle.cpp:100-119, disk.cpp:100-130, sphere.cpp:127-142, cylinder.cpp:105-107, :256-276)
static inline V3 xfPoint(const float *m, V3 p) { return mk(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7], m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]); }
static inline V3 xfVector(const float *m, V3 v) { return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z); }
static inline V3 xfNormal(const float *inv, V3 n) { return mk(inv[0] * n.x + inv[4] * n.y + inv[8] * n.z, inv[1] * n.x + inv[5] * n.y + inv[9] * n.z, inv[2] * n.x + inv[6] * n.y + inv[10] * n.z); }
static inline float length3(V3 a) { return std::sqrt(dot(a, a)); }
static void analyticPrepare(const mi_analytic &a, AnalyticD &d, V3 &lo, V3 &hi, V3 &tightLo, V3 &tightHi) {
    const float inf = std::numeric_limits<float>::infinity();
    std::memset(&d, 0, sizeof(d));
    std::memcpy(d.to_world, a.to_world, 48); std::memcpy(d.to_object, a.to_object, 48);
    d.type = a.type; d.radius = a.radius; d.length = a.length; d.material = a.bsdf; d.emitter = a.emitter;
    const float *M = a.to_world;
    lo = mk(inf, inf, inf); hi = mk(-inf, -inf, -inf);
    auto expand = [&](V3 q) { lo = vmin(lo, q); hi = vmax(hi, q); };
    V3 n = mk(0, 0, 0), dpdu = mk(0, 0, 0), center = mk(0, 0, 0);
    switch (a.type) {
    case MI_SHAPE_RECTANGLE: {
        dpdu = xfVector(M, mk(2, 0, 0)); V3 dpdv = xfVector(M, mk(0, 2, 0));
        n = normalize(xfNormal(a.to_object, mk(0, 0, 1)));
        d.inv_area = 1.0f / (length3(dpdu) * length3(dpdv));
        expand(xfPoint(M, mk(-1, -1, 0))); expand(xfPoint(M, mk(1, -1, 0))); expand(xfPoint(M, mk(1, 1, 0))); expand(xfPoint(M, mk(-1, 1, 0)));
        tightLo = lo; tightHi = hi; break;
    }
    case MI_SHAPE_DISK: {
        V3 du = xfVector(M, mk(1, 0, 0));
        n = normalize(xfNormal(a.to_object, mk(0, 0, 1)));
        d.inv_area = 1.0f / (MI_PI * length3(du) * length3(du));
        expand(xfPoint(M, mk(1, 0, 0))); expand(xfPoint(M, mk(-1, 0, 0))); expand(xfPoint(M, mk(0, 1, 0))); expand(xfPoint(M, mk(0, -1, 0)));
        // Disk::getAABB bounds four rim points only; the BVH needs the whole rim
        V3 c = xfPoint(M, mk(0, 0, 0)); float r = length3(du);
        tightLo = vmin(lo, c - mk(r, r, r)); tightHi = vmax(hi, c + mk(r, r, r)); break;
    }
    case MI_SHAPE_SPHERE: {
        center = xfPoint(M, mk(0, 0, 0));
        d.inv_area = 1 / (4 * MI_PI * a.radius * a.radius);
        lo = center - mk(a.radius, a.radius, a.radius); hi = center + mk(a.radius, a.radius, a.radius);
        tightLo = lo; tightHi = hi; break;
    }
    default: {
        d.inv_area = 1 / (2 * MI_PI * a.radius * a.length);
        V3 x1 = xfVector(M, mk(a.radius, 0, 0)), x2 = xfVector(M, mk(0, a.radius, 0));
        V3 p0 = xfPoint(M, mk(0, 0, 0)), p1 = xfPoint(M, mk(0, 0, a.length));
        float l[3], h[3];
        for (int i = 0; i < 3; ++i) {
            float range = std::sqrt(comp(x1, i) * comp(x1, i) + comp(x2, i) * comp(x2, i));
            l[i] = std::min(std::min(inf, comp(p0, i) - range), comp(p1, i) - range);
            h[i] = std::max(std::max(-inf, comp(p0, i) + range), comp(p1, i) + range);
        }
        lo = mk(l[0], l[1], l[2]); hi = mk(h[0], h[1], h[2]);
        tightLo = lo; tightHi = hi; break;
    }
    }
    d.n[0] = n.x; d.n[1] = n.y; d.n[2] = n.z; d.dpdu[0] = dpdu.x; d.dpdu[1] = dpdu.y; d.dpdu[2] = dpdu.z;
    d.center[0] = center.x; d.center[1] = center.y; d.center[2] = center.z;
}

struct BuildNode { V3 lo, hi; int left = -1, right = -1, first = 0, count = 0; };
struct Builder {
    std::vector<BuildNode> nodes; std::vector<uint32_t> order; const std::vector<V3> *tlo, *thi, *cen;
    const std::vector<uint8_t> *single = nullptr;   // primitives that must sit alone in their leaf (instances: the traversal enters them one at a time)
    int build(int first, int count, int depth) {
        int id = (int) nodes.size(); nodes.emplace_back();
        const float inf = std::numeric_limits<float>::infinity();
        V3 lo = mk(inf, inf, inf), hi = mk(-inf, -inf, -inf), clo = lo, chi = hi;
        for (int i = first; i < first + count; ++i) { uint32_t t = order[i]; lo = vmin(lo, (*tlo)[t]); hi = vmax(hi, (*thi)[t]); clo = vmin(clo, (*cen)[t]); chi = vmax(chi, (*cen)[t]); }
        nodes[id].lo = lo; nodes[id].hi = hi;
        auto makeLeaf = [&]() { nodes[id].first = first; nodes[id].count = count; return id; };
        bool mustSplit = false;
        if (single && count > 1) for (int i = first; i < first + count; ++i) mustSplit |= (*single)[order[i]] != 0;
        if (mustSplit) {              // keep splitting (object median) until the singleton primitives are alone
            V3 ce0 = chi - clo; int ax = ce0.x > ce0.y ? (ce0.x > ce0.z ? 0 : 2) : (ce0.y > ce0.z ? 1 : 2);
            std::sort(order.begin() + first, order.begin() + first + count, [&](uint32_t a, uint32_t b) { float x = comp((*cen)[a], ax), y = comp((*cen)[b], ax); return x < y || (x == y && a < b); });
            int m = first + count / 2;
            int l = build(first, m - first, depth + 1), r = build(m, first + count - m, depth + 1);
            nodes[id].left = l; nodes[id].right = r; return id;
        }
        if (count <= 2 || depth > 60) { if (count <= 8) return makeLeaf(); }
        // binned SAH, 16 bins per axis, all three centroid axes tried (round 1 / early round 2: the widest axis only; MI355PT_SAH1=1 restores that for A/B runs)
        static const bool widestOnly = [] { const char *e = getenv("MI355PT_SAH1"); return e && e[0] == '1'; }();
        V3 ce = chi - clo; const int widest = ce.x > ce.y ? (ce.x > ce.z ? 0 : 2) : (ce.y > ce.z ? 1 : 2);
        int mid = -1;
        {
            const int NB = 16; auto area = [](V3 l, V3 h) { V3 e = h - l; return 2.0f * (e.x * e.y + e.y * e.z + e.z * e.x); };
            float best = inf; int bestSplit = -1, bestAxis = -1;
            for (int axis = 0; axis < 3; ++axis) {
                if (widestOnly && axis != widest) continue;
                const float cmin = comp(clo, axis), cext = comp(ce, axis);
                if (!(cext > 0)) continue;
                int cnt[NB] = {0}; V3 blo[NB], bhi[NB];
                for (int b = 0; b < NB; ++b) { blo[b] = mk(inf, inf, inf); bhi[b] = mk(-inf, -inf, -inf); }
                auto binOf = [&](uint32_t t) { int b = (int) ((comp((*cen)[t], axis) - cmin) / cext * NB); return b < 0 ? 0 : (b >= NB ? NB - 1 : b); };
                for (int i = first; i < first + count; ++i) { uint32_t t = order[i]; int b = binOf(t); cnt[b]++; blo[b] = vmin(blo[b], (*tlo)[t]); bhi[b] = vmax(bhi[b], (*thi)[t]); }
                float rightArea[NB]; int rightCnt[NB]; V3 rl = mk(inf, inf, inf), rh = mk(-inf, -inf, -inf); int rc = 0;
                for (int b = NB - 1; b > 0; --b) { if (cnt[b]) { rl = vmin(rl, blo[b]); rh = vmax(rh, bhi[b]); } rc += cnt[b]; rightArea[b] = rc ? area(rl, rh) : 0; rightCnt[b] = rc; }
                V3 ll = mk(inf, inf, inf), lh = mk(-inf, -inf, -inf); int lc = 0;
                for (int b = 0; b < NB - 1; ++b) {
                    if (cnt[b]) { ll = vmin(ll, blo[b]); lh = vmax(lh, bhi[b]); } lc += cnt[b];
                    if (lc == 0 || rightCnt[b + 1] == 0) continue;
                    const float cost = area(ll, lh) * lc + rightArea[b + 1] * rightCnt[b + 1];
                    if (cost < best) { best = cost; bestSplit = b; bestAxis = axis; }
                }
            }
            const float leafCost = area(lo, hi) * count;
            if (bestSplit >= 0 && (count > 4 ? true : best + area(lo, hi) * 1.0f < leafCost)) {
                const int axis = bestAxis; const float cmin = comp(clo, axis), cext = comp(ce, axis);
                auto binOf = [&](uint32_t t) { int b = (int) ((comp((*cen)[t], axis) - cmin) / cext * NB); return b < 0 ? 0 : (b >= NB ? NB - 1 : b); };
                auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) { return binOf(t) <= bestSplit; });
                mid = (int) (it - order.begin());
            } else if (count <= 8 && (comp(ce, 0) > 0 || comp(ce, 1) > 0 || comp(ce, 2) > 0)) return makeLeaf();
        }
        if (mid <= first || mid >= first + count) {   // degenerate: all centroids equal -> split in the middle
            if (count <= 8) return makeLeaf();
            mid = first + count / 2;
        }
        int l = build(first, mid - first, depth + 1);
        int r = build(mid, first + count - mid, depth + 1);
        nodes[id].left = l; nodes[id].right = r;
        return id;
    }
};

static inline int32_t leafCode(int first, int count) { return ~(int32_t) (first * 8 + (count - 1)); }
#define BVH_EMPTY_CHILD 0x7FFFFFFD      // unused slot of a 4-wide node (its box is inverted, so it is never taken)

void SceneHost::commitHost() {
    const uint32_t nt = (uint32_t) (idx.size() / 3), na = (uint32_t) analytic.size(), ni = (uint32_t) instances.size(), np = nt + na + ni;
    nTris = nt;
    auto vert = [&](uint32_t i) { return mk(pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2]); };
    triShape.assign(nt, 0);
    for (uint32_t si = 0; si < shapes.size(); ++si) for (uint32_t t = 0; t < shapes[si].tri_count; ++t) triShape[shapes[si].first_tri + t] = si;

    // --- triangle records
    std::vector<TriAccelD> accel(nt); shade.assign(nt, TriShade{}); i2.assign(nt, 0);
    triuv.assign(uv.empty() ? 0 : nt, TriUV{}); anyUV = false;
    std::vector<V3> tlo(np), thi(np), cen(np);
    // flags of a shape's material as MIPathTracer::Li sees the (possibly nested) BSDF: bit1 EBackSide / ETransmission somewhere (dRec.refN = 0, records.inl:160-164),
    // bit2 no smooth component (no emitter sampling, path.cpp:173-174), bit3 anything but a plain diffuse record (class bit of the shading stage)
    auto leafBackside = [&](const mi_material &mat) { return (mat.flags & MI_BSDF_FLAG_TWOSIDED) != 0 || mat.type == MI_BSDF_DIELECTRIC || mat.type == MI_BSDF_ROUGHDIELECTRIC || mat.type == MI_BSDF_DIFFTRANS || mat.type == MI_BSDF_THINDIELECTRIC || mat.type == MI_BSDF_NULL; };
    // a `diffuse` with zero reflectance has no component at all -> not ESmooth -> Li skips emitter sampling (diffuse.cpp:99-102, path.cpp:174-176);
    // conductor / dielectric register delta components only
    auto leafSmooth = [&](const mi_material &mat) { return mat.type == MI_BSDF_DIFFUSE ? (((mat.flags >> 8) & 0xFFFFu) != 0 || std::max(std::max(mat.reflectance[0], mat.reflectance[1]), mat.reflectance[2]) > 0)
                                                                                         : (mat.type != MI_BSDF_CONDUCTOR && mat.type != MI_BSDF_DIELECTRIC && mat.type != MI_BSDF_THINDIELECTRIC && mat.type != MI_BSDF_NULL); };
    auto materialFlags = [&](int bsdf) {
        const mi_material *mat = &materials[bsdf]; bool masked = false, wrapped = false;
        if (mat->type == MI_BSDF_MASK) { masked = true; mat = &materials[mat->distr]; }          // mask.cpp:104-121: the nested BSDF's components + an ENull | EFrontSide | EBackSide one
        if (mat->type == MI_BSDF_BUMPMAP || mat->type == MI_BSDF_NORMALMAP) { wrapped = true; mat = &materials[mat->distr]; }     // the nested BSDF's component types (bumpmap.cpp:97-100)
        bool backside, smooth, coated = false;
        if (mat->type == MI_BSDF_COATING || mat->type == MI_BSDF_ROUGHCOATING) { coated = true; wrapped = true; mat = &materials[mat->distr]; }      // coating.cpp:166-173: the nested components + a delta reflection that is EFrontSide | EBackSide
        if (mat->type == MI_BSDF_BLEND) {                                                           // the two BSDFs' components (blendbsdf.cpp:103-134)
            backside = (mat->flags & MI_BSDF_FLAG_TWOSIDED) != 0; smooth = false; wrapped = true;
            for (int c = 0; c < 2; ++c) { const mi_material &ch = materials[(uint32_t) mat->eta[c]]; backside |= leafBackside(ch); smooth |= leafSmooth(ch); }
        } else if (mat->type == MI_BSDF_MIXTURE) {                                                         // the children's components (mixturebsdf.cpp:150-166)
            backside = (mat->flags & MI_BSDF_FLAG_TWOSIDED) != 0; smooth = false; wrapped = true;
            for (uint32_t c = 0; c < mat->distr; ++c) { const mi_material &ch = materials[(uint32_t) (c < 3 ? mat->reflectance[c] : mat->eta[0])]; backside |= leafBackside(ch); smooth |= leafSmooth(ch); }
        } else { backside = leafBackside(*mat); smooth = leafSmooth(*mat); }
        return ((backside || masked || coated) ? 2u : 0u) | (smooth ? 0u : 4u) | ((mat->type != MI_BSDF_DIFFUSE || masked || wrapped) ? 8u : 0u);
    };
    for (uint32_t t = 0; t < nt; ++t) {
        uint32_t a = idx[t * 3], b = idx[t * 3 + 1], c = idx[t * 3 + 2];
        V3 p0 = vert(a), p1 = vert(b), p2 = vert(c);
        triaccelLoad(accel[t], p0, p1, p2); accel[t].prim = t;
        const mi_shape &sh = shapes[triShape[t]];
        TriShade &ts = shade[t];
        ts.p0[0] = p0.x; ts.p0[1] = p0.y; ts.p0[2] = p0.z; ts.p1[0] = p1.x; ts.p1[1] = p1.y; ts.p1[2] = p1.z; ts.p2[0] = p2.x; ts.p2[1] = p2.y; ts.p2[2] = p2.z;
        ts.material = sh.bsdf; ts.emitter = sh.emitter;
        bool faceN = (sh.flags & 1u) || nrm.empty();
        const bool hasUV = (sh.flags & 2u) && !uv.empty();
        ts.flags = (faceN ? 1u : 0u) | materialFlags(sh.bsdf) | (hasUV ? 16u : 0u);
        ts.local_prim = t - sh.first_tri; ts.i0 = a; ts.i1 = b; i2[t] = c;
        // face frame: skdtree.h:367-371 (face normal), util.cpp:605-610 (computeShadingFrame with dpdu = p1 - p0)
        V3 side1 = p1 - p0, side2 = p2 - p0, fn = cross(side1, side2);
        float len = std::sqrt(dot(fn, fn));
        if (!(fn.x == 0 && fn.y == 0 && fn.z == 0)) { float r = 1.0f / len; fn = fn * r; }
        V3 dpdu = side1;
        if (hasUV) {                                             // TriMesh::computeUVTangents (trimesh.cpp:683-736)
            TriUV &tu = triuv[t]; anyUV = true;
            tu.uv0[0] = uv[a * 2]; tu.uv0[1] = uv[a * 2 + 1]; tu.uv1[0] = uv[b * 2]; tu.uv1[1] = uv[b * 2 + 1]; tu.uv2[0] = uv[c * 2]; tu.uv2[1] = uv[c * 2 + 1];
            float du1 = tu.uv1[0] - tu.uv0[0], dv1 = tu.uv1[1] - tu.uv0[1], du2 = tu.uv2[0] - tu.uv0[0], dv2 = tu.uv2[1] - tu.uv0[1];
            V3 n = cross(side1, side2); float length = std::sqrt(dot(n, n)); V3 tdu = mk(0, 0, 0), tdv = mk(0, 0, 0);
            if (length != 0) {
                float determinant = du1 * dv2 - dv1 * du2;
                if (determinant == 0) {                         // coordinateSystem(n / length, dpdu, dpdv), util.cpp:594-603
                    float r = 1.0f / length; V3 an = n * r;
                    if (std::fabs(an.x) > std::fabs(an.y)) { float invLen = 1.0f / std::sqrt(an.x * an.x + an.z * an.z); tdv = mk(an.z * invLen, 0.0f, -an.x * invLen); }
                    else { float invLen = 1.0f / std::sqrt(an.y * an.y + an.z * an.z); tdv = mk(0.0f, an.z * invLen, -an.y * invLen); }
                    tdu = cross(tdv, an);
                } else {
                    float invDet = 1.0f / determinant;
                    tdu = (side1 * dv2 - side2 * dv1) * invDet;
                    tdv = (side1 * (-du2) + side2 * du1) * invDet;
                }
            }
            tu.dpdu[0] = tdu.x; tu.dpdu[1] = tdu.y; tu.dpdu[2] = tdu.z; tu.dpdv[0] = tdv.x; tu.dpdv[1] = tdv.y; tu.dpdv[2] = tdv.z;
            dpdu = tdu;
        }
        V3 s = normalize(dpdu - fn * dot(fn, dpdu)), tt = cross(fn, s);
        ts.ng[0] = fn.x; ts.ng[1] = fn.y; ts.ng[2] = fn.z; ts.s[0] = s.x; ts.s[1] = s.y; ts.s[2] = s.z; ts.t[0] = tt.x; ts.t[1] = tt.y; ts.t[2] = tt.z;
        ts.i2 = c;
        if (!faceN) {      // a smooth triangle carries its three vertex normals instead of the (unused) face frame
            for (int k = 0; k < 3; ++k) { ts.s[k] = nrm[a * 3 + k]; ts.t[k] = nrm[b * 3 + k]; ts.n2[k] = nrm[c * 3 + k]; }
        }
        V3 lo = vmin(vmin(p0, p1), p2), hi = vmax(vmax(p0, p1), p2);
        // conservative padding: the Wald test is evaluated in its own arithmetic, boxes may only over-approximate
        V3 e = hi - lo; float mag = std::max(std::max(std::fabs(lo.x) + std::fabs(hi.x), std::fabs(lo.y) + std::fabs(hi.y)), std::fabs(lo.z) + std::fabs(hi.z));
        float pad = 1e-4f * std::max(std::max(e.x, e.y), e.z) + 2e-5f * mag + 1e-7f;
        tlo[t] = lo - mk(pad, pad, pad); thi[t] = hi + mk(pad, pad, pad); cen[t] = (lo + hi) * 0.5f;
    }
    // --- analytic shapes: device records, boxes
    analyticD.assign(na, AnalyticD{});
    std::vector<V3> alo(na), ahi(na);
    for (uint32_t i = 0; i < na; ++i) {
        V3 tl, th; analyticPrepare(analytic[i], analyticD[i], alo[i], ahi[i], tl, th);
        analyticD[i].flags = (analytic[i].flags & 1u) | materialFlags(analytic[i].bsdf);
        V3 e = th - tl; float mag = std::max(std::max(std::fabs(tl.x) + std::fabs(th.x), std::fabs(tl.y) + std::fabs(th.y)), std::fabs(tl.z) + std::fabs(th.z));
        float pad = 1e-4f * std::max(std::max(e.x, e.y), e.z) + 2e-5f * mag + 1e-7f;
        tlo[nt + i] = tl - mk(pad, pad, pad); thi[nt + i] = th + mk(pad, pad, pad); cen[nt + i] = (tl + th) * 0.5f;
        TriAccelD rec{}; rec.k = MI_K_ANALYTIC; rec.prim = nt + i; accel.push_back(rec);
    }
    // --- kd-tree boxes of the shape groups and of the scene = union of the member shapes' AABBs (ShapeKDTree::addShape, skdtree.cpp:68-77),
    //     enlarged like the kd-tree root (gkdtree.h:1213-1220); instance boxes = the 8 transformed corners of the group box (instance.cpp:46-64)
    const float inf = std::numeric_limits<float>::infinity();
    uint32_t ng = 0; for (const mi_shape &sh : shapes) ng = std::max(ng, sh.group);
    std::vector<V3> glo(ng + 1, mk(inf, inf, inf)), ghi(ng + 1, mk(-inf, -inf, -inf));      // slot ng = the scene level
    auto slotOf = [&](const mi_shape &sh) { return sh.group ? sh.group - 1 : ng; };
    auto enlarge = [](V3 &lo, V3 &hi) { const float eps = 1e-3f; V3 e1 = hi - lo; lo = lo - mk(e1.x * eps + eps, e1.y * eps + eps, e1.z * eps + eps);
                                        V3 e2 = hi - lo; hi = hi + mk(e2.x * eps + eps, e2.y * eps + eps, e2.z * eps + eps); };
    for (const mi_shape &sh : shapes) { uint32_t g = slotOf(sh); for (uint32_t v = 0; v < sh.vert_count; ++v) { V3 p = vert(sh.first_vert + v); glo[g] = vmin(glo[g], p); ghi[g] = vmax(ghi[g], p); } }
    for (uint32_t g = 0; g < ng; ++g) enlarge(glo[g], ghi[g]);
    for (uint32_t i = 0; i < na; ++i) { glo[ng] = vmin(glo[ng], alo[i]); ghi[ng] = vmax(ghi[ng], ahi[i]); }
    instancesD.assign(ni, InstanceD{});
    for (uint32_t i = 0; i < ni; ++i) {
        const mi_instance &in = instances[i]; const uint32_t g = in.group; V3 blo = mk(inf, inf, inf), bhi = mk(-inf, -inf, -inf);
        for (int c = 0; c < 8; ++c) {
            V3 q = xfPoint(in.to_world, mk(c & 1 ? ghi[g].x : glo[g].x, c & 2 ? ghi[g].y : glo[g].y, c & 4 ? ghi[g].z : glo[g].z));
            blo = vmin(blo, q); bhi = vmax(bhi, q);
        }
        glo[ng] = vmin(glo[ng], blo); ghi[ng] = vmax(ghi[ng], bhi);
        V3 e = bhi - blo; float mag = std::max(std::max(std::fabs(blo.x) + std::fabs(bhi.x), std::fabs(blo.y) + std::fabs(bhi.y)), std::fabs(blo.z) + std::fabs(bhi.z));
        float pad = 1e-4f * std::max(std::max(e.x, e.y), e.z) + 2e-5f * mag + 1e-7f;
        tlo[nt + na + i] = blo - mk(pad, pad, pad); thi[nt + na + i] = bhi + mk(pad, pad, pad); cen[nt + na + i] = (blo + bhi) * 0.5f;
        TriAccelD rec{}; rec.k = MI_K_INSTANCE; rec.prim = i; accel.push_back(rec);
        InstanceD &d = instancesD[i]; std::memcpy(d.to_world, in.to_world, 48); std::memcpy(d.to_object, in.to_object, 48);
        d.glo[0] = glo[g].x; d.glo[1] = glo[g].y; d.glo[2] = glo[g].z; d.ghi[0] = ghi[g].x; d.ghi[1] = ghi[g].y; d.ghi[2] = ghi[g].z; d.group = g; d.root = 0;
    }
    enlarge(glo[ng], ghi[ng]);
    aabbLo[0] = glo[ng].x; aabbLo[1] = glo[ng].y; aabbLo[2] = glo[ng].z; aabbHi[0] = ghi[ng].x; aabbHi[1] = ghi[ng].y; aabbHi[2] = ghi[ng].z;

    // --- participating media: device records; per primitive the (interior, exterior) pair of its shape (triangles, then analytic shapes)
    mediaD.assign(media.size(), MediumD{});
    for (size_t i = 0; i < media.size(); ++i) {
        MediumD &m = mediaD[i]; const mi_medium &in = media[i];
        for (int c = 0; c < 3; ++c) { m.sigma_s[c] = in.sigma_s[c]; m.sigma_t[c] = in.sigma_a[c] + in.sigma_s[c]; }     // Medium: m_sigmaT = m_sigmaA + m_sigmaS (medium.cpp:36)
        m.strategy = in.strategy; m.phase = in.phase; m.sampling_density = in.sampling_density; m.medium_sampling_weight = in.medium_sampling_weight; m.g = in.g;
    }
    primMedia.clear();
    if (!media.empty()) {
        primMedia.assign(nt + na, 0u);
        auto pairOf = [&](size_t shapeIndex) { return (uint32_t) (shapeMedia[shapeIndex * 2] + 1) | ((uint32_t) (shapeMedia[shapeIndex * 2 + 1] + 1) << 16); };
        for (uint32_t t = 0; t < nt; ++t) primMedia[t] = pairOf(triShape[t]);
        for (uint32_t i = 0; i < na; ++i) primMedia[nt + i] = pairOf(shapes.size() + i);
    }
    // --- BVHs: one per shape group, then the scene level (root = node 0 of its own range; the scene level is built LAST but must be node 0,
    //     so its nodes are emitted first and the groups appended)
    std::vector<uint8_t> single(np, 0); for (uint32_t i = 0; i < ni; ++i) single[nt + na + i] = 1;
    auto setBox = [](float *lo, float *hi, const BuildNode &n) { lo[0] = n.lo.x; lo[1] = n.lo.y; lo[2] = n.lo.z; hi[0] = n.hi.x; hi[1] = n.hi.y; hi[2] = n.hi.z; };
    auto emptyBox = [](float *lo, float *hi) { for (int i = 0; i < 3; ++i) { lo[i] = std::numeric_limits<float>::infinity(); hi[i] = -std::numeric_limits<float>::infinity(); } };
    nodes.clear(); tris.clear();
    std::vector<int> depthOf;     // depth of each emitted tree
    // builds the tree over `prims`, appends its nodes / leaf records, returns the device index of its root (always an inner node)
    // Node kind (measured, DESIGN.md "Tree scenes"): 4-wide quantised nodes win once the tree outgrows the caches close to the CUs (atrium, 0.6 M triangles:
    // +12 %); on small trees and on the two-level trees of instanced scenes the binary nodes' cheaper per-node arithmetic wins (instanced garden: +5 %).
    // MI355PT_BVH2 = 1 / 0 forces binary / wide (A/B runs, parity tests of both kinds).
    { const char *e2 = getenv("MI355PT_BVH2"); wideBvh = e2 && e2[0] ? e2[0] == '0' : (ni == 0 && nt >= 16384u); }
    std::vector<int> treeNeed, treeNeedDirect;    // traversal stack entries each emitted tree can need (one entry per level / child codes pushed one by one)
    auto emitTree = [&](const std::vector<uint32_t> &prims) -> int {
        Builder bld; bld.order = prims; bld.tlo = &tlo; bld.thi = &thi; bld.cen = &cen; bld.single = &single; bld.nodes.reserve(2 * prims.size() + 2);
        const int nodeBase = (int) nodes.size(), triBase = (int) tris.size();
        int root = prims.empty() ? -1 : bld.build(0, (int) prims.size(), 0);
        for (uint32_t i = 0; i < prims.size(); ++i) tris.push_back(accel[bld.order[i]]);
        if (wideBvh) {
            // Collapse the binary tree into 4-wide nodes (the inner child with the largest surface area is replaced by its two children until four slots are
            // taken), quantise the child boxes against the node's own box.  Returns the device index; `need` = stack entries from this node down.
            static_assert(sizeof(Bvh4Node) == sizeof(BvhNode), "both node kinds share one array");
            auto area = [](const BuildNode &n) { V3 e = n.hi - n.lo; return e.x * e.y + e.y * e.z + e.z * e.x; };
            struct Emit { std::vector<BvhNode> &nodes; Builder &bld; int triBase; decltype(area) &areaOf;
                int run(int id, int &need, int &needD) {
                    std::vector<int> kids;
                    if (id < 0) { /* empty tree */ } else if (bld.nodes[id].count > 0) kids.push_back(id); else { kids.push_back(bld.nodes[id].left); kids.push_back(bld.nodes[id].right); }
                    while (kids.size() < 4) {
                        int pick = -1; float best = -1;
                        for (size_t i = 0; i < kids.size(); ++i) if (bld.nodes[kids[i]].count == 0 && areaOf(bld.nodes[kids[i]]) > best) { best = areaOf(bld.nodes[kids[i]]); pick = (int) i; }
                        if (pick < 0) break;
                        const int k = kids[pick]; kids[pick] = bld.nodes[k].left; kids.insert(kids.begin() + pick + 1, bld.nodes[k].right);
                    }
                    const int dev = (int) nodes.size(); nodes.emplace_back();
                    const float inf = std::numeric_limits<float>::infinity(); V3 lo = mk(inf, inf, inf), hi = mk(-inf, -inf, -inf);
                    for (int k : kids) { lo = vmin(lo, bld.nodes[k].lo); hi = vmax(hi, bld.nodes[k].hi); }
                    if (kids.empty()) { lo = mk(0, 0, 0); hi = mk(0, 0, 0); }
                    Bvh4Node w; std::memset(&w, 0, sizeof(w)); w.org[0] = lo.x; w.org[1] = lo.y; w.org[2] = lo.z;
                    int ex[3];
                    for (int a = 0; a < 3; ++a) {      // step 2^e with 255 steps covering the extent (+ one step of slack for the rounding of org + q * step)
                        const float ext = comp(hi, a) - comp(lo, a); int e = 0; std::frexp(ext > 0 ? ext / 254.0f : 1e-30f, &e);      // ext / 254 = m 2^e, m in [0.5, 1): 2^e >= ext / 254
                        e = std::min(std::max(e + 127, 1), 254); ex[a] = e; (a == 0 ? w.step_x : (a == 1 ? w.step_y : w.step_z)) = std::ldexp(1.0f, e - 127);
                    }
                    int sub = 0, subD = 0;
                    for (int c = 0; c < 4; ++c) {
                        if (c >= (int) kids.size()) { for (int a = 0; a < 3; ++a) { w.qlo[a] |= 255u << (8 * c); } w.child[c] = BVH_EMPTY_CHILD; continue; }
                        const BuildNode &k = bld.nodes[kids[c]];
                        for (int a = 0; a < 3; ++a) {
                            const double step = std::ldexp(1.0, ex[a] - 127), o = comp(lo, a);
                            int ql = (int) std::floor(((double) comp(k.lo, a) - o) / step), qh = (int) std::ceil(((double) comp(k.hi, a) - o) / step);
                            ql = std::min(std::max(ql, 0), 255); qh = std::min(std::max(qh, 0), 255);
                            while (ql > 0 && (float) ((float) o + (float) ql * (float) step) > comp(k.lo, a)) --ql;          // the float reconstruction must enclose the child box
                            while (qh < 255 && (float) ((float) o + (float) qh * (float) step) < comp(k.hi, a)) ++qh;
                            w.qlo[a] |= (uint32_t) ql << (8 * c); w.qhi[a] |= (uint32_t) qh << (8 * c);
                        }
                        if (k.count > 0) w.child[c] = leafCode(triBase + k.first, k.count);
                        else { int need = 0, needD = 0; w.child[c] = run(kids[c], need, needD); sub = std::max(sub, need); subD = std::max(subD, needD); }
                    }
                    std::memcpy(&nodes[dev], &w, sizeof(w));
                    need = sub + (kids.size() > 1 ? 1 : 0);      // one stack entry per level: the node's pending children (trace.h)
                    needD = subD + (kids.size() > 1 ? (int) kids.size() - 1 : 0);      // child codes pushed one by one (trace_fused.h): up to kids - 1 siblings wait while a subtree is walked
                    return dev;
                } };
            Emit em{nodes, bld, triBase, area}; int need = 0, needD = 0; const int dev = em.run(root, need, needD); treeNeed.push_back(need + 1); treeNeedDirect.push_back(needD + 1);
            (void) nodeBase; return dev;
        }
        std::vector<int> devIndex(bld.nodes.size(), -1); int nInner = 0;
        for (size_t i = 0; i < bld.nodes.size(); ++i) if (bld.nodes[i].count == 0) devIndex[i] = nodeBase + nInner++;
        auto childCode = [&](int c) { const BuildNode &n = bld.nodes[c]; return n.count > 0 ? leafCode(triBase + n.first, n.count) : (int32_t) devIndex[c]; };
        if (root < 0) { BvhNode n{}; emptyBox(n.lo0, n.hi0); emptyBox(n.lo1, n.hi1); n.c0 = n.c1 = leafCode(0, 1); nodes.push_back(n); return nodeBase; }
        if (bld.nodes[root].count > 0) {      // the whole tree is one leaf: synthesise a root with one empty child
            BvhNode n{}; setBox(n.lo0, n.hi0, bld.nodes[root]); n.c0 = leafCode(triBase + bld.nodes[root].first, bld.nodes[root].count);
            emptyBox(n.lo1, n.hi1); n.c1 = n.c0; nodes.push_back(n); return nodeBase;
        }
        nodes.resize(nodeBase + nInner);
        for (size_t i = 0; i < bld.nodes.size(); ++i) {
            const BuildNode &b = bld.nodes[i]; if (b.count > 0) continue;
            BvhNode &n = nodes[devIndex[i]]; std::memset(&n, 0, sizeof(n));
            setBox(n.lo0, n.hi0, bld.nodes[b.left]); setBox(n.lo1, n.hi1, bld.nodes[b.right]);
            n.c0 = childCode(b.left); n.c1 = childCode(b.right);
        }
        return devIndex[root];
    };
    std::vector<std::vector<uint32_t> > members(ng + 1);
    for (const mi_shape &sh : shapes) { uint32_t g = slotOf(sh); for (uint32_t t = 0; t < sh.tri_count; ++t) members[g].push_back(sh.first_tri + t); }
    for (uint32_t t = nt; t < np; ++t) members[ng].push_back(t);
    emitTree(members[ng]);                 // the scene level first: a tree's root is the first node it emits, so the scene root is node 0
    // Hot nodes first (wide scene-level tree without instances): the fused walk (trace_fused.h) keeps the first nodes of the array in LDS, and a node is visited about
    // as often as its box is large (surface area heuristic) -- on the atrium the 128 most visited of 62 k nodes take 64 % of all node visits.  A pure renumbering
    // (root stays node 0): every traversal sees the same tree.
    if (wideBvh && ni == 0 && ng == 0 && nodes.size() > 1) {
        Bvh4Node *w = reinterpret_cast<Bvh4Node *>(nodes.data()); const size_t nn = nodes.size();
        std::vector<float> areaOf(nn, 0.0f); areaOf[0] = std::numeric_limits<float>::infinity();
        for (size_t i = 0; i < nn; ++i) for (int c = 0; c < 4; ++c) if (w[i].child[c] >= 0 && w[i].child[c] != BVH_EMPTY_CHILD) {
            const float st[3] = {w[i].step_x, w[i].step_y, w[i].step_z}; float e[3];
            for (int a = 0; a < 3; ++a) e[a] = (float) ((int) ((w[i].qhi[a] >> (8 * c)) & 0xFFu) - (int) ((w[i].qlo[a] >> (8 * c)) & 0xFFu)) * st[a];
            areaOf[w[i].child[c]] = e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
        }
        std::vector<uint32_t> byArea(nn); for (size_t i = 0; i < nn; ++i) byArea[i] = (uint32_t) i;
        std::stable_sort(byArea.begin(), byArea.end(), [&](uint32_t a, uint32_t b) { return areaOf[a] > areaOf[b]; });
        std::vector<int32_t> newIndex(nn); for (size_t i = 0; i < nn; ++i) newIndex[byArea[i]] = (int32_t) i;
        std::vector<BvhNode> re(nn);
        for (size_t i = 0; i < nn; ++i) {
            Bvh4Node n = w[byArea[i]];
            for (int c = 0; c < 4; ++c) if (n.child[c] >= 0 && n.child[c] != BVH_EMPTY_CHILD) n.child[c] = newIndex[n.child[c]];
            std::memcpy(&re[i], &n, sizeof(n));
        }
        nodes.swap(re);
    }
    std::vector<int> groupRoot(ng, 0);
    for (uint32_t g = 0; g < ng; ++g) groupRoot[g] = emitTree(members[g]);
    for (uint32_t i = 0; i < ni; ++i) instancesD[i].root = groupRoot[instances[i].group];
    // traversal stack need: scene tree + one return marker + the deepest group tree
    if (wideBvh) {      // treeNeed[0]: the scene level, then one entry per group
        int groupNeed = 0; for (uint32_t g = 0; g < ng; ++g) groupNeed = std::max(groupNeed, treeNeed[1 + g]);
        bvhDepth = treeNeed[0] + (ni ? 1 + groupNeed : 0);
        bvhStackDirect = treeNeedDirect[0];
    } else {
        struct Depth { const std::vector<BvhNode> &n; int of(int i) const { if (i < 0) return 0; int a = of(n[i].c0), b = of(n[i].c1); return 1 + (a > b ? a : b); } } dep{nodes};
        int groupDepth = 0; for (uint32_t g = 0; g < ng; ++g) groupDepth = std::max(groupDepth, dep.of(groupRoot[g]));
        bvhDepth = dep.of(0) + (ni ? 1 + groupDepth : 0);
        bvhStackDirect = dep.of(0);
    }
    // packet mode (no instances, <= MI_PACKET_MAX triangles): exact records in original order + pass-1 group records (pt_types.h PacketGroupD).  Coplanar
    // pairs that form a parallelogram (the two halves of a quad) share one record: for the vertices (X, Y, Z) of a triangle, taken cyclically, the partner is
    // the triangle on {Y, Z, Y + Z - X}.  Degenerate triangles (k = 3) never hit and are dropped.
    packetExact.assign(accel.begin(), accel.begin() + nt); packetGroups.clear(); packetGK[0] = packetGK[1] = packetGK[2] = 0;
    {
        float sx = 0; for (int i = 0; i < 3; ++i) sx = std::max(sx, std::max(std::fabs(aabbLo[i]), std::fabs(aabbHi[i])));
        packetScale = sx;
        if (nt <= MI_PACKET_MAX && ni == 0) {
            const float tol = 1e-6f * std::max(sx, 1e-20f);
            auto near = [&](V3 a, V3 b) { return std::fabs(a.x - b.x) <= tol && std::fabs(a.y - b.y) <= tol && std::fabs(a.z - b.z) <= tol; };
            std::vector<uint8_t> used(nt, 0); std::vector<PacketGroupD> byAxis[3];
            for (uint32_t t1 = 0; t1 < nt; ++t1) {
                if (used[t1] || accel[t1].k > 2) continue;
                used[t1] = 1;
                const V3 P[3] = {vert(idx[t1 * 3]), vert(idx[t1 * 3 + 1]), vert(idx[t1 * 3 + 2])};
                int partner = -1, rot = 0;
                for (uint32_t t2 = t1 + 1; t2 < nt && partner < 0; ++t2) {
                    if (used[t2] || accel[t2].k != accel[t1].k) continue;
                    const V3 Q[3] = {vert(idx[t2 * 3]), vert(idx[t2 * 3 + 1]), vert(idx[t2 * 3 + 2])};
                    for (int r = 0; r < 3 && partner < 0; ++r) {
                        const V3 X = P[r], Y = P[(r + 1) % 3], Z = P[(r + 2) % 3], X2 = Y + Z - X;
                        for (int a = 0; a < 3 && partner < 0; ++a)      // t2's vertex set == {Y, Z, X2} in any order
                            for (int b = 0; b < 3 && partner < 0; ++b) { if (b == a) continue; const int c = 3 - a - b;
                                if (near(Q[a], Y) && near(Q[b], Z) && near(Q[c], X2)) { partner = (int) t2; rot = r; } }
                    }
                }
                // relabel (X, Y, Z) -> (A*, B*, C*) = (Z, X, Y): a cyclic rotation (same plane, same orientation), shared edge YZ = C*A* <-> u* = 0
                const V3 X = P[rot], Y = P[(rot + 1) % 3], Z = P[(rot + 2) % 3];
                TriAccelD ta; triaccelLoad(ta, partner >= 0 ? Z : P[0], partner >= 0 ? X : P[1], partner >= 0 ? Y : P[2]);
                if (ta.k > 2) continue;
                if (partner >= 0) used[partner] = 1;
                PacketGroupD g; g.n_u = ta.n_u; g.n_v = ta.n_v; g.n_d = ta.n_d; g.a_u = ta.a_u; g.a_v = ta.a_v; g.b_nu = ta.b_nu; g.b_nv = ta.b_nv; g.c_nu = ta.c_nu; g.c_nv = ta.c_nv;
                g.margin = 1.1f * (std::fabs(ta.b_nu) + std::fabs(ta.b_nv) + std::fabs(ta.c_nu) + std::fabs(ta.c_nv));
                g.prim0 = t1; g.prim1 = partner >= 0 ? (uint32_t) partner : 0xFFFFFFFFu;
                byAxis[ta.k].push_back(g);
            }
            for (int axis = 0; axis < 3; ++axis) { packetGroups.insert(packetGroups.end(), byAxis[axis].begin(), byAxis[axis].end()); packetGK[axis] = (uint32_t) packetGroups.size(); }
        }
    }
    if (packetGroups.empty()) packetGroups.push_back(PacketGroupD{});
    if (packetExact.empty()) packetExact.push_back(TriAccelD{});

    // --- emitters (scene.cpp:383-388; pmf.h:56-58,103-116; trimesh.cpp:389-402; triangle.cpp:61-67)
    const uint32_t ne = (uint32_t) emitters.size();
    emitterCdf.assign(ne + 1, 0.0f); emittersD.assign(ne, EmitterD{}); areaCdf.clear(); emitterNorm = 0.0f;
    emitterX.assign((size_t) std::max(ne, 1u) * 16, 0.0f); hasDeltaEmitters = false;
    for (uint32_t e = 0; e < ne; ++e) emitterCdf[e + 1] = emitterCdf[e] + emitters[e].weight;
    if (ne) { float sum = emitterCdf[ne]; emitterNorm = sum > 0 ? 1.0f / sum : 0.0f; for (uint32_t e = 1; e <= ne; ++e) emitterCdf[e] *= emitterNorm; emitterCdf[ne] = 1.0f; }
    for (uint32_t e = 0; e < ne; ++e) {
        EmitterD &d = emittersD[e]; const mi_emitter &src = emitters[e];
        d.radiance[0] = src.radiance[0]; d.radiance[1] = src.radiance[1]; d.radiance[2] = src.radiance[2]; d.weight = src.weight;
        d.type = src.type; d.shape = src.shape;
        d.analytic = -1;
        float *x = &emitterX[e * 16];
        if (src.type == MI_EMITTER_POINT || src.type == MI_EMITTER_SPOT) { x[0] = src.to_world[3]; x[1] = src.to_world[7]; x[2] = src.to_world[11]; hasDeltaEmitters = true; }
        if (src.type == MI_EMITTER_COLLIMATED) { x[0] = src.to_world[3]; x[1] = src.to_world[7]; x[2] = src.to_world[11]; hasDeltaEmitters = true; }      // (selects the kernel variants whose sampleEmitterDirect knows emitter types >= 2)
        if (src.type == MI_EMITTER_DIRECTIONAL) { x[0] = src.to_world[2]; x[1] = src.to_world[6]; x[2] = src.to_world[10]; hasDeltaEmitters = true; }
        if (src.type == MI_EMITTER_SPOT) {                        // SpotEmitter constructor + configure (spot.cpp:70-96); trafo.inverse() of the rigid toWorld
            float beam = src.beam * (MI_PI / 180.0f), cutoff = src.cutoff * (MI_PI / 180.0f);
            x[13] = std::cos(beam); x[3] = std::cos(cutoff); x[14] = cutoff; x[15] = 1.0f / (cutoff - beam);
            const float *m = src.to_world; float *o = x + 4;
            float a[9] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]};
            float det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]); float id = 1.0f / det;
            o[0] = (a[4] * a[8] - a[5] * a[7]) * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
            o[3] = (a[5] * a[6] - a[3] * a[8]) * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
            o[6] = (a[3] * a[7] - a[4] * a[6]) * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
        }
        if (src.type != MI_EMITTER_AREA) continue;
        if ((size_t) src.shape >= shapes.size()) {               // area light on an analytic shape: no triangle CDF
            d.analytic = src.shape - (int32_t) shapes.size(); d.inv_area = analyticD[d.analytic].inv_area; continue;
        }
        const mi_shape &sh = shapes[src.shape];
        d.first_tri = sh.first_tri; d.tri_count = sh.tri_count; d.cdf_offset = (uint32_t) areaCdf.size();
        size_t base = areaCdf.size(); areaCdf.resize(base + sh.tri_count + 1); areaCdf[base] = 0.0f;
        for (uint32_t t = 0; t < sh.tri_count; ++t) {
            uint32_t prim = sh.first_tri + t;
            V3 p0 = vert(idx[prim * 3]), p1 = vert(idx[prim * 3 + 1]), p2 = vert(idx[prim * 3 + 2]);
            V3 c = cross(p1 - p0, p2 - p0);
            areaCdf[base + t + 1] = areaCdf[base + t] + 0.5f * std::sqrt(dot(c, c));
        }
        float sum = areaCdf[base + sh.tri_count], norm = 1.0f / sum;
        for (uint32_t t = 1; t <= sh.tri_count; ++t) areaCdf[base + t] *= norm;
        areaCdf[base + sh.tri_count] = 1.0f;
        d.inv_area = 1.0f / sum;
    }
    if (areaCdf.empty()) areaCdf.push_back(0.0f);

    /* --- reconstruction filter table (src/libcore/rfilter.cpp:37-56; eval of src/rfilters/box.cpp:31-48, gaussian.cpp:30-57, tent.cpp:36-38,
     * mitchell.cpp:49-61, catmullrom.cpp:36-49, lanczos.cpp:36-46).  filter_radius / filter_stddev carry: box radius; gaussian stddev (in
     * filter_stddev); mitchell B, C; lanczos lobes (in filter_radius) */
    {
        const uint32_t kind = filterKind;
        float radius = kind == 0 ? filterRadius + 1e-5f : kind == 1 ? 4.0f * filterStddev : kind == 2 ? 1.0f : kind == 5 ? (float) (int) filterRadius : 2.0f;
        float alpha = -1.0f / (2.0f * filterStddev * filterStddev), bias = std::exp(alpha * radius * radius);
        const float B = kind == 3 ? filterRadius : 0.0f, C = kind == 3 ? filterStddev : 0.5f;
        float sum = 0.0f;
        for (int i = 0; i < MI_FILTER_RES; ++i) {
            float x = (radius * (float) i) / (float) MI_FILTER_RES, v;
            if (kind == 0) v = std::fabs(x) <= radius ? 1.0f : 0.0f;
            else if (kind == 1) v = std::max(0.0f, std::exp(alpha * x * x) - bias);
            else if (kind == 2) v = std::max(0.0f, 1.0f - std::fabs(x / radius));
            else if (kind == 5) {
                float ax = std::fabs(x);
                if (ax < 1e-4f) v = 1.0f; else if (ax > radius) v = 0.0f;
                else { float x1 = MI_PI * ax, x2 = x1 / radius; v = (std::sin(x1) * std::sin(x2)) / (x1 * x2); }
            } else {
                float ax = std::fabs(x), x2 = ax * ax, x3 = x2 * ax;
                if (ax < 1) v = 1.0f / 6.0f * ((12 - 9 * B - 6 * C) * x3 + (-18 + 12 * B + 6 * C) * x2 + (6 - 2 * B));
                else if (ax < 2) v = 1.0f / 6.0f * ((-B - 6 * C) * x3 + (6 * B + 30 * C) * x2 + (-12 * B - 48 * C) * ax + (8 * B + 24 * C));
                else v = 0.0f;
            }
            filterValues[i] = v; sum += v;
        }
        filterValues[MI_FILTER_RES] = 0.0f;
        filterScale = (float) MI_FILTER_RES / radius; filterRadiusEff = radius;
        border = (int) std::ceil(radius - 0.5f);
        sum *= 2 * radius / (float) MI_FILTER_RES;
        float norm = 1.0f / sum;
        for (int i = 0; i < MI_FILTER_RES; ++i) filterValues[i] *= norm;
    }
    // --- environment emitter tables (envmap.cpp:264-330 configure, :336-347 createShape: sphere around kd-tree box + sensor position, x1.5)
    envIndex = -1; envConstant = false;
    for (uint32_t e = 0; e < ne; ++e) if (emitters[e].type == MI_EMITTER_ENVMAP || emitters[e].type == MI_EMITTER_CONSTANT) { envIndex = (int) e; envConstant = emitters[e].type == MI_EMITTER_CONSTANT; }
    {   // bounding spheres: environment emitters (envmap.cpp:336-347, constant.cpp:69-74: scene box incl. the sensor, x 1.5); directional.cpp:87-93 (kd-tree box, x 1.1)
        V3 blo = mk(aabbLo[0], aabbLo[1], aabbLo[2]), bhi = mk(aabbHi[0], aabbHi[1], aabbHi[2]);
        V3 c0 = (bhi + blo) * 0.5f, cm0 = c0 - bhi;
        dirBsCenter[0] = c0.x; dirBsCenter[1] = c0.y; dirBsCenter[2] = c0.z; dirBsRadius = std::sqrt(dot(cm0, cm0)) * 1.1f;
        V3 cam = mk(c2w[3], c2w[7], c2w[11]);
        blo = vmin(blo, cam); bhi = vmax(bhi, cam);
        V3 c = (bhi + blo) * 0.5f, cm = c - bhi;
        envBsCenter[0] = c.x; envBsCenter[1] = c.y; envBsCenter[2] = c.z; envBsRadius = std::max(MI_EPSILON, std::sqrt(dot(cm, cm)) * 1.5f);
    }
    if (envIndex >= 0 && !envConstant) {
        const int W = (int) envW, H = (int) envH;
        auto texel = [&](int x, int y) { const float *p = &envRGB[((size_t) y * W + x) * 3]; return mk(p[0], p[1], p[2]); };
        auto lum = [](V3 c) { return c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f; };
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) envToWorld3[i * 3 + j] = envToWorld[i * 4 + j];
        { const float *m = envToWorld3; float det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]); float id = 1.0f / det;
          float *o = envToLocal3;
          o[0] = (m[4] * m[8] - m[5] * m[7]) * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
          o[3] = (m[5] * m[6] - m[3] * m[8]) * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
          o[6] = (m[3] * m[7] - m[4] * m[6]) * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id; }
        envCdfCols.assign((size_t) (W + 1) * H, 0.0f); envCdfRows.assign((size_t) H + 1, 0.0f); envRowWeights.assign((size_t) H, 0.0f);
        size_t colPos = 0, rowPos = 0; float rowSum = 0.0f;
        envCdfRows[rowPos++] = 0;
        for (int y = 0; y < H; ++y) {
            float colSum = 0; envCdfCols[colPos++] = 0;
            for (int x = 0; x < W; ++x) { colSum += lum(texel(x, y)); envCdfCols[colPos++] = colSum; }
            float normalization = 1.0f / colSum;
            for (int x = 1; x < W; ++x) envCdfCols[colPos - x - 1] *= normalization;
            envCdfCols[colPos - 1] = 1.0f;
            float weight = std::sin(((float) y + 0.5f) * MI_PI / (float) H);
            envRowWeights[y] = weight; rowSum += colSum * weight; envCdfRows[rowPos++] = rowSum;
        }
        float normalization = 1.0f / rowSum;
        for (int y = 1; y < H; ++y) envCdfRows[rowPos - y - 1] *= normalization;
        envCdfRows[rowPos - 1] = 1.0f;
        // guide tables: the index lower_bound(cdf, b / K) for every bucket boundary (b / K is exact: K is a power of two), one table for the rows, one per row for the columns
        auto pow2ge = [](uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; };
        envGuideKR = std::min<uint32_t>(pow2ge((uint32_t) H), 4096u); envGuideKC = std::min<uint32_t>(pow2ge((uint32_t) W) / 4u > 0 ? pow2ge((uint32_t) W) / 4u : 1u, 1024u);
        auto guideOf = [](const float *cdf, uint32_t size, uint32_t K, uint16_t *out) {
            for (uint32_t b = 0; b <= K; ++b) { const float x = (float) b / (float) K; out[b] = b == K ? (uint16_t) (size + 1) : (uint16_t) (std::lower_bound(cdf, cdf + size + 1, x) - cdf); }
        };
        if ((uint32_t) W + 1 < 65535u && (uint32_t) H + 1 < 65535u) {
            envGuideRows.assign(envGuideKR + 1, 0); guideOf(envCdfRows.data(), (uint32_t) H, envGuideKR, envGuideRows.data());
            envGuideCols.assign((size_t) H * (envGuideKC + 1), 0);
            for (int y = 0; y < H; ++y) guideOf(envCdfCols.data() + (size_t) y * (W + 1), (uint32_t) W, envGuideKC, envGuideCols.data() + (size_t) y * (envGuideKC + 1));
        } else { envGuideRows.clear(); envGuideCols.clear(); envGuideKR = envGuideKC = 0; }
        envNormalization = 1.0f / (rowSum * (2 * MI_PI / (float) W) * (MI_PI / (float) H));
    }
    // Sobol film resolution (src/samplers/sobol.cpp:147-157)
    { uint32_t r = std::max(width, height), p = 1, l = 0; while (p < r) { p <<= 1; ++l; } resolution = (float) p; logRes = l; }
}

}  // namespace mi
