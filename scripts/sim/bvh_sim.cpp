// bvh_sim.cpp -- CPU model of the tree traversal stage (k_extend over 4-wide quantised nodes, trace.h) for design exploration WITHOUT the GPU:
// builds the scene's tree with the product's own builder (scene_build.cpp), generates the rays a path tracer would cast at bounce 1..D (cosine-weighted
// bounces off the hit surfaces), lays them out in path-pool segments exactly as k_generate / k_shade do (order-preserving compaction inside a segment), and
// runs the "while-while" walk in lock step for waves of 64 lanes under a given ray ORDER inside the segment.  Reports per bounce: nodes / triangles per ray,
// wave iterations (what the SIMD actually issues), SIMT efficiency, distinct 64-B lines touched per wave iteration.
// Test / measurement infrastructure only -- nothing here is linked into libmi355pt.so.
//   g++ -O2 -std=c++17 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ scripts/sim/bvh_sim.cpp mitsuba-im_amd/csrc/scene_build.cpp -o /tmp/bvh_sim
#include "../../mitsuba-im_amd/csrc/scene_host.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <set>

struct V3 { float x, y, z; };
static inline V3 mk(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline V3 normalize(V3 a) { float inv = 1.0f / std::sqrt(dot(a, a)); return a * inv; }

template <typename T> static std::vector<T> readFile(const std::string &p) {
    FILE *f = fopen(p.c_str(), "rb"); if (!f) { fprintf(stderr, "cannot open %s\n", p.c_str()); exit(1); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); std::vector<T> v(n / sizeof(T)); if (fread(v.data(), 1, n, f) != (size_t) n) exit(1); fclose(f); return v;
}

struct Ray { V3 o, d; float mint, maxt; uint32_t path; };
static uint32_t rngState = 12345u;
static inline float rnd() { rngState = rngState * 1664525u + 1013904223u; return (rngState >> 8) * (1.0f / 16777216.0f); }

namespace mi { void SceneHost::release() {} int SceneHost::upload(int) { return 0; } }
static mi::SceneHost H;
static inline float safeInv(float d) { float a = std::fabs(d) < 1e-30f ? std::copysign(1e-30f, d) : d; return 1.0f / a; }

static bool triTest(const TriAccelD &ta, V3 o, V3 d, float mint, float maxt, float &u, float &v, float &t) {
    if (ta.k > 2) return false;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}; static const int W[4] = {1, 2, 0, 1};
    const int k = ta.k, ku = W[k], kv = W[k + 1];
    const float o_u = oo[ku], o_v = oo[kv], o_k = oo[k], d_u = dd[ku], d_v = dd[kv], d_k = dd[k];
    t = (ta.n_d - o_u * ta.n_u - o_v * ta.n_v - o_k) / (d_u * ta.n_u + d_v * ta.n_v + d_k);
    if (!(t >= mint && t <= maxt)) return false;
    const float hu = o_u + t * d_u - ta.a_u, hv = o_v + t * d_v - ta.a_v;
    u = hv * ta.b_nu + hu * ta.b_nv; v = hu * ta.c_nu + hv * ta.c_nv;
    return u >= 0 && v >= 0 && u + v <= 1.0f;
}

// one lane of the walk; `step*` are called by the lock-step driver
struct Lane {
    V3 o, d, inv, oi; float mint, best; uint32_t prim; int cur, sp; int stk[40]; bool active; uint32_t nodes = 0, tris = 0;
    void init(const Ray &r) { o = r.o; d = r.d; inv = mk(safeInv(d.x), safeInv(d.y), safeInv(d.z)); oi = mk(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z); mint = r.mint; best = r.maxt; prim = 0xFFFFFFFFu; cur = 0; sp = 0; active = true; nodes = tris = 0; }
};
#define DONE 0x7FFFFFFF
static const Bvh4Node *N4() { return reinterpret_cast<const Bvh4Node *>(H.nodes.data()); }
static void pop(Lane &L) {
    if (L.sp > 0) { --L.sp; uint32_t e = (uint32_t) L.stk[L.sp]; uint32_t node = e & 0x7FFFFFu, slots = (e >> 23) & 0x3Fu, more = (e >> 29) & 3u;
        if (more) { L.stk[L.sp] = (int) (node | ((slots >> 2) << 23) | ((more - 1u) << 29)); ++L.sp; }
        L.cur = N4()[node].child[slots & 3u]; }
    else L.cur = DONE;
}
static std::vector<double> nodeVisits;
static int gOrder = 0;      // children after the nearest one: 0 by distance, 1 in slot order, 2 in reverse slot order, 3 no ordering at all (slot order incl. the first)
static void nodeStep(Lane &L) {       // trace.h traverse<..., WIDE = true>, one inner node
    const Bvh4Node &n = N4()[L.cur]; ++L.nodes; if (!nodeVisits.empty()) nodeVisits[L.cur] += 1;
    const float sx = n.step_x, sy = n.step_y, sz = n.step_z;
    const float bx = sx * L.inv.x, by = sy * L.inv.y, bz = sz * L.inv.z;
    const float ax = std::fmaf(n.org[0], L.inv.x, L.oi.x), ay = std::fmaf(n.org[1], L.inv.y, L.oi.y), az = std::fmaf(n.org[2], L.inv.z, L.oi.z);
    const uint32_t nxq = L.inv.x >= 0 ? n.qlo[0] : n.qhi[0], fxq = L.inv.x >= 0 ? n.qhi[0] : n.qlo[0], nyq = L.inv.y >= 0 ? n.qlo[1] : n.qhi[1], fyq = L.inv.y >= 0 ? n.qhi[1] : n.qlo[1], nzq = L.inv.z >= 0 ? n.qlo[2] : n.qhi[2], fzq = L.inv.z >= 0 ? n.qhi[2] : n.qlo[2];
    uint32_t key[4];
    for (int c = 0; c < 4; ++c) {
        const float tn = std::fmax(std::fmax(std::fmaf((float) ((nxq >> (8 * c)) & 0xFFu), bx, ax), std::fmaf((float) ((nyq >> (8 * c)) & 0xFFu), by, ay)), std::fmax(std::fmaf((float) ((nzq >> (8 * c)) & 0xFFu), bz, az), L.mint));
        const float tf = std::fmin(std::fmin(std::fmaf((float) ((fxq >> (8 * c)) & 0xFFu), bx, ax), std::fmaf((float) ((fyq >> (8 * c)) & 0xFFu), by, ay)), std::fmin(std::fmaf((float) ((fzq >> (8 * c)) & 0xFFu), bz, az), L.best));
        uint32_t tb; memcpy(&tb, &tn, 4);
        key[c] = (tn <= tf * 1.000002f + 1e-30f) ? ((tb & ~3u) | (uint32_t) c) : 0xFFFFFFFFu;
    }
    if (gOrder == 0) std::sort(key, key + 4);
    else {
        if (gOrder != 3) { int m = 0; for (int c = 1; c < 4; ++c) if (key[c] < key[m]) m = c; std::swap(key[0], key[m]); }
        // the rest: hits first, in slot order (or reverse)
        uint32_t rest[3]; int nr = 0; const int lo = gOrder == 3 ? 0 : 1;
        uint32_t all[4]; int na = 0; for (int c = lo; c < 4; ++c) if (key[c] != 0xFFFFFFFFu) all[na++] = key[c];
        std::sort(all, all + na, [](uint32_t a, uint32_t b) { return (a & 3u) < (b & 3u); }); if (gOrder == 2) std::reverse(all, all + na);
        (void) rest; (void) nr; for (int c = lo; c < 4; ++c) key[c] = (c - lo) < na ? all[c - lo] : 0xFFFFFFFFu;
    }
    if (key[0] == 0xFFFFFFFFu) pop(L);
    else {
        const uint32_t more = (key[1] != 0xFFFFFFFFu) + (key[2] != 0xFFFFFFFFu) + (key[3] != 0xFFFFFFFFu);
        if (more) { L.stk[L.sp++] = (int) ((uint32_t) L.cur | ((key[1] & 3u) << 23) | ((key[2] & 3u) << 25) | ((key[3] & 3u) << 27) | ((more - 1u) << 29)); }
        L.cur = n.child[key[0] & 3u];
    }
}
static inline uint32_t leafCount(int cur) { return ((uint32_t) ~cur & 7u) + 1u; }
static void leafTri(Lane &L, uint32_t i) {
    const uint32_t code = (uint32_t) ~L.cur, first = code >> 3; const TriAccelD &ta = H.tris[first + i]; ++L.tris;
    float u, v, t; if (triTest(ta, L.o, L.d, L.mint, L.best, u, v, t)) { if (L.prim == 0xFFFFFFFFu || t < L.best || (t == L.best && ta.prim < L.prim)) { L.best = t; L.prim = ta.prim; } }
}
// scalar walk (ray generation)
static bool trace(const Ray &r, float &t, uint32_t &prim) {
    Lane L; L.init(r);
    while (true) { while (L.cur >= 0 && L.cur != DONE) nodeStep(L); if (L.cur == DONE) break; for (uint32_t i = 0, c = leafCount(L.cur); i < c; ++i) leafTri(L, i); pop(L); }
    t = L.best; prim = L.prim; return prim != 0xFFFFFFFFu;
}


// stack depth of the fused walk (child codes pushed one by one): histogram of the deepest stack per ray
static int directDepth(const Ray &r) {
    Lane L; L.init(r); std::vector<int> st; int cur = 0, deepest = 0; float best = r.maxt; uint32_t bprim = 0xFFFFFFFFu;
    while (true) {
        if (cur >= 0) {
            const Bvh4Node &n = N4()[cur];
            const float sx = n.step_x, sy = n.step_y, sz = n.step_z, bx = sx * L.inv.x, by = sy * L.inv.y, bz = sz * L.inv.z;
            const float ax = std::fmaf(n.org[0], L.inv.x, L.oi.x), ay = std::fmaf(n.org[1], L.inv.y, L.oi.y), az = std::fmaf(n.org[2], L.inv.z, L.oi.z);
            const uint32_t nxq = L.inv.x >= 0 ? n.qlo[0] : n.qhi[0], fxq = L.inv.x >= 0 ? n.qhi[0] : n.qlo[0], nyq = L.inv.y >= 0 ? n.qlo[1] : n.qhi[1], fyq = L.inv.y >= 0 ? n.qhi[1] : n.qlo[1], nzq = L.inv.z >= 0 ? n.qlo[2] : n.qhi[2], fzq = L.inv.z >= 0 ? n.qhi[2] : n.qlo[2];
            uint32_t key[4];
            for (int c = 0; c < 4; ++c) {
                const float tn = std::fmax(std::fmax(std::fmaf((float) ((nxq >> (8 * c)) & 0xFFu), bx, ax), std::fmaf((float) ((nyq >> (8 * c)) & 0xFFu), by, ay)), std::fmax(std::fmaf((float) ((nzq >> (8 * c)) & 0xFFu), bz, az), L.mint));
                const float tf = std::fmin(std::fmin(std::fmaf((float) ((fxq >> (8 * c)) & 0xFFu), bx, ax), std::fmaf((float) ((fyq >> (8 * c)) & 0xFFu), by, ay)), std::fmin(std::fmaf((float) ((fzq >> (8 * c)) & 0xFFu), bz, az), best));
                uint32_t tb; memcpy(&tb, &tn, 4); key[c] = (tn <= tf * 1.000002f + 1e-30f) ? ((tb & ~3u) | (uint32_t) c) : 0xFFFFFFFFu;
            }
            std::sort(key, key + 4);
            for (int j = 3; j >= 1; --j) if (key[j] != 0xFFFFFFFFu) st.push_back(n.child[key[j] & 3u]);
            deepest = std::max(deepest, (int) st.size());
            if (key[0] != 0xFFFFFFFFu) { cur = n.child[key[0] & 3u]; continue; }
        } else {
            const uint32_t code = (uint32_t) ~cur, first = code >> 3, cnt = (code & 7u) + 1u;
            for (uint32_t i = 0; i < cnt; ++i) { float u, v, t; const TriAccelD &ta = H.tris[first + i]; if (triTest(ta, L.o, L.d, L.mint, best, u, v, t) && (bprim == 0xFFFFFFFFu || t < best || (t == best && ta.prim < bprim))) { best = t; bprim = ta.prim; } }
        }
        if (st.empty()) break; cur = st.back(); st.pop_back();
    }
    return deepest;
}

struct Stats { double rays = 0, nodes = 0, tris = 0, waveNodeIters = 0, waveTriIters = 0, waves = 0, nodeLaneSum = 0, triLaneSum = 0, nodeLines = 0, triLines = 0, popIters = 0; };
// lock-step walk of up to 64 rays: the "while-while" shape of trace.h (all lanes descend until each holds a leaf or is done; then all test their leaves)
static void waveWalk(const Ray *rays, int n, Stats &S) {
    static Lane L[64];
    for (int i = 0; i < n; ++i) L[i].init(rays[i]);
    S.waves += 1; S.rays += n;
    std::set<uint32_t> lines;
    while (true) {
        while (true) {      // inner nodes
            int act = 0; lines.clear();
            for (int i = 0; i < n; ++i) if (L[i].cur >= 0 && L[i].cur != DONE) { ++act; lines.insert((uint32_t) L[i].cur); }
            if (!act) break;
            for (int i = 0; i < n; ++i) if (L[i].cur >= 0 && L[i].cur != DONE) nodeStep(L[i]);
            S.waveNodeIters += 1; S.nodeLaneSum += act; S.nodeLines += lines.size();
        }
        int alive = 0; uint32_t maxc = 0;
        for (int i = 0; i < n; ++i) if (L[i].cur != DONE) { ++alive; maxc = std::max(maxc, leafCount(L[i].cur)); }
        if (!alive) break;
        for (uint32_t k = 0; k < maxc; ++k) {
            int act = 0; lines.clear();
            for (int i = 0; i < n; ++i) if (L[i].cur != DONE && k < leafCount(L[i].cur)) { ++act; lines.insert((((uint32_t) ~L[i].cur) >> 3) + k); leafTri(L[i], k); }
            S.waveTriIters += 1; S.triLaneSum += act; S.triLines += lines.size();
        }
        for (int i = 0; i < n; ++i) if (L[i].cur != DONE) pop(L[i]);
        S.popIters += 1;
    }
    for (int i = 0; i < n; ++i) { S.nodes += L[i].nodes; S.tris += L[i].tris; }
}


// A wave working through a STREAM of rays under a schedule: `thr` = refill idle lanes (from the stream) whenever fewer than thr lanes are busy at the top of the
// outer loop (thr = 1: only when the whole wave is done -- today's kernel; thr = 64: whenever a lane is idle); ifif = one fused loop in which every lane takes
// one step of whatever it needs (node or triangle) per iteration.  Returns the estimated VALU instructions issued by the wave.
struct Sched { const char *name; int thr; bool ifif; int spec; int triThr = 1; };
static double streamWalk(const std::vector<Ray> &stream, const Sched &sc, Stats &S) {
    static Lane L[64]; static uint32_t triPos[64]; static int post[64];
    for (int i = 0; i < 64; ++i) { L[i].cur = DONE; L[i].nodes = L[i].tris = 0; post[i] = DONE; }
    size_t next = 0; double cost = 0; S.waves += 1; S.rays += stream.size();
    auto retire = [&](int i) { S.nodes += L[i].nodes; S.tris += L[i].tris; L[i].nodes = L[i].tris = 0; };
    while (true) {
        int active = 0; for (int i = 0; i < 64; ++i) active += (L[i].cur != DONE || post[i] != DONE);
        if (!active && next >= stream.size()) break;
        if ((active < sc.thr || !active) && next < stream.size()) {
            for (int i = 0; i < 64 && next < stream.size(); ++i) if (L[i].cur == DONE && post[i] == DONE) { L[i].init(stream[next++]); triPos[i] = 0; }
            cost += 40;
        }
        if (sc.ifif) {
            bool anyNode = false, anyTri = false; int act = 0, nTri = 0, nNode = 0;
            for (int i = 0; i < 64; ++i) if (L[i].cur != DONE) { if (L[i].cur >= 0) ++nNode; else ++nTri; }
            const bool doTri = nTri >= sc.triThr || nNode == 0;      // the triangle block runs only when enough lanes want it (or nobody wants a node)
            for (int i = 0; i < 64; ++i) {
                if (L[i].cur == DONE) continue;
                if (L[i].cur >= 0) { ++act; nodeStep(L[i]); anyNode = true; if (L[i].cur == DONE) retire(i); else triPos[i] = 0; }
                else if (doTri) { ++act; leafTri(L[i], triPos[i]++); anyTri = true; if (triPos[i] >= leafCount(L[i].cur)) { pop(L[i]); triPos[i] = 0; if (L[i].cur == DONE) retire(i); } }
            }
            cost += (anyNode ? 175 : 0) + (anyTri ? 71 : 0) + 38; S.waveNodeIters += 1; S.waveTriIters += anyTri; S.nodeLaneSum += act;
            continue;
        }
        while (true) {      // inner nodes (spec: a lane that reaches its first leaf parks it and keeps descending)
            int act = 0;
            for (int i = 0; i < 64; ++i) {
                if (sc.spec && L[i].cur < 0 && L[i].cur != DONE && post[i] == DONE) { post[i] = L[i].cur; pop(L[i]); }
                if (L[i].cur >= 0 && L[i].cur != DONE) ++act;
            }
            if (!act) break;
            for (int i = 0; i < 64; ++i) if (L[i].cur >= 0 && L[i].cur != DONE) nodeStep(L[i]);
            cost += 120 + (sc.spec ? 8 : 0); S.waveNodeIters += 1; S.nodeLaneSum += act;
        }
        for (int pass = 0; pass < (sc.spec ? 2 : 1); ++pass) {
            uint32_t maxc = 0; int alive = 0;
            for (int i = 0; i < 64; ++i) { const int leaf = (sc.spec && pass == 0) ? post[i] : L[i].cur; if (leaf != DONE) { ++alive; maxc = std::max(maxc, leafCount(leaf)); } }
            if (!alive) continue;
            for (uint32_t k = 0; k < maxc; ++k) {
                int act = 0;
                for (int i = 0; i < 64; ++i) { const int leaf = (sc.spec && pass == 0) ? post[i] : L[i].cur; if (leaf != DONE && k < leafCount(leaf)) { ++act; const int keep = L[i].cur; L[i].cur = leaf; leafTri(L[i], k); L[i].cur = keep; } }
                cost += 45; S.waveTriIters += 1; S.triLaneSum += act;
            }
            if (sc.spec && pass == 0) { for (int i = 0; i < 64; ++i) post[i] = DONE; }
            else { for (int i = 0; i < 64; ++i) if (L[i].cur != DONE) { pop(L[i]); } cost += 15; S.popIters += 1; }
        }
        for (int i = 0; i < 64; ++i) if (L[i].cur == DONE && post[i] == DONE && (L[i].nodes || L[i].tris)) retire(i);
    }
    return cost;
}

static uint32_t part1by2(uint32_t x) { x &= 0x3FF; x = (x | (x << 16)) & 0x30000FF; x = (x | (x << 8)) & 0x300F00F; x = (x | (x << 4)) & 0x30C30C3; x = (x | (x << 2)) & 0x9249249; return x; }
static uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) { return part1by2(x) | (part1by2(y) << 1) | (part1by2(z) << 2); }

int main(int argc, char **argv) {
    std::string base = argc > 1 ? argv[1] : "/tmp/atrium";
    const int W = argc > 2 ? atoi(argv[2]) : 3840, Hh = argc > 3 ? atoi(argv[3]) : 2160;
    const int segStride = argc > 4 ? atoi(argv[4]) : 64;     // simulate every segStride-th segment
    const int maxDepth = argc > 5 ? atoi(argv[5]) : 4;
    gOrder = argc > 6 ? atoi(argv[6]) : 0;
    auto pos = readFile<float>(base + ".pos"); auto idx = readFile<uint32_t>(base + ".idx"); auto cam = readFile<float>(base + ".cam");
    H.pos = pos; H.idx = idx; mi_shape sh{}; sh.first_tri = 0; sh.tri_count = (uint32_t) (idx.size() / 3); sh.first_vert = 0; sh.vert_count = (uint32_t) (pos.size() / 3); sh.bsdf = 0; sh.emitter = -1; sh.flags = 1; sh.group = 0;
    H.shapes.push_back(sh); mi_material m{}; m.type = 0; m.reflectance[0] = m.reflectance[1] = m.reflectance[2] = 0.5f; H.materials.push_back(m);
    H.width = W; H.height = Hh; for (int i = 0; i < 16; ++i) H.c2w[i] = cam[i];
    setenv("MI355PT_BVH2", "0", 1);
    H.commitHost();
    nodeVisits.assign(H.nodes.size(), 0.0);
    printf("tree: %zu nodes (4-wide), %zu leaf records, stack need %d\n", H.nodes.size(), H.tris.size(), H.bvhDepth);
    // pinhole camera (xfov 60 degrees)
    const V3 co = mk(cam[3], cam[7], cam[11]); const V3 cx = mk(cam[0], cam[4], cam[8]), cy = mk(cam[1], cam[5], cam[9]), cz = mk(cam[2], cam[6], cam[10]);
    const float tanx = std::tan(30.0f * 3.14159265f / 180.0f), tany = tanx * Hh / W;
    struct Mode { const char *name; int layout; int segCap; int sort; };
    // layout 0: row-major pixels (k_generate today); 1: 64x64 pixel blocks of 8x8 sub-blocks.  sort 0: none; 1: direction octant; 2: octant + 10-bit origin Morton; 3: 6x... finer direction cells + Morton
    const Mode modes[] = {{"row-major, 4096/seg, unsorted", 0, 4096, 0}, {"blocked, 4096/seg, dir24+morton", 1, 4096, 4}};
    std::vector<double> depthHist(64, 0.0);
    for (const Mode &md : modes) {
        rngState = 777u;
        const uint64_t nPix = (uint64_t) W * Hh; const uint32_t cap = md.segCap; const uint64_t nSeg = (nPix + cap - 1) / cap;
        std::vector<Stats> S(maxDepth + 1);
        static const Sched scheds[] = {{"while-while, refill when the wave is done (today)", 1, false, 0}, {"refill below 32 busy lanes", 32, false, 0}, {"refill below 48", 48, false, 0}, {"refill below 64", 64, false, 0},
                                       {"fused if-if loop, refill when done", 1, true, 0}, {"fused if-if loop, refill below 32", 32, true, 0}, {"fused if-if loop, refill below 48", 48, true, 0}, {"fused if-if loop, refill below 64", 64, true, 0}, {"fused, refill below 48, triangle block at >= 8 lanes", 48, true, 0, 8}, {"fused, refill below 48, triangle block at >= 16 lanes", 48, true, 0, 16}, {"fused, refill below 48, triangle block at >= 24", 48, true, 0, 24}, {"fused, refill below 56, triangle block at >= 16", 56, true, 0, 16}, {"speculative (one parked leaf), refill when done", 1, false, 1}, {"speculative, refill below 48", 48, false, 1}};
        const int NS = (int) (sizeof(scheds) / sizeof(scheds[0]));
        std::vector<std::vector<double> > schedCost(NS, std::vector<double>(maxDepth + 1, 0.0)), schedIters = schedCost, schedLane = schedCost;
        for (uint64_t seg = 0; seg < nSeg; seg += segStride) {
            std::vector<Ray> rays;
            for (uint32_t i = 0; i < cap; ++i) {
                const uint64_t pl = seg * cap + i; if (pl >= nPix) break;
                uint32_t px, py;
                if (md.layout == 0) { px = (uint32_t) (pl % W); py = (uint32_t) (pl / W); }
                else {      // 64-row strips, 64-column blocks, 8-row sub-strips, 8-tall columns
                    const uint32_t strip = (uint32_t) (pl / ((uint64_t) 64 * W)); uint32_t r = (uint32_t) (pl % ((uint64_t) 64 * W)); const uint32_t hs = std::min<uint32_t>(64, Hh - strip * 64);
                    const uint32_t blk = r / (64 * hs); r %= 64 * hs; const uint32_t wb = std::min<uint32_t>(64, W - blk * 64);
                    const uint32_t sub = r / (8 * wb); r %= 8 * wb; const uint32_t h8 = std::min<uint32_t>(8, hs - sub * 8);
                    px = blk * 64 + r / h8; py = strip * 64 + sub * 8 + r % h8;
                }
                const float sx = (px + rnd()) / W * 2 - 1, sy = 1 - (py + rnd()) / Hh * 2;
                V3 dl = normalize(mk(-sx * tanx, sy * tany, 1.0f));
                Ray r; r.o = co; r.d = normalize(cx * dl.x + cy * dl.y + cz * dl.z); r.mint = 1e-4f; r.maxt = INFINITY; r.path = i; rays.push_back(r);
            }
            for (int depth = 1; depth <= maxDepth && !rays.empty(); ++depth) {
                // order inside the segment
                std::vector<uint32_t> order(rays.size()); for (size_t i = 0; i < order.size(); ++i) order[i] = (uint32_t) i;
                if (md.sort) {
                    V3 lo = mk(1e30f, 1e30f, 1e30f), hi = mk(-1e30f, -1e30f, -1e30f);
                    for (const Ray &r : rays) { lo = mk(std::min(lo.x, r.o.x), std::min(lo.y, r.o.y), std::min(lo.z, r.o.z)); hi = mk(std::max(hi.x, r.o.x), std::max(hi.y, r.o.y), std::max(hi.z, r.o.z)); }
                    // quantise against the SCENE box (a kernel has that as a constant)
                    lo = mk(H.aabbLo[0], H.aabbLo[1], H.aabbLo[2]); hi = mk(H.aabbHi[0], H.aabbHi[1], H.aabbHi[2]);
                    std::vector<uint64_t> key(rays.size());
                    for (size_t i = 0; i < rays.size(); ++i) {
                        const Ray &r = rays[i]; const uint32_t oct = (r.d.x < 0) | ((r.d.y < 0) << 1) | ((r.d.z < 0) << 2);
                        auto q = [&](float v, float l, float h) { float f = (v - l) / (h - l); int k = (int) (f * 1024.0f); return (uint32_t) std::min(std::max(k, 0), 1023); };
                        const uint32_t mo = morton3(q(r.o.x, lo.x, hi.x), q(r.o.y, lo.y, hi.y), q(r.o.z, lo.z, hi.z));
                        if (md.sort == 1) key[i] = oct;
                        else if (md.sort == 2) key[i] = ((uint64_t) oct << 30) | mo;
                        else if (md.sort == 3) key[i] = ((uint64_t) (mo >> 15) << 3 | oct) << 15 | (mo & 0x7FFF);
                        else { const V3 a = mk(std::fabs(r.d.x), std::fabs(r.d.y), std::fabs(r.d.z)); const uint32_t major = a.x > a.y ? (a.x > a.z ? 0 : 2) : (a.y > a.z ? 1 : 2);
                               key[i] = ((uint64_t) (oct * 3 + major) << 30) | mo; }
                    }
                    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
                }
                std::vector<Ray> ord(rays.size()); for (size_t i = 0; i < order.size(); ++i) ord[i] = rays[order[i]];
                for (size_t b = 0; b < ord.size(); b += 64) waveWalk(ord.data() + b, (int) std::min<size_t>(64, ord.size() - b), S[depth]);
                for (int w = 0; w < 4; ++w) {      // the four waves of the workgroup that owns the segment: wave w takes chunks w, w + 4, ... of 64 rays
                    std::vector<Ray> stream; for (size_t b = (size_t) w * 64; b < ord.size(); b += 256) for (size_t i = b; i < std::min(ord.size(), b + 64); ++i) stream.push_back(ord[i]);
                    if (stream.empty()) continue;
                    for (int k = 0; k < NS; ++k) { Stats tmp; schedCost[k][depth] += streamWalk(stream, scheds[k], tmp); schedIters[k][depth] += tmp.waveNodeIters; schedLane[k][depth] += tmp.nodeLaneSum; }
                }
                // next bounce (order preserving compaction in path order)
                if (md.layout == 0) for (const Ray &r : rays) { int dd = directDepth(r); ++depthHist[std::min(dd, 63)]; }
                std::vector<Ray> next;
                for (const Ray &r : rays) {
                    float t; uint32_t prim; if (!trace(r, t, prim)) continue;
                    const V3 p0 = mk(pos[idx[prim * 3] * 3], pos[idx[prim * 3] * 3 + 1], pos[idx[prim * 3] * 3 + 2]), p1 = mk(pos[idx[prim * 3 + 1] * 3], pos[idx[prim * 3 + 1] * 3 + 1], pos[idx[prim * 3 + 1] * 3 + 2]), p2 = mk(pos[idx[prim * 3 + 2] * 3], pos[idx[prim * 3 + 2] * 3 + 1], pos[idx[prim * 3 + 2] * 3 + 2]);
                    V3 n = normalize(cross(p1 - p0, p2 - p0)); if (dot(n, r.d) > 0) n = n * -1.0f;
                    const float u1 = rnd(), u2 = rnd(), rr = std::sqrt(u1), ph = 6.2831853f * u2; V3 s = std::fabs(n.x) > 0.5f ? normalize(cross(n, mk(0, 1, 0))) : normalize(cross(n, mk(1, 0, 0))), tt = cross(n, s);
                    V3 d = s * (rr * std::cos(ph)) + tt * (rr * std::sin(ph)) + n * std::sqrt(std::max(0.0f, 1 - u1));
                    Ray nr; nr.o = r.o + r.d * t; nr.d = normalize(d); nr.mint = 1e-4f; nr.maxt = INFINITY; nr.path = r.path; next.push_back(nr);
                }
                rays.swap(next);
            }
        }
        printf("\n== %s\n", md.name);
        double totalCost = 0, totalRays = 0;
        for (int d = 1; d <= maxDepth; ++d) {
            const Stats &s = S[d]; if (!s.rays) continue;
            const double cost = s.waveNodeIters * 120 + s.waveTriIters * 45 + s.popIters * 15;
            totalCost += cost; totalRays += s.rays;
            printf("  depth %d: %8.0f rays  nodes/ray %5.1f  tris/ray %5.1f | per wave: node iters %6.1f (eff %4.1f%%, %4.1f lines)  tri iters %5.1f (eff %4.1f%%, %4.1f lines) | VALU/ray est %6.0f\n", d, s.rays, s.nodes / s.rays, s.tris / s.rays,
                   s.waveNodeIters / s.waves, 100 * s.nodeLaneSum / (s.waveNodeIters * 64), s.nodeLines / s.waveNodeIters, s.waveTriIters / s.waves, 100 * s.triLaneSum / (s.waveTriIters * 64), s.triLines / s.waveTriIters, cost / s.rays);
        }
        printf("  all depths: est VALU/ray %.0f\n", totalCost / totalRays);
        for (int k = 0; k < NS; ++k) { double c = 0; printf("  schedule %-52s VALU/ray by depth:", scheds[k].name); for (int d = 1; d <= maxDepth; ++d) { c += schedCost[k][d]; printf(" %6.0f (eff %2.0f%%)", schedCost[k][d] / std::max(1.0, S[d].rays), 100 * schedLane[k][d] / std::max(1.0, schedIters[k][d] * 64)); } printf("  | all %.0f\n", c / totalRays); }
    }
    { std::vector<double> v = nodeVisits; std::sort(v.begin(), v.end(), [](double a, double b) { return a > b; }); double tot = 0; for (double x : v) tot += x; double acc = 0; printf("\nshare of node visits that go to the K most visited nodes:\n");
      for (size_t i = 0; i < v.size(); ++i) { acc += v[i]; if (i + 1 == 16 || i + 1 == 32 || i + 1 == 64 || i + 1 == 85 || i + 1 == 128 || i + 1 == 192 || i + 1 == 256 || i + 1 == 341 || i + 1 == 512 || i + 1 == 1024 || i + 1 == 2048 || i + 1 == 4096) printf("  K = %4zu: %5.1f%%\n", i + 1, 100 * acc / tot); } }
    { double tot = 0, acc = 0; for (double v : depthHist) tot += v; printf("\nfused walk: deepest stack per ray (entries: share of rays, cumulative)\n"); for (int i = 0; i < 64; ++i) if (depthHist[i]) { acc += depthHist[i]; printf("  %2d: %8.5f%%  %9.5f%%\n", i, 100 * depthHist[i] / tot, 100 * acc / tot); } printf("builder's bound: %d\n", H.bvhStackDirect); }
    return 0;
}
