// pt_device.h -- device-side arithmetic of the path tracer (gfx950).  Each function names the reference code it
// implements (paths relative to the reference tree).  Compiled with -ffp-contract=off and IEEE division / sqrt
// (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt): together with the fp64 restatement of glibc's sincosf below this makes
// every radiance sample reproducible bit for bit against a strict-IEEE CPU evaluation of the same formulas
// (DESIGN.md §"arithmetic contract"), which is what the parity tests check.
#pragma once
#include <hip/hip_runtime.h>
#include "pt_types.h"

#define DEV __device__ __forceinline__

struct v3 { float x, y, z; };
DEV v3 V(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV v3 operator*(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV v3 operator*(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
DEV v3 operator-(v3 a) { return V(-a.x, -a.y, -a.z); }
DEV float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV v3 cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV v3 normalize(v3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }   // TVector3::operator/= : recip, then multiply
DEV bool isZero(v3 a) { return a.x == 0 && a.y == 0 && a.z == 0; }
DEV float maxf(float a, float b) { return a > b ? a : b; }
DEV float minf(float a, float b) { return a < b ? a : b; }
DEV uint32_t minu(uint32_t a, uint32_t b) { return a < b ? a : b; }
DEV v3 ld3(const float *p) { return V(p[0], p[1], p[2]); }

// ---------------------------------------------------------------------------------------------- samplers
// include/mitsuba/core/qmc.h:146-156 sampleTEA
DEV uint64_t sampleTEA(uint32_t v0, uint32_t v1) {
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xA341316Cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xC8013EA4u);
        v1 += ((v0 << 4) + 0xAD90777Du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7E95761Eu);
    }
    return ((uint64_t) v1 << 32) + v0;
}
// src/libcore/random.cpp:626-634
DEV float bitsToFloat(uint32_t b) { return __uint_as_float((b >> 9) | 0x3f800000u) - 1.0f; }

// src/samplers/sobolseq.h:43-58 sampleSingle (the XOR sum starts from the scramble value).  `m32` may point to LDS or global memory.
DEV float sobolSample(const uint32_t *m32, uint64_t index, uint32_t dim, uint32_t scramble = 0u) {
    uint32_t result = scramble;
    for (uint32_t i = dim * MI_SOBOL_SIZE; index; index >>= 1, ++i)
        if (index & 1) result ^= m32[i];
    return minf((float) result * (1.0f / 4294967296.0f), MI_ONE_MINUS_EPS);
}
// The same XOR sum, four index bits per table lookup (tables built on the host from the same matrices: api.cpp buildNibbleTables).
// XOR is associative, so the result is bit-identical to sampleSingle.  `tab` may point to LDS or global memory.
typedef const __attribute__((address_space(3))) uint32_t *lds_u32_ptr;   // explicit LDS pointer: lookups compile to ds_read, not flat loads
template <typename P> struct SobolTabT { P tab; uint32_t nibs; uint32_t scramble; };
typedef SobolTabT<const uint32_t *> SobolTab;       // global memory
typedef SobolTabT<lds_u32_ptr> SobolTabLds;         // LDS copy
// nibs >= 8 always (mi_render_create pads with empty tables: entry 0 of a nibble table is 0, and index bits the render never sets read entry 0), so the eight
// nibbles of the low index word are eight independent lookups at compile-time offsets -- no loop, no dependent chain (a v_bfe + v_lshl_add per nibble, ds_read with
// an immediate offset, v_xor3) -- and the nibble extraction is shared between the two dimensions of a 2-D request.
template <typename P>
DEV uint32_t sobolBitsNib(SobolTabT<P> st, uint32_t lo, uint32_t hi, uint32_t dim) {
    P T = st.tab + dim * st.nibs * 16u;
    uint32_t result = (st.scramble ^ T[lo & 15u] ^ T[16u + ((lo >> 4) & 15u)]) ^ (T[32u + ((lo >> 8) & 15u)] ^ T[48u + ((lo >> 12) & 15u)] ^ T[64u + ((lo >> 16) & 15u)])
                    ^ (T[80u + ((lo >> 20) & 15u)] ^ T[96u + ((lo >> 24) & 15u)] ^ T[112u + (lo >> 28)]);
    for (uint32_t n = 8; n < st.nibs; ++n) result ^= T[n * 16u + ((hi >> (4u * (n - 8u))) & 15u)];
    return result;
}
template <typename P>
DEV float sobolSampleNib(SobolTabT<P> st, uint32_t lo, uint32_t hi, uint32_t dim) {
    return minf((float) sobolBitsNib(st, lo, hi, dim) * (1.0f / 4294967296.0f), MI_ONE_MINUS_EPS);
}
// Scene tables as seen by the shading kernel: either the global-memory arrays or (small scenes) a copy staged in LDS.  The pointer types
// carry the address space so that the LDS variant compiles to ds_read instead of flat loads.
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool L> struct AS { typedef const f4 *p4; typedef const float *pf; };
template <> struct AS<true> { typedef const __attribute__((address_space(3))) f4 *p4; typedef const __attribute__((address_space(3))) float *pf; };
template <bool L> struct Tabs {
    typename AS<L>::p4 shade4;      // TriShade: MI_SHADE_WORDS x 16 B per triangle
    typename AS<L>::p4 materials4;  // MaterialD: 4 x 16 B
    typename AS<L>::p4 emitters4;   // EmitterD: 3 x 16 B
    typename AS<L>::pf emitter_cdf, area_cdf;
};
template <bool L> DEV MaterialD loadMaterial(const Tabs<L> &t, int id) {
    f4 a = t.materials4[id * 4 + 0], b = t.materials4[id * 4 + 1], c = t.materials4[id * 4 + 2], e = t.materials4[id * 4 + 3];
    MaterialD m; m.type = __float_as_uint(a.x); m.flags = __float_as_uint(a.y); m.distr = __float_as_uint(a.z); m.alpha = a.w;
    m.reflectance[0] = b.x; m.reflectance[1] = b.y; m.reflectance[2] = b.z; m.eta[0] = b.w; m.eta[1] = c.x; m.eta[2] = c.y;
    m.k[0] = c.z; m.k[1] = c.w; m.k[2] = e.x; m.specular[0] = e.y; m.specular[1] = e.z; m.specular[2] = e.w;
    return m;
}
template <bool L> DEV EmitterD loadEmitter(const Tabs<L> &t, int id) {
    f4 a = t.emitters4[id * 3 + 0], b = t.emitters4[id * 3 + 1], c = t.emitters4[id * 3 + 2];
    EmitterD e; e.radiance[0] = a.x; e.radiance[1] = a.y; e.radiance[2] = a.z; e.weight = a.w;
    e.first_tri = __float_as_uint(b.x); e.tri_count = __float_as_uint(b.y); e.cdf_offset = __float_as_uint(b.z); e.inv_area = b.w;
    e.type = __float_as_uint(c.x); e.shape = __float_as_int(c.y); e.analytic = __float_as_int(c.z);
    return e;
}
// src/samplers/sobolseq.h:99-131 look_up (pixel coordinates flipped by the scramble value's top m bits); vdc / vdcInv = row (m-1) of the tables
DEV uint64_t sobolLookUp(const uint64_t *vdc, const uint64_t *vdcInv, uint32_t m, uint32_t frame, uint32_t px, uint32_t py, uint32_t scramble = 0u) {
    const uint32_t scr = scramble >> (32u - m); px ^= scr; py ^= scr;
    uint64_t index = (uint64_t) frame << (m << 1);
    uint64_t delta = 0;
    for (uint32_t c = 0; frame; frame >>= 1, ++c)
        if (frame & 1) delta ^= vdc[c];
    uint64_t b = (((uint64_t) px << m) | py) ^ delta;
    for (uint32_t c = 0; b; b >>= 1, ++c)
        if (b & 1) index ^= vdcInv[c];
    return index;
}

// Sampler state carried by a path.  Sobol: src/samplers/sobol.cpp:204-251; independent: the build-defined TEA stream.
struct SamplerState {
    uint32_t a, b;      // sobol: index lo/hi; independent: v0 (pixel ^ seed mix), sample index
    uint32_t dim;       // sobol: m_dimension; independent: call counter
};
template <typename P>
DEV float next1D(SamplerState &s, uint32_t kind, SobolTabT<P> st) {
    if (kind == 1) return sobolSampleNib(st, s.a, s.b, s.dim++);
    uint32_t v1 = (s.b << 8) | (s.dim++ & 0xFFu);
    return bitsToFloat((uint32_t) sampleTEA(s.a, v1));
}
// `first` is true only for the pixel-offset request of a sample (dimension 0), see sobol.cpp:239-245
template <typename P>
DEV void next2D(SamplerState &s, uint32_t kind, SobolTabT<P> st, float &x, float &y) {
    if (kind == 1) {
        if (s.dim + 1 >= 5 && s.dim < 5) s.dim = 5;                 // sobol.cpp:233-235 (m_arrayStartDim = m_arrayEndDim = 5)
        x = sobolSampleNib(st, s.a, s.b, s.dim++); y = sobolSampleNib(st, s.a, s.b, s.dim++);
    } else {
        uint32_t v1 = (s.b << 8) | (s.dim++ & 0xFFu);
        uint64_t r = sampleTEA(s.a, v1);
        x = bitsToFloat((uint32_t) r); y = bitsToFloat((uint32_t) (r >> 32));
    }
}

// ---------------------------------------------------------------------------------------------- warps
// glibc's sincosf, restated: the reference calls math::sincos = ::sincosf (include/mitsuba/core/math.h:219-221) in its warps, and a radiance
// sample can only equal the reference's bit for bit if the sine / cosine do.  glibc >= 2.28 (sysdeps/ieee754/flt-32/s_sincosf.c, the algorithm
// of ARM's optimized-routines) evaluates both in binary64: |y| < pi/4 -> two short polynomials in y; otherwise n = round(y * 2/pi),
// r = y - n * pi/2, the polynomials in r, swapped / negated by the quadrant.  Same operations, same order, no contraction: identical to glibc 2.35
// for EVERY float in [-8, 8] (2.18e9 arguments, scripts/check_sincosf.c; arguments on the path are <= 2 pi).  |y| >= 120 (never reached) falls
// back to the device library.
// A real function call (not inlined): the fp64 polynomial's registers then count once, not on top of the calling kernel's -- the all-diffuse shade kernel
// drops from 148 to 126 VGPRs (3 -> 4 waves per SIMD).  Returns (sin, cos) by value so that nothing goes through scratch memory.
__device__ __noinline__ static float2 glibcSincosf2(float y) {
    float sn, cs;
    const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;
    double x = (double) y, x2; int n = 0; bool neg = false;
    if (top < 0x3f4u) {                        // |y| < pi/4
        if (top < 0x398u) return make_float2(y, 1.0f);      // |y| < 2^-12
        x2 = x * x;
    } else if (top < 0x42fu) {                 // |y| < 120: reduce_fast
        const double r = x * 0x1.45F306DC9C883p+23;
        n = ((int) r + 0x800000) >> 24;
        x = x - (double) n * 0x1.921FB54442D18p0;
        x2 = x * x;
        if (((n & 3) == 1) || ((n & 3) == 2)) x = -x;         // sign[n & 3] = {1, -1, -1, 1}
        neg = (n & 2) != 0;
    } else return make_float2(sinf(y), cosf(y));
    const double c0 = neg ? -0x1p0 : 0x1p0, c1 = neg ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2, c2 = neg ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5,
                 c3 = neg ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10, c4 = neg ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double x4 = x2 * x2, x3 = x2 * x, cc2 = c3 + x2 * c4, ss1 = s2 + x2 * s3, cc1 = c0 + x2 * c1, x5 = x3 * x2, x6 = x4 * x2;
    const double s = x + x3 * s1, c = cc1 + x4 * c2;
    const float sv = (float) (s + x5 * ss1), cv = (float) (c + x6 * cc2);
    if (n & 1) { sn = cv; cs = sv; } else { sn = sv; cs = cv; }
    return make_float2(sn, cs);
}
DEV void glibcSincosf(float y, float &sn, float &cs) { const float2 r = glibcSincosf2(y); sn = r.x; cs = r.y; }
// The rest of libm the path calls, restated from glibc 2.35 like sincosf above (libm_glibc.h; every routine pinned against the host's libm by scripts/check_libm.c):
// real function calls for the same reason as glibcSincosf2.  From here on the plain names mean these routines in all device code.
#include "libm_glibc.h"
__device__ __noinline__ static float miExpf(float x) { return mi_expf(x); }
__device__ __noinline__ static float miLogf(float x) { return mi_logf(x); }
__device__ __noinline__ static float miPowf(float x, float y) { return mi_powf(x, y); }
__device__ __noinline__ static float miTanf(float x) { return mi_tanf(x); }
__device__ __noinline__ static float miAtanf(float x) { return mi_atanf(x); }
__device__ __noinline__ static float miAtan2f(float y, float x) { return mi_atan2f(y, x); }
__device__ __noinline__ static float miAcosf(float x) { return mi_acosf(x); }
DEV float miSinf(float x) { return glibcSincosf2(x).x; }      // glibc's sinf / cosf are sincosf's two halves (same polynomials, same reduction: scripts/check_sincosf.c)
DEV float miCosf(float x) { return glibcSincosf2(x).y; }
#define expf miExpf
#define logf miLogf
#define powf miPowf
#define tanf miTanf
#define atanf miAtanf
#define atan2f miAtan2f
#define acosf miAcosf
#define sinf miSinf
#define cosf miCosf
// src/libcore/warp.cpp:81-101 squareToUniformDiskConcentric
DEV void diskConcentric(float sx, float sy, float &ox, float &oy) {
    float r1 = 2.0f * sx - 1.0f, r2 = 2.0f * sy - 1.0f, r, phi, sn, cs;
    if (r1 == 0 && r2 == 0) r = phi = 0;
    else if (r1 * r1 > r2 * r2) { r = r1; phi = (MI_PI / 4.0f) * (r2 / r1); }
    else { r = r2; phi = (MI_PI / 2.0f) - (r1 / r2) * (MI_PI / 4.0f); }
    glibcSincosf(phi, sn, cs);
    ox = r * cs; oy = r * sn;
}
// warp.cpp:43-52 squareToCosineHemisphere
DEV v3 cosHemisphere(float sx, float sy) {
    float px, py; diskConcentric(sx, sy, px, py);
    float z = sqrtf(maxf(1.0f - px * px - py * py, 0.0f));
    if (z == 0) z = 1e-10f;
    return V(px, py, z);
}
// warp.cpp:76-79 squareToUniformTriangle
DEV void uniformTriangle(float sx, float sy, float &bx, float &by) { float a = sqrtf(maxf(1.0f - sx, 0.0f)); bx = 1 - a; by = a * sy; }

// ---------------------------------------------------------------------------------------------- camera
// src/sensors/perspective.cpp:271-287 + include/mitsuba/core/transform.h:108-125
DEV void cameraRay(const DScene &sc, float sx, float sy, v3 &o, v3 &d, float &mint, float &maxt) {
    const float *m = sc.s2c;
    float px = sx * sc.inv_res_x, py = sy * sc.inv_res_y, pz = 0.0f;
    float x = m[0] * px + m[1] * py + m[2] * pz + m[3];
    float y = m[4] * px + m[5] * py + m[6] * pz + m[7];
    float z = m[8] * px + m[9] * py + m[10] * pz + m[11];
    float w = m[12] * px + m[13] * py + m[14] * pz + m[15];
    v3 nearP = V(x, y, z);
    if (w != 1.0f) { float r = 1.0f / w; nearP = nearP * r; }
    v3 dl = normalize(nearP);
    float invZ = 1.0f / dl.z;
    mint = sc.near_clip * invZ; maxt = sc.far_clip * invZ;
    const float *c = sc.c2w;
    o = V(c[3], c[7], c[11]);
    d = V(c[0] * dl.x + c[1] * dl.y + c[2] * dl.z, c[4] * dl.x + c[5] * dl.y + c[6] * dl.z, c[8] * dl.x + c[9] * dl.y + c[10] * dl.z);
}

// ---------------------------------------------------------------------------------------------- ray / scene box
// include/mitsuba/core/aabb.h:308-339 + src/librender/skdtree.cpp:112-142 (closest) / :207-226 (any hit): scene-box clip + adaptive epsilon
DEV bool clipInterval(const DScene &sc, v3 o, v3 d, float rmint, float rmaxt, bool shadow, float &mint, float &maxt) {
    float nt = -INFINITY, ft = INFINITY;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float origin = oo[i], minv = sc.aabb_lo[i], maxv = sc.aabb_hi[i], di = dd[i];
        if (di == 0) { if (origin < minv || origin > maxv) return false; }
        else {
            float rcp = 1.0f / di;
            float t1 = (minv - origin) * rcp, t2 = (maxv - origin) * rcp;
            if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; }
            nt = maxf(t1, nt); ft = minf(t2, ft);
            if (!(nt <= ft)) return false;
        }
    }
    float rayMinT = rmint;
    if (rayMinT == MI_EPSILON) {
        float m = maxf(maxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
        if (!shadow) m = maxf(m, MI_EPSILON);
        rayMinT *= m;
    }
    if (rayMinT > nt) nt = rayMinT;
    if (rmaxt < ft) ft = rmaxt;
    mint = nt; maxt = ft;
    return ft > nt;
}

// include/mitsuba/render/triaccel.h:96-158 TriAccel::rayIntersect
DEV bool triIntersect(const TriAccelD &ta, v3 o, v3 d, float mint, float maxt, float &u, float &v, float &t) {
    float o_u, o_v, o_k, d_u, d_v, d_k;
    if (ta.k == 0) { o_u = o.y; o_v = o.z; o_k = o.x; d_u = d.y; d_v = d.z; d_k = d.x; }
    else if (ta.k == 1) { o_u = o.z; o_v = o.x; o_k = o.y; d_u = d.z; d_v = d.x; d_k = d.y; }
    else if (ta.k == 2) { o_u = o.x; o_v = o.y; o_k = o.z; d_u = d.x; d_v = d.y; d_k = d.z; }
    else return false;
    float tt = (ta.n_d - o_u * ta.n_u - o_v * ta.n_v - o_k) / (d_u * ta.n_u + d_v * ta.n_v + d_k);
    if (tt < mint || tt > maxt) return false;
    float hu = o_u + tt * d_u - ta.a_u, hv = o_v + tt * d_v - ta.a_v;
    float uu = hv * ta.b_nu + hu * ta.b_nv, vv = hu * ta.c_nu + hv * ta.c_nv;
    u = uu; v = vv; t = tt;
    return uu >= 0 && vv >= 0 && uu + vv <= 1.0f;
}

// ---------------------------------------------------------------------------------------------- hit record
// include/mitsuba/core/transform.h:126-135 transformAffine(Point), :172-181 operator()(Vector), :199-207 operator()(Normal); m = rows 0..2 of the 4x4
DEV v3 xfPoint(const float *m, v3 p) { return V(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7], m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]); }
DEV v3 xfVector(const float *m, v3 v) { return V(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z); }
DEV v3 xfNormal(const float *inv, v3 n) { return V(inv[0] * n.x + inv[4] * n.y + inv[8] * n.z, inv[1] * n.x + inv[5] * n.y + inv[9] * n.z, inv[2] * n.x + inv[6] * n.y + inv[10] * n.z); }
struct Hit {
    v3 p, ng, ns, s, t, wi; float dist; int material, emitter; uint32_t flags; float uvx, uvy;   // uvx, uvy: its.uv (texcoords; set by the EXT paths only)
};
// include/mitsuba/render/skdtree.h:343-428 fillIntersectionRecord<true> + src/libcore/util.cpp:605-610
// ---- bitmap textures: TMIPMap (include/mitsuba/render/mipmap.h): evalTexel :504-560, evalBox :562-566, evalBilinear :572-596, eval :625-705,
// evalEWA :746-818.  The pyramid is input data (the reference's Bitmap::resample output, half-precision texels widened to float).
DEV int imod(int a, int b) { int r = a % b; return r < 0 ? r + b : r; }
DEV v3 mipTexel(const DScene &sc, const TextureD &t, int level, int x, int y) {
    const uint32_t *L = sc.tex_levels + (t.first_level + (uint32_t) level) * 3u; const int w = (int) L[0], h = (int) L[1];
    if (x < 0 || x >= w) {
        if (t.wrap_u == 1u) x = imod(x, w);
        else if (t.wrap_u == 0u) x = x < 0 ? 0 : w - 1;
        else if (t.wrap_u == 2u) { x = imod(x, 2 * w); if (x >= w) x = 2 * w - x - 1; }
        else return t.wrap_u == 3u ? V(0, 0, 0) : V(1, 1, 1);
    }
    if (y < 0 || y >= h) {
        if (t.wrap_v == 1u) y = imod(y, h);
        else if (t.wrap_v == 0u) y = y < 0 ? 0 : h - 1;
        else if (t.wrap_v == 2u) { y = imod(y, 2 * h); if (y >= h) y = 2 * h - y - 1; }
        else return t.wrap_v == 3u ? V(0, 0, 0) : V(1, 1, 1);
    }
    return ld3(sc.tex_texels + L[2] + ((size_t) y * w + x) * 3);
}
DEV v3 mipBox(const DScene &sc, const TextureD &t, int level, float u, float v) {
    const uint32_t *L = sc.tex_levels + (t.first_level + (uint32_t) level) * 3u;
    return mipTexel(sc, t, level, (int) floorf(u * (float) (int) L[0]), (int) floorf(v * (float) (int) L[1]));
}
DEV v3 mipBilinear(const DScene &sc, const TextureD &t, int level, float uvx, float uvy) {
    if (!isfinite(uvx) || !isfinite(uvy)) return V(0, 0, 0);
    if (level >= (int) t.n_levels) return mipBox(sc, t, (int) t.n_levels - 1, uvx, uvy);
    const uint32_t *L = sc.tex_levels + (t.first_level + (uint32_t) level) * 3u;
    float u = uvx * (float) (int) L[0] - 0.5f, v = uvy * (float) (int) L[1] - 0.5f;
    int xPos = (int) floorf(u), yPos = (int) floorf(v);
    float dx1 = u - (float) xPos, dx2 = 1.0f - dx1, dy1 = v - (float) yPos, dy2 = 1.0f - dy1;
    v3 r = (mipTexel(sc, t, level, xPos, yPos) * dx2) * dy2;
    r = r + (mipTexel(sc, t, level, xPos, yPos + 1) * dx2) * dy1;
    r = r + (mipTexel(sc, t, level, xPos + 1, yPos) * dx1) * dy2;
    r = r + (mipTexel(sc, t, level, xPos + 1, yPos + 1) * dx1) * dy1;
    return r;
}
DEV v3 mipEwa(const DScene &sc, const TextureD &t, int level, float uvx, float uvy, float A, float B, float C) {
    if (!isfinite(A + B + C + uvx + uvy)) return V(0, 0, 0);
    if (level >= (int) t.n_levels) return mipBox(sc, t, (int) t.n_levels - 1, uvx, uvy);
    const uint32_t *L = sc.tex_levels + (t.first_level + (uint32_t) level) * 3u, *L0 = sc.tex_levels + t.first_level * 3u;
    float u = uvx * (float) (int) L[0] - 0.5f, v = uvy * (float) (int) L[1] - 0.5f;
    const float rx = (float) (int) L[0] / (float) (int) L0[0], ry = (float) (int) L[1] / (float) (int) L0[1];
    A /= rx * rx; B /= rx * ry; C /= ry * ry;
    float invDet = 1.0f / (-B * B + 4.0f * A * C), deltaU = 2.0f * sqrtf(C * invDet), deltaV = 2.0f * sqrtf(A * invDet);
    int u0 = (int) ceilf(u - deltaU), u1 = (int) floorf(u + deltaU), v0 = (int) ceilf(v - deltaV), v1 = (int) floorf(v + deltaV);
    float As = A * 64, Bs = B * 64, Cs = C * 64;
    v3 result = V(0, 0, 0); float denominator = 0.0f, ddq = 2 * As, uu0 = (float) u0 - u;
    for (int vt = v0; vt <= v1; ++vt) {
        const float vv = (float) vt - v;
        float q = As * uu0 * uu0 + (Bs * uu0 + Cs * vv) * vv, dq = As * (2 * uu0 + 1) + Bs * vv;
        for (int ut = u0; ut <= u1; ++ut) {
            if (q < 64.0f) {
                uint32_t qi = (uint32_t) q;
                if (qi < 64u) { const float weight = sc.mip_lut[(int) q]; result = result + mipTexel(sc, t, level, ut, vt) * weight; denominator += weight; }
            }
            q += dq; dq += ddq;
        }
    }
    if (denominator == 0) return mipBilinear(sc, t, level, uvx, uvy);
    float r = 1.0f / denominator; return result * r;
}
DEV float hypot2f(float a, float b) {                          // src/libcore/math.cpp:74-86
    float r;
    if (fabsf(a) > fabsf(b)) { r = b / a; r = fabsf(a) * sqrtf(1.0f + r * r); }
    else if (b != 0.0f) { r = a / b; r = fabsf(b) * sqrtf(1.0f + r * r); }
    else r = 0.0f;
    return r;
}
DEV float miLog2(float v) { const float invLn2 = 1.4426950408889634f; return (float) log((double) v) * invLn2; }   // math.cpp:103-106 (1 / logf(2))
DEV v3 mipEval(const DScene &sc, const TextureD &t, float uvx, float uvy, float d0x, float d0y, float d1x, float d1y) {
    if (t.filter == 0u) return mipBox(sc, t, 0, uvx, uvy);
    if (t.filter == 1u) return mipBilinear(sc, t, 0, uvx, uvy);
    const uint32_t *L0 = sc.tex_levels + t.first_level * 3u; const float sx = (float) (int) L0[0], sy = (float) (int) L0[1];
    float du0 = d0x * sx, dv0 = d0y * sy, du1 = d1x * sx, dv1 = d1y * sy;
    float A = dv0 * dv0 + dv1 * dv1, B = -2.0f * (du0 * dv0 + du1 * dv1), C = du0 * du0 + du1 * du1, F = A * C - B * B * 0.25f;
    float root = hypot2f(A - C, B), Aprime = 0.5f * (A + C - root), Cprime = 0.5f * (A + C + root),
          majorRadius = Aprime != 0 ? sqrtf(F / Aprime) : 0, minorRadius = Cprime != 0 ? sqrtf(F / Cprime) : 0;
    if (t.filter == 2u || !(minorRadius > 0) || !(majorRadius > 0) || F < 0) {
        float level = miLog2(maxf(majorRadius, MI_EPSILON)); int ilevel = (int) floorf(level);
        if (ilevel < 0) return mipBilinear(sc, t, 0, uvx, uvy);
        float a = level - (float) ilevel;
        return mipBilinear(sc, t, ilevel, uvx, uvy) * (1.0f - a) + mipBilinear(sc, t, ilevel + 1, uvx, uvy) * a;
    }
    if (minorRadius * t.max_anisotropy < majorRadius) {
        minorRadius = majorRadius / t.max_anisotropy;
        float theta = 0.5f * atanf(B / (A - C)); const float2 scT_ = glibcSincosf2(theta); float sinTheta = scT_.x, cosTheta = scT_.y;
        float a2 = majorRadius * majorRadius, b2 = minorRadius * minorRadius, sinTheta2 = sinTheta * sinTheta, cosTheta2 = cosTheta * cosTheta, sin2Theta = 2 * sinTheta * cosTheta;
        A = a2 * cosTheta2 + b2 * sinTheta2; B = (a2 - b2) * sin2Theta; C = a2 * sinTheta2 + b2 * cosTheta2; F = a2 * b2;
    }
    float scl = 1.0f / F; A *= scl; B *= scl; C *= scl;
    float level = maxf(0.0f, miLog2(minorRadius)); int ilevel = (int) level; float a = level - (float) ilevel;
    if (majorRadius < 1 || !(A > 0 && C > 0)) return mipBilinear(sc, t, ilevel, uvx, uvy);
    return mipEwa(sc, t, ilevel, uvx, uvy, A, B, C) * (1.0f - a) + mipEwa(sc, t, ilevel + 1, uvx, uvy, A, B, C) * a;
}
// Intersection::computePartials (src/librender/intersection.cpp:5-76) for the hit of a CAMERA ray; rxd / ryd: its differentials (common origin o)
DEV void computePartials(v3 p, v3 ng, v3 dpdu, v3 dpdv, v3 o, v3 rxd, v3 ryd, float *pa) {
    pa[0] = pa[1] = pa[2] = pa[3] = 0.0f;
    if (isZero(dpdu) && isZero(dpdv)) return;
    const float pp = dot(ng, p), pox = dot(ng, o), poy = dot(ng, o), prx = dot(ng, rxd), pry = dot(ng, ryd);
    if (prx == 0 || pry == 0) return;
    const float tx = (pp - pox) / prx, ty = (pp - poy) / pry;
    float absX = fabsf(ng.x), absY = fabsf(ng.y), absZ = fabsf(ng.z); int a0, a1;
    if (absX > absY && absX > absZ) { a0 = 1; a1 = 2; } else if (absY > absZ) { a0 = 0; a1 = 2; } else { a0 = 0; a1 = 1; }
    auto cmp = [](v3 q, int i) { return i == 0 ? q.x : (i == 1 ? q.y : q.z); };
    float A00 = cmp(dpdu, a0), A01 = cmp(dpdv, a0), A10 = cmp(dpdu, a1), A11 = cmp(dpdv, a1);
    v3 px = o + rxd * tx, py = o + ryd * ty;
    float Bx0 = cmp(px, a0) - cmp(p, a0), Bx1 = cmp(px, a1) - cmp(p, a1), By0 = cmp(py, a0) - cmp(p, a0), By1 = cmp(py, a1) - cmp(p, a1);
    float det = A00 * A11 - A01 * A10;
    if (fabsf(det) <= 0x1p-128f) { pa[0] = 1; pa[1] = 0; pa[2] = 1; return; }      // solveLinearSystem2x2 fails (util.cpp:529-541); the reference's second fallback assigns dudy twice
    float inverse = 1.0f / det;
    pa[0] = (A11 * Bx0 - A01 * Bx1) * inverse; pa[1] = (A00 * Bx1 - A10 * Bx0) * inverse;
    pa[2] = (A11 * By0 - A01 * By1) * inverse; pa[3] = (A00 * By1 - A10 * By0) * inverse;
}
// ray differentials of the sensor ray: perspective.cpp:290-295 + RayDifferential::scaleDifferential(1 / sqrt(spp)) (ray.h:163-168)
DEV void cameraDifferentials(const DScene &sc, float invSqrtSpp, float sx, float sy, v3 d, v3 &rxd, v3 &ryd) {
    const float *m = sc.s2c;
    float px = sx * sc.inv_res_x, py = sy * sc.inv_res_y, pz = 0.0f;
    float x = m[0] * px + m[1] * py + m[2] * pz + m[3], y = m[4] * px + m[5] * py + m[6] * pz + m[7], z = m[8] * px + m[9] * py + m[10] * pz + m[11], w = m[12] * px + m[13] * py + m[14] * pz + m[15];
    v3 nearP = V(x, y, z);
    if (w != 1.0f) { float r = 1.0f / w; nearP = nearP * r; }
    const float *c = sc.c2w;
    v3 a = normalize(nearP + ld3(sc.cam_dx)), b = normalize(nearP + ld3(sc.cam_dy));
    v3 rx = V(c[0] * a.x + c[1] * a.y + c[2] * a.z, c[4] * a.x + c[5] * a.y + c[6] * a.z, c[8] * a.x + c[9] * a.y + c[10] * a.z);
    v3 ry = V(c[0] * b.x + c[1] * b.y + c[2] * b.z, c[4] * b.x + c[5] * b.y + c[6] * b.z, c[8] * b.x + c[9] * b.y + c[10] * b.z);
    rxd = d + (rx - d) * invSqrtSpp; ryd = d + (ry - d) * invSqrtSpp;
}
// Checkerboard::eval (src/textures/checkerboard.cpp:68-76), GridTexture::eval (src/textures/gridtexture.cpp:63-77) under Texture2D::eval
// (src/librender/texture.cpp:112-121; these textures do not filter: usesRayDifferentials() = false)
DEV v3 textureEval(const TextureD &t, float u, float v) {
    float uvx = u * t.uscale + t.uoffset, uvy = v * t.vscale + t.voffset; bool first;
    if (t.type == 0) {
        int a = (int) (uvx * 2) % 2, b = (int) (uvy * 2) % 2; if (a < 0) a += 2; if (b < 0) b += 2;
        first = (2 * a - 1) * (2 * b - 1) == 1;
    } else {
        float x = uvx - (float) (int) floorf(uvx), y = uvy - (float) (int) floorf(uvy);
        if (x > .5) x -= 1;
        if (y > .5) y -= 1;
        first = !(fabsf(x) < t.line_width || fabsf(y) < t.line_width);
    }
    return first ? ld3(t.color0) : ld3(t.color1);
}
// UVT: honour texture coordinates (the EXT kernel variants): its.uv and dpdu = UV tangent for triangles whose mesh has texcoords
template <bool L, bool UVT = false>
DEV void fillHit(const DScene &sc, const Tabs<L> &tb, v3 d, float t, uint32_t prim, float u, float v, Hit &h) {
    typename AS<L>::p4 rec = tb.shade4 + prim * (uint32_t) MI_SHADE_WORDS;
    f4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4], r5 = rec[5];
    v3 p0 = V(r0.x, r0.y, r0.z), p1 = V(r1.x, r1.y, r1.z), p2 = V(r2.x, r2.y, r2.z);
    h.material = __float_as_int(r0.w); h.emitter = __float_as_int(r1.w); h.flags = __float_as_uint(r2.w);
    float bx = 1 - u - v, by = u, bz = v;
    h.dist = t;
    h.p = (p0 * bx + p1 * by) + p2 * bz;
    v3 fn = V(r3.x, r3.y, r3.z);
    h.uvx = by; h.uvy = bz;
    const bool hasUV = UVT && (h.flags & 16u);
    v3 tangent = V(0, 0, 0);
    if (hasUV) {
        const TriUV &tu = sc.triuv[prim];
        h.uvx = (tu.uv0[0] * bx + tu.uv1[0] * by) + tu.uv2[0] * bz; h.uvy = (tu.uv0[1] * bx + tu.uv1[1] * by) + tu.uv2[1] * bz;
        tangent = ld3(tu.dpdu);
    }
    if (h.flags & 1u) {          // face normals: the precomputed face frame (built from the UV tangent where there is one) is the shading frame
        h.ns = fn; h.ng = fn; h.s = V(r4.x, r4.y, r4.z); h.t = V(r5.x, r5.y, r5.z);
    } else {
        const f4 r6 = rec[6];      // the vertex normals travel in the record (words 4, 5, 6)
        v3 n = (V(r4.x, r4.y, r4.z) * bx + V(r5.x, r5.y, r5.z) * by) + V(r6.x, r6.y, r6.z) * bz;
        h.ns = normalize(n);
        if (dot(fn, h.ns) < 0) fn = -fn;
        h.ng = fn;
        v3 dpdu = hasUV ? tangent : p1 - p0;
        h.s = normalize(dpdu - h.ns * dot(h.ns, dpdu));
        h.t = cross(h.ns, h.s);
    }
    v3 md = -d;
    h.wi = V(dot(md, h.s), dot(md, h.t), dot(md, h.ns));
}
// Instance::fillIntersectionRecord (src/shapes/instance.cpp:126-141): the group member is filled in object space by
// fillIntersectionRecord<false> (p = ray(t) of the object-space ray, skdtree.h:361-365), then normals go through the inverse transpose,
// dpdu and p through the forward transform; the scene level recomputes the shading frame and wi (skdtree.h:425-426).
template <bool L>
DEV void fillHitInstanced(const DScene &sc, const Tabs<L> &tb, const InstanceD &in, v3 o, v3 d, float t, uint32_t prim, float u, float v, Hit &h) {
    typename AS<L>::p4 rec = tb.shade4 + prim * (uint32_t) MI_SHADE_WORDS;
    f4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4], r5 = rec[5];
    v3 p0 = V(r0.x, r0.y, r0.z), p1 = V(r1.x, r1.y, r1.z);
    h.material = __float_as_int(r0.w); h.emitter = __float_as_int(r1.w); h.flags = __float_as_uint(r2.w);
    float bx = 1 - u - v, by = u, bz = v;
    h.dist = t;
    v3 o2 = xfPoint(in.to_object, o), d2 = xfVector(in.to_object, d);
    v3 pObj = o2 + d2 * t;
    v3 fn = V(r3.x, r3.y, r3.z), ns;
    if (h.flags & 1u) ns = fn;
    else {
        const f4 r6 = rec[6];      // the vertex normals travel in the record (words 4, 5, 6)
        v3 n = (V(r4.x, r4.y, r4.z) * bx + V(r5.x, r5.y, r5.z) * by) + V(r6.x, r6.y, r6.z) * bz;
        ns = normalize(n);
        if (dot(fn, ns) < 0) fn = -fn;
    }
    h.ns = normalize(xfNormal(in.to_object, ns)); h.ng = normalize(xfNormal(in.to_object, fn));
    v3 dpduObj = p1 - p0; h.uvx = by; h.uvy = bz;
    if (h.flags & 16u) {
        const TriUV &tu = sc.triuv[prim];
        h.uvx = (tu.uv0[0] * bx + tu.uv1[0] * by) + tu.uv2[0] * bz; h.uvy = (tu.uv0[1] * bx + tu.uv1[1] * by) + tu.uv2[1] * bz;
        dpduObj = ld3(tu.dpdu);
    }
    v3 dpdu = xfVector(in.to_world, dpduObj);
    h.p = xfPoint(in.to_world, pObj);
    h.s = normalize(dpdu - h.ns * dot(h.ns, dpdu));
    h.t = cross(h.ns, h.s);
    v3 md = -d;
    h.wi = V(dot(md, h.s), dot(md, h.t), dot(md, h.ns));
}
DEV v3 toWorld(const Hit &h, v3 w) { return (h.s * w.x + h.t * w.y) + h.ns * w.z; }
DEV v3 toLocal(const Hit &h, v3 w) { return V(dot(w, h.s), dot(w, h.t), dot(w, h.ns)); }

// ---------------------------------------------------------------------------------------------- analytic shapes
// Non-triangle primitives behind Scene::rayIntersect (kd-tree leaf redirect include/mitsuba/render/skdtree.h:292-301, :330-333, record
// fill :421-427): src/shapes/rectangle.cpp, disk.cpp, sphere.cpp, cylinder.cpp.  The quadrics are solved in double precision like
// the reference (solveQuadraticDouble, src/libcore/util.cpp:489-527).
// include/mitsuba/core/transform.h:126-135 transformAffine(Point), :172-181 operator()(Vector), :199-207 operator()(Normal); m = rows 0..2 of the 4x4
// math::sincos(2.0f * M_PI * u) of squareToUniformSphere / squareToUniformCone (warp.cpp:29, :59) and Cylinder::samplePosition (cylinder.cpp:234)
DEV void sincos2pi(float u, float &sn, float &cs) { glibcSincosf((2.0f * MI_PI) * u, sn, cs); }
DEV bool solveQuadraticDouble(double a, double b, double c, double &x0, double &x1) {
    if (a == 0) { if (b != 0) { x0 = x1 = -c / b; return true; } return false; }
    double discrim = b * b - 4.0 * a * c;
    if (discrim < 0) return false;
    double temp, sqrtDiscrim = sqrt(discrim);
    if (b < 0) temp = -0.5 * (b - sqrtDiscrim); else temp = -0.5 * (b + sqrtDiscrim);
    x0 = temp / a; x1 = c / temp;
    if (x0 > x1) { double t = x0; x0 = x1; x1 = t; }
    return true;
}
DEV bool solveQuadratic(float a, float b, float c, float &x0, float &x1) {          // util.cpp:449-487
    if (a == 0) { if (b != 0) { x0 = x1 = -c / b; return true; } return false; }
    float discrim = b * b - 4.0f * a * c;
    if (discrim < 0) return false;
    float temp, sqrtDiscrim = sqrtf(discrim);
    if (b < 0) temp = -0.5f * (b - sqrtDiscrim); else temp = -0.5f * (b + sqrtDiscrim);
    x0 = temp / a; x1 = c / temp;
    if (x0 > x1) { float t = x0; x0 = x1; x1 = t; }
    return true;
}
DEV void coordinateSystem(v3 a, v3 &b, v3 &c) {                                     // util.cpp:594-603
    if (fabsf(a.x) > fabsf(a.y)) { float invLen = 1.0f / sqrtf(a.x * a.x + a.z * a.z); c = V(a.z * invLen, 0.0f, -a.x * invLen); }
    else { float invLen = 1.0f / sqrtf(a.y * a.y + a.z * a.z); c = V(0.0f, a.z * invLen, -a.y * invLen); }
    b = cross(c, a);
}
// Shape::rayIntersect: rectangle.cpp:125-148, disk.cpp:141-165, sphere.cpp:148-174, cylinder.cpp:128-166; ANY selects the shadow-ray
// overloads (sphere.cpp:176-194, cylinder.cpp:168-201).  u, v = the temp data (local x, y) of rectangle / disk.
template <bool ANY>
DEV bool analyticIntersect(const AnalyticD &sh, v3 o, v3 d, float mint, float maxt, float &t, float &u, float &v) {
    const uint32_t type = sh.type;
    if (type == MI_SHAPE_RECTANGLE || type == MI_SHAPE_DISK) {
        v3 ro = xfPoint(sh.to_object, o), rd = xfVector(sh.to_object, d);
        float hit = -ro.z / rd.z;
        if (!(hit >= mint && hit <= maxt)) return false;
        float lx = ro.x + hit * rd.x, ly = ro.y + hit * rd.y;
        if (type == MI_SHAPE_RECTANGLE ? (fabsf(lx) <= 1 && fabsf(ly) <= 1) : (lx * lx + ly * ly <= 1)) { t = hit; u = lx; v = ly; return true; }
        return false;
    }
    double nearT, farT;
    if (type == MI_SHAPE_SPHERE) {
        double ox = (double) o.x - (double) sh.center[0], oy = (double) o.y - (double) sh.center[1], oz = (double) o.z - (double) sh.center[2];
        double dx = d.x, dy = d.y, dz = d.z;
        double A = dx * dx + dy * dy + dz * dz, B = 2 * (ox * dx + oy * dy + oz * dz), C = (ox * ox + oy * oy + oz * oz) - (double) (sh.radius * sh.radius);
        if (!solveQuadraticDouble(A, B, C, nearT, farT)) return false;
        if (ANY) {
            if (nearT > maxt || farT < mint) return false;
            if (nearT < mint && farT > maxt) return false;
            t = 0; return true;
        }
        if (!(nearT <= maxt && farT >= mint)) return false;
        if (nearT < mint) { if (farT > maxt) return false; t = (float) farT; } else t = (float) nearT;
        u = 0; v = 0; return true;
    }
    v3 ro = xfPoint(sh.to_object, o), rd = xfVector(sh.to_object, d);
    double ox = ro.x, oy = ro.y, dx = rd.x, dy = rd.y;
    double A = dx * dx + dy * dy, B = 2 * (dx * ox + dy * oy), C = ox * ox + oy * oy - (double) (sh.radius * sh.radius);
    if (!solveQuadraticDouble(A, B, C, nearT, farT)) return false;
    if (ANY) { if (nearT > maxt || farT < mint) return false; }
    else if (!(nearT <= maxt && farT >= mint)) return false;
    double zPosNear = (double) ro.z + (double) rd.z * nearT, zPosFar = (double) ro.z + (double) rd.z * farT;
    if (zPosNear >= 0 && zPosNear <= sh.length && nearT >= mint) t = (float) nearT;
    else if (zPosFar >= 0 && zPosFar <= sh.length) { if (farT > maxt) return false; t = (float) farT; }
    else return false;
    u = 0; v = 0; return true;
}
// its.uv / its.dpdu / its.dpdv of an analytic hit (rectangle.cpp:161-163, disk.cpp:173-193, sphere.cpp:218-245, cylinder.cpp:204-216): textures need them, so they
// are evaluated on demand in the texture block of k_shade.  (lx, ly): the hit's local coordinates as stored by the intersection routine; pRay = ray(t)
DEV void analyticUV(const AnalyticD &sh, float lx, float ly, v3 pRay, float &uvx, float &uvy, v3 &dpdu, v3 &dpdv) {
    const uint32_t type = sh.type;
    if (type == MI_SHAPE_RECTANGLE) { uvx = 0.5f * (lx + 1); uvy = 0.5f * (ly + 1); dpdu = ld3(sh.dpdu); dpdv = xfVector(sh.to_world, V(0, 2, 0)); }
    else if (type == MI_SHAPE_DISK) {
        float r = sqrtf(lx * lx + ly * ly), invR = (r == 0) ? 0.0f : (1.0f / r);
        float phi = atan2f(ly, lx); if (phi < 0) phi += 2 * MI_PI;
        float cosPhi = lx * invR, sinPhi = ly * invR;
        if (r != 0) { dpdu = xfVector(sh.to_world, V(cosPhi, sinPhi, 0)); dpdv = xfVector(sh.to_world, V(-sinPhi, cosPhi, 0)); }
        else { dpdu = xfVector(sh.to_world, V(1, 0, 0)); dpdv = xfVector(sh.to_world, V(0, 1, 0)); }
        uvx = r; uvy = phi * MI_INV_TWOPI;
    } else if (type == MI_SHAPE_SPHERE) {
        const v3 c = ld3(sh.center); v3 p = c + normalize(pRay - c) * sh.radius;
        v3 local = xfVector(sh.to_object, p - c);
        float theta = acosf(minf(1.0f, maxf(-1.0f, local.z / sh.radius))), phi = atan2f(local.y, local.x); if (phi < 0) phi += 2 * MI_PI;
        uvx = phi * (0.5f * MI_INV_PI); uvy = theta * MI_INV_PI;
        dpdu = xfVector(sh.to_world, V(-local.y, local.x, 0) * (2 * MI_PI));
        float zrad = sqrtf(local.x * local.x + local.y * local.y), cosPhi = 0, sinPhi = 1;
        if (zrad > 0) { float inv = 1.0f / zrad; cosPhi = local.x * inv; sinPhi = local.y * inv; }
        dpdv = xfVector(sh.to_world, V(local.z * cosPhi, local.z * sinPhi, -sinf(theta) * sh.radius) * MI_PI);
    } else {
        v3 local = xfPoint(sh.to_object, pRay);
        float phi = atan2f(local.y, local.x); if (phi < 0) phi += 2 * MI_PI;
        uvx = phi / (2 * MI_PI); uvy = local.z / sh.length;
        dpdu = xfVector(sh.to_world, V(-local.y, local.x, 0) * (2 * MI_PI)); dpdv = xfVector(sh.to_world, V(0, 0, sh.length));
    }
}
// Shape::fillIntersectionRecord (rectangle.cpp:155-168, disk.cpp:172-200, sphere.cpp:196-245, cylinder.cpp:203-233) + computeShadingFrame
// + wi (skdtree.h:421-427).  Disk: the reference leaves geoFrame unset; defined as the shading normal.
DEV void fillHitAnalytic(const AnalyticD &sh, v3 o, v3 d, float t, float lx, float ly, Hit &h) {
    h.material = sh.material; h.emitter = sh.emitter; h.flags = sh.flags & 14u; h.dist = t; h.uvx = h.uvy = 0;
    v3 p = o + d * t, n, dpdu;
    const uint32_t type = sh.type;
    if (type == MI_SHAPE_RECTANGLE) { n = ld3(sh.n); dpdu = ld3(sh.dpdu); }
    else if (type == MI_SHAPE_DISK) {
        float r = sqrtf(lx * lx + ly * ly), invR = (r == 0) ? 0.0f : (1.0f / r);
        float cosPhi = lx * invR, sinPhi = ly * invR;
        dpdu = r != 0 ? xfVector(sh.to_world, V(cosPhi, sinPhi, 0)) : xfVector(sh.to_world, V(1, 0, 0));
        n = ld3(sh.n);
    } else if (type == MI_SHAPE_SPHERE) {
        const v3 c = ld3(sh.center);
        p = c + normalize(p - c) * sh.radius;
        v3 local = xfVector(sh.to_object, p - c);
        dpdu = xfVector(sh.to_world, V(-local.y, local.x, 0) * (2 * MI_PI));
        n = normalize(p - c);
        if (sh.flags & 1u) n = n * -1.0f;
    } else {
        v3 local = xfPoint(sh.to_object, p);
        dpdu = xfVector(sh.to_world, V(-local.y, local.x, 0) * (2 * MI_PI));
        v3 dpdv = xfVector(sh.to_world, V(0, 0, sh.length));
        n = cross(normalize(dpdu), normalize(dpdv));
        p = p + n * (sh.radius - sqrtf(local.x * local.x + local.y * local.y));
        if (sh.flags & 1u) n = n * -1.0f;
    }
    h.p = p; h.ng = n; h.ns = n;
    h.s = normalize(dpdu - n * dot(n, dpdu));
    h.t = cross(n, h.s);
    v3 md = -d;
    h.wi = V(dot(md, h.s), dot(md, h.t), dot(md, h.ns));
}
// warp.cpp:25-31 squareToUniformSphere, :54-63 squareToUniformCone
DEV v3 uniformSphere(float sx, float sy) {
    float z = 1.0f - 2.0f * sy, r = sqrtf(maxf(1.0f - z * z, 0.0f)), sinPhi, cosPhi;
    sincos2pi(sx, sinPhi, cosPhi);
    return V(r * cosPhi, r * sinPhi, z);
}
DEV v3 uniformCone(float cosCutoff, float sx, float sy) {
    float cosTheta = (1 - sx) + sx * cosCutoff, sinTheta = sqrtf(maxf(1.0f - cosTheta * cosTheta, 0.0f)), sinPhi, cosPhi;
    sincos2pi(sy, sinPhi, cosPhi);
    return V(cosPhi * sinTheta, sinPhi * sinTheta, cosTheta);
}
// Shape::sampleDirect (src/librender/shape.cpp:102-115) over samplePosition of rectangle (rectangle.cpp:215-221), disk (disk.cpp:252-260),
// cylinder (cylinder.cpp:235-250); Sphere::sampleDirect (sphere.cpp:275-346).  pdf = solid-angle density.
DEV void analyticSampleDirect(const AnalyticD &sh, v3 ref, float sx, float sy, v3 &p, v3 &n, v3 &dOut, float &dist, float &pdf) {
    const uint32_t type = sh.type;
    if (type == MI_SHAPE_SPHERE) {
        const float radius = sh.radius; const v3 center = ld3(sh.center);
        v3 refToCenter = center - ref;
        float refDist2 = dot(refToCenter, refToCenter), invRefDist = 1.0f / sqrtf(refDist2);
        float sinAlpha = radius * invRefDist;
        if (sinAlpha < 1 - MI_EPSILON) {
            float cosAlpha = sqrtf(maxf(1.0f - sinAlpha * sinAlpha, 0.0f));
            v3 fn = refToCenter * invRefDist, fs, ft; coordinateSystem(fn, fs, ft);
            v3 c = uniformCone(cosAlpha, sx, sy);
            v3 d = (fs * c.x + ft * c.y) + fn * c.z;
            pdf = (0.5f * MI_INV_PI) / (1 - cosAlpha);
            float projDist = dot(refToCenter, d);
            float baseT = refDist2 / projDist;
            v3 query = ref + d * baseT;
            v3 queryToCenter = center - query;
            float queryDist2 = dot(queryToCenter, queryToCenter), queryProjDist = dot(queryToCenter, d);
            float A = 1.0f, B = -2 * queryProjDist, C = queryDist2 - radius * radius, nearT, farT;
            if (!solveQuadratic(A, B, C, nearT, farT)) nearT = queryProjDist;
            dist = baseT + nearT;
            n = normalize(d * nearT - queryToCenter);
            p = center + n * radius;
            dOut = d;
        } else {
            v3 dl = uniformSphere(sx, sy);
            p = center + dl * radius; n = dl;
            v3 d = p - ref;
            float dist2 = dot(d, d); dist = sqrtf(dist2);
            { float r = 1.0f / dist; d = d * r; }
            dOut = d;
            pdf = sh.inv_area * dist2 / fabsf(dot(d, n));
        }
        if (sh.flags & 1u) n = n * -1.0f;
        return;
    }
    if (type == MI_SHAPE_RECTANGLE) { p = xfPoint(sh.to_world, V(sx * 2 - 1, sy * 2 - 1, 0)); n = ld3(sh.n); }
    else if (type == MI_SHAPE_DISK) { float px, py; diskConcentric(sx, sy, px, py); p = xfPoint(sh.to_world, V(px, py, 0)); n = ld3(sh.n); }
    else {
        float sinTheta, cosTheta; sincos2pi(sy, sinTheta, cosTheta);
        v3 pl = V(cosTheta * sh.radius, sinTheta * sh.radius, sx * sh.length), nl = V(cosTheta, sinTheta, 0.0f);
        if (sh.flags & 1u) nl = nl * -1.0f;
        p = xfPoint(sh.to_world, pl); n = normalize(xfNormal(sh.to_object, nl));
    }
    v3 d = p - ref;
    float distSquared = dot(d, d); dist = sqrtf(distSquared);
    { float r = 1.0f / dist; d = d * r; }
    float dp = fabsf(dot(d, n));
    pdf = sh.inv_area * (dp != 0 ? (distSquared / dp) : 0.0f);
    dOut = d;
}
// Shape::pdfDirect, ESolidAngle (shape.cpp:117-126); Sphere::pdfDirect (sphere.cpp:348-379)
DEV float analyticPdfDirect(const AnalyticD &sh, v3 ref, v3 d, v3 n, float dist) {
    if (sh.type == MI_SHAPE_SPHERE) {
        v3 refToCenter = ld3(sh.center) - ref;
        float invRefDist = 1.0f / sqrtf(dot(refToCenter, refToCenter)), sinAlpha = sh.radius * invRefDist;
        if (sinAlpha < 1 - MI_EPSILON) { float cosAlpha = sqrtf(maxf(1 - sinAlpha * sinAlpha, 0.0f)); return (0.5f * MI_INV_PI) / (1 - cosAlpha); }
        return sh.inv_area * dist * dist / fabsf(dot(d, n));
    }
    return sh.inv_area * (dist * dist) / fabsf(dot(d, n));
}

// ---------------------------------------------------------------------------------------------- rough conductor
// src/bsdfs/roughconductor.cpp:260-416 over src/bsdfs/microfacet.h (isotropic alpha, Beckmann / GGX, visible-normal sampling).
// exp/log/acos/atan2/tan/sin/cos/pow come from the device math library: this BSDF is tolerance-pinned, not bit-pinned.
DEV float fastexpf_(float x) { return (float) exp((double) x); }      // math::fastexp on Linux/x86_64 (include/mitsuba/core/math.h:185-199)
DEV float fastlogf_(float x) { return (float) log((double) x); }
// src/libcore/math.cpp:25-53 erfinv, :55-72 erf
DEV float miErfinv(float x) {
    float w = -fastlogf_((1.0f - x) * (1.0f + x)), p;
    if (w < 5.0f) {
        w = w - 2.5f; p = 2.81022636e-08f; p = 3.43273939e-07f + p * w; p = -3.5233877e-06f + p * w; p = -4.39150654e-06f + p * w;
        p = 0.00021858087f + p * w; p = -0.00125372503f + p * w; p = -0.00417768164f + p * w; p = 0.246640727f + p * w; p = 1.50140941f + p * w;
    } else {
        w = sqrtf(w) - 3.0f; p = -0.000200214257f; p = 0.000100950558f + p * w; p = 0.00134934322f + p * w; p = -0.00367342844f + p * w;
        p = 0.00573950773f + p * w; p = -0.0076224613f + p * w; p = 0.00943887047f + p * w; p = 1.00167406f + p * w; p = 2.83297682f + p * w;
    }
    return p * x;
}
DEV float miErf(float x) {
    const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f, p = 0.3275911f;
    float sign = copysignf(1.0f, x); x = fabsf(x);
    float t = 1.0f / (1.0f + p * x);
    float y = 1.0f - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * fastexpf_(-x * x);
    return sign * y;
}
// microfacet.h:190-237: distr 0 Beckmann, 1 GGX, 2 Phong (isotropic) / Ashikhmin-Shirley (anisotropic); exponents :717-720, interpolation :559-570
DEV float mfPhongExponent(float alpha) { return maxf(2.0f / (alpha * alpha) - 2.0f, 0.0f); }
DEV float mfInterpExponent(float au, float av, v3 v) {
    const float eu = mfPhongExponent(au), ev = mfPhongExponent(av), sinTheta2 = 1.0f - v.z * v.z;
    if (au == av || sinTheta2 <= 0x1p-128f) return eu;
    float invSinTheta2 = 1 / sinTheta2, cosPhi2 = v.x * v.x * invSinTheta2, sinPhi2 = v.y * v.y * invSinTheta2;
    return eu * cosPhi2 + ev * sinPhi2;
}
DEV float mfEval2(uint32_t distr, float au, float av, v3 m) {
    if (m.z <= 0) return 0.0f;
    float cosTheta2 = m.z * m.z;
    float beckmannExponent = ((m.x * m.x) / (au * au) + (m.y * m.y) / (av * av)) / cosTheta2;
    float result;
    if (distr == 0) result = fastexpf_(-beckmannExponent) / (MI_PI * au * av * cosTheta2 * cosTheta2);
    else if (distr == 1) { float root = (1.0f + beckmannExponent) * cosTheta2; result = 1.0f / (MI_PI * au * av * root * root); }
    else result = sqrtf((mfPhongExponent(au) + 2) * (mfPhongExponent(av) + 2)) * MI_INV_TWOPI * powf(m.z, mfInterpExponent(au, av, m));
    if (result * m.z < 1e-20f) result = 0;
    return result;
}
DEV float mfEval(uint32_t distr, float alpha, v3 m) {            // isotropic Beckmann / GGX (roughdielectric, roughplastic)
    if (m.z <= 0) return 0.0f;
    float cosTheta2 = m.z * m.z;
    float beckmannExponent = ((m.x * m.x) / (alpha * alpha) + (m.y * m.y) / (alpha * alpha)) / cosTheta2;
    float result;
    if (distr == 0) result = fastexpf_(-beckmannExponent) / (MI_PI * alpha * alpha * cosTheta2 * cosTheta2);
    else { float root = (1.0f + beckmannExponent) * cosTheta2; result = 1.0f / (MI_PI * alpha * alpha * root * root); }
    if (result * m.z < 1e-20f) result = 0;
    return result;
}
// microfacet.h:476-517 (+ math::hypot2, src/libcore/math.cpp:74-88)
DEV float mfSmithG1(uint32_t distr, float alpha, v3 v, v3 m) {
    if (dot(v, m) * v.z <= 0) return 0.0f;
    float temp = 1 - v.z * v.z;
    float tanTheta = temp <= 0.0f ? 0.0f : fabsf(sqrtf(temp) / v.z);
    if (tanTheta == 0.0f) return 1.0f;
    if (distr != 1u) {                                           // Beckmann, and Phong through the same fit (microfacet.h:489-491)
        float a = 1.0f / (alpha * tanTheta);
        if (a >= 1.6f) return 1.0f;
        float aSqr = a * a;
        return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
    }
    float root = alpha * tanTheta, r;
    if (1.0f > fabsf(root)) { r = root / 1.0f; r = 1.0f * sqrtf(1.0f + r * r); }
    else if (root != 0.0f) { r = 1.0f / root; r = fabsf(root) * sqrtf(1.0f + r * r); }
    else r = 0.0f;
    return 2.0f / (1.0f + r);
}
// microfacet.h:572-700
DEV void mfSampleVisible11(uint32_t distr, float thetaI, float sx, float sy, float &slx, float &sly) {
    const float SQRT_PI_INV = 1 / sqrtf(MI_PI);
    if (distr == 0) {
        if (thetaI < 1e-4f) { float r = sqrtf(-fastlogf_(1.0f - sx)), ph = 2 * MI_PI * sy; const float2 scp_ = glibcSincosf2(ph); slx = r * scp_.y; sly = r * scp_.x; return; }
        float tanThetaI = tanf(thetaI), cotThetaI = 1 / tanThetaI;
        float a = -1, c = miErf(cotThetaI);
        float sample_x = maxf(sx, 1e-6f);
        float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
        float b = c - (1 + c) * powf(1 - sample_x, fit);
        float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * expf(-cotThetaI * cotThetaI));
        int it = 0;
        while (++it < 10) {
            if (!(b >= a && b <= c)) b = 0.5f * (a + c);
            float invErf = miErfinv(b);
            float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * expf(-invErf * invErf)) - sample_x;
            float derivative = normalization * (1 - invErf * tanThetaI);
            if (fabsf(value) < 1e-5f) break;
            if (value > 0) c = b; else a = b;
            b -= value / derivative;
        }
        slx = miErfinv(b);
        sly = miErfinv(2.0f * maxf(sy, 1e-6f) - 1.0f);
    } else {
        if (thetaI < 1e-4f) { float r = sqrtf(maxf(sx / (1 - sx), 0.0f)), ph = 2 * MI_PI * sy; const float2 scp_ = glibcSincosf2(ph); slx = r * scp_.y; sly = r * scp_.x; return; }
        float tanThetaI = tanf(thetaI), a = 1 / tanThetaI;
        float G1 = 2.0f / (1.0f + sqrtf(maxf(1.0f + 1.0f / (a * a), 0.0f)));
        float A = 2.0f * sx / G1 - 1.0f;
        if (fabsf(A) == 1) A -= copysignf(1.0f, A) * MI_EPSILON;
        float tmp = 1.0f / (A * A - 1.0f), B = tanThetaI;
        float D = sqrtf(maxf(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.0f));
        float s1 = B * tmp - D, s2 = B * tmp + D;
        slx = (A < 0.0f || s2 > 1.0f / tanThetaI) ? s1 : s2;
        float S;
        if (sy > 0.5f) { S = 1.0f; sy = 2.0f * (sy - 0.5f); } else { S = -1.0f; sy = 2.0f * (0.5f - sy); }
        float z = (sy * (sy * (sy * (-0.365728915865723f) + 0.790235037209296f) - 0.424965825137544f) + 0.000152998850436920f) /
                  (sy * (sy * (sy * (sy * 0.169507819808272f - 0.397203533833404f) - 0.232500544458471f) + 1.0f) - 0.539825872510702f);
        sly = S * z * sqrtf(1.0f + slx * slx);
    }
}
// microfacet.h:420-466, :469-473
DEV v3 mfSampleVisible(uint32_t distr, float alpha, v3 wi_, float sx, float sy) {
    v3 wi = normalize(V(alpha * wi_.x, alpha * wi_.y, wi_.z));
    float theta = 0, phi = 0;
    if (wi.z < 0.99999f) { theta = acosf(wi.z); phi = atan2f(wi.y, wi.x); }
    const float2 scPhi_ = glibcSincosf2(phi); float sinPhi = scPhi_.x, cosPhi = scPhi_.y;
    float slx, sly; mfSampleVisible11(distr, theta, sx, sy, slx, sly);
    float rx = cosPhi * slx - sinPhi * sly, ry = sinPhi * slx + cosPhi * sly;
    rx *= alpha; ry *= alpha;
    float normalization = 1.0f / sqrtf(rx * rx + ry * ry + 1.0f);
    return V(-rx * normalization, -ry * normalization, normalization);
}
DEV float mfPdfVisible(uint32_t distr, float alpha, v3 wi, v3 m) {
    if (wi.z == 0) return 0.0f;
    return mfSmithG1(distr, alpha, wi, m) * fabsf(dot(wi, m)) * mfEval(distr, alpha, m) / fabsf(wi.z);
}
// src/libcore/util.cpp:741-763, per RGB channel
DEV v3 fresnelConductorExact(float cosThetaI, const float *eta, const float *k) {
    float cosThetaI2 = cosThetaI * cosThetaI, sinThetaI2 = 1 - cosThetaI2, sinThetaI4 = sinThetaI2 * sinThetaI2;
    float out[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float temp1 = eta[i] * eta[i] - k[i] * k[i] - sinThetaI2;
        float a2pb2 = sqrtf(maxf(temp1 * temp1 + k[i] * k[i] * eta[i] * eta[i] * 4, 0.0f));
        float a = sqrtf(maxf((a2pb2 + temp1) * 0.5f, 0.0f));
        float term1 = a2pb2 + cosThetaI2, term2 = a * (2 * cosThetaI);
        float Rs2 = (term1 - term2) / (term1 + term2);
        float term3 = a2pb2 * cosThetaI2 + sinThetaI4, term4 = term2 * sinThetaI2;
        float Rp2 = Rs2 * (term3 - term4) / (term3 + term4);
        out[i] = 0.5f * (Rp2 + Rs2);
    }
    return V(out[0], out[1], out[2]);
}
// microfacet.h:420-473 with separate roughness along the tangent / bitangent; :545-556 projectRoughness
DEV v3 mfSampleVisible2(uint32_t distr, float au, float av, v3 wi_, float sx, float sy) {
    v3 wi = normalize(V(au * wi_.x, av * wi_.y, wi_.z));
    float theta = 0, phi = 0;
    if (wi.z < 0.99999f) { theta = acosf(wi.z); phi = atan2f(wi.y, wi.x); }
    const float2 scPhi_ = glibcSincosf2(phi); float sinPhi = scPhi_.x, cosPhi = scPhi_.y;
    float slx, sly; mfSampleVisible11(distr, theta, sx, sy, slx, sly);
    float rx = cosPhi * slx - sinPhi * sly, ry = sinPhi * slx + cosPhi * sly;
    rx *= au; ry *= av;
    float normalization = 1.0f / sqrtf(rx * rx + ry * ry + 1.0f);
    return V(-rx * normalization, -ry * normalization, normalization);
}
DEV float mfProjectRoughness(float au, float av, v3 v) {
    float invSinTheta2 = 1 / (1.0f - v.z * v.z);
    if (au == av || invSinTheta2 <= 0) return au;
    float cosPhi2 = v.x * v.x * invSinTheta2, sinPhi2 = v.y * v.y * invSinTheta2;
    return sqrtf(cosPhi2 * au * au + sinPhi2 * av * av);
}
DEV float mfSmithG1_2(uint32_t distr, float au, float av, v3 v, v3 m) { return mfSmithG1(distr, mfProjectRoughness(au, av, v), v, m); }
// microfacet.h:722-731 sampleFirstQuadrant (Ashikhmin-Shirley); :286-392 sampleAll (all normals, density D(m) cos(theta_m))
DEV void mfSampleFirstQuadrant(float eu, float ev, float u1, float &phi, float &exponent) {
    phi = atanf(sqrtf((eu + 2.0f) / (ev + 2.0f)) * tanf(MI_PI * u1 * 0.5f));
    const float2 scPhi_ = glibcSincosf2(phi); float sinPhi = scPhi_.x, cosPhi = scPhi_.y;
    exponent = eu * cosPhi * cosPhi + ev * sinPhi * sinPhi;
}
DEV v3 mfSampleAll(uint32_t distr, float au, float av, float sx, float sy, float &pdf) {
    float cosThetaM = 0.0f, sinPhiM, cosPhiM, alphaSqr;
    if (distr <= 1u) {
        if (au == av) { float ph = (2.0f * MI_PI) * sy; const float2 sc_ = glibcSincosf2(ph); sinPhiM = sc_.x; cosPhiM = sc_.y; alphaSqr = au * au; }
        else {
            float phiM = atanf(av / au * tanf(MI_PI + 2 * MI_PI * sy)) + MI_PI * floorf(2 * sy + 0.5f);
            { const float2 sc_ = glibcSincosf2(phiM); sinPhiM = sc_.x; cosPhiM = sc_.y; }
            float cosSc = cosPhiM / au, sinSc = sinPhiM / av; alphaSqr = 1.0f / (cosSc * cosSc + sinSc * sinSc);
        }
        if (distr == 0u) {
            float tanThetaMSqr = alphaSqr * -fastlogf_(1.0f - sx);
            cosThetaM = 1.0f / sqrtf(1.0f + tanThetaMSqr);
            pdf = (1.0f - sx) / (MI_PI * au * av * cosThetaM * cosThetaM * cosThetaM);
        } else {
            float tanThetaMSqr = alphaSqr * sx / (1.0f - sx);
            cosThetaM = 1.0f / sqrtf(1.0f + tanThetaMSqr);
            float temp = 1 + tanThetaMSqr / alphaSqr;
            pdf = MI_INV_PI / (au * av * cosThetaM * cosThetaM * cosThetaM * temp * temp);
        }
    } else {
        const float eu = mfPhongExponent(au), ev = mfPhongExponent(av);
        float phiM, exponent;
        if (au == av) { phiM = (2.0f * MI_PI) * sy; exponent = eu; }
        else if (sy < 0.25f) mfSampleFirstQuadrant(eu, ev, 4 * sy, phiM, exponent);
        else if (sy < 0.5f) { mfSampleFirstQuadrant(eu, ev, 4 * (0.5f - sy), phiM, exponent); phiM = MI_PI - phiM; }
        else if (sy < 0.75f) { mfSampleFirstQuadrant(eu, ev, 4 * (sy - 0.5f), phiM, exponent); phiM += MI_PI; }
        else { mfSampleFirstQuadrant(eu, ev, 4 * (1 - sy), phiM, exponent); phiM = 2 * MI_PI - phiM; }
        { const float2 sc_ = glibcSincosf2(phiM); sinPhiM = sc_.x; cosPhiM = sc_.y; }
        cosThetaM = powf(sx, 1.0f / (exponent + 2.0f));
        pdf = sqrtf((eu + 2.0f) * (ev + 2.0f)) * MI_INV_TWOPI * powf(cosThetaM, exponent + 1.0f);
    }
    if (pdf < 1e-20f) pdf = 0;
    float sinThetaM = sqrtf(maxf(0.0f, 1 - cosThetaM * cosThetaM));
    return V(sinThetaM * cosPhiM, sinThetaM * sinPhiM, cosThetaM);
}
// The roughconductor's MicrofacetDistribution: flags bit1 sampleVisible, bit3 anisotropic (alphaV in reflectance[0]); Phong samples all normals (:141-145)
struct MfD { uint32_t distr; float au, av; bool visible; };
DEV MfD mfOfRoughConductor(const MaterialD &m) {
    MfD d; d.distr = m.distr; d.au = maxf(m.alpha, 1e-4f); d.av = (m.flags & 8u) ? maxf(m.reflectance[0], 1e-4f) : d.au;
    d.visible = (m.flags & 2u) != 0 && m.distr != 2u; return d;
}
DEV float mfdPdf(const MfD &d, v3 wi, v3 m) {                    // microfacet.h:269-274
    if (d.visible) { if (wi.z == 0) return 0.0f; return mfSmithG1_2(d.distr, d.au, d.av, wi, m) * fabsf(dot(wi, m)) * mfEval2(d.distr, d.au, d.av, m) / fabsf(wi.z); }
    return mfEval2(d.distr, d.au, d.av, m) * m.z;
}
DEV MfD mfOfRoughDielectric(const MaterialD &m) {               // alphaV in k[0] when flags bit3 is set
    MfD d; d.distr = m.distr; d.au = maxf(m.alpha, 1e-4f); d.av = (m.flags & 8u) ? maxf(m.k[0], 1e-4f) : d.au;
    d.visible = (m.flags & 2u) != 0 && m.distr != 2u; return d;
}
// roughdielectric.cpp:409-414: Walter et al.'s widened sampling distribution unless visible normals are sampled (scaleAlpha, microfacet.h:180-185)
DEV MfD mfSampling(const MfD &d, float cosThetaI) {
    MfD s = d;
    if (!d.visible) { float f = 1.2f - 0.2f * sqrtf(fabsf(cosThetaI)); s.au *= f; s.av *= f; }
    return s;
}
DEV v3 mfdSample(const MfD &d, v3 wi, float sx, float sy, float &pdf) {      // microfacet.h:239-248
    if (d.visible) { v3 m = mfSampleVisible2(d.distr, d.au, d.av, wi, sx, sy); pdf = mfdPdf(d, wi, m); return m; }
    return mfSampleAll(d.distr, d.au, d.av, sx, sy, pdf);
}
// src/bsdfs/roughconductor.cpp:260-297, :299-324, :373-425
DEV v3 rcEval(const MaterialD &m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return V(0, 0, 0);
    const MfD d = mfOfRoughConductor(m);
    v3 H = normalize(wo + wi);
    float D = mfEval2(d.distr, d.au, d.av, H);
    if (D == 0) return V(0, 0, 0);
    v3 F = fresnelConductorExact(dot(wi, H), m.eta, m.k) * ld3(m.specular);
    float G = mfSmithG1_2(d.distr, d.au, d.av, wi, H) * mfSmithG1_2(d.distr, d.au, d.av, wo, H);
    float model = D * G / (4.0f * wi.z);
    return F * model;
}
DEV float rcPdf(const MaterialD &m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return 0.0f;
    const MfD d = mfOfRoughConductor(m);
    v3 H = normalize(wo + wi);
    if (d.visible) return mfEval2(d.distr, d.au, d.av, H) * mfSmithG1_2(d.distr, d.au, d.av, wi, H) / (4.0f * wi.z);
    return mfdPdf(d, wi, H) / (4 * fabsf(dot(wo, H)));
}
DEV v3 rcSample(const MaterialD &mt, v3 wi, float u, float v, v3 &wo, float &pdf, float &eta) {
    if (wi.z < 0) return V(0, 0, 0);
    const MfD d = mfOfRoughConductor(mt);
    v3 m;
    if (d.visible) { m = mfSampleVisible2(d.distr, d.au, d.av, wi, u, v); pdf = mfdPdf(d, wi, m); }
    else m = mfSampleAll(d.distr, d.au, d.av, u, v, pdf);
    if (pdf == 0) return V(0, 0, 0);
    float c = 2 * dot(wi, m);
    wo = m * c - wi;
    eta = 1.0f;
    if (wo.z <= 0) return V(0, 0, 0);
    v3 F = fresnelConductorExact(dot(wi, m), mt.eta, mt.k) * ld3(mt.specular);
    float weight;
    if (d.visible) weight = mfSmithG1_2(d.distr, d.au, d.av, wo, m);
    else weight = mfEval2(d.distr, d.au, d.av, m) * (mfSmithG1_2(d.distr, d.au, d.av, wi, m) * mfSmithG1_2(d.distr, d.au, d.av, wo, m)) * dot(wi, m) / (pdf * wi.z);
    pdf /= 4.0f * dot(wo, m);
    return F * weight;
}

// ---------------------------------------------------------------------------------------------- smooth conductor / dielectric / plastic
// src/bsdfs/conductor.cpp:212-286, dielectric.cpp:224-342, plastic.cpp:248-453; eval / pdf are the solid-angle-measure versions the integrator
// calls while sampling emitters (delta components contribute 0 there).  MaterialD fields: conductor eta[3], k[3], specular[3]; dielectric
// eta[0] = intIOR / extIOR, specular = specularReflectance, reflectance = specularTransmittance; plastic eta[0], specular, reflectance =
// diffuseReflectance, k[0] = fresnelDiffuseReflectance(1 / eta) (m_fdrInt, plastic.cpp:200), flags bit2 = nonlinear.
#define MI_BSDF_T_ROUGHCONDUCTOR 1u
#define MI_BSDF_T_CONDUCTOR 2u
#define MI_BSDF_T_DIELECTRIC 3u
#define MI_BSDF_T_THINDIELECTRIC 8u
#define MI_BSDF_T_MASK 9u
#define MI_BSDF_T_NULL 13u
#define MI_BSDF_T_PLASTIC 4u
// src/libcore/util.cpp:653-683 fresnelDielectricExt
DEV float fresnelDielectricExt(float cosThetaI_, float &cosThetaT_, float eta) {
    if (eta == 1) { cosThetaT_ = -cosThetaI_; return 0.0f; }
    float scale = (cosThetaI_ > 0) ? 1 / eta : eta, cosThetaTSqr = 1 - (1 - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0f) { cosThetaT_ = 0.0f; return 1.0f; }
    float cosThetaI = fabsf(cosThetaI_), cosThetaT = sqrtf(cosThetaTSqr);
    float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5f * (Rs * Rs + Rp * Rp);
}
DEV v3 conductorSample(const MaterialD &m, v3 wi, v3 &wo, float &pdf, float &eta, bool &delta) {
    if (wi.z <= 0) return V(0, 0, 0);
    wo = V(-wi.x, -wi.y, wi.z); eta = 1.0f; pdf = 1; delta = true;
    return ld3(m.specular) * fresnelConductorExact(wi.z, m.eta, m.k);
}
DEV v3 dielectricSample(const MaterialD &m, v3 wi, float sx, v3 &wo, float &pdf, float &etaOut, bool &delta) {
    const float eta = m.eta[0], invEta = 1 / eta; float cosThetaT;
    float F = fresnelDielectricExt(wi.z, cosThetaT, eta);
    delta = true;
    if (sx <= F) { wo = V(-wi.x, -wi.y, wi.z); etaOut = 1.0f; pdf = F; return ld3(m.specular); }
    float scale_ = -(cosThetaT < 0 ? invEta : eta);
    wo = V(scale_ * wi.x, scale_ * wi.y, cosThetaT);
    etaOut = cosThetaT < 0 ? eta : invEta; pdf = 1 - F;
    float factor = cosThetaT < 0 ? invEta : eta;
    return ld3(m.reflectance) * (factor * factor);
}
DEV float luminance3(v3 c) { return c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f; }
DEV float plasticProbSpecular(const MaterialD &m, float Fi) {       // plastic.cpp:301-304; w = m_specularSamplingWeight (:204-207), derived on the host into eta[1] from the textures' averages
    const float w = m.eta[1];
    return (Fi * w) / (Fi * w + (1 - Fi) * (1 - w));
}
DEV v3 plasticDiffuse(const MaterialD &m) {
    v3 diff = ld3(m.reflectance); const float fdrInt = m.k[0];
    if (m.flags & 4u) return V(diff.x / (1.0f - diff.x * fdrInt), diff.y / (1.0f - diff.y * fdrInt), diff.z / (1.0f - diff.z * fdrInt));
    float r = 1.0f / (1 - fdrInt); return diff * r;
}
DEV v3 plasticEval(const MaterialD &m, v3 wi, v3 wo) {
    if (wo.z <= 0 || wi.z <= 0) return V(0, 0, 0);
    const float eta = m.eta[0], invEta2 = 1 / (eta * eta); float ct;
    float Fi = fresnelDielectricExt(wi.z, ct, eta), Fo = fresnelDielectricExt(wo.z, ct, eta);
    return plasticDiffuse(m) * (MI_INV_PI * wo.z * invEta2 * (1 - Fi) * (1 - Fo));
}
DEV float plasticPdf(const MaterialD &m, v3 wi, v3 wo) {
    if (wo.z <= 0 || wi.z <= 0) return 0.0f;
    float ct, Fi = fresnelDielectricExt(wi.z, ct, m.eta[0]);
    return MI_INV_PI * wo.z * (1 - plasticProbSpecular(m, Fi));
}
DEV v3 plasticSample(const MaterialD &m, v3 wi, float sx, float sy, v3 &wo, float &pdf, float &etaOut, bool &delta) {
    if (wi.z <= 0) return V(0, 0, 0);
    const float eta = m.eta[0], invEta2 = 1 / (eta * eta); float ct;
    float Fi = fresnelDielectricExt(wi.z, ct, eta);
    etaOut = 1.0f;
    float probSpecular = plasticProbSpecular(m, Fi);
    if (sx < probSpecular) {
        wo = V(-wi.x, -wi.y, wi.z); pdf = probSpecular; delta = true;
        float r = 1.0f / probSpecular; return (ld3(m.specular) * Fi) * r;
    }
    wo = cosHemisphere((sx - probSpecular) / (1 - probSpecular), sy);
    float Fo = fresnelDielectricExt(wo.z, ct, eta);
    pdf = (1 - probSpecular) * (MI_INV_PI * wo.z);
    return plasticDiffuse(m) * (invEta2 * (1 - Fi) * (1 - Fo) / (1 - probSpecular));
}

// ---------------------------------------------------------------------------------------------- rough dielectric, diffuse transmitter
// src/bsdfs/roughdielectric.cpp:274-617 over microfacet.h (isotropic alpha, Beckmann / GGX, sampleVisible = true): alpha, distr, eta[0] = intIOR / extIOR,
// specular = specularReflectance, reflectance = specularTransmittance.  sample() takes one more number from the path's sampler to choose between
// reflection and refraction (EUsesSampler, roughdielectric.cpp:480-481).  src/bsdfs/difftrans.cpp:78-120: reflectance = transmittance.
#define MI_BSDF_T_ROUGHDIELECTRIC 5u
#define MI_BSDF_T_DIFFTRANS 6u
DEV float signum_(float v) { return copysignf(1.0f, v); }
DEV v3 rdEval(const MaterialD &m, v3 wi, v3 wo) {
    if (wi.z == 0) return V(0, 0, 0);
    const float etaM = m.eta[0], invEta = 1 / etaM; const MfD d = mfOfRoughDielectric(m); const bool reflect = wi.z * wo.z > 0; v3 H;
    if (reflect) H = normalize(wo + wi);
    else { float eta = wi.z > 0 ? etaM : invEta; H = normalize(wi + wo * eta); }
    H = H * signum_(H.z);
    float D = mfEval2(d.distr, d.au, d.av, H); if (D == 0) return V(0, 0, 0);
    float ct, F = fresnelDielectricExt(dot(wi, H), ct, etaM);
    float G = mfSmithG1_2(d.distr, d.au, d.av, wi, H) * mfSmithG1_2(d.distr, d.au, d.av, wo, H);
    if (reflect) { float value = F * D * G / (4.0f * fabsf(wi.z)); return ld3(m.specular) * value; }
    float eta = wi.z > 0.0f ? etaM : invEta;
    float sqrtDenom = dot(wi, H) + eta * dot(wo, H);
    float value = ((1 - F) * D * G * eta * eta * dot(wi, H) * dot(wo, H)) / (wi.z * sqrtDenom * sqrtDenom);
    float factor = wi.z > 0 ? invEta : etaM;
    return ld3(m.reflectance) * fabsf(value * factor * factor);
}
DEV float rdPdf(const MaterialD &m, v3 wi, v3 wo) {
    const float etaM = m.eta[0], invEta = 1 / etaM; const MfD d = mfSampling(mfOfRoughDielectric(m), wi.z); const bool reflect = wi.z * wo.z > 0; v3 H; float dwh_dwo;
    if (reflect) { H = normalize(wo + wi); dwh_dwo = 1.0f / (4.0f * dot(wo, H)); }
    else { float eta = wi.z > 0 ? etaM : invEta; H = normalize(wi + wo * eta); float sqrtDenom = dot(wi, H) + eta * dot(wo, H); dwh_dwo = (eta * eta * dot(wo, H)) / (sqrtDenom * sqrtDenom); }
    H = H * signum_(H.z);
    float prob = mfdPdf(d, wi * signum_(wi.z), H);
    float ct, F = fresnelDielectricExt(dot(wi, H), ct, etaM);
    prob *= reflect ? F : (1 - F);
    return fabsf(prob * dwh_dwo);
}
DEV v3 rdSample(const MaterialD &mt, v3 wi, float sx, float sy, float extra, v3 &wo, float &pdf, float &etaOut) {
    const float etaM = mt.eta[0], invEta = 1 / etaM; const MfD d = mfOfRoughDielectric(mt);
    v3 wiS = wi * signum_(wi.z);
    float microfacetPDF; v3 m = mfdSample(mfSampling(d, wi.z), wiS, sx, sy, microfacetPDF);
    if (microfacetPDF == 0) return V(0, 0, 0);
    pdf = microfacetPDF;
    float cosThetaT, F = fresnelDielectricExt(dot(wi, m), cosThetaT, etaM);
    bool sampleReflection = true; v3 weight = V(1, 1, 1); float dwh_dwo;
    if (extra > F) { sampleReflection = false; pdf *= 1 - F; } else pdf *= F;
    if (sampleReflection) {
        float c = 2 * dot(wi, m); wo = m * c - wi; etaOut = 1.0f;
        if (wi.z * wo.z <= 0) return V(0, 0, 0);
        weight = weight * ld3(mt.specular);
        dwh_dwo = 1.0f / (4.0f * dot(wo, m));
    } else {
        if (cosThetaT == 0) return V(0, 0, 0);
        float e = cosThetaT < 0 ? 1 / etaM : etaM;                                   // refract(wi, n, eta, cosThetaT), src/libcore/util.cpp:769-774
        wo = m * (dot(wi, m) * e + cosThetaT) - wi * e;
        etaOut = cosThetaT < 0 ? etaM : invEta;
        if (wi.z * wo.z >= 0) return V(0, 0, 0);
        float factor = cosThetaT < 0 ? invEta : etaM;
        weight = weight * (ld3(mt.reflectance) * (factor * factor));
        float sqrtDenom = dot(wi, m) + etaOut * dot(wo, m);
        dwh_dwo = (etaOut * etaOut * dot(wo, m)) / (sqrtDenom * sqrtDenom);
    }
    if (d.visible) weight = weight * mfSmithG1_2(d.distr, d.au, d.av, wo, m);           // roughdielectric.cpp:607-612
    else weight = weight * fabsf(mfEval2(d.distr, d.au, d.av, m) * (mfSmithG1_2(d.distr, d.au, d.av, wi, m) * mfSmithG1_2(d.distr, d.au, d.av, wo, m)) * dot(wi, m) / (microfacetPDF * wi.z));
    pdf *= fabsf(dwh_dwo);
    return weight;
}
DEV v3 dtEval(const MaterialD &m, v3 wi, v3 wo) { if (wi.z * wo.z >= 0) return V(0, 0, 0); return ld3(m.reflectance) * (MI_INV_PI * fabsf(wo.z)); }
DEV float dtPdf(v3 wi, v3 wo) { if (wi.z * wo.z >= 0) return 0.0f; return fabsf(wo.z) * MI_INV_PI; }
DEV v3 dtSample(const MaterialD &m, v3 wi, float sx, float sy, v3 &wo, float &pdf, float &eta) {
    wo = cosHemisphere(sx, sy); if (wi.z > 0) wo.z *= -1;
    eta = 1.0f; pdf = fabsf(wo.z) * MI_INV_PI;
    return ld3(m.reflectance);
}

// ---------------------------------------------------------------------------------------------- rough plastic
// src/bsdfs/roughplastic.cpp:333-500; RoughTransmittance::eval with eta and alpha fixed (src/bsdfs/rtrans.h:183-193, :232) over evalCubicInterp1D
// (src/libcore/spline.cpp:23-60).  alpha, distr, eta[0], specular, reflectance = diffuseReflectance, flag bit2 nonlinear, k[0] = internal diffuse
// transmittance (roughplastic.cpp:372), k[1] / k[2] = offset / length of the external transmittance slice in DScene::material_tables.
#define MI_BSDF_T_ROUGHPLASTIC 7u
DEV float cubicInterp1D(float x, const float *values, uint32_t size, float mn, float mx) {
    if (!(x >= mn && x <= mx)) return 0.0f;
    float t = ((x - mn) * (float) (size - 1)) / (mx - mn);
    uint32_t k = (uint32_t) t; if (k > size - 2) k = size - 2;
    float f0 = values[k], f1 = values[k + 1], d0, d1;
    if (k > 0) d0 = 0.5f * (values[k + 1] - values[k - 1]); else d0 = values[k + 1] - values[k];
    if (k + 2 < size) d1 = 0.5f * (values[k + 2] - values[k]); else d1 = values[k + 1] - values[k];
    t = t - (float) k;
    float t2 = t * t, t3 = t2 * t;
    return (2 * t3 - 3 * t2 + 1) * f0 + (-2 * t3 + 3 * t2) * f1 + (t3 - 2 * t2 + t) * d0 + (t3 - t2) * d1;
}
DEV float rpTransmittance(const DScene &sc, const MaterialD &m, float cosTheta) {
    if (!(cosTheta >= 0)) return 0.0f;
    float warped = powf(fabsf(cosTheta), 0.25f);
    float result = cubicInterp1D(warped, sc.material_tables + (uint32_t) m.k[1], (uint32_t) m.k[2], 0.0f, 1.0f);
    return minf(1.0f, maxf(0.0f, result));
}
DEV float rpProbSpecular(const DScene &sc, const MaterialD &m, float cosThetaI) {
    float probSpecular = 1 - rpTransmittance(sc, m, cosThetaI);
    const float w = m.eta[1];                                        // m_specularSamplingWeight (roughplastic.cpp:244-246), host-derived like the plastic's
    return (probSpecular * w) / (probSpecular * w + (1 - probSpecular) * (1 - w));
}
DEV v3 rpEval(const DScene &sc, const MaterialD &m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return V(0, 0, 0);
    const float eta = m.eta[0], alpha = maxf(m.alpha, 1e-4f), invEta2 = 1.0f / (eta * eta);
    v3 H = normalize(wo + wi);
    float D, G, ct, F = fresnelDielectricExt(dot(wi, H), ct, eta);
    if (m.distr == 2u) { D = mfEval2(2u, alpha, alpha, H); G = mfSmithG1_2(2u, alpha, alpha, wi, H) * mfSmithG1_2(2u, alpha, alpha, wo, H); }   // Phong: roughness -> exponent (microfacet.h:98-110)
    else { D = mfEval(m.distr, alpha, H); G = mfSmithG1(m.distr, alpha, wi, H) * mfSmithG1(m.distr, alpha, wo, H); }
    float value = F * D * G / (4.0f * wi.z);
    v3 result = ld3(m.specular) * value;
    v3 diff = ld3(m.reflectance);
    float T12 = rpTransmittance(sc, m, wi.z), T21 = rpTransmittance(sc, m, wo.z), Fdr = 1 - m.k[0];
    if (m.flags & 4u) diff = V(diff.x / (1.0f - diff.x * Fdr), diff.y / (1.0f - diff.y * Fdr), diff.z / (1.0f - diff.z * Fdr));
    else { float r = 1.0f / (1 - Fdr); diff = diff * r; }
    return result + diff * (MI_INV_PI * wo.z * T12 * T21 * invEta2);
}
DEV float rpPdf(const DScene &sc, const MaterialD &m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return 0.0f;
    const float alpha = maxf(m.alpha, 1e-4f);
    v3 H = normalize(wo + wi);
    float probSpecular = rpProbSpecular(sc, m, wi.z), probDiffuse = 1 - probSpecular;
    float dwh_dwo = 1.0f / (4.0f * dot(wo, H));
    MfD d; d.distr = m.distr; d.au = d.av = alpha; d.visible = (m.flags & 2u) != 0 && m.distr != 2u;      // distr.pdf(wi, H): visible normals or all normals (roughplastic.cpp:432)
    float prob = mfdPdf(d, wi, H);
    float result = prob * dwh_dwo * probSpecular;
    result += probDiffuse * (MI_INV_PI * wo.z);
    return result;
}
DEV v3 rpSample(const DScene &sc, const MaterialD &mt, v3 wi, float sx, float sy, v3 &wo, float &pdf, float &etaOut) {
    if (wi.z <= 0) return V(0, 0, 0);
    const float alpha = maxf(mt.alpha, 1e-4f);
    float probSpecular = rpProbSpecular(sc, mt, wi.z); bool choseSpecular = true;
    if (sy < probSpecular) sy /= probSpecular; else { sy = (sy - probSpecular) / (1 - probSpecular); choseSpecular = false; }
    if (choseSpecular) {
        MfD d; d.distr = mt.distr; d.au = d.av = alpha; d.visible = (mt.flags & 2u) != 0 && mt.distr != 2u; float mpdf;
        v3 m = mfdSample(d, wi, sx, sy, mpdf);                                                // distr.sample(wi, sample) (roughplastic.cpp:483)
        float c = 2 * dot(wi, m); wo = m * c - wi;
        if (wo.z <= 0) return V(0, 0, 0);
    } else wo = cosHemisphere(sx, sy);
    etaOut = 1.0f;
    pdf = rpPdf(sc, mt, wi, wo);
    if (pdf == 0) return V(0, 0, 0);
    float r = 1.0f / pdf; return rpEval(sc, mt, wi, wo) * r;
}

// ---------------------------------------------------------------------------------------------- BSDFs
// src/bsdfs/diffuse.cpp:112-153; src/bsdfs/twosided.cpp:110-190 (flip to the front side).  RC = the scene holds non-diffuse materials
// (rough conductor, conductor, dielectric, plastic): the diffuse-only kernel variants carry none of that code.
// src/bsdfs/thindielectric.cpp:206-258: delta reflection or straight-through transmission (an ENull component: `nullComp`), internal bounces summed
// MI_THIN_SIGNED_COS (a flag the volpath_simple stage sets on its copy of the record): ThinDielectric has two sample() overloads; the one WITHOUT a pdf argument --
// the one volpath_simple calls (volpath_simple.cpp:234) -- feeds the SIGNED cosine to fresnelDielectricExt (thindielectric.cpp:263; the other takes |cos|, :212), so a
// pane hit from its back side reflects as if seen from inside the glass.  Restated as it is.
#define MI_THIN_SIGNED_COS (1u << 30)
DEV v3 thinDielectricSample(const MaterialD &m, v3 wi, float sx, v3 &wo, float &pdf, float &etaOut, bool &delta, bool &nullComp) {
    float ct, R = fresnelDielectricExt((m.flags & MI_THIN_SIGNED_COS) ? wi.z : fabsf(wi.z), ct, m.eta[0]), T = 1 - R;
    if (R < 1) R += T * T * R / (1 - R * R);
    etaOut = 1.0f; delta = true;
    if (sx <= R) { wo = V(-wi.x, -wi.y, wi.z); pdf = R; return ld3(m.specular); }
    nullComp = true; wo = V(-wi.x, -wi.y, -wi.z); pdf = 1 - R; return ld3(m.reflectance);
}
// src/bsdfs/roughdiffuse.cpp:131-261 (Oren-Nayar): alpha = the Beckmann-style roughness, distr = 1: useFastApprox.  `m_alpha->eval(its).average()` of the constant
// texture = ((0 + a) + a + a) * (1 / 3) (spectrum.h:481-486); Frame::sinTheta / cosPhi / sinPhi as in include/mitsuba/core/frame.h:107-154.
#define MI_BSDF_T_ROUGHDIFFUSE 14u
DEV float frameSinTheta(v3 v) { const float t = 1.0f - v.z * v.z; return t <= 0.0f ? 0.0f : sqrtf(t); }
DEV v3 roughDiffuseEval(const MaterialD &m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return V(0, 0, 0);
    const float conversionFactor = 1 / sqrtf(2.0f);
    const float avg = (((0.0f + m.alpha) + m.alpha) + m.alpha) * (1.0f / 3);
    const float sigma = avg * conversionFactor, sigma2 = sigma * sigma;
    const float sinThetaI = frameSinTheta(wi), sinThetaO = frameSinTheta(wo);
    float cosPhiDiff = 0;
    if (sinThetaI > MI_EPSILON && sinThetaO > MI_EPSILON) {
        const float sinPhiI = minf(1.0f, maxf(-1.0f, wi.y / sinThetaI)), cosPhiI = minf(1.0f, maxf(-1.0f, wi.x / sinThetaI));
        const float sinPhiO = minf(1.0f, maxf(-1.0f, wo.y / sinThetaO)), cosPhiO = minf(1.0f, maxf(-1.0f, wo.x / sinThetaO));
        cosPhiDiff = cosPhiI * cosPhiO + sinPhiI * sinPhiO;
    }
    const v3 rho = V(m.reflectance[0], m.reflectance[1], m.reflectance[2]);
    if (m.distr == 1u) {
        const float A = 1.0f - 0.5f * sigma2 / (sigma2 + 0.33f), B = 0.45f * sigma2 / (sigma2 + 0.09f);
        float sinAlpha, tanBeta;
        if (wi.z > wo.z) { sinAlpha = sinThetaO; tanBeta = sinThetaI / wi.z; } else { sinAlpha = sinThetaI; tanBeta = sinThetaO / wo.z; }
        return rho * (MI_INV_PI * wo.z * (A + B * maxf(cosPhiDiff, 0.0f) * sinAlpha * tanBeta));
    }
    const float thetaI = acosf(minf(1.0f, maxf(-1.0f, wi.z))), thetaO = acosf(minf(1.0f, maxf(-1.0f, wo.z)));
    const float alpha = maxf(thetaI, thetaO), beta = minf(thetaI, thetaO);
    float sinAlpha, sinBeta, tanBeta;
    if (wi.z > wo.z) { sinAlpha = sinThetaO; sinBeta = sinThetaI; tanBeta = sinThetaI / wi.z; } else { sinAlpha = sinThetaI; sinBeta = sinThetaO; tanBeta = sinThetaO / wo.z; }
    const float tmp = sigma2 / (sigma2 + 0.09f), tmp2 = (4 * MI_INV_PI * MI_INV_PI) * alpha * beta, tmp3 = 2 * beta * MI_INV_PI;
    const float C1 = 1.0f - 0.5f * sigma2 / (sigma2 + 0.33f); float C2 = 0.45f * tmp; const float C3 = 0.125f * tmp * tmp2 * tmp2, C4 = 0.17f * sigma2 / (sigma2 + 0.13f);
    if (cosPhiDiff > 0) C2 *= sinAlpha; else C2 *= sinAlpha - tmp3 * tmp3 * tmp3;
    const float tanHalf = (sinAlpha + sinBeta) / (sqrtf(maxf(0.0f, 1.0f - sinAlpha * sinAlpha)) + sqrtf(maxf(0.0f, 1.0f - sinBeta * sinBeta)));
    const v3 snglScat = rho * (C1 + cosPhiDiff * C2 * tanBeta + (1.0f - fabsf(cosPhiDiff)) * C3 * tanHalf);
    const v3 dblScat = V(rho.x * rho.x, rho.y * rho.y, rho.z * rho.z) * (C4 * (1.0f - cosPhiDiff * tmp3 * tmp3));
    return (snglScat + dblScat) * (MI_INV_PI * wo.z);
}
DEV float roughDiffusePdf(v3 wi, v3 wo) { if (wi.z <= 0 || wo.z <= 0) return 0.0f; return MI_INV_PI * wo.z; }
DEV v3 roughDiffuseSample(const MaterialD &m, v3 wi, float sx, float sy, v3 &wo, float &pdf, float &eta) {      // roughdiffuse.cpp:239-253: eval / pdf, Spectrum / Float = multiplication by the reciprocal (spectrum.h:415-425)
    if (wi.z <= 0) return V(0, 0, 0);
    wo = cosHemisphere(sx, sy); eta = 1.0f; pdf = MI_INV_PI * wo.z;
    const v3 f = roughDiffuseEval(m, wi, wo); const float recip = 1.0f / pdf;
    return V(f.x * recip, f.y * recip, f.z * recip);
}
// src/bsdfs/phong.cpp:130-256 (modified Phong): reflectance = diffuseReflectance, specular = specularReflectance, alpha = exponent, k[0] = m_specularSamplingWeight
// (configure(), :104-108: luminance of the specular average over the sum of both); two components (glossy 0, diffuse 1), always queried together on this path
#define MI_BSDF_T_PHONG 15u
DEV float phongExponent(const MaterialD &m) { return (((0.0f + m.alpha) + m.alpha) + m.alpha) * (1.0f / 3); }      // m_exponent->eval(its).average()
DEV v3 phongEval(const MaterialD &m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return V(0, 0, 0);
    v3 result = V(0, 0, 0);
    const float alpha = dot(wo, V(-wi.x, -wi.y, wi.z)), exponent = phongExponent(m);
    if (alpha > 0.0f) result = ld3(m.specular) * ((exponent + 2) * MI_INV_TWOPI * powf(alpha, exponent));
    result = result + ld3(m.reflectance) * MI_INV_PI;
    return result * wo.z;
}
DEV float phongPdf(const MaterialD &m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return 0.0f;
    const float diffuseProb = MI_INV_PI * wo.z; float specProb = 0.0f;
    const float alpha = dot(wo, V(-wi.x, -wi.y, wi.z)), exponent = phongExponent(m);
    if (alpha > 0) specProb = powf(alpha, exponent) * (exponent + 1.0f) / (2.0f * MI_PI);
    return m.k[0] * specProb + (1 - m.k[0]) * diffuseProb;
}
DEV v3 phongSample(const MaterialD &m, v3 wi, float sx, float sy, v3 &wo, float &pdf, float &eta) {
    const float w = m.k[0]; bool choseSpecular = true;
    if (sx <= w) sx /= w; else { sx = (sx - w) / (1 - w); choseSpecular = false; }
    if (choseSpecular) {
        const v3 R = V(-wi.x, -wi.y, wi.z); const float exponent = phongExponent(m);
        const float sinAlpha = sqrtf(1 - powf(sy, 2 / (exponent + 1))), cosAlpha = powf(sy, 1 / (exponent + 1)), phi = (2.0f * MI_PI) * sx;
        const float2 sc = glibcSincosf2(phi);
        const v3 local = V(sinAlpha * sc.y, sinAlpha * sc.x, cosAlpha);
        v3 fs, ft; coordinateSystem(R, fs, ft);
        wo = (fs * local.x + ft * local.y) + R * local.z;
        if (wo.z <= 0) return V(0, 0, 0);
    } else wo = cosHemisphere(sx, sy);
    eta = 1.0f; pdf = phongPdf(m, wi, wo);
    if (pdf == 0) return V(0, 0, 0);
    const v3 f = phongEval(m, wi, wo); const float recip = 1.0f / pdf;
    return V(f.x * recip, f.y * recip, f.z * recip);
}
// src/bsdfs/ward.cpp:178-338 (anisotropic Ward in its three variants): reflectance = diffuseReflectance, specular = specularReflectance, alpha = alphaU, k[1] = alphaV,
// distr = variant (0 ward, 1 ward-duer, 2 balanced), k[0] = m_specularSamplingWeight (:160-164).  std::pow(Float, int) promotes to binary64 (so do the factors around it);
// H.z^4 and H.z^3 are single roundings of exact 48-bit squares here.  math::fastexp / fastlog are the binary64 routines (math.h:185-195).
#define MI_BSDF_T_WARD 16u
DEV float avg3(float a) { return (((0.0f + a) + a) + a) * (1.0f / 3); }      // a constant texture's eval(its).average()
DEV float wardExp(v3 H, float alphaU, float alphaV) {
    const float factor2 = H.x / alphaU, factor3 = H.y / alphaV;
    return fastexpf_(-(factor2 * factor2 + factor3 * factor3) / (H.z * H.z));
}
DEV v3 wardEval(const MaterialD &m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return V(0, 0, 0);
    v3 result = V(0, 0, 0);
    const v3 H = wi + wo; const float alphaU = avg3(m.alpha), alphaV = avg3(m.k[1]);
    float factor1;
    if (m.distr == 0u) factor1 = 1.0f / (4.0f * MI_PI * alphaU * alphaV * sqrtf(wi.z * wo.z));
    else if (m.distr == 1u) factor1 = 1.0f / (4.0f * MI_PI * alphaU * alphaV * wi.z * wo.z);
    else { const double z2 = (double) H.z * (double) H.z; factor1 = (float) ((double) dot(H, H) / ((double) (MI_PI * alphaU * alphaV) * (z2 * z2))); }
    const float specRef = factor1 * wardExp(H, alphaU, alphaV);
    if (specRef > 1e-10f) result = ld3(m.specular) * specRef;
    result = result + ld3(m.reflectance) * MI_INV_PI;
    return result * wo.z;
}
DEV float wardPdf(const MaterialD &m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return 0.0f;
    const float alphaU = avg3(m.alpha), alphaV = avg3(m.k[1]);
    const v3 H = normalize(wi + wo);
    const double z2 = (double) H.z * (double) H.z;
    const float factor1 = (float) (1.0 / ((double) (4.0f * MI_PI * alphaU * alphaV * dot(H, wi)) * (z2 * (double) H.z)));
    const float specProb = factor1 * wardExp(H, alphaU, alphaV), diffuseProb = MI_INV_PI * wo.z;
    return m.k[0] * specProb + (1 - m.k[0]) * diffuseProb;
}
DEV v3 wardSample(const MaterialD &m, v3 wi, float sx, float sy, v3 &wo, float &pdf, float &eta) {
    const float w = m.k[0]; bool choseSpecular = true;
    if (sx <= w) sx /= w; else { sx = (sx - w) / (1 - w); choseSpecular = false; }
    if (choseSpecular) {
        const float alphaU = avg3(m.alpha), alphaV = avg3(m.k[1]);
        float phiH = atanf(alphaV / alphaU * tanf(2.0f * MI_PI * sy));
        if (sy > 0.5f) phiH += MI_PI;
        const float2 scPhi = glibcSincosf2(phiH);
        const float cosPhiH = scPhi.y, sinPhiH = sqrtf(maxf(0.0f, 1.0f - cosPhiH * cosPhiH));
        const float thetaH = atanf(sqrtf(maxf(0.0f, -fastlogf_(sx) / ((cosPhiH * cosPhiH) / (alphaU * alphaU) + (sinPhiH * sinPhiH) / (alphaV * alphaV)))));
        const float2 scTheta = glibcSincosf2(thetaH);
        const v3 H = V(scTheta.x * scPhi.y, scTheta.x * scPhi.x, scTheta.y);
        wo = H * (2.0f * dot(wi, H)) - wi;
        if (wo.z <= 0.0f) return V(0, 0, 0);
    } else wo = cosHemisphere(sx, sy);
    eta = 1.0f; pdf = wardPdf(m, wi, wo);
    if (pdf == 0) return V(0, 0, 0);
    const v3 f = wardEval(m, wi, wo); const float recip = 1.0f / pdf;
    return V(f.x * recip, f.y * recip, f.z * recip);
}
template <bool RC> DEV v3 bsdfEval(const DScene &sc, const MaterialD &m, v3 wi, v3 wo) {
    if ((m.flags & 1u) && wi.z < 0) { wi.z = -wi.z; wo.z = -wo.z; }
    if (RC && m.type != 0) {
        if (m.type == MI_BSDF_T_ROUGHCONDUCTOR) return rcEval(m, wi, wo);
        if (m.type == MI_BSDF_T_PLASTIC) return plasticEval(m, wi, wo);
        if (m.type == MI_BSDF_T_ROUGHDIELECTRIC) return rdEval(m, wi, wo);
        if (m.type == MI_BSDF_T_DIFFTRANS) return dtEval(m, wi, wo);
        if (m.type == MI_BSDF_T_ROUGHPLASTIC) return rpEval(sc, m, wi, wo);
        if (m.type == MI_BSDF_T_ROUGHDIFFUSE) return roughDiffuseEval(m, wi, wo);
        if (m.type == MI_BSDF_T_PHONG) return phongEval(m, wi, wo);
        if (m.type == MI_BSDF_T_WARD) return wardEval(m, wi, wo);
        return V(0, 0, 0);
    }
    if (wi.z <= 0 || wo.z <= 0) return V(0, 0, 0);
    float f = MI_INV_PI * wo.z;
    return V(m.reflectance[0] * f, m.reflectance[1] * f, m.reflectance[2] * f);
}
template <bool RC> DEV float bsdfPdf(const DScene &sc, const MaterialD &m, v3 wi, v3 wo) {
    if ((m.flags & 1u) && wi.z < 0) { wi.z = -wi.z; wo.z = -wo.z; }
    if (RC && m.type != 0) {
        if (m.type == MI_BSDF_T_ROUGHCONDUCTOR) return rcPdf(m, wi, wo);
        if (m.type == MI_BSDF_T_PLASTIC) return plasticPdf(m, wi, wo);
        if (m.type == MI_BSDF_T_ROUGHDIELECTRIC) return rdPdf(m, wi, wo);
        if (m.type == MI_BSDF_T_DIFFTRANS) return dtPdf(wi, wo);
        if (m.type == MI_BSDF_T_ROUGHPLASTIC) return rpPdf(sc, m, wi, wo);
        if (m.type == MI_BSDF_T_ROUGHDIFFUSE) return roughDiffusePdf(wi, wo);
        if (m.type == MI_BSDF_T_PHONG) return phongPdf(m, wi, wo);
        if (m.type == MI_BSDF_T_WARD) return wardPdf(m, wi, wo);
        return 0.0f;
    }
    if (wi.z <= 0 || wo.z <= 0) return 0.0f;
    return MI_INV_PI * wo.z;
}
// delta = the sampled component is a Dirac delta (bRec.sampledType & BSDF::EDelta, path.cpp:259-260)
// `extra`: one more value from the path's sampler, drawn by the caller iff bsdfUsesSampler(m) (BSDF::EUsesSampler)
DEV bool bsdfUsesSampler(const MaterialD &m) { return m.type == MI_BSDF_T_ROUGHDIELECTRIC; }
template <bool RC> DEV v3 bsdfSample(const DScene &sc, const MaterialD &m, v3 wi, float u, float v, float extra, v3 &wo, float &pdf, float &eta, bool &delta, bool &nullComp) {
    bool flipped = false; delta = false; nullComp = false;
    if ((m.flags & 1u) && wi.z < 0) { wi.z = -wi.z; flipped = true; }
    if (RC && m.type != 0) {
        v3 w;
        if (m.type == MI_BSDF_T_ROUGHCONDUCTOR) w = rcSample(m, wi, u, v, wo, pdf, eta);
        else if (m.type == MI_BSDF_T_CONDUCTOR) w = conductorSample(m, wi, wo, pdf, eta, delta);
        else if (m.type == MI_BSDF_T_DIELECTRIC) w = dielectricSample(m, wi, u, wo, pdf, eta, delta);
        else if (m.type == MI_BSDF_T_ROUGHDIELECTRIC) w = rdSample(m, wi, u, v, extra, wo, pdf, eta);
        else if (m.type == MI_BSDF_T_DIFFTRANS) w = dtSample(m, wi, u, v, wo, pdf, eta);
        else if (m.type == MI_BSDF_T_ROUGHPLASTIC) w = rpSample(sc, m, wi, u, v, wo, pdf, eta);
        else if (m.type == MI_BSDF_T_ROUGHDIFFUSE) w = roughDiffuseSample(m, wi, u, v, wo, pdf, eta);
        else if (m.type == MI_BSDF_T_PHONG) w = phongSample(m, wi, u, v, wo, pdf, eta);
        else if (m.type == MI_BSDF_T_WARD) w = wardSample(m, wi, u, v, wo, pdf, eta);
        else if (m.type == MI_BSDF_T_THINDIELECTRIC) w = thinDielectricSample(m, wi, u, wo, pdf, eta, delta, nullComp);
        else if (m.type == MI_BSDF_T_NULL) { wo = V(-wi.x, -wi.y, -wi.z); pdf = 1.0f; eta = 1.0f; delta = true; nullComp = true; w = V(1, 1, 1); }      // src/bsdfs/null.cpp:56-66
        else w = plasticSample(m, wi, u, v, wo, pdf, eta, delta);
        if (flipped && !isZero(w) && pdf != 0) wo.z = -wo.z;      // twosided.cpp:176-180
        return w;
    }
    if (wi.z <= 0) return V(0, 0, 0);
    wo = cosHemisphere(u, v); eta = 1.0f; pdf = MI_INV_PI * wo.z;
    if (flipped) wo.z = -wo.z;
    return V(m.reflectance[0], m.reflectance[1], m.reflectance[2]);
}

// ---------------------------------------------------------------------------------------------- BSDF adapters: mixturebsdf, bumpmap, normalmap
// src/bsdfs/mixturebsdf.cpp:115-276 (path tracer: component = -1).  Record: distr = number of children (2..4), their material indices as numbers in
// reflectance[0..2], eta[0], their weights in k[0..2], specular[0]; flags bit0: the mixture sits inside a `twosided`.  Children are plain BSDF records (no
// textures, at most one of them with a delta component: validated by mi_scene_set_materials).  Weights are rescaled when they sum to more than one
// (ensureEnergyConservation), selection probabilities = the normalised running sums of a DiscreteDistribution (include/mitsuba/core/pmf.h:56-58, 103-116).
#define MI_BSDF_T_MIXTURE 10u
#define MI_BSDF_T_BUMPMAP 11u
#define MI_BSDF_T_NORMALMAP 12u
#define MI_BSDF_T_BLEND 18u
struct MixD { int n; float w0, w1, w2, w3, c1, c2, c3; uint32_t i0, i1, i2, i3; bool blend; };      // weights, cdf[1..3] (cdf[0] = 0, cdf[n] = 1), child records
DEV MixD mixOf(const MaterialD &m) {
    if (m.type == MI_BSDF_T_BLEND) {      // blendbsdf.cpp:138-141: weight = clamp(m_weight->eval(its).average(), 0, 1) on the second BSDF, 1 - weight on the first; `reflectance` = the texture's value or (w, w, w)
        MixD b; b.n = 2; b.blend = true; b.w1 = minf(1.0f, maxf(0.0f, (((0.0f + m.reflectance[0]) + m.reflectance[1]) + m.reflectance[2]) * (1.0f / 3))); b.w0 = 1 - b.w1; b.w2 = b.w3 = 0.0f;
        b.i0 = (uint32_t) m.eta[0]; b.i1 = (uint32_t) m.eta[1]; b.i2 = b.i3 = 0u; b.c1 = b.w0; b.c2 = b.c3 = 1.0f;
        return b;
    }
    MixD x; x.blend = false; x.n = (int) m.distr; x.w0 = m.k[0]; x.w1 = m.k[1]; x.w2 = x.n > 2 ? m.k[2] : 0.0f; x.w3 = x.n > 3 ? m.specular[0] : 0.0f;
    x.i0 = (uint32_t) m.reflectance[0]; x.i1 = (uint32_t) m.reflectance[1]; x.i2 = (uint32_t) m.reflectance[2]; x.i3 = (uint32_t) m.eta[0];
    float total = x.w0 + x.w1; if (x.n > 2) total += x.w2; if (x.n > 3) total += x.w3;
    if (total > 1) { const float sc_ = 1.0f / total; x.w0 *= sc_; x.w1 *= sc_; x.w2 *= sc_; x.w3 *= sc_; }
    float c1 = 0.0f + x.w0, c2 = c1 + x.w1, c3 = c2 + x.w2, c4 = c3 + x.w3;
    const float sum = x.n == 2 ? c2 : (x.n == 3 ? c3 : c4), norm = 1.0f / sum;
    c1 *= norm; c2 *= norm; c3 *= norm;
    x.c1 = c1; x.c2 = x.n == 2 ? 1.0f : c2; x.c3 = x.n <= 3 ? 1.0f : c3;
    return x;
}
DEV float mixCdf(const MixD &x, int i) { return i <= 0 ? 0.0f : (i == 1 ? x.c1 : (i == 2 ? x.c2 : (i == 3 ? x.c3 : 1.0f))); }      // cdf[n] = 1 is baked into c2 / c3
DEV float mixProb(const MixD &x, int i) { if (x.blend) return i == 0 ? x.w0 : x.w1; return (i + 1 >= x.n ? 1.0f : mixCdf(x, i + 1)) - mixCdf(x, i); }
DEV float mixWeight(const MixD &x, int i) { return i == 0 ? x.w0 : (i == 1 ? x.w1 : (i == 2 ? x.w2 : x.w3)); }
DEV uint32_t mixChild(const MixD &x, int i) { return i == 0 ? x.i0 : (i == 1 ? x.i1 : (i == 2 ? x.i2 : x.i3)); }
template <bool RC, bool MIX, bool L> DEV v3 mxEval(const DScene &sc, const Tabs<L> &tb, const MaterialD &m, v3 wi, v3 wo) {
    if (!MIX || (m.type != MI_BSDF_T_MIXTURE && m.type != MI_BSDF_T_BLEND)) return bsdfEval<RC>(sc, m, wi, wo);
    if ((m.flags & 1u) && wi.z < 0) { wi.z = -wi.z; wo.z = -wo.z; }
    const MixD x = mixOf(m); v3 r = V(0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i < x.n) r = r + bsdfEval<RC>(sc, loadMaterial(tb, (int) mixChild(x, i)), wi, wo) * mixWeight(x, i);
    return r;
}
template <bool RC, bool MIX, bool L> DEV float mxPdf(const DScene &sc, const Tabs<L> &tb, const MaterialD &m, v3 wi, v3 wo) {
    if (!MIX || (m.type != MI_BSDF_T_MIXTURE && m.type != MI_BSDF_T_BLEND)) return bsdfPdf<RC>(sc, m, wi, wo);
    if ((m.flags & 1u) && wi.z < 0) { wi.z = -wi.z; wo.z = -wo.z; }
    const MixD x = mixOf(m); float r = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i < x.n) r += bsdfPdf<RC>(sc, loadMaterial(tb, (int) mixChild(x, i)), wi, wo) * mixProb(x, i);
    return r;
}
// `extra`: functor drawing one more sampler value, called only if the chosen BSDF asks for it (BSDF::EUsesSampler)
template <bool RC, bool MIX, bool L, typename F> DEV v3 mxSample(const DScene &sc, const Tabs<L> &tb, const MaterialD &m, v3 wi, float u, float v, F extra, v3 &wo, float &pdf, float &eta, bool &delta, bool &nullComp) {
    if (!MIX || (m.type != MI_BSDF_T_MIXTURE && m.type != MI_BSDF_T_BLEND)) { const float e = (RC && bsdfUsesSampler(m)) ? extra() : 0.0f; return bsdfSample<RC>(sc, m, wi, u, v, e, wo, pdf, eta, delta, nullComp); }
    const bool flip = (m.flags & 1u) && wi.z < 0; if (flip) wi.z = -wi.z;
    const MixD x = mixOf(m);
    // m_pdf.sampleReuse(sample.x) (pmf.h:124-190): lower_bound over the cdf = number of entries below u, minus one; zero-probability entries are skipped forward
    int lo = 0;
#pragma unroll
    for (int j = 0; j <= 4; ++j) if (j <= x.n && (j >= x.n ? 1.0f : mixCdf(x, j)) < u) ++lo;
    int entry = lo > 0 ? lo - 1 : 0; if (entry > x.n - 1) entry = x.n - 1;
    if (x.blend) {                                   // blendbsdf.cpp:226-232: `sample.x < weights[0]`, rescaled by the weights themselves
        if (u < x.w0) { entry = 0; u /= x.w0; } else { entry = 1; u = (u - x.w0) / x.w1; }
    } else {
    while (entry < x.n && mixProb(x, entry) == 0) ++entry;
    const float c0 = mixCdf(x, entry), c1 = entry + 1 >= x.n ? 1.0f : mixCdf(x, entry + 1);
    u = (u - c0) / (c1 - c0);
    }
    const MaterialD child = loadMaterial(tb, (int) mixChild(x, entry));
    const float e = bsdfUsesSampler(child) ? extra() : 0.0f;
    v3 result = bsdfSample<RC>(sc, child, wi, u, v, e, wo, pdf, eta, delta, nullComp);
    if (isZero(result)) return result;
    result = result * (mixWeight(x, entry) * pdf); pdf *= mixProb(x, entry);
    if (!delta) {                                    // measure of the sampled component: the other BSDFs contribute in solid angle only
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < x.n && i != entry) {
            const MaterialD other = loadMaterial(tb, (int) mixChild(x, i));
            pdf += bsdfPdf<RC>(sc, other, wi, wo) * mixProb(x, i);
            result = result + bsdfEval<RC>(sc, other, wi, wo) * mixWeight(x, i);
        }
    }
    { const float r = 1.0f / pdf; result = result * r; }
    if (flip) wo.z = -wo.z;
    return result;
}
// ---- smooth dielectric coating (src/bsdfs/coating.cpp:205-372) around the level below.  `coat`: the layer's record (eta[0], alpha = thickness, reflectance = sigmaA, specular,
// k[0] = m_specularSamplingWeight as configure() derives it, filled in at commit); `m`: the nested record.  Queried as the path tracer does: all components, solid angle.
#define MI_BSDF_T_COATING 17u
DEV v3 ctRefractIn(v3 wi, float eta, float invEta, float &R) { float cosThetaT; R = fresnelDielectricExt(fabsf(wi.z), cosThetaT, eta); return V(invEta * wi.x, invEta * wi.y, -copysignf(1.0f, wi.z) * cosThetaT); }
DEV v3 ctRefractOut(v3 wi, float eta, float invEta, float &R) { float cosThetaT; R = fresnelDielectricExt(fabsf(wi.z), cosThetaT, invEta); return V(eta * wi.x, eta * wi.y, -copysignf(1.0f, wi.z) * cosThetaT); }
DEV v3 ctAbsorb(const MaterialD &c, v3 result, v3 wiP, v3 woP) {
    const v3 sigmaA = ld3(c.reflectance) * c.alpha;
    if (isZero(sigmaA)) return result;
    const float f = 1 / fabsf(wiP.z) + 1 / fabsf(woP.z);
    return V(result.x * fastexpf_(-sigmaA.x * f), result.y * fastexpf_(-sigmaA.y * f), result.z * fastexpf_(-sigmaA.z * f));
}
DEV float ctProbSpecular(const MaterialD &c, float R12) { return (R12 * c.k[0]) / (R12 * c.k[0] + (1 - R12) * (1 - c.k[0])); }
// ---- rough dielectric coating (src/bsdfs/roughcoating.cpp:236-443): a microfacet interface (isotropic MicrofacetDistribution) over the nested record; directions enter and
// leave by Snell's law alone (refractTo), the energy balance is the rough-transmittance slice (table at k[1], length k[2]); sample() re-evaluates pdf and value (:432-438)
#define MI_BSDF_T_ROUGHCOATING 19u
DEV v3 rctRefractTo(bool interior, v3 wi, float eta, float invEtaC) {
    const float invEta = interior ? invEtaC : eta; const bool entering = wi.z > 0.0f;
    const float sinThetaTSqr = invEta * invEta * (1.0f - wi.z * wi.z);
    if (sinThetaTSqr >= 1.0f) return V(0, 0, 0);
    const float cosThetaT = sqrtf(1.0f - sinThetaTSqr);
    return V(invEta * wi.x, invEta * wi.y, entering ? cosThetaT : -cosThetaT);
}
DEV MfD rctDistr(const MaterialD &c) { MfD d; d.distr = (uint32_t) c.eta[2]; d.au = d.av = maxf(avg3(c.alpha), 1e-4f); d.visible = (c.flags & 2u) != 0 && d.distr != 2u; return d; }
DEV float rctProbSpecular(const DScene &sc, const MaterialD &c, float cosThetaI) {
    const float p = 1 - rpTransmittance(sc, c, fabsf(cosThetaI)), w = c.k[0];
    return (p * w) / (p * w + (1 - p) * (1 - w));
}
DEV v3 rctAbsorb(const MaterialD &c, v3 result, v3 wiP, v3 woP) {
    const v3 sigmaA = ld3(c.reflectance) * c.eta[1];
    if (isZero(sigmaA)) return result;
    const float f = 1 / fabsf(wiP.z) + 1 / fabsf(woP.z);
    return V(result.x * fastexpf_(-sigmaA.x * f), result.y * fastexpf_(-sigmaA.y * f), result.z * fastexpf_(-sigmaA.z * f));
}
template <bool RC, bool MIX, bool L> DEV v3 rctEval(const DScene &sc, const Tabs<L> &tb, const MaterialD &c, const MaterialD &m, v3 wi, v3 wo) {
    const float eta = c.eta[0], invEta = 1 / eta; const MfD d = rctDistr(c);
    if ((c.flags & 1u) && wi.z < 0) { wi.z = -wi.z; wo.z = -wo.z; }
    v3 result = V(0, 0, 0);
    if (wo.z * wi.z > 0) {
        const v3 H = normalize(wo + wi) * copysignf(1.0f, wo.z); float ct;
        const float D = mfEval2(d.distr, d.au, d.av, H), F = fresnelDielectricExt(fabsf(dot(wi, H)), ct, eta);
        const float G = mfSmithG1_2(d.distr, d.au, d.av, wi, H) * mfSmithG1_2(d.distr, d.au, d.av, wo, H);
        const float value = F * D * G / (4.0f * fabsf(wi.z));
        result = result + ld3(c.specular) * value;
    }
    const v3 wiP = rctRefractTo(true, wi, eta, invEta), woP = rctRefractTo(true, wo, eta, invEta);
    v3 nested = (mxEval<RC, MIX>(sc, tb, m, wiP, woP) * rpTransmittance(sc, c, fabsf(wi.z))) * rpTransmittance(sc, c, fabsf(wo.z));
    nested = rctAbsorb(c, nested, wiP, woP);
    nested = nested * (invEta * invEta * wo.z / woP.z);
    return result + nested;
}
template <bool RC, bool MIX, bool L> DEV float rctPdf(const DScene &sc, const Tabs<L> &tb, const MaterialD &c, const MaterialD &m, v3 wi, v3 wo) {
    const float eta = c.eta[0], invEta = 1 / eta; const MfD d = rctDistr(c);
    if ((c.flags & 1u) && wi.z < 0) { wi.z = -wi.z; wo.z = -wo.z; }
    const v3 H = normalize(wo + wi) * copysignf(1.0f, wo.z);
    const float probSpecular = rctProbSpecular(sc, c, wi.z), probNested = 1 - probSpecular; float result = 0.0f;
    if (wo.z * wi.z > 0) {
        const float dwh_dwo = 1.0f / (4.0f * fabsf(dot(wo, H))), prob = mfdPdf(d, wi, H);
        result = prob * dwh_dwo * probSpecular;
    }
    const v3 wiP = rctRefractTo(true, wi, eta, invEta), woP = rctRefractTo(true, wo, eta, invEta);
    float prob = mxPdf<RC, MIX>(sc, tb, m, wiP, woP);
    prob *= invEta * invEta * wo.z / woP.z;
    result += prob * probNested;
    return result;
}
template <bool RC, bool MIX, bool L, typename F> DEV v3 rctSample(const DScene &sc, const Tabs<L> &tb, const MaterialD &c, const MaterialD &m, v3 wi, float sx, float sy, F extra, v3 &wo, float &pdf, float &etaOut, bool &delta, bool &nullComp) {
    const float eta = c.eta[0], invEta = 1 / eta; const MfD d = rctDistr(c);
    const bool flip = (c.flags & 1u) && wi.z < 0; if (flip) wi.z = -wi.z;
    const float probSpecular = rctProbSpecular(sc, c, wi.z); bool choseSpecular = true;
    if (sy < probSpecular) sy /= probSpecular; else { sy = (sy - probSpecular) / (1 - probSpecular); choseSpecular = false; }
    delta = false; nullComp = false;
    if (choseSpecular) {
        float mpdf; const v3 mm = mfdSample(d, wi, sx, sy, mpdf);
        const float cc = 2 * dot(wi, mm); wo = mm * cc - wi; etaOut = 1.0f;
        if (wo.z * wi.z <= 0) return V(0, 0, 0);
    } else {
        const v3 wiP = rctRefractTo(true, wi, eta, invEta); v3 woP = V(0, 0, 0);
        const v3 r = mxSample<RC, MIX>(sc, tb, m, wiP, sx, sy, extra, woP, pdf, etaOut, delta, nullComp);
        if (isZero(r)) return V(0, 0, 0);
        wo = rctRefractTo(false, woP, eta, invEta);
        if (isZero(wo)) return V(0, 0, 0);
    }
    pdf = rctPdf<RC, MIX>(sc, tb, c, m, wi, wo);
    if (pdf == 0) return V(0, 0, 0);
    const float r = 1.0f / pdf; const v3 result = rctEval<RC, MIX>(sc, tb, c, m, wi, wo) * r;
    if (flip) wo.z = -wo.z;
    return result;
}
// (`coat`: index of the layer's record, -1 = no coating; the record is re-read where it is needed rather than carried through the shade stage: 16 registers)
template <bool RC, bool MIX, bool L> DEV v3 ctEval(const DScene &sc, const Tabs<L> &tb, int coat, const MaterialD &m, v3 wi, v3 wo) {
    if (!MIX || coat < 0) return mxEval<RC, MIX>(sc, tb, m, wi, wo);
    const MaterialD c = loadMaterial(tb, coat);
    if (c.type == MI_BSDF_T_ROUGHCOATING) return rctEval<RC, MIX>(sc, tb, c, m, wi, wo);
    const float eta = c.eta[0], invEta = 1 / eta;
    if ((c.flags & 1u) && wi.z < 0) { wi.z = -wi.z; wo.z = -wo.z; }
    float R12, R21; const v3 wiP = ctRefractIn(wi, eta, invEta, R12), woP = ctRefractIn(wo, eta, invEta, R21);
    if (R12 == 1 || R21 == 1) return V(0, 0, 0);
    v3 result = (mxEval<RC, MIX>(sc, tb, m, wiP, woP) * (1 - R12)) * (1 - R21);
    result = ctAbsorb(c, result, wiP, woP);
    return result * (invEta * invEta * wo.z / woP.z);
}
template <bool RC, bool MIX, bool L> DEV float ctPdf(const DScene &sc, const Tabs<L> &tb, int coat, const MaterialD &m, v3 wi, v3 wo) {
    if (!MIX || coat < 0) return mxPdf<RC, MIX>(sc, tb, m, wi, wo);
    const MaterialD c = loadMaterial(tb, coat);
    if (c.type == MI_BSDF_T_ROUGHCOATING) return rctPdf<RC, MIX>(sc, tb, c, m, wi, wo);
    const float eta = c.eta[0], invEta = 1 / eta;
    if ((c.flags & 1u) && wi.z < 0) { wi.z = -wi.z; wo.z = -wo.z; }
    float R12, R21; const v3 wiP = ctRefractIn(wi, eta, invEta, R12); const float probSpecular = ctProbSpecular(c, R12);
    const v3 woP = ctRefractIn(wo, eta, invEta, R21);
    if (R12 == 1 || R21 == 1) return 0.0f;
    float pdf = mxPdf<RC, MIX>(sc, tb, m, wiP, woP);
    pdf *= invEta * invEta * wo.z / woP.z;
    return pdf * (1 - probSpecular);
}
template <bool RC, bool MIX, bool L, typename F> DEV v3 ctSample(const DScene &sc, const Tabs<L> &tb, int coat, const MaterialD &m, v3 wi, float u, float v, F extra, v3 &wo, float &pdf, float &etaOut, bool &delta, bool &nullComp) {
    if (!MIX || coat < 0) return mxSample<RC, MIX>(sc, tb, m, wi, u, v, extra, wo, pdf, etaOut, delta, nullComp);
    const MaterialD c = loadMaterial(tb, coat);
    if (c.type == MI_BSDF_T_ROUGHCOATING) return rctSample<RC, MIX>(sc, tb, c, m, wi, u, v, extra, wo, pdf, etaOut, delta, nullComp);
    const float eta = c.eta[0], invEta = 1 / eta;
    const bool flip = (c.flags & 1u) && wi.z < 0; if (flip) wi.z = -wi.z;
    float R12; const v3 wiP = ctRefractIn(wi, eta, invEta, R12); const float probSpecular = ctProbSpecular(c, R12);
    v3 result;
    if (u < probSpecular) {
        wo = V(-wi.x, -wi.y, wi.z); etaOut = 1.0f; pdf = probSpecular; delta = true; nullComp = false;
        result = ld3(c.specular) * (R12 / pdf);
    } else {
        u = (u - probSpecular) / (1 - probSpecular);
        if (R12 == 1.0f) return V(0, 0, 0);
        v3 woP = V(0, 0, 0);
        result = mxSample<RC, MIX>(sc, tb, m, wiP, u, v, extra, woP, pdf, etaOut, delta, nullComp);
        if (isZero(result)) return V(0, 0, 0);
        result = ctAbsorb(c, result, wiP, woP);
        float R21; wo = ctRefractOut(woP, eta, invEta, R21);
        if (R21 == 1.0f) return V(0, 0, 0);
        pdf *= 1.0f - probSpecular; { const float r = 1.0f / (1.0f - probSpecular); result = result * r; }
        result = result * ((1 - R12) * (1 - R21));
        if (!delta) pdf *= invEta * invEta * wo.z / woP.z;
    }
    if (flip) wo.z = -wo.z;
    return result;
}
// bumpmap / normalmap (src/bsdfs/bumpmap.cpp:140-250, normalmap.cpp:108-260): the nested BSDF runs in a perturbed shading frame (ps, pt, pn); directions whose
// cosines differ in sign between the hit's frame and the perturbed one are rejected.  Texture2D::evalGradient (src/librender/texture.cpp:123-141): finite
// differences over eps = 1e-4 for procedural textures, the bilinear gradient of MIP level 0 for bitmaps (bitmap.cpp:459-483, mipmap.h:602-627).
DEV void mipGradientBilinear(const DScene &sc, const TextureD &t, float uvx, float uvy, v3 &gu, v3 &gv) {
    gu = gv = V(0, 0, 0);
    if (!isfinite(uvx) || !isfinite(uvy)) return;
    const uint32_t *L = sc.tex_levels + t.first_level * 3u; const float sx = (float) (int) L[0], sy = (float) (int) L[1];
    float u = uvx * sx - 0.5f, v = uvy * sy - 0.5f;
    int xPos = (int) floorf(u), yPos = (int) floorf(v);
    float dx = u - (float) xPos, dy = v - (float) yPos;
    v3 p00 = mipTexel(sc, t, 0, xPos, yPos), p10 = mipTexel(sc, t, 0, xPos + 1, yPos), p01 = mipTexel(sc, t, 0, xPos, yPos + 1), p11 = mipTexel(sc, t, 0, xPos + 1, yPos + 1);
    v3 tmp = (p01 + p10) - p11;
    gu = ((p10 + p00 * (dy - 1)) - tmp * dy) * sx;
    gv = ((p01 + p00 * (dx - 1)) - tmp * dx) * sy;
}
DEV void textureGradient(const DScene &sc, const TextureD &t, float u, float v, v3 &gu, v3 &gv) {
    const float uvx = u * t.uscale + t.uoffset, uvy = v * t.vscale + t.voffset;
    if (t.type == 2u) { if (t.filter != 0u) mipGradientBilinear(sc, t, uvx, uvy, gu, gv); else gu = gv = V(0, 0, 0); }
    else {
        TextureD raw = t; raw.uscale = raw.vscale = 1.0f; raw.uoffset = raw.voffset = 0.0f;
        const float eps = MI_EPSILON;
        v3 value = textureEval(raw, uvx, uvy), valueU = textureEval(raw, uvx + eps, uvy), valueV = textureEval(raw, uvx, uvy + eps);
        gu = (valueU - value) * (1 / eps); gv = (valueV - value) * (1 / eps);
    }
    gu = gu * t.uscale; gv = gv * t.vscale;
}
DEV void perturbFrame(const DScene &sc, const MaterialD &m, const Hit &h, float uvx, float uvy, v3 dpdu, v3 dpdv, v3 &ps, v3 &pt, v3 &pn) {
    const TextureD &tx = sc.textures[((m.flags >> 8) & 0xFFFFu) - 1u];
    if (m.type == MI_BSDF_T_BUMPMAP) {
        v3 gu, gv; textureGradient(sc, tx, uvx, uvy, gu, gv);
        gu = gu * m.alpha; gv = gv * m.alpha;                                 // ScaleTexture::evalGradient (src/textures/scale.cpp:93-97)
        const float dDispDu = luminance3(gu), dDispDv = luminance3(gv);
        v3 du = dpdu + h.ns * (dDispDu - dot(h.ns, dpdu)), dv = dpdv + h.ns * (dDispDv - dot(h.ns, dpdv));
        v3 n = normalize(cross(du, dv));
        ps = normalize(du - n * dot(n, du)); pt = cross(n, ps);
        if (dot(n, h.ng) < 0) n = n * -1.0f;
        pn = n;
    } else {
        v3 c = tx.type == 2u ? (tx.filter != 0u ? mipBilinear(sc, tx, 0, uvx * tx.uscale + tx.uoffset, uvy * tx.vscale + tx.voffset) : mipBox(sc, tx, 0, uvx * tx.uscale + tx.uoffset, uvy * tx.vscale + tx.voffset)) : textureEval(tx, uvx, uvy);
        v3 nl = V(2 * c.x - 1, 2 * c.y - 1, 2 * c.z - 1);
        v3 n = normalize((h.s * nl.x + h.t * nl.y) + h.ns * nl.z);
        ps = normalize(dpdu - n * dot(n, dpdu)); pt = cross(n, ps); pn = n;
    }
}
DEV v3 frameToLocal(v3 fs, v3 ft, v3 fn, v3 w) { return V(dot(w, fs), dot(w, ft), dot(w, fn)); }
DEV v3 frameToWorld(v3 fs, v3 ft, v3 fn, v3 w) { return (fs * w.x + ft * w.y) + fn * w.z; }

// ---------------------------------------------------------------------------------------------- environment emitter
// src/emitters/envmap.cpp (level-0 bilinear lookups only; the reference EWA-filters camera rays that see the sky directly, :398-411).
// atan2/acos/sin/cos come from the device math library: tolerance-pinned like the rough conductor.
DEV v3 mat3(const float *m, v3 v) { return V(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z); }
DEV float luminance(v3 c) { return c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f; }
// include/mitsuba/render/mipmap.h:504-560 evalTexel (u: ERepeat, v: EClamp)
DEV v3 envTexel(const DScene &sc, int x, int y) {
    if (x < 0 || x >= sc.env_w) { int r = x % sc.env_w; x = r < 0 ? r + sc.env_w : r; }
    y = y < 0 ? 0 : (y >= sc.env_h ? sc.env_h - 1 : y);
    return ld3(sc.env_rgb + ((size_t) y * sc.env_w + x) * 3);
}
// The device library's atan2f / acosf / sinf / cosf as real calls (not inlined): their expansions are register-hungry and sit inside the environment-map code of k_shade<ENV>
DEV float2 miSinCosf(float x) { return glibcSincosf2(x); }
// mipmap.h:576-597 evalBilinear(0, uv)
DEV v3 envBilinear(const DScene &sc, float uvx, float uvy) {
    if (!isfinite(uvx) || !isfinite(uvy)) return V(0, 0, 0);
    float u = uvx * (float) sc.env_w - 0.5f, v = uvy * (float) sc.env_h - 0.5f;
    int xPos = (int) floorf(u), yPos = (int) floorf(v);
    float dx1 = u - (float) xPos, dx2 = 1.0f - dx1, dy1 = v - (float) yPos, dy2 = 1.0f - dy1;
    v3 r = (envTexel(sc, xPos, yPos) * dx2) * dy2;
    r = r + (envTexel(sc, xPos, yPos + 1) * dx2) * dy1;
    r = r + (envTexel(sc, xPos + 1, yPos) * dx1) * dy2;
    r = r + (envTexel(sc, xPos + 1, yPos + 1) * dx1) * dy1;
    return r;
}
// envmap.cpp:384-416 evalEnvironment, no differentials
DEV v3 envEval(const DScene &sc, v3 d) {
    if (sc.env_constant) return ld3(sc.emitters[sc.env_index].radiance);      // ConstantBackgroundEmitter::evalEnvironment (constant.cpp:244-246)
    v3 v = mat3(sc.env_to_local, d);
    float uvx = miAtan2f(v.x, -v.z) * MI_INV_TWOPI, uvy = miAcosf(minf(1.0f, maxf(-1.0f, v.y))) * MI_INV_PI;
    return envBilinear(sc, uvx, uvy) * sc.env_scale;
}
// envmap.cpp:664-669 sampleReuse
DEV uint32_t envSampleReuse(const float *cdf, uint32_t size, float &sample) {
    uint32_t lo = 0, hi = size + 1;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (cdf[mid] < sample) lo = mid + 1; else hi = mid; }
    uint32_t index = lo > 0 ? lo - 1 : 0; if (index > size - 1) index = size - 1;
    sample = (sample - cdf[index]) / (cdf[index + 1] - cdf[index]);
    return index;
}
// the same search started from a guide table: lower_bound(cdf, sample) lies in [guide[b], guide[b + 1]] for b = floor(sample * K) -- identical index, fewer dependent loads
DEV uint32_t envSampleReuseG(const float *cdf, uint32_t size, const uint16_t *guide, uint32_t K, float &sample) {
    uint32_t b = (uint32_t) (sample * (float) K); if (b >= K) b = K - 1u;
    uint32_t lo = guide[b], hi = guide[b + 1u];
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (cdf[mid] < sample) lo = mid + 1; else hi = mid; }
    uint32_t index = lo > 0 ? lo - 1 : 0; if (index > size - 1) index = size - 1;
    sample = (sample - cdf[index]) / (cdf[index + 1] - cdf[index]);
    return index;
}
DEV float intervalToTent(float sample) {               // src/libcore/warp.cpp:142-155
    float sign;
    if (sample < 0.5f) { sign = 1; sample *= 2; } else { sign = -1; sample = 2 * (sample - 0.5f); }
    return sign * (1 - sqrtf(sample));
}
DEV void envBilinearPair(const DScene &sc, float px, float py, v3 &value1, v3 &value2, int &yPos) {
    int xPos = (int) floorf(px); yPos = (int) floorf(py);
    float dx1 = px - (float) xPos, dx2 = 1.0f - dx1, dy1 = py - (float) yPos, dy2 = 1.0f - dy1;
    value1 = (envTexel(sc, xPos, yPos) * dx2) * dy2 + (envTexel(sc, xPos + 1, yPos) * dx1) * dy2;
    value2 = (envTexel(sc, xPos, yPos + 1) * dx2) * dy1 + (envTexel(sc, xPos + 1, yPos + 1) * dx1) * dy1;
}
DEV int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// envmap.cpp:571-608 internalSampleDirection
DEV void envSampleDirection(const DScene &sc, float sx, float sy, v3 &d, v3 &value, float &pdf) {
    uint32_t row, col;
    if (sc.env_guide_rows) {
        row = envSampleReuseG(sc.env_cdf_rows, (uint32_t) sc.env_h, sc.env_guide_rows_t, sc.env_guide_rows, sy);
        col = envSampleReuseG(sc.env_cdf_cols + (size_t) row * (sc.env_w + 1), (uint32_t) sc.env_w, sc.env_guide_cols_t + (size_t) row * (sc.env_guide_cols + 1u), sc.env_guide_cols, sx);
    } else {
        row = envSampleReuse(sc.env_cdf_rows, (uint32_t) sc.env_h, sy);
        col = envSampleReuse(sc.env_cdf_cols + (size_t) row * (sc.env_w + 1), (uint32_t) sc.env_w, sx);
    }
    float px = (float) col + intervalToTent(sx), py = (float) row + intervalToTent(sy);
    v3 v1, v2; int yPos; envBilinearPair(sc, px, py, v1, v2, yPos);
    value = (v1 + v2) * sc.env_scale;
    pdf = (luminance(v1) * sc.env_row_weights[clampi(yPos, 0, sc.env_h - 1)] + luminance(v2) * sc.env_row_weights[clampi(yPos + 1, 0, sc.env_h - 1)]) * sc.env_normalization;
    float ph = sc.env_pixel_w * (px + 0.5f), th = sc.env_pixel_h * (py + 0.5f);
    const float2 scp = miSinCosf(ph), sct = miSinCosf(th); const float sinPhi = scp.x, cosPhi = scp.y, sinTheta = sct.x, cosTheta = sct.y;
    d = V(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta);
    pdf /= maxf(fabsf(sinTheta), MI_EPSILON);
}
// envmap.cpp:611-638 internalPdfDirection
DEV float envPdfDirection(const DScene &sc, v3 d) {
    float uvx = miAtan2f(d.x, -d.z) * MI_INV_TWOPI, uvy = miAcosf(minf(1.0f, maxf(-1.0f, d.y))) * MI_INV_PI;
    if (!isfinite(uvx) || !isfinite(uvy)) return 0.0f;
    float u = uvx * (float) sc.env_w - 0.5f, v = uvy * (float) sc.env_h - 0.5f;
    v3 v1, v2; int yPos; envBilinearPair(sc, u, v, v1, v2, yPos);
    float sinTheta = sqrtf(maxf(1 - d.y * d.y, 0.0f));
    return (luminance(v1) * sc.env_row_weights[clampi(yPos, 0, sc.env_h - 1)] + luminance(v2) * sc.env_row_weights[clampi(yPos + 1, 0, sc.env_h - 1)])
           * sc.env_normalization / maxf(fabsf(sinTheta), MI_EPSILON);
}
// evalEnvironment + pdfDirect of ONE direction (a BSDF-sampled ray that left the scene: path.cpp:234-264 asks for both): the lat-long coordinates (atan2f, acosf) and
// the four texels are the same in envEval and envPdfDirection -- computed / fetched once here, each result then in its own routine's operation order (bit-identical).
DEV void envEvalAndPdf(const DScene &sc, v3 dWorld, v3 &value, float &pdfSA) {
    const v3 d = mat3(sc.env_to_local, dWorld);
    const float uvx = miAtan2f(d.x, -d.z) * MI_INV_TWOPI, uvy = miAcosf(minf(1.0f, maxf(-1.0f, d.y))) * MI_INV_PI;
    if (!isfinite(uvx) || !isfinite(uvy)) { value = V(0, 0, 0); pdfSA = 0.0f; return; }
    const float u = uvx * (float) sc.env_w - 0.5f, v = uvy * (float) sc.env_h - 0.5f;
    const int xPos = (int) floorf(u), yPos = (int) floorf(v);
    const float dx1 = u - (float) xPos, dx2 = 1.0f - dx1, dy1 = v - (float) yPos, dy2 = 1.0f - dy1;
    const v3 t00 = envTexel(sc, xPos, yPos), t01 = envTexel(sc, xPos, yPos + 1), t10 = envTexel(sc, xPos + 1, yPos), t11 = envTexel(sc, xPos + 1, yPos + 1);
    v3 r = (t00 * dx2) * dy2; r = r + (t01 * dx2) * dy1; r = r + (t10 * dx1) * dy2; r = r + (t11 * dx1) * dy1;      // envBilinear
    value = r * sc.env_scale;
    const v3 v1 = (t00 * dx2) * dy2 + (t10 * dx1) * dy2, v2 = (t01 * dx2) * dy1 + (t11 * dx1) * dy1;               // envBilinearPair
    const float sinTheta = sqrtf(maxf(1 - d.y * d.y, 0.0f));
    pdfSA = (luminance(v1) * sc.env_row_weights[clampi(yPos, 0, sc.env_h - 1)] + luminance(v2) * sc.env_row_weights[clampi(yPos + 1, 0, sc.env_h - 1)])
            * sc.env_normalization / maxf(fabsf(sinTheta), MI_EPSILON);
}
// include/mitsuba/core/bsphere.h:88-95 + src/libcore/util.cpp:449-487 solveQuadratic
DEV bool bsphereIntersect(const DScene &sc, v3 ro, v3 rd, float &nearT, float &farT) {
    v3 o = ro - ld3(sc.env_bs_center);
    float A = dot(rd, rd), B = 2 * dot(o, rd), C = dot(o, o) - sc.env_bs_radius * sc.env_bs_radius;
    if (A == 0) { if (B != 0) { nearT = farT = -C / B; return true; } return false; }
    float discrim = B * B - 4.0f * A * C;
    if (discrim < 0) return false;
    float temp, sqrtDiscrim = sqrtf(discrim);
    if (B < 0) temp = -0.5f * (B - sqrtDiscrim); else temp = -0.5f * (B + sqrtDiscrim);
    float x0 = temp / A, x1 = C / temp;
    if (x0 > x1) { float t = x0; x0 = x1; x1 = t; }
    nearT = x0; farT = x1; return true;
}

// ---------------------------------------------------------------------------------------------- emitters
// include/mitsuba/core/pmf.h:124-137 DiscreteDistribution::sample
template <typename P>
DEV uint32_t cdfSample(P cdf, uint32_t n, float x) {
    uint32_t lo = 0, hi = n + 1;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (cdf[mid] < x) lo = mid + 1; else hi = mid; }
    uint32_t index = lo > 0 ? lo - 1 : 0; if (index > n - 1) index = n - 1;
    while (cdf[index + 1] - cdf[index] == 0 && index < n) ++index;
    return index;
}
// The same search and the two entries around its result (the caller's sample reuse) for tables of up to three bins -- emitter selection and the triangles of a quad light in
// most scenes -- from ONE round trip: cdf[0..n] is fetched whole and lower_bound is the number of entries below x (the table is non-decreasing).  The binary search and the
// reuse after it are five dependent loads, and in the shade stage every dependent load is a latency the resident waves cannot cover.  `whole` is uniform over
// the launch (DScene::search_flags): with light meshes of both kinds in one scene the lanes of a wave would run both searches one after the other (Veach-MIS: -2 %).
template <typename P>
DEV uint32_t cdfSampleReuse(P cdf, uint32_t n, float x, float &a0, float &a1, bool whole) {
    if (whole && n <= 3u) {
        const float c0 = cdf[0], c1 = cdf[1], c2 = cdf[n >= 2u ? 2u : n], c3 = cdf[n >= 3u ? 3u : n];
        const uint32_t lo = (c0 < x ? 1u : 0u) + (c1 < x ? 1u : 0u) + ((n >= 2u && c2 < x) ? 1u : 0u) + ((n >= 3u && c3 < x) ? 1u : 0u);
        uint32_t index = lo > 0 ? lo - 1 : 0; if (index > n - 1) index = n - 1;
        if (index == 0u && c1 - c0 == 0) index = 1u;
        if (index == 1u && n > 1u && c2 - c1 == 0) index = 2u;
        if (index == 2u && n > 2u && c3 - c2 == 0) index = 3u;
        if (index >= n) { a0 = cdf[index]; a1 = cdf[index + 1]; return index; }      // (a table that ends in empty bins: what the loop below reads)
        a0 = index == 0u ? c0 : index == 1u ? c1 : c2; a1 = index == 0u ? c1 : index == 1u ? c2 : c3;
        return index;
    }
    const uint32_t index = cdfSample(cdf, n, x);
    a0 = cdf[index]; a1 = cdf[index + 1];
    return index;
}
struct Direct { v3 p, n, d; float dist, pdf; float em_pdf; /* probability of the chosen emitter (RAW) */ int emitter; bool delta; /* !isOnSurface: point / spot / directional */ };
// src/emitters/area.cpp:106-111
template <bool L>
DEV v3 emitterEval(const Tabs<L> &tb, int e, v3 ns, v3 d) {
    if (dot(ns, d) <= 0) return V(0, 0, 0);
    f4 a = tb.emitters4[e * 3]; return V(a.x, a.y, a.z);
}
// Scene::sampleEmitterDirect (src/librender/scene.cpp:860-884) without the visibility test (the shadow queue does it)
// -> AreaLight::sampleDirect (src/emitters/area.cpp:160-176) -> Shape::sampleDirect (src/librender/shape.cpp:102-115)
// -> TriMesh::samplePosition (src/librender/trimesh.cpp:413-425) -> Triangle::sample (src/libcore/triangle.cpp:24-59)
// RAW (Scene::sampleAttenuatedEmitterDirect, scene.cpp:886-931): the value is NOT yet divided by the emitter-selection probability (dr.em_pdf): the transmittance joins it first
template <bool ENV, bool AN, bool L, bool RAW = false>
DEV v3 sampleEmitterDirect(const DScene &sc, const Tabs<L> &tb, v3 ref, v3 refN, float sx, float sy, Direct &dr) {
    float c0, c1; const uint32_t ei = cdfSampleReuse(tb.emitter_cdf, sc.n_emitters, sx, c0, c1, (sc.search_flags & 1u) != 0);
    float emPdf = c1 - c0;
    sx = (sx - c0) / (c1 - c0);
    const EmitterD em = loadEmitter(tb, (int) ei);
    dr.delta = false;
    if ((ENV || AN) && em.type >= 2) {
        v3 value = V(0, 0, 0); dr.pdf = 0.0f;
        if (ENV && em.type == 2) {                               // ConstantBackgroundEmitter::sampleDirect (constant.cpp:175-217)
            v3 d; float pdf, nearT, farT;
            if (!isZero(refN)) {
                v3 l = cosHemisphere(sx, sy); pdf = MI_INV_PI * l.z;
                v3 fs, ft; coordinateSystem(refN, fs, ft);
                d = (fs * l.x + ft * l.y) + refN * l.z;
            } else { d = uniformSphere(sx, sy); pdf = MI_INV_FOURPI; }
            if (!bsphereIntersect(sc, ref, d, nearT, farT)) return V(0, 0, 0);
            if (!(nearT < 0 && farT > 0)) return V(0, 0, 0);
            dr.p = ref + d * farT; dr.n = normalize(ld3(sc.env_bs_center) - dr.p); dr.d = d; dr.dist = farT; dr.pdf = pdf;
            if (!isZero(refN) && dot(d, refN) <= 0) value = V(0, 0, 0);          // pdf stays non-zero: the shadow ray is still traced
            else { float r = 1.0f / pdf; value = V(em.radiance[0] * r, em.radiance[1] * r, em.radiance[2] * r); }
        } else if (AN && (em.type == 3 || em.type == 4)) {       // PointEmitter / SpotEmitter::sampleDirect (point.cpp:133-149, spot.cpp:108-128, 187-203)
            const float *x = sc.emitter_x + ei * 16u;
            dr.delta = true;
            dr.p = V(x[0], x[1], x[2]);
            dr.d = dr.p - ref; dr.dist = sqrtf(dot(dr.d, dr.d));
            float invDist = 1.0f / dr.dist; dr.d = dr.d * invDist; dr.n = V(0, 0, 0); dr.pdf = 1.0f;
            v3 I = V(em.radiance[0], em.radiance[1], em.radiance[2]);
            if (em.type == 4) {
                v3 local = mat3(x + 4, -dr.d); float cosTheta = local.z, f;
                if (cosTheta <= x[3]) f = 0.0f;
                else if (cosTheta >= x[13]) f = 1.0f;
                else f = (x[14] - acosf(cosTheta)) * x[15];
                I = I * f;
            }
            value = I * (invDist * invDist);
        } else if (AN && em.type == 5) {                         // DirectionalEmitter::sampleDirect (directional.cpp:159-180)
            const float *x = sc.emitter_x + ei * 16u;
            dr.delta = true;
            v3 d = V(x[0], x[1], x[2]);
            v3 diskCenter = ld3(sc.dir_bs_center) - d * sc.dir_bs_radius;
            float distance = dot(ref - diskCenter, d);
            if (distance < 0) return V(0, 0, 0);
            dr.p = ref - d * distance; dr.d = -d; dr.n = d; dr.dist = distance; dr.pdf = 1.0f;
            value = V(em.radiance[0], em.radiance[1], em.radiance[2]);
        }
        // (type 7, CollimatedBeamEmitter::sampleDirect, collimated.cpp:129-133: "direct sampling always fails for a response function on a 0D space" -- pdf stays 0)
        if (dr.pdf != 0) {
            dr.emitter = (int) ei; dr.pdf *= emPdf;
            if (RAW) dr.em_pdf = emPdf; else { float r = 1.0f / emPdf; value = value * r; }
            return value;
        }
        return V(0, 0, 0);
    }
    if (ENV && em.type == 1) {                                   // EnvironmentMap::sampleDirect (envmap.cpp:520-547)
        v3 value, dl; float pdf, nearT, farT;
        envSampleDirection(sc, sx, sy, dl, value, pdf);
        v3 dw = mat3(sc.env_to_world, dl);
        if (isZero(value) || pdf == 0 || !bsphereIntersect(sc, ref, dw, nearT, farT) || nearT >= 0 || farT <= 0) { dr.pdf = 0.0f; return V(0, 0, 0); }
        dr.pdf = pdf; dr.p = ref + dw * farT; dr.n = normalize(ld3(sc.env_bs_center) - dr.p); dr.dist = farT; dr.d = dw;
        { float r = 1.0f / pdf; value = value * r; }
        dr.emitter = (int) ei; dr.pdf *= emPdf;
        if (RAW) dr.em_pdf = emPdf; else { float r = 1.0f / emPdf; value = value * r; }
        return value;
    }
    if (AN && em.analytic >= 0) {                                // area light on an analytic shape: no sample reuse
        analyticSampleDirect(sc.analytic[em.analytic], ref, sx, sy, dr.p, dr.n, dr.d, dr.dist, dr.pdf);
    } else {
    typename AS<L>::pf acdf = tb.area_cdf + em.cdf_offset;
    float a0, a1; const uint32_t ti = cdfSampleReuse(acdf, em.tri_count, sy, a0, a1, (sc.search_flags & 2u) != 0);
    sy = (sy - a0) / (a1 - a0);
    uint32_t prim = em.first_tri + ti;
    typename AS<L>::p4 rec = tb.shade4 + prim * (uint32_t) MI_SHADE_WORDS;
    f4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
    v3 p0 = V(r0.x, r0.y, r0.z), p1 = V(r1.x, r1.y, r1.z), p2 = V(r2.x, r2.y, r2.z);
    float bx, by; uniformTriangle(sx, sy, bx, by);
    v3 sideA = p1 - p0, sideB = p2 - p0;
    dr.p = (p0 + sideA * bx) + sideB * by;
    if (__float_as_uint(r2.w) & 1u) dr.n = normalize(cross(sideA, sideB));
    else {
        f4 r4 = rec[4], r5 = rec[5];
        const f4 r6 = rec[6];
        dr.n = normalize((V(r4.x, r4.y, r4.z) * (1.0f - bx - by) + V(r5.x, r5.y, r5.z) * bx) + V(r6.x, r6.y, r6.z) * by);
    }
    dr.pdf = em.inv_area;
    dr.d = dr.p - ref;
    float distSquared = dot(dr.d, dr.d);
    dr.dist = sqrtf(distSquared);
    { float r = 1.0f / dr.dist; dr.d = dr.d * r; }
    float dp = fabsf(dot(dr.d, dr.n));
    dr.pdf *= dp != 0 ? (distSquared / dp) : 0.0f;
    }
    v3 value;
    if (dot(dr.d, refN) >= 0 && dot(dr.d, dr.n) < 0 && dr.pdf != 0) {
        float r = 1.0f / dr.pdf; value = V(em.radiance[0] * r, em.radiance[1] * r, em.radiance[2] * r);
    } else { dr.pdf = 0.0f; value = V(0, 0, 0); }
    if (dr.pdf != 0) {
        dr.emitter = (int) ei;
        dr.pdf *= emPdf;
        if (RAW) dr.em_pdf = emPdf; else { float r = 1.0f / emPdf; value = value * r; }
        return value;
    }
    return V(0, 0, 0);
}
// Scene::pdfEmitterDirect (scene.cpp:981-984) -> AreaLight::pdfDirect (area.cpp:178-184) -> Shape::pdfDirect (shape.cpp:117-126);
// `facingRef` = (dot(d, refN) >= 0) evaluated where refN was still known (the previous vertex)
template <bool AN, bool L>
DEV float pdfEmitterDirect(const DScene &sc, const Tabs<L> &tb, int e, v3 ref, v3 d, v3 n, float dist, bool facingRef) {
    f4 a = tb.emitters4[e * 3], b = tb.emitters4[e * 3 + 1];     // a.w = weight, b.w = inv_area
    float pdf;
    if (facingRef && dot(d, n) < 0) {
        int an = AN ? __float_as_int(tb.emitters4[e * 3 + 2].z) : -1;
        if (AN && an >= 0) pdf = analyticPdfDirect(sc.analytic[an], ref, d, n, dist);
        else pdf = b.w * (dist * dist) / fabsf(dot(d, n));
    } else pdf = 0.0f;
    return pdf * (a.w * sc.emitter_norm);
}
DEV float miWeight(float a, float b) { a *= a; b *= b; return a / (a + b); }   // src/integrators/path/path.cpp:296-300

// ---------------------------------------------------------------------------------------------- participating media (volumetric integrators)
// include/mitsuba/core/math.h:185-195 (Linux x86_64): fastexp / fastlog go through the DOUBLE-precision routines, rounded to float
__device__ __noinline__ static float miFastExp(float v) { return (float) exp((double) v); }
__device__ __noinline__ static float miFastLog(float v) { return (float) log((double) v); }
struct MediumRec { float t; v3 p; v3 transmittance; float pdfSuccess, pdfFailure; };
// HomogeneousMedium::evalTransmittance (src/medium/homogeneous.cpp:266-273) over [mint, maxt] of a ray
DEV v3 mediumTransmittance(const MediumD &m, float mint, float maxt) {
    const float negLength = mint - maxt;
    return V(m.sigma_t[0] != 0 ? miFastExp(m.sigma_t[0] * negLength) : 1.0f, m.sigma_t[1] != 0 ? miFastExp(m.sigma_t[1] * negLength) : 1.0f, m.sigma_t[2] != 0 ? miFastExp(m.sigma_t[2] * negLength) : 1.0f);
}
// HomogeneousMedium::sampleDistance (homogeneous.cpp:275-349), strategies balance / single / manual; draws one or two 1-D samples
template <typename P>
DEV bool mediumSampleDistance(const MediumD &m, v3 o, v3 d, float mint, float maxt, SamplerState &ss, uint32_t kind, SobolTabT<P> st, MediumRec &r) {
    float rnd = next1D(ss, kind, st), sampled, density = m.sampling_density;
    if (rnd < m.medium_sampling_weight) {
        rnd /= m.medium_sampling_weight;
        if (m.strategy == 0u) { int ch = (int) (next1D(ss, kind, st) * 3); if (ch > 2) ch = 2; density = ch == 0 ? m.sigma_t[0] : (ch == 1 ? m.sigma_t[1] : m.sigma_t[2]); }
        sampled = -miFastLog(1 - rnd) / density;
    } else sampled = INFINITY;
    const float distSurf = maxt - mint; bool success = true;
    if (sampled < distSurf) {
        r.t = sampled + mint; r.p = o + d * r.t;
        if (r.p.x == o.x && r.p.y == o.y && r.p.z == o.z) success = false;
    } else { sampled = distSurf; success = false; }
    if (m.strategy == 0u) {
        r.pdfFailure = 0; r.pdfSuccess = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) { const float tmp = miFastExp(-m.sigma_t[i] * sampled); r.pdfFailure += tmp; r.pdfSuccess += m.sigma_t[i] * tmp; }
        r.pdfFailure /= 3; r.pdfSuccess /= 3;
    } else { r.pdfFailure = miFastExp(-density * sampled); r.pdfSuccess = density * r.pdfFailure; }
    r.transmittance = V(miFastExp(m.sigma_t[0] * (-sampled)), miFastExp(m.sigma_t[1] * (-sampled)), miFastExp(m.sigma_t[2] * (-sampled)));
    r.pdfSuccess = r.pdfSuccess * m.medium_sampling_weight;
    r.pdfFailure = m.medium_sampling_weight * r.pdfFailure + (1 - m.medium_sampling_weight);
    if (maxf(maxf(r.transmittance.x, r.transmittance.y), r.transmittance.z) < 1e-20f) r.transmittance = V(0, 0, 0);
    return success;
}
// IsotropicPhaseFunction::eval (src/phase/isotropic.cpp:74-76), HGPhaseFunction::eval (src/phase/hg.cpp:108-111); wi = -ray.d
DEV float phaseEval(const MediumD &m, v3 wi, v3 wo) {
    if (m.phase == 0u) return MI_INV_FOURPI;
    const float g = m.g, temp = 1.0f + g * g + 2.0f * g * dot(wi, wo);
    return MI_INV_FOURPI * (1 - g * g) / (temp * sqrtf(temp));
}
// IsotropicPhaseFunction::sample (isotropic.cpp:61-66), HGPhaseFunction::sample (hg.cpp:74-99); the weight is 1 for both
DEV v3 phaseSample(const MediumD &m, v3 wi, float sx, float sy) {
    if (m.phase == 0u) return uniformSphere(sx, sy);
    const float g = m.g; float cosTheta;
    if (fabsf(g) < MI_EPSILON) cosTheta = 1 - 2 * sx;
    else { const float sqrTerm = (1 - g * g) / (1 - g + 2 * g * sx); cosTheta = (1 + g * g - sqrTerm * sqrTerm) / (2 * g); }
    float sinTheta = sqrtf(maxf(1.0f - cosTheta * cosTheta, 0.0f)), sinPhi, cosPhi;
    sincos2pi(sy, sinPhi, cosPhi);
    const v3 n = V(-wi.x, -wi.y, -wi.z); v3 fs, ft; coordinateSystem(n, fs, ft);          // Frame(-pRec.wi).toWorld
    return (fs * (sinTheta * cosPhi) + ft * (sinTheta * sinPhi)) + n * cosTheta;
}
// bsdf->eval(bRec, EDiscrete) with typeMask = ENull for a straight pass-through (scene.cpp:679-685, volpath.cpp:399-402): `null` -> 1, `thindielectric` -> its transmittance
// with the internal bounces summed (thindielectric.cpp:155-178); cosWi = Frame::cosTheta(bRec.wi)
DEV bool materialHasNull(uint32_t type) { return type == MI_BSDF_T_NULL || type == MI_BSDF_T_THINDIELECTRIC || type == MI_BSDF_T_MASK; }      // mask.cpp:107-108: a mask always has an ENull component
DEV v3 materialNullEval(const MaterialD &m, float cosWi) {
    if (m.type == MI_BSDF_T_NULL) return V(1, 1, 1);
    float ct, R = fresnelDielectricExt(fabsf(cosWi), ct, m.eta[0]), T = 1 - R;
    if (R < 1) R += T * T * R / (1 - R * R);
    return ld3(m.reflectance) * (1 - R);
}
// the same at a hit: a `mask` answers 1 - opacity (mask.cpp:120-121), its opacity texture looked up at the hit's uv without differentials (level 0 of a bitmap).  walk:
// the hit comes from ShapeKDTree::rayIntersect(ray, t, shape, n, uv), which gives a scene-level triangle mesh WITHOUT texture coordinates uv = (0, 0)
// (skdtree.cpp:182-184), not the barycentrics
// ... and of a record that may be a mixturebsdf: its children's components (mixturebsdf.cpp:150-166)
template <bool L> DEV bool surfaceHasNull(const Tabs<L> &tb, const MaterialD &m) {
    if (m.type != MI_BSDF_T_MIXTURE) return materialHasNull(m.type);
    const MixD x = mixOf(m); bool r = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i < x.n) { const uint32_t ct = loadMaterial(tb, (int) mixChild(x, i)).type; r |= ct == MI_BSDF_T_NULL || ct == MI_BSDF_T_THINDIELECTRIC; }
    return r;
}
template <bool L> DEV v3 surfaceNullEval(const DScene &sc, const Tabs<L> &tb, const MaterialD &m, v3 o, v3 d, float t, uint32_t prim, float u, float v, int inst, float cosWi, bool walk) {
    if (m.type == MI_BSDF_T_MIXTURE) {           // MixtureBSDF::eval (mixturebsdf.cpp:176-183) under EDiscrete / typeMask = ENull: weight x the pass-through value of every child that has an ENull lobe
        const MixD x = mixOf(m); v3 r = V(0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < x.n) { const MaterialD c = loadMaterial(tb, (int) mixChild(x, i)); if (c.type == MI_BSDF_T_NULL || c.type == MI_BSDF_T_THINDIELECTRIC) r = r + materialNullEval(c, cosWi) * mixWeight(x, i); }
        return r;
    }
    if (m.type != MI_BSDF_T_MASK) return materialNullEval(m, cosWi);
    float uvx = 0, uvy = 0;
    if (inst >= 0) { Hit hh; fillHitInstanced(sc, tb, sc.instances[inst], o, d, t, prim, u, v, hh); uvx = hh.uvx; uvy = hh.uvy; }
    else if (prim >= sc.n_tris) { v3 du, dv; analyticUV(sc.analytic[prim - sc.n_tris], u, v, o + d * t, uvx, uvy, du, dv); }
    else { Hit hh; fillHit<L, true>(sc, tb, d, t, prim, u, v, hh); if (!walk || (hh.flags & 16u)) { uvx = hh.uvx; uvy = hh.uvy; } }
    v3 op = ld3(m.reflectance); const uint32_t tex = (m.flags >> 8) & 0xFFFFu;
    if (tex) {
        const TextureD &tx = sc.textures[tex - 1];
        if (tx.type == 2u) { const float a = uvx * tx.uscale + tx.uoffset, b = uvy * tx.vscale + tx.voffset; op = tx.filter != 0u ? mipBilinear(sc, tx, 0, a, b) : mipBox(sc, tx, 0, a, b); }
        else op = textureEval(tx, uvx, uvy);
    }
    return V(1.0f - op.x, 1.0f - op.y, 1.0f - op.z);
}
// Shape::isMediumTransition / Intersection::getTargetMedium (include/mitsuba/render/records.inl:77-86): index of the medium on the side `d` points to, -1 = none
DEV int targetMedium(uint32_t pm, v3 n, v3 d) { return (int) (dot(d, n) > 0 ? (pm >> 16) : (pm & 0xFFFFu)) - 1; }

// include/mitsuba/core/rfilter.h:76-77
DEV float filterEvalDiscretized(const DScene &sc, float x) {
    int i = (int) fabsf(x * sc.filter_scale); if (i > MI_FILTER_RES) i = MI_FILTER_RES; return sc.filter_values[i];
}
