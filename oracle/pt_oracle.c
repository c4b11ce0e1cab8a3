/* oracle/pt_oracle.c -- TEST INFRASTRUCTURE (see pt_oracle.h).  Plain-C CPU restatement of the reference's
 * unidirectional path tracer: MIPathTracer::Li over Scene::rayIntersect + BSDF / emitter / sampler plugins.
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 *
 * Arithmetic contract (shared with the HIP product only as a written specification, DESIGN.md §"arithmetic"):
 *   - IEEE binary32, round-to-nearest, no contraction (compiled with -ffp-contract=off, no fast-math);
 *     expressions are evaluated left to right exactly as written; x/len of vectors is "recip then multiply"
 *     like the reference's TVector3::operator/= (include/mitsuba/core/vector.h:546-553).
 *   - sin/cos of the warps (concentric disk, uniform sphere / cone, cylinder) are glibc's sincosf, as in the reference
 *     (math::sincos, include/mitsuba/core/math.h:219-221); the HIP product restates glibc's algorithm in fp64 (pt_device.h
 *     glibcSincosf, pinned against glibc over every float in [-8, 8] by tests/test_host_logic.py), so CPU and GPU still agree bit for bit.
 *   - closest hit = minimum t over all triangles passing the TriAccel test in [mint, maxt]; ties are broken
 *     towards the lower global triangle index, which makes the result independent of traversal order
 *     (the kd-tree of the reference keeps the last-tested of equal-t hits: SURVEY.md §7; scenes avoid coincident geometry).
 */
#define _GNU_SOURCE
#include "pt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <pthread.h>

#define EPSILON 1e-4f            /* include/mitsuba/core/constants.h:28 */
#define SHADOW_EPSILON 1e-3f     /* constants.h:29 */
#define INV_PI 0.31830988618379067154f  /* constants.h:64 */
#define INV_FOURPI 0.07957747154594766788f /* constants.h:66 */
#define ONE_MINUS_EPS 0.999999940395355225f /* constants.h:51 */
#define M_PI_F 3.14159265358979323846f
#define KD_AABB_EPSILON 1e-3f    /* include/mitsuba/render/gkdtree.h:50 */
#define FILTER_RES 31            /* include/mitsuba/core/rfilter.h:28 */

typedef struct { float x, y, z; } v3;
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline v3 normalize(v3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return scale(a, inv); }
static inline int is_zero(v3 a) { return a.x == 0 && a.y == 0 && a.z == 0; }
static inline float maxf(float a, float b) { return a > b ? a : b; }
static inline float minf(float a, float b) { return a < b ? a : b; }
static inline float comp(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

typedef struct { uint32_t k; float n_u, n_v, n_d, a_u, a_v, b_nu, b_nv, c_nu, c_nv; } triaccel;
typedef struct { v3 lo, hi; int32_t left, right, first, count; } bvh_node;   /* leaf: count > 0 */

typedef struct {
    v3 dpdu, dpdv;    /* its.dpdu / its.dpdv (world space): feed Intersection::computePartials */
    float uvx, uvy;   /* its.uv: interpolated texture coordinates (meshes with texcoords), else the barycentrics */
    int valid; float t; v3 p, ng, ns, s, tt; float u, v; v3 wi; uint32_t prim, shape; int32_t material, emitter; int32_t instance;
    v3 ng_raw;        /* the `n` of ShapeKDTree::rayIntersect(ray, t, shape, n, uv) (skdtree.cpp:144-205): the triangle's face normal BEFORE it is flipped to the shading side; analytic shapes / instances: ng */
} hit_t;

/* a material + the table it refers to (roughplastic); bsdf_* functions receive &mat->m and may cast back */
typedef struct mat_s { orc_material m; const float *table; } mat_t;
struct orc_scene {
    orc_scene_desc d;
    uint32_t *tex_levels; float *tex_texels; float mip_lut[64]; v3 cam_dx, cam_dy;   /* MIP pyramids (input data); EWA weight table; perspective.cpp:159-163 */
    float *uv; float *tangents; /* per triangle: dpdu xyz, dpdv xyz (TriMesh::computeUVTangents) */ orc_texture *textures;
    float *pos, *nrm; uint32_t *idx; orc_shape *shapes; struct mat_s *materials; orc_emitter *emitters; float *material_tables;
    uint32_t *tri_shape; orc_medium *media;
    struct analytic_s *analytic; uint32_t n_analytic, n_prims;
    orc_instance *instances; uint32_t n_instances, n_groups; int *group_root; v3 *group_lo, *group_hi;   /* per shape group: BVH root, kd-tree box (enlarged) */
    uint32_t *group_first, *group_count;     /* per group: its range in group_prims (for the brute-force cross-check) */ uint32_t *group_prims;   /* primitive index space: [0, n_tris) triangles, then the analytic shapes */
    triaccel *accel;
    v3 aabb_lo, aabb_hi;            /* kd-tree root box incl. the reference's enlargement */
    bvh_node *nodes; uint32_t *bvh_tris; int n_nodes;
    /* emitters */
    float *emitter_cdf; float emitter_norm;
    float **area_cdf; float *inv_area;   /* per emitter */
    /* film */
    float filter_values[FILTER_RES + 1]; float filter_radius, filter_scale; int border;
    uint32_t log_res; float resolution;
    float inv_res_x, inv_res_y;
    /* environment emitter (src/emitters/envmap.cpp); env_index = its position in the emitter list or -1 */
    int env_index; int env_w, env_h; float *env_rgb; float *env_cdf_cols, *env_cdf_rows, *env_row_weights;
    float env_normalization, env_scale, env_to_world[9], env_to_local[9], env_pixel_w, env_pixel_h;
    v3 env_bs_center; float env_bs_radius;
    int env_constant;                     /* the environment emitter is `constant` (src/emitters/constant.cpp), radiance in emitters[env_index] */
    /* delta emitters: spot constants (spot.cpp:91-96), directional bounding sphere (directional.cpp:87-93) */
    float *spot_cos_beam, *spot_cos_cutoff, *spot_inv_transition, *spot_cutoff, *spot_to_local;   /* per emitter; to_local = 9 floats each */
    v3 dir_bs_center; float dir_bs_radius;
};

/* ------------------------------------------------------------------------------------------------ samplers */
/* include/mitsuba/core/qmc.h:146-156 sampleTEA */
uint64_t orc_tea(uint32_t v0, uint32_t v1, int rounds) {
    uint32_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xA341316Cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xC8013EA4u);
        v1 += ((v0 << 4) + 0xAD90777Du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7E95761Eu);
    }
    return ((uint64_t) v1 << 32) + v0;
}
/* src/libcore/random.cpp:626-634 nextFloat bit trick */
static inline float bits_to_float(uint32_t b) { union { uint32_t u; float f; } x; x.u = (b >> 9) | 0x3f800000u; return x.f - 1.0f; }

/* SobolSampler's scramble value (src/samplers/sobol.cpp:92-102): 0 stays 0, a frame number goes through sampleTEA (4 rounds) */
static inline uint64_t sobol_scramble(const orc_scene *s) {
    const uint64_t v = s->d.seed;
    return (s->d.sampler == 1 && v) ? orc_tea((uint32_t) v, (uint32_t) (v >> 32), 4) : 0;
}
/* src/samplers/sobolseq.h:43-58 sampleSingle: the XOR sum starts from the (low 32 bits of the) scramble value */
float orc_sobol_sample(const orc_scene *s, uint64_t index, uint32_t dim) {
    uint32_t result = (uint32_t) sobol_scramble(s);
    for (uint32_t i = dim * 52; index; index >>= 1, ++i)
        if (index & 1) result ^= s->d.sobol_matrices32[i];
    return minf((float) result * (1.0f / 4294967296.0f), ONE_MINUS_EPS);
}
/* src/samplers/sobolseq.h:99-131 look_up; the pixel coordinates are flipped by the scramble value's top m bits (:117-124) */
uint64_t orc_sobol_look_up(const orc_scene *s, uint32_t m, uint32_t frame, uint32_t px, uint32_t py) {
    const uint32_t m2 = m << 1;
    uint64_t index = (uint64_t) frame << m2;
    uint64_t delta = 0;
    for (uint32_t c = 0; frame; frame >>= 1, ++c)
        if (frame & 1) delta ^= s->d.sobol_vdc[(m - 1) * 52 + c];
    const uint64_t scr = (sobol_scramble(s) & 0xFFFFFFFFull) >> (32 - m);
    uint64_t b = (((uint64_t) (px ^ scr) << m) | (py ^ scr)) ^ delta;
    for (uint32_t c = 0; b; b >>= 1, ++c)
        if (b & 1) index ^= s->d.sobol_vdc_inv[(m - 1) * 52 + c];
    return index;
}

/* Sampler state: src/samplers/sobol.cpp:171-251 (generate/setSampleIndex/next1D/next2D, m_arrayStartDim = m_arrayEndDim = 5)
 * and the build-defined independent stream (mirrored in oracle/ref_build/harness.cpp SeededIndependent). */
typedef struct {
    const orc_scene *sc; int kind;
    uint32_t px, py; uint64_t sample_index, sobol_index; uint32_t dim;   /* sobol */
    uint32_t v0, call;                                                  /* independent */
    float *log; int nlog;
} sampler_t;

static void sampler_begin(sampler_t *sp, const orc_scene *sc, uint32_t px, uint32_t py, uint64_t sample_index, float *log) {
    sp->sc = sc; sp->kind = (int) sc->d.sampler; sp->px = px; sp->py = py; sp->sample_index = sample_index;
    sp->dim = 0; sp->log = log; sp->nlog = 0;
    if (sp->kind == 1) {
        /* sobol.cpp:204-216 setSampleIndex */
        if (sc->log_res > 1) sp->sobol_index = orc_sobol_look_up(sc, sc->log_res, (uint32_t) sample_index, px, py);
        else sp->sobol_index = sample_index;
    } else {
        sp->v0 = (py * sc->d.width + px) ^ ((uint32_t) sc->d.seed * 0x9E3779B9u);
        sp->call = 0;
    }
}
static inline void slog(sampler_t *sp, float v) { if (sp->log && sp->nlog < 64) sp->log[sp->nlog] = v; sp->nlog++; }
static float next1D(sampler_t *sp) {
    float v;
    if (sp->kind == 1) {
        if (sp->dim >= 5 && sp->dim < 5) sp->dim = 5;          /* sobol.cpp:220-221 with start == end == 5: never taken */
        v = orc_sobol_sample(sp->sc, sp->sobol_index, sp->dim++);
    } else {
        uint32_t v1 = ((uint32_t) sp->sample_index << 8) | (sp->call++ & 0xFFu);
        v = bits_to_float((uint32_t) orc_tea(sp->v0, v1, 4));
    }
    slog(sp, v); return v;
}
static void next2D(sampler_t *sp, float *x, float *y) {
    if (sp->kind == 1) {
        if (sp->dim + 1 >= 5 && sp->dim < 5) sp->dim = 5;      /* sobol.cpp:233-235: a 2D request at dim 4 skips to 5 */
        if (sp->dim == 0 && sp->sobol_index != sp->sample_index) {  /* sobol.cpp:241-243 */
            *x = orc_sobol_sample(sp->sc, sp->sobol_index, sp->dim++) * sp->sc->resolution - (float) (int32_t) sp->px;
            *y = orc_sobol_sample(sp->sc, sp->sobol_index, sp->dim++) * sp->sc->resolution - (float) (int32_t) sp->py;
        } else {
            *x = orc_sobol_sample(sp->sc, sp->sobol_index, sp->dim++);
            *y = orc_sobol_sample(sp->sc, sp->sobol_index, sp->dim++);
        }
    } else {
        uint32_t v1 = ((uint32_t) sp->sample_index << 8) | (sp->call++ & 0xFFu);
        uint64_t r = orc_tea(sp->v0, v1, 4);
        *x = bits_to_float((uint32_t) r); *y = bits_to_float((uint32_t) (r >> 32));
    }
    slog(sp, *x); slog(sp, *y);
}

/* ------------------------------------------------------------------------------------------------ SFMT19937 (KAT only) */
/* src/libcore/random.cpp:72-96 parameters, :397-406 init_gen_rand (64-bit seed), :322-346 period_certification,
 * :353-390 gen_rand_all (SFMT recursion of Saito & Matsumoto), :288-296 gen_rand64, :626-634 nextFloat */
#define SFMT_N 156
#define SFMT_N32 624
#define SFMT_N64 312
typedef struct { uint32_t s[SFMT_N32]; int idx; } sfmt_t;
static void sfmt_init(sfmt_t *st, uint64_t seed) {
    uint64_t p[SFMT_N64]; p[0] = seed;
    for (int i = 1; i < SFMT_N64; ++i) p[i] = 6364136223846793005ULL * (p[i - 1] ^ (p[i - 1] >> 62)) + (uint64_t) i;
    memcpy(st->s, p, sizeof(p));  /* little-endian: psfmt64 aliases psfmt32 */
    st->idx = SFMT_N32;
    static const uint32_t parity[4] = {0x00000001u, 0, 0, 0x13c9e684u};
    uint32_t inner = 0;
    for (int i = 0; i < 4; ++i) inner ^= st->s[i] & parity[i];
    for (int i = 16; i > 0; i >>= 1) inner ^= inner >> i;
    if ((inner & 1) == 1) return;
    for (int i = 0; i < 4; ++i) { uint32_t work = 1; for (int j = 0; j < 32; ++j) { if (work & parity[i]) { st->s[i] ^= work; return; } work <<= 1; } }
}
static void sfmt_recursion(uint32_t *r, const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d) {
    /* 128-bit shifts by SL2 = SR2 = 1 byte; 32-bit shifts SL1 = 18, SR1 = 11; masks MSK1..4 */
    static const uint32_t msk[4] = {0xdfffffefu, 0xddfecb7fu, 0xbffaffffu, 0xbffffff6u};
    uint64_t al = ((uint64_t) a[1] << 32) | a[0], ah = ((uint64_t) a[3] << 32) | a[2];
    uint64_t cl = ((uint64_t) c[1] << 32) | c[0], ch = ((uint64_t) c[3] << 32) | c[2];
    uint64_t xh = (ah << 8) | (al >> 56), xl = al << 8;         /* lshift128(a, 1 byte) */
    uint64_t yl = (cl >> 8) | (ch << 56), yh = ch >> 8;         /* rshift128(c, 1 byte) */
    uint32_t x[4] = {(uint32_t) xl, (uint32_t) (xl >> 32), (uint32_t) xh, (uint32_t) (xh >> 32)};
    uint32_t y[4] = {(uint32_t) yl, (uint32_t) (yl >> 32), (uint32_t) yh, (uint32_t) (yh >> 32)};
    for (int i = 0; i < 4; ++i) r[i] = a[i] ^ x[i] ^ ((b[i] >> 11) & msk[i]) ^ y[i] ^ (d[i] << 18);
}
static void sfmt_gen_all(sfmt_t *st) {
    uint32_t *s = st->s; const int POS1 = 122;
    uint32_t *r1 = &s[(SFMT_N - 2) * 4], *r2 = &s[(SFMT_N - 1) * 4];
    int i = 0;
    for (; i < SFMT_N - POS1; ++i) { sfmt_recursion(&s[i * 4], &s[i * 4], &s[(i + POS1) * 4], r1, r2); r1 = r2; r2 = &s[i * 4]; }
    for (; i < SFMT_N; ++i) { sfmt_recursion(&s[i * 4], &s[i * 4], &s[(i + POS1 - SFMT_N) * 4], r1, r2); r1 = r2; r2 = &s[i * 4]; }
}
static uint64_t sfmt_next64(sfmt_t *st) {
    if (st->idx >= SFMT_N32) { sfmt_gen_all(st); st->idx = 0; }
    uint64_t r = ((uint64_t) st->s[st->idx + 1] << 32) | st->s[st->idx]; st->idx += 2; return r;
}
void orc_sfmt_sequence(uint64_t seed, uint64_t n, uint64_t *out) { sfmt_t st; sfmt_init(&st, seed); for (uint64_t i = 0; i < n; ++i) out[i] = sfmt_next64(&st); }
void orc_sfmt_floats(uint64_t seed, uint64_t n, float *out) { sfmt_t st; sfmt_init(&st, seed); for (uint64_t i = 0; i < n; ++i) out[i] = bits_to_float((uint32_t) (sfmt_next64(&st) & 0xFFFFFFFFu)); }

/* ------------------------------------------------------------------------------------------------ warps */
/* src/libcore/warp.cpp:81-101 squareToUniformDiskConcentric; math::sincos = ::sincosf (include/mitsuba/core/math.h:219-221) */
static void disk_concentric(float sx, float sy, float *ox, float *oy) {
    float r1 = 2.0f * sx - 1.0f, r2 = 2.0f * sy - 1.0f, r, phi, sn, cs;
    if (r1 == 0 && r2 == 0) r = phi = 0;
    else if (r1 * r1 > r2 * r2) { r = r1; phi = (M_PI_F / 4.0f) * (r2 / r1); }
    else { r = r2; phi = (M_PI_F / 2.0f) - (r1 / r2) * (M_PI_F / 4.0f); }
    sincosf(phi, &sn, &cs);
    *ox = r * cs; *oy = r * sn;
}
/* warp.cpp:43-52 squareToCosineHemisphere; math::safe_sqrt = sqrt(max(x,0)) (include/mitsuba/core/math.h:260-267) */
static v3 cos_hemisphere(float sx, float sy) {
    float px, py; disk_concentric(sx, sy, &px, &py);
    float z = sqrtf(maxf(1.0f - px * px - py * py, 0.0f));
    if (z == 0) z = 1e-10f;
    return V(px, py, z);
}
/* warp.cpp:76-79 squareToUniformTriangle */
static void uniform_triangle(float sx, float sy, float *bx, float *by) { float a = sqrtf(maxf(1.0f - sx, 0.0f)); *bx = 1 - a; *by = a * sy; }
void orc_warp(float u, float v, float *o) { v3 h = cos_hemisphere(u, v); o[0] = h.x; o[1] = h.y; o[2] = h.z; uniform_triangle(u, v, &o[3], &o[4]); disk_concentric(u, v, &o[5], &o[6]); }

/* ------------------------------------------------------------------------------------------------ geometry */
static inline v3 vert(const orc_scene *s, uint32_t i) { return V(s->pos[i * 3], s->pos[i * 3 + 1], s->pos[i * 3 + 2]); }
static inline v3 vnrm(const orc_scene *s, uint32_t i) { return V(s->nrm[i * 3], s->nrm[i * 3 + 1], s->nrm[i * 3 + 2]); }

/* include/mitsuba/render/triaccel.h:61-94 TriAccel::load */
static void triaccel_load(triaccel *ta, v3 A, v3 B, v3 C) {
    static const int wald[4] = {1, 2, 0, 1};
    v3 b = sub(C, A), c = sub(B, A), N = cross(c, b);
    int k = 0;
    for (int j = 0; j < 3; ++j) if (fabsf(comp(N, j)) > fabsf(comp(N, k))) k = j;
    int u = wald[k], v = wald[k + 1];
    float n_k = comp(N, k), denom = comp(b, u) * comp(c, v) - comp(b, v) * comp(c, u);
    memset(ta, 0, sizeof(*ta));
    if (denom == 0) { ta->k = 3; return; }
    ta->k = (uint32_t) k;
    ta->n_u = comp(N, u) / n_k; ta->n_v = comp(N, v) / n_k; ta->n_d = dot(A, N) / n_k;
    ta->b_nu = comp(b, u) / denom; ta->b_nv = -comp(b, v) / denom;
    ta->a_u = comp(A, u); ta->a_v = comp(A, v);
    ta->c_nu = comp(c, v) / denom; ta->c_nv = -comp(c, u) / denom;
}
/* triaccel.h:96-158 TriAccel::rayIntersect */
static inline int triaccel_intersect(const triaccel *ta, v3 o, v3 d, float mint, float maxt, float *u, float *v, float *t) {
    float o_u, o_v, o_k, d_u, d_v, d_k;
    switch (ta->k) {
        case 0: o_u = o.y; o_v = o.z; o_k = o.x; d_u = d.y; d_v = d.z; d_k = d.x; break;
        case 1: o_u = o.z; o_v = o.x; o_k = o.y; d_u = d.z; d_v = d.x; d_k = d.y; break;
        case 2: o_u = o.x; o_v = o.y; o_k = o.z; d_u = d.x; d_v = d.y; d_k = d.z; break;
        default: return 0;
    }
    float tt = (ta->n_d - o_u * ta->n_u - o_v * ta->n_v - o_k) / (d_u * ta->n_u + d_v * ta->n_v + d_k);
    if (tt < mint || tt > maxt) return 0;   /* NaN falls through both comparisons, exactly like the reference */
    float hu = o_u + tt * d_u - ta->a_u, hv = o_v + tt * d_v - ta->a_v;
    float uu = hv * ta->b_nu + hu * ta->b_nv, vv = hu * ta->c_nu + hv * ta->c_nv;
    *u = uu; *v = vv; *t = tt;
    return uu >= 0 && vv >= 0 && uu + vv <= 1.0f;
}
void orc_triaccel(const orc_scene *s, uint32_t tri, float *o) {
    const triaccel *a = &s->accel[tri];
    o[0] = (float) a->k; o[1] = a->n_u; o[2] = a->n_v; o[3] = a->n_d; o[4] = a->a_u; o[5] = a->a_v; o[6] = a->b_nu; o[7] = a->b_nv; o[8] = a->c_nu; o[9] = a->c_nv;
}

/* ------------------------------------------------------------------------------------------------ analytic shapes */
/* Non-triangle primitives behind Scene::rayIntersect: the kd-tree leaf redirects to Shape::rayIntersect for them
 * (include/mitsuba/render/skdtree.h:292-301, :330-333) and to Shape::fillIntersectionRecord afterwards (:421-427).
 * Transforms come with the descriptor (to_world = the shape's m_objectToWorld after its constructor, to_object = its inverse). */
enum { SH_RECTANGLE = 0, SH_DISK = 1, SH_SPHERE = 2, SH_CYLINDER = 3 };
typedef struct analytic_s {
    orc_analytic a;
    v3 n;            /* rectangle / disk: normalize(objectToWorld(Normal(0,0,1))) (rectangle.cpp:106, disk.cpp:185) */
    v3 dpdu;         /* rectangle: objectToWorld(Vector(2,0,0)) (rectangle.cpp:104) */
    v3 center;       /* sphere: objectToWorld(Point(0,0,0)) (sphere.cpp:127) */
    float inv_area;
    v3 lo, hi;       /* Shape::getAABB */
} analytic_t;
/* include/mitsuba/core/transform.h:126-135 transformAffine(Point), :172-181 operator()(Vector), :199-207 operator()(Normal) */
static inline v3 xf_point(const float *m, v3 p) { return V(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7], m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]); }
static inline v3 xf_vector(const float *m, v3 v) { return V(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z); }
static inline v3 xf_normal(const float *inv, v3 n) { return V(inv[0] * n.x + inv[4] * n.y + inv[8] * n.z, inv[1] * n.x + inv[5] * n.y + inv[9] * n.z, inv[2] * n.x + inv[6] * n.y + inv[10] * n.z); }
static inline float length3(v3 a) { return sqrtf(dot(a, a)); }

/* math::sincos(2.0f * M_PI * u) of squareToUniformSphere / squareToUniformCone (warp.cpp:29, :59) and Cylinder::samplePosition (cylinder.cpp:234) */
static void sincos_2pi(float u, float *sn, float *cs) { sincosf((2.0f * M_PI_F) * u, sn, cs); }
/* src/libcore/util.cpp:489-527 solveQuadraticDouble */
static int solve_quadratic_double(double a, double b, double c, double *x0, double *x1) {
    if (a == 0) { if (b != 0) { *x0 = *x1 = -c / b; return 1; } return 0; }
    double discrim = b * b - 4.0 * a * c;
    if (discrim < 0) return 0;
    double temp, sqrtDiscrim = sqrt(discrim);
    if (b < 0) temp = -0.5 * (b - sqrtDiscrim); else temp = -0.5 * (b + sqrtDiscrim);
    *x0 = temp / a; *x1 = c / temp;
    if (*x0 > *x1) { double t = *x0; *x0 = *x1; *x1 = t; }
    return 1;
}
/* util.cpp:449-487 solveQuadratic */
static int solve_quadratic(float a, float b, float c, float *x0, float *x1) {
    if (a == 0) { if (b != 0) { *x0 = *x1 = -c / b; return 1; } return 0; }
    float discrim = b * b - 4.0f * a * c;
    if (discrim < 0) return 0;
    float temp, sqrtDiscrim = sqrtf(discrim);
    if (b < 0) temp = -0.5f * (b - sqrtDiscrim); else temp = -0.5f * (b + sqrtDiscrim);
    *x0 = temp / a; *x1 = c / temp;
    if (*x0 > *x1) { float t = *x0; *x0 = *x1; *x1 = t; }
    return 1;
}
/* util.cpp:594-603 coordinateSystem */
static void coordinate_system(v3 a, v3 *b, v3 *c) {
    if (fabsf(a.x) > fabsf(a.y)) { float invLen = 1.0f / sqrtf(a.x * a.x + a.z * a.z); *c = V(a.z * invLen, 0.0f, -a.x * invLen); }
    else { float invLen = 1.0f / sqrtf(a.y * a.y + a.z * a.z); *c = V(0.0f, a.z * invLen, -a.y * invLen); }
    *b = cross(*c, a);
}
/* Shape::rayIntersect(ray, mint, maxt, t, temp): rectangle.cpp:125-148, disk.cpp:141-165, sphere.cpp:148-174, cylinder.cpp:128-166.
 * any != 0 selects the shadow-ray overloads (rectangle.cpp:150-153, disk.cpp:167-170, sphere.cpp:176-194, cylinder.cpp:168-201),
 * which differ from the closest-hit ones for the quadrics.  *u,*v = the temp data (local x, y) of rectangle / disk. */
static int analytic_intersect(const analytic_t *sh, v3 o, v3 d, float mint, float maxt, int any, float *t, float *u, float *v) {
    switch (sh->a.type) {
    case SH_RECTANGLE: case SH_DISK: {
        v3 ro = xf_point(sh->a.to_object, o), rd = xf_vector(sh->a.to_object, d);
        float hit = -ro.z / rd.z;
        if (!(hit >= mint && hit <= maxt)) return 0;
        float lx = ro.x + hit * rd.x, ly = ro.y + hit * rd.y;                 /* Ray::operator()(t) = o + t * d */
        if (sh->a.type == SH_RECTANGLE ? (fabsf(lx) <= 1 && fabsf(ly) <= 1) : (lx * lx + ly * ly <= 1)) { *t = hit; *u = lx; *v = ly; return 1; }
        return 0;
    }
    case SH_SPHERE: {
        double ox = (double) o.x - (double) sh->center.x, oy = (double) o.y - (double) sh->center.y, oz = (double) o.z - (double) sh->center.z;
        double dx = d.x, dy = d.y, dz = d.z;
        double A = dx * dx + dy * dy + dz * dz, B = 2 * (ox * dx + oy * dy + oz * dz), C = (ox * ox + oy * oy + oz * oz) - (double) (sh->a.radius * sh->a.radius);
        double nearT, farT;
        if (!solve_quadratic_double(A, B, C, &nearT, &farT)) return 0;
        if (any) {
            if (nearT > maxt || farT < mint) return 0;
            if (nearT < mint && farT > maxt) return 0;
            *t = 0; return 1;
        }
        if (!(nearT <= maxt && farT >= mint)) return 0;
        if (nearT < mint) { if (farT > maxt) return 0; *t = (float) farT; } else *t = (float) nearT;
        *u = 0; *v = 0; return 1;
    }
    case SH_CYLINDER: {
        v3 ro = xf_point(sh->a.to_object, o), rd = xf_vector(sh->a.to_object, d);
        double ox = ro.x, oy = ro.y, dx = rd.x, dy = rd.y;
        double A = dx * dx + dy * dy, B = 2 * (dx * ox + dy * oy), C = ox * ox + oy * oy - (double) (sh->a.radius * sh->a.radius);
        double nearT, farT;
        if (!solve_quadratic_double(A, B, C, &nearT, &farT)) return 0;
        if (any) { if (nearT > maxt || farT < mint) return 0; }
        else if (!(nearT <= maxt && farT >= mint)) return 0;
        double zPosNear = (double) ro.z + (double) rd.z * nearT, zPosFar = (double) ro.z + (double) rd.z * farT;
        if (zPosNear >= 0 && zPosNear <= sh->a.length && nearT >= mint) { *t = (float) nearT; }
        else if (zPosFar >= 0 && zPosFar <= sh->a.length) { if (farT > maxt) return 0; *t = (float) farT; }
        else return 0;
        *u = 0; *v = 0; return 1;
    }
    }
    return 0;
}
/* Shape::fillIntersectionRecord: rectangle.cpp:155-168, disk.cpp:172-200, sphere.cpp:196-245, cylinder.cpp:203-233: p, normals and dpdu
 * (uv / dpdv feed textures only and are not restated).  Disk::fillIntersectionRecord leaves geoFrame unset in the reference; it is
 * defined as the shading normal here. */
static void analytic_fill(const analytic_t *sh, v3 o, v3 d, float t, float lx, float ly, v3 *p, v3 *ng, v3 *ns, v3 *dpdu) {
    *p = add(o, scale(d, t));
    switch (sh->a.type) {
    case SH_RECTANGLE: *ng = sh->n; *ns = sh->n; *dpdu = sh->dpdu; break;
    case SH_DISK: {
        float r = sqrtf(lx * lx + ly * ly), invR = (r == 0) ? 0.0f : (1.0f / r);
        float cosPhi = lx * invR, sinPhi = ly * invR;
        *dpdu = r != 0 ? xf_vector(sh->a.to_world, V(cosPhi, sinPhi, 0)) : xf_vector(sh->a.to_world, V(1, 0, 0));
        *ns = sh->n; *ng = sh->n; break;
    }
    case SH_SPHERE: {
        *p = add(sh->center, scale(normalize(sub(*p, sh->center)), sh->a.radius));           /* SINGLE_PRECISION re-projection */
        v3 local = xf_vector(sh->a.to_object, sub(*p, sh->center));
        *dpdu = xf_vector(sh->a.to_world, scale(V(-local.y, local.x, 0), 2 * M_PI_F));
        v3 n = normalize(sub(*p, sh->center));
        if (sh->a.flags & 1u) n = scale(n, -1.0f);
        *ng = n; *ns = n; break;
    }
    default: {
        v3 local = xf_point(sh->a.to_object, *p);
        *dpdu = xf_vector(sh->a.to_world, scale(V(-local.y, local.x, 0), 2 * M_PI_F));
        v3 dpdv = xf_vector(sh->a.to_world, V(0, 0, sh->a.length));
        v3 n = cross(normalize(*dpdu), normalize(dpdv));
        *p = add(*p, scale(n, sh->a.radius - sqrtf(local.x * local.x + local.y * local.y)));
        if (sh->a.flags & 1u) n = scale(n, -1.0f);
        *ng = n; *ns = n; break;
    }
    }
}
#ifndef INV_TWOPI
#define INV_TWOPI 0.15915494309189533577f
#endif
/* its.uv / its.dpdu / its.dpdv of an analytic hit (rectangle.cpp:161-163, disk.cpp:173-193, sphere.cpp:218-245, cylinder.cpp:204-216): needed by textures
 * only, so evaluated on demand.  (lx, ly) = the hit's local coordinates as the intersection routine stored them (rectangle, disk); p_ray = ray(t) */
static void analytic_uv(const analytic_t *sh, float lx, float ly, v3 p_ray, float *uvx, float *uvy, v3 *dpdu, v3 *dpdv) {
    switch (sh->a.type) {
    case SH_RECTANGLE: *uvx = 0.5f * (lx + 1); *uvy = 0.5f * (ly + 1); *dpdu = sh->dpdu; *dpdv = xf_vector(sh->a.to_world, V(0, 2, 0)); break;
    case SH_DISK: {
        float r = sqrtf(lx * lx + ly * ly), invR = (r == 0) ? 0.0f : (1.0f / r);
        float phi = atan2f(ly, lx); if (phi < 0) phi += 2 * M_PI_F;
        float cosPhi = lx * invR, sinPhi = ly * invR;
        if (r != 0) { *dpdu = xf_vector(sh->a.to_world, V(cosPhi, sinPhi, 0)); *dpdv = xf_vector(sh->a.to_world, V(-sinPhi, cosPhi, 0)); }
        else { *dpdu = xf_vector(sh->a.to_world, V(1, 0, 0)); *dpdv = xf_vector(sh->a.to_world, V(0, 1, 0)); }
        *uvx = r; *uvy = phi * INV_TWOPI; break;
    }
    case SH_SPHERE: {
        v3 p = add(sh->center, scale(normalize(sub(p_ray, sh->center)), sh->a.radius));
        v3 local = xf_vector(sh->a.to_object, sub(p, sh->center));
        float theta = acosf(minf(1.0f, maxf(-1.0f, local.z / sh->a.radius))), phi = atan2f(local.y, local.x); if (phi < 0) phi += 2 * M_PI_F;
        *uvx = phi * (0.5f * INV_PI); *uvy = theta * INV_PI;
        *dpdu = xf_vector(sh->a.to_world, scale(V(-local.y, local.x, 0), 2 * M_PI_F));
        float zrad = sqrtf(local.x * local.x + local.y * local.y), cosPhi = 0, sinPhi = 1;
        if (zrad > 0) { float inv = 1.0f / zrad; cosPhi = local.x * inv; sinPhi = local.y * inv; }
        *dpdv = xf_vector(sh->a.to_world, scale(V(local.z * cosPhi, local.z * sinPhi, -sinf(theta) * sh->a.radius), M_PI_F));
        break;
    }
    default: {
        v3 local = xf_point(sh->a.to_object, p_ray);
        float phi = atan2f(local.y, local.x); if (phi < 0) phi += 2 * M_PI_F;
        *uvx = phi / (2 * M_PI_F); *uvy = local.z / sh->a.length;
        *dpdu = xf_vector(sh->a.to_world, scale(V(-local.y, local.x, 0), 2 * M_PI_F)); *dpdv = xf_vector(sh->a.to_world, V(0, 0, sh->a.length));
        break;
    }
    }
}
/* Shape::samplePosition: rectangle.cpp:215-221, disk.cpp:252-260, cylinder.cpp:235-250 (sphere: see analytic_sample_direct) */
static void analytic_sample_position(const analytic_t *sh, float sx, float sy, v3 *p, v3 *n) {
    switch (sh->a.type) {
    case SH_RECTANGLE: *p = xf_point(sh->a.to_world, V(sx * 2 - 1, sy * 2 - 1, 0)); *n = sh->n; break;
    case SH_DISK: { float px, py; disk_concentric(sx, sy, &px, &py); *p = xf_point(sh->a.to_world, V(px, py, 0)); *n = sh->n; break; }
    default: {
        float sinTheta, cosTheta; sincos_2pi(sy, &sinTheta, &cosTheta);
        v3 pl = V(cosTheta * sh->a.radius, sinTheta * sh->a.radius, sx * sh->a.length), nl = V(cosTheta, sinTheta, 0.0f);
        if (sh->a.flags & 1u) nl = scale(nl, -1.0f);
        *p = xf_point(sh->a.to_world, pl); *n = normalize(xf_normal(sh->a.to_object, nl)); break;
    }
    }
}
/* warp.cpp:25-31 squareToUniformSphere, :54-63 squareToUniformCone */
static v3 uniform_sphere(float sx, float sy) {
    float z = 1.0f - 2.0f * sy, r = sqrtf(maxf(1.0f - z * z, 0.0f)), sinPhi, cosPhi;
    sincos_2pi(sx, &sinPhi, &cosPhi);
    return V(r * cosPhi, r * sinPhi, z);
}
static v3 uniform_cone(float cosCutoff, float sx, float sy) {
    float cosTheta = (1 - sx) + sx * cosCutoff, sinTheta = sqrtf(maxf(1.0f - cosTheta * cosTheta, 0.0f)), sinPhi, cosPhi;
    sincos_2pi(sy, &sinPhi, &cosPhi);
    return V(cosPhi * sinTheta, sinPhi * sinTheta, cosTheta);
}
/* Shape::sampleDirect (src/librender/shape.cpp:102-115) for rectangle / disk / cylinder; Sphere::sampleDirect (sphere.cpp:275-346).
 * Outputs the solid-angle density in *pdf. */
static void analytic_sample_direct(const analytic_t *sh, v3 ref, float sx, float sy, v3 *p, v3 *n, v3 *dOut, float *dist, float *pdf) {
    if (sh->a.type == SH_SPHERE) {
        const float radius = sh->a.radius;
        v3 refToCenter = sub(sh->center, ref);
        float refDist2 = dot(refToCenter, refToCenter), invRefDist = 1.0f / sqrtf(refDist2);
        float sinAlpha = radius * invRefDist;
        if (sinAlpha < 1 - EPSILON) {
            float cosAlpha = sqrtf(maxf(1.0f - sinAlpha * sinAlpha, 0.0f));
            v3 fn = scale(refToCenter, invRefDist), fs, ft; coordinate_system(fn, &fs, &ft);        /* Frame(n), include/mitsuba/core/frame.h:53-55 */
            v3 c = uniform_cone(cosAlpha, sx, sy);
            v3 d = add(add(scale(fs, c.x), scale(ft, c.y)), scale(fn, c.z));
            *pdf = (0.5f * INV_PI) / (1 - cosAlpha);                                                /* INV_TWOPI / (1 - cosCutoff), warp.h:74-76 */
            float projDist = dot(refToCenter, d);
            float baseT = refDist2 / projDist;
            v3 query = add(ref, scale(d, baseT));
            v3 queryToCenter = sub(sh->center, query);
            float queryDist2 = dot(queryToCenter, queryToCenter), queryProjDist = dot(queryToCenter, d);
            float A = 1.0f, B = -2 * queryProjDist, C = queryDist2 - radius * radius, nearT, farT;
            if (!solve_quadratic(A, B, C, &nearT, &farT)) nearT = queryProjDist;
            *dist = baseT + nearT;
            *n = normalize(sub(scale(d, nearT), queryToCenter));
            *p = add(sh->center, scale(*n, radius));
            *dOut = d;
        } else {
            v3 dl = uniform_sphere(sx, sy);
            *p = add(sh->center, scale(dl, radius)); *n = dl;
            v3 d = sub(*p, ref);
            float dist2 = dot(d, d); *dist = sqrtf(dist2);
            { float r = 1.0f / *dist; d = scale(d, r); }
            *dOut = d;
            *pdf = sh->inv_area * dist2 / fabsf(dot(d, *n));
        }
        if (sh->a.flags & 1u) *n = scale(*n, -1.0f);
        return;
    }
    analytic_sample_position(sh, sx, sy, p, n);
    v3 d = sub(*p, ref);
    float distSquared = dot(d, d); *dist = sqrtf(distSquared);
    { float r = 1.0f / *dist; d = scale(d, r); }
    float dp = fabsf(dot(d, *n));
    *pdf = sh->inv_area * (dp != 0 ? (distSquared / dp) : 0.0f);
    *dOut = d;
}
/* Shape::pdfDirect, measure = ESolidAngle (shape.cpp:117-126); Sphere::pdfDirect (sphere.cpp:348-379) */
static float analytic_pdf_direct(const analytic_t *sh, v3 ref, v3 d, v3 n, float dist) {
    if (sh->a.type == SH_SPHERE) {
        v3 refToCenter = sub(sh->center, ref);
        float invRefDist = 1.0f / length3(refToCenter), sinAlpha = sh->a.radius * invRefDist;
        if (sinAlpha < 1 - EPSILON) { float cosAlpha = sqrtf(maxf(1 - sinAlpha * sinAlpha, 0.0f)); return (0.5f * INV_PI) / (1 - cosAlpha); }
    }
    if (sh->a.type == SH_SPHERE) return sh->inv_area * dist * dist / fabsf(dot(d, n));   /* sphere.cpp:370-372 (association as written there) */
    return sh->inv_area * (dist * dist) / fabsf(dot(d, n));
}
/* derived constants + Shape::getAABB (rectangle.cpp:112-119, disk.cpp:118-130, sphere.cpp:137-142, cylinder.cpp:256-276) */
static void analytic_prepare(analytic_t *sh) {
    const float *M = sh->a.to_world; v3 lo = V(INFINITY, INFINITY, INFINITY), hi = V(-INFINITY, -INFINITY, -INFINITY);
#define EXPAND(pt) do { v3 q_ = (pt); lo = V(minf(lo.x, q_.x), minf(lo.y, q_.y), minf(lo.z, q_.z)); hi = V(maxf(hi.x, q_.x), maxf(hi.y, q_.y), maxf(hi.z, q_.z)); } while (0)
    sh->n = V(0, 0, 0); sh->dpdu = V(0, 0, 0); sh->center = V(0, 0, 0);
    switch (sh->a.type) {
    case SH_RECTANGLE: {
        sh->dpdu = xf_vector(M, V(2, 0, 0)); v3 dpdv = xf_vector(M, V(0, 2, 0));
        sh->n = normalize(xf_normal(sh->a.to_object, V(0, 0, 1)));
        sh->inv_area = 1.0f / (length3(sh->dpdu) * length3(dpdv));
        EXPAND(xf_point(M, V(-1, -1, 0))); EXPAND(xf_point(M, V(1, -1, 0))); EXPAND(xf_point(M, V(1, 1, 0))); EXPAND(xf_point(M, V(-1, 1, 0)));
        break;
    }
    case SH_DISK: {
        v3 dpdu = xf_vector(M, V(1, 0, 0));
        sh->n = normalize(xf_normal(sh->a.to_object, V(0, 0, 1)));
        sh->inv_area = 1.0f / (M_PI_F * length3(dpdu) * length3(dpdu));
        EXPAND(xf_point(M, V(1, 0, 0))); EXPAND(xf_point(M, V(-1, 0, 0))); EXPAND(xf_point(M, V(0, 1, 0))); EXPAND(xf_point(M, V(0, -1, 0)));
        break;
    }
    case SH_SPHERE: {
        sh->center = xf_point(M, V(0, 0, 0));
        sh->inv_area = 1 / (4 * M_PI_F * sh->a.radius * sh->a.radius);
        lo = V(sh->center.x - sh->a.radius, sh->center.y - sh->a.radius, sh->center.z - sh->a.radius);
        hi = V(sh->center.x + sh->a.radius, sh->center.y + sh->a.radius, sh->center.z + sh->a.radius);
        break;
    }
    default: {
        sh->inv_area = 1 / (2 * M_PI_F * sh->a.radius * sh->a.length);
        v3 x1 = xf_vector(M, V(sh->a.radius, 0, 0)), x2 = xf_vector(M, V(0, sh->a.radius, 0));
        v3 p0 = xf_point(M, V(0, 0, 0)), p1 = xf_point(M, V(0, 0, sh->a.length));
        float l[3], h[3];
        for (int i = 0; i < 3; ++i) {
            float range = sqrtf(comp(x1, i) * comp(x1, i) + comp(x2, i) * comp(x2, i));
            l[i] = minf(minf(INFINITY, comp(p0, i) - range), comp(p1, i) - range);
            h[i] = maxf(maxf(-INFINITY, comp(p0, i) + range), comp(p1, i) + range);
        }
        lo = V(l[0], l[1], l[2]); hi = V(h[0], h[1], h[2]);
        break;
    }
    }
#undef EXPAND
    sh->lo = lo; sh->hi = hi;
}

/* include/mitsuba/core/aabb.h:308-339 TAABB::rayIntersect(ray, nearT, farT) (dRcp = 1/d, include/mitsuba/core/ray.h:72-83) */
static int aabb_ray(v3 lo, v3 hi, v3 o, v3 d, float *nearT, float *farT) {
    float nt = -INFINITY, ft = INFINITY;
    for (int i = 0; i < 3; ++i) {
        float origin = comp(o, i), minv = comp(lo, i), maxv = comp(hi, i), di = comp(d, i);
        if (di == 0) { if (origin < minv || origin > maxv) return 0; }
        else {
            float rcp = 1.0f / di;
            float t1 = (minv - origin) * rcp, t2 = (maxv - origin) * rcp;
            if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; }
            nt = maxf(t1, nt); ft = minf(t2, ft);
            if (!(nt <= ft)) return 0;
        }
    }
    *nearT = nt; *farT = ft; return 1;
}
/* src/librender/skdtree.cpp:112-142 (closest, shadow=0) and :207-226 (any hit, shadow=1): AABB clip + adaptive epsilon */
static int clip_interval(const orc_scene *s, v3 o, v3 d, float rmint, float rmaxt, int shadow, float *mint, float *maxt) {
    float nt, ft;
    if (!aabb_ray(s->aabb_lo, s->aabb_hi, o, d, &nt, &ft)) return 0;
    float rayMinT = rmint;
    if (rayMinT == EPSILON) {
        float m = maxf(maxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
        if (!shadow) m = maxf(m, EPSILON);
        rayMinT *= m;
    }
    if (rayMinT > nt) nt = rayMinT;
    if (rmaxt < ft) ft = rmaxt;
    *mint = nt; *maxt = ft;
    return ft > nt;
}

/* conservative slab test for the oracle's own BVH (not part of the reference; only prunes) */
static inline int box_overlap(const bvh_node *n, v3 o, v3 inv, float mint, float maxt) {
    float t0 = mint, t1 = maxt;
    for (int i = 0; i < 3; ++i) {
        float a = (comp(n->lo, i) - comp(o, i)) * comp(inv, i), b = (comp(n->hi, i) - comp(o, i)) * comp(inv, i);
        if (a != a || b != b) continue;          /* 0 * inf: the ray lies in the slab plane -> do not prune on this axis */
        float lo = minf(a, b), hi = maxf(a, b);
        t0 = maxf(t0, lo); t1 = minf(t1, hi);
    }
    return t0 <= t1 * 1.0000005f + 1e-30f;
}
static inline int better(float t, uint32_t prim, int32_t inst, float bt, uint32_t bprim, int32_t binst) { return t < bt || (t == bt && (prim < bprim || (prim == bprim && inst < binst))); }

/* One primitive of a leaf: triangle (TriAccel), analytic shape or instance.  Instance::rayIntersect (src/shapes/instance.cpp:91-108): the ray
 * goes to the group's object space (trafo.inverse()(ray)), then ShapeKDTree::rayIntersect(ray, mint, maxt, t, temp) / (ray, mint, maxt)
 * (include/mitsuba/render/skdtree.h:431-452): clip [mint, maxt] against the group's kd-tree box, then walk the group's tree. */
static int traverse_from(const orc_scene *s, int root, int brute_first, int brute_count, v3 o, v3 d, float mint, float maxt, int shadow,
                         float *bt, uint32_t *bprim, int32_t *binst, float *bu, float *bv);
static int prim_intersect(const orc_scene *s, uint32_t prim, v3 o, v3 d, float mint, float best, int shadow, int brute, float *t, uint32_t *hprim, int32_t *hinst, float *u, float *v) {
    const uint32_t nt = s->d.n_tris, na = s->n_analytic;
    *hprim = prim; *hinst = -1;
    if (prim < nt) return triaccel_intersect(&s->accel[prim], o, d, mint, best, u, v, t);
    if (prim < nt + na) return analytic_intersect(&s->analytic[prim - nt], o, d, mint, best, shadow, t, u, v);
    const int32_t ii = (int32_t) (prim - nt - na); const orc_instance *in = &s->instances[ii]; const uint32_t g = in->group;
    v3 o2 = xf_point(in->to_object, o), d2 = xf_vector(in->to_object, d);
    float nearT, farT;
    if (!aabb_ray(s->group_lo[g], s->group_hi[g], o2, d2, &nearT, &farT)) return 0;
    float mi = mint > nearT ? mint : nearT, ma = best < farT ? best : farT;
    if (!(ma > mi)) return 0;
    int32_t dummy;
    if (!traverse_from(s, brute ? -1 : s->group_root[g], (int) s->group_first[g], (int) s->group_count[g], o2, d2, mi, ma, shadow, t, hprim, &dummy, u, v)) return 0;
    *hinst = ii; return 1;
}
/* root >= 0: walk the oracle's BVH from that node; root < 0: O(N) loop over group_prims[brute_first .. +brute_count) (cross-check) */
static int traverse_from(const orc_scene *s, int root, int brute_first, int brute_count, v3 o, v3 d, float mint, float maxt, int shadow,
                         float *bt, uint32_t *bprim, int32_t *binst, float *bu, float *bv) {
    int found = 0; float best = maxt; uint32_t bestPrim = 0xFFFFFFFFu; int32_t bestInst = -1;
    if (root < 0) {
        for (int i = 0; i < brute_count; ++i) {
            uint32_t prim = s->group_prims[brute_first + i], hp; int32_t hi; float u, v, t;
            if (prim_intersect(s, prim, o, d, mint, best, shadow, 1, &t, &hp, &hi, &u, &v)) {
                if (shadow) return 1;
                if (!found || better(t, hp, hi, best, bestPrim, bestInst)) { best = t; bestPrim = hp; bestInst = hi; *bu = u; *bv = v; found = 1; }
            }
        }
    } else {
        v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        int stack[128], sp = 0; stack[sp++] = root;
        while (sp) {
            const bvh_node *n = &s->nodes[stack[--sp]];
            if (!box_overlap(n, o, inv, mint, best)) continue;
            if (n->count > 0) {
                for (int i = 0; i < n->count; ++i) {
                    uint32_t prim = s->bvh_tris[n->first + i], hp; int32_t hi; float u, v, t;
                    if (prim_intersect(s, prim, o, d, mint, best, shadow, 0, &t, &hp, &hi, &u, &v)) {
                        if (shadow) return 1;
                        if (!found || better(t, hp, hi, best, bestPrim, bestInst)) { best = t; bestPrim = hp; bestInst = hi; *bu = u; *bv = v; found = 1; }
                    }
                }
            } else { stack[sp++] = n->left; stack[sp++] = n->right; }
        }
    }
    if (found) { *bt = best; *bprim = bestPrim; *binst = bestInst; }
    return found;
}
/* scene level: group 0 of the tables = the scene's own primitives */
static int traverse(const orc_scene *s, v3 o, v3 d, float mint, float maxt, int shadow, float *bt, uint32_t *bprim, int32_t *binst, float *bu, float *bv) {
    return traverse_from(s, s->group_root[s->n_groups], 0, 0, o, d, mint, maxt, shadow, bt, bprim, binst, bu, bv);
}
static int traverse_brute(const orc_scene *s, v3 o, v3 d, float mint, float maxt, int shadow, float *bt, uint32_t *bprim, int32_t *binst, float *bu, float *bv) {
    return traverse_from(s, -1, (int) s->group_first[s->n_groups], (int) s->group_count[s->n_groups], o, d, mint, maxt, shadow, bt, bprim, binst, bu, bv);
}

/* include/mitsuba/render/skdtree.h:343-428 fillIntersectionRecord<true> + src/libcore/util.cpp:605-610 computeShadingFrame */
static void fill_hit(const orc_scene *s, v3 o, v3 d, float t, uint32_t prim, int32_t inst, float u, float v, hit_t *h) {
    h->instance = inst;
    if (prim >= s->d.n_tris) {                       /* skdtree.h:421-427: shape->fillIntersectionRecord, computeShadingFrame, wi */
        const analytic_t *sh = &s->analytic[prim - s->d.n_tris]; v3 dpdu;
        h->valid = 1; h->t = t; h->u = u; h->v = v; h->prim = 0; h->shape = s->d.n_shapes + (prim - s->d.n_tris);
        analytic_fill(sh, o, d, t, u, v, &h->p, &h->ng, &h->ns, &dpdu); h->dpdu = dpdu; h->dpdv = V(0, 0, 0); h->ng_raw = h->ng;
        { v3 du, dv; analytic_uv(sh, u, v, add(o, scale(d, t)), &h->uvx, &h->uvy, &du, &dv); h->dpdv = dv; }
        h->s = normalize(sub(dpdu, scale(h->ns, dot(h->ns, dpdu))));
        h->tt = cross(h->ns, h->s);
        v3 md = neg(d);
        h->wi = V(dot(md, h->s), dot(md, h->tt), dot(md, h->ns));
        h->material = sh->a.bsdf; h->emitter = sh->a.emitter;
        return;
    }
    uint32_t shape = s->tri_shape[prim]; const orc_shape *sh = &s->shapes[shape];
    uint32_t i0 = s->idx[prim * 3], i1 = s->idx[prim * 3 + 1], i2 = s->idx[prim * 3 + 2];
    v3 p0 = vert(s, i0), p1 = vert(s, i1), p2 = vert(s, i2);
    float bx = 1 - u - v, by = u, bz = v;
    h->valid = 1; h->t = t; h->u = u; h->v = v; h->prim = prim - sh->first_tri; h->shape = shape;
    h->p = add(add(scale(p0, bx), scale(p1, by)), scale(p2, bz));
    const orc_instance *in = inst >= 0 ? &s->instances[inst] : NULL;
    if (in) {   /* Instance::fillIntersectionRecord (instance.cpp:126-141): fillIntersectionRecord<false> in object space -> p = ray(t) of the object-space ray */
        v3 o2 = xf_point(in->to_object, o), d2 = xf_vector(in->to_object, d);
        h->p = add(o2, scale(d2, t));
    }
    v3 side1 = sub(p1, p0), side2 = sub(p2, p0);
    v3 fn = cross(side1, side2);
    float len = sqrtf(dot(fn, fn));
    if (!is_zero(fn)) { float r = 1.0f / len; fn = scale(fn, r); }
    h->ng_raw = fn;
    int smooth = s->nrm != NULL && !(sh->flags & 1u);
    if (smooth) {
        v3 n = add(add(scale(vnrm(s, i0), bx), scale(vnrm(s, i1), by)), scale(vnrm(s, i2), bz));
        h->ns = normalize(n);
        if (dot(fn, h->ns) < 0) fn = neg(fn);
    } else h->ns = fn;
    h->ng = fn;
    v3 dpdu = side1;   /* skdtree.h:373-380: the UV tangent of the triangle when the mesh has texcoords, else the first edge */
    h->uvx = by; h->uvy = bz;                                   /* skdtree.h:402-408 */
    v3 dpdv = side2;
    if (s->uv && (sh->flags & 2u)) {
        const float *t0 = &s->uv[i0 * 2], *t1 = &s->uv[i1 * 2], *t2 = &s->uv[i2 * 2];
        h->uvx = (t0[0] * bx + t1[0] * by) + t2[0] * bz; h->uvy = (t0[1] * bx + t1[1] * by) + t2[1] * bz;
        dpdu = V(s->tangents[prim * 6], s->tangents[prim * 6 + 1], s->tangents[prim * 6 + 2]);
        dpdv = V(s->tangents[prim * 6 + 3], s->tangents[prim * 6 + 4], s->tangents[prim * 6 + 5]);
    }
    if (in) {   /* instance.cpp:134-139: normals through the inverse transpose, dpdu / p through the forward transform; then the scene-level computeShadingFrame + wi */
        h->ns = normalize(xf_normal(in->to_object, h->ns)); h->ng = normalize(xf_normal(in->to_object, h->ng));
        dpdu = xf_vector(in->to_world, dpdu); dpdv = xf_vector(in->to_world, dpdv); h->p = xf_point(in->to_world, h->p);
        h->ng_raw = h->ng;
    }
    h->dpdu = dpdu; h->dpdv = dpdv;
    h->s = normalize(sub(dpdu, scale(h->ns, dot(h->ns, dpdu))));
    h->tt = cross(h->ns, h->s);
    v3 md = neg(d);
    h->wi = V(dot(md, h->s), dot(md, h->tt), dot(md, h->ns));   /* Frame::toLocal */
    h->material = sh->bsdf; h->emitter = sh->emitter;
}
static int g_brute = 0;
void orc_set_brute(int b) { g_brute = b; }
static int ray_intersect(const orc_scene *s, v3 o, v3 d, float rmint, float rmaxt, hit_t *h, int brute) {
    brute |= g_brute;
    float mint, maxt, t = 0, u = 0, v = 0; uint32_t prim = 0; int32_t inst = -1;
    h->valid = 0;
    if (!clip_interval(s, o, d, rmint, rmaxt, 0, &mint, &maxt)) return 0;
    int found = brute ? traverse_brute(s, o, d, mint, maxt, 0, &t, &prim, &inst, &u, &v) : traverse(s, o, d, mint, maxt, 0, &t, &prim, &inst, &u, &v);
    if (!found) return 0;
    fill_hit(s, o, d, t, prim, inst, u, v, h);
    return 1;
}
static int ray_occluded(const orc_scene *s, v3 o, v3 d, float rmint, float rmaxt) {
    float mint, maxt, t, u, v; uint32_t prim; int32_t inst;
    if (!clip_interval(s, o, d, rmint, rmaxt, 1, &mint, &maxt)) return 0;
    if (g_brute) return traverse_brute(s, o, d, mint, maxt, 1, &t, &prim, &inst, &u, &v);
    return traverse(s, o, d, mint, maxt, 1, &t, &prim, &inst, &u, &v);
}
static v3 to_world(const hit_t *h, v3 w) { return add(add(scale(h->s, w.x), scale(h->tt, w.y)), scale(h->ns, w.z)); }
static v3 to_local(const hit_t *h, v3 w) { return V(dot(w, h->s), dot(w, h->tt), dot(w, h->ns)); }

static void hit_out(const hit_t *h, float *o) {
    o[0] = h->t; o[1] = h->p.x; o[2] = h->p.y; o[3] = h->p.z; o[4] = h->ng.x; o[5] = h->ng.y; o[6] = h->ng.z;
    o[7] = h->ns.x; o[8] = h->ns.y; o[9] = h->ns.z; o[10] = h->s.x; o[11] = h->s.y; o[12] = h->s.z;
    o[13] = h->u; o[14] = h->v; o[15] = h->wi.x; o[16] = h->wi.y; o[17] = h->wi.z; o[18] = (float) h->prim; o[19] = (float) h->shape; o[20] = (float) h->instance; o[21] = h->uvx; o[22] = h->uvy;
}
int orc_ray_intersect(const orc_scene *s, const float *r, float *out) { hit_t h; int ok = ray_intersect(s, V(r[0], r[1], r[2]), V(r[4], r[5], r[6]), r[3], r[7], &h, 0); if (ok) hit_out(&h, out); return ok; }
int orc_ray_intersect_brute(const orc_scene *s, const float *r, float *out) { hit_t h; int ok = ray_intersect(s, V(r[0], r[1], r[2]), V(r[4], r[5], r[6]), r[3], r[7], &h, 1); if (ok) hit_out(&h, out); return ok; }
int orc_ray_occluded(const orc_scene *s, const float *r) { return ray_occluded(s, V(r[0], r[1], r[2]), V(r[4], r[5], r[6]), r[3], r[7]); }

/* ------------------------------------------------------------------------------------------------ camera */
/* src/sensors/perspective.cpp:271-287 sampleRayDifferential (ray part), include/mitsuba/core/transform.h:108-125 */
static void camera_ray(const orc_scene *s, float sx, float sy, v3 *o, v3 *d, float *mint, float *maxt) {
    const float *m = s->d.sample_to_camera;
    float px = sx * s->inv_res_x, py = sy * s->inv_res_y, pz = 0.0f;
    float x = m[0] * px + m[1] * py + m[2] * pz + m[3];
    float y = m[4] * px + m[5] * py + m[6] * pz + m[7];
    float z = m[8] * px + m[9] * py + m[10] * pz + m[11];
    float w = m[12] * px + m[13] * py + m[14] * pz + m[15];
    v3 nearP = V(x, y, z);
    if (w != 1.0f) { float r = 1.0f / w; nearP = scale(nearP, r); }   /* TPoint3::operator/ : recip then multiply */
    v3 dl = normalize(nearP);
    float invZ = 1.0f / dl.z;
    *mint = s->d.near_clip * invZ; *maxt = s->d.far_clip * invZ;
    const float *c = s->d.cam_to_world;
    *o = V(c[3], c[7], c[11]);
    *d = V(c[0] * dl.x + c[1] * dl.y + c[2] * dl.z, c[4] * dl.x + c[5] * dl.y + c[6] * dl.z, c[8] * dl.x + c[9] * dl.y + c[10] * dl.z);
}
/* ray differentials of the sensor ray: perspective.cpp:290-295 (rxDirection / ryDirection through nearP + m_dx / m_dy, :159-163), then
 * RayDifferential::scaleDifferential(1 / sqrt(spp)) (include/mitsuba/core/ray.h:163-168; integrator.cpp:145-146, 182 / :403-405, 425) */
static void camera_differentials(const orc_scene *s, float sx, float sy, v3 d, v3 *rxd, v3 *ryd) {
    const float *m = s->d.sample_to_camera;
    float px = sx * s->inv_res_x, py = sy * s->inv_res_y;
    float x = m[0] * px + m[1] * py + m[3], y = m[4] * px + m[5] * py + m[7], z = m[8] * px + m[9] * py + m[11], w = m[12] * px + m[13] * py + m[15];
    x = m[0] * px + m[1] * py + m[2] * 0.0f + m[3]; y = m[4] * px + m[5] * py + m[6] * 0.0f + m[7]; z = m[8] * px + m[9] * py + m[10] * 0.0f + m[11]; w = m[12] * px + m[13] * py + m[14] * 0.0f + m[15];
    v3 nearP = V(x, y, z);
    if (w != 1.0f) { float r = 1.0f / w; nearP = scale(nearP, r); }
    const float *c = s->d.cam_to_world;
    v3 a = normalize(add(nearP, s->cam_dx)), b = normalize(add(nearP, s->cam_dy));
    v3 rx = V(c[0] * a.x + c[1] * a.y + c[2] * a.z, c[4] * a.x + c[5] * a.y + c[6] * a.z, c[8] * a.x + c[9] * a.y + c[10] * a.z);
    v3 ry = V(c[0] * b.x + c[1] * b.y + c[2] * b.z, c[4] * b.x + c[5] * b.y + c[6] * b.z, c[8] * b.x + c[9] * b.y + c[10] * b.z);
    const float amount = 1.0f / sqrtf((float) s->d.spp);
    *rxd = add(d, scale(sub(rx, d), amount)); *ryd = add(d, scale(sub(ry, d), amount));
}
void orc_camera_ray(const orc_scene *s, float sx, float sy, float *o8) { v3 o, d; camera_ray(s, sx, sy, &o, &d, &o8[3], &o8[7]); o8[0] = o.x; o8[1] = o.y; o8[2] = o.z; o8[4] = d.x; o8[5] = d.y; o8[6] = d.z; }

/* ------------------------------------------------------------------------------------------------ BSDFs */
#define BSDF_FLAG_TWOSIDED 1u
/* BSDF type bits that matter on this path: ESmooth (all supported BSDFs are smooth), EBackSide (twosided.cpp:99-102) */
enum { BSDF_DIFFUSE = 0, BSDF_ROUGHCONDUCTOR = 1, BSDF_CONDUCTOR = 2, BSDF_DIELECTRIC = 3, BSDF_PLASTIC = 4, BSDF_ROUGHDIELECTRIC = 5, BSDF_DIFFTRANS = 6, BSDF_ROUGHPLASTIC = 7, BSDF_THINDIELECTRIC = 8, BSDF_MASK = 9, BSDF_MIXTURE = 10, BSDF_BUMPMAP = 11, BSDF_NORMALMAP = 12, BSDF_NULL = 13, BSDF_ROUGHDIFFUSE = 14, BSDF_PHONG = 15, BSDF_WARD = 16, BSDF_COATING = 17, BSDF_BLEND = 18, BSDF_ROUGHCOATING = 19 };
#define BSDF_FLAG_NONLINEAR 4u
/* BSDF type has ETransmission or EBackSide -> dRec.refN = 0 (records.inl:160-164): twosided wrapper; dielectric (dielectric.cpp:199-202) */
static int material_has_backside(const orc_material *m) { return (m->flags & BSDF_FLAG_TWOSIDED) != 0 || m->type == BSDF_DIELECTRIC || m->type == BSDF_ROUGHDIELECTRIC || m->type == BSDF_DIFFTRANS || m->type == BSDF_THINDIELECTRIC || m->type == BSDF_NULL; }
/* BSDF::ESmooth: a `diffuse` whose reflectance is identically zero registers NO component (src/bsdfs/diffuse.cpp:99-102), so its type
 * is 0 and MIPathTracer::Li skips emitter sampling -- and the sampler request that goes with it (path.cpp:174-176) */
static int material_is_smooth(const orc_material *m) {
    if (m->type == BSDF_DIFFUSE) return ((m->flags >> 8) & 0xFFFFu) != 0 || maxf(maxf(m->reflectance[0], m->reflectance[1]), m->reflectance[2]) > 0;   /* a textured reflectance always registers the component */
    if (m->type == BSDF_CONDUCTOR || m->type == BSDF_DIELECTRIC || m->type == BSDF_THINDIELECTRIC || m->type == BSDF_NULL) return 0;     /* delta components only (conductor.cpp:201-202, dielectric.cpp:199-202, thindielectric.cpp:117-120) */
    return 1;
}

/* src/bsdfs/diffuse.cpp:112-153 */
static v3 diffuse_eval(const orc_material *m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return V(0, 0, 0);
    float f = INV_PI * wo.z;
    return V(m->reflectance[0] * f, m->reflectance[1] * f, m->reflectance[2] * f);
}
static float diffuse_pdf(v3 wi, v3 wo) { if (wi.z <= 0 || wo.z <= 0) return 0.0f; return INV_PI * wo.z; }
static v3 diffuse_sample(const orc_material *m, v3 wi, float u, float v, v3 *wo, float *pdf, float *eta) {
    if (wi.z <= 0) return V(0, 0, 0);
    *wo = cos_hemisphere(u, v); *eta = 1.0f; *pdf = INV_PI * wo->z;
    return V(m->reflectance[0], m->reflectance[1], m->reflectance[2]);
}
/* ---- rough conductor: src/bsdfs/roughconductor.cpp:260-416 over src/bsdfs/microfacet.h (isotropic alpha, Beckmann / GGX,
 * visible-normal sampling).  Transcendentals come from libm here and from the GPU math library on the device, so this BSDF is
 * tolerance-pinned (not bit-pinned) between oracle and HIP path; math::fastexp/fastlog are double exp/log in the reference
 * (include/mitsuba/core/math.h:185-199). */
static inline float fastexpf_(float x) { return (float) exp((double) x); }
static inline float fastlogf_(float x) { return (float) log((double) x); }
/* src/libcore/math.cpp:25-53 erfinv (Giles), :55-72 erf (A&S 7.1.26) */
static float mi_erfinv(float x) {
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
static float mi_erf(float x) {
    const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f, p = 0.3275911f;
    float sign = copysignf(1.0f, x); x = fabsf(x);
    float t = 1.0f / (1.0f + p * x);
    float y = 1.0f - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * fastexpf_(-x * x);
    return sign * y;
}
#ifndef INV_TWOPI
#define INV_TWOPI 0.15915494309189533577f
#endif
#define RCPOVERFLOW 0x1p-128f
/* microfacet.h:190-237 eval: distr 0 Beckmann, 1 GGX, 2 Phong (isotropic) / Ashikhmin-Shirley (anisotropic); exponents :717-720, :559-570 */
static inline float mf_phong_exponent(float alpha) { return maxf(2.0f / (alpha * alpha) - 2.0f, 0.0f); }
static float mf_interp_exponent(float au, float av, v3 v) {
    const float eu = mf_phong_exponent(au), ev = mf_phong_exponent(av), sinTheta2 = 1.0f - v.z * v.z;
    if (au == av || sinTheta2 <= RCPOVERFLOW) return eu;
    float invSinTheta2 = 1 / sinTheta2, cosPhi2 = v.x * v.x * invSinTheta2, sinPhi2 = v.y * v.y * invSinTheta2;
    return eu * cosPhi2 + ev * sinPhi2;
}
static float mf_eval2(uint32_t distr, float au, float av, v3 m) {
    if (m.z <= 0) return 0.0f;
    float cosTheta2 = m.z * m.z;
    float beckmannExponent = ((m.x * m.x) / (au * au) + (m.y * m.y) / (av * av)) / cosTheta2;
    float result;
    if (distr == 0) result = fastexpf_(-beckmannExponent) / (M_PI_F * au * av * cosTheta2 * cosTheta2);
    else if (distr == 1) { float root = (1.0f + beckmannExponent) * cosTheta2; result = 1.0f / (M_PI_F * au * av * root * root); }
    else result = sqrtf((mf_phong_exponent(au) + 2) * (mf_phong_exponent(av) + 2)) * INV_TWOPI * powf(m.z, mf_interp_exponent(au, av, m));
    if (result * m.z < 1e-20f) result = 0;
    return result;
}
static float mf_eval(uint32_t distr, float alpha, v3 m) { return mf_eval2(distr, alpha, alpha, m); }
/* microfacet.h:476-517 smithG1 with the roughness projected on v (projectRoughness :545-556); Phong uses the Beckmann fit */
static float mf_smith_g1(uint32_t distr, float alpha, v3 v, v3 m) {
    if (dot(v, m) * v.z <= 0) return 0.0f;
    float temp = 1 - v.z * v.z;
    float tanTheta = temp <= 0.0f ? 0.0f : fabsf(sqrtf(temp) / v.z);       /* frame.h:122-127 */
    if (tanTheta == 0.0f) return 1.0f;
    if (distr != 1) {
        float a = 1.0f / (alpha * tanTheta);
        if (a >= 1.6f) return 1.0f;
        float aSqr = a * a;
        return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
    } else {
        float root = alpha * tanTheta;
        /* math::hypot2(1, root), src/libcore/math.cpp:74-88 */
        float r;
        if (1.0f > fabsf(root)) { r = root / 1.0f; r = 1.0f * sqrtf(1.0f + r * r); }
        else if (root != 0.0f) { r = 1.0f / root; r = fabsf(root) * sqrtf(1.0f + r * r); }
        else r = 0.0f;
        return 2.0f / (1.0f + r);
    }
}
/* microfacet.h:572-700 sampleVisible11 */
static void mf_sample_visible11(uint32_t distr, float thetaI, float sx, float sy, float *slx, float *sly) {
    const float SQRT_PI_INV = 1 / sqrtf(M_PI_F);
    if (distr == 0) {
        if (thetaI < 1e-4f) {
            float r = sqrtf(-fastlogf_(1.0f - sx)), ph = 2 * M_PI_F * sy;
            *slx = r * cosf(ph); *sly = r * sinf(ph); return;
        }
        float tanThetaI = tanf(thetaI), cotThetaI = 1 / tanThetaI;
        float a = -1, c = mi_erf(cotThetaI);
        float sample_x = maxf(sx, 1e-6f);
        float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
        float b = c - (1 + c) * powf(1 - sample_x, fit);
        float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * expf(-cotThetaI * cotThetaI));
        int it = 0;
        while (++it < 10) {
            if (!(b >= a && b <= c)) b = 0.5f * (a + c);
            float invErf = mi_erfinv(b);
            float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * expf(-invErf * invErf)) - sample_x;
            float derivative = normalization * (1 - invErf * tanThetaI);
            if (fabsf(value) < 1e-5f) break;
            if (value > 0) c = b; else a = b;
            b -= value / derivative;
        }
        *slx = mi_erfinv(b);
        *sly = mi_erfinv(2.0f * maxf(sy, 1e-6f) - 1.0f);
    } else {
        if (thetaI < 1e-4f) {
            float r = sqrtf(maxf(sx / (1 - sx), 0.0f)), ph = 2 * M_PI_F * sy;
            *slx = r * cosf(ph); *sly = r * sinf(ph); return;
        }
        float tanThetaI = tanf(thetaI), a = 1 / tanThetaI;
        float G1 = 2.0f / (1.0f + sqrtf(maxf(1.0f + 1.0f / (a * a), 0.0f)));
        float A = 2.0f * sx / G1 - 1.0f;
        if (fabsf(A) == 1) A -= copysignf(1.0f, A) * EPSILON;
        float tmp = 1.0f / (A * A - 1.0f), B = tanThetaI;
        float D = sqrtf(maxf(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.0f));
        float slope_x_1 = B * tmp - D, slope_x_2 = B * tmp + D;
        *slx = (A < 0.0f || slope_x_2 > 1.0f / tanThetaI) ? slope_x_1 : slope_x_2;
        float S;
        if (sy > 0.5f) { S = 1.0f; sy = 2.0f * (sy - 0.5f); } else { S = -1.0f; sy = 2.0f * (0.5f - sy); }
        float z = (sy * (sy * (sy * (-0.365728915865723f) + 0.790235037209296f) - 0.424965825137544f) + 0.000152998850436920f) /
                  (sy * (sy * (sy * (sy * 0.169507819808272f - 0.397203533833404f) - 0.232500544458471f) + 1.0f) - 0.539825872510702f);
        *sly = S * z * sqrtf(1.0f + *slx * *slx);
    }
}
/* microfacet.h:420-466 sampleVisible */
static v3 mf_sample_visible(uint32_t distr, float alpha, v3 wi_, float sx, float sy) {
    v3 wi = normalize(V(alpha * wi_.x, alpha * wi_.y, wi_.z));
    float theta = 0, phi = 0;
    if (wi.z < 0.99999f) { theta = acosf(wi.z); phi = atan2f(wi.y, wi.x); }
    float sinPhi = sinf(phi), cosPhi = cosf(phi);
    float slx, sly; mf_sample_visible11(distr, theta, sx, sy, &slx, &sly);
    float rx = cosPhi * slx - sinPhi * sly, ry = sinPhi * slx + cosPhi * sly;
    rx *= alpha; ry *= alpha;
    float normalization = 1.0f / sqrtf(rx * rx + ry * ry + 1.0f);
    return V(-rx * normalization, -ry * normalization, normalization);
}
/* microfacet.h:469-473 pdfVisible */
static float mf_pdf_visible(uint32_t distr, float alpha, v3 wi, v3 m) {
    if (wi.z == 0) return 0.0f;
    return mf_smith_g1(distr, alpha, wi, m) * fabsf(dot(wi, m)) * mf_eval(distr, alpha, m) / fabsf(wi.z);
}
/* microfacet.h:420-473 sampleVisible / pdfVisible with separate roughness along the tangent and bitangent */
static v3 mf_sample_visible2(uint32_t distr, float au, float av, v3 wi_, float sx, float sy) {
    v3 wi = normalize(V(au * wi_.x, av * wi_.y, wi_.z));
    float theta = 0, phi = 0;
    if (wi.z < 0.99999f) { theta = acosf(wi.z); phi = atan2f(wi.y, wi.x); }
    float sinPhi = sinf(phi), cosPhi = cosf(phi);
    float slx, sly; mf_sample_visible11(distr, theta, sx, sy, &slx, &sly);
    float rx = cosPhi * slx - sinPhi * sly, ry = sinPhi * slx + cosPhi * sly;
    rx *= au; ry *= av;
    float normalization = 1.0f / sqrtf(rx * rx + ry * ry + 1.0f);
    return V(-rx * normalization, -ry * normalization, normalization);
}
/* microfacet.h:545-556 projectRoughness */
static float mf_project_roughness(float au, float av, v3 v) {
    float invSinTheta2 = 1 / (1.0f - v.z * v.z);
    if (au == av || invSinTheta2 <= 0) return au;
    float cosPhi2 = v.x * v.x * invSinTheta2, sinPhi2 = v.y * v.y * invSinTheta2;
    return sqrtf(cosPhi2 * au * au + sinPhi2 * av * av);
}
static float mf_smith_g1_2(uint32_t distr, float au, float av, v3 v, v3 m) { return mf_smith_g1(distr, mf_project_roughness(au, av, v), v, m); }
/* microfacet.h:722-731 sampleFirstQuadrant (Ashikhmin-Shirley) */
static void mf_sample_first_quadrant(float eu, float ev, float u1, float *phi, float *exponent) {
    *phi = atanf(sqrtf((eu + 2.0f) / (ev + 2.0f)) * tanf(M_PI_F * u1 * 0.5f));
    float sinPhi = sinf(*phi), cosPhi = cosf(*phi);
    *exponent = eu * cosPhi * cosPhi + ev * sinPhi * sinPhi;
}
/* microfacet.h:286-392 sampleAll: all normals, density D(m) cos(theta_m) */
static v3 mf_sample_all(uint32_t distr, float au, float av, float sx, float sy, float *pdf) {
    float cosThetaM = 0.0f, sinPhiM, cosPhiM, alphaSqr;
    if (distr == 0 || distr == 1) {
        if (au == av) { float ph = (2.0f * M_PI_F) * sy; sinPhiM = sinf(ph); cosPhiM = cosf(ph); alphaSqr = au * au; }
        else {
            float phiM = atanf(av / au * tanf(M_PI_F + 2 * M_PI_F * sy)) + M_PI_F * floorf(2 * sy + 0.5f);
            sinPhiM = sinf(phiM); cosPhiM = cosf(phiM);
            float cosSc = cosPhiM / au, sinSc = sinPhiM / av; alphaSqr = 1.0f / (cosSc * cosSc + sinSc * sinSc);
        }
        if (distr == 0) {
            float tanThetaMSqr = alphaSqr * -fastlogf_(1.0f - sx);
            cosThetaM = 1.0f / sqrtf(1.0f + tanThetaMSqr);
            *pdf = (1.0f - sx) / (M_PI_F * au * av * cosThetaM * cosThetaM * cosThetaM);
        } else {
            float tanThetaMSqr = alphaSqr * sx / (1.0f - sx);
            cosThetaM = 1.0f / sqrtf(1.0f + tanThetaMSqr);
            float temp = 1 + tanThetaMSqr / alphaSqr;
            *pdf = INV_PI / (au * av * cosThetaM * cosThetaM * cosThetaM * temp * temp);
        }
    } else {
        const float eu = mf_phong_exponent(au), ev = mf_phong_exponent(av);
        float phiM, exponent;
        if (au == av) { phiM = (2.0f * M_PI_F) * sy; exponent = eu; }
        else if (sy < 0.25f) mf_sample_first_quadrant(eu, ev, 4 * sy, &phiM, &exponent);
        else if (sy < 0.5f) { mf_sample_first_quadrant(eu, ev, 4 * (0.5f - sy), &phiM, &exponent); phiM = M_PI_F - phiM; }
        else if (sy < 0.75f) { mf_sample_first_quadrant(eu, ev, 4 * (sy - 0.5f), &phiM, &exponent); phiM += M_PI_F; }
        else { mf_sample_first_quadrant(eu, ev, 4 * (1 - sy), &phiM, &exponent); phiM = 2 * M_PI_F - phiM; }
        sinPhiM = sinf(phiM); cosPhiM = cosf(phiM);
        cosThetaM = powf(sx, 1.0f / (exponent + 2.0f));
        *pdf = sqrtf((eu + 2.0f) * (ev + 2.0f)) * INV_TWOPI * powf(cosThetaM, exponent + 1.0f);
    }
    if (*pdf < 1e-20f) *pdf = 0;
    float sinThetaM = sqrtf(maxf(0.0f, 1 - cosThetaM * cosThetaM));
    return V(sinThetaM * cosPhiM, sinThetaM * sinPhiM, cosThetaM);
}
/* microfacet.h:239-274 sample / pdf: visible normals or all normals; Phong never samples visible normals (:141-145) */
typedef struct { uint32_t distr; float au, av; int visible; } mfd_t;
static mfd_t mfd_of_roughconductor(const orc_material *m) {      /* flags bit1 sampleVisible, bit3 anisotropic: alphaV in reflectance[0] */
    mfd_t d; d.distr = m->distr; d.au = maxf(m->alpha, 1e-4f); d.av = (m->flags & 8u) ? maxf(m->reflectance[0], 1e-4f) : d.au;
    d.visible = (m->flags & 2u) != 0 && m->distr != 2; return d;
}
static mfd_t mfd_of_roughdielectric(const orc_material *m) {     /* alphaV in k[0] when flags bit3 is set */
    mfd_t d; d.distr = m->distr; d.au = maxf(m->alpha, 1e-4f); d.av = (m->flags & 8u) ? maxf(m->k[0], 1e-4f) : d.au;
    d.visible = (m->flags & 2u) != 0 && m->distr != 2; return d;
}
/* roughdielectric.cpp:409-414: Walter et al.'s trick -- sample from a slightly wider distribution unless visible normals are sampled (scaleAlpha, microfacet.h:180-185) */
static mfd_t mfd_sampling(const mfd_t *d, float cosThetaI) {
    mfd_t s = *d;
    if (!d->visible) { float f = 1.2f - 0.2f * sqrtf(fabsf(cosThetaI)); s.au *= f; s.av *= f; }
    return s;
}
static float mfd_pdf(const mfd_t *d, v3 wi, v3 m) {
    if (d->visible) { if (wi.z == 0) return 0.0f; return mf_smith_g1_2(d->distr, d->au, d->av, wi, m) * fabsf(dot(wi, m)) * mf_eval2(d->distr, d->au, d->av, m) / fabsf(wi.z); }
    return mf_eval2(d->distr, d->au, d->av, m) * m.z;
}
static v3 mfd_sample(const mfd_t *d, v3 wi, float sx, float sy, float *pdf) {
    if (d->visible) { v3 m = mf_sample_visible2(d->distr, d->au, d->av, wi, sx, sy); *pdf = mfd_pdf(d, wi, m); return m; }
    return mf_sample_all(d->distr, d->au, d->av, sx, sy, pdf);
}
/* src/libcore/util.cpp:741-763 fresnelConductorExact, per RGB channel */
static v3 fresnel_conductor_exact(float cosThetaI, const float *eta, const float *k) {
    float cosThetaI2 = cosThetaI * cosThetaI, sinThetaI2 = 1 - cosThetaI2, sinThetaI4 = sinThetaI2 * sinThetaI2;
    float out[3];
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
static v3 rc_eval(const orc_material *m, v3 wi, v3 wo) {                /* roughconductor.cpp:260-297 */
    if (wi.z <= 0 || wo.z <= 0) return V(0, 0, 0);
    const mfd_t d = mfd_of_roughconductor(m);
    v3 H = normalize(add(wo, wi));
    float D = mf_eval2(d.distr, d.au, d.av, H);
    if (D == 0) return V(0, 0, 0);
    v3 F = mul(fresnel_conductor_exact(dot(wi, H), m->eta, m->k), V(m->specular[0], m->specular[1], m->specular[2]));
    float G = mf_smith_g1_2(d.distr, d.au, d.av, wi, H) * mf_smith_g1_2(d.distr, d.au, d.av, wo, H);
    float model = D * G / (4.0f * wi.z);
    return scale(F, model);
}
static float rc_pdf(const orc_material *m, v3 wi, v3 wo) {              /* roughconductor.cpp:299-324 */
    if (wi.z <= 0 || wo.z <= 0) return 0.0f;
    const mfd_t d = mfd_of_roughconductor(m);
    v3 H = normalize(add(wo, wi));
    if (d.visible) return mf_eval2(d.distr, d.au, d.av, H) * mf_smith_g1_2(d.distr, d.au, d.av, wi, H) / (4.0f * wi.z);
    return mfd_pdf(&d, wi, H) / (4 * fabsf(dot(wo, H)));
}
static v3 rc_sample(const orc_material *mt, v3 wi, float u, float v, v3 *wo, float *pdf, float *eta) {   /* roughconductor.cpp:373-425 */
    if (wi.z < 0) return V(0, 0, 0);
    const mfd_t d = mfd_of_roughconductor(mt);
    v3 m = mfd_sample(&d, wi, u, v, pdf);
    if (*pdf == 0) return V(0, 0, 0);
    float c = 2 * dot(wi, m);
    *wo = sub(scale(m, c), wi);                                  /* reflect(wi, m) = 2 dot(wi,m) m - wi */
    *eta = 1.0f;
    if (wo->z <= 0) return V(0, 0, 0);
    v3 F = mul(fresnel_conductor_exact(dot(wi, m), mt->eta, mt->k), V(mt->specular[0], mt->specular[1], mt->specular[2]));
    float weight;
    if (d.visible) weight = mf_smith_g1_2(d.distr, d.au, d.av, *wo, m);
    else weight = mf_eval2(d.distr, d.au, d.av, m) * (mf_smith_g1_2(d.distr, d.au, d.av, wi, m) * mf_smith_g1_2(d.distr, d.au, d.av, *wo, m)) * dot(wi, m) / (*pdf * wi.z);
    *pdf /= 4.0f * dot(*wo, m);
    return scale(F, weight);
}

/* dispatch incl. src/bsdfs/twosided.cpp:110-190 (flip wi/wo to the front side when wi.z < 0) */
static inline float luminance(v3 c) { return c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f; }   /* include/mitsuba/core/spectrum.h:725-727 */
/* ---- smooth conductor / dielectric / plastic: src/bsdfs/conductor.cpp:212-286, dielectric.cpp:224-342, plastic.cpp:248-453.
 * eval / pdf are the solid-angle-measure versions the integrator calls during emitter sampling (delta components contribute 0 there).
 * Material fields: conductor eta[3], k[3], specular[3]; dielectric eta[0] = intIOR / extIOR, specular = specularReflectance,
 * reflectance = specularTransmittance; plastic eta[0], specular, reflectance = diffuseReflectance, k[0] = fresnelDiffuseReflectance(1/eta)
 * (m_fdrInt, plastic.cpp:200), flag bit2 = nonlinear. */
/* src/libcore/util.cpp:653-683 fresnelDielectricExt */
static float fresnel_dielectric_ext(float cosThetaI_, float *cosThetaT_, float eta) {
    if (eta == 1) { *cosThetaT_ = -cosThetaI_; return 0.0f; }
    float scale = (cosThetaI_ > 0) ? 1 / eta : eta, cosThetaTSqr = 1 - (1 - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0f) { *cosThetaT_ = 0.0f; return 1.0f; }
    float cosThetaI = fabsf(cosThetaI_), cosThetaT = sqrtf(cosThetaTSqr);
    float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    *cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5f * (Rs * Rs + Rp * Rp);
}
static v3 conductor_sample(const orc_material *m, v3 wi, v3 *wo, float *pdf, float *eta, int *delta) {
    if (wi.z <= 0) return V(0, 0, 0);
    *wo = V(-wi.x, -wi.y, wi.z); *eta = 1.0f; *pdf = 1; *delta = 1;
    return mul(V(m->specular[0], m->specular[1], m->specular[2]), fresnel_conductor_exact(wi.z, m->eta, m->k));
}
/* src/bsdfs/thindielectric.cpp:206-258: a thin slab -- delta reflection or straight-through transmission (an ENull component: the path counts as unscattered,
 * *delta = 2), reflectance with the internal bounces summed (R' = R + TRT + TR^3T + ...).  Fields: eta[0], specular = specularReflectance, reflectance = specularTransmittance */
/* THIN_SIGNED_COS (set on the scene's thindielectric records when the integrator is volpath_simple): that integrator calls the sample() overload WITHOUT a pdf argument
 * (volpath_simple.cpp:234), which feeds the SIGNED cosine to fresnelDielectricExt (thindielectric.cpp:263; the other overload takes |cos|, :212) */
#define THIN_SIGNED_COS (1u << 30)
static v3 thindielectric_sample(const orc_material *m, v3 wi, float sx, v3 *wo, float *pdf, float *etaOut, int *delta) {
    float ct, R = fresnel_dielectric_ext((m->flags & THIN_SIGNED_COS) ? wi.z : fabsf(wi.z), &ct, m->eta[0]), T = 1 - R;
    if (R < 1) R += T * T * R / (1 - R * R);
    *etaOut = 1.0f;
    if (sx <= R) { *delta = 1; *wo = V(-wi.x, -wi.y, wi.z); *pdf = R; return V(m->specular[0], m->specular[1], m->specular[2]); }
    *delta = 2; *wo = neg(wi); *pdf = 1 - R; return V(m->reflectance[0], m->reflectance[1], m->reflectance[2]);
}
static v3 dielectric_sample(const orc_material *m, v3 wi, float sx, v3 *wo, float *pdf, float *etaOut, int *delta) {
    const float eta = m->eta[0], invEta = 1 / eta; float cosThetaT;
    float F = fresnel_dielectric_ext(wi.z, &cosThetaT, eta);
    *delta = 1;
    if (sx <= F) { *wo = V(-wi.x, -wi.y, wi.z); *etaOut = 1.0f; *pdf = F; return V(m->specular[0], m->specular[1], m->specular[2]); }
    float scale_ = -(cosThetaT < 0 ? invEta : eta);
    *wo = V(scale_ * wi.x, scale_ * wi.y, cosThetaT);
    *etaOut = cosThetaT < 0 ? eta : invEta; *pdf = 1 - F;
    float factor = cosThetaT < 0 ? invEta : eta;
    return scale(V(m->reflectance[0], m->reflectance[1], m->reflectance[2]), factor * factor);
}
/* plastic.cpp:204-207 / roughplastic.cpp:244-246 m_specularSamplingWeight = sAvg / (dAvg + sAvg) from the textures' getAverage(): derived once at scene creation
 * (orc_scene_create) into the otherwise unused eta[1] of the material copy, because a textured diffuse reflectance contributes its AVERAGE here, not its local value */
static float plastic_spec_weight(const orc_material *m) { return m->eta[1]; }
static v3 plastic_diffuse(const orc_material *m) {              /* diff /= 1 - fdrInt  (or the nonlinear form) */
    v3 diff = V(m->reflectance[0], m->reflectance[1], m->reflectance[2]); const float fdrInt = m->k[0];
    if (m->flags & BSDF_FLAG_NONLINEAR) return V(diff.x / (1.0f - diff.x * fdrInt), diff.y / (1.0f - diff.y * fdrInt), diff.z / (1.0f - diff.z * fdrInt));
    float r = 1.0f / (1 - fdrInt); return scale(diff, r);
}
static v3 plastic_eval(const orc_material *m, v3 wi, v3 wo) {
    if (wo.z <= 0 || wi.z <= 0) return V(0, 0, 0);
    const float eta = m->eta[0], invEta2 = 1 / (eta * eta); float ct;
    float Fi = fresnel_dielectric_ext(wi.z, &ct, eta), Fo = fresnel_dielectric_ext(wo.z, &ct, eta);
    return scale(plastic_diffuse(m), INV_PI * wo.z * invEta2 * (1 - Fi) * (1 - Fo));
}
static float plastic_prob_specular(const orc_material *m, float Fi) { float w = plastic_spec_weight(m); return (Fi * w) / (Fi * w + (1 - Fi) * (1 - w)); }
static float plastic_pdf(const orc_material *m, v3 wi, v3 wo) {
    if (wo.z <= 0 || wi.z <= 0) return 0.0f;
    float ct, Fi = fresnel_dielectric_ext(wi.z, &ct, m->eta[0]);
    return INV_PI * wo.z * (1 - plastic_prob_specular(m, Fi));
}
static v3 plastic_sample(const orc_material *m, v3 wi, float sx, float sy, v3 *wo, float *pdf, float *etaOut, int *delta) {
    if (wi.z <= 0) return V(0, 0, 0);
    const float eta = m->eta[0], invEta2 = 1 / (eta * eta); float ct;
    float Fi = fresnel_dielectric_ext(wi.z, &ct, eta);
    *etaOut = 1.0f;
    float probSpecular = plastic_prob_specular(m, Fi);
    if (sx < probSpecular) {
        *wo = V(-wi.x, -wi.y, wi.z); *pdf = probSpecular; *delta = 1;
        float r = 1.0f / probSpecular; return scale(scale(V(m->specular[0], m->specular[1], m->specular[2]), Fi), r);
    }
    *wo = cos_hemisphere((sx - probSpecular) / (1 - probSpecular), sy);
    float Fo = fresnel_dielectric_ext(wo->z, &ct, eta);
    *pdf = (1 - probSpecular) * (INV_PI * wo->z);
    return scale(plastic_diffuse(m), invEta2 * (1 - Fi) * (1 - Fo) / (1 - probSpecular));
}

/* ---- rough dielectric: src/bsdfs/roughdielectric.cpp:274-617 over the whole MicrofacetDistribution (flags bit1 sampleVisible, bit3 anisotropic with alphaV in k[0]).
 * Fields: alpha, distr, eta[0] = intIOR / extIOR, specular = specularReflectance, reflectance = specularTransmittance.  sample() draws one more
 * number from the path's sampler to choose reflection / refraction (EUsesSampler, roughdielectric.cpp:480-481). */
static inline float signum_(float v) { return copysignf(1.0f, v); }
static v3 rd_eval(const orc_material *m, v3 wi, v3 wo) {
    if (wi.z == 0) return V(0, 0, 0);
    const float etaM = m->eta[0], invEta = 1 / etaM; const mfd_t d = mfd_of_roughdielectric(m); const int reflect = wi.z * wo.z > 0; v3 H;
    if (reflect) H = normalize(add(wo, wi));
    else { float eta = wi.z > 0 ? etaM : invEta; H = normalize(add(wi, scale(wo, eta))); }
    H = scale(H, signum_(H.z));
    float D = mf_eval2(d.distr, d.au, d.av, H); if (D == 0) return V(0, 0, 0);
    float ct, F = fresnel_dielectric_ext(dot(wi, H), &ct, etaM);
    float G = mf_smith_g1_2(d.distr, d.au, d.av, wi, H) * mf_smith_g1_2(d.distr, d.au, d.av, wo, H);
    if (reflect) { float value = F * D * G / (4.0f * fabsf(wi.z)); return scale(V(m->specular[0], m->specular[1], m->specular[2]), value); }
    float eta = wi.z > 0.0f ? etaM : invEta;
    float sqrtDenom = dot(wi, H) + eta * dot(wo, H);
    float value = ((1 - F) * D * G * eta * eta * dot(wi, H) * dot(wo, H)) / (wi.z * sqrtDenom * sqrtDenom);
    float factor = wi.z > 0 ? invEta : etaM;
    return scale(V(m->reflectance[0], m->reflectance[1], m->reflectance[2]), fabsf(value * factor * factor));
}
static float rd_pdf(const orc_material *m, v3 wi, v3 wo) {
    const float etaM = m->eta[0], invEta = 1 / etaM; const mfd_t d0 = mfd_of_roughdielectric(m), d = mfd_sampling(&d0, wi.z); const int reflect = wi.z * wo.z > 0; v3 H; float dwh_dwo;
    if (reflect) { H = normalize(add(wo, wi)); dwh_dwo = 1.0f / (4.0f * dot(wo, H)); }
    else { float eta = wi.z > 0 ? etaM : invEta; H = normalize(add(wi, scale(wo, eta))); float sqrtDenom = dot(wi, H) + eta * dot(wo, H); dwh_dwo = (eta * eta * dot(wo, H)) / (sqrtDenom * sqrtDenom); }
    H = scale(H, signum_(H.z));
    float prob = mfd_pdf(&d, scale(wi, signum_(wi.z)), H);
    float ct, F = fresnel_dielectric_ext(dot(wi, H), &ct, etaM);
    prob *= reflect ? F : (1 - F);
    return fabsf(prob * dwh_dwo);
}
static v3 rd_sample(const orc_material *mt, v3 wi, float sx, float sy, float extra, v3 *wo, float *pdf, float *etaOut) {
    const float etaM = mt->eta[0], invEta = 1 / etaM; const mfd_t d = mfd_of_roughdielectric(mt), sd = mfd_sampling(&d, wi.z);
    v3 wiS = scale(wi, signum_(wi.z));
    float microfacetPDF; v3 m = mfd_sample(&sd, wiS, sx, sy, &microfacetPDF);
    if (microfacetPDF == 0) return V(0, 0, 0);
    *pdf = microfacetPDF;
    float cosThetaT, F = fresnel_dielectric_ext(dot(wi, m), &cosThetaT, etaM);
    int sampleReflection = 1; v3 weight = V(1, 1, 1); float dwh_dwo;
    if (extra > F) { sampleReflection = 0; *pdf *= 1 - F; } else *pdf *= F;
    if (sampleReflection) {
        float c = 2 * dot(wi, m); *wo = sub(scale(m, c), wi); *etaOut = 1.0f;
        if (wi.z * wo->z <= 0) return V(0, 0, 0);
        weight = mul(weight, V(mt->specular[0], mt->specular[1], mt->specular[2]));
        dwh_dwo = 1.0f / (4.0f * dot(*wo, m));
    } else {
        if (cosThetaT == 0) return V(0, 0, 0);
        float e = cosThetaT < 0 ? 1 / etaM : etaM;                                   /* refract(wi, n, eta, cosThetaT), src/libcore/util.cpp:769-774 */
        *wo = sub(scale(m, dot(wi, m) * e + cosThetaT), scale(wi, e));
        *etaOut = cosThetaT < 0 ? etaM : invEta;
        if (wi.z * wo->z >= 0) return V(0, 0, 0);
        float factor = cosThetaT < 0 ? invEta : etaM;
        weight = mul(weight, scale(V(mt->reflectance[0], mt->reflectance[1], mt->reflectance[2]), factor * factor));
        float sqrtDenom = dot(wi, m) + *etaOut * dot(*wo, m);
        dwh_dwo = (*etaOut * *etaOut * dot(*wo, m)) / (sqrtDenom * sqrtDenom);
    }
    if (d.visible) weight = scale(weight, mf_smith_g1_2(d.distr, d.au, d.av, *wo, m));          /* roughdielectric.cpp:607-612 */
    else weight = scale(weight, fabsf(mf_eval2(d.distr, d.au, d.av, m) * (mf_smith_g1_2(d.distr, d.au, d.av, wi, m) * mf_smith_g1_2(d.distr, d.au, d.av, *wo, m)) * dot(wi, m) / (microfacetPDF * wi.z)));
    *pdf *= fabsf(dwh_dwo);
    return weight;
}
/* ---- diffuse transmitter: src/bsdfs/difftrans.cpp:78-120; reflectance = transmittance */
static v3 dt_eval(const orc_material *m, v3 wi, v3 wo) { if (wi.z * wo.z >= 0) return V(0, 0, 0); return scale(V(m->reflectance[0], m->reflectance[1], m->reflectance[2]), INV_PI * fabsf(wo.z)); }
static float dt_pdf(v3 wi, v3 wo) { if (wi.z * wo.z >= 0) return 0.0f; return fabsf(wo.z) * INV_PI; }
static v3 dt_sample(const orc_material *m, v3 wi, float sx, float sy, v3 *wo, float *pdf, float *eta) {
    *wo = cos_hemisphere(sx, sy); if (wi.z > 0) wo->z *= -1;
    *eta = 1.0f; *pdf = fabsf(wo->z) * INV_PI;
    return V(m->reflectance[0], m->reflectance[1], m->reflectance[2]);
}

/* ---- rough plastic: src/bsdfs/roughplastic.cpp:333-500; RoughTransmittance::eval with eta and alpha fixed (src/bsdfs/rtrans.h:183-193, :232) over
 * evalCubicInterp1D (src/libcore/spline.cpp:23-60).  Fields: alpha, distr, eta[0], specular, reflectance = diffuseReflectance, flag bit2 nonlinear,
 * k[0] = internal diffuse transmittance (m_internalRoughTransmittance->evalDiffuse(alpha), roughplastic.cpp:372), k[1] / k[2] = offset / length of
 * the external transmittance slice in the scene's material tables (setEta(eta) + setAlpha(alpha), roughplastic.cpp:291-299). */
static float cubic_interp_1d(float x, const float *values, size_t size, float min, float max) {
    if (!(x >= min && x <= max)) return 0.0f;
    float t = ((x - min) * (float) (size - 1)) / (max - min);
    size_t k = (size_t) t; if (k > size - 2) k = size - 2;
    float f0 = values[k], f1 = values[k + 1], d0, d1;
    if (k > 0) d0 = 0.5f * (values[k + 1] - values[k - 1]); else d0 = values[k + 1] - values[k];
    if (k + 2 < size) d1 = 0.5f * (values[k + 2] - values[k]); else d1 = values[k + 1] - values[k];
    t = t - (float) k;
    float t2 = t * t, t3 = t2 * t;
    return (2 * t3 - 3 * t2 + 1) * f0 + (-2 * t3 + 3 * t2) * f1 + (t3 - 2 * t2 + t) * d0 + (t3 - t2) * d1;
}
static float rp_transmittance(const orc_material *m, float cosTheta) {
    if (!(cosTheta >= 0)) return 0.0f;
    float warped = powf(fabsf(cosTheta), 0.25f);
    float result = cubic_interp_1d(warped, ((const mat_t *) m)->table, (size_t) m->k[2], 0.0f, 1.0f);
    return minf(1.0f, maxf(0.0f, result));
}
static float rp_prob_specular(const orc_material *m, float cosThetaI) {
    float probSpecular = 1 - rp_transmittance(m, cosThetaI), w = plastic_spec_weight(m);
    return (probSpecular * w) / (probSpecular * w + (1 - probSpecular) * (1 - w));
}
static v3 rp_eval(const orc_material *m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return V(0, 0, 0);
    const float eta = m->eta[0], alpha = maxf(m->alpha, 1e-4f), invEta2 = 1.0f / (eta * eta);
    v3 H = normalize(add(wo, wi));
    float D, G, ct, F = fresnel_dielectric_ext(dot(wi, H), &ct, eta);
    if (m->distr == 2) { D = mf_eval2(2, alpha, alpha, H); G = mf_smith_g1_2(2, alpha, alpha, wi, H) * mf_smith_g1_2(2, alpha, alpha, wo, H); }   /* Phong: roughness -> exponent (microfacet.h:98-110) */
    else { D = mf_eval(m->distr, alpha, H); G = mf_smith_g1(m->distr, alpha, wi, H) * mf_smith_g1(m->distr, alpha, wo, H); }
    float value = F * D * G / (4.0f * wi.z);
    v3 result = scale(V(m->specular[0], m->specular[1], m->specular[2]), value);
    v3 diff = V(m->reflectance[0], m->reflectance[1], m->reflectance[2]);
    float T12 = rp_transmittance(m, wi.z), T21 = rp_transmittance(m, wo.z), Fdr = 1 - m->k[0];
    if (m->flags & BSDF_FLAG_NONLINEAR) diff = V(diff.x / (1.0f - diff.x * Fdr), diff.y / (1.0f - diff.y * Fdr), diff.z / (1.0f - diff.z * Fdr));
    else { float r = 1.0f / (1 - Fdr); diff = scale(diff, r); }
    return add(result, scale(diff, INV_PI * wo.z * T12 * T21 * invEta2));
}
static float rp_pdf(const orc_material *m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return 0.0f;
    const float alpha = maxf(m->alpha, 1e-4f);
    v3 H = normalize(add(wo, wi));
    float probSpecular = rp_prob_specular(m, wi.z), probDiffuse = 1 - probSpecular;
    float dwh_dwo = 1.0f / (4.0f * dot(wo, H));
    const mfd_t d = {m->distr, alpha, alpha, (m->flags & 2u) != 0 && m->distr != 2};        /* distr.pdf(wi, H): visible normals or all normals (roughplastic.cpp:432) */
    float prob = mfd_pdf(&d, wi, H);
    float result = prob * dwh_dwo * probSpecular;
    result += probDiffuse * (INV_PI * wo.z);
    return result;
}
static v3 rp_sample(const orc_material *mt, v3 wi, float sx, float sy, v3 *wo, float *pdf, float *etaOut) {
    if (wi.z <= 0) return V(0, 0, 0);
    const float alpha = maxf(mt->alpha, 1e-4f);
    float probSpecular = rp_prob_specular(mt, wi.z); int choseSpecular = 1;
    if (sy < probSpecular) sy /= probSpecular; else { sy = (sy - probSpecular) / (1 - probSpecular); choseSpecular = 0; }
    if (choseSpecular) {
        const mfd_t d = {mt->distr, alpha, alpha, (mt->flags & 2u) != 0 && mt->distr != 2}; float mpdf;
        v3 m = mfd_sample(&d, wi, sx, sy, &mpdf);                             /* distr.sample(wi, sample) (roughplastic.cpp:483) */
        float c = 2 * dot(wi, m); *wo = sub(scale(m, c), wi);
        if (wo->z <= 0) return V(0, 0, 0);
    } else *wo = cos_hemisphere(sx, sy);
    *etaOut = 1.0f;
    *pdf = rp_pdf(mt, wi, *wo);
    if (*pdf == 0) return V(0, 0, 0);
    float r = 1.0f / *pdf; return scale(rp_eval(mt, wi, *wo), r);       /* Spectrum / Float */
}

/* ---- rough diffuse (Oren-Nayar): src/bsdfs/roughdiffuse.cpp:131-261.  alpha = m_alpha (constant), distr = 1: useFastApprox; m_alpha->eval(its).average() of the
 * constant texture = ((0 + a) + a + a) * (1 / 3) (include/mitsuba/core/spectrum.h:481-486); Frame::sinTheta / cosPhi / sinPhi: include/mitsuba/core/frame.h:107-154;
 * math::clamp = min(max, max(min, v)) (math.h:51-53); sample() returns eval / pdf = eval * (1 / pdf) (spectrum.h:415-425) */
static float frame_sin_theta(v3 v) { float t = 1.0f - v.z * v.z; if (t <= 0.0f) return 0.0f; return sqrtf(t); }
static v3 roughdiffuse_eval(const orc_material *m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return V(0, 0, 0);
    const float conversionFactor = 1 / sqrtf(2.0f);
    float avg = 0.0f; avg += m->alpha; avg += m->alpha; avg += m->alpha; avg = avg * (1.0f / 3);
    float sigma = avg * conversionFactor, sigma2 = sigma * sigma;
    float sinThetaI = frame_sin_theta(wi), sinThetaO = frame_sin_theta(wo), cosPhiDiff = 0;
    if (sinThetaI > EPSILON && sinThetaO > EPSILON) {
        float sinPhiI = minf(1.0f, maxf(-1.0f, wi.y / sinThetaI)), cosPhiI = minf(1.0f, maxf(-1.0f, wi.x / sinThetaI));
        float sinPhiO = minf(1.0f, maxf(-1.0f, wo.y / sinThetaO)), cosPhiO = minf(1.0f, maxf(-1.0f, wo.x / sinThetaO));
        cosPhiDiff = cosPhiI * cosPhiO + sinPhiI * sinPhiO;
    }
    v3 rho = V(m->reflectance[0], m->reflectance[1], m->reflectance[2]);
    if (m->distr == 1u) {
        float A = 1.0f - 0.5f * sigma2 / (sigma2 + 0.33f), B = 0.45f * sigma2 / (sigma2 + 0.09f), sinAlpha, tanBeta;
        if (wi.z > wo.z) { sinAlpha = sinThetaO; tanBeta = sinThetaI / wi.z; } else { sinAlpha = sinThetaI; tanBeta = sinThetaO / wo.z; }
        return scale(rho, INV_PI * wo.z * (A + B * maxf(cosPhiDiff, 0.0f) * sinAlpha * tanBeta));
    }
    float thetaI = acosf(minf(1.0f, maxf(-1.0f, wi.z))), thetaO = acosf(minf(1.0f, maxf(-1.0f, wo.z)));
    float alpha = maxf(thetaI, thetaO), beta = minf(thetaI, thetaO), sinAlpha, sinBeta, tanBeta;
    if (wi.z > wo.z) { sinAlpha = sinThetaO; sinBeta = sinThetaI; tanBeta = sinThetaI / wi.z; } else { sinAlpha = sinThetaI; sinBeta = sinThetaO; tanBeta = sinThetaO / wo.z; }
    float tmp = sigma2 / (sigma2 + 0.09f), tmp2 = (4 * INV_PI * INV_PI) * alpha * beta, tmp3 = 2 * beta * INV_PI;
    float C1 = 1.0f - 0.5f * sigma2 / (sigma2 + 0.33f), C2 = 0.45f * tmp, C3 = 0.125f * tmp * tmp2 * tmp2, C4 = 0.17f * sigma2 / (sigma2 + 0.13f);
    if (cosPhiDiff > 0) C2 *= sinAlpha; else C2 *= sinAlpha - tmp3 * tmp3 * tmp3;
    float tanHalf = (sinAlpha + sinBeta) / (sqrtf(maxf(0.0f, 1.0f - sinAlpha * sinAlpha)) + sqrtf(maxf(0.0f, 1.0f - sinBeta * sinBeta)));
    v3 snglScat = scale(rho, C1 + cosPhiDiff * C2 * tanBeta + (1.0f - fabsf(cosPhiDiff)) * C3 * tanHalf);
    v3 dblScat = scale(V(rho.x * rho.x, rho.y * rho.y, rho.z * rho.z), C4 * (1.0f - cosPhiDiff * tmp3 * tmp3));
    return scale(add(snglScat, dblScat), INV_PI * wo.z);
}
static float roughdiffuse_pdf(v3 wi, v3 wo) { if (wi.z <= 0 || wo.z <= 0) return 0.0f; return INV_PI * wo.z; }
static v3 roughdiffuse_sample(const orc_material *m, v3 wi, float sx, float sy, v3 *wo, float *pdf, float *eta) {
    if (wi.z <= 0) return V(0, 0, 0);
    *wo = cos_hemisphere(sx, sy); *eta = 1.0f; *pdf = INV_PI * wo->z;
    v3 f = roughdiffuse_eval(m, wi, *wo); float recip = 1.0f / *pdf;
    return V(f.x * recip, f.y * recip, f.z * recip);
}

/* ---- modified Phong: src/bsdfs/phong.cpp:130-256.  reflectance = diffuseReflectance, specular = specularReflectance, alpha = exponent (a constant texture:
 * eval(its).average() = ((0 + e) + e + e) * (1 / 3)), k[0] = m_specularSamplingWeight (configure(), :104-108).  M_PI is the float constant (constants.h:63, 80);
 * Frame(R).toWorld (frame.h:83-85) over coordinateSystem (util.cpp:594-603) */
static float phong_exponent(const orc_material *m) { float e = 0.0f; e += m->alpha; e += m->alpha; e += m->alpha; return e * (1.0f / 3); }
static v3 phong_eval(const orc_material *m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return V(0, 0, 0);
    v3 result = V(0, 0, 0);
    float alpha = dot(wo, V(-wi.x, -wi.y, wi.z)), exponent = phong_exponent(m);
    if (alpha > 0.0f) result = scale(V(m->specular[0], m->specular[1], m->specular[2]), (exponent + 2) * INV_TWOPI * powf(alpha, exponent));
    result = add(result, scale(V(m->reflectance[0], m->reflectance[1], m->reflectance[2]), INV_PI));
    return scale(result, wo.z);
}
static float phong_pdf(const orc_material *m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return 0.0f;
    float diffuseProb = INV_PI * wo.z, specProb = 0.0f;
    float alpha = dot(wo, V(-wi.x, -wi.y, wi.z)), exponent = phong_exponent(m);
    if (alpha > 0) specProb = powf(alpha, exponent) * (exponent + 1.0f) / (2.0f * M_PI_F);
    return m->k[0] * specProb + (1 - m->k[0]) * diffuseProb;
}
static v3 phong_sample(const orc_material *m, v3 wi, float sx, float sy, v3 *wo, float *pdf, float *eta) {
    float w = m->k[0]; int choseSpecular = 1;
    if (sx <= w) sx /= w; else { sx = (sx - w) / (1 - w); choseSpecular = 0; }
    if (choseSpecular) {
        v3 R = V(-wi.x, -wi.y, wi.z), fs, ft; float exponent = phong_exponent(m);
        float sinAlpha = sqrtf(1 - powf(sy, 2 / (exponent + 1))), cosAlpha = powf(sy, 1 / (exponent + 1)), phi = (2.0f * M_PI_F) * sx;
        v3 local = V(sinAlpha * cosf(phi), sinAlpha * sinf(phi), cosAlpha);
        coordinate_system(R, &fs, &ft);
        *wo = add(add(scale(fs, local.x), scale(ft, local.y)), scale(R, local.z));
        if (wo->z <= 0) return V(0, 0, 0);
    } else *wo = cos_hemisphere(sx, sy);
    *eta = 1.0f; *pdf = phong_pdf(m, wi, *wo);
    if (*pdf == 0) return V(0, 0, 0);
    v3 f = phong_eval(m, wi, *wo); float recip = 1.0f / *pdf;
    return V(f.x * recip, f.y * recip, f.z * recip);
}

/* ---- Ward (ward / ward-duer / balanced): src/bsdfs/ward.cpp:178-338.  reflectance = diffuseReflectance, specular = specularReflectance, alpha = alphaU, k[1] = alphaV,
 * distr = variant, k[0] = m_specularSamplingWeight (:160-164).  std::pow(Float, int) is the binary64 pow (C++11 promotion), and the factors around it are evaluated in
 * binary64 as the usual arithmetic conversions make them; math::fastexp / fastlog = exp / log in binary64 (math.h:185-195); sphericalDirection: util.cpp:581-592 */
static float avg3(float a) { float e = 0.0f; e += a; e += a; e += a; return e * (1.0f / 3); }
static float ward_exp(v3 H, float alphaU, float alphaV) {
    float factor2 = H.x / alphaU, factor3 = H.y / alphaV;
    return fastexpf_(-(factor2 * factor2 + factor3 * factor3) / (H.z * H.z));
}
static v3 ward_eval(const orc_material *m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return V(0, 0, 0);
    v3 result = V(0, 0, 0), H = add(wi, wo); float alphaU = avg3(m->alpha), alphaV = avg3(m->k[1]), factor1;
    if (m->distr == 0u) factor1 = 1.0f / (4.0f * M_PI_F * alphaU * alphaV * sqrtf(wi.z * wo.z));
    else if (m->distr == 1u) factor1 = 1.0f / (4.0f * M_PI_F * alphaU * alphaV * wi.z * wo.z);
    else factor1 = (float) (dot(H, H) / (M_PI_F * alphaU * alphaV * pow((double) H.z, 4.0)));
    float specRef = factor1 * ward_exp(H, alphaU, alphaV);
    if (specRef > 1e-10f) result = scale(V(m->specular[0], m->specular[1], m->specular[2]), specRef);
    result = add(result, scale(V(m->reflectance[0], m->reflectance[1], m->reflectance[2]), INV_PI));
    return scale(result, wo.z);
}
static float ward_pdf(const orc_material *m, v3 wi, v3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return 0.0f;
    float alphaU = avg3(m->alpha), alphaV = avg3(m->k[1]); v3 H = normalize(add(wi, wo));
    float factor1 = (float) (1.0f / (4.0f * M_PI_F * alphaU * alphaV * dot(H, wi) * pow((double) H.z, 3.0)));
    float specProb = factor1 * ward_exp(H, alphaU, alphaV), diffuseProb = INV_PI * wo.z;
    return m->k[0] * specProb + (1 - m->k[0]) * diffuseProb;
}
static v3 ward_sample(const orc_material *m, v3 wi, float sx, float sy, v3 *wo, float *pdf, float *eta) {
    float w = m->k[0]; int choseSpecular = 1;
    if (sx <= w) sx /= w; else { sx = (sx - w) / (1 - w); choseSpecular = 0; }
    if (choseSpecular) {
        float alphaU = avg3(m->alpha), alphaV = avg3(m->k[1]);
        float phiH = atanf(alphaV / alphaU * tanf(2.0f * M_PI_F * sy));
        if (sy > 0.5f) phiH += M_PI_F;
        float cosPhiH = cosf(phiH), sinPhiH = sqrtf(maxf(0.0f, 1.0f - cosPhiH * cosPhiH));
        float thetaH = atanf(sqrtf(maxf(0.0f, -fastlogf_(sx) / ((cosPhiH * cosPhiH) / (alphaU * alphaU) + (sinPhiH * sinPhiH) / (alphaV * alphaV)))));
        float sinTheta, cosTheta, sinPhi, cosPhi; sincosf(thetaH, &sinTheta, &cosTheta); sincosf(phiH, &sinPhi, &cosPhi);
        v3 H = V(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
        *wo = sub(scale(H, 2.0f * dot(wi, H)), wi);
        if (wo->z <= 0.0f) return V(0, 0, 0);
    } else *wo = cos_hemisphere(sx, sy);
    *eta = 1.0f; *pdf = ward_pdf(m, wi, *wo);
    if (*pdf == 0) return V(0, 0, 0);
    v3 f = ward_eval(m, wi, *wo); float recip = 1.0f / *pdf;
    return V(f.x * recip, f.y * recip, f.z * recip);
}

static v3 bsdf_eval(const orc_material *m, v3 wi, v3 wo) {
    if ((m->flags & BSDF_FLAG_TWOSIDED) && wi.z < 0) { wi.z = -wi.z; wo.z = -wo.z; }
    switch (m->type) {
        case BSDF_ROUGHCONDUCTOR: return rc_eval(m, wi, wo);
        case BSDF_CONDUCTOR: case BSDF_DIELECTRIC: case BSDF_THINDIELECTRIC: case BSDF_NULL: return V(0, 0, 0);
        case BSDF_PLASTIC: return plastic_eval(m, wi, wo);
        case BSDF_ROUGHDIELECTRIC: return rd_eval(m, wi, wo);
        case BSDF_DIFFTRANS: return dt_eval(m, wi, wo);
        case BSDF_ROUGHPLASTIC: return rp_eval(m, wi, wo);
        case BSDF_ROUGHDIFFUSE: return roughdiffuse_eval(m, wi, wo);
        case BSDF_PHONG: return phong_eval(m, wi, wo);
        case BSDF_WARD: return ward_eval(m, wi, wo);
        default: return diffuse_eval(m, wi, wo);
    }
}
static float bsdf_pdf(const orc_material *m, v3 wi, v3 wo) {
    if ((m->flags & BSDF_FLAG_TWOSIDED) && wi.z < 0) { wi.z = -wi.z; wo.z = -wo.z; }
    switch (m->type) {
        case BSDF_ROUGHCONDUCTOR: return rc_pdf(m, wi, wo);
        case BSDF_CONDUCTOR: case BSDF_DIELECTRIC: case BSDF_THINDIELECTRIC: case BSDF_NULL: return 0.0f;
        case BSDF_PLASTIC: return plastic_pdf(m, wi, wo);
        case BSDF_ROUGHDIELECTRIC: return rd_pdf(m, wi, wo);
        case BSDF_DIFFTRANS: return dt_pdf(wi, wo);
        case BSDF_ROUGHPLASTIC: return rp_pdf(m, wi, wo);
        case BSDF_ROUGHDIFFUSE: return roughdiffuse_pdf(wi, wo);
        case BSDF_PHONG: return phong_pdf(m, wi, wo);
        case BSDF_WARD: return ward_pdf(m, wi, wo);
        default: return diffuse_pdf(wi, wo);
    }
}
/* *delta = the sampled component is a Dirac delta (bRec.sampledType & BSDF::EDelta, path.cpp:259-260) */
/* sp: the path's sampler, used by BSDFs with EUsesSampler (may be NULL for the unit entry point: then `extra_unit` stands in) */
static float g_extra_unit = 0.5f;
static v3 bsdf_sample(const orc_material *m, v3 wi, float u, float v, v3 *wo, float *pdf, float *eta, int *delta, sampler_t *sp) {
    int flipped = 0; *delta = 0;
    if ((m->flags & BSDF_FLAG_TWOSIDED) && wi.z < 0) { wi.z = -wi.z; flipped = 1; }
    v3 w;
    switch (m->type) {
        case BSDF_ROUGHCONDUCTOR: w = rc_sample(m, wi, u, v, wo, pdf, eta); break;
        case BSDF_CONDUCTOR: w = conductor_sample(m, wi, wo, pdf, eta, delta); break;
        case BSDF_DIELECTRIC: w = dielectric_sample(m, wi, u, wo, pdf, eta, delta); break;
        case BSDF_PLASTIC: w = plastic_sample(m, wi, u, v, wo, pdf, eta, delta); break;
        case BSDF_ROUGHDIELECTRIC: w = rd_sample(m, wi, u, v, sp ? next1D(sp) : g_extra_unit, wo, pdf, eta); break;
        case BSDF_DIFFTRANS: w = dt_sample(m, wi, u, v, wo, pdf, eta); break;
        case BSDF_ROUGHPLASTIC: w = rp_sample(m, wi, u, v, wo, pdf, eta); break;
        case BSDF_ROUGHDIFFUSE: w = roughdiffuse_sample(m, wi, u, v, wo, pdf, eta); break;
        case BSDF_PHONG: w = phong_sample(m, wi, u, v, wo, pdf, eta); break;
        case BSDF_WARD: w = ward_sample(m, wi, u, v, wo, pdf, eta); break;
        case BSDF_THINDIELECTRIC: w = thindielectric_sample(m, wi, u, wo, pdf, eta, delta); break;
        case BSDF_NULL: *wo = neg(wi); *pdf = 1.0f; *eta = 1.0f; *delta = 2; w = V(1, 1, 1); break;      /* src/bsdfs/null.cpp:56-66: the index-matched boundary, sampledType = ENull */
        default: w = diffuse_sample(m, wi, u, v, wo, pdf, eta); break;
    }
    if (flipped && !is_zero(w) && *pdf != 0) wo->z = -wo->z;      /* twosided.cpp:176-180 */
    return w;
}
/* a hit's material with its textures evaluated and its wrappers resolved (below): mask -> bumpmap / normalmap -> mixturebsdf -> plain BSDFs */
typedef struct { mat_t inner; int masked; int pdfless; v3 opacity; float prob;
                 int bumped; v3 ps, pt, pn; const hit_t *its;           /* bumpmap / normalmap: the perturbed shading frame; the hit's own frame stays the query frame */
                 int coated; mat_t coatm; float coat_w;                              /* coating: the layer's record and its m_specularSamplingWeight; `inner` is the nested BSDF */
                 int blend;                                              /* blendbsdf: two children in mix[0..1], w[1] = the (textured) weight, w[0] = 1 - w[1]; selection by `sample.x < w[0]` */
                 int n_mix; mat_t mix[4]; float w[4], p[4], cdf[5]; } smat_t;  /* mixturebsdf: children, weights, normalised selection probabilities */
static smat_t resolve_material(const orc_scene *s, uint32_t material, const hit_t *its, int want_partials, v3 o, const v3 *rxd, const v3 *ryd);
static v3 sm_eval(const smat_t *sm, v3 wi, v3 wo);
static float sm_pdf(const smat_t *sm, v3 wi, v3 wo);
static v3 sm_sample(const smat_t *sm, v3 wi, float u, float v, v3 *wo, float *pdf, float *eta, int *delta, sampler_t *sp);
void orc_bsdf_sample(const orc_scene *s, uint32_t mi, const float *wi, float u, float v, float *o) {
    v3 wo = V(0, 0, 0); float pdf = 0, eta = 0; int delta; const smat_t sm = resolve_material(s, mi, NULL, 0, V(0, 0, 0), NULL, NULL); v3 w = sm_sample(&sm, V(wi[0], wi[1], wi[2]), u, v, &wo, &pdf, &eta, &delta, NULL);
    o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = pdf; o[4] = wo.x; o[5] = wo.y; o[6] = wo.z; o[7] = eta;
}
void orc_bsdf_eval(const orc_scene *s, uint32_t mi, const float *wi, const float *wo, float *o) {
    const smat_t sm = resolve_material(s, mi, NULL, 0, V(0, 0, 0), NULL, NULL);
    v3 e = sm_eval(&sm, V(wi[0], wi[1], wi[2]), V(wo[0], wo[1], wo[2]));
    o[0] = e.x; o[1] = e.y; o[2] = e.z; o[3] = sm_pdf(&sm, V(wi[0], wi[1], wi[2]), V(wo[0], wo[1], wo[2]));
}

/* ------------------------------------------------------------------------------------------------ environment emitter */
#ifndef INV_TWOPI
#define INV_TWOPI 0.15915494309189533577f
#endif
static inline v3 mat3(const float *m, v3 v) { return V(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z); }
/* include/mitsuba/render/mipmap.h:504-560 evalTexel, level 0, u: ERepeat, v: EClamp (envmap.cpp:181-183) */
static v3 env_texel(const orc_scene *s, int x, int y) {
    if (x < 0 || x >= s->env_w) { int r = x % s->env_w; x = r < 0 ? r + s->env_w : r; }
    if (y < 0) y = 0; else if (y >= s->env_h) y = s->env_h - 1;
    const float *p = s->env_rgb + ((size_t) y * s->env_w + x) * 3; return V(p[0], p[1], p[2]);
}
/* mipmap.h:576-597 evalBilinear(0, uv) */
static v3 env_bilinear(const orc_scene *s, float uvx, float uvy) {
    if (!isfinite(uvx) || !isfinite(uvy)) return V(0, 0, 0);
    float u = uvx * (float) s->env_w - 0.5f, v = uvy * (float) s->env_h - 0.5f;
    int xPos = (int) floorf(u), yPos = (int) floorf(v);
    float dx1 = u - (float) xPos, dx2 = 1.0f - dx1, dy1 = v - (float) yPos, dy2 = 1.0f - dy1;
    v3 r = scale(scale(env_texel(s, xPos, yPos), dx2), dy2);
    r = add(r, scale(scale(env_texel(s, xPos, yPos + 1), dx2), dy1));
    r = add(r, scale(scale(env_texel(s, xPos + 1, yPos), dx1), dy2));
    r = add(r, scale(scale(env_texel(s, xPos + 1, yPos + 1), dx1), dy1));
    return r;
}
/* envmap.cpp:384-416 evalEnvironment without ray differentials (level-0 bilinear lookup) */
static v3 env_eval_map(const orc_scene *s, v3 d);
static v3 env_eval(const orc_scene *s, v3 d) {                     /* ConstantBackgroundEmitter::evalEnvironment (constant.cpp:244-246) */
    if (s->env_constant) { const float *r = s->emitters[s->env_index].radiance; return V(r[0], r[1], r[2]); }
    return env_eval_map(s, d);
}
static v3 env_eval_map(const orc_scene *s, v3 d) {
    v3 v = mat3(s->env_to_local, d);
    float uvx = atan2f(v.x, -v.z) * INV_TWOPI, uvy = acosf(minf(1.0f, maxf(-1.0f, v.y))) * INV_PI;
    return scale(env_bilinear(s, uvx, uvy), s->env_scale);
}
/* envmap.cpp:664-669 sampleReuse over a float CDF */
static uint32_t env_sample_reuse(const float *cdf, uint32_t size, float *sample) {
    uint32_t lo = 0, hi = size + 1;
    while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (cdf[mid] < *sample) lo = mid + 1; else hi = mid; }
    int64_t e = (int64_t) lo - 1; if (e < 0) e = 0;
    uint32_t index = (uint32_t) e; if (index > size - 1) index = size - 1;
    *sample = (*sample - cdf[index]) / (cdf[index + 1] - cdf[index]);
    return index;
}
static inline float interval_to_tent(float sample) {                    /* src/libcore/warp.cpp:142-155 */
    float sign;
    if (sample < 0.5f) { sign = 1; sample *= 2; } else { sign = -1; sample = 2 * (sample - 0.5f); }
    return sign * (1 - sqrtf(sample));
}
static void env_bilinear_pair(const orc_scene *s, float px, float py, v3 *value1, v3 *value2, int *yPosOut) {
    int xPos = (int) floorf(px), yPos = (int) floorf(py);
    float dx1 = px - (float) xPos, dx2 = 1.0f - dx1, dy1 = py - (float) yPos, dy2 = 1.0f - dy1;
    *value1 = add(scale(scale(env_texel(s, xPos, yPos), dx2), dy2), scale(scale(env_texel(s, xPos + 1, yPos), dx1), dy2));
    *value2 = add(scale(scale(env_texel(s, xPos, yPos + 1), dx2), dy1), scale(scale(env_texel(s, xPos + 1, yPos + 1), dx1), dy1));
    *yPosOut = yPos;
}
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* envmap.cpp:571-608 internalSampleDirection */
static void env_sample_direction(const orc_scene *s, float sx, float sy, v3 *d, v3 *value, float *pdf) {
    uint32_t row = env_sample_reuse(s->env_cdf_rows, (uint32_t) s->env_h, &sy);
    uint32_t col = env_sample_reuse(s->env_cdf_cols + (size_t) row * (s->env_w + 1), (uint32_t) s->env_w, &sx);
    float px = (float) col + interval_to_tent(sx), py = (float) row + interval_to_tent(sy);
    v3 v1, v2; int yPos; env_bilinear_pair(s, px, py, &v1, &v2, &yPos);
    *value = scale(add(v1, v2), s->env_scale);
    *pdf = (luminance(v1) * s->env_row_weights[clampi(yPos, 0, s->env_h - 1)] + luminance(v2) * s->env_row_weights[clampi(yPos + 1, 0, s->env_h - 1)]) * s->env_normalization;
    float ph = s->env_pixel_w * (px + 0.5f), th = s->env_pixel_h * (py + 0.5f);
    float sinPhi = sinf(ph), cosPhi = cosf(ph), sinTheta = sinf(th), cosTheta = cosf(th);
    *d = V(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta);
    *pdf /= maxf(fabsf(sinTheta), EPSILON);
}
/* envmap.cpp:611-638 internalPdfDirection */
static float env_pdf_direction(const orc_scene *s, v3 d) {
    float uvx = atan2f(d.x, -d.z) * INV_TWOPI, uvy = acosf(minf(1.0f, maxf(-1.0f, d.y))) * INV_PI;
    if (!isfinite(uvx) || !isfinite(uvy)) return 0.0f;
    float u = uvx * (float) s->env_w - 0.5f, v = uvy * (float) s->env_h - 0.5f;
    v3 v1, v2; int yPos; env_bilinear_pair(s, u, v, &v1, &v2, &yPos);
    float sinTheta = sqrtf(maxf(1 - d.y * d.y, 0.0f));
    return (luminance(v1) * s->env_row_weights[clampi(yPos, 0, s->env_h - 1)] + luminance(v2) * s->env_row_weights[clampi(yPos + 1, 0, s->env_h - 1)])
           * s->env_normalization / maxf(fabsf(sinTheta), EPSILON);
}
/* include/mitsuba/core/bsphere.h:88-95 + src/libcore/util.cpp:449-487 solveQuadratic */
static int bsphere_intersect(v3 center, float radius, v3 ro, v3 rd, float *nearT, float *farT) {
    v3 o = sub(ro, center);
    float A = dot(rd, rd), B = 2 * dot(o, rd), C = dot(o, o) - radius * radius;
    if (A == 0) { if (B != 0) { *nearT = *farT = -C / B; return 1; } return 0; }
    float discrim = B * B - 4.0f * A * C;
    if (discrim < 0) return 0;
    float temp, sqrtDiscrim = sqrtf(discrim);
    if (B < 0) temp = -0.5f * (B - sqrtDiscrim); else temp = -0.5f * (B + sqrtDiscrim);
    float x0 = temp / A, x1 = C / temp;
    if (x0 > x1) { float t = x0; x0 = x1; x1 = t; }
    *nearT = x0; *farT = x1; return 1;
}

/* ------------------------------------------------------------------------------------------------ emitters */
/* include/mitsuba/core/pmf.h:124-137 DiscreteDistribution::sample (lower_bound over cdf[0..n]) */
static uint32_t cdf_sample(const float *cdf, uint32_t n, float x) {
    uint32_t lo = 0, hi = n + 1;                       /* first element of cdf[0..n] that is >= x */
    while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (cdf[mid] < x) lo = mid + 1; else hi = mid; }
    int64_t e = (int64_t) lo - 1; if (e < 0) e = 0;
    uint32_t index = (uint32_t) e; if (index > n - 1) index = n - 1;
    while (cdf[index + 1] - cdf[index] == 0 && index < n) ++index;
    return index;
}
typedef struct { v3 ref, p, n, d; float dist, pdf; float em_pdf; /* probability of the chosen emitter (attenuated mode) */ int32_t emitter; int delta; /* !isOnSurface: point / spot / directional */ } direct_t;

/* src/emitters/area.cpp:106-111 AreaLight::eval */
static v3 emitter_eval(const orc_scene *s, int32_t e, v3 ns, v3 d) {
    if (dot(ns, d) <= 0) return V(0, 0, 0);
    const float *r = s->emitters[e].radiance; return V(r[0], r[1], r[2]);
}
/* Scene::sampleEmitterDirect (src/librender/scene.cpp:860-884) -> AreaLight::sampleDirect (src/emitters/area.cpp:160-176)
 * -> Shape::sampleDirect (src/librender/shape.cpp:102-115) -> TriMesh::samplePosition (src/librender/trimesh.cpp:413-425)
 * -> Triangle::sample (src/libcore/triangle.cpp:24-59) */
static v3 sample_emitter_direct(const orc_scene *s, v3 ref, v3 refN, float sx, float sy, direct_t *dr, int test_visibility, uint64_t *shadow_rays) {
    uint32_t ne = s->d.n_emitters;
    uint32_t ei = cdf_sample(s->emitter_cdf, ne, sx);
    float emPdf = s->emitter_cdf[ei + 1] - s->emitter_cdf[ei];
    sx = (sx - s->emitter_cdf[ei]) / (s->emitter_cdf[ei + 1] - s->emitter_cdf[ei]);
    const orc_emitter *em = &s->emitters[ei];
    dr->delta = 0;
    if (em->type >= 2) {
        v3 value; dr->pdf = 0.0f;
        if (em->type == 2) {
            /* ConstantBackgroundEmitter::sampleDirect (constant.cpp:175-217) */
            v3 d; float pdf, nearT, farT;
            if (!is_zero(refN)) {
                v3 l = cos_hemisphere(sx, sy); pdf = INV_PI * l.z;
                v3 fs, ft; coordinate_system(refN, &fs, &ft);
                d = add(add(scale(fs, l.x), scale(ft, l.y)), scale(refN, l.z));
            } else { d = uniform_sphere(sx, sy); pdf = INV_FOURPI; }
            if (!bsphere_intersect(s->env_bs_center, s->env_bs_radius, ref, d, &nearT, &farT)) return V(0, 0, 0);
            if (!(nearT < 0 && farT > 0)) return V(0, 0, 0);
            dr->p = add(ref, scale(d, farT)); dr->n = normalize(sub(s->env_bs_center, dr->p)); dr->d = d; dr->dist = farT; dr->pdf = pdf;
            if (!is_zero(refN) && dot(d, refN) <= 0) value = V(0, 0, 0);       /* NB pdf stays non-zero: the shadow ray is still traced */
            else { float r = 1.0f / pdf; value = V(em->radiance[0] * r, em->radiance[1] * r, em->radiance[2] * r); }
        } else if (em->type == 3 || em->type == 4) {
            /* PointEmitter::sampleDirect (point.cpp:133-149), SpotEmitter::sampleDirect (spot.cpp:187-203) + falloffCurve (:108-128) */
            dr->delta = 1;
            dr->p = V(em->to_world[3], em->to_world[7], em->to_world[11]);
            dr->d = sub(dr->p, ref); dr->dist = length3(dr->d);
            float invDist = 1.0f / dr->dist; dr->d = scale(dr->d, invDist); dr->n = V(0, 0, 0); dr->pdf = 1.0f;
            v3 I = V(em->radiance[0], em->radiance[1], em->radiance[2]);
            if (em->type == 4) {
                v3 local = mat3(&s->spot_to_local[ei * 9], neg(dr->d)); float cosTheta = local.z, f;
                if (cosTheta <= s->spot_cos_cutoff[ei]) f = 0.0f;
                else if (cosTheta >= s->spot_cos_beam[ei]) f = 1.0f;
                else f = (s->spot_cutoff[ei] - acosf(cosTheta)) * s->spot_inv_transition[ei];
                I = scale(I, f);
            }
            value = scale(I, invDist * invDist);
        } else if (em->type == 7) {
            /* CollimatedBeamEmitter::sampleDirect (collimated.cpp:129-133): direct sampling always fails for a response function on a 0-D space */
            dr->pdf = 0.0f; return V(0, 0, 0);
        } else {
            /* DirectionalEmitter::sampleDirect (directional.cpp:159-180) */
            dr->delta = 1;
            v3 d = V(em->to_world[2], em->to_world[6], em->to_world[10]);
            v3 diskCenter = sub(s->dir_bs_center, scale(d, s->dir_bs_radius));
            float distance = dot(sub(ref, diskCenter), d);
            if (distance < 0) return V(0, 0, 0);
            dr->p = sub(ref, scale(d, distance)); dr->d = neg(d); dr->n = d; dr->dist = distance; dr->pdf = 1.0f;
            value = V(em->radiance[0], em->radiance[1], em->radiance[2]);
        }
        if (dr->pdf != 0) {                                                     /* scene.cpp:870-883 */
            if (test_visibility == 1) {
                if (shadow_rays) ++*shadow_rays;
                if (ray_occluded(s, ref, dr->d, EPSILON, dr->dist * (1 - SHADOW_EPSILON))) return V(0, 0, 0);
            }
            dr->emitter = (int32_t) ei; dr->pdf *= emPdf;
            dr->em_pdf = emPdf; if (test_visibility != 2) { float r = 1.0f / emPdf; value = scale(value, r); }
            return value;
        }
        return V(0, 0, 0);
    }
    if (em->type == 1) {
        /* EnvironmentMap::sampleDirect (src/emitters/envmap.cpp:520-547) */
        v3 value, dl; float pdf, nearT, farT;
        env_sample_direction(s, sx, sy, &dl, &value, &pdf);
        v3 dw = mat3(s->env_to_world, dl);
        if (is_zero(value) || pdf == 0 || !bsphere_intersect(s->env_bs_center, s->env_bs_radius, ref, dw, &nearT, &farT) || nearT >= 0 || farT <= 0) {
            dr->pdf = 0.0f; return V(0, 0, 0);
        }
        dr->pdf = pdf; dr->p = add(ref, scale(dw, farT)); dr->n = normalize(sub(s->env_bs_center, dr->p)); dr->dist = farT; dr->d = dw;
        { float r = 1.0f / pdf; value = scale(value, r); }
        if (test_visibility == 1) {
            if (shadow_rays) ++*shadow_rays;
            if (ray_occluded(s, ref, dr->d, EPSILON, dr->dist * (1 - SHADOW_EPSILON))) return V(0, 0, 0);
        }
        dr->emitter = (int32_t) ei; dr->pdf *= emPdf;
        dr->em_pdf = emPdf; if (test_visibility != 2) { float r = 1.0f / emPdf; value = scale(value, r); }
        return value;
    }
    dr->ref = ref;
    if ((uint32_t) em->shape >= s->d.n_shapes) {       /* area light on an analytic shape: Shape::sampleDirect / Sphere::sampleDirect, no sample reuse */
        analytic_sample_direct(&s->analytic[(uint32_t) em->shape - s->d.n_shapes], ref, sx, sy, &dr->p, &dr->n, &dr->d, &dr->dist, &dr->pdf);
    } else {
    const orc_shape *sh = &s->shapes[em->shape];
    uint32_t ti = cdf_sample(s->area_cdf[ei], sh->tri_count, sy);
    sy = (sy - s->area_cdf[ei][ti]) / (s->area_cdf[ei][ti + 1] - s->area_cdf[ei][ti]);
    uint32_t prim = sh->first_tri + ti;
    uint32_t i0 = s->idx[prim * 3], i1 = s->idx[prim * 3 + 1], i2 = s->idx[prim * 3 + 2];
    v3 p0 = vert(s, i0), p1 = vert(s, i1), p2 = vert(s, i2);
    float bx, by; uniform_triangle(sx, sy, &bx, &by);
    v3 sideA = sub(p1, p0), sideB = sub(p2, p0);
    dr->p = add(add(p0, scale(sideA, bx)), scale(sideB, by));
    if (s->nrm != NULL && !(sh->flags & 1u))
        dr->n = normalize(add(add(scale(vnrm(s, i0), 1.0f - bx - by), scale(vnrm(s, i1), bx)), scale(vnrm(s, i2), by)));
    else dr->n = normalize(cross(sideA, sideB));
    dr->pdf = s->inv_area[ei];
    dr->d = sub(dr->p, ref);
    float distSquared = dot(dr->d, dr->d);
    dr->dist = sqrtf(distSquared);
    { float r = 1.0f / dr->dist; dr->d = scale(dr->d, r); }
    float dp = fabsf(dot(dr->d, dr->n));
    dr->pdf *= dp != 0 ? (distSquared / dp) : 0.0f;
    }
    v3 value;
    if (dot(dr->d, refN) >= 0 && dot(dr->d, dr->n) < 0 && dr->pdf != 0) {
        float r = 1.0f / dr->pdf; value = V(em->radiance[0] * r, em->radiance[1] * r, em->radiance[2] * r);   /* Spectrum / Float */
    } else { dr->pdf = 0.0f; value = V(0, 0, 0); }
    if (dr->pdf != 0) {
        if (test_visibility == 1) {
            if (shadow_rays) ++*shadow_rays;
            if (ray_occluded(s, ref, dr->d, EPSILON, dr->dist * (1 - SHADOW_EPSILON))) return V(0, 0, 0);
        }
        dr->emitter = (int32_t) ei;
        dr->pdf *= emPdf;
        dr->em_pdf = emPdf; if (test_visibility != 2) { float r = 1.0f / emPdf; value = scale(value, r); }     /* Spectrum /= Float */
        return value;
    }
    return V(0, 0, 0);
}
/* Scene::pdfEmitterDirect (scene.cpp:981-984) -> AreaLight::pdfDirect (area.cpp:178-184) -> Shape::pdfDirect (shape.cpp:117-126);
 * pdfEmitterDiscrete (include/mitsuba/render/scene.h:848-850) */
static float pdf_emitter_direct(const orc_scene *s, const direct_t *dr, v3 refN) {
    float pdf;
    if (s->emitters[dr->emitter].type >= 3)                                    /* point / spot / directional pdfDirect with measure = EDiscrete (point.cpp:151-153) */
        return 1.0f * (s->emitters[dr->emitter].weight * s->emitter_norm);
    if (s->emitters[dr->emitter].type == 2) {                                  /* ConstantBackgroundEmitter::pdfDirect, ESolidAngle (constant.cpp:219-233) */
        float pdfSA = !is_zero(refN) ? INV_PI * maxf(0.0f, dot(dr->d, refN)) : INV_FOURPI;
        return pdfSA * (s->emitters[dr->emitter].weight * s->emitter_norm);
    }
    if (s->emitters[dr->emitter].type == 1)                                    /* EnvironmentMap::pdfDirect, measure = ESolidAngle (envmap.cpp:549-560) */
        return env_pdf_direction(s, mat3(s->env_to_local, dr->d)) * (s->emitters[dr->emitter].weight * s->emitter_norm);
    if (dot(dr->d, refN) >= 0 && dot(dr->d, dr->n) < 0) {
        int32_t shape = s->emitters[dr->emitter].shape;
        if ((uint32_t) shape >= s->d.n_shapes) pdf = analytic_pdf_direct(&s->analytic[(uint32_t) shape - s->d.n_shapes], dr->ref, dr->d, dr->n, dr->dist);
        else pdf = s->inv_area[dr->emitter] * (dr->dist * dr->dist) / fabsf(dot(dr->d, dr->n));
    } else pdf = 0.0f;
    return pdf * (s->emitters[dr->emitter].weight * s->emitter_norm);
}
void orc_sample_emitter_direct(const orc_scene *s, const float *rp, const float *rn, float u, float v, float *o) {
    direct_t dr; memset(&dr, 0, sizeof(dr));
    v3 refN = V(rn[0], rn[1], rn[2]);
    v3 val = sample_emitter_direct(s, V(rp[0], rp[1], rp[2]), refN, u, v, &dr, 1, NULL);
    o[0] = val.x; o[1] = val.y; o[2] = val.z; o[3] = dr.p.x; o[4] = dr.p.y; o[5] = dr.p.z; o[6] = dr.d.x; o[7] = dr.d.y; o[8] = dr.d.z;
    o[9] = dr.dist; o[10] = dr.pdf; o[11] = is_zero(val) ? 0.0f : pdf_emitter_direct(s, &dr, refN);
}

/* ---- bitmap textures: TMIPMap (include/mitsuba/render/mipmap.h): evalTexel :504-560, evalBox :562-566, evalBilinear :572-596, eval :625-705,
 * evalEWA :746-818; the pyramid itself is input data (the reference resamples with Bitmap::resample; fixtures come from oracle/_ref/harness mipmap) */
static inline int imod(int a, int b) { int r = a % b; return r < 0 ? r + b : r; }
static v3 mip_texel(const orc_scene *s, const orc_texture *t, int level, int x, int y) {
    const uint32_t *L = &s->tex_levels[(t->first_level + (uint32_t) level) * 3]; const int w = (int) L[0], h = (int) L[1];
    if (x < 0 || x >= w) switch (t->wrap_u) {
        case 1: x = imod(x, w); break;
        case 0: x = x < 0 ? 0 : w - 1; break;
        case 2: x = imod(x, 2 * w); if (x >= w) x = 2 * w - x - 1; break;
        case 3: return V(0, 0, 0);
        default: return V(1, 1, 1);
    }
    if (y < 0 || y >= h) switch (t->wrap_v) {
        case 1: y = imod(y, h); break;
        case 0: y = y < 0 ? 0 : h - 1; break;
        case 2: y = imod(y, 2 * h); if (y >= h) y = 2 * h - y - 1; break;
        case 3: return V(0, 0, 0);
        default: return V(1, 1, 1);
    }
    const float *p = &s->tex_texels[L[2] + ((size_t) y * w + x) * 3]; return V(p[0], p[1], p[2]);
}
static v3 mip_box(const orc_scene *s, const orc_texture *t, int level, float u, float v) {
    const uint32_t *L = &s->tex_levels[(t->first_level + (uint32_t) level) * 3];
    return mip_texel(s, t, level, (int) floorf(u * (float) (int) L[0]), (int) floorf(v * (float) (int) L[1]));
}
static v3 mip_bilinear(const orc_scene *s, const orc_texture *t, int level, float uvx, float uvy) {
    if (!isfinite(uvx) || !isfinite(uvy)) return V(0, 0, 0);
    if (level >= (int) t->n_levels) return mip_box(s, t, (int) t->n_levels - 1, uvx, uvy);
    const uint32_t *L = &s->tex_levels[(t->first_level + (uint32_t) level) * 3];
    float u = uvx * (float) (int) L[0] - 0.5f, v = uvy * (float) (int) L[1] - 0.5f;
    int xPos = (int) floorf(u), yPos = (int) floorf(v);
    float dx1 = u - (float) xPos, dx2 = 1.0f - dx1, dy1 = v - (float) yPos, dy2 = 1.0f - dy1;
    v3 r = scale(scale(mip_texel(s, t, level, xPos, yPos), dx2), dy2);
    r = add(r, scale(scale(mip_texel(s, t, level, xPos, yPos + 1), dx2), dy1));
    r = add(r, scale(scale(mip_texel(s, t, level, xPos + 1, yPos), dx1), dy2));
    r = add(r, scale(scale(mip_texel(s, t, level, xPos + 1, yPos + 1), dx1), dy1));
    return r;
}
static v3 mip_ewa(const orc_scene *s, const orc_texture *t, int level, float uvx, float uvy, float A, float B, float C) {
    if (!isfinite(A + B + C + uvx + uvy)) return V(0, 0, 0);
    if (level >= (int) t->n_levels) return mip_box(s, t, (int) t->n_levels - 1, uvx, uvy);
    const uint32_t *L = &s->tex_levels[(t->first_level + (uint32_t) level) * 3], *L0 = &s->tex_levels[t->first_level * 3];
    float u = uvx * (float) (int) L[0] - 0.5f, v = uvy * (float) (int) L[1] - 0.5f;
    const float rx = (float) (int) L[0] / (float) (int) L0[0], ry = (float) (int) L[1] / (float) (int) L0[1];     /* m_sizeRatio[level] */
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
                if (qi < 64) { const float weight = s->mip_lut[(int) q]; result = add(result, scale(mip_texel(s, t, level, ut, vt), weight)); denominator += weight; }
            }
            q += dq; dq += ddq;
        }
    }
    if (denominator == 0) return mip_bilinear(s, t, level, uvx, uvy);
    { float r = 1.0f / denominator; return scale(result, r); }
}
static float hypot2f(float a, float b) {                       /* src/libcore/math.cpp:74-86 */
    float r;
    if (fabsf(a) > fabsf(b)) { r = b / a; r = fabsf(a) * sqrtf(1.0f + r * r); }
    else if (b != 0.0f) { r = a / b; r = fabsf(b) * sqrtf(1.0f + r * r); }
    else r = 0.0f;
    return r;
}
static inline float mi_log2(float v) { const float invLn2 = 1.0f / logf(2.0f); return (float) log((double) v) * invLn2; }   /* math.cpp:103-106 */
static v3 mip_eval(const orc_scene *s, const orc_texture *t, float uvx, float uvy, float d0x, float d0y, float d1x, float d1y) {
    if (t->filter == 0) return mip_box(s, t, 0, uvx, uvy);
    if (t->filter == 1) return mip_bilinear(s, t, 0, uvx, uvy);
    const uint32_t *L0 = &s->tex_levels[t->first_level * 3]; const float sx = (float) (int) L0[0], sy = (float) (int) L0[1];
    float du0 = d0x * sx, dv0 = d0y * sy, du1 = d1x * sx, dv1 = d1y * sy;
    float A = dv0 * dv0 + dv1 * dv1, B = -2.0f * (du0 * dv0 + du1 * dv1), C = du0 * du0 + du1 * du1, F = A * C - B * B * 0.25f;
    float root = hypot2f(A - C, B), Aprime = 0.5f * (A + C - root), Cprime = 0.5f * (A + C + root),
          majorRadius = Aprime != 0 ? sqrtf(F / Aprime) : 0, minorRadius = Cprime != 0 ? sqrtf(F / Cprime) : 0;
    if (t->filter == 2 || !(minorRadius > 0) || !(majorRadius > 0) || F < 0) {
        float level = mi_log2(maxf(majorRadius, EPSILON)); int ilevel = (int) floorf(level);
        if (ilevel < 0) return mip_bilinear(s, t, 0, uvx, uvy);
        float a = level - (float) ilevel;
        return add(scale(mip_bilinear(s, t, ilevel, uvx, uvy), 1.0f - a), scale(mip_bilinear(s, t, ilevel + 1, uvx, uvy), a));
    }
    if (minorRadius * t->max_anisotropy < majorRadius) {
        minorRadius = majorRadius / t->max_anisotropy;
        float theta = 0.5f * atanf(B / (A - C)), sinTheta = sinf(theta), cosTheta = cosf(theta);
        float a2 = majorRadius * majorRadius, b2 = minorRadius * minorRadius, sinTheta2 = sinTheta * sinTheta, cosTheta2 = cosTheta * cosTheta, sin2Theta = 2 * sinTheta * cosTheta;
        A = a2 * cosTheta2 + b2 * sinTheta2; B = (a2 - b2) * sin2Theta; C = a2 * sinTheta2 + b2 * cosTheta2; F = a2 * b2;
    }
    float sc = 1.0f / F; A *= sc; B *= sc; C *= sc;
    float level = maxf(0.0f, mi_log2(minorRadius)); int ilevel = (int) level; float a = level - (float) ilevel;
    if (majorRadius < 1 || !(A > 0 && C > 0)) return mip_bilinear(s, t, ilevel, uvx, uvy);
    return add(scale(mip_ewa(s, t, ilevel, uvx, uvy, A, B, C), 1.0f - a), scale(mip_ewa(s, t, ilevel + 1, uvx, uvy, A, B, C), a));
}
/* Intersection::computePartials (src/librender/intersection.cpp:5-76) for a hit of the CAMERA ray (rx/ry: its differentials, common origin o) */
static int compute_partials(const hit_t *h, v3 o, v3 rxd, v3 ryd, float *dudx, float *dvdx, float *dudy, float *dvdy) {
    *dudx = *dvdx = *dudy = *dvdy = 0.0f;
    if (is_zero(h->dpdu) && is_zero(h->dpdv)) return 1;
    const float pp = dot(h->ng, h->p), pox = dot(h->ng, o), poy = dot(h->ng, o), prx = dot(h->ng, rxd), pry = dot(h->ng, ryd);
    if (prx == 0 || pry == 0) return 1;
    const float tx = (pp - pox) / prx, ty = (pp - poy) / pry;
    float absX = fabsf(h->ng.x), absY = fabsf(h->ng.y), absZ = fabsf(h->ng.z); int a0, a1;
    if (absX > absY && absX > absZ) { a0 = 1; a1 = 2; } else if (absY > absZ) { a0 = 0; a1 = 2; } else { a0 = 0; a1 = 1; }
    float A00 = comp(h->dpdu, a0), A01 = comp(h->dpdv, a0), A10 = comp(h->dpdu, a1), A11 = comp(h->dpdv, a1);
    v3 px = add(o, scale(rxd, tx)), py = add(o, scale(ryd, ty));
    float Bx0 = comp(px, a0) - comp(h->p, a0), Bx1 = comp(px, a1) - comp(h->p, a1), By0 = comp(py, a0) - comp(h->p, a0), By1 = comp(py, a1) - comp(h->p, a1);
    float det = A00 * A11 - A01 * A10;
    if (fabsf(det) <= 0x1p-128f) { *dudx = 1; *dvdx = 0; *dudy = 0; *dudy = 1; return 1; }       /* solveLinearSystem2x2 fails (util.cpp:529-541); NB the reference's second fallback assigns dudy twice */
    float inverse = 1.0f / det;
    *dudx = (A11 * Bx0 - A01 * Bx1) * inverse; *dvdx = (A00 * Bx1 - A10 * Bx0) * inverse;
    *dudy = (A11 * By0 - A01 * By1) * inverse; *dvdy = (A00 * By1 - A10 * By0) * inverse;
    return 1;
}
/* ---- 2-D procedural textures: Texture2D::eval (src/librender/texture.cpp:112-121, no filtering: usesRayDifferentials() = false), Checkerboard::eval
 * (src/textures/checkerboard.cpp:68-76), GridTexture::eval (src/textures/gridtexture.cpp:63-77) */
/* partials: NULL (no ray differentials: every hit but the camera ray's) or {dudx, dvdx, dudy, dvdy} */
static v3 texture_eval(const orc_scene *s, const orc_texture *t, float u, float v, const float *partials) {
    float uvx = u * t->uscale + t->uoffset, uvy = v * t->vscale + t->voffset;
    if (t->type == 2) {                                           /* BitmapTexture::eval (src/textures/bitmap.cpp:434-502) */
        if (partials) return mip_eval(s, t, uvx, uvy, partials[0] * t->uscale, partials[1] * t->vscale, partials[2] * t->uscale, partials[3] * t->vscale);
        return t->filter != 0 ? mip_bilinear(s, t, 0, uvx, uvy) : mip_box(s, t, 0, uvx, uvy);
    }
    int first;
    if (t->type == 0) {
        int a = (int) (uvx * 2) % 2, b = (int) (uvy * 2) % 2; if (a < 0) a += 2; if (b < 0) b += 2;
        int x = 2 * a - 1, y = 2 * b - 1;
        first = x * y == 1;
    } else {
        float x = uvx - (float) (int) floorf(uvx), y = uvy - (float) (int) floorf(uvy);
        if (x > .5) x -= 1;
        if (y > .5) y -= 1;
        first = !(fabsf(x) < t->line_width || fabsf(y) < t->line_width);
    }
    return first ? V(t->color0[0], t->color0[1], t->color0[2]) : V(t->color1[0], t->color1[1], t->color1[2]);
}
/* ---- mask: src/bsdfs/mask.cpp:104-229.  A material record of type 9 puts an opacity (its `reflectance`, constant or textured) in front of the nested material
 * record `distr`: eval = nested * opacity, pdf = nested * luminance(opacity); sample picks the nested BSDF with probability prob = luminance(opacity) (sample.x
 * rescaled) or passes straight through (wo = -wi, an ENull component: the path stays "unscattered").  Its component list holds EBackSide -> refN = 0. */
static void apply_texture(const orc_scene *s, mat_t *m, const hit_t *its, const float *partials_or_null, int want_partials, v3 o, const v3 *rxd, const v3 *ryd) {
    uint32_t tex = (m->m.flags >> 8) & 0xFFFFu; if (!tex) return;
    float pa[4]; const float *partials = partials_or_null;
    if (want_partials && s->textures[tex - 1].type == 2) { compute_partials(its, o, *rxd, *ryd, &pa[0], &pa[1], &pa[2], &pa[3]); partials = pa; }
    v3 c = texture_eval(s, &s->textures[tex - 1], its->uvx, its->uvy, partials); m->m.reflectance[0] = c.x; m->m.reflectance[1] = c.y; m->m.reflectance[2] = c.z;
}
/* ---- bumpmap / normalmap: src/bsdfs/bumpmap.cpp:140-250, normalmap.cpp:108-260.  A record of type 11 / 12 perturbs the shading frame of the nested record
 * `distr` from its bound texture (bumpmap: luminance of the displacement's uv gradient, `alpha` = the factor of an enclosing `scale` texture; normalmap: the
 * texel as a tangent-space normal); eval / pdf / sample run the nested BSDF in the perturbed frame and reject directions whose cosines differ in sign between
 * the two frames.  Texture2D::evalGradient (src/librender/texture.cpp:123-141): finite differences over eps = 1e-4 for the procedural textures, the bilinear
 * gradient of MIP level 0 for bitmaps (src/textures/bitmap.cpp:459-483, include/mitsuba/render/mipmap.h:602-627). */
static void mip_gradient_bilinear(const orc_scene *s, const orc_texture *t, float uvx, float uvy, v3 *gu, v3 *gv) {
    *gu = *gv = V(0, 0, 0);
    if (!isfinite(uvx) || !isfinite(uvy)) return;
    const uint32_t *L = &s->tex_levels[t->first_level * 3]; const float sx = (float) (int) L[0], sy = (float) (int) L[1];
    float u = uvx * sx - 0.5f, v = uvy * sy - 0.5f;
    int xPos = (int) floorf(u), yPos = (int) floorf(v);
    float dx = u - (float) xPos, dy = v - (float) yPos;
    v3 p00 = mip_texel(s, t, 0, xPos, yPos), p10 = mip_texel(s, t, 0, xPos + 1, yPos), p01 = mip_texel(s, t, 0, xPos, yPos + 1), p11 = mip_texel(s, t, 0, xPos + 1, yPos + 1);
    v3 tmp = sub(add(p01, p10), p11);
    *gu = scale(sub(add(p10, scale(p00, dy - 1)), scale(tmp, dy)), sx);
    *gv = scale(sub(add(p01, scale(p00, dx - 1)), scale(tmp, dx)), sy);
}
static void texture_gradient(const orc_scene *s, const orc_texture *t, float u, float v, v3 *gu, v3 *gv) {
    float uvx = u * t->uscale + t->uoffset, uvy = v * t->vscale + t->voffset;
    if (t->type == 2) { if (t->filter != 0) mip_gradient_bilinear(s, t, uvx, uvy, gu, gv); else *gu = *gv = V(0, 0, 0); }
    else {
        orc_texture raw = *t; raw.uscale = raw.vscale = 1.0f; raw.uoffset = raw.voffset = 0.0f;      /* eval(uv) on the already transformed coordinates */
        const float eps = EPSILON;
        v3 value = texture_eval(s, &raw, uvx, uvy, NULL), valueU = texture_eval(s, &raw, uvx + eps, uvy, NULL), valueV = texture_eval(s, &raw, uvx, uvy + eps, NULL);
        *gu = scale(sub(valueU, value), 1 / eps); *gv = scale(sub(valueV, value), 1 / eps);
    }
    *gu = scale(*gu, t->uscale); *gv = scale(*gv, t->vscale);
}
static void perturb_frame(const orc_scene *s, const orc_material *m, const hit_t *its, v3 *ps, v3 *pt, v3 *pn) {
    const orc_texture *tx = &s->textures[((m->flags >> 8) & 0xFFFFu) - 1];
    if (m->type == BSDF_BUMPMAP) {                                        /* BumpMap::getFrame (bumpmap.cpp:140-163) */
        v3 gu, gv; texture_gradient(s, tx, its->uvx, its->uvy, &gu, &gv);
        gu = scale(gu, m->alpha); gv = scale(gv, m->alpha);               /* ScaleTexture::evalGradient (src/textures/scale.cpp:93-97) */
        float dDispDu = luminance(gu), dDispDv = luminance(gv);
        v3 dpdu = add(its->dpdu, scale(its->ns, dDispDu - dot(its->ns, its->dpdu)));
        v3 dpdv = add(its->dpdv, scale(its->ns, dDispDv - dot(its->ns, its->dpdv)));
        v3 n = normalize(cross(dpdu, dpdv));
        *ps = normalize(sub(dpdu, scale(n, dot(n, dpdu)))); *pt = cross(n, *ps);
        if (dot(n, its->ng) < 0) n = scale(n, -1.0f);
        *pn = n;
    } else {                                                              /* NormalMap::getFrame (normalmap.cpp:108-127): Texture::eval(its, false) */
        v3 c = texture_eval(s, tx, its->uvx, its->uvy, NULL);
        v3 nl = V(2 * c.x - 1, 2 * c.y - 1, 2 * c.z - 1);
        v3 n = normalize(add(add(scale(its->s, nl.x), scale(its->tt, nl.y)), scale(its->ns, nl.z)));
        *ps = normalize(sub(its->dpdu, scale(n, dot(n, its->dpdu)))); *pt = cross(n, *ps); *pn = n;
    }
}
static smat_t resolve_material(const orc_scene *s, uint32_t material, const hit_t *its, int want_partials, v3 o, const v3 *rxd, const v3 *ryd) {
    smat_t sm; sm.inner = s->materials[material]; sm.masked = 0; sm.pdfless = s->d.integrator == 1;      /* volpath_simple calls BSDF::sample(bRec, sample), the overload without a pdf */
     sm.opacity = V(1, 1, 1); sm.prob = 1.0f; sm.bumped = 0; sm.n_mix = 0; sm.its = its; sm.coated = 0; sm.blend = 0;
    if (its && sm.inner.m.type != BSDF_BUMPMAP && sm.inner.m.type != BSDF_NORMALMAP) apply_texture(s, &sm.inner, its, NULL, want_partials, o, rxd, ryd);
    if (sm.inner.m.type == BSDF_MASK) {
        sm.masked = 1; sm.opacity = V(sm.inner.m.reflectance[0], sm.inner.m.reflectance[1], sm.inner.m.reflectance[2]); sm.prob = luminance(sm.opacity);
        sm.inner = s->materials[sm.inner.m.distr];
        if (its && sm.inner.m.type != BSDF_BUMPMAP && sm.inner.m.type != BSDF_NORMALMAP) apply_texture(s, &sm.inner, its, NULL, want_partials, o, rxd, ryd);
    }
    if (sm.inner.m.type == BSDF_BUMPMAP || sm.inner.m.type == BSDF_NORMALMAP) {
        if (its) { sm.bumped = 1; perturb_frame(s, &sm.inner.m, its, &sm.ps, &sm.pt, &sm.pn); }
        sm.inner = s->materials[sm.inner.m.distr];
        if (its) apply_texture(s, &sm.inner, its, NULL, want_partials, o, rxd, ryd);
    }
    if (sm.inner.m.type == BSDF_ROUGHCOATING) {                           /* RoughCoating::configure (roughcoating.cpp:183-230): eta[0] = eta, eta[1] = thickness, eta[2] = microfacet type, alpha, k[1..2] = the transmittance slice */
        sm.coated = 2; sm.coatm = sm.inner; const orc_material *c = &sm.coatm.m;
        float e0 = fastexpf_(c->reflectance[0] * (-2 * c->eta[1])), e1 = fastexpf_(c->reflectance[1] * (-2 * c->eta[1])), e2 = fastexpf_(c->reflectance[2] * (-2 * c->eta[1]));
        float avgAbsorption = 0.0f; avgAbsorption += e0; avgAbsorption += e1; avgAbsorption += e2; avgAbsorption = avgAbsorption * (1.0f / 3);
        sm.coat_w = 1.0f / (avgAbsorption + 1.0f);
        sm.inner = s->materials[c->distr];
        if (its) apply_texture(s, &sm.inner, its, NULL, want_partials, o, rxd, ryd);
    }
    if (sm.inner.m.type == BSDF_COATING) {                                /* SmoothCoating::configure (coating.cpp:160-192): reflectance = sigmaA, alpha = thickness, eta[0] = eta, specular */
        sm.coated = 1; sm.coatm = sm.inner;
        float e0 = fastexpf_(sm.coatm.m.reflectance[0] * (-2 * sm.coatm.m.alpha)), e1 = fastexpf_(sm.coatm.m.reflectance[1] * (-2 * sm.coatm.m.alpha)), e2 = fastexpf_(sm.coatm.m.reflectance[2] * (-2 * sm.coatm.m.alpha));
        float avgAbsorption = 0.0f; avgAbsorption += e0; avgAbsorption += e1; avgAbsorption += e2; avgAbsorption = avgAbsorption * (1.0f / 3);
        sm.coat_w = 1.0f / (avgAbsorption + 1.0f);
        sm.inner = s->materials[sm.coatm.m.distr];
        if (its) apply_texture(s, &sm.inner, its, NULL, want_partials, o, rxd, ryd);
    }
    if (sm.inner.m.type == BSDF_BLEND) {                                  /* BlendBSDF (blendbsdf.cpp:138-141): weight = clamp(m_weight->eval(its).average(), 0, 1); `reflectance` holds the texture's value (or w, w, w) */
        const orc_material *bl = &sm.inner.m; sm.n_mix = 2; sm.blend = 1;
        float avg = 0.0f; avg += bl->reflectance[0]; avg += bl->reflectance[1]; avg += bl->reflectance[2]; avg = avg * (1.0f / 3);
        sm.w[1] = minf(1.0f, maxf(0.0f, avg)); sm.w[0] = 1 - sm.w[1]; sm.p[0] = sm.w[0]; sm.p[1] = sm.w[1];
        for (int i = 0; i < 2; ++i) { sm.mix[i] = s->materials[(uint32_t) bl->eta[i]]; sm.mix[i].m.flags &= ~THIN_SIGNED_COS; }
    }
    if (sm.inner.m.type == BSDF_MIXTURE) {                                /* MixtureBSDF::configure (mixturebsdf.cpp:115-169) */
        const orc_material *mx = &sm.inner.m; sm.n_mix = (int) mx->distr; float total = 0;
        for (int i = 0; i < sm.n_mix; ++i) {
            sm.mix[i] = s->materials[(uint32_t) (i < 3 ? mx->reflectance[i] : mx->eta[0])]; sm.w[i] = i < 3 ? mx->k[i] : mx->specular[0];
            sm.mix[i].m.flags &= ~THIN_SIGNED_COS;                        /* both MixtureBSDF::sample overloads call the child's sample WITH a pdf (mixturebsdf.cpp:217, :251) */
            total += sm.w[i];
        }
        if (total > 1) { float sc = 1.0f / total; for (int i = 0; i < sm.n_mix; ++i) sm.w[i] *= sc; }      /* ensureEnergyConservation (default true) */
        /* DiscreteDistribution: append + normalize (include/mitsuba/core/pmf.h:56-58, 103-116): cdf by running sums, normalised by the total */
        float *cdf = sm.cdf; cdf[0] = 0; for (int i = 0; i < sm.n_mix; ++i) cdf[i + 1] = cdf[i] + sm.w[i];
        float norm = 1.0f / cdf[sm.n_mix]; for (int i = 1; i <= sm.n_mix; ++i) cdf[i] *= norm; cdf[sm.n_mix] = 1.0f;
        for (int i = 0; i < sm.n_mix; ++i) sm.p[i] = cdf[i + 1] - cdf[i];
    }
    return sm;
}
/* the nested level below mask and bump / normal map: a plain BSDF or a mixture of plain BSDFs (mixturebsdf.cpp:171-276, path tracer: component = -1) */
static v3 mx_eval(const smat_t *sm, v3 wi, v3 wo) {
    if (!sm->n_mix) return bsdf_eval(&sm->inner.m, wi, wo);
    int flip = (sm->inner.m.flags & BSDF_FLAG_TWOSIDED) && wi.z < 0; if (flip) { wi.z = -wi.z; wo.z = -wo.z; }     /* twosided(mixture) */
    v3 r = V(0, 0, 0); for (int i = 0; i < sm->n_mix; ++i) r = add(r, scale(bsdf_eval(&sm->mix[i].m, wi, wo), sm->w[i]));
    return r;
}
static float mx_pdf(const smat_t *sm, v3 wi, v3 wo) {
    if (!sm->n_mix) return bsdf_pdf(&sm->inner.m, wi, wo);
    int flip = (sm->inner.m.flags & BSDF_FLAG_TWOSIDED) && wi.z < 0; if (flip) { wi.z = -wi.z; wo.z = -wo.z; }
    float r = 0; for (int i = 0; i < sm->n_mix; ++i) r += bsdf_pdf(&sm->mix[i].m, wi, wo) * sm->p[i];
    return r;
}
static v3 mx_sample(const smat_t *sm, v3 wi, float u, float v, v3 *wo, float *pdf, float *eta, int *delta, sampler_t *sp) {
    if (!sm->n_mix) return bsdf_sample(&sm->inner.m, wi, u, v, wo, pdf, eta, delta, sp);
    int flip = (sm->inner.m.flags & BSDF_FLAG_TWOSIDED) && wi.z < 0; if (flip) wi.z = -wi.z;
    /* m_pdf.sampleReuse(sample.x) (pmf.h:124-190): lower_bound over the cdf, then rescale */
    const float *cdf = sm->cdf;
    int entry = 0;
    if (sm->blend) {                                                      /* blendbsdf.cpp:226-232 */
        if (u < sm->w[0]) { entry = 0; u /= sm->w[0]; } else { entry = 1; u = (u - sm->w[0]) / sm->w[1]; }
    } else { int lo = 0, hi = sm->n_mix + 1; while (lo < hi) { int mid = (lo + hi) >> 1; if (cdf[mid] < u) lo = mid + 1; else hi = mid; }
                     entry = lo > 0 ? lo - 1 : 0; if (entry > sm->n_mix - 1) entry = sm->n_mix - 1;
                     while (cdf[entry + 1] - cdf[entry] == 0 && entry < sm->n_mix) ++entry;
                     u = (u - cdf[entry]) / (cdf[entry + 1] - cdf[entry]); }
    v3 result = bsdf_sample(&sm->mix[entry].m, wi, u, v, wo, pdf, eta, delta, sp);
    if (is_zero(result)) return result;
    result = scale(result, sm->w[entry] * *pdf); *pdf *= sm->p[entry];
    if (!*delta)                                                          /* measure of the sampled component: the others contribute in solid angle only */
        for (int i = 0; i < sm->n_mix; ++i) {
            if (i == entry) continue;
            *pdf += bsdf_pdf(&sm->mix[i].m, wi, *wo) * sm->p[i];
            result = add(result, scale(bsdf_eval(&sm->mix[i].m, wi, *wo), sm->w[i]));
        }
    { float r = 1.0f / *pdf; result = scale(result, r); }                 /* Spectrum::operator/(Float): reciprocal, then multiply */
    if (flip) wo->z = -wo->z;
    return result;
}
/* ---- smooth dielectric coating: src/bsdfs/coating.cpp:205-372 around the level below (queried with component = -1, typeMask = EAll, measure = ESolidAngle on this path) */
static v3 ct_refract_in(v3 wi, float eta, float invEta, float *R) {      /* :214-218; math::signum = copysignf(1, x) (math.h:270-278) */
    float cosThetaT; *R = fresnel_dielectric_ext(fabsf(wi.z), &cosThetaT, eta);
    return V(invEta * wi.x, invEta * wi.y, -copysignf(1.0f, wi.z) * cosThetaT);
}
static v3 ct_refract_out(v3 wi, float eta, float invEta, float *R) {     /* :221-225 */
    float cosThetaT; *R = fresnel_dielectric_ext(fabsf(wi.z), &cosThetaT, invEta);
    return V(eta * wi.x, eta * wi.y, -copysignf(1.0f, wi.z) * cosThetaT);
}
static v3 ct_absorb(const orc_material *c, v3 result, v3 wiP, v3 woP) {  /* Spectrum::exp = math::fastexp per channel (spectrum.h:521-526) */
    v3 sigmaA = scale(V(c->reflectance[0], c->reflectance[1], c->reflectance[2]), c->alpha);
    if (is_zero(sigmaA)) return result;
    float f = 1 / fabsf(wiP.z) + 1 / fabsf(woP.z);
    return mul(result, V(fastexpf_(-sigmaA.x * f), fastexpf_(-sigmaA.y * f), fastexpf_(-sigmaA.z * f)));
}
/* ---- rough dielectric coating: src/bsdfs/roughcoating.cpp:236-443.  The layer's interface is a microfacet surface (MicrofacetDistribution(type, alpha, sampleVisible),
 * isotropic); directions enter and leave the layer by Snell's law alone (refractTo, :239-259), the energy balance is the precomputed rough transmittance (rtrans.h, the
 * same 1-D slice as roughplastic's: table at k[1], length k[2]).  Nested BSDFs with delta lobes are not supported here (they would need EDiscrete queries) */
static v3 rct_refract_to(int interior, v3 wi, float eta, float invEtaC) {
    float invEta = interior ? invEtaC : eta; int entering = wi.z > 0.0f;
    float sinThetaTSqr = invEta * invEta * (1.0f - wi.z * wi.z);
    if (sinThetaTSqr >= 1.0f) return V(0, 0, 0);
    float cosThetaT = sqrtf(1.0f - sinThetaTSqr);
    return V(invEta * wi.x, invEta * wi.y, entering ? cosThetaT : -cosThetaT);
}
static mfd_t rct_distr(const orc_material *c) { mfd_t d; d.distr = (uint32_t) c->eta[2]; d.au = d.av = maxf(avg3(c->alpha), 1e-4f); d.visible = (c->flags & 2u) != 0 && d.distr != 2; return d; }
static float rct_prob_specular(const smat_t *sm, float cosThetaI) {
    float p = 1 - rp_transmittance(&sm->coatm.m, fabsf(cosThetaI)), w = sm->coat_w;
    return (p * w) / (p * w + (1 - p) * (1 - w));
}
static v3 rct_absorb(const orc_material *c, v3 result, v3 wiP, v3 woP) {
    v3 sigmaA = scale(V(c->reflectance[0], c->reflectance[1], c->reflectance[2]), c->eta[1]);
    if (is_zero(sigmaA)) return result;
    float f = 1 / fabsf(wiP.z) + 1 / fabsf(woP.z);
    return mul(result, V(fastexpf_(-sigmaA.x * f), fastexpf_(-sigmaA.y * f), fastexpf_(-sigmaA.z * f)));
}
static v3 rct_eval(const smat_t *sm, v3 wi, v3 wo) {                      /* :261-322 */
    const orc_material *c = &sm->coatm.m; const float eta = c->eta[0], invEta = 1 / eta; const mfd_t d = rct_distr(c);
    if ((c->flags & BSDF_FLAG_TWOSIDED) && wi.z < 0) { wi.z = -wi.z; wo.z = -wo.z; }
    v3 result = V(0, 0, 0);
    if (wo.z * wi.z > 0) {
        v3 H = scale(normalize(add(wo, wi)), copysignf(1.0f, wo.z)); float ct;
        float D = mf_eval2(d.distr, d.au, d.av, H), F = fresnel_dielectric_ext(fabsf(dot(wi, H)), &ct, eta);
        float G = mf_smith_g1_2(d.distr, d.au, d.av, wi, H) * mf_smith_g1_2(d.distr, d.au, d.av, wo, H);
        float value = F * D * G / (4.0f * fabsf(wi.z));
        result = add(result, scale(V(c->specular[0], c->specular[1], c->specular[2]), value));
    }
    v3 wiP = rct_refract_to(1, wi, eta, invEta), woP = rct_refract_to(1, wo, eta, invEta);
    v3 nested = scale(scale(mx_eval(sm, wiP, woP), rp_transmittance(c, fabsf(wi.z))), rp_transmittance(c, fabsf(wo.z)));
    nested = rct_absorb(c, nested, wiP, woP);
    nested = scale(nested, invEta * invEta * wo.z / woP.z);
    return add(result, nested);
}
static float rct_pdf(const smat_t *sm, v3 wi, v3 wo) {                    /* :324-383 */
    const orc_material *c = &sm->coatm.m; const float eta = c->eta[0], invEta = 1 / eta; const mfd_t d = rct_distr(c);
    if ((c->flags & BSDF_FLAG_TWOSIDED) && wi.z < 0) { wi.z = -wi.z; wo.z = -wo.z; }
    v3 H = scale(normalize(add(wo, wi)), copysignf(1.0f, wo.z));
    float probSpecular = rct_prob_specular(sm, wi.z), probNested = 1 - probSpecular, result = 0.0f;
    if (wo.z * wi.z > 0) {
        float dwh_dwo = 1.0f / (4.0f * fabsf(dot(wo, H))), prob = mfd_pdf(&d, wi, H);
        result = prob * dwh_dwo * probSpecular;
    }
    v3 wiP = rct_refract_to(1, wi, eta, invEta), woP = rct_refract_to(1, wo, eta, invEta);
    float prob = mx_pdf(sm, wiP, woP);
    prob *= invEta * invEta * wo.z / woP.z;
    result += prob * probNested;
    return result;
}
static v3 rct_sample(const smat_t *sm, v3 wi, float sx, float sy, v3 *wo, float *pdf, float *etaOut, int *delta, sampler_t *sp) {      /* :385-443 */
    const orc_material *c = &sm->coatm.m; const float eta = c->eta[0], invEta = 1 / eta; const mfd_t d = rct_distr(c);
    int flip = (c->flags & BSDF_FLAG_TWOSIDED) && wi.z < 0; if (flip) wi.z = -wi.z;
    float probSpecular = rct_prob_specular(sm, wi.z); int choseSpecular = 1;
    if (sy < probSpecular) sy /= probSpecular; else { sy = (sy - probSpecular) / (1 - probSpecular); choseSpecular = 0; }
    *delta = 0;
    if (choseSpecular) {
        float mpdf; v3 m = mfd_sample(&d, wi, sx, sy, &mpdf);
        float cc = 2 * dot(wi, m); *wo = sub(scale(m, cc), wi); *etaOut = 1.0f;
        if (wo->z * wi.z <= 0) return V(0, 0, 0);
    } else {
        v3 wiP = rct_refract_to(1, wi, eta, invEta), woP = V(0, 0, 0);
        v3 r = mx_sample(sm, wiP, sx, sy, &woP, pdf, etaOut, delta, sp);      /* (the nested pdf stays in *pdf when the way out fails, as in the reference: the weight is zero then) */
        if (is_zero(r)) return V(0, 0, 0);
        *wo = rct_refract_to(0, woP, eta, invEta);
        if (is_zero(*wo)) return V(0, 0, 0);
    }
    *pdf = rct_pdf(sm, wi, *wo);                                          /* "guard against numerical imprecisions": pdf and value are re-evaluated (:432-438) */
    if (*pdf == 0) return V(0, 0, 0);
    float r = 1.0f / *pdf; v3 result = scale(rct_eval(sm, wi, *wo), r);
    if (flip) wo->z = -wo->z;
    return result;
}
static v3 ct_eval(const smat_t *sm, v3 wi, v3 wo) {                       /* :227-265 */
    if (!sm->coated) return mx_eval(sm, wi, wo);
    if (sm->coated == 2) return rct_eval(sm, wi, wo);
    const orc_material *c = &sm->coatm.m; const float eta = c->eta[0], invEta = 1 / eta;
    if ((c->flags & BSDF_FLAG_TWOSIDED) && wi.z < 0) { wi.z = -wi.z; wo.z = -wo.z; }
    float R12, R21; v3 wiP = ct_refract_in(wi, eta, invEta, &R12), woP = ct_refract_in(wo, eta, invEta, &R21);
    if (R12 == 1 || R21 == 1) return V(0, 0, 0);
    v3 result = scale(scale(mx_eval(sm, wiP, woP), 1 - R12), 1 - R21);
    result = ct_absorb(c, result, wiP, woP);
    return scale(result, invEta * invEta * wo.z / woP.z);
}
static float ct_prob_specular(const smat_t *sm, float R12) { return (R12 * sm->coat_w) / (R12 * sm->coat_w + (1 - R12) * (1 - sm->coat_w)); }
static float ct_pdf(const smat_t *sm, v3 wi, v3 wo) {                     /* :267-303 */
    if (!sm->coated) return mx_pdf(sm, wi, wo);
    if (sm->coated == 2) return rct_pdf(sm, wi, wo);
    const orc_material *c = &sm->coatm.m; const float eta = c->eta[0], invEta = 1 / eta;
    if ((c->flags & BSDF_FLAG_TWOSIDED) && wi.z < 0) { wi.z = -wi.z; wo.z = -wo.z; }
    float R12, R21; v3 wiP = ct_refract_in(wi, eta, invEta, &R12); float probSpecular = ct_prob_specular(sm, R12);
    v3 woP = ct_refract_in(wo, eta, invEta, &R21);
    if (R12 == 1 || R21 == 1) return 0.0f;
    float pdf = mx_pdf(sm, wiP, woP);
    pdf *= invEta * invEta * wo.z / woP.z;
    return pdf * (1 - probSpecular);
}
static v3 ct_sample(const smat_t *sm, v3 wi, float u, float v, v3 *wo, float *pdf, float *etaOut, int *delta, sampler_t *sp) {      /* :305-372 */
    if (!sm->coated) return mx_sample(sm, wi, u, v, wo, pdf, etaOut, delta, sp);
    if (sm->coated == 2) return rct_sample(sm, wi, u, v, wo, pdf, etaOut, delta, sp);
    const orc_material *c = &sm->coatm.m; const float eta = c->eta[0], invEta = 1 / eta;
    int flip = (c->flags & BSDF_FLAG_TWOSIDED) && wi.z < 0; if (flip) wi.z = -wi.z;
    float R12; v3 wiP = ct_refract_in(wi, eta, invEta, &R12); float probSpecular = ct_prob_specular(sm, R12);
    v3 result;
    if (u < probSpecular) {
        *wo = V(-wi.x, -wi.y, wi.z); *etaOut = 1.0f; *pdf = probSpecular; *delta = 1;
        result = scale(V(c->specular[0], c->specular[1], c->specular[2]), R12 / *pdf);
    } else {
        u = (u - probSpecular) / (1 - probSpecular);
        if (R12 == 1.0f) return V(0, 0, 0);
        v3 woP = V(0, 0, 0);
        result = mx_sample(sm, wiP, u, v, &woP, pdf, etaOut, delta, sp);
        if (is_zero(result)) return V(0, 0, 0);
        result = ct_absorb(c, result, wiP, woP);
        float R21; *wo = ct_refract_out(woP, eta, invEta, &R21);
        if (R21 == 1.0f) return V(0, 0, 0);
        *pdf *= 1.0f - probSpecular; { float r = 1.0f / (1.0f - probSpecular); result = scale(result, r); }
        result = scale(result, (1 - R12) * (1 - R21));
        if (!*delta) *pdf *= invEta * invEta * wo->z / woP.z;
    }
    if (flip) wo->z = -wo->z;
    return result;
}
static inline v3 frame_to_local(v3 fs, v3 ft, v3 fn, v3 w) { return V(dot(w, fs), dot(w, ft), dot(w, fn)); }
static inline v3 frame_to_world(v3 fs, v3 ft, v3 fn, v3 w) { return add(add(scale(fs, w.x), scale(ft, w.y)), scale(fn, w.z)); }
/* bump / normal map level (bumpmap.cpp:165-250): wi, wo are given in the hit's own frame */
static v3 bp_eval(const smat_t *sm, v3 wi, v3 wo) {
    if (!sm->bumped) return ct_eval(sm, wi, wo);
    const hit_t *h = sm->its; v3 wiP = frame_to_local(sm->ps, sm->pt, sm->pn, frame_to_world(h->s, h->tt, h->ns, wi)), woP = frame_to_local(sm->ps, sm->pt, sm->pn, frame_to_world(h->s, h->tt, h->ns, wo));
    if (wo.z * woP.z <= 0) return V(0, 0, 0);
    return ct_eval(sm, wiP, woP);
}
static float bp_pdf(const smat_t *sm, v3 wi, v3 wo) {
    if (!sm->bumped) return ct_pdf(sm, wi, wo);
    const hit_t *h = sm->its; v3 wiP = frame_to_local(sm->ps, sm->pt, sm->pn, frame_to_world(h->s, h->tt, h->ns, wi)), woP = frame_to_local(sm->ps, sm->pt, sm->pn, frame_to_world(h->s, h->tt, h->ns, wo));
    if (wo.z * woP.z <= 0) return 0.0f;
    return ct_pdf(sm, wiP, woP);
}
static v3 bp_sample(const smat_t *sm, v3 wi, float u, float v, v3 *wo, float *pdf, float *eta, int *delta, sampler_t *sp) {
    if (!sm->bumped) return ct_sample(sm, wi, u, v, wo, pdf, eta, delta, sp);
    const hit_t *h = sm->its; v3 wiP = frame_to_local(sm->ps, sm->pt, sm->pn, frame_to_world(h->s, h->tt, h->ns, wi)), woP = V(0, 0, 0);
    v3 result = ct_sample(sm, wiP, u, v, &woP, pdf, eta, delta, sp);
    if (!is_zero(result)) {
        *wo = frame_to_local(h->s, h->tt, h->ns, frame_to_world(sm->ps, sm->pt, sm->pn, woP));
        if (wo->z * woP.z <= 0) return V(0, 0, 0);
    }
    return result;
}
static v3 sm_eval(const smat_t *sm, v3 wi, v3 wo) { v3 e = bp_eval(sm, wi, wo); return sm->masked ? mul(e, sm->opacity) : e; }
static float sm_pdf(const smat_t *sm, v3 wi, v3 wo) { float p = bp_pdf(sm, wi, wo); return sm->masked ? p * sm->prob : p; }
static v3 sm_sample(const smat_t *sm, v3 wi, float u, float v, v3 *wo, float *pdf, float *eta, int *delta, sampler_t *sp) {
    if (!sm->masked) return bp_sample(sm, wi, u, v, wo, pdf, eta, delta, sp);
    if (u < sm->prob) {                                                  /* mask.cpp:196-201; the pdf-less overload multiplies by 1 / prob instead (:164-167) */
        const float invProb = 1.0f / sm->prob;
        if (sm->pdfless) u *= invProb; else u /= sm->prob;
        v3 w = bp_sample(sm, wi, u, v, wo, pdf, eta, delta, sp);
        v3 r = sm->pdfless ? V(w.x * sm->opacity.x * invProb, w.y * sm->opacity.y * invProb, w.z * sm->opacity.z * invProb)
                           : V(w.x * sm->opacity.x / sm->prob, w.y * sm->opacity.y / sm->prob, w.z * sm->opacity.z / sm->prob);
        *pdf *= sm->prob; return r;
    }
    *wo = neg(wi); *eta = 1.0f; *delta = 2; *pdf = 1 - sm->prob;       /* :202-208 */
    return V((1.0f - sm->opacity.x) / *pdf, (1.0f - sm->opacity.y) / *pdf, (1.0f - sm->opacity.z) / *pdf);
}
static int sm_is_smooth(const smat_t *sm) {
    if (!sm->n_mix) return material_is_smooth(&sm->inner.m);
    int r = 0; for (int i = 0; i < sm->n_mix; ++i) r |= material_is_smooth(&sm->mix[i].m); return r;      /* the mixture's components are its children's (mixturebsdf.cpp:150-166) */
}
static int sm_has_backside(const smat_t *sm) {
    if (sm->masked || sm->coated) return 1;                              /* coating.cpp:172-173: the layer's delta component is EFrontSide | EBackSide */
    if (!sm->n_mix) return material_has_backside(&sm->inner.m);
    int r = (sm->inner.m.flags & BSDF_FLAG_TWOSIDED) != 0; for (int i = 0; i < sm->n_mix; ++i) r |= material_has_backside(&sm->mix[i].m); return r;
}

/* envmap.cpp:384-416 evalEnvironment WITH ray differentials (the sensor ray, path.cpp:139-141): texture-space partials, then TMIPMap::eval over the
 * map's pyramid (input data; record s->d.env_texture - 1 of the texture table) */
static v3 env_eval_filtered(const orc_scene *s, v3 d, v3 rxd, v3 ryd) {
    v3 v = mat3(s->env_to_local, d);
    float uvx = atan2f(v.x, -v.z) * INV_TWOPI, uvy = acosf(minf(1.0f, maxf(-1.0f, v.y))) * INV_PI;
    v3 dvdx = sub(mat3(s->env_to_local, rxd), v), dvdy = sub(mat3(s->env_to_local, ryd), v);
    float t1 = INV_TWOPI / (v.x * v.x + v.z * v.z), t2 = -INV_PI / maxf(sqrtf(maxf(0.0f, 1.0f - v.y * v.y)), EPSILON);
    v3 value = mip_eval(s, &s->textures[s->d.env_texture - 1], uvx, uvy, t1 * (dvdx.z * v.x - dvdx.x * v.z), t2 * dvdx.y, t1 * (dvdy.z * v.x - dvdy.x * v.z), t2 * dvdy.y);
    return scale(value, s->env_scale);
}

/* ------------------------------------------------------------------------------------------------ the Li loop */
static inline float mi_weight(float a, float b) { a *= a; b *= b; return a / (a + b); }   /* path.cpp:296-300 */

/* src/integrators/path/path.cpp:119-294 MIPathTracer::Li (no media, no subsurface) */
static v3 path_li(const orc_scene *s, v3 o, v3 d, float mint, float maxt, sampler_t *sp, int *out_depth, uint64_t *counters, float *alpha, const v3 *rxd, const v3 *ryd) {
    const int maxDepth = s->d.max_depth, rrDepth = s->d.rr_depth;
    const int strict = s->d.strict_normals != 0, hide = s->d.hide_emitters != 0;
    hit_t its; v3 Li = V(0, 0, 0); int scattered = 0; int depth = 1;
    int emitted_radiance = 1;                           /* rRec.type & EEmittedRadiance; cleared after the first bounce (path.cpp:274) */
    ++counters[0];
    ray_intersect(s, o, d, mint, maxt, &its, 0);        /* records.inl:117-145 */
    *alpha = (s->d.opacity && !its.valid) ? 0.0f : 1.0f; /* records.inl:121-137 (no media): EOpacity -> 1 on a hit, 0 on a miss; else newQuery's 1 */
    v3 throughput = V(1, 1, 1); float eta = 1.0f;
    while (depth <= maxDepth || maxDepth < 0) {
        if (!its.valid) {                                /* path.cpp:136-143 (reached by camera rays only; BSDF-ray misses are handled below) */
            if (s->env_index >= 0 && emitted_radiance && (!hide || scattered))
                Li = add(Li, mul(throughput, (s->d.env_texture && !s->env_constant && depth == 1 && !scattered) ? env_eval_filtered(s, d, *rxd, *ryd) : env_eval(s, d)));
            break;
        }
        /* its.getBSDF(ray) -> computePartials: only the camera ray carries differentials (records.inl:68-75); textured parameters are evaluated at the hit's uv */
        const smat_t smat = resolve_material(s, its.material, &its, depth == 1 && !scattered, o, rxd, ryd); const smat_t *bsdf = &smat;
        if (its.emitter >= 0 && emitted_radiance && (!hide || scattered))
            Li = add(Li, mul(throughput, emitter_eval(s, its.emitter, its.ns, neg(d))));
        if ((depth >= maxDepth && maxDepth > 0) || (strict && dot(d, its.ng) * its.wi.z >= 0)) break;

        /* direct illumination sampling (path.cpp:172-200), only for BSDFs with a smooth component */
        v3 refN = sm_has_backside(bsdf) ? V(0, 0, 0) : its.ns;      /* records.inl:160-164 */
        direct_t dRec; memset(&dRec, 0, sizeof(dRec)); dRec.ref = its.p;
        if (sm_is_smooth(bsdf)) {
            float sx, sy; next2D(sp, &sx, &sy);
            v3 value = sample_emitter_direct(s, its.p, refN, sx, sy, &dRec, 1, &counters[1]);
            if (!is_zero(value)) {
                v3 wo = to_local(&its, dRec.d);
                v3 bsdfVal = sm_eval(bsdf, its.wi, wo);
                if (!is_zero(bsdfVal) && (!strict || dot(its.ng, dRec.d) * wo.z > 0)) {
                    float bsdfPdf = dRec.delta ? 0.0f : sm_pdf(bsdf, its.wi, wo);   /* emitter->isOnSurface() && measure == ESolidAngle (path.cpp:191-192) */
                    float weight = mi_weight(dRec.pdf, bsdfPdf);
                    Li = add(Li, scale(mul(mul(throughput, value), bsdfVal), weight));
                }
            }
        }
        /* BSDF sampling (path.cpp:207-219) */
        float bsdfPdf = 0, bEta = 1; v3 woL = V(0, 0, 0);
        float sx, sy; next2D(sp, &sx, &sy);
        int sampledDelta = 0;
        v3 bsdfWeight = sm_sample(bsdf, its.wi, sx, sy, &woL, &bsdfPdf, &bEta, &sampledDelta, sp);
        if (is_zero(bsdfWeight)) break;
        scattered |= sampledDelta != 2;                  /* path.cpp:213: scattered |= bRec.sampledType != BSDF::ENull */
        v3 wo = to_world(&its, woL);
        float woDotGeoN = dot(its.ng, wo);
        if (strict && woDotGeoN * woL.z <= 0) break;

        int hitEmitter = 0; v3 value = V(0, 0, 0);
        o = its.p; d = wo;
        ++counters[0];
        if (ray_intersect(s, o, d, EPSILON, INFINITY, &its, 0)) {      /* path.cpp:225-233 */
            if (its.emitter >= 0) {
                value = emitter_eval(s, its.emitter, its.ns, neg(d));
                dRec.p = its.p; dRec.n = its.ns; dRec.d = d; dRec.dist = its.t; dRec.emitter = its.emitter;   /* records.inl:181-189 setQuery */
                hitEmitter = 1;
            }
        } else {                                         /* path.cpp:234-248 */
            if (s->env_index < 0) break;
            if (hide && !scattered) break;
            value = env_eval(s, d);
            float nearT, farT;                           /* envmap.cpp:362-378 fillDirectSamplingRecord */
            if (!bsphere_intersect(s->env_bs_center, s->env_bs_radius, o, d, &nearT, &farT) || nearT > 0 || farT < 0) break;
            dRec.p = add(o, scale(d, farT)); dRec.n = normalize(sub(s->env_bs_center, dRec.p)); dRec.d = d; dRec.dist = farT; dRec.emitter = s->env_index;
            hitEmitter = 1;
        }

        throughput = mul(throughput, bsdfWeight); eta *= bEta;
        if (hitEmitter) {                                /* path.cpp:257-264 */
            float lumPdf = sampledDelta ? 0.0f : pdf_emitter_direct(s, &dRec, refN);     /* !(bRec.sampledType & BSDF::EDelta) (path.cpp:259-260) */
            Li = add(Li, scale(mul(throughput, value), mi_weight(bsdfPdf, lumPdf)));
        }
        if (!its.valid) break;                           /* path.cpp:272 */
        emitted_radiance = 0;                            /* rRec.type = ERadianceNoEmission */
        if (depth++ >= rrDepth) {                        /* path.cpp:276-286 */
            float q = minf(maxf(maxf(throughput.x, throughput.y), throughput.z) * eta * eta, 0.95f);
            if (next1D(sp) >= q) break;
            float r = 1.0f / q; throughput = scale(throughput, r);
        }
    }
    counters[2] += (uint64_t) depth;
    *out_depth = depth;
    return Li;
}

/* ------------------------------------------------------------------------------------------------ participating media */
/* include/mitsuba/core/math.h:185-195 (Linux x86_64): fastexp / fastlog go through the double-precision routines */
static inline float mi_fastexp(float v) { return (float) exp((double) v); }
static inline float mi_fastlog(float v) { return (float) log((double) v); }
typedef struct { float t; v3 p; v3 transmittance; float pdf_success, pdf_failure; } mrec_t;
static inline void medium_sigma_t(const orc_medium *m, float *st) { for (int i = 0; i < 3; ++i) st[i] = m->sigma_a[i] + m->sigma_s[i]; }     /* medium.cpp:36 */
/* HomogeneousMedium::evalTransmittance (homogeneous.cpp:266-273) over [mint, maxt] of a ray */
static v3 medium_transmittance(const orc_medium *m, float mint, float maxt) {
    float st[3], r[3]; medium_sigma_t(m, st); const float negLength = mint - maxt;
    for (int i = 0; i < 3; ++i) r[i] = st[i] != 0 ? mi_fastexp(st[i] * negLength) : 1.0f;
    return V(r[0], r[1], r[2]);
}
/* HomogeneousMedium::sampleDistance (homogeneous.cpp:275-349), strategies balance / single / manual */
static int medium_sample_distance(const orc_medium *m, v3 o, v3 d, float mint, float maxt, sampler_t *sp, mrec_t *r) {
    float st[3]; medium_sigma_t(m, st);
    float rnd = next1D(sp), sampled, density = m->sampling_density;
    if (rnd < m->medium_sampling_weight) {
        rnd /= m->medium_sampling_weight;
        if (m->strategy == 0) { int ch = (int) (next1D(sp) * 3); if (ch > 2) ch = 2; density = st[ch]; }
        sampled = -mi_fastlog(1 - rnd) / density;
    } else sampled = INFINITY;
    const float distSurf = maxt - mint; int success = 1;
    if (sampled < distSurf) {
        r->t = sampled + mint; r->p = add(o, scale(d, r->t));
        if (r->p.x == o.x && r->p.y == o.y && r->p.z == o.z) success = 0;
    } else { sampled = distSurf; success = 0; }
    if (m->strategy == 0) {
        r->pdf_failure = 0; r->pdf_success = 0;
        for (int i = 0; i < 3; ++i) { float tmp = mi_fastexp(-st[i] * sampled); r->pdf_failure += tmp; r->pdf_success += st[i] * tmp; }
        r->pdf_failure /= 3; r->pdf_success /= 3;
    } else { r->pdf_failure = mi_fastexp(-density * sampled); r->pdf_success = density * r->pdf_failure; }
    r->transmittance = V(mi_fastexp(st[0] * (-sampled)), mi_fastexp(st[1] * (-sampled)), mi_fastexp(st[2] * (-sampled)));
    r->pdf_success = r->pdf_success * m->medium_sampling_weight;
    r->pdf_failure = m->medium_sampling_weight * r->pdf_failure + (1 - m->medium_sampling_weight);
    if (maxf(maxf(r->transmittance.x, r->transmittance.y), r->transmittance.z) < 1e-20f) r->transmittance = V(0, 0, 0);
    return success;
}
/* IsotropicPhaseFunction::eval (isotropic.cpp:74-76), HGPhaseFunction::eval (hg.cpp:108-111); wi = -ray.d, wo = the sampled direction */
static float phase_eval(const orc_medium *m, v3 wi, v3 wo) {
    if (m->phase == 0) return INV_FOURPI;
    const float g = m->g, temp = 1.0f + g * g + 2.0f * g * dot(wi, wo);
    return INV_FOURPI * (1 - g * g) / (temp * sqrtf(temp));
}
/* IsotropicPhaseFunction::sample (isotropic.cpp:61-66), HGPhaseFunction::sample (hg.cpp:74-99): weight 1 */
static v3 phase_sample(const orc_medium *m, v3 wi, sampler_t *sp) {
    float sx, sy; next2D(sp, &sx, &sy);
    if (m->phase == 0) return uniform_sphere(sx, sy);
    const float g = m->g; float cosTheta;
    if (fabsf(g) < EPSILON) cosTheta = 1 - 2 * sx;
    else { float sqrTerm = (1 - g * g) / (1 - g + 2 * g * sx); cosTheta = (1 + g * g - sqrTerm * sqrTerm) / (2 * g); }
    float sinTheta = sqrtf(maxf(1.0f - cosTheta * cosTheta, 0.0f)), sinPhi, cosPhi;
    sincos_2pi(sy, &sinPhi, &cosPhi);
    v3 n = neg(wi), fs, ft; coordinate_system(n, &fs, &ft);           /* Frame(-pRec.wi).toWorld */
    v3 l = V(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
    return add(add(scale(fs, l.x), scale(ft, l.y)), scale(n, l.z));
}
static inline int shape_interior(const orc_scene *s, uint32_t shape) { return s->d.shape_media ? s->d.shape_media[shape * 2] : -1; }
static inline int shape_exterior(const orc_scene *s, uint32_t shape) { return s->d.shape_media ? s->d.shape_media[shape * 2 + 1] : -1; }
static inline int is_medium_transition(const orc_scene *s, uint32_t shape) { return shape_interior(s, shape) >= 0 || shape_exterior(s, shape) >= 0; }   /* shape.h isMediumTransition */
static inline int target_medium(const orc_scene *s, uint32_t shape, v3 n, v3 d) { return dot(d, n) > 0 ? shape_exterior(s, shape) : shape_interior(s, shape); }   /* records.inl:81-86 */
/* ShapeKDTree::rayIntersect(ray, t, shape, n, uv) (skdtree.cpp:144-205): closest hit under the shadow-ray epsilon rule, counted as a shadow ray */
static int ray_intersect_n(const orc_scene *s, v3 o, v3 d, float rmint, float rmaxt, hit_t *h, uint64_t *shadow_rays) {
    float mint, maxt, t = 0, u = 0, v = 0; uint32_t prim = 0; int32_t inst = -1;
    h->valid = 0; h->t = INFINITY; if (shadow_rays) ++*shadow_rays;
    if (!clip_interval(s, o, d, rmint, rmaxt, 1, &mint, &maxt)) return 0;
    int found = g_brute ? traverse_brute(s, o, d, mint, maxt, 0, &t, &prim, &inst, &u, &v) : traverse(s, o, d, mint, maxt, 0, &t, &prim, &inst, &u, &v);
    if (!found) return 0;
    fill_hit(s, o, d, t, prim, inst, u, v, h);
    return 1;
}
/* bsdf->eval(bRec, EDiscrete) with typeMask = ENull for a straight pass-through (wo = -wi): `null` -> 1 (null.cpp:48-50), `thindielectric` -> its transmittance with
 * the internal bounces summed (thindielectric.cpp:155-178); cosWi = Frame::cosTheta(bRec.wi).  has_null: the BSDF's type carries ENull at all */
static int leaf_has_null(const orc_material *m) { return m->type == BSDF_NULL || m->type == BSDF_THINDIELECTRIC; }
#define MIX_CHILD(m, c) ((uint32_t) ((c) < 3 ? (m)->reflectance[c] : (m)->eta[0]))
/* mask.cpp:107-108: always an ENull component; mixturebsdf.cpp:150-166: its children's components */
static int surface_has_null(const orc_scene *s, const orc_material *m) {
    if (m->type == BSDF_MIXTURE) { for (uint32_t c = 0; c < m->distr && c < 4; ++c) if (leaf_has_null(&s->materials[MIX_CHILD(m, c)].m)) return 1; return 0; }
    return leaf_has_null(m) || m->type == BSDF_MASK;
}
static v3 material_null_eval(const orc_material *m, float cosWi) {
    if (m->type == BSDF_NULL) return V(1, 1, 1);
    float ct, R = fresnel_dielectric_ext(fabsf(cosWi), &ct, m->eta[0]), T = 1 - R;
    if (R < 1) R += T * T * R / (1 - R * R);
    return scale(V(m->reflectance[0], m->reflectance[1], m->reflectance[2]), 1 - R);
}
/* the same at a hit: a `mask` answers 1 - opacity (mask.cpp:120-121), its opacity texture looked up at the hit's uv without differentials.  walk: the hit comes from
 * ShapeKDTree::rayIntersect(ray, t, shape, n, uv), which gives a scene-level triangle mesh WITHOUT texture coordinates uv = (0, 0) (skdtree.cpp:182-184), not the barycentrics */
static v3 surface_null_eval(const orc_scene *s, const hit_t *its, float cosWi, int walk) {
    const mat_t *mt = &s->materials[its->material];
    if (mt->m.type == BSDF_MIXTURE) {              /* MixtureBSDF::eval (mixturebsdf.cpp:171-183) under EDiscrete / typeMask = ENull: sum of weight x the children's pass-through values (0 for a child without an ENull lobe) */
        const smat_t sm = resolve_material(s, (uint32_t) its->material, NULL, 0, V(0, 0, 0), NULL, NULL); v3 r = V(0, 0, 0);
        for (int i = 0; i < sm.n_mix; ++i) if (leaf_has_null(&sm.mix[i].m)) r = add(r, scale(material_null_eval(&sm.mix[i].m, cosWi), sm.w[i])); else r = add(r, scale(V(0, 0, 0), sm.w[i]));
        return r;
    }
    if (mt->m.type != BSDF_MASK) return material_null_eval(&mt->m, cosWi);
    mat_t tmp = *mt; hit_t h = *its;
    if (walk && its->instance < 0 && its->prim < s->d.n_tris && !(s->uv && (s->shapes[its->shape].flags & 2u))) { h.uvx = 0; h.uvy = 0; }
    apply_texture(s, &tmp, &h, NULL, 0, V(0, 0, 0), NULL, NULL);
    return V(1.0f - tmp.m.reflectance[0], 1.0f - tmp.m.reflectance[1], 1.0f - tmp.m.reflectance[2]);
}
/* Scene::evalTransmittance (scene.cpp:650-713): walk from p1 to p2 through index-matched (`null`) boundaries, attenuating by the media in between */
static v3 eval_transmittance(const orc_scene *s, v3 p1, int p1OnSurface, v3 p2, int p2OnSurface, int medium, int *interactions, uint64_t *shadow_rays) {
    v3 d = sub(p2, p1); float remaining = length3(d); { float r = 1.0f / remaining; d = scale(d, r); }
    const float lengthFactor = p2OnSurface ? (1 - SHADOW_EPSILON) : 1;
    v3 o = p1; float mint = p1OnSurface ? EPSILON : 0, maxt = remaining * lengthFactor;
    v3 transmittance = V(1, 1, 1); hit_t its; const int maxInteractions = *interactions; *interactions = 0;
    while (remaining > 0) {
        int surface = ray_intersect_n(s, o, d, mint, maxt, &its, shadow_rays);
        if (surface && (*interactions == maxInteractions || !surface_has_null(s, &s->materials[its.material].m))) return V(0, 0, 0);     /* !(bsdf->getType() & BSDF::ENull) */
        if (medium >= 0) transmittance = mul(transmittance, medium_transmittance(&s->media[medium], 0, minf(its.t, remaining)));
        if (!surface || is_zero(transmittance)) break;
        /* its.geoFrame = Frame(n); wo = toLocal(ray.d); bRec(its, -wo, wo) with typeMask = ENull (scene.cpp:679-685): cosTheta(wi) = -dot(d, n) */
        transmittance = mul(transmittance, surface_null_eval(s, &its, -dot(d, its.ng_raw), 1));
        if (is_medium_transition(s, its.shape)) {
            if (medium != target_medium(s, its.shape, its.ng_raw, neg(d))) return V(0, 0, 0);        /* medium inconsistency */
            medium = target_medium(s, its.shape, its.ng_raw, d);
        }
        if (++*interactions > 100) break;
        o = add(o, scale(d, its.t)); remaining -= its.t; maxt = remaining * lengthFactor; mint = EPSILON;
    }
    return transmittance;
}
/* Scene::sampleAttenuatedEmitterDirect (scene.cpp:886-931), both overloads: `its` = NULL for a medium interaction */
static v3 sample_attenuated_emitter_direct(const orc_scene *s, v3 ref, v3 refN, const hit_t *its, int medium, int *interactions, float sx, float sy, direct_t *dr, uint64_t *shadow_rays) {
    v3 value = sample_emitter_direct(s, ref, refN, sx, sy, dr, 2, NULL);
    if (dr->pdf == 0) return V(0, 0, 0);
    if (its && is_medium_transition(s, its->shape)) medium = target_medium(s, its->shape, its->ng, dr->d);
    v3 tr = eval_transmittance(s, ref, its != NULL, dr->p, !dr->delta, medium, interactions, shadow_rays);
    float r = 1.0f / dr->em_pdf;
    return mul(value, scale(tr, r));                  /* value *= evalTransmittance(...) / emPdf */
}

/* RadianceQueryRecord::rayIntersect, EOpacity (records.inl:121-137) */
static float sensor_ray_alpha(const orc_scene *s, v3 o, v3 d, const hit_t *its, int medium, uint64_t *shadow_rays) {
    const int hit = its->valid;
    if (!s->d.opacity || (hit && !is_medium_transition(s, its->shape))) return 1.0f;
    if (!hit && medium < 0) return 0.0f;
    /* :131-134: 1 - average transmittance of the sensor's medium over twice the scene's bounding-sphere radius.  Scene::getBSphere: the kd-tree box expanded by the
     * sensor's and the point / spot emitters' positions (scene.cpp:394-421, aabb.cpp:44-47) */
    v3 lo = s->aabb_lo, hi = s->aabb_hi, c;
    v3 cam = V(s->d.cam_to_world[3], s->d.cam_to_world[7], s->d.cam_to_world[11]);
    lo = V(minf(lo.x, cam.x), minf(lo.y, cam.y), minf(lo.z, cam.z)); hi = V(maxf(hi.x, cam.x), maxf(hi.y, cam.y), maxf(hi.z, cam.z));
    for (uint32_t e = 0; e < s->d.n_emitters; ++e) if (s->emitters[e].type == 3 || s->emitters[e].type == 4 || s->emitters[e].type == 7) {
        v3 p = V(s->emitters[e].to_world[3], s->emitters[e].to_world[7], s->emitters[e].to_world[11]);
        lo = V(minf(lo.x, p.x), minf(lo.y, p.y), minf(lo.z, p.z)); hi = V(maxf(hi.x, p.x), maxf(hi.y, p.y), maxf(hi.z, p.z));
    }
    c = scale(add(hi, lo), 0.5f); float dist = length3(sub(c, hi)) * 2;
    v3 p2 = add(o, scale(d, dist)), tr;
    if (hit) {                                           /* :128-130: what lies behind the boundary, seen from the hit point */
        int unused = 0x7FFFFFFF;
        tr = eval_transmittance(s, its->p, 1, p2, 0, target_medium(s, its->shape, its->ng, d), &unused, shadow_rays);
    } else { v3 dd = sub(p2, o); tr = medium_transmittance(&s->media[medium], 0.0f, length3(dd)); }     /* (no surface on the way: the walk of :131-134 reduces to this) */
    return 1 - ((0.0f + tr.x) + tr.y + tr.z) * (1.0f / 3);
}
/* src/integrators/path/volpath_simple.cpp:84-289 SimpleVolumetricPathTracer::Li (no subsurface) */
static v3 volpath_simple_li(const orc_scene *s, v3 o, v3 d, float mint, float maxt, sampler_t *sp, int *out_depth, uint64_t *counters, float *alpha, const v3 *rxd, const v3 *ryd) {
    const int maxDepth = s->d.max_depth, rrDepth = s->d.rr_depth;
    const int strict = s->d.strict_normals != 0, hide = s->d.hide_emitters != 0;
    hit_t its; v3 Li = V(0, 0, 0); int depth = 1, medium = s->d.sensor_medium;
    int nullChain = 1, scattered = 0; float eta = 1.0f;
    ++counters[0];
    if (!ray_intersect(s, o, d, mint, maxt, &its, 0)) its.t = INFINITY;
    *alpha = sensor_ray_alpha(s, o, d, &its, medium, &counters[1]);
    v3 throughput = V(1, 1, 1);
    int emitted = 1, others = 1;                                  /* rRec.type: EEmittedRadiance / the bits of ERadianceNoEmission (they only ever change together) */
    if (maxDepth == 1) others = 0;
    int differentials = 1;                                        /* the sensor ray carries them; every later `ray = Ray(...)` does not */
    while (depth <= maxDepth || maxDepth < 0) {
        mrec_t mRec;
        if (medium >= 0 && medium_sample_distance(&s->media[medium], o, d, 0, its.t, sp, &mRec)) {
            const orc_medium *m = &s->media[medium];
            { float r = 1.0f / mRec.pdf_success; throughput = mul(throughput, scale(mul(V(m->sigma_s[0], m->sigma_s[1], m->sigma_s[2]), mRec.transmittance), r)); }
            if (others) {                                        /* EDirectMediumRadiance */
                direct_t dRec; memset(&dRec, 0, sizeof(dRec)); dRec.ref = mRec.p;
                int interactions = maxDepth - depth - 1; float sx, sy; next2D(sp, &sx, &sy);
                v3 value = sample_attenuated_emitter_direct(s, mRec.p, V(0, 0, 0), NULL, medium, &interactions, sx, sy, &dRec, &counters[1]);
                if (!is_zero(value)) Li = add(Li, scale(mul(throughput, value), phase_eval(m, neg(d), dRec.d)));
            }
            if ((depth + 1 >= maxDepth && maxDepth > 0) || !others) break;
            v3 wo = phase_sample(m, neg(d), sp);                /* weight 1 for both phase functions */
            o = mRec.p; d = wo; mint = 0; maxt = INFINITY; differentials = 0;
            ++counters[0];
            if (!ray_intersect(s, o, d, 0.0f, INFINITY, &its, 0)) its.t = INFINITY;
            nullChain = 0; scattered = 1;
        } else {
            if (medium >= 0) { float r = 1.0f / mRec.pdf_failure; throughput = mul(throughput, scale(mRec.transmittance, r)); }
            if (!its.valid) {
                if (s->env_index >= 0 && emitted && (!hide || scattered)) {
                    v3 value = mul(throughput, (s->d.env_texture && !s->env_constant && differentials) ? env_eval_filtered(s, d, *rxd, *ryd) : env_eval(s, d));
                    if (medium >= 0) value = mul(value, medium_transmittance(&s->media[medium], mint, maxt));
                    Li = add(Li, value);
                }
                break;
            }
            if (its.emitter >= 0 && emitted && (!hide || scattered))
                Li = add(Li, mul(throughput, emitter_eval(s, its.emitter, its.ns, neg(d))));
            if (strict && (-dot(its.ng, d)) * its.wi.z < 0) break;
            const smat_t smat = resolve_material(s, its.material, &its, differentials, o, rxd, ryd); const smat_t *bsdf = &smat;
            if (others && sm_is_smooth(bsdf)) {                  /* EDirectSurfaceRadiance */
                v3 refN = sm_has_backside(bsdf) ? V(0, 0, 0) : its.ns;
                direct_t dRec; memset(&dRec, 0, sizeof(dRec)); dRec.ref = its.p;
                int interactions = maxDepth - depth - 1; float sx, sy; next2D(sp, &sx, &sy);
                v3 value = sample_attenuated_emitter_direct(s, its.p, refN, &its, medium, &interactions, sx, sy, &dRec, &counters[1]);
                if (!is_zero(value)) {
                    v3 wo = to_local(&its, dRec.d);
                    if (!strict || dot(its.ng, dRec.d) * wo.z > 0) Li = add(Li, mul(mul(throughput, value), sm_eval(bsdf, its.wi, wo)));
                }
            }
            float bsdfPdf = 0, bEta = 1; v3 woL = V(0, 0, 0); float sx, sy; next2D(sp, &sx, &sy); int sampledDelta = 0;
            v3 bsdfVal = sm_sample(bsdf, its.wi, sx, sy, &woL, &bsdfPdf, &bEta, &sampledDelta, sp);
            if (is_zero(bsdfVal)) break;
            int rt_others = (depth + 1 < maxDepth || maxDepth < 0) && others, rt_emitted = 0;
            if ((depth < maxDepth || maxDepth < 0) && others && sampledDelta && (sampledDelta != 2 || nullChain)) { rt_emitted = 1; nullChain = 1; }
            else nullChain &= sampledDelta == 2;
            if (!rt_others && !rt_emitted) break;
            others = rt_others; emitted = rt_emitted;
            v3 wo = to_world(&its, woL);
            if (dot(its.ng, wo) * woL.z <= 0 && strict) break;
            throughput = mul(throughput, bsdfVal); eta *= bEta;
            if (is_medium_transition(s, its.shape)) medium = target_medium(s, its.shape, its.ng, wo);
            o = its.p; d = wo; mint = EPSILON; maxt = INFINITY; differentials = 0;
            ++counters[0];
            if (!ray_intersect(s, o, d, EPSILON, INFINITY, &its, 0)) its.t = INFINITY;
            scattered |= sampledDelta != 2;
        }
        if (depth++ >= rrDepth) {
            float q = minf(maxf(maxf(throughput.x, throughput.y), throughput.z) * eta * eta, 0.95f);
            if (next1D(sp) >= q) break;
            float r = 1.0f / q; throughput = scale(throughput, r);
        }
    }
    counters[2] += (uint64_t) depth;
    *out_depth = depth;
    return Li;
}

/* VolumetricPathTracer::rayIntersectAndLookForEmitter (src/integrators/path/volpath.cpp:370-431): the FIRST intersection of the ray goes to *first (the path continues
 * from it); the walk goes on through index-matched boundaries looking for an emitter and returns its attenuated emittance (no environment emitters in volumetric scenes here) */
static void look_for_emitter(const orc_scene *s, int medium, int maxInteractions, v3 o, v3 d, float mint, hit_t *first, direct_t *dRec, v3 *value, uint64_t *counters) {
    hit_t its2, *its = first; v3 transmittance = V(1, 1, 1); int surface = 0, interactions = 0;
    *value = V(0, 0, 0);
    while (1) {
        ++counters[0];
        surface = ray_intersect(s, o, d, mint, INFINITY, its, 0); if (!surface) its->t = INFINITY;
        if (medium >= 0) transmittance = mul(transmittance, medium_transmittance(&s->media[medium], 0, its->t));
        if (surface && (interactions == maxInteractions || !surface_has_null(s, &s->materials[its->material].m) || its->emitter >= 0)) break;
        if (!surface) break;
        if (is_zero(transmittance)) return;
        if (is_medium_transition(s, its->shape)) medium = target_medium(s, its->shape, its->ng, d);
        /* wo = its->shFrame.toLocal(ray.d); bRec(*its, -wo, wo), typeMask = ENull (volpath.cpp:399-402): cosTheta(wi) = -dot(d, ns) */
        transmittance = mul(transmittance, surface_null_eval(s, its, -dot(d, its->ns), 0));
        o = add(o, scale(d, its->t)); mint = EPSILON; its = &its2;
        if (++interactions > 100) return;
    }
    if (surface && its->emitter >= 0) {                  /* dRec.setQuery(ray, *its) (records.inl:170-178): dist = the LAST segment's t; ref / refN stay */
        dRec->p = its->p; dRec->n = its->ns; dRec->d = d; dRec->dist = its->t; dRec->emitter = its->emitter;
        *value = mul(transmittance, emitter_eval(s, its->emitter, its->ns, neg(d)));
    } else if (!surface && s->env_index >= 0) {          /* volpath.cpp:421-426: env->fillDirectSamplingRecord(dRec, ray) (envmap.cpp:362-378, from the ADVANCED origin), evalEnvironment without differentials */
        float nearT, farT;
        if (bsphere_intersect(s->env_bs_center, s->env_bs_radius, o, d, &nearT, &farT) && !(nearT > 0) && !(farT < 0)) {
            dRec->p = add(o, scale(d, farT)); dRec->n = normalize(sub(s->env_bs_center, dRec->p)); dRec->d = d; dRec->dist = farT; dRec->emitter = s->env_index;
            *value = mul(transmittance, env_eval(s, d));
        }
    }
}
static float phase_pdf(const orc_medium *m, v3 wi, v3 wo) { return phase_eval(m, wi, wo); }       /* PhaseFunction::pdf (src/librender/phase.cpp:21-23); isotropic: the same constant */

/* src/integrators/path/volpath.cpp:84-343 VolumetricPathTracer::Li (no subsurface, no environment emitter) */
static v3 volpath_li(const orc_scene *s, v3 o, v3 d, float mint, float maxt, sampler_t *sp, int *out_depth, uint64_t *counters, float *alpha, const v3 *rxd, const v3 *ryd) {
    const int maxDepth = s->d.max_depth, rrDepth = s->d.rr_depth;
    const int strict = s->d.strict_normals != 0, hide = s->d.hide_emitters != 0;
    hit_t its; v3 Li = V(0, 0, 0); int depth = 1, medium = s->d.sensor_medium, scattered = 0; float eta = 1.0f;
    ++counters[0];
    if (!ray_intersect(s, o, d, mint, maxt, &its, 0)) its.t = INFINITY;
    *alpha = sensor_ray_alpha(s, o, d, &its, medium, &counters[1]);
    v3 throughput = V(1, 1, 1); int emitted = 1;                  /* rRec.type is ERadiance or ERadianceNoEmission throughout */
    int differentials = 1;
    while (depth <= maxDepth || maxDepth < 0) {
        mrec_t mRec;
        if (medium >= 0 && medium_sample_distance(&s->media[medium], o, d, 0, its.t, sp, &mRec)) {
            const orc_medium *m = &s->media[medium];
            if (depth >= maxDepth && maxDepth != -1) break;
            { float r = 1.0f / mRec.pdf_success; throughput = mul(throughput, scale(mul(V(m->sigma_s[0], m->sigma_s[1], m->sigma_s[2]), mRec.transmittance), r)); }
            direct_t dRec; memset(&dRec, 0, sizeof(dRec)); dRec.ref = mRec.p;
            {   int interactions = maxDepth - depth - 1; float sx, sy; next2D(sp, &sx, &sy);
                v3 value = sample_attenuated_emitter_direct(s, mRec.p, V(0, 0, 0), NULL, medium, &interactions, sx, sy, &dRec, &counters[1]);
                if (!is_zero(value)) {
                    float phaseVal = phase_eval(m, neg(d), dRec.d);
                    if (phaseVal != 0) {
                        float phasePdf = dRec.delta ? 0.0f : phase_pdf(m, neg(d), dRec.d);
                        Li = add(Li, scale(scale(mul(throughput, value), phaseVal), mi_weight(dRec.pdf, phasePdf)));
                    }
                }
            }
            v3 wi = neg(d); v3 wo = phase_sample(m, wi, sp); float phasePdf = phase_pdf(m, wi, wo);        /* isotropic.cpp:68-72, hg.cpp:101-106: weight 1 */
            o = mRec.p; d = wo; mint = 0; maxt = INFINITY; differentials = 0;
            v3 value;
            look_for_emitter(s, medium, maxDepth - depth - 1, o, d, 0.0f, &its, &dRec, &value, counters);
            if (!is_zero(value)) {
                float emitterPdf = pdf_emitter_direct(s, &dRec, V(0, 0, 0));
                Li = add(Li, scale(mul(throughput, value), mi_weight(phasePdf, emitterPdf)));
            }
            emitted = 0;
        } else {
            if (medium >= 0) { float r = 1.0f / mRec.pdf_failure; throughput = mul(throughput, scale(mRec.transmittance, r)); }
            if (!its.valid) {                                     /* volpath.cpp:181-192 */
                if (s->env_index >= 0 && emitted && (!hide || scattered)) {
                    v3 value = mul(throughput, (s->d.env_texture && !s->env_constant && differentials) ? env_eval_filtered(s, d, *rxd, *ryd) : env_eval(s, d));
                    if (medium >= 0) value = mul(value, medium_transmittance(&s->media[medium], mint, maxt));
                    Li = add(Li, value);
                }
                break;
            }
            if (its.emitter >= 0 && emitted && (!hide || scattered))
                Li = add(Li, mul(throughput, emitter_eval(s, its.emitter, its.ns, neg(d))));
            if (depth >= maxDepth && maxDepth != -1) break;
            if ((-dot(its.ng, d)) * its.wi.z < 0 && strict) break;
            const smat_t smat = resolve_material(s, its.material, &its, differentials, o, rxd, ryd); const smat_t *bsdf = &smat;
            v3 refN = sm_has_backside(bsdf) ? V(0, 0, 0) : its.ns;
            direct_t dRec; memset(&dRec, 0, sizeof(dRec)); dRec.ref = its.p;
            if (sm_is_smooth(bsdf)) {
                int interactions = maxDepth - depth - 1; float sx, sy; next2D(sp, &sx, &sy);
                v3 value = sample_attenuated_emitter_direct(s, its.p, refN, &its, medium, &interactions, sx, sy, &dRec, &counters[1]);
                if (!is_zero(value)) {
                    v3 wo = to_local(&its, dRec.d); v3 bsdfVal = sm_eval(bsdf, its.wi, wo);
                    if (!is_zero(bsdfVal) && (!strict || dot(its.ng, dRec.d) * wo.z > 0)) {
                        float bsdfPdf = dRec.delta ? 0.0f : sm_pdf(bsdf, its.wi, wo);
                        Li = add(Li, scale(mul(mul(throughput, value), bsdfVal), mi_weight(dRec.pdf, bsdfPdf)));
                    }
                }
            }
            float bsdfPdf = 0, bEta = 1; v3 woL = V(0, 0, 0); float sx, sy; next2D(sp, &sx, &sy); int sampledDelta = 0;
            v3 bsdfWeight = sm_sample(bsdf, its.wi, sx, sy, &woL, &bsdfPdf, &bEta, &sampledDelta, sp);
            if (is_zero(bsdfWeight)) break;
            v3 wo = to_world(&its, woL);
            if (dot(its.ng, wo) * woL.z <= 0 && strict) break;
            v3 po = its.p;
            throughput = mul(throughput, bsdfWeight); eta *= bEta;
            if (is_medium_transition(s, its.shape)) medium = target_medium(s, its.shape, its.ng, wo);
            o = po; d = wo; mint = EPSILON; maxt = INFINITY; differentials = 0;
            if (sampledDelta == 2) {                              /* index-matched boundary: no emitter search, no Russian roulette, `scattered` unchanged */
                emitted = scattered ? 0 : 1;
                ++counters[0];
                if (!ray_intersect(s, o, d, EPSILON, INFINITY, &its, 0)) its.t = INFINITY;
                depth++;
                continue;
            }
            v3 value;
            look_for_emitter(s, medium, maxDepth - depth - 1, o, d, EPSILON, &its, &dRec, &value, counters);
            if (!is_zero(value)) {
                float emitterPdf = sampledDelta ? 0.0f : pdf_emitter_direct(s, &dRec, refN);
                Li = add(Li, scale(mul(throughput, value), mi_weight(bsdfPdf, emitterPdf)));
            }
            emitted = 0;
        }
        if (depth++ >= rrDepth) {
            float q = minf(maxf(maxf(throughput.x, throughput.y), throughput.z) * eta * eta, 0.95f);
            if (next1D(sp) >= q) break;
            float r = 1.0f / q; throughput = scale(throughput, r);
        }
        scattered = 1;
    }
    counters[2] += (uint64_t) depth;
    *out_depth = depth;
    return Li;
}

/* one pixel sample: src/librender/integrator.cpp:171-186 (renderBlock body) */
static v3 pixel_sample(const orc_scene *s, uint32_t px, uint32_t py, uint64_t sidx, float *pos, int *depth, uint64_t *counters, float *log, int *nlog, float *alpha) {
    sampler_t sp; sampler_begin(&sp, s, px, py, sidx, log);
    float jx, jy; next2D(&sp, &jx, &jy);
    pos[0] = (float) (int32_t) px + jx; pos[1] = (float) (int32_t) py + jy;
    v3 o, d; float mint, maxt; camera_ray(s, pos[0], pos[1], &o, &d, &mint, &maxt);
    v3 rxd, ryd; camera_differentials(s, pos[0], pos[1], d, &rxd, &ryd);
    v3 li = s->d.integrator == 2 ? volpath_li(s, o, d, mint, maxt, &sp, depth, counters, alpha, &rxd, &ryd) : s->d.integrator == 1 ? volpath_simple_li(s, o, d, mint, maxt, &sp, depth, counters, alpha, &rxd, &ryd) : path_li(s, o, d, mint, maxt, &sp, depth, counters, alpha, &rxd, &ryd);
    if (nlog) *nlog = sp.nlog;
    return li;
}
void orc_render_samples(const orc_scene *s, const uint32_t *pairs, uint64_t n, float *out_li, float *out_pos, int32_t *out_depth, int32_t *out_nvals, float *out_vals) {
    uint64_t counters[3] = {0, 0, 0};
    for (uint64_t i = 0; i < n; ++i) {
        int depth = 0, nlog = 0; float pos[2], alpha;
        if (out_vals) for (int k = 0; k < 64; ++k) out_vals[i * 64 + k] = -1.0f;
        v3 li = pixel_sample(s, pairs[i * 3], pairs[i * 3 + 1], pairs[i * 3 + 2], pos, &depth, counters, out_vals ? &out_vals[i * 64] : NULL, &nlog, &alpha);
        out_li[i * 3] = li.x; out_li[i * 3 + 1] = li.y; out_li[i * 3 + 2] = li.z;
        if (out_pos) { out_pos[i * 2] = pos[0]; out_pos[i * 2 + 1] = pos[1]; }
        if (out_depth) out_depth[i] = depth;
        if (out_nvals) out_nvals[i] = nlog;
    }
}

/* ------------------------------------------------------------------------------------------------ film */
/* include/mitsuba/core/rfilter.h:76-77 evalDiscretized */
float orc_filter_eval_discretized(const orc_scene *s, float x) {
    int i = (int) fabsf(x * s->filter_scale); if (i > FILTER_RES) i = FILTER_RES; return s->filter_values[i];
}
void orc_filter_table(const orc_scene *s, float *o, float *radius, int *border) { memcpy(o, s->filter_values, sizeof(s->filter_values)); *radius = s->filter_radius; *border = s->border; }
int orc_film_border(const orc_scene *s) { return s->border; }
/* include/mitsuba/render/imageblock.h:161-221 ImageBlock::put for a film-sized block at offset 0, 5 channels (RGB, alpha, weight) */
static void film_put(const orc_scene *s, float *film, float sxp, float syp, v3 value, float alpha) {
    float vals[5] = {value.x, value.y, value.z, alpha, 1.0f};
    for (int i = 0; i < 5; ++i) if (!isfinite(vals[i]) || vals[i] < 0) return;
    const int W = (int) s->d.width + 2 * s->border, H = (int) s->d.height + 2 * s->border;
    const float r = s->filter_radius;
    float posx = sxp - 0.5f - (float) (0 - s->border), posy = syp - 0.5f - (float) (0 - s->border);
    int minx = (int) ceilf(posx - r), miny = (int) ceilf(posy - r), maxx = (int) floorf(posx + r), maxy = (int) floorf(posy + r);
    if (minx < 0) minx = 0;
    if (miny < 0) miny = 0;
    if (maxx > W - 1) maxx = W - 1;
    if (maxy > H - 1) maxy = H - 1;
    for (int y = miny; y <= maxy; ++y) {
        float wy = orc_filter_eval_discretized(s, (float) y - posy);
        for (int x = minx; x <= maxx; ++x) {
            float w = orc_filter_eval_discretized(s, (float) x - posx) * wy;
            float *dst = film + ((size_t) y * W + x) * 5;
            for (int k = 0; k < 5; ++k) dst[k] += w * vals[k];
        }
    }
}
typedef struct { const orc_scene *s; uint32_t s0, s1, y0, y1; float *film; uint64_t counters[3]; int tid, nthreads; pthread_mutex_t *mtx; } job_t;
static void *image_worker(void *arg) {
    job_t *j = (job_t *) arg; const orc_scene *s = j->s;
    /* each worker owns whole pixel rows; splats of neighbouring rows overlap only inside the filter footprint -> lock per put
       for footprints > 1 pixel is avoided by giving every worker a private film that is summed at the end */
    for (uint32_t y = j->y0 + (uint32_t) j->tid; y < j->y1; y += (uint32_t) j->nthreads)
        for (uint32_t x = 0; x < s->d.width; ++x)
            for (uint32_t k = j->s0; k < j->s1; ++k) {
                float pos[2], alpha; int depth;
                v3 li = pixel_sample(s, x, y, k, pos, &depth, j->counters, NULL, NULL, &alpha);
                film_put(s, j->film, pos[0], pos[1], li, alpha);
            }
    return NULL;
}
void orc_render_image(const orc_scene *s, uint32_t s0, uint32_t s1, uint32_t y0, uint32_t y1, int n_threads, float *film, uint64_t *counters) {
    size_t n = (size_t) (s->d.width + 2 * s->border) * (s->d.height + 2 * s->border) * 5;
    if (n_threads < 1) n_threads = 1;
    job_t *jobs = (job_t *) calloc((size_t) n_threads, sizeof(job_t)); pthread_t *th = (pthread_t *) calloc((size_t) n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; ++t) {
        jobs[t].s = s; jobs[t].s0 = s0; jobs[t].s1 = s1; jobs[t].y0 = y0; jobs[t].y1 = y1; jobs[t].tid = t; jobs[t].nthreads = n_threads;
        jobs[t].film = t == 0 ? film : (float *) calloc(n, sizeof(float));
        if (t > 0) pthread_create(&th[t], NULL, image_worker, &jobs[t]);
    }
    image_worker(&jobs[0]);
    counters[0] = counters[1] = counters[2] = 0;
    for (int t = 0; t < n_threads; ++t) {
        if (t > 0) { pthread_join(th[t], NULL); for (size_t i = 0; i < n; ++i) film[i] += jobs[t].film[i]; free(jobs[t].film); }
        for (int c = 0; c < 3; ++c) counters[c] += jobs[t].counters[c];
    }
    free(jobs); free(th);
}

/* ------------------------------------------------------------------------------------------------ scene setup */
static const v3 *g_sort_cent; static int g_sort_axis;      /* scene construction is single-threaded */
static int cmp_centroid(const void *a, const void *b) {
    float x = comp(g_sort_cent[*(const uint32_t *) a], g_sort_axis), y = comp(g_sort_cent[*(const uint32_t *) b], g_sort_axis);
    if (x < y) return -1; if (x > y) return 1;
    return *(const uint32_t *) a < *(const uint32_t *) b ? -1 : 1;
}
static int build_bvh(orc_scene *s, uint32_t *tris, int first, int count, v3 *cent, v3 *tlo, v3 *thi) {
    int id = s->n_nodes++; bvh_node *n = &s->nodes[id];
    v3 lo = V(INFINITY, INFINITY, INFINITY), hi = V(-INFINITY, -INFINITY, -INFINITY), clo = lo, chi = hi;
    for (int i = first; i < first + count; ++i) {
        uint32_t t = tris[i];
        lo = V(minf(lo.x, tlo[t].x), minf(lo.y, tlo[t].y), minf(lo.z, tlo[t].z)); hi = V(maxf(hi.x, thi[t].x), maxf(hi.y, thi[t].y), maxf(hi.z, thi[t].z));
        clo = V(minf(clo.x, cent[t].x), minf(clo.y, cent[t].y), minf(clo.z, cent[t].z)); chi = V(maxf(chi.x, cent[t].x), maxf(chi.y, cent[t].y), maxf(chi.z, cent[t].z));
    }
    /* pad: the TriAccel test is not watertight w.r.t. the exact triangle, so boxes must only ever over-approximate */
    v3 ext = sub(hi, lo); float pad = 1e-4f * maxf(maxf(ext.x, ext.y), ext.z) + 1e-4f * maxf(maxf(fabsf(lo.x) + fabsf(hi.x), fabsf(lo.y) + fabsf(hi.y)), fabsf(lo.z) + fabsf(hi.z)) + 1e-6f;
    n->lo = V(lo.x - pad, lo.y - pad, lo.z - pad); n->hi = V(hi.x + pad, hi.y + pad, hi.z + pad);
    if (count <= 4) { n->first = first; n->count = count; n->left = n->right = -1; return id; }
    v3 ce = sub(chi, clo); int axis = ce.x > ce.y ? (ce.x > ce.z ? 0 : 2) : (ce.y > ce.z ? 1 : 2);
    /* median split by sorting on the centroid coordinate */
    g_sort_cent = cent; g_sort_axis = axis;
    qsort(tris + first, (size_t) count, sizeof(uint32_t), cmp_centroid);
    int half = count / 2;
    n->count = 0; n->first = 0;
    int l = build_bvh(s, tris, first, half, cent, tlo, thi);
    int r = build_bvh(s, tris, first + half, count - half, cent, tlo, thi);
    s->nodes[id].left = l; s->nodes[id].right = r;
    return id;
}
static void *dup(const void *p, size_t n) { if (!p) return NULL; void *q = malloc(n ? n : 1); memcpy(q, p, n); return q; }

orc_scene *orc_scene_create(const orc_scene_desc *d) {
    for (uint32_t i = 0; i < d->n_emitters; ++i) if (d->emitters[i].type > 7 || d->emitters[i].type == 6) return NULL;      /* 6 = the reference's compound `sunsky`: not restated; 7 = collimated */
    if (d->integrator != 0) for (uint32_t i = 0; i < d->n_materials; ++i) {                      /* volumetric walks see through plain records, masks and mixtures (surface_has_null), not through bumpmap / normalmap */
        const orc_material *m = &d->materials[i]; int bad = 0;
        #define ORC_NULL_LOBE(j) ((j) < d->n_materials && (d->materials[j].type == BSDF_NULL || d->materials[j].type == BSDF_THINDIELECTRIC))
        if ((m->type == BSDF_BUMPMAP || m->type == BSDF_NORMALMAP) && m->distr < d->n_materials) {
            const orc_material *nm = &d->materials[m->distr]; bad |= ORC_NULL_LOBE(m->distr);
            if (nm->type == BSDF_MIXTURE) for (uint32_t c = 0; c < nm->distr && c < 4; ++c) bad |= ORC_NULL_LOBE(MIX_CHILD(nm, c));
        }
        #undef ORC_NULL_LOBE
        if (bad) return NULL;
    }
    orc_scene *s = (orc_scene *) calloc(1, sizeof(orc_scene));
    s->d = *d;
    s->pos = (float *) dup(d->pos, (size_t) d->n_verts * 12); s->nrm = (float *) dup(d->nrm, (size_t) d->n_verts * 12);
    s->idx = (uint32_t *) dup(d->idx, (size_t) d->n_tris * 12);
    s->shapes = (orc_shape *) dup(d->shapes, d->n_shapes * sizeof(orc_shape));
    s->material_tables = (float *) dup(d->material_tables, (size_t) (d->material_tables ? d->n_material_tables : 0) * 4);
    s->materials = (mat_t *) calloc(d->n_materials ? d->n_materials : 1, sizeof(mat_t));
    for (uint32_t i = 0; i < d->n_materials; ++i) { s->materials[i].m = d->materials[i]; s->materials[i].table = s->material_tables ? s->material_tables + (size_t) d->materials[i].k[1] : NULL; }
    s->d.material_tables = NULL;
    if (d->integrator == 1) for (uint32_t i = 0; i < d->n_materials; ++i) if (s->materials[i].m.type == BSDF_THINDIELECTRIC) s->materials[i].m.flags |= THIN_SIGNED_COS;
    s->emitters = (orc_emitter *) dup(d->emitters, d->n_emitters * sizeof(orc_emitter));
    s->media = (orc_medium *) dup(d->media, (size_t) (d->media ? d->n_media : 0) * sizeof(orc_medium)); s->d.media = NULL;
    s->d.shape_media = (const int32_t *) dup(d->shape_media, (size_t) (d->shape_media ? (d->n_shapes + d->n_analytic) : 0) * 8);
    if (!d->n_media) s->d.sensor_medium = -1;
    s->uv = (float *) dup(d->uv, (size_t) d->n_verts * 8);
    s->textures = (orc_texture *) dup(d->textures, (size_t) (d->textures ? d->n_textures : 0) * sizeof(orc_texture)); s->d.textures = NULL;
    for (uint32_t i = 0; i < d->n_materials; ++i) {       /* Texture::getAverage(): checkerboard.cpp:102-104, gridtexture.cpp:116-121, bitmap.cpp:504-514 (the bitmap's average is input: color0) */
        orc_material *m = &s->materials[i].m; if (m->type != BSDF_PLASTIC && m->type != BSDF_ROUGHPLASTIC) continue;
        v3 dAvg = V(m->reflectance[0], m->reflectance[1], m->reflectance[2]); const uint32_t tex = (m->flags >> 8) & 0xFFFFu;
        if (tex && tex <= d->n_textures) {
            const orc_texture *t = &s->textures[tex - 1]; v3 c0 = V(t->color0[0], t->color0[1], t->color0[2]), c1 = V(t->color1[0], t->color1[1], t->color1[2]);
            if (t->type == 0) dAvg = scale(add(c0, c1), 0.5f);
            else if (t->type == 1) { float iw = maxf(0.0f, 1 - 2 * t->line_width), ia = iw * iw, la = 1 - ia; dAvg = add(scale(c1, la), scale(c0, ia)); }
            else dAvg = c0;
        }
        float dl = luminance(dAvg), sl = luminance(V(m->specular[0], m->specular[1], m->specular[2]));
        m->eta[1] = sl / (dl + sl);
    }
    s->tex_levels = (uint32_t *) dup(d->texture_levels, (size_t) (d->texture_levels ? d->n_texture_levels : 0) * 12);
    s->tex_texels = (float *) dup(d->texture_texels, (size_t) (d->texture_texels ? d->n_texture_texels : 0) * 4);
    s->d.texture_levels = NULL; s->d.texture_texels = NULL;
    for (int i = 0; i < 64; ++i) { float r2 = (float) i / 63.0f; s->mip_lut[i] = fastexpf_(-2.0f * r2) - fastexpf_(-2.0f); }   /* mipmap.h:297-302 */
    {   /* PerspectiveCameraImpl::configure (perspective.cpp:159-163): m_dx / m_dy = sampleToCamera(1/w, 0, 0) - sampleToCamera(0), w-divided points */
        const float *m = d->sample_to_camera; float irx = 1.0f / (float) d->width, iry = 1.0f / (float) d->height;
        v3 p0, px, py; float w;
        w = m[15]; p0 = V(m[3], m[7], m[11]); if (w != 1.0f) { float r = 1.0f / w; p0 = scale(p0, r); }
        w = m[12] * irx + m[13] * 0.0f + m[14] * 0.0f + m[15]; px = V(m[0] * irx + m[1] * 0.0f + m[2] * 0.0f + m[3], m[4] * irx + m[5] * 0.0f + m[6] * 0.0f + m[7], m[8] * irx + m[9] * 0.0f + m[10] * 0.0f + m[11]); if (w != 1.0f) { float r = 1.0f / w; px = scale(px, r); }
        w = m[12] * 0.0f + m[13] * iry + m[14] * 0.0f + m[15]; py = V(m[0] * 0.0f + m[1] * iry + m[2] * 0.0f + m[3], m[4] * 0.0f + m[5] * iry + m[6] * 0.0f + m[7], m[8] * 0.0f + m[9] * iry + m[10] * 0.0f + m[11]); if (w != 1.0f) { float r = 1.0f / w; py = scale(py, r); }
        s->cam_dx = sub(px, p0); s->cam_dy = sub(py, p0);
    }
    s->tri_shape = (uint32_t *) calloc(d->n_tris, 4);
    for (uint32_t i = 0; i < d->n_shapes; ++i) for (uint32_t t = 0; t < s->shapes[i].tri_count; ++t) s->tri_shape[s->shapes[i].first_tri + t] = i;
    /* TriAccel table (skdtree.cpp:79-105) + scene box (union of mesh AABBs, enlarged as in gkdtree.h:1213-1220) */
    s->accel = (triaccel *) calloc(d->n_tris ? d->n_tris : 1, sizeof(triaccel));
    /* TriMesh::computeUVTangents (src/librender/trimesh.cpp:683-736) for meshes with texcoords */
    s->tangents = NULL;
    if (s->uv) {
        s->tangents = (float *) calloc((size_t) (d->n_tris ? d->n_tris : 1) * 6, 4);
        for (uint32_t t = 0; t < d->n_tris; ++t) {
            if (!(s->shapes[s->tri_shape[t]].flags & 2u)) continue;
            uint32_t i0 = s->idx[t * 3], i1 = s->idx[t * 3 + 1], i2 = s->idx[t * 3 + 2];
            v3 dP1 = sub(vert(s, i1), vert(s, i0)), dP2 = sub(vert(s, i2), vert(s, i0));
            float du1 = s->uv[i1 * 2] - s->uv[i0 * 2], dv1 = s->uv[i1 * 2 + 1] - s->uv[i0 * 2 + 1], du2 = s->uv[i2 * 2] - s->uv[i0 * 2], dv2 = s->uv[i2 * 2 + 1] - s->uv[i0 * 2 + 1];
            v3 n = cross(dP1, dP2); float length = sqrtf(dot(n, n)); v3 dpdu, dpdv;
            if (length == 0) continue;
            float determinant = du1 * dv2 - dv1 * du2;
            if (determinant == 0) { float r = 1.0f / length; coordinate_system(scale(n, r), &dpdu, &dpdv); }
            else {
                float invDet = 1.0f / determinant;
                dpdu = scale(sub(scale(dP1, dv2), scale(dP2, dv1)), invDet);
                dpdv = scale(add(scale(dP1, -du2), scale(dP2, du1)), invDet);
            }
            float *o = &s->tangents[t * 6]; o[0] = dpdu.x; o[1] = dpdu.y; o[2] = dpdu.z; o[3] = dpdv.x; o[4] = dpdv.y; o[5] = dpdv.z;
        }
    }
    s->n_analytic = d->analytic ? d->n_analytic : 0; s->n_instances = d->instances ? d->n_instances : 0;
    s->n_prims = d->n_tris + s->n_analytic + s->n_instances;
    s->analytic = (analytic_t *) calloc(s->n_analytic ? s->n_analytic : 1, sizeof(analytic_t));
    for (uint32_t i = 0; i < s->n_analytic; ++i) { s->analytic[i].a = d->analytic[i]; analytic_prepare(&s->analytic[i]); }
    s->instances = (orc_instance *) dup(d->instances, s->n_instances * sizeof(orc_instance));
    s->d.analytic = NULL; s->d.instances = NULL;
    s->n_groups = 0;
    for (uint32_t i = 0; i < d->n_shapes; ++i) if (s->shapes[i].group > s->n_groups) s->n_groups = s->shapes[i].group;
    const uint32_t ng = s->n_groups;                                 /* table slot ng = the scene level */
    s->group_root = (int *) calloc(ng + 1, sizeof(int)); s->group_lo = (v3 *) calloc(ng + 1, sizeof(v3)); s->group_hi = (v3 *) calloc(ng + 1, sizeof(v3));
    s->group_first = (uint32_t *) calloc(ng + 1, 4); s->group_count = (uint32_t *) calloc(ng + 1, 4);
    v3 *cent = (v3 *) calloc(s->n_prims ? s->n_prims : 1, sizeof(v3)), *tlo = (v3 *) calloc(s->n_prims ? s->n_prims : 1, sizeof(v3)), *thi = (v3 *) calloc(s->n_prims ? s->n_prims : 1, sizeof(v3));
    for (uint32_t t = 0; t < d->n_tris; ++t) {
        v3 a = vert(s, s->idx[t * 3]), b = vert(s, s->idx[t * 3 + 1]), c = vert(s, s->idx[t * 3 + 2]);
        triaccel_load(&s->accel[t], a, b, c);
        tlo[t] = V(minf(minf(a.x, b.x), c.x), minf(minf(a.y, b.y), c.y), minf(minf(a.z, b.z), c.z));
        thi[t] = V(maxf(maxf(a.x, b.x), c.x), maxf(maxf(a.y, b.y), c.y), maxf(maxf(a.z, b.z), c.z));
        cent[t] = scale(add(tlo[t], thi[t]), 0.5f);
    }
    /* kd-tree boxes: per shape group and for the scene = union of the member shapes' AABBs (ShapeKDTree::addShape, skdtree.cpp:68-77),
     * enlarged as in gkdtree.h:1213-1220 */
    for (uint32_t g = 0; g <= ng; ++g) { s->group_lo[g] = V(INFINITY, INFINITY, INFINITY); s->group_hi[g] = V(-INFINITY, -INFINITY, -INFINITY); }
#define GROW(g, q) do { v3 q_ = (q); s->group_lo[g] = V(minf(s->group_lo[g].x, q_.x), minf(s->group_lo[g].y, q_.y), minf(s->group_lo[g].z, q_.z)); s->group_hi[g] = V(maxf(s->group_hi[g].x, q_.x), maxf(s->group_hi[g].y, q_.y), maxf(s->group_hi[g].z, q_.z)); } while (0)
#define ENLARGE(g) do { const float eps = KD_AABB_EPSILON; v3 lo_ = s->group_lo[g], hi_ = s->group_hi[g]; \
      v3 e1 = sub(hi_, lo_); lo_ = sub(lo_, V(e1.x * eps + eps, e1.y * eps + eps, e1.z * eps + eps)); \
      v3 e2 = sub(hi_, lo_); hi_ = add(hi_, V(e2.x * eps + eps, e2.y * eps + eps, e2.z * eps + eps)); s->group_lo[g] = lo_; s->group_hi[g] = hi_; } while (0)
    for (uint32_t i = 0; i < d->n_shapes; ++i) { uint32_t g = s->shapes[i].group ? s->shapes[i].group - 1 : ng; for (uint32_t v = 0; v < s->shapes[i].vert_count; ++v) GROW(g, vert(s, s->shapes[i].first_vert + v)); }
    for (uint32_t g = 0; g < ng; ++g) ENLARGE(g);
    for (uint32_t i = 0; i < s->n_analytic; ++i) {
        const analytic_t *a = &s->analytic[i]; uint32_t t = d->n_tris + i;
        GROW(ng, a->lo); GROW(ng, a->hi);
        tlo[t] = a->lo; thi[t] = a->hi;
        if (a->a.type == SH_DISK) {      /* Disk::getAABB bounds four rim points only; the oracle's own BVH needs the whole rim */
            v3 c = xf_point(a->a.to_world, V(0, 0, 0)); float r = length3(xf_vector(a->a.to_world, V(1, 0, 0)));
            tlo[t] = V(minf(tlo[t].x, c.x - r), minf(tlo[t].y, c.y - r), minf(tlo[t].z, c.z - r)); thi[t] = V(maxf(thi[t].x, c.x + r), maxf(thi[t].y, c.y + r), maxf(thi[t].z, c.z + r));
        }
        cent[t] = scale(add(tlo[t], thi[t]), 0.5f);
    }
    for (uint32_t i = 0; i < s->n_instances; ++i) {                  /* Instance::getAABB (instance.cpp:46-64): the 8 corners of the group's kd-tree box, transformed */
        const orc_instance *in = &s->instances[i]; uint32_t t = d->n_tris + s->n_analytic + i, g = in->group;
        v3 blo = V(INFINITY, INFINITY, INFINITY), bhi = V(-INFINITY, -INFINITY, -INFINITY);
        for (int c = 0; c < 8; ++c) {
            v3 q = xf_point(in->to_world, V(c & 1 ? s->group_hi[g].x : s->group_lo[g].x, c & 2 ? s->group_hi[g].y : s->group_lo[g].y, c & 4 ? s->group_hi[g].z : s->group_lo[g].z));
            blo = V(minf(blo.x, q.x), minf(blo.y, q.y), minf(blo.z, q.z)); bhi = V(maxf(bhi.x, q.x), maxf(bhi.y, q.y), maxf(bhi.z, q.z));
        }
        GROW(ng, blo); GROW(ng, bhi); tlo[t] = blo; thi[t] = bhi; cent[t] = scale(add(blo, bhi), 0.5f);
    }
    ENLARGE(ng);
#undef GROW
#undef ENLARGE
    s->aabb_lo = s->group_lo[ng]; s->aabb_hi = s->group_hi[ng];
    /* the oracle's own BVHs: one per shape group plus the scene level, over disjoint segments of bvh_tris */
    s->nodes = (bvh_node *) calloc(2 * (size_t) s->n_prims + 4 * (size_t) (ng + 2), sizeof(bvh_node));
    s->bvh_tris = (uint32_t *) calloc(s->n_prims + 1, 4); s->group_prims = s->bvh_tris;
    uint32_t fill = 0;
    for (uint32_t g = 0; g <= ng; ++g) {
        s->group_first[g] = fill;
        for (uint32_t i = 0; i < d->n_shapes; ++i) if ((s->shapes[i].group ? s->shapes[i].group - 1 : ng) == g)
            for (uint32_t t = 0; t < s->shapes[i].tri_count; ++t) s->bvh_tris[fill++] = s->shapes[i].first_tri + t;
        if (g == ng) for (uint32_t t = d->n_tris; t < s->n_prims; ++t) s->bvh_tris[fill++] = t;
        s->group_count[g] = fill - s->group_first[g];
    }
    s->n_nodes = 0;
    for (uint32_t g = 0; g <= ng; ++g) {
        if (s->group_count[g] == 0) { int id = s->n_nodes++; s->nodes[id].lo = V(INFINITY, INFINITY, INFINITY); s->nodes[id].hi = V(-INFINITY, -INFINITY, -INFINITY); s->nodes[id].count = 0; s->nodes[id].left = s->nodes[id].right = id; s->nodes[id].first = 0; s->group_root[g] = -2; continue; }
        s->group_root[g] = build_bvh(s, s->bvh_tris, (int) s->group_first[g], (int) s->group_count[g], cent, tlo, thi);
    }
    free(cent); free(tlo); free(thi);
    /* emitter selection PDF (scene.cpp:383-388; pmf.h:56-58 append, :103-116 normalize) */
    uint32_t ne = d->n_emitters;
    s->emitter_cdf = (float *) calloc(ne + 1, 4); s->area_cdf = (float **) calloc(ne ? ne : 1, sizeof(float *)); s->inv_area = (float *) calloc(ne ? ne : 1, 4);
    for (uint32_t e = 0; e < ne; ++e) s->emitter_cdf[e + 1] = s->emitter_cdf[e] + s->emitters[e].weight;
    if (ne) { float sum = s->emitter_cdf[ne]; s->emitter_norm = sum > 0 ? 1.0f / sum : 0.0f; for (uint32_t e = 1; e <= ne; ++e) s->emitter_cdf[e] *= s->emitter_norm; s->emitter_cdf[ne] = 1.0f; }
    /* per-mesh area distribution (trimesh.cpp:389-402 prepareSamplingTable; triangle.cpp:61-67 surfaceArea) */
    for (uint32_t e = 0; e < ne; ++e) {
        if (s->emitters[e].type != 0 || (uint32_t) s->emitters[e].shape >= d->n_shapes) continue;
        const orc_shape *sh = &s->shapes[s->emitters[e].shape]; uint32_t nt = sh->tri_count;
        float *cdf = (float *) calloc(nt + 1, 4);
        for (uint32_t t = 0; t < nt; ++t) {
            uint32_t prim = sh->first_tri + t;
            v3 p0 = vert(s, s->idx[prim * 3]), p1 = vert(s, s->idx[prim * 3 + 1]), p2 = vert(s, s->idx[prim * 3 + 2]);
            v3 c = cross(sub(p1, p0), sub(p2, p0));
            cdf[t + 1] = cdf[t] + 0.5f * sqrtf(dot(c, c));
        }
        float sum = cdf[nt]; float norm = 1.0f / sum;
        for (uint32_t t = 1; t <= nt; ++t) cdf[t] *= norm;
        cdf[nt] = 1.0f;
        s->area_cdf[e] = cdf; s->inv_area[e] = 1.0f / sum;
    }
    /* reconstruction filter table (src/libcore/rfilter.cpp:37-56; eval of src/rfilters/box.cpp:31-48, gaussian.cpp:30-57, tent.cpp:36-38,
     * mitchell.cpp:49-61, catmullrom.cpp:36-49, lanczos.cpp:36-46).  filter_radius / filter_stddev carry: box radius; gaussian stddev (in
     * filter_stddev); mitchell B, C; lanczos lobes (in filter_radius) */
    {
        const uint32_t kind = d->filter;
        float radius = kind == 0 ? d->filter_radius + 1e-5f : kind == 1 ? 4.0f * d->filter_stddev : kind == 2 ? 1.0f : kind == 5 ? (float) (int) d->filter_radius : 2.0f;
        float alpha = -1.0f / (2.0f * d->filter_stddev * d->filter_stddev), bias = expf(alpha * radius * radius);
        const float B = kind == 3 ? d->filter_radius : 0.0f, C = kind == 3 ? d->filter_stddev : 0.5f;
        float sum = 0.0f;
        for (int i = 0; i < FILTER_RES; ++i) {
            float x = (radius * (float) i) / (float) FILTER_RES, v;
            if (kind == 0) v = fabsf(x) <= radius ? 1.0f : 0.0f;
            else if (kind == 1) v = maxf(0.0f, expf(alpha * x * x) - bias);
            else if (kind == 2) v = maxf(0.0f, 1.0f - fabsf(x / radius));
            else if (kind == 5) {
                float ax = fabsf(x);
                if (ax < 1e-4f) v = 1.0f; else if (ax > radius) v = 0.0f;
                else { float x1 = M_PI_F * ax, x2 = x1 / radius; v = (sinf(x1) * sinf(x2)) / (x1 * x2); }
            } else {
                float ax = fabsf(x), x2 = ax * ax, x3 = x2 * ax;
                if (ax < 1) v = 1.0f / 6.0f * ((12 - 9 * B - 6 * C) * x3 + (-18 + 12 * B + 6 * C) * x2 + (6 - 2 * B));
                else if (ax < 2) v = 1.0f / 6.0f * ((-B - 6 * C) * x3 + (6 * B + 30 * C) * x2 + (-12 * B - 48 * C) * ax + (8 * B + 24 * C));
                else v = 0.0f;
            }
            s->filter_values[i] = v; sum += v;
        }
        s->filter_values[FILTER_RES] = 0.0f;
        s->filter_scale = (float) FILTER_RES / radius; s->filter_radius = radius;
        s->border = (int) ceilf(radius - 0.5f);
        sum *= 2 * radius / (float) FILTER_RES;
        float norm = 1.0f / sum;
        for (int i = 0; i < FILTER_RES; ++i) s->filter_values[i] *= norm;
    }
    /* environment emitter tables (envmap.cpp:264-330 configure; :336-347 createShape: sphere around kd-tree box + sensor position, x1.5) */
    s->env_index = -1;
    for (uint32_t e = 0; e < ne; ++e) if (s->emitters[e].type == 1 || s->emitters[e].type == 2) { s->env_index = (int) e; s->env_constant = s->emitters[e].type == 2; }
    if (s->env_constant) {   /* ConstantBackgroundEmitter::createShape (constant.cpp:69-74): scene AABB incl. the sensor, radius x 1.5 */
        v3 blo = s->aabb_lo, bhi = s->aabb_hi; v3 cam = V(d->cam_to_world[3], d->cam_to_world[7], d->cam_to_world[11]);
        blo = V(minf(blo.x, cam.x), minf(blo.y, cam.y), minf(blo.z, cam.z)); bhi = V(maxf(bhi.x, cam.x), maxf(bhi.y, cam.y), maxf(bhi.z, cam.z));
        v3 c = scale(add(bhi, blo), 0.5f); v3 cm = sub(c, bhi);
        s->env_bs_center = c; s->env_bs_radius = maxf(EPSILON, sqrtf(dot(cm, cm)) * 1.5f);
    }
    {   /* DirectionalEmitter::createShape (directional.cpp:87-93): kd-tree AABB bounding sphere x 1.1 */
        v3 c = scale(add(s->aabb_hi, s->aabb_lo), 0.5f), cm = sub(c, s->aabb_hi);
        s->dir_bs_center = c; s->dir_bs_radius = sqrtf(dot(cm, cm)) * 1.1f;
    }
    s->spot_cos_beam = (float *) calloc(ne ? ne : 1, 4); s->spot_cos_cutoff = (float *) calloc(ne ? ne : 1, 4); s->spot_inv_transition = (float *) calloc(ne ? ne : 1, 4);
    s->spot_cutoff = (float *) calloc(ne ? ne : 1, 4); s->spot_to_local = (float *) calloc(ne ? ne : 1, 36);
    for (uint32_t e = 0; e < ne; ++e) if (s->emitters[e].type == 4) {       /* SpotEmitter constructor + configure (spot.cpp:70-96); trafo.inverse() of a rigid toWorld */
        float beam = s->emitters[e].beam * (M_PI_F / 180.0f), cutoff = s->emitters[e].cutoff * (M_PI_F / 180.0f);
        s->spot_cos_beam[e] = cosf(beam); s->spot_cos_cutoff[e] = cosf(cutoff); s->spot_cutoff[e] = cutoff; s->spot_inv_transition[e] = 1.0f / (cutoff - beam);
        const float *m = s->emitters[e].to_world; float *o = &s->spot_to_local[e * 9];
        float a[9] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]};
        float det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]); float id = 1.0f / det;
        o[0] = (a[4] * a[8] - a[5] * a[7]) * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
        o[3] = (a[5] * a[6] - a[3] * a[8]) * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
        o[6] = (a[3] * a[7] - a[4] * a[6]) * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    }
    if (s->env_index >= 0 && !s->env_constant && d->env_rgb) {
        const int W = (int) d->env_w, H = (int) d->env_h; s->env_w = W; s->env_h = H; s->env_scale = d->env_scale;
        s->env_rgb = (float *) dup(d->env_rgb, (size_t) W * H * 12);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) s->env_to_world[i * 3 + j] = d->env_to_world[i * 4 + j];
        /* inverse of a rotation (+ uniform scale) handed over by the caller as the transpose would be wrong in general: invert the 3x3 */
        { const float *m = s->env_to_world; float det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]); float id = 1.0f / det;
          float *o = s->env_to_local;
          o[0] = (m[4] * m[8] - m[5] * m[7]) * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
          o[3] = (m[5] * m[6] - m[3] * m[8]) * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
          o[6] = (m[3] * m[7] - m[4] * m[6]) * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id; }
        s->env_cdf_cols = (float *) calloc((size_t) (W + 1) * H, 4); s->env_cdf_rows = (float *) calloc((size_t) H + 1, 4); s->env_row_weights = (float *) calloc((size_t) H, 4);
        size_t colPos = 0, rowPos = 0; float rowSum = 0.0f;
        s->env_cdf_rows[rowPos++] = 0;
        for (int y = 0; y < H; ++y) {
            float colSum = 0; s->env_cdf_cols[colPos++] = 0;
            for (int x = 0; x < W; ++x) { colSum += luminance(env_texel(s, x, y)); s->env_cdf_cols[colPos++] = colSum; }
            float normalization = 1.0f / colSum;
            for (int x = 1; x < W; ++x) s->env_cdf_cols[colPos - x - 1] *= normalization;
            s->env_cdf_cols[colPos - 1] = 1.0f;
            float weight = sinf(((float) y + 0.5f) * M_PI_F / (float) H);
            s->env_row_weights[y] = weight; rowSum += colSum * weight; s->env_cdf_rows[rowPos++] = rowSum;
        }
        float normalization = 1.0f / rowSum;
        for (int y = 1; y < H; ++y) s->env_cdf_rows[rowPos - y - 1] *= normalization;
        s->env_cdf_rows[rowPos - 1] = 1.0f;
        s->env_normalization = 1.0f / (rowSum * (2 * M_PI_F / (float) W) * (M_PI_F / (float) H));
        s->env_pixel_w = 2 * M_PI_F / (float) W; s->env_pixel_h = M_PI_F / (float) H;
        /* Scene::initialize: m_aabb = kd-tree box, expanded by the sensor's position (scene.cpp:399-403); AABB::getBSphere (aabb.cpp:44-47) */
        v3 blo = s->aabb_lo, bhi = s->aabb_hi; v3 cam = V(d->cam_to_world[3], d->cam_to_world[7], d->cam_to_world[11]);
        blo = V(minf(blo.x, cam.x), minf(blo.y, cam.y), minf(blo.z, cam.z)); bhi = V(maxf(bhi.x, cam.x), maxf(bhi.y, cam.y), maxf(bhi.z, cam.z));
        v3 c = scale(add(bhi, blo), 0.5f); v3 cm = sub(c, bhi);
        s->env_bs_center = c; s->env_bs_radius = maxf(EPSILON, sqrtf(dot(cm, cm)) * 1.5f);
    }
    /* Sobol film resolution (sobol.cpp:147-157 setFilmResolution, bucketed) */
    { uint32_t r = d->width > d->height ? d->width : d->height, p = 1, l = 0; while (p < r) { p <<= 1; ++l; } s->resolution = (float) p; s->log_res = l; }
    s->inv_res_x = 1.0f / (float) d->width; s->inv_res_y = 1.0f / (float) d->height;
    return s;
}
void orc_scene_destroy(orc_scene *s) {
    if (!s) return;
    for (uint32_t e = 0; e < s->d.n_emitters; ++e) free(s->area_cdf[e]);   /* NULL for the environment emitter */
    free(s->area_cdf); free(s->inv_area); free(s->emitter_cdf); free(s->nodes); free(s->bvh_tris); free(s->accel); free(s->tri_shape); free(s->analytic); free(s->instances); free(s->group_root); free(s->group_lo); free(s->group_hi); free(s->group_first); free(s->group_count);
    free(s->spot_cos_beam); free(s->spot_cos_cutoff); free(s->spot_inv_transition); free(s->spot_cutoff); free(s->spot_to_local);
    free(s->env_rgb); free(s->env_cdf_cols); free(s->env_cdf_rows); free(s->env_row_weights);
    free(s->tex_levels); free(s->tex_texels); free(s->uv); free(s->tangents); free(s->textures); free(s->pos); free(s->nrm); free(s->idx); free(s->shapes); free(s->materials); free(s->material_tables); free(s->emitters); free(s);
}
