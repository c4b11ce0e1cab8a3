// api.cpp -- C-ABI of libmi355pt.so (include/mi355pt.h): scene upload, wavefront batch scheduling, film read-back.
// Host orchestration only; all arithmetic on the sample path lives in the kernel translation units (kernels_*.hip over trace.h, trace_fused.h, shade.h, pt_device.h).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>
#include <set>
#include "scene_host.h"
#include "queues.h"

extern "C" {
void mi_launch_generate(const DScene &, const RenderConst &, const Queues &, const BatchDesc &, uint32_t, hipStream_t);
void mi_launch_extend(const DScene &, const Queues &, int, uint32_t, hipStream_t);
void mi_launch_shade_d(const DScene &, const RenderConst &, const Queues &, int, uint32_t, size_t, hipStream_t);
void mi_launch_shade_d_env(const DScene &, const RenderConst &, const Queues &, int, uint32_t, size_t, hipStream_t);
void mi_launch_shade_rc(const DScene &, const RenderConst &, const Queues &, int, uint32_t, size_t, hipStream_t);
void mi_launch_shade_rc_env(const DScene &, const RenderConst &, const Queues &, int, uint32_t, size_t, hipStream_t);
void mi_launch_shade_rcw(const DScene &, const RenderConst &, const Queues &, int, uint32_t, size_t, hipStream_t);
void mi_launch_shade_rcw_env(const DScene &, const RenderConst &, const Queues &, int, uint32_t, size_t, hipStream_t);
void mi_launch_shadow(const DScene &, const Queues &, uint32_t, hipStream_t);
bool mi_fused_walk(const DScene &);
uint32_t mi_fused_grid(void);
void mi_launch_extend_fused(const DScene &, const Queues &, int, uint32_t *, hipStream_t);
void mi_launch_shadow_fused(const DScene &, const Queues &, uint32_t *, hipStream_t);
void mi_launch_shade_vol(const DScene &, const RenderConst &, const Queues &, int, uint32_t, size_t, hipStream_t);
void mi_launch_shadow_vol(const DScene &, const Queues &, uint32_t, hipStream_t);
void mi_launch_shade_volmis(const DScene &, const RenderConst &, const Queues &, int, uint32_t, size_t, hipStream_t);
void mi_launch_shadow_volmis(const DScene &, const Queues &, uint32_t, hipStream_t);
void mi_launch_film(const DScene &, const Queues &, const BatchDesc &, float *, float *, hipStream_t);
void mi_launch_env_primary(const DScene &, const RenderConst &, const Queues &, int, uint32_t, hipStream_t);
void mi_launch_film_layout(const float *, const float *, float *, int, int, int, int, hipStream_t);
void mi_launch_gather_samples(const Queues &, const uint32_t *, uint64_t, float *, hipStream_t);
void mi_launch_film_add(float *, const float *, size_t, hipStream_t);
void mi_launch_debug_intersect(const DScene &, const float *, uint64_t, int, float *, int *, hipStream_t);
void mi_launch_ray_intersect(const DScene &, const float *, uint64_t, mi_intersection *, hipStream_t);
void mi_launch_debug_sobol(const DScene &, const uint32_t *, uint64_t, uint32_t, unsigned long long *, float *, hipStream_t);
void mi_launch_debug_camera(const DScene &, const float *, uint64_t, float *, hipStream_t);
void mi_launch_debug_sincosf(const float *, uint64_t, float *, hipStream_t);
void mi_launch_debug_libm(int, const float *, const float *, uint64_t, float *, hipStream_t);
}

// Shading stage dispatch.  Dynamic LDS: Sobol nibble tables + (small scenes) the scene tables + (scenes with non-diffuse BSDFs) the per-wave path-order list.
// (Round 2 experiment, removed: two launches per bounce -- diffuse hits through the diffuse-only kernel, the rest through the full kernel -- measured SLOWER
// than one launch of the full kernel over the class-sorted list, 1579 vs 1675 Msamples/s on the Veach scene: a wave of diffuse hits already skips the other
// BSDFs' code at run time, and the stage runs at two waves per SIMD either way)
static void mi_launch_shade(const DScene &scIn, bool ldsTables, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, hipStream_t st) {
    DScene sc = scIn; sc.small_tables = ldsTables ? 1u : 0u;
    size_t lds = rc.sampler == 1 ? (size_t) rc.nib_dims * rc.nib_count * 64 : 16;
    const bool env = sc.env_index >= 0;
    if (sc.small_tables) lds += 16 + 4 * ((size_t) sc.n_tris * (4 * MI_SHADE_WORDS) + sc.n_materials * 16 + sc.n_emitters * 12 + ((sc.n_emitters + 4) & ~3u) + sc.area_cdf_len);
    RenderConst rcl = rc; rcl.order_offset_words = 0;
    if (sc.has_roughconductor && q.cap <= 8192u) {      // path-order list: only where it still fits the 64 KB a launch may request (else unsorted shading)
        const uint32_t off = (uint32_t) ((lds + 15) / 16 * 4); const size_t total = (size_t) off * 4 + (size_t) q.cap * 2 * 4 + 16;
        if (total <= 64 * 1024) { rcl.order_offset_words = off; lds = total; }
    }
    if (!sc.has_roughconductor) (env ? mi_launch_shade_d_env : mi_launch_shade_d)(sc, rcl, q, buf, grid, lds, st);
    else if (sc.has_adapters & 1u) (env ? mi_launch_shade_rcw_env : mi_launch_shade_rcw)(sc, rcl, q, buf, grid, lds, st);      // mixturebsdf / bumpmap / normalmap present
    else (env ? mi_launch_shade_rc_env : mi_launch_shade_rc)(sc, rcl, q, buf, grid, lds, st);
}
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(MI_ERR_DEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

static std::vector<uint32_t> g_sobolM32; static std::vector<uint64_t> g_sobolVdc, g_sobolVdcInv; static uint32_t g_sobolDims = 0;

struct mi_scene { mi::SceneHost h; };

struct mi_render {
    mi_scene *scene = nullptr; mi_render_params p{}; RenderConst rc{};
    DScene sc{};   // this render's view of the scene: the scene's record, with the triangle packet switched off for the volumetric integrators (their stages walk the tree, which every scene has)
    Queues q{}; std::vector<void *> allocs; uint64_t poolPaths = 0; uint32_t grid = 0, gridExtend = 0, gridShade = 0, gridShadow = 0;
    float *film = nullptr, *spill = nullptr; size_t filmFloats = 0; float *layoutTmp = nullptr, *mergeTmp = nullptr;   // mergeTmp: staging for another device's film (mi_render_merge_film)   // film: own-pixel sums; spill: cross-pixel splats (atomics)
    hipStream_t stream = nullptr; hipEvent_t evBegin = nullptr, evEnd = nullptr;
    std::atomic<int> cancel{0};
    bool profiling = false; std::vector<hipEvent_t> evPool; std::vector<int> evTag;   // tag: 0 generate/film, 1 extend, 2 shade, 3 shadow
    mi_stats stats{};
    uint64_t samplesTotal = 0, launchesAll = 0; uint64_t mergedRays = 0, mergedShadow = 0, mergedPathLen = 0, mergedSamples = 0;   // counters of replicas merged into this film
    bool ldsTables = false;   // this render stages the scene tables in LDS (scene eligible and everything fits 64 KB together with the Sobol tables and the order list)
    uint32_t *pollHost = nullptr; size_t pollWords = 0; hipEvent_t pollEv = nullptr;   // unbounded depth: pinned landing buffer + event for the survivor-count poll
    uint32_t *dNib = nullptr; void *dSobolTabs = nullptr;   // dSobolTabs: the three look_up tables of k_generate (frame, px, py), one allocation
    // optional further path pools + streams: consecutive batches go round-robin through them, so the ALU-bound traversal kernels of one batch
    // overlap the latency-bound shading kernels of the others on the same CUs (MI355PT_STREAMS = 1..4 pools, default 2)
    enum { kMaxPools = 4 };
    Queues qx[kMaxPools - 1]{}; std::vector<void *> allocsx[kMaxPools - 1]; hipStream_t streamx[kMaxPools - 1] = {}; hipEvent_t filmDone[kMaxPools] = {}, joinEv[kMaxPools] = {}; int nStreams = 1;
    Queues &pool(int i) { return i ? qx[i - 1] : q; }
    hipStream_t poolStream(int i) { return i ? streamx[i - 1] : stream; }
};

extern "C" {

const char *mi_last_error(void) { return g_err.c_str(); }

int mi_set_sobol_tables(const uint32_t *m32, uint32_t dims, const uint64_t *vdc, const uint64_t *vdcInv) {
    if (!m32 || !vdc || !vdcInv || dims < 2) return fail(MI_ERR_INVALID, "mi_set_sobol_tables: null table or dims < 2");
    g_sobolM32.assign(m32, m32 + (size_t) dims * MI_SOBOL_SIZE); g_sobolDims = dims;
    g_sobolVdc.assign(vdc, vdc + 16 * MI_SOBOL_SIZE); g_sobolVdcInv.assign(vdcInv, vdcInv + 16 * MI_SOBOL_SIZE);
    return MI_OK;
}

// Reads mitsuba-im_amd/data/sobol_tables.bin ("MISOBOL1", dims, rows, matrices32[dims][52], vdc[rows][52], vdc_inv[rows][52]).
int mi_load_sobol_tables(const char *path) {
    FILE *f = fopen(path, "rb"); if (!f) return fail(MI_ERR_INVALID, std::string("mi_load_sobol_tables: cannot open ") + path);
    char magic[8]; uint32_t hd[2]; bool ok = fread(magic, 1, 8, f) == 8 && !memcmp(magic, "MISOBOL1", 8) && fread(hd, 4, 2, f) == 2 && hd[1] == 16;
    std::vector<uint32_t> m32; std::vector<uint64_t> vdc, vdi;
    if (ok) { m32.resize((size_t) hd[0] * MI_SOBOL_SIZE); vdc.resize(16 * MI_SOBOL_SIZE); vdi.resize(16 * MI_SOBOL_SIZE);
              ok = fread(m32.data(), 4, m32.size(), f) == m32.size() && fread(vdc.data(), 8, vdc.size(), f) == vdc.size() && fread(vdi.data(), 8, vdi.size(), f) == vdi.size(); }
    fclose(f);
    if (!ok) return fail(MI_ERR_INVALID, std::string("mi_load_sobol_tables: malformed file ") + path);
    return mi_set_sobol_tables(m32.data(), hd[0], vdc.data(), vdi.data());
}
}  // extern "C"
#include <dlfcn.h>
// hosts that never call mi_set_sobol_tables (the adapter plugin): look for data/sobol_tables.bin next to this shared library
static void ensureSobolTables() {
    if (g_sobolDims) return;
    const char *env = getenv("MI355PT_DATA");
    if (env && mi_load_sobol_tables((std::string(env) + "/sobol_tables.bin").c_str()) == MI_OK) return;
    Dl_info info;
    if (dladdr((void *) &mi_set_sobol_tables, &info) && info.dli_fname) {
        std::string dir(info.dli_fname); size_t slash = dir.rfind('/'); dir = slash == std::string::npos ? "." : dir.substr(0, slash);
        (void) mi_load_sobol_tables((dir + "/data/sobol_tables.bin").c_str());
    }
}
extern "C" {

// ------------------------------------------------------------------------------------------------ scene
int mi_scene_create(mi_scene **out) { if (!out) return fail(MI_ERR_INVALID, "mi_scene_create: out is null"); *out = new mi_scene(); return MI_OK; }
void mi_scene_destroy(mi_scene *s) { delete s; }

int mi_scene_set_triangles(mi_scene *s, const float *pos, const float *nrm, const float *uv, const uint32_t *idx,
                           uint32_t nv, uint32_t nt, const mi_shape *shapes, uint32_t ns) {
    if (s && nt == 0 && ns == 0) { s->h.pos.clear(); s->h.idx.clear(); s->h.nrm.clear(); s->h.shapes.clear(); s->h.committed = false; return MI_OK; }   // analytic-only scene
    if (!s || !pos || !idx || !shapes || !ns) return fail(MI_ERR_INVALID, "mi_scene_set_triangles: null argument");
    for (uint32_t i = 0; i < ns; ++i) {
        const mi_shape &sh = shapes[i];
        if ((uint64_t) sh.first_tri + sh.tri_count > nt || (uint64_t) sh.first_vert + sh.vert_count > nv || sh.tri_count == 0)
            return fail(MI_ERR_INVALID, "mi_scene_set_triangles: shape range outside the arrays (or an empty mesh)");
    }
    for (uint64_t i = 0; i < (uint64_t) nt * 3; ++i) if (idx[i] >= nv) return fail(MI_ERR_INVALID, "mi_scene_set_triangles: vertex index out of range");
    s->h.pos.assign(pos, pos + (size_t) nv * 3); s->h.idx.assign(idx, idx + (size_t) nt * 3);
    if (nrm) s->h.nrm.assign(nrm, nrm + (size_t) nv * 3); else s->h.nrm.clear();
    if (uv) s->h.uv.assign(uv, uv + (size_t) nv * 2); else s->h.uv.clear();
    s->h.shapes.assign(shapes, shapes + ns); s->h.committed = false;
    return MI_OK;
}
int mi_scene_set_analytic(mi_scene *s, const mi_analytic *a, uint32_t n) {
    if (!s || (n && !a)) return fail(MI_ERR_INVALID, "mi_scene_set_analytic: null argument");
    for (uint32_t i = 0; i < n; ++i) {
        if (a[i].type > MI_SHAPE_CYLINDER) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_analytic: only rectangle, disk, sphere and cylinder are implemented");
        if (a[i].type >= MI_SHAPE_SPHERE && !(a[i].radius > 0)) return fail(MI_ERR_INVALID, "Cannot create spheres of radius <= 0");      // sphere.cpp:130-131
        if (a[i].type == MI_SHAPE_CYLINDER && !(a[i].length > 0)) return fail(MI_ERR_INVALID, "mi_scene_set_analytic: cylinder of length <= 0");
        if (a[i].to_world[12] != 0 || a[i].to_world[13] != 0 || a[i].to_world[14] != 0 || a[i].to_world[15] != 1.0f)
            return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_analytic: toWorld must be affine");
    }
    s->h.analytic.assign(a, a + n); s->h.committed = false; return MI_OK;
}
int mi_scene_set_textures(mi_scene *s, const mi_texture *t, uint32_t n) {
    if (!s || (n && !t)) return fail(MI_ERR_INVALID, "mi_scene_set_textures: null argument");
    for (uint32_t i = 0; i < n; ++i) {
        if (t[i].type > MI_TEXTURE_BITMAP) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_textures: implemented textures: checkerboard, gridtexture, bitmap");
        if (t[i].type == MI_TEXTURE_BITMAP && (t[i].wrap_u > 4 || t[i].wrap_v > 4 || t[i].filter > 3 || !t[i].n_levels)) return fail(MI_ERR_INVALID, "mi_scene_set_textures: bad wrap mode / filter type / level count");
    }
    s->h.textures.assign(t, t + n); s->h.committed = false; return MI_OK;
}
int mi_scene_set_texture_data(mi_scene *s, const uint32_t *levels, uint32_t nLevels, const float *texels, uint64_t nTexels) {
    if (!s || !levels || !texels || !nLevels || !nTexels) return fail(MI_ERR_INVALID, "mi_scene_set_texture_data: null argument");
    for (uint32_t i = 0; i < nLevels; ++i)
        if (!levels[i * 3] || !levels[i * 3 + 1] || (uint64_t) levels[i * 3 + 2] + (uint64_t) levels[i * 3] * levels[i * 3 + 1] * 3 > nTexels) return fail(MI_ERR_INVALID, "mi_scene_set_texture_data: MIP level outside the texel buffer");
    s->h.texLevels.assign(levels, levels + (size_t) nLevels * 3); s->h.texTexels.assign(texels, texels + nTexels); s->h.committed = false; return MI_OK;
}
int mi_scene_set_envmap_filter(mi_scene *s, int32_t texture) {
    if (!s) return fail(MI_ERR_INVALID, "mi_scene_set_envmap_filter: null argument");
    if (texture < -1) return fail(MI_ERR_INVALID, "mi_scene_set_envmap_filter: texture index must be >= -1");
    s->h.envTexture = texture; s->h.committed = false; return MI_OK;
}
int mi_scene_set_material_tables(mi_scene *s, const float *data, uint32_t n) {
    if (!s || (n && !data)) return fail(MI_ERR_INVALID, "mi_scene_set_material_tables: null argument");
    s->h.materialTables.assign(data, data + n); s->h.committed = false; return MI_OK;
}
int mi_scene_set_instances(mi_scene *s, const mi_instance *a, uint32_t n) {
    if (!s || (n && !a)) return fail(MI_ERR_INVALID, "mi_scene_set_instances: null argument");
    for (uint32_t i = 0; i < n; ++i)
        if (a[i].to_world[12] != 0 || a[i].to_world[13] != 0 || a[i].to_world[14] != 0 || a[i].to_world[15] != 1.0f) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_instances: toWorld must be affine");
    s->h.instances.assign(a, a + n); s->h.committed = false; return MI_OK;
}
int mi_scene_set_media(mi_scene *s, const mi_medium *media, uint32_t n, const int32_t *shapeMedia, uint32_t nPairs, int32_t sensorMedium) {
    if (!s || (n && (!media || !shapeMedia))) return fail(MI_ERR_INVALID, "mi_scene_set_media: null argument");
    if (n > 254) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_media: at most 254 media");
    for (uint32_t i = 0; i < n; ++i) {
        if (media[i].strategy > MI_MEDIUM_MANUAL) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_media: sampling strategies balance, single and manual are implemented (not `maximum`)");   // homogeneous.cpp:192-226
        if (media[i].phase > MI_PHASE_HG) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_media: phase functions isotropic and hg are implemented");
        if (media[i].phase == MI_PHASE_HG && !(media[i].g > -1 && media[i].g < 1)) return fail(MI_ERR_INVALID, "The anisotropy parameter 'g' must be in the range (-1, 1)!");   // hg.cpp:50-51
        for (int c = 0; c < 3; ++c) if (!(media[i].sigma_a[c] >= 0) || !(media[i].sigma_s[c] >= 0)) return fail(MI_ERR_INVALID, "mi_scene_set_media: negative coefficient");
    }
    for (uint32_t i = 0; i < nPairs * 2u; ++i) if (shapeMedia[i] < -1 || shapeMedia[i] >= (int32_t) n) return fail(MI_ERR_INVALID, "mi_scene_set_media: a shape refers to a missing medium");
    if (sensorMedium < -1 || sensorMedium >= (int32_t) n) return fail(MI_ERR_INVALID, "mi_scene_set_media: the sensor refers to a missing medium");
    s->h.media.assign(media, media + n); s->h.shapeMedia.assign(shapeMedia, shapeMedia + (size_t) nPairs * 2u); s->h.sensorMedium = sensorMedium; s->h.committed = false; return MI_OK;
}
int mi_scene_set_materials(mi_scene *s, const mi_material *m, uint32_t n) {
    if (!s || !m || !n) return fail(MI_ERR_INVALID, "mi_scene_set_materials: null argument");
    auto isWrapper = [](uint32_t t) { return t == MI_BSDF_MASK || t == MI_BSDF_MIXTURE || t == MI_BSDF_BUMPMAP || t == MI_BSDF_NORMALMAP || t == MI_BSDF_COATING || t == MI_BSDF_BLEND || t == MI_BSDF_ROUGHCOATING; };
    auto hasDelta = [](uint32_t t) { return t == MI_BSDF_CONDUCTOR || t == MI_BSDF_DIELECTRIC || t == MI_BSDF_THINDIELECTRIC || t == MI_BSDF_PLASTIC; };
    for (uint32_t i = 0; i < n; ++i) {
        if (m[i].type > MI_BSDF_ROUGHCOATING) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: implemented BSDFs: diffuse, roughdiffuse, phong, ward, coating, roughcoating, blendbsdf, roughconductor, conductor, dielectric, plastic, roughdielectric, difftrans, roughplastic, thindielectric, mask, mixturebsdf, bumpmap, normalmap (those without transmission optionally twosided)");
        if (m[i].type == MI_BSDF_MASK && (m[i].distr >= n || m[m[i].distr].type == MI_BSDF_MASK || (m[i].flags & MI_BSDF_FLAG_TWOSIDED))) return fail(MI_ERR_INVALID, "mi_scene_set_materials: a mask refers to its nested material record by index (not another mask) and cannot itself be twosided");
        if (m[i].type == MI_BSDF_BLEND) {
            int deltas = 0;
            for (int c = 0; c < 2; ++c) {
                const float idxf = m[i].eta[c];
                if (!(idxf >= 0) || idxf >= (float) n || isWrapper(m[(uint32_t) idxf].type)) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: the two BSDFs of a blendbsdf are plain BSDF records (indices in eta[0], eta[1])");
                if (((m[(uint32_t) idxf].flags >> 8) & 0xFFFFu)) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: textures on the BSDFs inside a blendbsdf are not implemented");
                deltas += hasDelta(m[(uint32_t) idxf].type);
            }
            if (deltas > 1) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: a blendbsdf of two BSDFs that both have a Dirac delta component is not implemented");
        }
        if (m[i].type == MI_BSDF_ROUGHCOATING) {
            if (m[i].distr >= n || isWrapper(m[m[i].distr].type)) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: a roughcoating nests a plain BSDF record (index in `distr`)");
            const mi_material &nm = m[m[i].distr];
            if ((nm.flags & MI_BSDF_FLAG_TWOSIDED) || hasDelta(nm.type) || nm.type == MI_BSDF_ROUGHDIELECTRIC || nm.type == MI_BSDF_DIFFTRANS || nm.type == MI_BSDF_NULL)
                return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: the BSDF under a roughcoating is a reflective one without a Dirac delta lobe, `twosided` goes on the coating");
            if (!(m[i].eta[0] > 0) || m[i].eta[0] == 1.0f) return fail(MI_ERR_INVALID, "The interior and exterior indices of refraction must be positive and differ!");      // roughcoating.cpp:126-128
            if (m[i].eta[2] != 0.0f && m[i].eta[2] != 1.0f && m[i].eta[2] != 2.0f) return fail(MI_ERR_INVALID, "Specified an invalid distribution, must be \"beckmann\", \"ggx\", or \"phong\"/\"as\"!");
            if ((m[i].flags >> 8) & 0xFFFFu) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: a textured sigmaA is not implemented");
        }
        if (m[i].type == MI_BSDF_COATING) {
            // adapters nest in the order mask -> bumpmap / normalmap -> coating -> plain BSDF (a coating over a mixturebsdf, or as the child of one, is not implemented)
            if (m[i].distr >= n || isWrapper(m[m[i].distr].type)) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: a coating nests a plain BSDF record (index in `distr`)");
            const mi_material &nm = m[m[i].distr];
            if ((nm.flags & MI_BSDF_FLAG_TWOSIDED) || nm.type == MI_BSDF_DIELECTRIC || nm.type == MI_BSDF_ROUGHDIELECTRIC || nm.type == MI_BSDF_DIFFTRANS || nm.type == MI_BSDF_THINDIELECTRIC || nm.type == MI_BSDF_NULL)
                return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: the BSDF under a coating is a reflective one, `twosided` goes on the coating");
            if (!(m[i].eta[0] > 0) || m[i].eta[0] == 1.0f) return fail(MI_ERR_INVALID, "The interior and exterior indices of refraction must be positive and differ!");      // coating.cpp:119-121
            if ((m[i].flags >> 8) & 0xFFFFu) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: a textured sigmaA is not implemented");
        }
        if (m[i].type == MI_BSDF_BUMPMAP || m[i].type == MI_BSDF_NORMALMAP) {
            // adapters nest in the order mask -> bumpmap / normalmap -> mixturebsdf -> plain BSDF
            if (m[i].distr >= n || m[m[i].distr].type == MI_BSDF_MASK || m[m[i].distr].type == MI_BSDF_BUMPMAP || m[m[i].distr].type == MI_BSDF_NORMALMAP) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: a bumpmap / normalmap nests a plain BSDF or a mixturebsdf (record index in `distr`)");
            if (m[i].flags & MI_BSDF_FLAG_TWOSIDED) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: put `twosided` on the BSDF nested in a bumpmap / normalmap, not on the adapter");
            if (!((m[i].flags >> 8) & 0xFFFFu)) return fail(MI_ERR_INVALID, m[i].type == MI_BSDF_BUMPMAP ? "A displacement texture must be specified" : "A normal map texture must be specified");   // bumpmap.cpp:88-89
        }
        if (m[i].type == MI_BSDF_MIXTURE) {
            if (m[i].distr < 2 || m[i].distr > 4) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: a mixturebsdf holds 2..4 BSDFs");
            float total = 0; int deltas = 0;
            for (uint32_t c = 0; c < m[i].distr; ++c) {
                const float idxf = c < 3 ? m[i].reflectance[c] : m[i].eta[0], w = c < 3 ? m[i].k[c] : m[i].specular[0];
                if (!(idxf >= 0) || idxf >= (float) n || isWrapper(m[(uint32_t) idxf].type)) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: the children of a mixturebsdf are plain BSDF records (indices in reflectance[0..2], eta[0])");
                if (((m[(uint32_t) idxf].flags >> 8) & 0xFFFFu)) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: textures on the children of a mixturebsdf are not implemented");
                if (!(w >= 0)) return fail(MI_ERR_INVALID, "Invalid BSDF weight!");                                    // mixturebsdf.cpp:82-83
                total += w; deltas += hasDelta(m[(uint32_t) idxf].type);
            }
            if (!(total > 0)) return fail(MI_ERR_INVALID, "The weights must sum to a value greater than zero!");       // mixturebsdf.cpp:126-127
            if (deltas > 1) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: a mixturebsdf with more than one child that has a Dirac delta component is not implemented");
        }
        if ((m[i].type == MI_BSDF_DIELECTRIC || m[i].type == MI_BSDF_ROUGHDIELECTRIC || m[i].type == MI_BSDF_DIFFTRANS || m[i].type == MI_BSDF_THINDIELECTRIC) && (m[i].flags & MI_BSDF_FLAG_TWOSIDED)) return fail(MI_ERR_INVALID, "Only BSDFs without a transmission component can be nested!");   // twosided.cpp:86-88
        if ((m[i].type == MI_BSDF_DIELECTRIC || m[i].type == MI_BSDF_PLASTIC || m[i].type == MI_BSDF_ROUGHDIELECTRIC || m[i].type == MI_BSDF_ROUGHPLASTIC || m[i].type == MI_BSDF_THINDIELECTRIC) && !(m[i].eta[0] > 0)) return fail(MI_ERR_INVALID, "The interior and exterior indices of refraction must be positive!");
        if (m[i].type == MI_BSDF_ROUGHPLASTIC && (m[i].distr > 2 || (m[i].flags & MI_BSDF_FLAG_ANISOTROPIC)))
            return fail(MI_ERR_INVALID, "The 'roughplastic' plugin currently does not support anisotropic microfacet distributions!");        // roughplastic.cpp:225-227
        if ((m[i].type == MI_BSDF_ROUGHCONDUCTOR || m[i].type == MI_BSDF_ROUGHDIELECTRIC) && m[i].distr > 2) return fail(MI_ERR_INVALID, "Specified an invalid distribution, must be \"beckmann\", \"ggx\", or \"phong\"/\"as\"!");   // microfacet.h:113-115
        if ((m[i].flags & MI_BSDF_FLAG_ANISOTROPIC) && m[i].type != MI_BSDF_ROUGHCONDUCTOR && m[i].type != MI_BSDF_ROUGHDIELECTRIC && m[i].type != MI_BSDF_WARD) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_materials: anisotropic roughness is implemented for roughconductor and roughdielectric");
    }
    s->h.materials.assign(m, m + n); s->h.committed = false; return MI_OK;
}
int mi_scene_set_emitters(mi_scene *s, const mi_emitter *e, uint32_t n) {
    if (!s || (n && !e)) return fail(MI_ERR_INVALID, "mi_scene_set_emitters: null argument");
    uint32_t nEnv = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (e[i].type > MI_EMITTER_COLLIMATED || e[i].type == 6u) return fail(MI_ERR_UNSUPPORTED, "mi_scene_set_emitters: implemented emitters: area, envmap, constant, point, spot, directional, collimated");
        nEnv += e[i].type == MI_EMITTER_ENVMAP || e[i].type == MI_EMITTER_CONSTANT;
        if (e[i].type == MI_EMITTER_SPOT && !(e[i].cutoff >= e[i].beam && e[i].beam >= 0 && e[i].cutoff > 0)) return fail(MI_ERR_INVALID, "mi_scene_set_emitters: spot needs cutoffAngle >= beamWidth >= 0");   // spot.cpp:77
    }
    if (nEnv > 1) return fail(MI_ERR_INVALID, "The scene may only contain one environment emitter");      // scene.cpp:542-543
    s->h.emitters.assign(e, e + n); s->h.committed = false; return MI_OK;
}
int mi_scene_set_envmap(mi_scene *s, const float *rgb, uint32_t w, uint32_t h, const float *toWorld, float scale) {
    if (!s || !rgb || !toWorld || w < 2 || h < 2 || w > 0xFFFF || h > 0xFFFF) return fail(MI_ERR_INVALID, "mi_scene_set_envmap: bad argument (2..65535 texels per side)");
    s->h.envRGB.assign(rgb, rgb + (size_t) w * h * 3); s->h.envW = w; s->h.envH = h; memcpy(s->h.envToWorld, toWorld, 64); s->h.envScale = scale; s->h.committed = false;
    return MI_OK;
}
int mi_scene_set_camera(mi_scene *s, const float *s2c, const float *c2w, float nearClip, float farClip) {
    if (!s || !s2c || !c2w) return fail(MI_ERR_INVALID, "mi_scene_set_camera: null argument");
    memcpy(s->h.s2c, s2c, 64); memcpy(s->h.c2w, c2w, 64); s->h.nearClip = nearClip; s->h.farClip = farClip; s->h.haveCamera = true; s->h.committed = false;
    return MI_OK;
}
int mi_scene_set_film(mi_scene *s, uint32_t w, uint32_t h, uint32_t kind, float radius, float stddev) {
    if (!s || !w || !h || kind > 5 || (kind == 5 && !(radius >= 1))) return fail(MI_ERR_INVALID, "mi_scene_set_film: bad argument");
    s->h.width = w; s->h.height = h; s->h.filterKind = kind; s->h.filterRadius = radius; s->h.filterStddev = stddev; s->h.haveFilm = true; s->h.committed = false;
    return MI_OK;
}

}  // extern "C"

namespace mi {
static int bvhDepthOf(const std::vector<BvhNode> &nodes, int n) {
    if (n < 0) return 0;
    int a = bvhDepthOf(nodes, nodes[n].c0), b = bvhDepthOf(nodes, nodes[n].c1);
    return 1 + (a > b ? a : b);
}
template <typename T> static int up(void **dst, const std::vector<T> &v) {
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    hipError_t e = hipMalloc(dst, bytes); if (e != hipSuccess) return 1;
    if (!v.empty()) { e = hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); if (e != hipSuccess) return 1; }
    return 0;
}
void SceneHost::release() {
    void **ps[] = {&dMedia, &dPrimMedia, &dPacketGroups, &dPacketExact, &dTexLevels, &dTexTexels, &dMipLut, &dTriUV, &dTextures, &dMaterialTables, &dInstances, &dEmitterX, &dAnalytic, &dNodes, &dTris, &dShade, &dI2, &dNrm, &dMaterials, &dEmitters, &dEmitterCdf, &dAreaCdf, &dFilter, &dSobolM32, &dSobolVdc, &dSobolVdcInv, &dEnvRGB, &dEnvCols, &dEnvRows, &dEnvWeights, &dEnvGuideRows, &dEnvGuideCols};
    for (void **p : ps) if (*p) { (void) hipFree(*p); *p = nullptr; }
}
int SceneHost::upload(int dev) {
    release(); device = dev;
    if (hipSetDevice(dev) != hipSuccess) return 1;
    std::vector<MaterialD> mats(materials.size());
    for (size_t i = 0; i < materials.size(); ++i) memcpy(&mats[i], &materials[i], sizeof(MaterialD));
    for (MaterialD &m : mats) if (m.type == MI_BSDF_ROUGHCOATING) {      // RoughCoating::configure (roughcoating.cpp:205-209): the same weight, thickness in eta[1]
        float avg = 0.0f; for (int c = 0; c < 3; ++c) avg += (float) exp((double) (m.reflectance[c] * (-2 * m.eta[1])));
        avg = avg * (1.0f / 3); m.k[0] = 1.0f / (avg + 1.0f);
    }
    for (MaterialD &m : mats) if (m.type == MI_BSDF_COATING) {      // SmoothCoating::configure (coating.cpp:182-186): m_specularSamplingWeight from the layer's average absorption -> k[0]
        float avg = 0.0f; for (int c = 0; c < 3; ++c) avg += (float) exp((double) (m.reflectance[c] * (-2 * m.alpha)));      // Spectrum::exp = math::fastexp per channel, then average()
        avg = avg * (1.0f / 3); m.k[0] = 1.0f / (avg + 1.0f);
    }
    for (MaterialD &m : mats) {          // plastic / roughplastic: m_specularSamplingWeight = sAvg / (dAvg + sAvg) over Texture::getAverage() (plastic.cpp:204-207, roughplastic.cpp:244-246) -> eta[1]
        if (m.type != MI_BSDF_PLASTIC && m.type != MI_BSDF_ROUGHPLASTIC) continue;
        float d[3] = {m.reflectance[0], m.reflectance[1], m.reflectance[2]}; const uint32_t tex = (m.flags >> 8) & 0xFFFFu;
        if (tex && tex <= textures.size()) {     // checkerboard.cpp:102-104, gridtexture.cpp:116-121; a bitmap's average is input (color0, from TMIPMap::getAverage)
            const mi_texture &t = textures[tex - 1];
            for (int c = 0; c < 3; ++c) {
                if (t.type == MI_TEXTURE_CHECKERBOARD) d[c] = (t.color0[c] + t.color1[c]) * 0.5f;
                else if (t.type == MI_TEXTURE_GRID) { const float iw = std::max(0.0f, 1 - 2 * t.line_width), ia = iw * iw, la = 1 - ia; d[c] = t.color1[c] * la + t.color0[c] * ia; }
                else d[c] = t.color0[c];
            }
        }
        const float dl = d[0] * 0.212671f + d[1] * 0.715160f + d[2] * 0.072169f, sl = m.specular[0] * 0.212671f + m.specular[1] * 0.715160f + m.specular[2] * 0.072169f;
        m.eta[1] = sl / (dl + sl);
    }
    std::vector<float> filt(filterValues, filterValues + MI_FILTER_RES + 1);
    // tree nodes and leaf records share ONE allocation (nodes first): the fused walk (trace_fused.h) addresses both through one base + a 32-bit byte offset
    std::vector<unsigned char> geo(nodes.size() * sizeof(BvhNode) + std::max<size_t>(tris.size(), 1) * sizeof(TriAccelD) + 16);
    if (!nodes.empty()) std::memcpy(geo.data(), nodes.data(), nodes.size() * sizeof(BvhNode));
    if (!tris.empty()) std::memcpy(geo.data() + nodes.size() * sizeof(BvhNode), tris.data(), tris.size() * sizeof(TriAccelD));
    dTris = nullptr;
    int bad = up(&dNodes, geo) | up(&dShade, shade) | up(&dI2, i2) | up(&dNrm, nrm) | up(&dMaterials, mats) |
              up(&dEmitters, emittersD) | up(&dAnalytic, analyticD) | up(&dInstances, instancesD) | up(&dMaterialTables, materialTables) | up(&dTriUV, triuv) | up(&dTextures, textures) | up(&dEmitterX, emitterX) | up(&dEmitterCdf, emitterCdf) | up(&dAreaCdf, areaCdf) | up(&dFilter, filt);
    if (bad) return 1;
    d = DScene{};
    if (g_sobolDims && logRes <= 16) {
        const uint32_t row = logRes >= 1 ? logRes - 1 : 0;   // look_up is only used for logRes > 1 (sobol.cpp:209-215)
        std::vector<uint64_t> vdc(g_sobolVdc.begin() + row * MI_SOBOL_SIZE, g_sobolVdc.begin() + (row + 1) * MI_SOBOL_SIZE);
        std::vector<uint64_t> vdi(g_sobolVdcInv.begin() + row * MI_SOBOL_SIZE, g_sobolVdcInv.begin() + (row + 1) * MI_SOBOL_SIZE);
        if (up(&dSobolM32, g_sobolM32) | up(&dSobolVdc, vdc) | up(&dSobolVdcInv, vdi)) return 1;
        d.sobol_m32 = (const uint32_t *) dSobolM32; d.sobol_vdc = (const uint64_t *) dSobolVdc; d.sobol_vdc_inv = (const uint64_t *) dSobolVdcInv;
        d.sobol_dims = g_sobolDims;
    }
    d.nodes = (const BvhNode *) dNodes; d.tris = (const TriAccelD *) ((const unsigned char *) dNodes + nodes.size() * sizeof(BvhNode)); d.geo_bytes = geo.size(); d.shade = (const TriShade *) dShade; d.i2 = (const uint32_t *) dI2;
    d.nrm = (const float *) dNrm; d.materials = (const MaterialD *) dMaterials; d.emitters = (const EmitterD *) dEmitters;
    d.emitter_cdf = (const float *) dEmitterCdf; d.area_cdf = (const float *) dAreaCdf; d.filter_values = (const float *) dFilter;
    d.analytic = (const AnalyticD *) dAnalytic; d.n_analytic = (uint32_t) analyticD.size();
    {   // EWA weight table (mipmap.h:297-302; math::fastexp on Linux/x86_64 = (float) exp((double) x)); PerspectiveCameraImpl::m_dx / m_dy (perspective.cpp:159-163)
        std::vector<float> lut(64); for (int i = 0; i < 64; ++i) { float r2 = (float) i / 63.0f; lut[i] = (float) std::exp((double) (-2.0f * r2)) - (float) std::exp((double) -2.0f); }
        if (up(&dTexLevels, texLevels) | up(&dTexTexels, texTexels) | up(&dMipLut, lut)) return 1;
        d.tex_levels = (const uint32_t *) dTexLevels; d.tex_texels = (const float *) dTexTexels; d.mip_lut = (const float *) dMipLut;
        const float *m = s2c; const float irx = 1.0f / (float) width, iry = 1.0f / (float) height;
        auto pt = [&](float px, float py, float *o) {
            float x = m[0] * px + m[1] * py + m[2] * 0.0f + m[3], y = m[4] * px + m[5] * py + m[6] * 0.0f + m[7], z = m[8] * px + m[9] * py + m[10] * 0.0f + m[11], w = m[12] * px + m[13] * py + m[14] * 0.0f + m[15];
            if (w != 1.0f) { float r = 1.0f / w; x *= r; y *= r; z *= r; }
            o[0] = x; o[1] = y; o[2] = z; };
        float p0[3], px[3], py[3]; pt(0.0f, 0.0f, p0); pt(irx, 0.0f, px); pt(0.0f, iry, py);
        for (int i = 0; i < 3; ++i) { d.cam_dx[i] = px[i] - p0[i]; d.cam_dy[i] = py[i] - p0[i]; }
    }
    d.material_tables = (const float *) dMaterialTables; d.triuv = (const TriUV *) dTriUV; d.textures = (const TextureD *) dTextures;
    bool matTextures = false; for (const auto &m : mats) if ((m.flags >> 8) & 0xFFFFu) matTextures = true;
    d.n_textures = matTextures ? (uint32_t) textures.size() : 0u; d.env_texture = envTexture >= 0 ? (uint32_t) envTexture + 1u : 0u;
    d.instances = (const InstanceD *) dInstances; d.n_instances = (uint32_t) instancesD.size();
    d.emitter_x = (const float *) dEmitterX; d.env_constant = envConstant ? 1u : 0u; d.ext = (!analyticD.empty() || !instancesD.empty() || hasDeltaEmitters || anyUV || matTextures) ? 1u : 0u;
    memcpy(d.dir_bs_center, dirBsCenter, 12); d.dir_bs_radius = dirBsRadius;
    d.n_tris = nTris; d.n_nodes = (uint32_t) nodes.size(); d.n_emitters = (uint32_t) emittersD.size(); d.n_materials = (uint32_t) mats.size();
    d.emitter_norm = emitterNorm;
    for (int i = 0; i < 3; ++i) { d.aabb_lo[i] = aabbLo[i]; d.aabb_hi[i] = aabbHi[i]; }
    memcpy(d.s2c, s2c, 64); memcpy(d.c2w, c2w, 64);
    d.near_clip = nearClip; d.far_clip = farClip; d.inv_res_x = 1.0f / (float) width; d.inv_res_y = 1.0f / (float) height;
    d.width = width; d.height = height;
    d.filter_radius = filterRadiusEff; d.filter_scale = filterScale; d.border = border;
    d.log_res = logRes; d.resolution = resolution;
    d.env_index = envIndex;
    d.env_bs_radius = envBsRadius; memcpy(d.env_bs_center, envBsCenter, 12);
    if (envIndex >= 0 && !envConstant) {
        if (up(&dEnvRGB, envRGB) | up(&dEnvCols, envCdfCols) | up(&dEnvRows, envCdfRows) | up(&dEnvWeights, envRowWeights)) return 1;
        if (!envGuideRows.empty()) { if (up(&dEnvGuideRows, envGuideRows) | up(&dEnvGuideCols, envGuideCols)) return 1;
            d.env_guide_rows_t = (const uint16_t *) dEnvGuideRows; d.env_guide_cols_t = (const uint16_t *) dEnvGuideCols; d.env_guide_rows = envGuideKR; d.env_guide_cols = envGuideKC; }
        d.env_rgb = (const float *) dEnvRGB; d.env_cdf_cols = (const float *) dEnvCols; d.env_cdf_rows = (const float *) dEnvRows; d.env_row_weights = (const float *) dEnvWeights;
        d.env_w = (int) envW; d.env_h = (int) envH; d.env_normalization = envNormalization; d.env_scale = envScale;
        d.env_pixel_w = 2 * MI_PI / (float) envW; d.env_pixel_h = MI_PI / (float) envH; d.env_bs_radius = envBsRadius;
        memcpy(d.env_to_world, envToWorld3, 36); memcpy(d.env_to_local, envToLocal3, 36); memcpy(d.env_bs_center, envBsCenter, 12);
    }
    d.bvh_depth = (uint32_t) bvhDepth; d.bvh_wide = wideBvh ? 1u : 0u; d.bvh_stack_direct = (uint32_t) bvhStackDirect;
    d.area_cdf_len = (uint32_t) areaCdf.size();
    { uint32_t maxLightTris = 0; for (const EmitterD &e : emittersD) if (e.tri_count > maxLightTris) maxLightTris = e.tri_count;
      const char *sf = getenv("MI355PT_SEARCH"); d.search_flags = sf ? (uint32_t) atoi(sf) : ((emittersD.size() <= 3 ? 1u : 0u) | (maxLightTris <= 3u ? 2u : 0u)); }
    { const char *ns = getenv("MI355PT_NO_LDS_TABLES");
      d.small_tables = (nTris <= 400 && mats.size() <= 64 && emittersD.size() <= 32 && areaCdf.size() <= 2048 && !(ns && ns[0] == '1')) ? 1u : 0u; }   // ELIGIBLE for LDS staging; mi_render_create decides per render whether it fits next to the Sobol tables
    d.has_roughconductor = 0; d.has_diffuse = 0;
    d.has_adapters = 0;
    for (const mi_material &m : materials) { if (m.type != MI_BSDF_DIFFUSE) d.has_roughconductor = 1; else d.has_diffuse = 1; if (m.type == MI_BSDF_MIXTURE || m.type == MI_BSDF_BUMPMAP || m.type == MI_BSDF_NORMALMAP || m.type == MI_BSDF_COATING || m.type == MI_BSDF_ROUGHCOATING) d.has_adapters |= 1u; if (m.type == MI_BSDF_COATING || m.type == MI_BSDF_BLEND || m.type == MI_BSDF_ROUGHCOATING) d.has_adapters |= 4u; if (m.type == MI_BSDF_BLEND) d.has_adapters |= 1u; }   // any non-diffuse material -> k_shade<RC = true>; both kinds -> two shading launches per bounce (class split)
    // bit 1: ENull lobes the volumetric walks have to evaluate through a wrapper -- a `mask`, or a mixturebsdf with a `null` / `thindielectric` child (surfaceNullEval; the NX kernel variants)
    for (const mi_material &m : materials) {
        if (m.type == MI_BSDF_MASK) d.has_adapters |= 2u;
        if (m.type == MI_BSDF_MIXTURE) for (uint32_t c = 0; c < m.distr && c < 4; ++c) { const uint32_t ci = (uint32_t) (c < 3 ? m.reflectance[c] : m.eta[0]); if (ci < materials.size() && (materials[ci].type == MI_BSDF_NULL || materials[ci].type == MI_BSDF_THINDIELECTRIC)) d.has_adapters |= 2u; }
    }
    const char *noPacket = getenv("MI355PT_NO_PACKET");
    d.packet_n = (nTris <= MI_PACKET_MAX && analyticD.size() <= MI_ANALYTIC_PACKET_MAX && instancesD.empty() && media.empty() && !(noPacket && noPacket[0] == '1')) ? (uint32_t) tris.size() : 0;   // used as a flag
    if (up(&dPacketGroups, packetGroups) | up(&dPacketExact, packetExact)) return 1;
    // participating media: scenes that carry them always walk the tree (the transmittance walk of the volumetric shadow stage is a closest-hit traversal)
    if (!media.empty()) { if (up(&dMedia, mediaD) | up(&dPrimMedia, primMedia)) return 1; }
    d.media = (const MediumD *) dMedia; d.prim_media = (const uint32_t *) dPrimMedia; d.n_media = (uint32_t) mediaD.size(); d.sensor_medium = media.empty() ? -1 : sensorMedium;
    d.packet_groups = (const PacketGroupD *) dPacketGroups; d.packet_exact = (const TriAccelD *) dPacketExact; d.packet_scale = packetScale;
    for (int i = 0; i < 3; ++i) d.packet_gk[i] = packetGK[i];
    committed = true;
    return 0;
}
}  // namespace mi

extern "C" {

int mi_scene_commit(mi_scene *s, uint32_t device) {
    if (!s) return fail(MI_ERR_INVALID, "mi_scene_commit: null scene");
    if ((s->h.idx.empty() && s->h.analytic.empty()) || s->h.materials.empty() || !s->h.haveCamera || !s->h.haveFilm)
        return fail(MI_ERR_INVALID, "mi_scene_commit: geometry (triangles and / or analytic shapes), materials, camera and film must be set first");
    {   uint32_t ng = 0; for (const mi_shape &sh : s->h.shapes) ng = std::max(ng, sh.group);
        for (const mi_shape &sh : s->h.shapes) if (sh.group && sh.emitter >= 0) return fail(MI_ERR_INVALID, "Instancing of emitters is not supported");   // shapegroup.cpp:75-76
        for (const mi_instance &in : s->h.instances) if (in.group >= ng) return fail(MI_ERR_INVALID, "A reference to a 'shapegroup' must be specified!");   // instance.cpp:41-44
    }
    for (const mi_material &m : s->h.materials) {
        const uint32_t tex = (m.flags >> 8) & 0xFFFFu;
        if (tex && tex <= s->h.textures.size() && s->h.textures[tex - 1].type == MI_TEXTURE_BITMAP &&
            (size_t) s->h.textures[tex - 1].first_level + s->h.textures[tex - 1].n_levels > s->h.texLevels.size() / 3) return fail(MI_ERR_INVALID, "mi_scene_commit: bitmap texture without its MIP levels (mi_scene_set_texture_data)");
        if (tex && (tex > s->h.textures.size() || (m.type != MI_BSDF_DIFFUSE && m.type != MI_BSDF_ROUGHDIFFUSE && m.type != MI_BSDF_PLASTIC && m.type != MI_BSDF_ROUGHPLASTIC && m.type != MI_BSDF_DIFFTRANS && m.type != MI_BSDF_MASK && m.type != MI_BSDF_BLEND && m.type != MI_BSDF_BUMPMAP && m.type != MI_BSDF_NORMALMAP)))
            return fail(MI_ERR_UNSUPPORTED, "mi_scene_commit: textures bind to diffuse.reflectance, plastic / roughplastic.diffuseReflectance, difftrans.transmittance, mask.opacity or are the map of a bumpmap / normalmap (and must exist)");
    }
    if (s->h.envTexture >= 0) {       // MIP pyramid of the environment map (camera-ray lookups, envmap.cpp:398-411)
        if (!s->h.envW) return fail(MI_ERR_INVALID, "mi_scene_commit: mi_scene_set_envmap_filter without an environment map");
        if ((size_t) s->h.envTexture >= s->h.textures.size()) return fail(MI_ERR_INVALID, "mi_scene_commit: mi_scene_set_envmap_filter refers to a missing texture record");
        const mi_texture &t = s->h.textures[s->h.envTexture];
        if (t.type != MI_TEXTURE_BITMAP || (size_t) t.first_level + t.n_levels > s->h.texLevels.size() / 3 || s->h.texLevels[(size_t) t.first_level * 3] != s->h.envW || s->h.texLevels[(size_t) t.first_level * 3 + 1] != s->h.envH)
            return fail(MI_ERR_INVALID, "mi_scene_commit: the environment map's pyramid must be a bitmap texture record whose level 0 has the map's size");
    }
    auto wantsTangents = [&](int32_t b) {          // anisotropic roughness, or a bumpmap / normalmap anywhere under the shape's material (their components are EAnisotropic, bumpmap.cpp:99-100)
        if (b < 0 || (size_t) b >= s->h.materials.size()) return false;
        const mi_material *mm = &s->h.materials[b];
        if (mm->type == MI_BSDF_MASK && mm->distr < s->h.materials.size()) mm = &s->h.materials[mm->distr];
        if ((mm->type == MI_BSDF_COATING || mm->type == MI_BSDF_ROUGHCOATING) && mm->distr < s->h.materials.size()) mm = &s->h.materials[mm->distr];
        return (mm->flags & MI_BSDF_FLAG_ANISOTROPIC) != 0 || mm->type == MI_BSDF_BUMPMAP || mm->type == MI_BSDF_NORMALMAP;
    };
    for (const mi_shape &sh : s->h.shapes)         // TriMesh::computeUVTangents (trimesh.cpp:683-692): such BSDFs take their tangents from the texture coordinates
        if (wantsTangents(sh.bsdf) && !((sh.flags & 2u) && !s->h.uv.empty()))
            return fail(MI_ERR_INVALID, "computeUVTangents(): texture coordinates are required to generate tangent vectors. If you want to render with an anisotropic material, please make sure that all associated shapes have valid texture coordinates.");
    for (const mi_material &m : s->h.materials)
        if ((m.type == MI_BSDF_ROUGHPLASTIC || m.type == MI_BSDF_ROUGHCOATING) && (m.k[2] < 2 || m.k[1] < 0 || (size_t) m.k[1] + (size_t) m.k[2] > s->h.materialTables.size()))
            return fail(MI_ERR_INVALID, "mi_scene_commit: roughplastic material without its rough-transmittance slice (mi_scene_set_material_tables)");
    for (const mi_analytic &a : s->h.analytic) {
        if (a.bsdf < 0 || (size_t) a.bsdf >= s->h.materials.size()) return fail(MI_ERR_INVALID, "mi_scene_commit: analytic shape refers to a missing material");
        if (a.emitter >= (int32_t) s->h.emitters.size()) return fail(MI_ERR_INVALID, "mi_scene_commit: analytic shape refers to a missing emitter");
    }
    for (const mi_shape &sh : s->h.shapes) {
        if (sh.bsdf < 0 || (size_t) sh.bsdf >= s->h.materials.size()) return fail(MI_ERR_INVALID, "mi_scene_commit: shape refers to a missing material");
        if (sh.emitter >= (int32_t) s->h.emitters.size()) return fail(MI_ERR_INVALID, "mi_scene_commit: shape refers to a missing emitter");
        if (!(sh.flags & 1u) && s->h.nrm.empty()) return fail(MI_ERR_INVALID, "mi_scene_commit: smooth-shaded mesh without vertex normals (pass faceNormals or normals)");
    }
    for (const mi_emitter &e : s->h.emitters) {
        if (e.type == MI_EMITTER_AREA && (e.shape < 0 || (size_t) e.shape >= s->h.shapes.size() + s->h.analytic.size())) return fail(MI_ERR_INVALID, "mi_scene_commit: area emitter without a shape");
        if (e.type == MI_EMITTER_ENVMAP && s->h.envRGB.empty()) return fail(MI_ERR_INVALID, "mi_scene_commit: envmap emitter listed but mi_scene_set_envmap was not called");
    }
    if (!s->h.media.empty()) {
        if (s->h.shapeMedia.size() != (s->h.shapes.size() + s->h.analytic.size()) * 2u) return fail(MI_ERR_INVALID, "mi_scene_commit: mi_scene_set_media needs one (interior, exterior) pair per mesh and per analytic shape");
        for (size_t i = 0; i < s->h.shapes.size(); ++i)
            if (s->h.shapes[i].group && (s->h.shapeMedia[i * 2] >= 0 || s->h.shapeMedia[i * 2 + 1] >= 0)) return fail(MI_ERR_UNSUPPORTED, "mi_scene_commit: media on the members of a shape group are not implemented");
    }
    if (s->h.emitters.empty()) return fail(MI_ERR_UNSUPPORTED, "mi_scene_commit: scene without emitters (the reference would add a sunsky emitter)");
    ensureSobolTables();
    s->h.commitHost();
    if (s->h.bvhDepth > 32) return fail(MI_ERR_UNSUPPORTED, "mi_scene_commit: BVH deeper than the traversal stack (32)");
    if (s->h.wideBvh && s->h.nodes.size() >= (1u << 23)) return fail(MI_ERR_UNSUPPORTED, "mi_scene_commit: more than 2^23 BVH nodes");
    if (getenv("MI355PT_VERBOSE")) fprintf(stderr, "[mi355pt] %zu triangles, %zu analytic shapes, %zu instances, %zu BVH nodes, depth %d\n", s->h.idx.size() / 3, s->h.analytic.size(), s->h.instances.size(), s->h.nodes.size(), s->h.bvhDepth);
    int devCount = 0; HIPCHK(hipGetDeviceCount(&devCount));
    if ((int) device >= devCount) return fail(MI_ERR_DEVICE, "mi_scene_commit: no such HIP device");
    if (s->h.upload((int) device)) return fail(MI_ERR_DEVICE, std::string("mi_scene_commit: upload failed: ") + hipGetErrorString(hipGetLastError()));
    return MI_OK;
}

// A replica of a committed scene on another (or the same) HIP device: multi-device renders hold one full scene copy per device (the reference ships the
// serialized scene to every worker, src/librender/renderjob.cpp + sched_remote.cpp).  The host-side build (BVH, tables) is reused, only the upload is repeated.
int mi_scene_clone(mi_scene *s, uint32_t device, mi_scene **out) {
    if (!s || !out) return fail(MI_ERR_INVALID, "mi_scene_clone: null argument");
    if (!s->h.committed) return fail(MI_ERR_INVALID, "mi_scene_clone: scene not committed");
    int devCount = 0; HIPCHK(hipGetDeviceCount(&devCount));
    if ((int) device >= devCount) return fail(MI_ERR_DEVICE, "mi_scene_clone: no such HIP device");
    mi_scene *c = new mi_scene();
    c->h = s->h;                                    // inputs + host-derived data
    {   // the copy must not own the source's device allocations
        void **ps[] = {&c->h.dMedia, &c->h.dPrimMedia, &c->h.dPacketGroups, &c->h.dPacketExact, &c->h.dTexLevels, &c->h.dTexTexels, &c->h.dMipLut, &c->h.dTriUV, &c->h.dTextures, &c->h.dMaterialTables, &c->h.dInstances, &c->h.dEmitterX, &c->h.dAnalytic, &c->h.dNodes, &c->h.dTris, &c->h.dShade, &c->h.dI2, &c->h.dNrm, &c->h.dMaterials, &c->h.dEmitters, &c->h.dEmitterCdf, &c->h.dAreaCdf, &c->h.dFilter, &c->h.dSobolM32, &c->h.dSobolVdc, &c->h.dSobolVdcInv, &c->h.dEnvRGB, &c->h.dEnvCols, &c->h.dEnvRows, &c->h.dEnvWeights, &c->h.dEnvGuideRows, &c->h.dEnvGuideCols};
        for (void **p : ps) *p = nullptr;
        c->h.committed = false;
    }
    if (c->h.upload((int) device)) { std::string msg = std::string("mi_scene_clone: upload failed: ") + hipGetErrorString(hipGetLastError()); delete c; return fail(MI_ERR_DEVICE, msg); }
    *out = c; return MI_OK;
}

// ------------------------------------------------------------------------------------------------ render
static int allocQ(std::vector<void *> &allocs, void **p, size_t bytes) {
    HIPCHK(hipMalloc(p, bytes)); allocs.push_back(*p); return MI_OK;
}
#define ALLOC(ptr, type, count) do { void *p_ = nullptr; int rc_ = allocQ(allocs, &p_, sizeof(type) * (size_t) (count)); if (rc_) return rc_; ptr = (type *) p_; } while (0)

static void freePoolQ(Queues &Q, std::vector<void *> &allocs) { for (void *p : allocs) (void) hipFree(p); allocs.clear(); Q = Queues{}; }      // no dangling queue pointer survives
// allocates the queues of ONE pool into an empty Q (freePoolQ first); on failure the caller releases whatever was allocated
static int allocPoolQ(mi_render *r, uint64_t paths, Queues &Q, std::vector<void *> &allocs) {
    auto envU = [](const char *name, uint32_t dflt) { const char *v = getenv(name); return v && v[0] ? (uint32_t) atoi(v) : dflt; };
    // segments of the path pool (each owned by one wave in the shade stage): 16384, or -- where the shade stage sorts a segment's paths by material class and its
    // index list must fit LDS -- as many as keep a segment at ~1000 slots (C3 at 64 M paths: 1916 Msamples/s with 16384 segments, 1983 with 65536)
    uint32_t grid = envU("MI355PT_SEGMENTS", r->scene->h.d.has_roughconductor ? (uint32_t) std::min<uint64_t>(std::max<uint64_t>(16384u, paths / 1024u), 1u << 18) : 16384u);
    uint64_t minGrid = (paths + 63) / 64; if (grid > minGrid) grid = (uint32_t) std::max<uint64_t>(minGrid, 1);
    if (r->rc.integrator != MI_INTEGRATOR_PATH && (paths + grid - 1) / grid > 65472u) grid = (uint32_t) ((paths + 65471u) / 65472u);      // the volumetric stages count two kinds of shadow records per segment in 16 bits each
    uint64_t cap = (paths + grid - 1) / grid; cap = (cap + 63) / 64 * 64;
    r->grid = grid; Q.cap = (uint32_t) cap; Q.n_seg = grid;
    // workgroups launched per stage (each walks segments b, b + grid, ...): sized to the stage's occupancy on 256 CUs
    r->gridExtend = std::min(grid, envU("MI355PT_GRID_EXTEND", 4096u)); r->gridShade = std::min(grid, envU("MI355PT_GRID_SHADE", r->scene->h.d.has_roughconductor ? 512u : 768u));      // 3 workgroups per CU since the diffuse kernels hold 4 waves per SIMD (C2: 512 -> 2970, 640 -> 3022, 768 -> 3078, 896 -> 2890 Msamples/s); the 2-wave microfacet kernels stay at 2
    r->gridShadow = std::min(grid, envU("MI355PT_GRID_SHADOW", 4096u));
    const uint64_t slots = cap * grid;
    if (slots >= (1ull << 28)) return fail(MI_ERR_INVALID, "mi_render_run: a path pool holds fewer than 2^28 slots (the stages address the queues through 32-bit byte offsets); lower planes_per_batch");
    {   // MI355PT_POOL_LIMIT (bytes per pool; tests): behave as if the card had no more room than this -- mi_render_run then falls back to smaller batches
        const char *lim = getenv("MI355PT_POOL_LIMIT");
        if (lim && atoll(lim) > 0 && slots * 224ull > (uint64_t) atoll(lim)) return fail(MI_ERR_DEVICE, "mi_render_run: path pool larger than MI355PT_POOL_LIMIT");
    }
    for (int b = 0; b < 2; ++b) {
        ALLOC(Q.rayO[b], float4, slots); ALLOC(Q.rayD[b], float4, slots);
        ALLOC(Q.st0[b], uint4, slots); ALLOC(Q.st1[b], float4, slots); ALLOC(Q.st2[b], float, slots);
        if (r->scene->h.d.env_constant) ALLOC(Q.st3[b], float, slots); else Q.st3[b] = nullptr;
        ALLOC(Q.count[b], uint32_t, grid);
    }
    if (r->scene->h.d.n_instances) ALLOC(Q.hitInst, int32_t, slots); else Q.hitInst = nullptr;
    // shadow records: 48 B; the volumetric integrators add throughput and BSDF / phase value (80 B); volpath leaves up to two records per path and pass (kernels_volmis.hip)
    const uint64_t shSlots = r->rc.integrator != MI_INTEGRATOR_PATH ? slots * 2 : slots;
    ALLOC(Q.hit, float4, slots); ALLOC(Q.shO, float4, shSlots); ALLOC(Q.shD, float4, shSlots); ALLOC(Q.shC, float4, shSlots);
    if (r->rc.integrator != MI_INTEGRATOR_PATH) { ALLOC(Q.shT, float4, shSlots); ALLOC(Q.shX, float4, shSlots); } else { Q.shT = nullptr; Q.shX = nullptr; }
    ALLOC(Q.acc, float4, slots); ALLOC(Q.pos, float2, slots); ALLOC(Q.shCount, uint32_t, grid);
    ALLOC(Q.counters, unsigned long long, 4);
    ALLOC(Q.ticket, uint32_t, MI_TICKETS);
    Q.stkSpill = nullptr;
    if (mi_fused_walk(r->scene->h.d) && r->scene->h.d.bvh_stack_direct > 10u)      // FZ_LDS_STACK (trace_fused.h) entries live in LDS, the rest of the builder's bound here
        ALLOC(Q.stkSpill, int32_t, (size_t) (r->scene->h.d.bvh_stack_direct - 10u) * mi_fused_grid() * 256u);
    HIPCHK(hipMemset(Q.counters, 0, 32));
    return MI_OK;
}
// (Re)allocates every pool for `paths` paths.  The ray counters survive (they are cleared by mi_render_clear only).  Nothing is valid until ALL pools are
// allocated: on any failure every pool is released, every queue pointer nulled and poolPaths = 0, so a later run / samples call allocates afresh instead of
// launching on freed buffers; the counters saved before the release move into the handle's merged totals.
static int allocPool(mi_render *r, uint64_t paths) {
    unsigned long long keep[mi_render::kMaxPools][4] = {};
    for (int i = 0; i < r->nStreams; ++i) { Queues &Q = r->pool(i); if (Q.counters && hipMemcpy(keep[i], Q.counters, 32, hipMemcpyDeviceToHost) != hipSuccess) { (void) hipGetLastError(); memset(keep[i], 0, 32); } }
    r->poolPaths = 0;
    for (int i = 0; i < r->nStreams; ++i) freePoolQ(r->pool(i), i ? r->allocsx[i - 1] : r->allocs);
    int rc = MI_OK;
    for (int i = 0; i < r->nStreams && !rc; ++i) {
        rc = allocPoolQ(r, paths, r->pool(i), i ? r->allocsx[i - 1] : r->allocs);
        if (!rc && hipMemcpy(r->pool(i).counters, keep[i], 32, hipMemcpyHostToDevice) != hipSuccess) rc = fail(MI_ERR_DEVICE, "mi_render_run: cannot restore the ray counters");
    }
    if (rc) {
        (void) hipGetLastError();
        for (int i = 0; i < r->nStreams; ++i) { freePoolQ(r->pool(i), i ? r->allocsx[i - 1] : r->allocs); r->mergedRays += keep[i][0]; r->mergedShadow += keep[i][1]; r->mergedPathLen += keep[i][2]; }
        return rc;
    }
    r->poolPaths = paths;
    return MI_OK;
}

// LDS bytes k_shade stages for a small scene (shading records, materials, emitters, CDFs); must match mi_launch_shade (above) and shade.h
static size_t smallTableBytes(const mi::SceneHost &h) {
    if (!h.d.small_tables) return 0;
    return 16 + 4 * ((size_t) h.d.n_tris * (4 * MI_SHADE_WORDS) + h.d.n_materials * 16 + h.d.n_emitters * 12 + ((h.d.n_emitters + 4) & ~3u) + h.d.area_cdf_len);
}
int mi_render_create(mi_scene *s, const mi_render_params *p, mi_render **out) {
    if (!s || !p || !out) return fail(MI_ERR_INVALID, "mi_render_create: null argument");
    if (!s->h.committed) return fail(MI_ERR_INVALID, "mi_render_create: scene not committed");
    if (p->rr_depth <= 0) return fail(MI_ERR_INVALID, "'rrDepth' must be set to a value greater than zero!");                       // integrator.cpp:221-222
    if (p->max_depth <= 0 && p->max_depth != -1) return fail(MI_ERR_INVALID, "'maxDepth' must be set to -1 (infinite) or a value greater than zero!");   // :224-225
    if (p->max_depth > 250) return fail(MI_ERR_UNSUPPORTED, "mi_render_create: maxDepth > 250");
    if (p->sampler > 1) return fail(MI_ERR_INVALID, "mi_render_create: unknown sampler");
    if (p->integrator != MI_INTEGRATOR_PATH && (s->h.d.has_adapters & 4u)) return fail(MI_ERR_UNSUPPORTED, "mi_render_create: `coating`, `roughcoating` and `blendbsdf` are implemented for the path integrator only");
    if (p->integrator > MI_INTEGRATOR_VOLPATH) return fail(MI_ERR_UNSUPPORTED, "mi_render_create: integrators path (0), volpath_simple (1) and volpath (2) are implemented");
    const bool vol = p->integrator != MI_INTEGRATOR_PATH;
    if (vol) {       // what the volumetric stages (kernels_vol.hip) are built for
        // a bumpmap / normalmap over a `null` / `thindielectric` has an ENull lobe the transmittance walks would have to evaluate in a perturbed frame (the reference reads an unset shading frame there); plain records, `mask` and `mixturebsdf` are seen through (surfaceNullEval)
        auto nullLobe = [&](uint32_t i) { return i < s->h.materials.size() && (s->h.materials[i].type == MI_BSDF_NULL || s->h.materials[i].type == MI_BSDF_THINDIELECTRIC); };
        for (const mi_material &m : s->h.materials) {
            bool bad = false;
            if ((m.type == MI_BSDF_BUMPMAP || m.type == MI_BSDF_NORMALMAP) && m.distr < s->h.materials.size()) {
                const mi_material &nm = s->h.materials[m.distr]; bad = nullLobe(m.distr);
                if (nm.type == MI_BSDF_MIXTURE) for (uint32_t c = 0; c < nm.distr && c < 4; ++c) bad |= nullLobe((uint32_t) (c < 3 ? nm.reflectance[c] : nm.eta[0]));
            }
            if (bad) return fail(MI_ERR_UNSUPPORTED, "mi_render_create: volpath_simple / volpath with a null / thindielectric BSDF inside a bumpmap / normalmap is not implemented");
        }
        if (p->max_depth > 250) return fail(MI_ERR_UNSUPPORTED, "mi_render_create: maxDepth beyond 250");
    }
    if (p->sampler == MI_SAMPLER_SOBOL) {
        if (!s->h.d.sobol_m32) return fail(MI_ERR_INVALID, "mi_render_create: Sobol tables not loaded before mi_scene_commit (mi_set_sobol_tables)");
        // dimensions consumed: 2 + per bounce (2 NEE + 2 BSDF + 1 RR, + 1 where a BSDF draws from the sampler itself: roughdielectric, EUsesSampler) + the dim-4 skip
        // (sobol.cpp:218-251 aborts beyond the table)
        int depth = p->max_depth < 0 ? 250 : p->max_depth;
        int perBounce = 5; for (const mi_material &m : s->h.materials) if (m.type == MI_BSDF_ROUGHDIELECTRIC) perBounce = 6;
        if (vol) perBounce += 2;      // + the one or two draws of HomogeneousMedium::sampleDistance per iteration
        if (p->max_depth > 0 && (uint32_t) (3 + perBounce * depth) > s->h.d.sobol_dims) return fail(MI_ERR_INVALID, "Lookup dimension exceeds the direction number table size! You may have to reduce the 'maxDepth' parameter of your integrator.");
        if (p->max_depth < 0) return fail(MI_ERR_UNSUPPORTED, "mi_render_create: maxDepth = -1 with the Sobol sampler needs more dimensions than are loaded");
    }
    if (p->sampler == MI_SAMPLER_INDEPENDENT && p->max_depth > 0) {
        // the build-defined independent stream numbers a path's draws with 8 bits: a path that could draw more than 256 values would re-read
        // its own stream from call 0 -- refused by name rather than silently correlated (unbounded depth keeps the documented period)
        int perBounce = 5; for (const mi_material &m : s->h.materials) if (m.type == MI_BSDF_ROUGHDIELECTRIC) perBounce = 6;
        if (vol) perBounce += 2;
        if (2 + perBounce * p->max_depth > 256) return fail(MI_ERR_UNSUPPORTED, "mi_render_create: the independent sampler stream numbers 256 draws per path; maxDepth x draws per bounce exceeds that (use the Sobol sampler or a smaller maxDepth)");
    }
    HIPCHK(hipSetDevice(s->h.device));
    mi_render *r = new mi_render(); r->scene = s; r->p = *p;
    r->sc = s->h.d; if (vol) r->sc.packet_n = 0;      // packet or tree is decided per render: volpath / volpath_simple on a <= 64-triangle scene without media walk its tree
    struct Guard { mi_render *r; ~Guard() { if (r) mi_render_destroy(r); } } guard{r};      // every early return below releases what was created so far
    r->rc.max_depth = p->max_depth; r->rc.rr_depth = p->rr_depth; r->rc.strict_normals = p->strict_normals; r->rc.hide_emitters = p->hide_emitters; r->rc.opacity = p->opacity;
    r->rc.sobol_scramble = 0;
    // volumetric integrators: the radiance-type bits and the sensor's medium of a fresh path (kernels_vol.hip; volpath_simple.cpp:103-104: maxDepth = 1 gathers emitted radiance only)
    {   // Scene::getBSphere() (aabb.cpp:44-47) of the kd-tree box expanded by the sensor's and the point / spot emitters' positions (scene.cpp:394-421)
        float lo[3], hi[3]; for (int i = 0; i < 3; ++i) { lo[i] = s->h.aabbLo[i]; hi[i] = s->h.aabbHi[i]; }
        auto expand = [&](float x, float y, float z) { const float v[3] = {x, y, z}; for (int i = 0; i < 3; ++i) { lo[i] = std::min(lo[i], v[i]); hi[i] = std::max(hi[i], v[i]); } };
        expand(s->h.c2w[3], s->h.c2w[7], s->h.c2w[11]);
        for (const mi_emitter &e : s->h.emitters) if (e.type == MI_EMITTER_POINT || e.type == MI_EMITTER_SPOT || e.type == MI_EMITTER_COLLIMATED) expand(e.to_world[3], e.to_world[7], e.to_world[11]);
        float c[3], d2 = 0; for (int i = 0; i < 3; ++i) { c[i] = (hi[i] + lo[i]) * 0.5f; const float d = c[i] - hi[i]; d2 += d * d; }
        r->rc.alpha_dist = std::sqrt(d2) * 2;
    }
    r->rc.integrator = p->integrator;
    r->rc.state_init = p->integrator == MI_INTEGRATOR_VOLPATH_SIMPLE ? ((1u << 16) | (p->max_depth == 1 ? 0u : (1u << 17)) | (1u << 18) | ((uint32_t) (s->h.d.sensor_medium + 1) << 20))
                     : p->integrator == MI_INTEGRATOR_VOLPATH ? ((1u << 16) | ((uint32_t) (s->h.d.sensor_medium + 1) << 20)) : 0u;      // volpath: ERadiance, nothing else (kernels_volmis.hip)
    if (p->sampler == MI_SAMPLER_SOBOL && p->seed) {          // SobolSampler: a nonzero `scramble` goes through sampleTEA (sobol.cpp:96-102; qmc.h:146-156, 4 rounds)
        uint32_t v0 = (uint32_t) p->seed, v1 = (uint32_t) (p->seed >> 32), sum = 0;
        for (int i = 0; i < 4; ++i) {
            sum += 0x9e3779b9u;
            v0 += ((v1 << 4) + 0xA341316Cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xC8013EA4u);
            v1 += ((v0 << 4) + 0xAD90777Du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7E95761Eu);
        }
        r->rc.sobol_scramble = v0;                           // single precision build: the low 32 bits of (v1 << 32) + v0 (sobolseq.h:87-96)
    }
    r->rc.sampler = p->sampler; r->rc.seed_mix = (uint32_t) p->seed * 0x9E3779B9u; r->rc.inv_sqrt_spp = 1.0f / std::sqrt((float) (p->spp ? p->spp : 1u));
    if (p->sampler == MI_SAMPLER_SOBOL) {
        // fold the direction matrices into 4-bit lookup tables for the dimensions / index bits this render can touch
        int perBounceDims = 5; for (const mi_material &m : s->h.materials) if (m.type == MI_BSDF_ROUGHDIELECTRIC) perBounceDims = 6;
        if (vol) perBounceDims += 2;
        uint32_t sppBits = 0; while ((1ull << sppBits) < p->spp) ++sppBits;
        const uint32_t bits = (s->h.logRes > 1 ? 2 * s->h.logRes : 0) + sppBits + 1;
        const uint32_t nibs = std::max<uint32_t>(8, (bits + 3) / 4), dims      /* at least the eight nibbles of the low index word (pt_device.h sobolBitsNib unrolls them) */ = std::min<uint32_t>(s->h.d.sobol_dims, (uint32_t) (4 + perBounceDims * p->max_depth));
        std::vector<uint32_t> nib((size_t) dims * nibs * 16);
        for (uint32_t dmn = 0; dmn < dims; ++dmn) for (uint32_t n = 0; n < nibs; ++n) for (uint32_t v = 0; v < 16; ++v) {
            uint32_t x = 0; for (uint32_t b = 0; b < 4; ++b) if (((v >> b) & 1u) && 4 * n + b < MI_SOBOL_SIZE) x ^= g_sobolM32[(size_t) dmn * MI_SOBOL_SIZE + 4 * n + b];
            nib[((size_t) dmn * nibs + n) * 16 + v] = x;
        }
        if ((size_t) dims * nibs * 64 + 64 > 144 * 1024) return fail(MI_ERR_UNSUPPORTED, "mi_render_create: the Sobol lookup tables (maxDepth x index bits) exceed the 144 KB of LDS the shading stage may request (reduce maxDepth or spp)");
        HIPCHK(hipMalloc((void **) &r->dNib, nib.size() * 4)); HIPCHK(hipMemcpy(r->dNib, nib.data(), nib.size() * 4, hipMemcpyHostToDevice));
        r->rc.sobol_nib = r->dNib; r->rc.nib_count = nibs; r->rc.nib_dims = dims;
        // look_up tables of k_generate.  index(frame, px, py) = (frame << 2m) ^ Inv * ((px << m | py) ^ Delta * frame)  (sobolseq.h:99-131, all XOR-linear) =
        // F[frame] ^ PX[px] ^ PY[py] with F = (frame << 2m) ^ Inv Delta frame, PX = Inv (px << m), PY = Inv py; and dimension d of the sample is M_d * index,
        // linear as well: every table entry carries {index lo, hi, M_0 * index, M_1 * index}.  The scramble flips pixel bits before the lookup and is XORed
        // into the sample afterwards (k_generate), so the tables do not depend on it.
        const uint32_t m = s->h.logRes;
        if (m > 1) {
            if (2 * m + sppBits > 52 || p->spp > (1u << 22)) return fail(MI_ERR_UNSUPPORTED, "mi_render_create: Sobol index beyond 52 bits (film resolution x sample count)");
            const uint64_t *vdc = g_sobolVdc.data() + (size_t) (m - 1) * MI_SOBOL_SIZE, *vdi = g_sobolVdcInv.data() + (size_t) (m - 1) * MI_SOBOL_SIZE;
            auto mulVec = [&](const uint64_t *cols, uint64_t b) { uint64_t x = 0; for (uint32_t c = 0; b; b >>= 1, ++c) if (b & 1) x ^= cols[c]; return x; };
            auto dimBits = [&](uint32_t dmn, uint64_t index) { uint32_t x = 0; for (uint32_t c = 0; index; index >>= 1, ++c) if (index & 1) x ^= g_sobolM32[(size_t) dmn * MI_SOBOL_SIZE + c]; return x; };
            const size_t res = (size_t) 1 << m, nF = std::max<uint32_t>(p->spp, 1u);
            std::vector<uint32_t> tab((nF + 2 * res) * 4);
            auto put = [&](size_t e, uint64_t idx) { tab[e * 4] = (uint32_t) idx; tab[e * 4 + 1] = (uint32_t) (idx >> 32); tab[e * 4 + 2] = dimBits(0, idx); tab[e * 4 + 3] = dimBits(1, idx); };
            for (size_t f = 0; f < nF; ++f) put(f, ((uint64_t) f << (2 * m)) ^ mulVec(vdi, mulVec(vdc, f)));
            for (size_t x = 0; x < res; ++x) { put(nF + x, mulVec(vdi, (uint64_t) x << m)); put(nF + res + x, mulVec(vdi, (uint64_t) x)); }
            HIPCHK(hipMalloc(&r->dSobolTabs, tab.size() * 4)); HIPCHK(hipMemcpy(r->dSobolTabs, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
            r->rc.sobol_frame = (const uint4 *) r->dSobolTabs; r->rc.sobol_nframes = (uint32_t) nF; r->rc.sobol_px = r->rc.sobol_frame + nF; r->rc.sobol_py = r->rc.sobol_px + res;
        }
    }
    HIPCHK(hipStreamCreate(&r->stream)); HIPCHK(hipEventCreate(&r->evBegin)); HIPCHK(hipEventCreate(&r->evEnd));
    { const char *ns = getenv("MI355PT_STREAMS"); r->nStreams = ns && ns[0] >= '1' && ns[0] <= '4' ? ns[0] - '0' : 2; }
    for (int i = 1; i < r->nStreams; ++i) HIPCHK(hipStreamCreate(&r->streamx[i - 1]));
    for (int i = 0; i < r->nStreams; ++i) { HIPCHK(hipEventCreateWithFlags(&r->filmDone[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&r->joinEv[i], hipEventDisableTiming)); }
    {   // scene tables in LDS: only if they fit next to the Sobol tables, leaving room for the order list of mixed-material scenes (8 KB at the default segment size)
        const size_t nibBytes = p->sampler == MI_SAMPLER_SOBOL ? (size_t) r->rc.nib_dims * r->rc.nib_count * 64 : 16;
        r->ldsTables = s->h.d.small_tables && nibBytes + smallTableBytes(s->h) + (s->h.d.has_roughconductor ? 8192 + 64 : 64) <= 64 * 1024;
    }
    const int W = (int) s->h.width + 2 * s->h.border, H = (int) s->h.height + 2 * s->h.border;
    r->filmFloats = (size_t) W * H * 5;
    HIPCHK(hipMalloc((void **) &r->film, r->filmFloats * 4)); HIPCHK(hipMemset(r->film, 0, r->filmFloats * 4));
    HIPCHK(hipMalloc((void **) &r->spill, r->filmFloats * 4)); HIPCHK(hipMemset(r->spill, 0, r->filmFloats * 4));
    HIPCHK(hipMalloc((void **) &r->layoutTmp, r->filmFloats * 4));
    guard.r = nullptr; *out = r; return MI_OK;
}
void mi_render_destroy(mi_render *r) {
    if (!r) return;
    (void) hipSetDevice(r->scene->h.device);
    for (void *p : r->allocs) (void) hipFree(p);
    for (auto &a : r->allocsx) for (void *p : a) (void) hipFree(p);
    for (hipEvent_t e : r->filmDone) if (e) (void) hipEventDestroy(e);
    for (hipEvent_t e : r->joinEv) if (e) (void) hipEventDestroy(e);
    for (hipStream_t st : r->streamx) if (st) (void) hipStreamDestroy(st);
    if (r->film) (void) hipFree(r->film);
    if (r->spill) (void) hipFree(r->spill);
    if (r->layoutTmp) (void) hipFree(r->layoutTmp);
    if (r->mergeTmp) (void) hipFree(r->mergeTmp);
    if (r->dNib) (void) hipFree(r->dNib);
    if (r->pollHost) (void) hipHostFree(r->pollHost);
    if (r->pollEv) (void) hipEventDestroy(r->pollEv);
    if (r->dSobolTabs) (void) hipFree(r->dSobolTabs);
    for (hipEvent_t e : r->evPool) (void) hipEventDestroy(e);
    if (r->evBegin) (void) hipEventDestroy(r->evBegin);
    if (r->evEnd) (void) hipEventDestroy(r->evEnd);
    if (r->stream) (void) hipStreamDestroy(r->stream);
    delete r;
}
int mi_render_clear(mi_render *r) {
    if (!r) return fail(MI_ERR_INVALID, "mi_render_clear: null"); HIPCHK(hipSetDevice(r->scene->h.device));
    HIPCHK(hipMemsetAsync(r->film, 0, r->filmFloats * 4, r->stream)); HIPCHK(hipMemsetAsync(r->spill, 0, r->filmFloats * 4, r->stream));
    if (r->q.counters) HIPCHK(hipMemsetAsync(r->q.counters, 0, 32, r->stream));
    for (Queues &Q : r->qx) if (Q.counters) HIPCHK(hipMemsetAsync(Q.counters, 0, 32, r->stream));
    HIPCHK(hipStreamSynchronize(r->stream)); r->samplesTotal = 0; r->mergedRays = r->mergedShadow = r->mergedPathLen = r->mergedSamples = 0; r->cancel.store(0); return MI_OK;
}
void mi_render_cancel(mi_render *r) { if (r) r->cancel.store(1); }
int mi_render_set_profiling(mi_render *r, int enabled) { if (!r) return fail(MI_ERR_INVALID, "null"); r->profiling = enabled != 0; return MI_OK; }

static void mark(mi_render *r, int tag, size_t &used, hipStream_t st = nullptr) {
    if (!r->profiling || (st && st != r->stream)) return;      // stage events are recorded on the first stream only
    if (used >= r->evPool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; r->evPool.push_back(e); r->evTag.push_back(0); }
    (void) hipEventRecord(r->evPool[used], r->stream); r->evTag[used] = tag; ++used;
}

// trace one batch: paths = tile pixels x planes (or an explicit list), all bounces
static int traceBatch(mi_render *r, const BatchDesc &bd, const uint32_t *list, size_t &evUsed, int pool = 0) {
    const DScene &sc = r->sc; hipStream_t st = r->poolStream(pool); Queues &Q = r->pool(pool);
    (void) list;
    mark(r, 0, evUsed, st);
    const bool fused = r->rc.integrator == MI_INTEGRATOR_PATH && mi_fused_walk(sc);      // trace_fused.h: persistent waves fetch segments through per-launch tickets
    if (fused) HIPCHK(hipMemsetAsync(Q.ticket, 0, MI_TICKETS * sizeof(uint32_t), st));
    mi_launch_generate(sc, r->rc, Q, bd, r->grid, st);
    int buf = 0; const int maxDepth = r->rc.max_depth > 0 ? r->rc.max_depth : 250;
    for (int depth = 1; depth <= maxDepth; ++depth) {
        mark(r, 1, evUsed, st);
        if (fused && 2 * depth + 1 < MI_TICKETS) mi_launch_extend_fused(sc, Q, buf, Q.ticket + 2 * depth, st); else mi_launch_extend(sc, Q, buf, r->gridExtend, st);
        ++r->launchesAll;
        if (r->rc.integrator != MI_INTEGRATOR_PATH) {      // the same loop over media: its own shade and shadow stages (kernels_vol.hip, kernels_volmis.hip)
            const size_t lds = r->rc.sampler == MI_SAMPLER_SOBOL ? (size_t) r->rc.nib_dims * r->rc.nib_count * 64 : 16; const bool mis = r->rc.integrator == MI_INTEGRATOR_VOLPATH;
            mark(r, 2, evUsed, st); (mis ? mi_launch_shade_volmis : mi_launch_shade_vol)(sc, r->rc, Q, buf, r->gridShade, lds, st);
            if (depth < maxDepth || (depth == 1 && r->rc.opacity)) { mark(r, 3, evUsed, st); (mis ? mi_launch_shadow_volmis : mi_launch_shadow_vol)(sc, Q, r->gridShadow, st); }      // (depth 1 with EOpacity: the alpha walks)
        } else {
        if (depth == 1 && sc.env_texture && !r->rc.hide_emitters) mi_launch_env_primary(sc, r->rc, Q, buf, r->gridExtend, st);   // camera rays that see the sky: filtered lookup (envmap.cpp:398-411)
        mark(r, 2, evUsed, st); mi_launch_shade(sc, r->ldsTables, r->rc, Q, buf, r->gridShade, st);
        if (depth < maxDepth) { mark(r, 3, evUsed, st); if (fused && 2 * depth + 1 < MI_TICKETS) mi_launch_shadow_fused(sc, Q, Q.ticket + 2 * depth + 1, st); else mi_launch_shadow(sc, Q, r->gridShadow, st); }
        }
        buf ^= 1;
        if (r->rc.max_depth < 0 && (depth % 4) == 0) {   // unbounded depth: poll the survivor counts every few bounces
            // survivor counts land in a pinned buffer (a true asynchronous copy), the host waits on an event of THIS stream only -- the other pool's stream keeps running
            if (r->pollWords < r->grid) { if (r->pollHost) (void) hipHostFree(r->pollHost); r->pollHost = nullptr; HIPCHK(hipHostMalloc((void **) &r->pollHost, (size_t) r->grid * 4, hipHostMallocDefault)); r->pollWords = r->grid; }
            if (!r->pollEv) HIPCHK(hipEventCreateWithFlags(&r->pollEv, hipEventDisableTiming));
            HIPCHK(hipMemcpyAsync(r->pollHost, Q.count[buf], (size_t) r->grid * 4, hipMemcpyDeviceToHost, st)); HIPCHK(hipEventRecord(r->pollEv, st)); HIPCHK(hipEventSynchronize(r->pollEv));
            uint64_t alive = 0; for (uint32_t k = 0; k < r->grid; ++k) alive += r->pollHost[k];
            if (!alive) break;
        }
    }
    mark(r, 0, evUsed, st);
    HIPCHK(hipGetLastError());
    return MI_OK;
}

int mi_render_run(mi_render *r, mi_tile tile, uint32_t s0, uint32_t s1) { return mi_render_run_rows(r, tile, 1, s0, s1); }

int mi_render_run_rows(mi_render *r, mi_tile tile, uint32_t rowStride, uint32_t s0, uint32_t s1) {
    if (!r) return fail(MI_ERR_INVALID, "mi_render_run: null");
    if (rowStride == 0) return fail(MI_ERR_INVALID, "mi_render_run_rows: row stride must be >= 1");
    const mi::SceneHost &h = r->scene->h;
    if (tile.x1 <= tile.x0 || tile.y1 <= tile.y0 || tile.x1 > h.width || tile.y1 > h.height) return fail(MI_ERR_INVALID, "mi_render_run: tile outside the film");
    if (s1 < s0 || s1 > r->p.spp) return fail(MI_ERR_INVALID, "mi_render_run: sample range outside [0, spp]");
    if (r->p.sampler == MI_SAMPLER_INDEPENDENT && s1 > (1u << 24)) return fail(MI_ERR_INVALID, "mi_render_run: independent stream supports < 2^24 samples per pixel");
    HIPCHK(hipSetDevice(h.device));
    const uint32_t nrows = (tile.y1 - tile.y0 + rowStride - 1) / rowStride;          // rows y0, y0 + stride, ... below y1
    const uint32_t npix = (tile.x1 - tile.x0) * nrows;
    uint32_t planes = r->p.planes_per_batch;
    if (!planes) { static const uint64_t target = [] { const char *v = getenv("MI355PT_BATCH_PATHS"); return v && atoll(v) > 0 ? (uint64_t) atoll(v) : (uint64_t) (64u << 20); }(); planes = (uint32_t) std::max<uint64_t>(1, target / npix); }   // ~64 M paths in flight per pool (measured: 16 M 2873, 32 M 3025, 64 M 3055, 128 M 2993 Msamples/s on C2; C4 483 / 501 / 509 / 511)
    if (planes > s1 - s0) planes = std::max<uint32_t>(1, s1 - s0);
    if (!r->p.planes_per_batch && s1 - s0 >= (uint32_t) r->nStreams) {      // automatic batches: as many as keep every path pool / stream busy, equally sized (a job of one
        const uint32_t total = s1 - s0, ns = (uint32_t) r->nStreams;        // 64 M batch would run on one stream: 1569 instead of 1712 Msamples/s on a 32-spp 1080p frame)
        uint32_t nb = (total + planes - 1) / planes; nb = (nb + ns - 1) / ns * ns;
        planes = (total + nb - 1) / nb;
    }
    uint64_t need = (uint64_t) npix * planes;
    if (need > 0xFFFFFF00ull) return fail(MI_ERR_INVALID, "mi_render_run: batch larger than 2^32 paths");
    while (need > r->poolPaths) {      // the pool only grows: a short last batch or a smaller tile reuses it
        int rc = allocPool(r, need);
        if (!rc) break;
        // not enough free device memory for pools of this size (another process on the card, a smaller part): halve the automatic batch and try again
        (void) hipGetLastError(); r->poolPaths = 0;
        if (r->p.planes_per_batch || planes <= 1) return rc;
        planes = (planes + 1) / 2; need = (uint64_t) npix * planes;
    }
    size_t evUsed = 0; r->launchesAll = 0;
    HIPCHK(hipEventRecord(r->evBegin, r->stream));
    const uint32_t nBatches = (s1 - s0 + planes - 1) / planes;
    const int nPools = (int) std::min<uint32_t>((uint32_t) r->nStreams, std::max(nBatches, 1u));      // more than one batch: round-robin over the pools / streams
    if (nPools > 1) HIPCHK(hipEventRecord(r->joinEv[0], r->stream));
    for (int i = 1; i < nPools; ++i) HIPCHK(hipStreamWaitEvent(r->poolStream(i), r->joinEv[0], 0));
    int batch = 0, lastFilmPool = -1;
    for (uint32_t s = s0; s < s1; s += planes, ++batch) {
        if (r->cancel.exchange(0)) { HIPCHK(hipDeviceSynchronize()); return fail(MI_CANCELLED, "render cancelled"); }      // consumed where it is observed
        const int pool = batch % nPools; hipStream_t st = r->poolStream(pool);
        BatchDesc bd{}; bd.tile = tile; bd.n_pix = npix; bd.n_planes = std::min(planes, s1 - s); bd.sample_begin = s; bd.n_paths = (uint64_t) npix * bd.n_planes; bd.list = nullptr; bd.row_stride = rowStride;
        int rc = traceBatch(r, bd, nullptr, evUsed, pool); if (rc) return rc;
        // film accumulation stays in batch order (own-pixel sums are plain read-modify-writes): wait for the previous batch's film kernel
        if (nPools > 1 && lastFilmPool >= 0) HIPCHK(hipStreamWaitEvent(st, r->filmDone[lastFilmPool], 0));
        mi_launch_film(h.d, r->pool(pool), bd, r->film, r->spill, st);
        if (nPools > 1) { HIPCHK(hipEventRecord(r->filmDone[pool], st)); lastFilmPool = pool; }
        r->samplesTotal += bd.n_paths;
    }
    for (int i = 1; i < nPools; ++i) { HIPCHK(hipEventRecord(r->joinEv[i], r->poolStream(i))); HIPCHK(hipStreamWaitEvent(r->stream, r->joinEv[i], 0)); }
    mark(r, 0, evUsed);
    HIPCHK(hipEventRecord(r->evEnd, r->stream));
    HIPCHK(hipStreamSynchronize(r->stream));
    HIPCHK(hipGetLastError());
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, r->evBegin, r->evEnd)); r->stats.render_ms = ms;
    r->stats.extend_ms = r->stats.shade_ms = r->stats.shadow_ms = r->stats.other_ms = 0; r->stats.extend_launches = 0;
    if (r->profiling) {
        for (size_t i = 0; i + 1 < evUsed; ++i) {
            float t = 0; if (hipEventElapsedTime(&t, r->evPool[i], r->evPool[i + 1]) != hipSuccess) continue;
            switch (r->evTag[i]) { case 1: r->stats.extend_ms += t; r->stats.extend_launches++; break; case 2: r->stats.shade_ms += t; break; case 3: r->stats.shadow_ms += t; break; default: r->stats.other_ms += t; }
        }
    }
    return MI_OK;
}

int mi_render_stats(mi_render *r, mi_stats *out) {
    if (!r || !out) return fail(MI_ERR_INVALID, "mi_render_stats: null");
    HIPCHK(hipSetDevice(r->scene->h.device));
    unsigned long long c[4] = {0, 0, 0, 0};
    if (r->q.counters) HIPCHK(hipMemcpy(c, r->q.counters, 32, hipMemcpyDeviceToHost));
    for (Queues &Q : r->qx) if (Q.counters) { unsigned long long c2[4]; HIPCHK(hipMemcpy(c2, Q.counters, 32, hipMemcpyDeviceToHost)); for (int i = 0; i < 4; ++i) c[i] += c2[i]; }
    r->stats.rays = c[0] + r->mergedRays; r->stats.shadow_rays = c[1] + r->mergedShadow; r->stats.path_length_sum = c[2] + r->mergedPathLen; r->stats.samples = r->samplesTotal + r->mergedSamples; r->stats.extend_rays = c[0]; r->stats.extend_launches_all = r->launchesAll;
    *out = r->stats; return MI_OK;
}

int mi_render_film_size(mi_render *r, int layout, uint32_t *height, uint32_t *width, uint32_t *channels, uint32_t *border) {
    if (!r || layout < 0 || layout > 2) return fail(MI_ERR_INVALID, "mi_render_film_size: bad argument");
    const mi::SceneHost &h = r->scene->h; const uint32_t b = (uint32_t) h.border;
    if (height) *height = layout == 2 ? h.height : h.height + 2 * b;
    if (width) *width = layout == 2 ? h.width : h.width + 2 * b;
    if (channels) *channels = layout == 0 ? 5 : (layout == 1 ? 4 : 3);
    if (border) *border = layout == 2 ? 0 : b;
    return MI_OK;
}
static size_t layoutFloats(mi_render *r, int layout) { uint32_t hh, ww, cc, bb; mi_render_film_size(r, layout, &hh, &ww, &cc, &bb); return (size_t) hh * ww * cc; }
int mi_render_read_film_device(mi_render *r, int layout, void *dev) {
    if (!r || !dev || layout < 0 || layout > 2) return fail(MI_ERR_INVALID, "mi_render_read_film_device: bad argument");
    const mi::SceneHost &h = r->scene->h; HIPCHK(hipSetDevice(h.device));
    mi_launch_film_layout(r->film, r->spill, (float *) dev, (int) h.width + 2 * h.border, (int) h.height + 2 * h.border, h.border, layout, r->stream);
    HIPCHK(hipStreamSynchronize(r->stream)); HIPCHK(hipGetLastError());
    return MI_OK;
}
int mi_render_read_film(mi_render *r, int layout, float *host) {
    if (!r || !host) return fail(MI_ERR_INVALID, "mi_render_read_film: null");
    int rc = mi_render_read_film_device(r, layout, r->layoutTmp); if (rc) return rc;
    HIPCHK(hipMemcpy(host, r->layoutTmp, layoutFloats(r, layout) * 4, hipMemcpyDeviceToHost));
    return MI_OK;
}

// dst += src (raw film sums: own-pixel planes and spill planes).  The merge step of a multi-device render, where every device renders its own rows of the
// frame (mi_render_run_rows) into its own film: the reference merges worker blocks with Film::put under a mutex (src/librender/renderproc.cpp:142-149).
// Same device: one add kernel; different devices: peer copy (xGMI, peer access enabled once per pair) into a staging buffer on dst's device, then the add; devices
// without peer access: staged through the host.  Both renders must be idle.  The host mirror calls this along a reduction tree (integrator_host.cpp).
int mi_render_merge_film(mi_render *dst, mi_render *src) {
    if (!dst || !src || dst == src) return fail(MI_ERR_INVALID, "mi_render_merge_film: two different render handles are needed");
    if (dst->filmFloats != src->filmFloats) return fail(MI_ERR_INVALID, "mi_render_merge_film: the two renders have different films");
    const int dd = dst->scene->h.device, sd = src->scene->h.device; const size_t n = dst->filmFloats;
    HIPCHK(hipSetDevice(dd));
    const float *sFilm = src->film, *sSpill = src->spill;
    if (dd != sd) {
        // Peer access is set up once per (destination, source) pair: where the two devices can reach each other (xGMI inside a node) the add kernel's input is copied
        // device to device; where they cannot, the film is staged through a host buffer -- never silently assumed.
        if (!dst->mergeTmp) HIPCHK(hipMalloc((void **) &dst->mergeTmp, 2 * n * 4));
        int can = 0; HIPCHK(hipDeviceCanAccessPeer(&can, dd, sd));
        if (can) {
            static std::mutex peerMutex; static std::set<std::pair<int, int> > enabled;
            { std::lock_guard<std::mutex> l(peerMutex);
              if (!enabled.count({dd, sd})) { hipError_t e = hipDeviceEnablePeerAccess(sd, 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(MI_ERR_DEVICE, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e)); (void) hipGetLastError(); enabled.insert({dd, sd}); } }
            HIPCHK(hipMemcpyPeerAsync(dst->mergeTmp, dd, src->film, sd, n * 4, dst->stream));
            HIPCHK(hipMemcpyPeerAsync(dst->mergeTmp + n, dd, src->spill, sd, n * 4, dst->stream));
        } else {
            std::vector<float> host(2 * n);
            HIPCHK(hipSetDevice(sd)); HIPCHK(hipMemcpy(host.data(), src->film, n * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(host.data() + n, src->spill, n * 4, hipMemcpyDeviceToHost));
            HIPCHK(hipSetDevice(dd)); HIPCHK(hipMemcpy(dst->mergeTmp, host.data(), 2 * n * 4, hipMemcpyHostToDevice));
        }
        sFilm = dst->mergeTmp; sSpill = dst->mergeTmp + n;
    }
    mi_launch_film_add(dst->film, sFilm, n, dst->stream); mi_launch_film_add(dst->spill, sSpill, n, dst->stream);
    HIPCHK(hipStreamSynchronize(dst->stream)); HIPCHK(hipGetLastError());
    unsigned long long c[4]; mi_stats st{};      // the merged handle reports the sum of the ray counters too
    (void) c; if (mi_render_stats(src, &st) == MI_OK) { dst->mergedRays += st.rays; dst->mergedShadow += st.shadow_rays; dst->mergedPathLen += st.path_length_sum; dst->mergedSamples += st.samples; }
    return MI_OK;
}

int mi_render_samples(mi_render *r, const uint32_t *pairs, uint64_t n, float *outLi) {
    if (!r || !pairs || !outLi || !n) return fail(MI_ERR_INVALID, "mi_render_samples: null argument");
    const mi::SceneHost &h = r->scene->h; HIPCHK(hipSetDevice(h.device));
    for (uint64_t i = 0; i < n; ++i) if (pairs[i * 3] >= h.width || pairs[i * 3 + 1] >= h.height) return fail(MI_ERR_INVALID, "mi_render_samples: pixel outside the film");
    if (n > r->poolPaths) { int rc = allocPool(r, n); if (rc) return rc; }
    uint32_t *dList = nullptr; float *dOut = nullptr; uint32_t *dSlots = nullptr;
    HIPCHK(hipMalloc((void **) &dList, n * 12)); HIPCHK(hipMalloc((void **) &dOut, n * 12)); HIPCHK(hipMalloc((void **) &dSlots, n * 4));
    HIPCHK(hipMemcpy(dList, pairs, n * 12, hipMemcpyHostToDevice));
    std::vector<uint32_t> slots(n); for (uint64_t i = 0; i < n; ++i) slots[i] = (uint32_t) i;
    HIPCHK(hipMemcpy(dSlots, slots.data(), n * 4, hipMemcpyHostToDevice));
    BatchDesc bd{}; bd.tile = mi_tile{0, 0, h.width, h.height}; bd.n_pix = (uint32_t) n; bd.n_planes = 1; bd.sample_begin = 0; bd.n_paths = n; bd.list = dList; bd.row_stride = 1;
    size_t evUsed = 0; bool prof = r->profiling; r->profiling = false;
    if (!r->q.counters) return fail(MI_ERR_DEVICE, "mi_render_samples: no path pool");
    unsigned long long keep[4]; HIPCHK(hipMemcpy(keep, r->q.counters, 32, hipMemcpyDeviceToHost));     // the parity entry point leaves the ray counters untouched
    int rc = traceBatch(r, bd, dList, evUsed); r->profiling = prof;
    if (!rc) { hipError_t e = hipStreamSynchronize(r->stream); if (e == hipSuccess) e = hipMemcpy(r->q.counters, keep, 32, hipMemcpyHostToDevice); if (e != hipSuccess) rc = fail(MI_ERR_DEVICE, hipGetErrorString(e)); }
    if (!rc) { mi_launch_gather_samples(r->q, dSlots, n, dOut, r->stream); hipError_t e = hipStreamSynchronize(r->stream); if (e != hipSuccess) rc = fail(MI_ERR_DEVICE, hipGetErrorString(e)); }
    if (!rc) { hipError_t e = hipMemcpy(outLi, dOut, n * 12, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = fail(MI_ERR_DEVICE, hipGetErrorString(e)); }
    (void) hipFree(dList); (void) hipFree(dOut); (void) hipFree(dSlots);
    return rc;
}

// ------------------------------------------------------------------------------------------------ unit-level device entry points
}  // extern "C"
template <typename F> static int withBuffers(const void *in, size_t inBytes, void *out, size_t outBytes, F f) {
    void *dIn = nullptr, *dOut = nullptr;
    HIPCHK(hipMalloc(&dIn, inBytes)); HIPCHK(hipMalloc(&dOut, outBytes));
    HIPCHK(hipMemcpy(dIn, in, inBytes, hipMemcpyHostToDevice));
    f(dIn, dOut);
    hipError_t e = hipDeviceSynchronize(); if (e == hipSuccess) e = hipMemcpy(out, dOut, outBytes, hipMemcpyDeviceToHost);
    (void) hipFree(dIn); (void) hipFree(dOut);
    if (e != hipSuccess) return fail(MI_ERR_DEVICE, hipGetErrorString(e));
    return MI_OK;
}
extern "C" {
int mi_debug_intersect_inst(mi_scene *s, const float *rays, uint64_t n, int anyHit, float *out, int32_t *outInst) {
    if (!s || !s->h.committed || !rays || !out || !n) return fail(MI_ERR_INVALID, "mi_debug_intersect: bad argument");
    HIPCHK(hipSetDevice(s->h.device));
    void *dInst = nullptr; if (outInst) HIPCHK(hipMalloc(&dInst, n * 4));
    int rc = withBuffers(rays, n * 32, out, n * 16, [&](void *i, void *o) { mi_launch_debug_intersect(s->h.d, (const float *) i, n, anyHit, (float *) o, (int *) dInst, nullptr); });
    if (!rc && outInst) { hipError_t e = hipMemcpy(outInst, dInst, n * 4, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = fail(MI_ERR_DEVICE, hipGetErrorString(e)); }
    if (dInst) (void) hipFree(dInst);
    return rc;
}
int mi_scene_ray_intersect(mi_scene *s, const float *rays, uint64_t n, mi_intersection *out) {
    if (!s || !s->h.committed || !rays || !out || !n) return fail(MI_ERR_INVALID, "mi_scene_ray_intersect: bad argument (the scene must be committed)");
    HIPCHK(hipSetDevice(s->h.device));
    return withBuffers(rays, n * 32, out, n * sizeof(mi_intersection), [&](void *i, void *o) { mi_launch_ray_intersect(s->h.d, (const float *) i, n, (mi_intersection *) o, nullptr); });
}
int mi_debug_intersect(mi_scene *s, const float *rays, uint64_t n, int anyHit, float *out) { return mi_debug_intersect_inst(s, rays, n, anyHit, out, nullptr); }
int mi_debug_sobol(mi_scene *s, const uint32_t *in, uint64_t n, uint32_t ndims, uint64_t *outIdx, float *outVals) {
    if (!s || !s->h.committed || !in || !outIdx || !outVals || !n || !s->h.d.sobol_m32 || ndims > s->h.d.sobol_dims) return fail(MI_ERR_INVALID, "mi_debug_sobol: bad argument");
    HIPCHK(hipSetDevice(s->h.device));
    void *dIdx = nullptr; HIPCHK(hipMalloc(&dIdx, n * 8));
    int rc = withBuffers(in, n * 12, outVals, n * ndims * 4, [&](void *i, void *o) { mi_launch_debug_sobol(s->h.d, (const uint32_t *) i, n, ndims, (unsigned long long *) dIdx, (float *) o, nullptr); });
    if (!rc) { hipError_t e = hipMemcpy(outIdx, dIdx, n * 8, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = fail(MI_ERR_DEVICE, hipGetErrorString(e)); }
    (void) hipFree(dIdx); return rc;
}
int mi_debug_sincosf(const float *x, uint64_t n, float *out) {
    if (!x || !out || !n) return fail(MI_ERR_INVALID, "mi_debug_sincosf: bad argument");
    return withBuffers(x, n * 4, out, n * 8, [&](void *i, void *o) { mi_launch_debug_sincosf((const float *) i, n, (float *) o, nullptr); });
}
int mi_debug_libm(int fn, const float *x, const float *y, uint64_t n, float *out) {
    if (!x || !out || !n || fn < 0 || fn > 6 || ((fn == 2 || fn == 5) && !y)) return fail(MI_ERR_INVALID, "mi_debug_libm: bad argument");
    std::vector<float> xy(2 * n); memcpy(xy.data(), x, n * 4); if (y) memcpy(xy.data() + n, y, n * 4);
    return withBuffers(xy.data(), n * 8, out, n * 4, [&](void *i, void *o) { mi_launch_debug_libm(fn, (const float *) i, y ? (const float *) i + n : nullptr, n, (float *) o, nullptr); });
}
int mi_debug_camera_rays(mi_scene *s, const float *pos, uint64_t n, float *out) {
    if (!s || !s->h.committed || !pos || !out || !n) return fail(MI_ERR_INVALID, "mi_debug_camera_rays: bad argument");
    HIPCHK(hipSetDevice(s->h.device));
    return withBuffers(pos, n * 8, out, n * 32, [&](void *i, void *o) { mi_launch_debug_camera(s->h.d, (const float *) i, n, (float *) o, nullptr); });
}

}  // extern "C"
