/* include/mi355pt.h -- C-ABI of libmi355pt.so: the MI355X (gfx950) wavefront path tracer behind the reference's
 * Integrator / ResponsiveIntegrator boundary.
 *
 * Plain C: opaque handles, plain pointers and sizes, every call returns an int status (0 = MI_OK) and leaves a
 * message for mi_last_error().  No exceptions cross this boundary.  What each entry point replaces in the reference
 * (paths relative to the reference tree) is cited next to it; the adapter plugin (mitsuba-im_amd/csrc/adapter/path_hip.cpp,
 * INTEGRATION.md) fills these calls from a live mitsuba::Scene.
 */
#ifndef MI355PT_H
#define MI355PT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_ERR_INVALID 1     /* bad argument / call order            */
#define MI_ERR_DEVICE 2      /* HIP error (message has the HIP text)  */
#define MI_ERR_UNSUPPORTED 3 /* scene feature outside the hot path    */
#define MI_CANCELLED 4       /* mi_render_cancel() was observed       */

typedef struct mi_scene mi_scene;
typedef struct mi_render mi_render;

/* One mesh of the flattened scene = one mitsuba::TriMesh (include/mitsuba/render/trimesh.h:122-160). */
typedef struct {
    uint32_t first_tri, tri_count, first_vert, vert_count;
    int32_t bsdf;        /* index into the material table (Shape::getBSDF, include/mitsuba/render/shape.h) */
    int32_t emitter;     /* index into the emitter table or -1 (Shape::getEmitter)                          */
    uint32_t flags;      /* bit0: face normals (TriMesh "faceNormals", src/librender/trimesh.cpp:70); bit1: the mesh has texture coordinates
                            (`uv` of mi_scene_set_triangles): its.uv is interpolated from them and the triangle's UV tangent replaces the first
                            edge as dpdu (TriMesh::computeUVTangents, trimesh.cpp:683-736; skdtree.h:373-376, 402-408) */
    uint32_t group;      /* 0: the mesh is a scene shape; g > 0: it is a member of shape group g - 1 (src/shapes/shapegroup.cpp) and
                            appears in the scene only through mi_instance records; group members cannot be emitters (shapegroup.cpp:75-76) */
} mi_shape;

/* src/shapes/instance.cpp: one placement of shape group `group`; to_world = the instance transform, to_object = its inverse as the
 * reference computes it.  Groups hold triangle meshes; nested instancing is not permitted (shapegroup.cpp:71-72). */
typedef struct { uint32_t group; uint32_t pad[3]; float to_world[16], to_object[16]; } mi_instance;
/* Participating medium: HomogeneousMedium (src/medium/homogeneous.cpp) with its phase function (src/phase/isotropic.cpp, src/phase/hg.cpp).  sigma_a / sigma_s:
 * the final coefficients (`scale` and presets folded in by the caller, medium.cpp:27-37); strategy: MI_MEDIUM_BALANCE / SINGLE / MANUAL (`maximum` is refused);
 * sampling_density, medium_sampling_weight: as the constructor derives them (homogeneous.cpp:168-222); phase: MI_PHASE_ISOTROPIC / MI_PHASE_HG with mean cosine g */
#define MI_MEDIUM_BALANCE 0
#define MI_MEDIUM_SINGLE 1
#define MI_MEDIUM_MANUAL 2
#define MI_PHASE_ISOTROPIC 0
#define MI_PHASE_HG 1
typedef struct { float sigma_a[3], sigma_s[3]; uint32_t strategy; float sampling_density, medium_sampling_weight; uint32_t phase; float g; uint32_t pad; } mi_medium;
#define MI_INTEGRATOR_PATH 0            /* MIPathTracer, src/integrators/path/path.cpp */
#define MI_INTEGRATOR_VOLPATH_SIMPLE 1  /* SimpleVolumetricPathTracer, src/integrators/path/volpath_simple.cpp */
#define MI_INTEGRATOR_VOLPATH 2         /* VolumetricPathTracer, src/integrators/path/volpath.cpp: multiple importance sampling, emitters found through index-matched boundaries */

#define MI_BSDF_DIFFUSE 0         /* src/bsdfs/diffuse.cpp: reflectance                                                              */
#define MI_BSDF_ROUGHCONDUCTOR 1  /* src/bsdfs/roughconductor.cpp + microfacet.h: alpha (alphaU), distr, eta, k, specular; flags: sampleVisible, anisotropic */
#define MI_BSDF_CONDUCTOR 2       /* src/bsdfs/conductor.cpp: eta, k (already divided by the exterior index), specular               */
#define MI_BSDF_DIELECTRIC 3      /* src/bsdfs/dielectric.cpp: eta[0] = intIOR / extIOR, specular = specularReflectance, reflectance = specularTransmittance */
#define MI_BSDF_PLASTIC 4         /* src/bsdfs/plastic.cpp: eta[0] = intIOR / extIOR, specular = specularReflectance, reflectance = diffuseReflectance,
                                     k[0] = fresnelDiffuseReflectance(1 / eta, false) (SmoothPlastic::m_fdrInt, plastic.cpp:200)       */
#define MI_BSDF_ROUGHDIELECTRIC 5 /* src/bsdfs/roughdielectric.cpp: alpha, distr, eta[0], specular = specularReflectance, reflectance = specularTransmittance */
#define MI_BSDF_DIFFTRANS 6       /* src/bsdfs/difftrans.cpp: reflectance = transmittance                                            */
#define MI_BSDF_ROUGHPLASTIC 7    /* src/bsdfs/roughplastic.cpp: alpha, distr, eta[0], specular, reflectance = diffuseReflectance; k[0] = internal diffuse rough
                                     transmittance (m_internalRoughTransmittance->evalDiffuse(alpha), roughplastic.cpp:372), k[1] / k[2] = offset / length of the
                                     external rough-transmittance slice (RoughTransmittance after setEta + setAlpha, src/bsdfs/rtrans.h:292-388) in the table
                                     buffer of mi_scene_set_material_tables */
#define MI_BSDF_THINDIELECTRIC 8  /* src/bsdfs/thindielectric.cpp: eta[0] = intIOR / extIOR, specular = specularReflectance, reflectance = specularTransmittance (ENull transmission) */
#define MI_BSDF_MASK 9            /* src/bsdfs/mask.cpp: reflectance = opacity (constant or a bound texture), distr = index of the nested material record; ENull pass-through lobe */
#define MI_BSDF_MIXTURE 10        /* src/bsdfs/mixturebsdf.cpp: distr = number of children (2..4); their material record indices as numbers in reflectance[0..2], eta[0],
                                     their weights in k[0..2], specular[0] (rescaled when they sum to more than one, ensureEnergyConservation).  Children are plain
                                     BSDF records without textures, at most one of them with a Dirac delta component */
#define MI_BSDF_BUMPMAP 11        /* src/bsdfs/bumpmap.cpp: distr = index of the nested record (a plain BSDF or a mixturebsdf), bound texture (MI_BSDF_TEXTURE) = the displacement,
                                     alpha = the factor of an enclosing <texture type="scale"> (1 = none); meshes need texture coordinates */
#define MI_BSDF_NORMALMAP 12      /* src/bsdfs/normalmap.cpp: distr = nested record, bound texture = tangent-space normals (rgb = 0.5 + 0.5 n).  Adapters nest in the
                                     order mask -> bumpmap / normalmap -> mixturebsdf -> plain BSDF */
#define MI_BSDF_NULL 13           /* src/bsdfs/null.cpp: the index-matched boundary of a participating medium (straight pass-through, ENull; no parameters) */
#define MI_BSDF_ROUGHDIFFUSE 14    /* src/bsdfs/roughdiffuse.cpp (Oren-Nayar): reflectance (constant or a bound texture), alpha, distr = 1: useFastApprox */
#define MI_BSDF_PHONG 15           /* src/bsdfs/phong.cpp (modified Phong): reflectance = diffuseReflectance, specular = specularReflectance, alpha = exponent,
                                      k[0] = specular sampling weight = lum(specular) / (lum(diffuse) + lum(specular)) (phong.cpp:104-108); constants only */
#define MI_BSDF_WARD 16            /* src/bsdfs/ward.cpp: reflectance = diffuseReflectance, specular = specularReflectance, alpha = alphaU, k[1] = alphaV, distr = variant
                                      (0 ward, 1 ward-duer, 2 balanced), k[0] = specular sampling weight as for phong; constants only */
#define MI_BSDF_COATING 17         /* src/bsdfs/coating.cpp (smooth dielectric layer): distr = nested record (a plain BSDF without transmission, not twosided itself), eta[0] = intIOR /
                                      extIOR, alpha = thickness, reflectance = sigmaA, specular = specularReflectance; may be twosided, and may sit under a mask / bumpmap / normalmap.
                                      Path integrator only */
#define MI_BSDF_BLEND 18           /* src/bsdfs/blendbsdf.cpp: eta[0], eta[1] = the two child records (plain BSDFs, as for mixturebsdf), reflectance = (w, w, w) for a constant
                                      weight or the value of the bound `weight` texture; may be twosided and may sit under a mask / bumpmap / normalmap.  Path integrator only */
#define MI_BSDF_ROUGHCOATING 19    /* src/bsdfs/roughcoating.cpp (rough dielectric layer): distr = nested record (a plain reflective BSDF without a delta lobe), eta[0] = intIOR / extIOR,
                                      eta[1] = thickness, eta[2] = microfacet distribution (0 beckmann, 1 ggx, 2 phong), alpha, flags bit 1 = sampleVisible, reflectance = sigmaA, specular,
                                      k[1], k[2] = offset / length of its rough-transmittance slice in the material tables (as for roughplastic).  Path integrator only */
#define MI_BSDF_FLAG_TWOSIDED 1u  /* wrapped in src/bsdfs/twosided.cpp             */
#define MI_BSDF_FLAG_SAMPLE_VISIBLE 2u
#define MI_BSDF_FLAG_NONLINEAR 4u /* plastic "nonlinear" */
#define MI_BSDF_FLAG_ANISOTROPIC 8u /* alphaU = alpha, alphaV = reflectance[0] (roughconductor) / k[0] (roughdielectric) (src/bsdfs/microfacet.h:116-127); mesh shapes need texture coordinates */
#define MI_BSDF_TEXTURE(i) (((uint32_t) (i) + 1u) << 8)   /* flags bits 8..23: texture i (mi_scene_set_textures) bound to `reflectance`: diffuse.reflectance, plastic / roughplastic.diffuseReflectance, difftrans.transmittance, mask.opacity; a bitmap texture record carries its average in color0 (plastic lobe weights, plastic.cpp:204-207) */

/* 2-D procedural textures over Texture2D (src/librender/texture.cpp:81-121: uv * scale + offset): src/textures/checkerboard.cpp, gridtexture.cpp */
#define MI_TEXTURE_CHECKERBOARD 0
#define MI_TEXTURE_GRID 1
#define MI_TEXTURE_BITMAP 2       /* src/textures/bitmap.cpp over TMIPMap (include/mitsuba/render/mipmap.h).  The MIP pyramid is INPUT DATA: levels
                                     [first_level, first_level + n_levels) of mi_scene_set_texture_data, exactly as the reference builds them (Bitmap::resample
                                     with a 2-lobed Lanczos filter, texels rounded to half; the adapter obtains them from the reference's own code).
                                     wrap_u / wrap_v: 0 clamp, 1 repeat, 2 mirror, 3 zero, 4 one (ReconstructionFilter::EBoundaryCondition); filter: 0 nearest,
                                     1 bilinear, 2 trilinear, 3 ewa (EMIPFilterType); camera hits filter through Intersection::computePartials */
typedef struct { uint32_t type; float color0[3], color1[3]; float line_width; float uoffset, voffset, uscale, vscale;
                 uint32_t wrap_u, wrap_v, filter; float max_anisotropy; uint32_t first_level, n_levels; } mi_texture;
typedef struct {
    uint32_t type, flags, distr;  /* distr: 0 beckmann, 1 ggx, 2 phong / Ashikhmin-Shirley (roughconductor, roughdielectric, roughplastic; samples all normals, microfacet.h:141-145) */
    float alpha;
    float reflectance[3], eta[3], k[3], specular[3];
} mi_material;

#define MI_EMITTER_AREA 0         /* src/emitters/area.cpp   */
#define MI_EMITTER_ENVMAP 1       /* src/emitters/envmap.cpp */
#define MI_EMITTER_CONSTANT 2     /* src/emitters/constant.cpp: `radiance`                                                   */
#define MI_EMITTER_POINT 3        /* src/emitters/point.cpp: `radiance` = intensity, position = translation of to_world     */
#define MI_EMITTER_SPOT 4         /* src/emitters/spot.cpp: intensity, to_world, cutoff / beam = cutoffAngle / beamWidth in degrees (no texture) */
#define MI_EMITTER_DIRECTIONAL 5  /* src/emitters/directional.cpp: `radiance` = irradiance, travel direction = to_world z axis */
/* 6: reserved (scene files: the compound `sunsky`, expanded by the reference itself inside the drop-in plugin) */
#define MI_EMITTER_COLLIMATED 7   /* src/emitters/collimated.cpp: `radiance` = power, position = translation of to_world.  A 0-D emitter: sampleDirect always fails (collimated.cpp:129-133), so in this
                                     unidirectional integrator it contributes nothing itself -- but it takes its share of the emitter-selection probability, exactly as in the reference */
/* shape: area lights only (index into the shape list; >= n_shapes: analytic shape shape - n_shapes), else -1 */
typedef struct { uint32_t type; int32_t shape; float radiance[3]; float weight; float cutoff, beam; float to_world[16]; } mi_emitter;

/* Analytic shapes behind Scene::rayIntersect (kd-tree leaf redirect, include/mitsuba/render/skdtree.h:292-301): src/shapes/rectangle.cpp,
 * disk.cpp, sphere.cpp, cylinder.cpp.  to_world = the shape's objectToWorld AFTER its constructor (sphere.cpp:113-123 and
 * cylinder.cpp:85-106 split the scale off into radius / length; rectangle.cpp:82-84 and disk.cpp:83-87 fold flipNormals into the
 * transform), to_object = its inverse as the reference computes it; radius: sphere, cylinder; length: cylinder. */
#define MI_SHAPE_RECTANGLE 0
#define MI_SHAPE_DISK 1
#define MI_SHAPE_SPHERE 2
#define MI_SHAPE_CYLINDER 3
#define MI_ANALYTIC_FLIP_NORMALS 1u   /* sphere / cylinder "flipNormals" */
typedef struct {
    uint32_t type; int32_t bsdf, emitter; uint32_t flags;
    float to_world[16], to_object[16];
    float radius, length; float pad[2];
} mi_analytic;

/* Integrator + sampler parameters: MonteCarloIntegrator properties (src/librender/integrator.cpp:191-226),
 * sampler properties (src/samplers/sobol.cpp:86-107; src/samplers/independent.cpp:51-60). */
#define MI_SAMPLER_INDEPENDENT 0  /* build-defined counter-based stream, seedable */
#define MI_SAMPLER_SOBOL 1
typedef struct {
    int32_t max_depth, rr_depth;
    uint32_t strict_normals, hide_emitters;
    uint32_t sampler, spp;
    uint64_t seed;               /* independent: seed of the counter stream; sobol: the sampler's `scramble` value (0 = unscrambled; src/samplers/sobol.cpp:92-102) */
    uint32_t device;             /* HIP device ordinal */
    uint32_t planes_per_batch;   /* sample planes traced per wavefront batch (0 = auto) */
    uint32_t opacity;            /* 1: alpha = 1 where the camera ray hits a surface, else 0 (RadianceQueryRecord::EOpacity, records.inl:121-137:
                                    the responsive drivers and films with an alpha channel); 0: alpha = 1 (classic film without alpha, integrator.cpp:160-161) */
    uint32_t integrator;         /* MI_INTEGRATOR_PATH (0, default), MI_INTEGRATOR_VOLPATH_SIMPLE or MI_INTEGRATOR_VOLPATH: the same loop over participating media (mi_scene_set_media);
                                    anything else: MI_ERR_UNSUPPORTED.  (Round 1 had `fast_math` here; one set of kernels ships: strict IEEE arithmetic) */
} mi_render_params;

typedef struct { uint32_t x0, y0, x1, y1; } mi_tile;   /* pixel rectangle [x0,x1) x [y0,y1) in GLOBAL film coordinates */

typedef struct {
    uint64_t rays, shadow_rays, path_length_sum, samples;  /* = reference StatsCounters "Normal rays traced", "Shadow rays traced",
                                                              avgPathLength (src/librender/skdtree.cpp:46-47, src/integrators/path/path.cpp:24) */
    double render_ms;                                      /* device time of the last mi_render_run (HIP events) */
    double extend_ms, shade_ms, shadow_ms, other_ms;       /* per-stage device time of the last run (0 unless profiling enabled) */
    uint64_t extend_launches, extend_rays;                 /* extend launches TIMED in the last run (first stream only) / closest-hit rays since clear */
    uint64_t extend_launches_all;                          /* extend launches of the last run on all streams */
} mi_stats;

const char *mi_last_error(void);

/* Sobol' tables (data): matrices32[dims][52], vdc[16][52], vdc_inv[16][52] (the tables of src/samplers/sobolseq.cpp:33,106537,107241).
 * Must be called once before a Sobol render; mitsuba-im_amd loads them from mitsuba-im_amd/data/sobol_tables.bin. */
int mi_set_sobol_tables(const uint32_t *matrices32, uint32_t dims, const uint64_t *vdc, const uint64_t *vdc_inv);
/* Same from a file; mi_scene_commit loads <dir of libmi355pt.so>/data/sobol_tables.bin (or $MI355PT_DATA/sobol_tables.bin) by itself
 * when no tables were set. */
int mi_load_sobol_tables(const char *path);

/* -- scene: replaces Scene::initialize -> ShapeKDTree::build (src/librender/scene.cpp:330-392, src/librender/skdtree.cpp:68-105) -- */
int mi_scene_create(mi_scene **out);
void mi_scene_destroy(mi_scene *s);
/* TriMesh::getVertexPositions/getVertexNormals/getVertexTexcoords/getTriangles of all meshes, concatenated; nrm/uv may be NULL */
int mi_scene_set_triangles(mi_scene *s, const float *pos, const float *nrm, const float *uv, const uint32_t *idx,
                           uint32_t n_verts, uint32_t n_tris, const mi_shape *shapes, uint32_t n_shapes);
/* analytic shapes, numbered after the meshes: shape index n_shapes + i, primitive index n_tris + i (Scene::getShapes order with the
 * meshes first).  Either call may be omitted, but a scene needs at least one primitive. */
int mi_scene_set_analytic(mi_scene *s, const mi_analytic *shapes, uint32_t n);
/* instances of shape groups; primitive index of the i-th: n_tris + n_analytic + i (they come last in Scene::getShapes order here) */
int mi_scene_set_instances(mi_scene *s, const mi_instance *instances, uint32_t n);
/* Media of the scene and who refers to them (Shape::addChild "interior" / "exterior", src/librender/shape.cpp:156-170; Sensor medium, src/librender/emitter.cpp:51-54):
 * shape_media[(n_shapes + n_analytic)][2] = (interior, exterior) medium index of every mesh, then every analytic shape, -1 = none; sensor_medium = the medium the
 * camera sits in or -1.  Only the volumetric integrators look at them. */
int mi_scene_set_media(mi_scene *s, const mi_medium *media, uint32_t n, const int32_t *shape_media, uint32_t n_pairs, int32_t sensor_medium);
int mi_scene_set_materials(mi_scene *s, const mi_material *materials, uint32_t n);
int mi_scene_set_textures(mi_scene *s, const mi_texture *textures, uint32_t n);
/* MIP levels of the bitmap textures: levels[n_levels][3] = (width, height, offset of the level's first float in `texels`), RGB floats row-major */
int mi_scene_set_texture_data(mi_scene *s, const uint32_t *levels, uint32_t n_levels, const float *texels, uint64_t n_texels);
int mi_scene_set_material_tables(mi_scene *s, const float *data, uint32_t n);   /* float tables the materials refer to by offset (roughplastic) */
int mi_scene_set_emitters(mi_scene *s, const mi_emitter *emitters, uint32_t n);    /* Scene::getEmitters order; samplingWeight in .weight */
int mi_scene_set_envmap(mi_scene *s, const float *rgb, uint32_t w, uint32_t h, const float *to_world16, float scale);
/* MIP pyramid of the environment map = record `texture` of mi_scene_set_textures (type bitmap, u repeats, v clamps, EWA, anisotropy 10: the settings of
 * src/emitters/envmap.cpp:144-145, 182-185; level 0 = the map itself): camera rays that leave the scene then get the filtered lookup of
 * EnvironmentMap::evalEnvironment (envmap.cpp:398-411).  -1 (default): level-0 bilinear lookups for them as for every other ray */
int mi_scene_set_envmap_filter(mi_scene *s, int32_t texture);
/* PerspectiveCameraImpl: m_sampleToCamera, world transform, clip planes (src/sensors/perspective.cpp:126-178) */
int mi_scene_set_camera(mi_scene *s, const float *sample_to_camera16, const float *to_world16, float near_clip, float far_clip);
/* Film crop size + reconstruction filter (src/librender/film.cpp:92; src/rfilters/<name>.cpp): kind 0 box(radius), 1 gaussian(stddev), 2 tent,
 * 3 mitchell (B in `radius`, C in `stddev`), 4 catmullrom, 5 lanczos (lobes in `radius`) */
int mi_scene_set_film(mi_scene *s, uint32_t width, uint32_t height, uint32_t filter_kind, float radius, float stddev);
int mi_scene_commit(mi_scene *s, uint32_t device);   /* BVH build + TriAccel table + upload */
/* A replica of a committed scene on HIP device `device` (may be the source's own): multi-device renders keep one scene copy per device, as the reference
 * ships the scene to every worker (src/librender/renderjob.cpp, sched_remote.cpp).  The host-side build is reused, only the upload is repeated. */
int mi_scene_clone(mi_scene *s, uint32_t device, mi_scene **out);

/* -- render: replaces SamplingIntegrator::renderBlock / ImageOrderIntegrator::render over MIPathTracer::Li
 *    (src/librender/integrator.cpp:141-189, :336-402, :469-486; src/integrators/path/path.cpp:119-294) -- */
int mi_render_create(mi_scene *s, const mi_render_params *p, mi_render **out);
void mi_render_destroy(mi_render *r);
/* Trace sample planes [sample_begin, sample_end) of every pixel of `tile` and accumulate them into the device film
 * (ImageBlock::put, include/mitsuba/render/imageblock.h:161-221).  One caller per handle. */
int mi_render_run(mi_render *r, mi_tile tile, uint32_t sample_begin, uint32_t sample_end);
/* Same for the rows y0, y0 + row_stride, ... (< y1) of the tile only: interleaved row ownership across ranks for multi-GPU load balance
 * (rank k of N renders tile {0, k, W, H} with row_stride N). */
int mi_render_run_rows(mi_render *r, mi_tile tile, uint32_t row_stride, uint32_t sample_begin, uint32_t sample_end);
int mi_render_clear(mi_render *r);                  /* ImageBlock::clear */
void mi_render_cancel(mi_render *r);                /* Integrator::cancel: thread-safe flag, observed before every batch: the run that sees it returns MI_CANCELLED
                                                       and consumes it (a cancel issued between two runs stops the next one); mi_render_clear drops a pending cancel */
/* Film read-back.  layout 0: raw ImageBlock sums (H+2b) x (W+2b) x 5 {R,G,B,alpha,weight} incl. border (classic, ESpectrumAlphaWeight);
 * layout 1: (H+2b) x (W+2b) x 4 RGBA sums (responsive target, src/im-mts/scene.cpp:317-321); layout 2: H x W x 3 developed RGB = sum/weight. */
int mi_render_film_size(mi_render *r, int layout, uint32_t *height, uint32_t *width, uint32_t *channels, uint32_t *border);
int mi_render_read_film(mi_render *r, int layout, float *host_out);
int mi_render_read_film_device(mi_render *r, int layout, void *device_out);   /* device pointer (e.g. a torch tensor) for the RCCL reduce */
/* dst += src (raw film sums + ray counters): merge step of a render whose rows were split over several handles / devices with mi_render_run_rows; replaces
 * Film::put(block) under the RenderQueue mutex (src/librender/renderproc.cpp:142-149).  Other device: peer copy over xGMI, then one add kernel.  Both idle. */
int mi_render_merge_film(mi_render *dst, mi_render *src);
/* Debug / parity: Li of individual (px, py, sampleIndex) triples through the very same kernels; out_li[n*3] */
int mi_render_samples(mi_render *r, const uint32_t *pairs, uint64_t n, float *out_li);
int mi_render_stats(mi_render *r, mi_stats *out);
int mi_render_set_profiling(mi_render *r, int enabled);   /* per-stage HIP-event timing: events are recorded between the stage launches of the first stream, nothing is serialised (off by default) */

/* bool Scene::rayIntersect(const Ray &ray, Intersection &its) for a batch of rays (include/mitsuba/render/scene.h:187-243): rays8 = (o.xyz, mint, d.xyz, maxt) per
 * ray, host pointers; the scene must be committed.  Record = the fields of mitsuba::Intersection the path consumes (include/mitsuba/render/shape.h:36-170). */
typedef struct {
    uint32_t valid;                      /* 0: no intersection in [mint, maxt]; the other fields are then unset */
    float t, p[3], ng[3];                /* its.t, its.p, its.geoFrame.n */
    float ns[3], s[3], tt[3];            /* its.shFrame.n / .s / .t */
    float uv[2], wi[3], bary[2];         /* its.uv, its.wi (local), the barycentrics (u, v) of a triangle hit / the shape's temp data */
    uint32_t prim; int32_t instance;     /* global primitive index (triangles first, then analytic shapes); instance index or -1 */
    int32_t material, emitter;           /* material / emitter table index (its.shape->getBSDF() / getEmitter()), -1: none */
} mi_intersection;
int mi_scene_ray_intersect(mi_scene *s, const float *rays8, uint64_t n, mi_intersection *out);

/* Unit-level device entry points used by the parity tests (each runs a small kernel over n items) */
int mi_debug_intersect(mi_scene *s, const float *rays8, uint64_t n, int any_hit, float *out_hits4);   /* t,u,v,prim (prim<0: miss) */
int mi_debug_intersect_inst(mi_scene *s, const float *rays8, uint64_t n, int any_hit, float *out_hits4, int32_t *out_instance);   /* + instance index of the hit (-1: scene-level primitive) */
int mi_debug_sobol(mi_scene *s, const uint32_t *px_py_k, uint64_t n, uint32_t ndims, uint64_t *out_index, float *out_values);
int mi_debug_camera_rays(mi_scene *s, const float *sample_pos2, uint64_t n, float *out_rays8);
int mi_debug_libm(int fn, const float *x, const float *y, uint64_t n, float *out);   /* the device restatements of glibc's routines (libm_glibc.h) as the kernels call them:
                                      fn 0 expf(x), 1 logf(x), 2 powf(x, y), 3 tanf(x), 4 atanf(x), 5 atan2f(x, y), 6 acosf(x); y may be NULL for the one-argument routines */
int mi_debug_sincosf(const float *x, uint64_t n, float *out_sin_cos2);   /* the device restatement of glibc's sincosf (warps): out[2i] = sin, out[2i+1] = cos */

#ifdef __cplusplus
}
#endif
#endif
