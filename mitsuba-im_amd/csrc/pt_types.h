// pt_types.h -- POD records shared by the host-side scene build and the gfx950 kernels.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define MI_EPSILON 1e-4f                    // reference include/mitsuba/core/constants.h:28
#define MI_SHADOW_EPSILON 1e-3f             // constants.h:29
#define MI_INV_PI 0.31830988618379067154f   // constants.h:64
#define MI_INV_TWOPI 0.15915494309189533577f
#define MI_ONE_MINUS_EPS 0.999999940395355225f
#define MI_PI 3.14159265358979323846f
#define MI_FILTER_RES 31                    // include/mitsuba/core/rfilter.h:28
#define MI_SOBOL_SIZE 52                    // src/samplers/sobolseq.h:31

// Wald triangle record in BVH leaf order, 48 B = three 16-B loads (reference include/mitsuba/render/triaccel.h:37-52)
struct TriAccelD {
    uint32_t k; float n_u, n_v, n_d;
    float a_u, a_v, b_nu, b_nv;
    float c_nu, c_nv; uint32_t prim; uint32_t pad;
};

// Packet mode, pass 1: the approximate Wald record of a coplanar PAIR of triangles forming a parallelogram, or of a single triangle (prim1 = 0xFFFFFFFF), 48 B =
// three 16-B scalar loads.  The record is TriAccel::load of the first triangle with its vertices cyclically relabelled (A*, B*, C*) so that the shared edge
// is A*C* (u* = 0): with s = u* + v* both triangles lie in 0 <= s <= 1, 0 <= v* <= 1, triangle prim0 on the side u* >= 0, its partner prim1 on u* <= 0.
// margin = 1.1 (|b_nu| + |b_nv| + |c_nu| + |c_nv|): barycentric error per unit of distance error (trace.h).
struct PacketGroupD {
    float n_u, n_v, n_d, a_u;
    float a_v, b_nu, b_nv, c_nu;
    float c_nv, margin; uint32_t prim0, prim1;
};

// BVH2 node, 64 B: both child boxes live in the parent, so one node fetch decides both descents.
// child >= 0: inner node index; child < 0: leaf, ~child = first_tri * 8 + (count - 1), count <= 8.
struct BvhNode {
    float lo0[3]; int32_t c0;
    float hi0[3]; int32_t c1;
    float lo1[3]; int32_t pad0;
    float hi1[3]; int32_t pad1;
};

// 4-wide node with quantised child boxes, 64 B (same size and array as BvhNode): child box c = org + q * step per axis, q in 0..255 (lo rounded down,
// hi rounded up: the boxes only grow), step = a power of two stored as a float (the walk multiplies it with 1 / d without decoding anything).
// child: >= 0 inner node index, < 0 leaf code as in BvhNode; an unused slot has an inverted box (lo 255, hi 0).
struct Bvh4Node {
    float org[3]; float step_x;           // 16-B word 0
    uint32_t qlo[3], qhi[3];              // words 1 / 2: per axis the four children's bytes, child c in bits 8c..8c+7
    float step_y, step_z;
    int32_t child[4];                     // word 3
};

// Per-triangle shading record in ORIGINAL triangle order, 128 B = eight 16-B words = ONE aligned cache line: everything a hit needs of its triangle arrives with
// one line fetch (round 2: a 96-B record straddling lines + the index of the third vertex + three vertex normals from three more lines -- the shade stage of the
// 251 k-triangle atrium moved 2.3 x its algorithmic bytes).  Words 4 / 5 hold the face frame (s, t) of a face-normal triangle or the first two VERTEX NORMALS of a
// smooth one (its frame is recomputed from the interpolated normal anyway), word 6 the third vertex normal.
#define MI_SHADE_WORDS 8
struct TriShade {
    float p0[3]; int32_t material;
    float p1[3]; int32_t emitter;
    float p2[3]; uint32_t flags;          // bit4: the mesh has texture coordinates (TriUV record: uv + UV tangents); bit0 face normals, bit1 material has a back side (twosided), bit2 BSDF without a smooth component (no NEE), bit3 rough conductor (material class for sorted shading)
    float ng[3]; uint32_t local_prim;
    float s[3]; uint32_t i0;              // smooth triangles: vertex normal 0 (i0, i1, i2: the vertex indices, kept for tools)
    float t[3]; uint32_t i1;              // smooth triangles: vertex normal 1
    float n2[3]; uint32_t i2;             // smooth triangles: vertex normal 2
    float pad[4];
};

// Per-triangle texture-coordinate record (meshes with texcoords only), 48 B: its.uv interpolation (skdtree.h:402-408) and the UV tangents of
// TriMesh::computeUVTangents (src/librender/trimesh.cpp:683-736), which replace the first edge as dpdu (skdtree.h:373-376)
struct TriUV { float uv0[2], uv1[2], uv2[2]; float dpdu[3], dpdv[3]; };
// 2-D procedural texture (src/textures/checkerboard.cpp, gridtexture.cpp over Texture2D), 48 B
// type 2 = bitmap (src/textures/bitmap.cpp over TMIPMap, include/mitsuba/render/mipmap.h): MIP levels [first_level, first_level + n_levels) of
// DScene::tex_levels (w, h, offset into tex_texels), wrap = ReconstructionFilter::EBoundaryCondition, filter = EMIPFilterType; 72 B
struct TextureD { uint32_t type; float color0[3], color1[3]; float line_width; float uoffset, voffset, uscale, vscale;
                  uint32_t wrap_u, wrap_v, filter; float max_anisotropy; uint32_t first_level, n_levels; };

struct MaterialD {                        // 64 B; flags bits 8..23: texture index + 1 bound to `reflectance`
    uint32_t type, flags, distr; float alpha;
    float reflectance[3], eta[3], k[3], specular[3];
};

struct EmitterD {                         // 48 B
    float radiance[3]; float weight;
    uint32_t first_tri, tri_count, cdf_offset; float inv_area;
    uint32_t type; int32_t shape; int32_t analytic; uint32_t pad;   // analytic >= 0: area light on analytic shape `analytic` (no triangle CDF)
};

// Analytic shape (reference src/shapes/{rectangle,disk,sphere,cylinder}.cpp), 160 B = ten 16-B loads.  Primitive index of the i-th
// analytic shape = n_tris + i; in BVH leaves it appears as a TriAccelD record with k = MI_K_ANALYTIC and prim = that index
// (the reference marks such records with k = KNoTriangleFlag, include/mitsuba/render/skdtree.h:292-301).
#define MI_SHAPE_RECTANGLE 0
#define MI_SHAPE_DISK 1
#define MI_SHAPE_SPHERE 2
#define MI_SHAPE_CYLINDER 3
#define MI_K_ANALYTIC 4u
#define MI_K_INSTANCE 5u                      // leaf record of an instance (src/shapes/instance.cpp): prim = instance index; always alone in its leaf
#define MI_INV_FOURPI 0.07957747154594766788f   // constants.h:66
#define MI_ANALYTIC_PACKET_MAX 16
// One placement of a shape group (reference src/shapes/instance.cpp + shapegroup.cpp), 128 B.  The group's own BVH lives in the same node array.
struct InstanceD {
    float to_world[12];                   // rows 0..2 of the instance transform
    float to_object[12];                  // rows 0..2 of its inverse
    float glo[3]; int32_t root;           // the group's kd-tree box (enlarged, gkdtree.h:1213-1220); root node of the group's BVH
    float ghi[3]; uint32_t group;
};

struct AnalyticD {
    float to_world[12];                   // rows 0..2 of objectToWorld
    float to_object[12];                  // rows 0..2 of worldToObject
    float n[3]; uint32_t type;            // rectangle / disk: world normal
    float dpdu[3]; uint32_t flags;        // rectangle: dpdu; flags bit0 flipNormals, bit1 back side (twosided), bit2 BSDF without smooth component, bit3 rough conductor
    float center[3]; float radius;        // sphere centre; sphere / cylinder radius
    float length, inv_area; int32_t material, emitter;
};

// Homogeneous participating medium + phase function (mi_medium), 64 B: sigma_t precomputed (Medium: m_sigmaT = m_sigmaA + m_sigmaS)
struct MediumD { float sigma_s[3]; uint32_t strategy; float sigma_t[3]; uint32_t phase; float sampling_density, medium_sampling_weight, g, pad; float pad2[4]; };

// Everything a kernel needs to know about the scene; passed by value.
struct DScene {
    const BvhNode *nodes; const TriAccelD *tris; const TriShade *shade; const uint32_t *i2; const float *nrm;
    const MaterialD *materials; const EmitterD *emitters; const float *emitter_cdf; const float *area_cdf;
    const AnalyticD *analytic; uint32_t n_analytic;
    const InstanceD *instances; uint32_t n_instances;   // hit records of instanced triangles carry the instance index in Queues::hitInst
    uint32_t ext;                         // analytic shapes or delta emitters present: selects the k_shade<..., EXT> variants
    // scene-level emitters beyond envmap (src/emitters/constant.cpp, point.cpp, spot.cpp, directional.cpp): per emitter 16 floats
    //   [0..2] position (point, spot) / travel direction (directional); spot: [3] cos(cutoff), [4..12] world->local 3x3, [13] cos(beam), [14] cutoff, [15] 1/(cutoff-beam)
    const TriUV *triuv; const TextureD *textures; uint32_t n_textures, env_texture;   // n_textures: bound to materials (kernel variant); env_texture: index + 1 of the environment map's MIP pyramid record, 0 = none
    const uint32_t *tex_levels; const float *tex_texels; const float *mip_lut;   // MIP pyramids (input data); EWA weight table (mipmap.h:297-302)
    float cam_dx[3], cam_dy[3];           // PerspectiveCameraImpl::m_dx / m_dy (perspective.cpp:159-163)
    const float *material_tables;                    // float tables referenced by materials (roughplastic: k[1] = offset, k[2] = length)
    const float *emitter_x; uint32_t env_constant;   // env_constant: the environment emitter (env_index) is `constant`; radiance in its EmitterD
    float dir_bs_center[3], dir_bs_radius;           // DirectionalEmitter::createShape: kd-tree box bounding sphere x 1.1
    uint32_t n_tris, n_nodes, n_emitters, n_materials;
    float emitter_norm;
    float aabb_lo[3], aabb_hi[3];         // kd-tree root box of the reference incl. its enlargement (gkdtree.h:1213-1220)
    float s2c[16], c2w[16];               // sampleToCamera, cameraToWorld
    float near_clip, far_clip, inv_res_x, inv_res_y;
    uint32_t width, height;
    // film
    const float *filter_values;           // [MI_FILTER_RES + 1], global memory (indexed per lane)
    float filter_radius, filter_scale; int32_t border;
    // sobol
    const uint32_t *sobol_m32; const uint64_t *sobol_vdc, *sobol_vdc_inv;   // vdc rows for m = log_res only
    uint32_t sobol_dims, log_res; float resolution;
    // traversal variant: packet_n > 0 -> the whole scene is ONE triangle packet held in constant memory (scenes of <= MI_PACKET_MAX
    // triangles: every lane tests every triangle with wave-uniform operands, no stack, no divergence); else BVH of depth bvh_depth
    uint64_t geo_bytes;                   // nodes + leaf records live in one allocation of this many bytes (nodes first)
    uint32_t search_flags;                // table searches of emitter sampling fetched whole (cdfSampleReuse): bit 0 the emitter-selection table (<= 3 emitters), bit 1 the triangle tables of the area lights (every light mesh <= 3 triangles); MI355PT_SEARCH overrides (A/B runs)
    uint32_t bvh_stack_direct;            // stack entries per lane the fused walk of trace_fused.h can need on the scene-level tree
    uint32_t packet_n, bvh_depth, bvh_wide;   // bvh_depth: traversal stack entries the tree(s) can need; bvh_wide: the node array holds Bvh4Node records
    // packet mode (trace.h): pass-1 records (PacketGroupD, sorted by projection axis: [0,gk[0]) axis 0, [gk[0],gk[1]) axis 1, [gk[1],gk[2]) axis 2; degenerate
    // triangles dropped), exact Wald records in ORIGINAL triangle order for pass 2, largest |coordinate| of the scene box (error-margin scale)
    const struct PacketGroupD *packet_groups; const TriAccelD *packet_exact; uint32_t packet_gk[3]; float packet_scale;
    // participating media (volumetric integrators only): prim_media[primitive] = (interior + 1) | (exterior + 1) << 16 for triangles, then analytic shapes; 0 = none
    const MediumD *media; const uint32_t *prim_media; uint32_t n_media; int32_t sensor_medium;
    uint32_t has_adapters;   // bit 0: mixturebsdf / bumpmap / normalmap records present (the WRAP variants of k_shade); bit 1: ENull lobes behind a mask / mixture (NX variants of the volumetric shadow stages)
    uint32_t has_roughconductor, has_diffuse;   // non-diffuse / plain diffuse materials present: select the shade kernel variants (both: two launches per bounce, shade.h)
    uint32_t small_tables, area_cdf_len;   // small_tables: shading records / materials / emitters / CDFs fit the LDS staging budget
    // environment emitter (reference src/emitters/envmap.cpp); env_index = its position in the emitter list, -1 = none
    const float *env_rgb, *env_cdf_cols, *env_cdf_rows, *env_row_weights;
    // guide tables of the two CDF searches (scene_build.cpp): guide[b] = lower_bound(cdf, b / K) for K = env_guide_rows / env_guide_cols buckets (powers of two), so a
    // search starts in [guide[b], guide[b + 1]] instead of [0, size + 1] -- the same index, a third of the dependent loads
    const uint16_t *env_guide_rows_t, *env_guide_cols_t; uint32_t env_guide_rows, env_guide_cols;
    int32_t env_index, env_w, env_h;
    float env_normalization, env_scale, env_pixel_w, env_pixel_h, env_bs_radius;
    float env_to_world[9], env_to_local[9], env_bs_center[3];
};
#define MI_PACKET_MAX 64

struct RenderConst {
    int32_t max_depth, rr_depth; uint32_t strict_normals, hide_emitters, opacity;
    uint32_t sampler; uint32_t seed_mix;  // independent: seed * 0x9E3779B9
    uint32_t sobol_scramble;              // Sobol: low 32 bits of sampleTEA(scramble) (src/samplers/sobol.cpp:92-102), 0 = unscrambled
    // Sobol' direction matrices folded into 4-bit lookup tables: nib[dim][n][v] = XOR of matrices32[dim*52 + 4n + b] over the bits b of v
    const uint32_t *sobol_nib; uint32_t nib_count, nib_dims;
    // sobol::look_up (src/samplers/sobolseq.h:99-131) and the first two sample dimensions are XOR-linear in (frame, px, py): three tables of
    // {index lo, index hi, dim-0 bits, dim-1 bits} (api.cpp buildSobolLookupTables), XORed together in k_generate; null when log_res <= 1
    const uint4 *sobol_frame, *sobol_px, *sobol_py; uint32_t sobol_nframes;   // frames beyond the table (parity entry point only) take sobolLookUp
    float inv_sqrt_spp;                   // RayDifferential::scaleDifferential amount (integrator.cpp:145-146, 403-405)
    uint32_t integrator;                  // MI_INTEGRATOR_*
    float alpha_dist;                     // volumetric integrators, EOpacity: twice the radius of the scene's bounding sphere (records.inl:128-134)
    uint32_t state_init;                  // bits ORed into the state word st0.w of a fresh path (volumetric integrators: radiance-type bits, the sensor's medium)
    uint32_t order_offset_words;          // dynamic-LDS offset of the material-sort index list (0 = no sorting); set per launch
};
