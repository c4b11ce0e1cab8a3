// scene_host.h -- host-side scene object behind the mi_scene handle.
#pragma once
#include "../../include/mi355pt.h"
#include "pt_types.h"
#include <string>
#include <vector>

namespace mi {

struct SceneHost {
    // inputs (mi_scene_set_*)
    std::vector<float> pos, nrm, uv; std::vector<uint32_t> idx; std::vector<mi_shape> shapes;
    std::vector<mi_material> materials; std::vector<mi_emitter> emitters; std::vector<mi_analytic> analytic; std::vector<mi_instance> instances; std::vector<float> materialTables; std::vector<mi_texture> textures; std::vector<uint32_t> texLevels; std::vector<float> texTexels; int32_t envTexture = -1;
    std::vector<mi_medium> media; std::vector<int32_t> shapeMedia; int32_t sensorMedium = -1;   // mi_scene_set_media
    std::vector<uint16_t> envGuideRows, envGuideCols; uint32_t envGuideKR = 0, envGuideKC = 0; void *dEnvGuideRows = nullptr, *dEnvGuideCols = nullptr;
    std::vector<MediumD> mediaD; std::vector<uint32_t> primMedia;   // derived: device records; per primitive (interior + 1) | (exterior + 1) << 16
    void *dMedia = nullptr, *dPrimMedia = nullptr;
    float s2c[16] = {0}, c2w[16] = {0}; float nearClip = 0, farClip = 0; bool haveCamera = false;
    uint32_t width = 0, height = 0, filterKind = 0; float filterRadius = 0.5f, filterStddev = 0.5f; bool haveFilm = false;
    std::vector<float> envRGB; uint32_t envW = 0, envH = 0; float envToWorld[16] = {0}, envScale = 1.0f;
    // derived on the host (scene_build.cpp)
    std::vector<uint32_t> triShape, i2; std::vector<TriAccelD> tris; std::vector<TriShade> shade; std::vector<BvhNode> nodes;
    std::vector<TriUV> triuv; bool anyUV = false;   // per-triangle uv + UV tangents (only when some mesh has texcoords)
    bool wideBvh = true;   // 4-wide quantised nodes (default) or the binary tree (MI355PT_BVH2=1)
    std::vector<InstanceD> instancesD; int bvhStackDirect = 0;   // stack entries of the fused walk (trace_fused.h: child codes pushed one by one), scene-level tree
    int bvhDepth = 0;   // stack entries the traversal needs (scene tree + return marker + deepest group tree)
    std::vector<AnalyticD> analyticD; uint32_t nTris = 0;   // analytic shapes: primitive index nTris + i; `tris` (BVH leaf order) holds a k = MI_K_ANALYTIC record for each
    std::vector<TriAccelD> packetExact;   // Wald records in ORIGINAL triangle order (packet mode, <= MI_PACKET_MAX triangles): pass 2 of trace.h
    std::vector<PacketGroupD> packetGroups; uint32_t packetGK[3] = {0, 0, 0}; float packetScale = 1.0f;   // pass-1 records sorted by projection axis
    std::vector<EmitterD> emittersD; std::vector<float> emitterCdf, areaCdf, emitterX; float emitterNorm = 0;
    bool envConstant = false, hasDeltaEmitters = false; float dirBsCenter[3] = {0, 0, 0}, dirBsRadius = 0;
    float aabbLo[3], aabbHi[3];
    float filterValues[MI_FILTER_RES + 1]; float filterRadiusEff = 0, filterScale = 0; int border = 0;
    float resolution = 1; uint32_t logRes = 0;
    int envIndex = -1; std::vector<float> envCdfCols, envCdfRows, envRowWeights; float envNormalization = 0, envToWorld3[9], envToLocal3[9], envBsCenter[3], envBsRadius = 0;
    // device
    bool committed = false; int device = 0;
    void *dNodes = nullptr, *dTris = nullptr, *dShade = nullptr, *dI2 = nullptr, *dNrm = nullptr, *dMaterials = nullptr, *dEmitters = nullptr,
         *dEmitterCdf = nullptr, *dAnalytic = nullptr, *dInstances = nullptr, *dMaterialTables = nullptr, *dTriUV = nullptr, *dTextures = nullptr, *dTexLevels = nullptr, *dTexTexels = nullptr, *dMipLut = nullptr, *dEmitterX = nullptr, *dAreaCdf = nullptr, *dFilter = nullptr, *dEnvRGB = nullptr, *dEnvCols = nullptr, *dEnvRows = nullptr, *dEnvWeights = nullptr, *dPacketGroups = nullptr, *dPacketExact = nullptr, *dSobolM32 = nullptr, *dSobolVdc = nullptr, *dSobolVdcInv = nullptr;
    DScene d{};

    void commitHost();          // scene_build.cpp
    int upload(int device);     // api.cpp
    void release();
    ~SceneHost() { release(); }
};

}  // namespace mi
