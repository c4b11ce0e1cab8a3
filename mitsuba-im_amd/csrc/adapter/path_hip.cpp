// adapter/path_hip.cpp -- the drop-in plugin: `path_hip.so` for Mitsuba-IM's plugin directory.
//
// Compiled AGAINST THE REFERENCE'S HEADERS with the reference's defines (include/mitsuba/core/cobject.h:102-110 plugin
// entry points, src/integrators/mark_integrator.cpp:3 marker string).  It exposes the MI355X path tracer as
//   * a classic mitsuba::Integrator ("path_hip", same properties as `path`: maxDepth, rrDepth, strictNormals, hideEmitters), and
//   * a ResponsiveIntegrator through makeResponsiveIntegrator() (include/mitsuba/render/integrator2.h:49-100),
// flattening the live mitsuba::Scene into the C-ABI of libmi355pt.so (include/mi355pt.h).  All GPU work runs on threadIdx 0.
// HIP / C-ABI errors become Log(EError) (which throws std::runtime_error, as everywhere in the reference).
#include <mitsuba/render/scene.h>
#include <mitsuba/render/integrator.h>
#include <mitsuba/render/integrator2.h>
#include <mitsuba/render/trimesh.h>
#include <mitsuba/render/sensor.h>
#include <mitsuba/render/film.h>
#include <mitsuba/render/emitter.h>
#include <mitsuba/render/bsdf.h>
#include <mitsuba/render/medium.h>
#include <mitsuba/render/phase.h>
#include <mitsuba/render/sampler.h>
#include <mitsuba/render/imageblock.h>
#include <mitsuba/render/renderqueue.h>
#include <mitsuba/core/rfilter.h>
#include <mitsuba/core/plugin.h>
#include <mitsuba/core/fresolver.h>
#include <mitsuba/core/bitmap.h>
#include <mitsuba/core/half.h>
#include <mitsuba/core/mstream.h>
#include <mitsuba/core/serialization.h>
#include <instance.h>     // src/shapes/instance.h (pulls in shapegroup.h, which has no include guard): inline accessors only (getShapeGroup, getWorldTransform, getKDTree)
#include <mitsuba/render/mipmap.h>
#include <rtrans.h>       // src/bsdfs/rtrans.h: RoughTransmittance, as RoughPlastic::configure uses it
#include <ior.h>   // src/bsdfs/ior.h: lookupIOR, as used by RoughConductor's constructor
#include "../integrator_host.h"

MTS_NAMESPACE_BEGIN

namespace {

struct FlatScene {
    std::vector<float> uv; bool anyUV = false; std::vector<mi_texture> textures; std::vector<uint32_t> texLevels; std::vector<float> texTexels; int32_t envTexture = -1;
    std::vector<float> pos, nrm; std::vector<uint32_t> idx; std::vector<mi_shape> shapes; std::vector<mi_material> materials; std::vector<mi_emitter> emitters;
    std::vector<mi_analytic> analytic; std::vector<const Shape *> analyticShapes; std::vector<mi_instance> instances; std::vector<float> materialTables;
    bool anyNormals = false;
    std::vector<mi_medium> media; std::vector<int32_t> shapeMedia; int32_t sensorMedium = -1;      // participating media (volumetric = true)
    std::vector<float> envRGB; uint32_t envW = 0, envH = 0; float envToWorld[16], envScale = 1.0f;
};

#define MI_CHECK(call) do { int rc_ = (call); if (rc_ != MI_OK) SLog(EError, "path_hip: %s failed: %s", #call, mi_last_error()); } while (0)

/// RoughPlastic::configure (src/bsdfs/roughplastic.cpp:281-299): external rough transmittance reduced to a 1-D slice (setEta, setAlpha), internal
/// diffuse transmittance; the slice is appended to `tables` and referenced from the material (k[1] offset, k[2] length, k[0] Tdiff_int)
struct RTAccess : RoughTransmittance {
    RTAccess(MicrofacetDistribution::EType t) : RoughTransmittance(t) { }
    const Float *trans() const { return m_trans; } size_t thetaSamples() const { return m_thetaSamples; }
};
static std::vector<float> *g_tables = NULL;      // FlatScene::materialTables of the flatten() in progress (single-threaded: preprocess)
static void roughPlasticTables(mi_material &m) {
    ref<RTAccess> ext = new RTAccess(m.distr == 1 ? MicrofacetDistribution::EGGX : m.distr == 2 ? MicrofacetDistribution::EPhong : MicrofacetDistribution::EBeckmann);
    ext->checkEta(m.eta[0]); ext->checkAlpha(m.alpha);
    ref<RoughTransmittance> internal = ext->clone();
    ext->setEta(m.eta[0]); internal->setEta(1 / m.eta[0]); ext->setAlpha(m.alpha);
    m.k[0] = internal->evalDiffuse(m.alpha); m.k[1] = (float) g_tables->size(); m.k[2] = (float) ext->thetaSamples();
    g_tables->insert(g_tables->end(), ext->trans(), ext->trans() + ext->thetaSamples());
}

/// `twosided` keeps its nested BSDF private (src/bsdfs/twosided.cpp:197-198), the nested object's Properties are gone once it is configured from a
/// parent, and every BSDF keeps its textures private, so wrappers and spatially varying BSDFs are read through the one public door that shows their
/// content: serialisation.  InstanceManager::serialize (src/libcore/serialization.cpp:76-91) writes id, class name, then the object's own serialize();
/// the layouts parsed here are TwoSidedBRDF (twosided.cpp:79-84), RoughConductor (roughconductor.cpp:219-229), SmoothConductor (conductor.cpp:204-209),
/// SmoothPlastic (plastic.cpp:180-187), RoughPlastic (roughplastic.cpp:247-257), SmoothDiffuse (diffuse.cpp:163-167), RoughDiffuse (roughdiffuse.cpp:269-275), DiffuseTransmitter
/// (difftrans.cpp:66-70), constant textures (src/librender/basictexture.cpp:29-49), Texture2D (src/librender/texture.cpp:106-110) with Checkerboard /
/// GridTexture / BitmapTexture (bitmap.cpp:404-432).
static std::vector<mi_texture> *g_textures = NULL; static std::vector<uint32_t> *g_texLevels = NULL; static std::vector<float> *g_texTexels = NULL;
static std::vector<mi_material> *g_materials = NULL;       // FlatScene::materials of the flatten() in progress: a `mask` appends its nested BSDF's record
struct NestedReader {
    ref<MemoryStream> ms; std::map<uint32_t, std::vector<float> > seen; std::map<uint32_t, int> seenTexture;
    int lastTexture = -1;                 ///< index into g_textures when the last texture() call met a spatially varying texture, else -1
    int spatialTexture(const std::string &tcls);
    /// a texture reference -> its constant value(s), or (checkerboard / gridtexture / bitmap) a texture record (`lastTexture`); back-references resolve through `seen`
    std::vector<float> texture() {
        lastTexture = -1;
        uint32_t id = ms->readUInt(); if (id == 0) SLog(EError, "path_hip: missing texture inside a BSDF");
        if (seen.count(id)) { lastTexture = seenTexture[id]; return seen[id]; }
        std::string cls = ms->readString(); std::vector<float> v;
        if (cls == "ConstantSpectrumTexture") { Spectrum sp(ms.get()); Float r, g, b; sp.toLinearRGB(r, g, b); v = {(float) r, (float) g, (float) b}; }
        else if (cls == "ConstantFloatTexture") v = {(float) ms->readFloat()};
        else if (cls == "Checkerboard" || cls == "GridTexture" || cls == "BitmapTexture") { lastTexture = spatialTexture(cls); v = {0.5f, 0.5f, 0.5f}; }
        else SLog(EError, "path_hip: texture \"%s\" is not implemented (constant values, checkerboard, gridtexture, bitmap)", cls.c_str());
        seen[id] = v; seenTexture[id] = lastTexture; return v;
    }
    /// a parameter the path only takes as a constant
    std::vector<float> constant(const char *what) {
        std::vector<float> v = texture();
        if (lastTexture >= 0) SLog(EError, "path_hip: a texture on \"%s\" is not implemented (textures drive diffuse.reflectance, plastic / roughplastic.diffuseReflectance, difftrans.transmittance)", what);
        return v;
    }
    void rgb(float *dst) { Spectrum sp(ms.get()); Float r, g, b; sp.toLinearRGB(r, g, b); dst[0] = r; dst[1] = g; dst[2] = b; }
};
/// Texture2D::serialize + the texture's own fields -> a mi_texture record (bitmaps: the MIP pyramid rebuilt with the reference's own TMIPMap, average in color0)
int NestedReader::spatialTexture(const std::string &tcls) {
    NestedReader &rd = *this;
    mi_texture t; memset(&t, 0, sizeof(t)); t.type = tcls == "GridTexture" ? MI_TEXTURE_GRID : tcls == "BitmapTexture" ? MI_TEXTURE_BITMAP : MI_TEXTURE_CHECKERBOARD;
    t.uoffset = rd.ms->readFloat(); t.voffset = rd.ms->readFloat(); t.uscale = rd.ms->readFloat(); t.vscale = rd.ms->readFloat();
    if (t.type == MI_TEXTURE_BITMAP) {
        // BitmapTexture::serialize (src/textures/bitmap.cpp:404-432): parameters + the image file's bytes; the MIP pyramid is rebuilt with the reference's
        // own code exactly as BitmapTexture's constructors do (2-lobed Lanczos, TMIPMap<Color3, Color3h>; :193-214, :363-401)
        rd.ms->readString(); t.filter = rd.ms->readUInt(); t.wrap_u = rd.ms->readUInt(); t.wrap_v = rd.ms->readUInt();
        Float gamma = rd.ms->readFloat(); t.max_anisotropy = rd.ms->readFloat(); std::string channel = rd.ms->readString();
        size_t size = rd.ms->readSize(); ref<MemoryStream> img = new MemoryStream(size); rd.ms->copyTo(img, size); img->seek(0);
        ref<Bitmap> bitmap = new Bitmap(Bitmap::EAuto, img); if (gamma != 0) bitmap->setGamma(gamma);
        if (!channel.empty()) {              // a texture from one channel of the image (bitmap.cpp:261-266, findChannel :303-321): a luminance pyramid
            int found = -1;
            for (int i = 0; i < bitmap->getChannelCount(); ++i) { std::string nm = bitmap->getChannelName(i); std::transform(nm.begin(), nm.end(), nm.begin(), ::tolower); if (nm == channel) found = i; }
            if (found < 0) SLog(EError, "Channel \"%s\" not found!", channel.c_str());
            bitmap = bitmap->extractChannel(found);
            if (channel == "a") bitmap->setGamma(1.0f);
        }
        Properties rp("lanczos"); rp.setInteger("lobes", 2);
        ref<ReconstructionFilter> rf = static_cast<ReconstructionFilter *>(PluginManager::getInstance()->createObject(MTS_CLASS(ReconstructionFilter), rp)); rf->configure();
        typedef TSpectrum<Float, 3> Color3; typedef TSpectrum<half, 3> Color3h; typedef TSpectrum<Float, 1> Color1; typedef TSpectrum<half, 1> Color1h;
        const bool lum = bitmap->getPixelFormat() == Bitmap::ELuminance || bitmap->getPixelFormat() == Bitmap::ELuminanceAlpha;
        if (!lum && bitmap->getPixelFormat() != Bitmap::ERGB && bitmap->getPixelFormat() != Bitmap::ERGBA) SLog(EError, "path_hip: The input image has an unsupported pixel format!");
        t.first_level = (uint32_t) (g_texLevels->size() / 3);
        auto pushLevel = [&](const Bitmap *lb, int ch) {
            g_texLevels->push_back((uint32_t) lb->getWidth()); g_texLevels->push_back((uint32_t) lb->getHeight()); g_texLevels->push_back((uint32_t) g_texTexels->size());
            const half *hp = lb->getFloat16Data(); const size_t n = (size_t) lb->getWidth() * lb->getHeight();
            for (size_t i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) g_texTexels->push_back((float) hp[i * ch + (ch == 3 ? c : 0)]);
        };
        if (lum) { ref<TMIPMap<Color1, Color1h> > mip = new TMIPMap<Color1, Color1h>(bitmap, Bitmap::ELuminance, Bitmap::EFloat, rf, (ReconstructionFilter::EBoundaryCondition) t.wrap_u, (ReconstructionFilter::EBoundaryCondition) t.wrap_v, (EMIPFilterType) t.filter, t.max_anisotropy);
                   t.n_levels = (uint32_t) mip->getLevels(); for (int l = 0; l < mip->getLevels(); ++l) pushLevel(mip->toBitmap(l), 1);
                   t.color0[0] = t.color0[1] = t.color0[2] = mip->getAverage()[0]; }                    // BitmapTexture::getAverage (bitmap.cpp:504-514)
        else { ref<TMIPMap<Color3, Color3h> > mip = new TMIPMap<Color3, Color3h>(bitmap, Bitmap::ERGB, Bitmap::EFloat, rf, (ReconstructionFilter::EBoundaryCondition) t.wrap_u, (ReconstructionFilter::EBoundaryCondition) t.wrap_v, (EMIPFilterType) t.filter, t.max_anisotropy);
               t.n_levels = (uint32_t) mip->getLevels(); for (int l = 0; l < mip->getLevels(); ++l) pushLevel(mip->toBitmap(l), 3);
               Color3 avg = mip->getAverage(); for (int c = 0; c < 3; ++c) t.color0[c] = avg[c]; }
    } else { rd.rgb(t.color0); rd.rgb(t.color1); if (t.type == MI_TEXTURE_GRID) t.line_width = rd.ms->readFloat(); }
    g_textures->push_back(t);
    return (int) g_textures->size() - 1;
}
/// one nested / serialised BSDF of class `cls` (its BSDF::serialize bool already consumed) -> material; false: leave it to the generic path
static bool readNestedInstance(NestedReader &rd, mi_material &m);
static bool readSerializedBSDF(NestedReader &rd, const std::string &cls, mi_material &m, bool acceptConstantDiffuse = false) {
    auto bind = [&](const std::vector<float> &v) { if (rd.lastTexture >= 0) m.flags |= MI_BSDF_TEXTURE(rd.lastTexture); memcpy(m.reflectance, v.data(), 12); };
    if (cls == "RoughConductor") {
        uint32_t distr = rd.ms->readUInt(); bool sampleVisible = rd.ms->readBool();
        std::vector<float> au = rd.constant("alphaU"), av = rd.constant("alphaV"), spec = rd.constant("specularReflectance");
        // MicrofacetDistribution::EType: EBeckmann 0, EGGX 1, EPhong 2 (microfacet.h:46-52); sampleVisible as stored (Phong clears it in the constructor, :141-145)
        if (distr > 2) SLog(EError, "path_hip: unknown microfacet distribution");
        m.type = MI_BSDF_ROUGHCONDUCTOR; m.distr = distr; m.alpha = au[0]; if (sampleVisible) m.flags |= MI_BSDF_FLAG_SAMPLE_VISIBLE;
        if (au[0] != av[0]) { m.flags |= MI_BSDF_FLAG_ANISOTROPIC; m.reflectance[0] = av[0]; }
        memcpy(m.specular, spec.data(), 12); rd.rgb(m.eta); rd.rgb(m.k);
    } else if (cls == "SmoothConductor") {
        std::vector<float> spec = rd.constant("specularReflectance"); m.type = MI_BSDF_CONDUCTOR; memcpy(m.specular, spec.data(), 12); rd.rgb(m.eta); rd.rgb(m.k);
    } else if (cls == "RoughPlastic") {                                 // roughplastic.cpp:247-257
        uint32_t distr = rd.ms->readUInt(); bool sampleVisible = rd.ms->readBool();
        std::vector<float> spec = rd.constant("specularReflectance"), diff = rd.texture(); bind(diff);
        std::vector<float> alpha = rd.constant("alpha");
        if (distr > 2) SLog(EError, "path_hip: roughplastic: unknown microfacet distribution %u", distr);
        m.type = MI_BSDF_ROUGHPLASTIC; if (sampleVisible && distr != 2) m.flags |= MI_BSDF_FLAG_SAMPLE_VISIBLE; m.distr = distr; m.alpha = alpha[0];
        memcpy(m.specular, spec.data(), 12);
        m.eta[0] = rd.ms->readFloat(); if (rd.ms->readBool()) m.flags |= MI_BSDF_FLAG_NONLINEAR;
        roughPlasticTables(m);
    } else if (cls == "SmoothPlastic") {
        m.type = MI_BSDF_PLASTIC; m.eta[0] = rd.ms->readFloat(); if (rd.ms->readBool()) m.flags |= MI_BSDF_FLAG_NONLINEAR;
        std::vector<float> spec = rd.constant("specularReflectance"), diff = rd.texture(); bind(diff); memcpy(m.specular, spec.data(), 12);
        m.k[0] = fresnelDiffuseReflectance(1 / m.eta[0], false);
    } else if (cls == "SmoothDiffuse") {
        std::vector<float> refl = rd.texture(); if (rd.lastTexture < 0 && !acceptConstantDiffuse) return false;      // constant reflectance: the generic path reads it through the public interface
        m.type = MI_BSDF_DIFFUSE; bind(refl);
    } else if (cls == "Mask") {                                         // mask.cpp:92-96: opacity texture, then the nested BSDF
        std::vector<float> op = rd.texture(); const int opTex = rd.lastTexture;
        mi_material nested; memset(&nested, 0, sizeof(nested));
        if (!readNestedInstance(rd, nested)) SLog(EError, "path_hip: the BSDF nested in `mask` is not implemented");      // (a plain BSDF, a mixturebsdf or a bumpmap / normalmap)
        if (nested.type == MI_BSDF_MASK) SLog(EError, "path_hip: a mask nested in a mask is not implemented");
        const uint32_t keepFlags = m.flags;
        memset(&m, 0, sizeof(m)); m.type = MI_BSDF_MASK; m.flags = keepFlags; m.distr = (uint32_t) g_materials->size(); g_materials->push_back(nested);
        memcpy(m.reflectance, op.data(), 12); if (opTex >= 0) m.flags |= MI_BSDF_TEXTURE(opTex);
    } else if (cls == "RoughCoating") {                                 // roughcoating.cpp:174-185: type, sampleVisible, nested, sigmaA, specularReflectance, alpha, eta, thickness
        const uint32_t distr = rd.ms->readUInt(); const bool sampleVisible = rd.ms->readBool();
        mi_material nested; memset(&nested, 0, sizeof(nested));
        if (!readNestedInstance(rd, nested)) SLog(EError, "path_hip: the BSDF nested in `roughcoating` is not implemented");
        if (nested.type == MI_BSDF_MASK || nested.type == MI_BSDF_MIXTURE || nested.type == MI_BSDF_BUMPMAP || nested.type == MI_BSDF_NORMALMAP || nested.type == MI_BSDF_COATING || nested.type == MI_BSDF_ROUGHCOATING || nested.type == MI_BSDF_BLEND || (nested.flags & MI_BSDF_FLAG_TWOSIDED))
            SLog(EError, "path_hip: a roughcoating over an adapter (mask, mixturebsdf, blendbsdf, bumpmap, normalmap, twosided, coating) is not implemented");
        std::vector<float> sa = rd.constant("sigmaA"), spec = rd.constant("specularReflectance"), alpha = rd.constant("alpha");
        const float eta = rd.ms->readFloat(), thickness = rd.ms->readFloat();
        if (distr > 2) SLog(EError, "path_hip: unknown microfacet distribution");
        const uint32_t keepFlags = m.flags;
        memset(&m, 0, sizeof(m)); m.type = MI_BSDF_ROUGHCOATING; m.flags = keepFlags | (sampleVisible ? MI_BSDF_FLAG_SAMPLE_VISIBLE : 0u);
        m.eta[0] = eta; m.alpha = alpha[0]; m.distr = distr; roughPlasticTables(m);      // (the slice for (distribution, eta, alpha); k[0] is overwritten at commit)
        m.distr = (uint32_t) g_materials->size(); g_materials->push_back(nested);
        m.eta[1] = thickness; m.eta[2] = (float) distr; memcpy(m.reflectance, sa.data(), 12); memcpy(m.specular, spec.data(), 12);
    } else if (cls == "BlendBSDF") {                                    // blendbsdf.cpp:94-101: the weight texture, then the two BSDFs; the children become records of their own
        std::vector<float> w = rd.texture(); const int wTex = rd.lastTexture;
        const uint32_t keepFlags = m.flags; memset(&m, 0, sizeof(m)); m.type = MI_BSDF_BLEND; m.flags = keepFlags;
        for (int i = 0; i < 2; ++i) {
            mi_material child; memset(&child, 0, sizeof(child));
            if (!readNestedInstance(rd, child) || (child.type >= MI_BSDF_MASK && child.type != MI_BSDF_ROUGHDIFFUSE && child.type != MI_BSDF_PHONG && child.type != MI_BSDF_WARD))
                SLog(EError, "path_hip: this BSDF inside a blendbsdf is not implemented (plain BSDFs, optionally twosided)");
            if ((child.flags >> 8) & 0xFFFFu) SLog(EError, "path_hip: textures on the BSDFs inside a blendbsdf are not implemented");
            m.eta[i] = (float) g_materials->size(); g_materials->push_back(child);
        }
        m.reflectance[0] = m.reflectance[1] = m.reflectance[2] = w[0]; if (wTex >= 0) m.flags |= MI_BSDF_TEXTURE(wTex);
    } else if (cls == "SmoothCoating") {                                // coating.cpp:150-158: eta, thickness, the nested BSDF, sigmaA, specularReflectance
        const float eta = rd.ms->readFloat(), thickness = rd.ms->readFloat();
        mi_material nested; memset(&nested, 0, sizeof(nested));
        if (!readNestedInstance(rd, nested)) SLog(EError, "path_hip: the BSDF nested in `coating` is not implemented");
        if (nested.type == MI_BSDF_MASK || nested.type == MI_BSDF_MIXTURE || nested.type == MI_BSDF_BUMPMAP || nested.type == MI_BSDF_NORMALMAP || nested.type == MI_BSDF_COATING || (nested.flags & MI_BSDF_FLAG_TWOSIDED))
            SLog(EError, "path_hip: a coating over an adapter (mask, mixturebsdf, bumpmap, normalmap, twosided, coating) is not implemented");
        std::vector<float> sa = rd.constant("sigmaA"), spec = rd.constant("specularReflectance");
        const uint32_t keepFlags = m.flags;
        memset(&m, 0, sizeof(m)); m.type = MI_BSDF_COATING; m.flags = keepFlags; m.distr = (uint32_t) g_materials->size(); g_materials->push_back(nested);
        m.eta[0] = eta; m.alpha = thickness; memcpy(m.reflectance, sa.data(), 12); memcpy(m.specular, spec.data(), 12);
    } else if (cls == "DiffuseTransmitter") {
        std::vector<float> tr = rd.texture(); m.type = MI_BSDF_DIFFTRANS; bind(tr);
    } else if (cls == "Phong") {                                        // phong.cpp:262-268: diffuse, specular, exponent; the sampling weight as configure() derives it (:104-108)
        std::vector<float> diff = rd.constant("diffuseReflectance"), spec = rd.constant("specularReflectance"), ex = rd.constant("exponent");
        m.type = MI_BSDF_PHONG; memcpy(m.reflectance, diff.data(), 12); memcpy(m.specular, spec.data(), 12); m.alpha = ex[0];
        Spectrum d, s; d.fromLinearRGB(diff[0], diff[1], diff[2]); s.fromLinearRGB(spec[0], spec[1], spec[2]);
        for (int c = 0; c < 3; ++c) if (diff[c] + spec[c] > 1.0f) SLog(EError, "path_hip: phong with diffuseReflectance + specularReflectance > 1 (rescaled by the reference) is not implemented");
        const Float dAvg = d.getLuminance(), sAvg = s.getLuminance(); m.k[0] = sAvg / (dAvg + sAvg);
    } else if (cls == "Ward") {                                         // ward.cpp:340-348: variant, diffuse, specular, alphaU, alphaV; the sampling weight as configure() derives it (:160-164)
        const uint32_t variant = rd.ms->readUInt();
        std::vector<float> diff = rd.constant("diffuseReflectance"), spec = rd.constant("specularReflectance"), au = rd.constant("alphaU"), av = rd.constant("alphaV");
        if (variant > 2) SLog(EError, "path_hip: unknown ward variant");
        m.type = MI_BSDF_WARD; m.distr = variant; memcpy(m.reflectance, diff.data(), 12); memcpy(m.specular, spec.data(), 12); m.alpha = au[0]; m.k[1] = av[0]; if (au[0] != av[0]) m.flags |= MI_BSDF_FLAG_ANISOTROPIC;
        Spectrum d, s; d.fromLinearRGB(diff[0], diff[1], diff[2]); s.fromLinearRGB(spec[0], spec[1], spec[2]);
        for (int c = 0; c < 3; ++c) if (diff[c] + spec[c] > 1.0f) SLog(EError, "path_hip: ward with diffuseReflectance + specularReflectance > 1 (rescaled by the reference) is not implemented");
        const Float dAvg = d.getLuminance(), sAvg = s.getLuminance(); m.k[0] = sAvg / (dAvg + sAvg);
    } else if (cls == "RoughDiffuse") {                                 // roughdiffuse.cpp:269-275: reflectance, alpha, useFastApprox
        std::vector<float> refl = rd.texture(); m.type = MI_BSDF_ROUGHDIFFUSE; bind(refl);
        std::vector<float> a = rd.constant("alpha"); m.alpha = a[0]; m.distr = rd.ms->readBool() ? 1u : 0u;
    } else if (cls == "MixtureBSDF") {                                  // mixturebsdf.cpp:104-113: count, then (weight, BSDF) pairs; the children become records of their own
        const size_t count = rd.ms->readSize();
        if (count < 2 || count > 4) SLog(EError, "path_hip: a mixturebsdf with %i BSDFs is not implemented (2..4)", (int) count);
        const uint32_t keepFlags = m.flags; memset(&m, 0, sizeof(m)); m.type = MI_BSDF_MIXTURE; m.flags = keepFlags; m.distr = (uint32_t) count;
        for (size_t i = 0; i < count; ++i) {
            const float w = rd.ms->readFloat(); mi_material child; memset(&child, 0, sizeof(child));
            if (!readNestedInstance(rd, child) || (child.type >= MI_BSDF_MASK && child.type != MI_BSDF_ROUGHDIFFUSE && child.type != MI_BSDF_PHONG && child.type != MI_BSDF_WARD)) SLog(EError, "path_hip: this BSDF inside a mixturebsdf is not implemented (plain BSDFs, optionally twosided)");
            const float idx = (float) g_materials->size(); g_materials->push_back(child);
            if (i < 3) { m.reflectance[i] = idx; m.k[i] = w; } else { m.eta[0] = idx; m.specular[0] = w; }
        }
    } else if (cls == "BumpMap" || cls == "NormalMap") {                // bumpmap.cpp:104-109 / normalmap.cpp:75-80: the nested BSDF, then the displacement / normal texture
        mi_material nested; memset(&nested, 0, sizeof(nested));
        if (!readNestedInstance(rd, nested) || nested.type == MI_BSDF_MASK || nested.type == MI_BSDF_BUMPMAP || nested.type == MI_BSDF_NORMALMAP) SLog(EError, "path_hip: the BSDF nested in `%s` is not implemented", cls.c_str());
        const uint32_t keepFlags = m.flags; memset(&m, 0, sizeof(m)); m.type = cls == "BumpMap" ? MI_BSDF_BUMPMAP : MI_BSDF_NORMALMAP; m.flags = keepFlags; m.alpha = 1.0f;
        m.distr = (uint32_t) g_materials->size(); g_materials->push_back(nested);
        // the map: a 2-D texture, for the bump map optionally inside one <texture type="scale"> (src/textures/scale.cpp:153-157: nested texture, then the factor)
        const size_t at = rd.ms->getPos(); const uint32_t id = rd.ms->readUInt(); const std::string tcls = id ? rd.ms->readString() : std::string();
        if (tcls == "ScalingTexture" && m.type == MI_BSDF_BUMPMAP) {
            rd.texture(); float sc3[3]; rd.rgb(sc3);
            if (sc3[0] != sc3[1] || sc3[0] != sc3[2]) SLog(EError, "path_hip: a coloured `scale` around a bump map is not implemented");
            m.alpha = sc3[0];
        } else { rd.ms->seek(at); rd.texture(); }
        if (rd.lastTexture < 0) SLog(EError, "path_hip: the map of a `%s` must be a checkerboard, gridtexture or bitmap texture", cls.c_str());
        m.flags |= MI_BSDF_TEXTURE(rd.lastTexture);
    } else return false;
    return true;
}
/// an instance reference inside a serialised BSDF (id, class name, BSDF::serialize bool, content), `twosided` unwrapped
static bool readNestedInstance(NestedReader &rd, mi_material &m) {
    uint32_t id = rd.ms->readUInt(); if (id == 0) return false;
    std::string cls = rd.ms->readString(); rd.ms->readBool();
    if (cls == "TwoSidedBRDF") {
        uint32_t id0 = rd.ms->readUInt(); std::string inner = rd.ms->readString(); rd.ms->readBool();
        m.flags |= MI_BSDF_FLAG_TWOSIDED;
        if (!readSerializedBSDF(rd, inner, m, true)) return false;
        if (rd.ms->readUInt() != id0) SLog(EError, "path_hip: twosided with two different nested BSDFs is not implemented");
        return true;
    }
    return readSerializedBSDF(rd, cls, m, true);
}
static bool convertTwoSided(const BSDF *bsdf, mi_material &m) {
    NestedReader rd; rd.ms = new MemoryStream(); ref<InstanceManager> mgr = new InstanceManager();
    mgr->serialize(rd.ms, bsdf); rd.ms->seek(0);
    rd.ms->readUInt(); if (rd.ms->readString() != "TwoSidedBRDF") return false;
    rd.ms->readBool();                                                  // BSDF::serialize: m_ensureEnergyConservation (bsdf.cpp:43-46)
    uint32_t id0 = rd.ms->readUInt(); std::string cls = rd.ms->readString(); rd.ms->readBool();
    memset(&m, 0, sizeof(m)); m.flags = MI_BSDF_FLAG_TWOSIDED;
    const size_t nTex = g_textures->size(), nLev = g_texLevels->size(), nTxl = g_texTexels->size();
    if (!readSerializedBSDF(rd, cls, m)) { g_textures->resize(nTex); g_texLevels->resize(nLev); g_texTexels->resize(nTxl); return false; }   // twosided(diffuse) and anything else: the generic component check
    if (rd.ms->readUInt() != id0) SLog(EError, "path_hip: twosided with two different nested BSDFs is not implemented");
    return true;
}
/// A spatially varying BSDF outside `twosided`: read from its serialised form like the nested ones
static bool convertSpatiallyVarying(const BSDF *bsdf, mi_material &m) {
    NestedReader rd; rd.ms = new MemoryStream(); ref<InstanceManager> mgr = new InstanceManager();
    mgr->serialize(rd.ms, bsdf); rd.ms->seek(0);
    rd.ms->readUInt(); std::string cls = rd.ms->readString(); rd.ms->readBool();
    memset(&m, 0, sizeof(m));
    return readSerializedBSDF(rd, cls, m);
}

/// BSDF -> mi_material.  Only what the hot path implements; anything else is reported, never silently approximated.
static mi_material convertBSDF(const BSDF *bsdf) {
    mi_material m; memset(&m, 0, sizeof(m));
    if (bsdf->getClass()->getName() == "TwoSidedBRDF" && convertTwoSided(bsdf, m)) return m;
    if (bsdf->getClass()->getName() == "RoughCoating" || bsdf->getClass()->getName() == "BlendBSDF" || bsdf->getClass()->getName() == "SmoothCoating" || bsdf->getClass()->getName() == "Mask" || bsdf->getClass()->getName() == "MixtureBSDF" || bsdf->getClass()->getName() == "BumpMap" || bsdf->getClass()->getName() == "NormalMap") {   // nested BSDFs, weights and maps are private: serialised form
        if (convertSpatiallyVarying(bsdf, m)) return m;
        SLog(EError, "path_hip: this `%s` is not implemented", bsdf->getClass()->getName().c_str());
    }
    if ((bsdf->getType() & BSDF::ESpatiallyVarying) && bsdf->getClass()->getName() != "TwoSidedBRDF") {      // textures are private members: never fall through to the Properties (constants only)
        if (convertSpatiallyVarying(bsdf, m)) return m;
        SLog(EError, "path_hip: spatially varying BSDF \"%s\": textures are implemented on diffuse.reflectance, plastic / roughplastic.diffuseReflectance and difftrans.transmittance", bsdf->getClass()->getName().c_str());
    }
    if (bsdf->getClass()->getName() == "Ward") {
        if (convertSpatiallyVarying(bsdf, m)) return m;
        SLog(EError, "path_hip: this `ward` is not implemented");
    }
    if (bsdf->getClass()->getName() == "Phong") {
        if (convertSpatiallyVarying(bsdf, m)) return m;
        SLog(EError, "path_hip: this `phong` is not implemented");
    }
    if (bsdf->getClass()->getName() == "RoughDiffuse") {      // the constructor wraps its parameters in textures: the serialised form holds them either way
        if (convertSpatiallyVarying(bsdf, m)) return m;
        SLog(EError, "path_hip: this `roughdiffuse` is not implemented");
    }
    if (bsdf->getClass()->getName() == "Null") { m.type = MI_BSDF_NULL; return m; }      // src/bsdfs/null.cpp: the index-matched boundary of a medium
    if (bsdf->getClass()->getName() == "RoughConductor") {
        // same derivation as RoughConductor's constructor (src/bsdfs/roughconductor.cpp:170-207): eta / k from the properties or from
        // data/ior/<material>.{eta,k}.spd, divided by the exterior IOR; isotropic alpha; Beckmann / GGX with visible-normal sampling
        const Properties &props = bsdf->getProperties();
        std::string material = props.getString("material", "Cu"), distr = props.getString("distribution", "beckmann");
        std::transform(material.begin(), material.end(), material.begin(), ::tolower); std::transform(distr.begin(), distr.end(), distr.begin(), ::tolower);
        Spectrum intEta, intK;
        if (material == "none") { intEta = Spectrum(0.0f); intK = Spectrum(1.0f); }
        else {
            ref<FileResolver> fr = Thread::getThread()->getFileResolver(); std::string name = props.getString("material", "Cu");
            intEta.fromContinuousSpectrum(InterpolatedSpectrum(fr->resolve(fs::pathstr("data/ior/" + name + ".eta.spd"))));
            intK.fromContinuousSpectrum(InterpolatedSpectrum(fr->resolve(fs::pathstr("data/ior/" + name + ".k.spd"))));
        }
        Float extEta = lookupIOR(props, "extEta", "air");
        Spectrum eta = props.getSpectrum("eta", intEta) / extEta, k = props.getSpectrum("k", intK) / extEta, spec = props.getSpectrum("specularReflectance", Spectrum(1.0f));
        // MicrofacetDistribution(props) (microfacet.h:98-146): distribution, alpha | alphaU + alphaV, sampleVisible (never for phong / as)
        if (distr != "beckmann" && distr != "ggx" && distr != "phong" && distr != "as") SLog(EError, "Specified an invalid distribution \"%s\", must be \"beckmann\", \"ggx\", or \"phong\"/\"as\"!", distr.c_str());
        m.type = MI_BSDF_ROUGHCONDUCTOR; m.distr = distr == "ggx" ? 1u : distr == "beckmann" ? 0u : 2u;
        m.flags = (props.getBoolean("sampleVisible", true) && m.distr != 2u) ? MI_BSDF_FLAG_SAMPLE_VISIBLE : 0u;
        if (props.hasProperty("alphaU") || props.hasProperty("alphaV")) {
            if (!props.hasProperty("alphaU") || !props.hasProperty("alphaV")) SLog(EError, "Microfacet model: both 'alphaU' and 'alphaV' must be specified.");
            m.alpha = props.getFloat("alphaU"); const Float av = props.getFloat("alphaV");
            if (av != m.alpha) { m.flags |= MI_BSDF_FLAG_ANISOTROPIC; m.reflectance[0] = av; }
        } else m.alpha = props.getFloat("alpha", 0.1f);
        Float r, g, b;
        eta.toLinearRGB(r, g, b); m.eta[0] = r; m.eta[1] = g; m.eta[2] = b;
        k.toLinearRGB(r, g, b); m.k[0] = r; m.k[1] = g; m.k[2] = b;
        spec.toLinearRGB(r, g, b); m.specular[0] = r; m.specular[1] = g; m.specular[2] = b;
        return m;
    }
    {   // smooth conductor / dielectric / plastic: parameters as their constructors derive them (conductor.cpp:155-178, dielectric.cpp:150-167, plastic.cpp:147-168, 199-201)
        const std::string cls = bsdf->getClass()->getName(); const Properties &props = bsdf->getProperties(); Float r, g, b;
        auto rgb3 = [&](const Spectrum &sp, float *dst) { sp.toLinearRGB(r, g, b); dst[0] = r; dst[1] = g; dst[2] = b; };
        if (cls == "SmoothConductor") {
            std::string material = props.getString("material", "Cu"), lower = material; std::transform(lower.begin(), lower.end(), lower.begin(), ::tolower);
            Spectrum intEta, intK;
            if (lower == "none") { intEta = Spectrum(0.0f); intK = Spectrum(1.0f); }
            else {
                ref<FileResolver> fr = Thread::getThread()->getFileResolver();
                intEta.fromContinuousSpectrum(InterpolatedSpectrum(fr->resolve(fs::pathstr("data/ior/" + material + ".eta.spd"))));
                intK.fromContinuousSpectrum(InterpolatedSpectrum(fr->resolve(fs::pathstr("data/ior/" + material + ".k.spd"))));
            }
            Float extEta = lookupIOR(props, "extEta", "air");
            m.type = MI_BSDF_CONDUCTOR; rgb3(props.getSpectrum("eta", intEta) / extEta, m.eta); rgb3(props.getSpectrum("k", intK) / extEta, m.k);
            rgb3(props.getSpectrum("specularReflectance", Spectrum(1.0f)), m.specular);
            return m;
        }
        if (cls == "ThinDielectric") {           // thindielectric.cpp:70-86
            m.type = MI_BSDF_THINDIELECTRIC; m.eta[0] = lookupIOR(props, "intIOR", "bk7") / lookupIOR(props, "extIOR", "air");
            rgb3(props.getSpectrum("specularReflectance", Spectrum(1.0f)), m.specular); rgb3(props.getSpectrum("specularTransmittance", Spectrum(1.0f)), m.reflectance);
            return m;
        }
        if (cls == "SmoothDielectric") {
            m.type = MI_BSDF_DIELECTRIC; m.eta[0] = lookupIOR(props, "intIOR", "bk7") / lookupIOR(props, "extIOR", "air");
            rgb3(props.getSpectrum("specularReflectance", Spectrum(1.0f)), m.specular); rgb3(props.getSpectrum("specularTransmittance", Spectrum(1.0f)), m.reflectance);
            return m;
        }
        if (cls == "RoughDielectric") {
            std::string distr = props.getString("distribution", "beckmann"); std::transform(distr.begin(), distr.end(), distr.begin(), ::tolower);
            // MicrofacetDistribution(props) (microfacet.h:98-146), as for the rough conductor; alphaV travels in k[0]
            if (distr != "beckmann" && distr != "ggx" && distr != "phong" && distr != "as") SLog(EError, "Specified an invalid distribution \"%s\", must be \"beckmann\", \"ggx\", or \"phong\"/\"as\"!", distr.c_str());
            m.type = MI_BSDF_ROUGHDIELECTRIC; m.distr = distr == "ggx" ? 1u : distr == "beckmann" ? 0u : 2u;
            m.flags = (props.getBoolean("sampleVisible", true) && m.distr != 2u) ? MI_BSDF_FLAG_SAMPLE_VISIBLE : 0u;
            if (props.hasProperty("alphaU") || props.hasProperty("alphaV")) {
                if (!props.hasProperty("alphaU") || !props.hasProperty("alphaV")) SLog(EError, "Microfacet model: both 'alphaU' and 'alphaV' must be specified.");
                m.alpha = props.getFloat("alphaU"); const Float av = props.getFloat("alphaV");
                if (av != m.alpha) { m.flags |= MI_BSDF_FLAG_ANISOTROPIC; m.k[0] = av; }
            } else m.alpha = props.getFloat("alpha", 0.1f);
            m.eta[0] = lookupIOR(props, "intIOR", "bk7") / lookupIOR(props, "extIOR", "air");
            rgb3(props.getSpectrum("specularReflectance", Spectrum(1.0f)), m.specular); rgb3(props.getSpectrum("specularTransmittance", Spectrum(1.0f)), m.reflectance);
            return m;
        }
        if (cls == "DiffuseTransmitter") {
            m.type = MI_BSDF_DIFFTRANS; rgb3(props.getSpectrum(props.hasProperty("transmittance") ? "transmittance" : "diffuseTransmittance", Spectrum(.5f)), m.reflectance);   // difftrans.cpp:52-57
            return m;
        }
        if (cls == "RoughPlastic") {
            std::string distr = props.getString("distribution", "beckmann"); std::transform(distr.begin(), distr.end(), distr.begin(), ::tolower);
            if (distr != "beckmann" && distr != "ggx" && distr != "phong" && distr != "as") SLog(EError, "Specified an invalid distribution \"%s\", must be \"beckmann\", \"ggx\", or \"phong\"/\"as\"!", distr.c_str());   // microfacet.h:113-115
            m.distr = distr == "ggx" ? 1u : distr == "beckmann" ? 0u : 2u;
            m.type = MI_BSDF_ROUGHPLASTIC; m.flags = (props.getBoolean("sampleVisible", true) && m.distr != 2u) ? MI_BSDF_FLAG_SAMPLE_VISIBLE : 0u; m.alpha = props.getFloat("alpha", 0.1f);
            m.eta[0] = lookupIOR(props, "intIOR", "polypropylene") / lookupIOR(props, "extIOR", "air");
            if (props.getBoolean("nonlinear", false)) m.flags |= MI_BSDF_FLAG_NONLINEAR;
            rgb3(props.getSpectrum("specularReflectance", Spectrum(1.0f)), m.specular); rgb3(props.getSpectrum("diffuseReflectance", Spectrum(0.5f)), m.reflectance);
            roughPlasticTables(m);
            return m;
        }
        if (cls == "SmoothPlastic") {
            m.type = MI_BSDF_PLASTIC; m.eta[0] = lookupIOR(props, "intIOR", "polypropylene") / lookupIOR(props, "extIOR", "air");
            m.k[0] = fresnelDiffuseReflectance(1 / m.eta[0], false);
            if (props.getBoolean("nonlinear", false)) m.flags |= MI_BSDF_FLAG_NONLINEAR;
            rgb3(props.getSpectrum("specularReflectance", Spectrum(1.0f)), m.specular); rgb3(props.getSpectrum("diffuseReflectance", Spectrum(0.5f)), m.reflectance);
            return m;
        }
    }
    if (bsdf->getType() & BSDF::ESpatiallyVarying)
        SLog(EError, "path_hip: spatially varying BSDF \"%s\" inside twosided: textures are implemented on diffuse.reflectance and plastic / roughplastic.diffuseReflectance", bsdf->getClass()->getName().c_str());
    bool backSide = false;
    for (int i = 0; i < bsdf->getComponentCount(); ++i) {
        unsigned int type = bsdf->getType(i);
        if (!(type & BSDF::EDiffuseReflection))
            SLog(EError, "path_hip: BSDF \"%s\" is not implemented (diffuse, roughconductor, conductor, dielectric, plastic; all but dielectric optionally twosided)", bsdf->getClass()->getName().c_str());
        backSide |= (type & BSDF::EBackSide) != 0;
    }
    Intersection its; its.uv = Point2(0.5f); its.p = Point(0.0f); its.hasUVPartials = false;
    Spectrum refl = bsdf->getDiffuseReflectance(its); Float r, g, b; refl.toLinearRGB(r, g, b);
    m.type = MI_BSDF_DIFFUSE; m.flags = backSide ? MI_BSDF_FLAG_TWOSIDED : 0; m.reflectance[0] = r; m.reflectance[1] = g; m.reflectance[2] = b;
    return m;
}

/// Analytic shape -> mi_analytic.  The shapes keep their transforms private, so each record is rebuilt from the shape's Properties exactly
/// as its constructor does (rectangle.cpp:80-85, disk.cpp:82-87, sphere.cpp:108-132, cylinder.cpp:83-107), with the reference's own Transform code.
static bool convertAnalytic(const Shape *shape, mi_analytic &a) {
    const std::string cls = shape->getClass()->getName(); const Properties &props = shape->getProperties();
    memset(&a, 0, sizeof(a)); a.bsdf = -1; a.emitter = -1; a.radius = 1.0f; a.length = 1.0f;
    Transform toWorld;
    if (cls == "Rectangle") {
        a.type = MI_SHAPE_RECTANGLE; toWorld = props.getTransform("toWorld", Transform());
        if (props.getBoolean("flipNormals", false)) toWorld = toWorld * Transform::scale(Vector(1, 1, -1));
    } else if (cls == "Disk") {
        a.type = MI_SHAPE_DISK;
        ref<const AnimatedTransform> at = props.getAnimatedTransform("toWorld", Transform());
        if (!at->isStatic()) SLog(EError, "path_hip: animated transforms are not implemented");
        toWorld = at->eval(0);
        if (props.getBoolean("flipNormals", false)) toWorld = toWorld * Transform::scale(Vector(1, 1, -1));
    } else if (cls == "Sphere") {
        a.type = MI_SHAPE_SPHERE;
        toWorld = Transform::translate(Vector(props.getPoint("center", Point(0.0f))));
        Float radius = props.getFloat("radius", 1.0f);
        if (props.hasProperty("toWorld")) {
            Transform objectToWorld = props.getTransform("toWorld");
            Float r = objectToWorld(Vector(1, 0, 0)).length();
            toWorld = objectToWorld * Transform::scale(Vector(1 / r)) * toWorld;
            radius *= r;
        }
        a.radius = radius; a.flags = props.getBoolean("flipNormals", false) ? MI_ANALYTIC_FLIP_NORMALS : 0u;
    } else if (cls == "Cylinder") {
        a.type = MI_SHAPE_CYLINDER;
        Float radius = props.getFloat("radius", 1.0f);
        Point p1 = props.getPoint("p0", Point(0.0f, 0.0f, 0.0f)), p2 = props.getPoint("p1", Point(0.0f, 0.0f, 1.0f));
        Vector d = p2 - p1; Float length = d.length();
        toWorld = Transform::translate(Vector(p1)) * Transform::fromFrame(Frame(d / length)) * Transform::scale(Vector(radius, radius, length));
        if (props.hasProperty("toWorld")) toWorld = props.getTransform("toWorld") * toWorld;
        a.radius = toWorld(Vector(1, 0, 0)).length(); a.length = toWorld(Vector(0, 0, 1)).length();
        toWorld = toWorld * Transform::scale(Vector(1 / a.radius, 1 / a.radius, 1 / a.length));
        a.flags = props.getBoolean("flipNormals", false) ? MI_ANALYTIC_FLIP_NORMALS : 0u;
    } else return false;
    const Matrix4x4 &m = toWorld.getMatrix(), &inv = toWorld.getInverseMatrix();
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a.to_world[i * 4 + j] = m(i, j); a.to_object[i * 4 + j] = inv(i, j); }
    return true;
}

static void flatten(const Scene *scene, FlatScene &fs) {
    g_tables = &fs.materialTables; g_materials = &fs.materials; g_textures = &fs.textures; g_texLevels = &fs.texLevels; g_texTexels = &fs.texTexels;
    const std::vector<TriMesh *> &meshes = scene->getMeshes();
    std::map<const BSDF *, int> bsdfIndex; std::vector<const Instance *> insts;
    // non-mesh shapes: rectangle / disk / sphere / cylinder become analytic records (numbered after the meshes); anything else is refused
    for (size_t si = 0; si < scene->getShapes().size(); ++si) {
        const Shape *shape = scene->getShapes()[si].get();
        if (shape->getClass()->derivesFrom(MTS_CLASS(TriMesh))) continue;
        if (shape->getClass()->getName() == "Instance") { insts.push_back(static_cast<const Instance *>(shape)); continue; }
        mi_analytic a;
        if (!convertAnalytic(shape, a))
            SLog(EError, "path_hip: shape \"%s\" is not implemented (triangle meshes, rectangle, disk, sphere, cylinder, instance / shapegroup of meshes)", shape->getClass()->getName().c_str());
        fs.analytic.push_back(a); fs.analyticShapes.push_back(shape);
    }
    // shape groups reached through the instances (src/shapes/instance.cpp, shapegroup.cpp): their meshes follow the scene meshes, tagged with the group
    std::map<const ShapeGroup *, uint32_t> groupIndex; std::vector<const TriMesh *> allMeshes(meshes.begin(), meshes.end()); std::vector<uint32_t> meshGroup(meshes.size(), 0);
    for (const Instance *inst : insts) {
        const ShapeGroup *grp = inst->getShapeGroup();
        if (!groupIndex.count(grp)) {
            uint32_t g = (uint32_t) groupIndex.size(); groupIndex[grp] = g;
            for (const Shape *member : grp->getKDTree()->getShapes()) {
                if (!member->getClass()->derivesFrom(MTS_CLASS(TriMesh))) SLog(EError, "path_hip: shape groups are implemented for triangle meshes (found \"%s\")", member->getClass()->getName().c_str());
                allMeshes.push_back(static_cast<const TriMesh *>(member)); meshGroup.push_back(g + 1);
            }
        }
        if (!inst->getWorldTransform()->isStatic()) SLog(EError, "path_hip: animated transforms are not implemented");
        const Transform &trafo = inst->getWorldTransform()->eval(0);
        mi_instance mi_; memset(&mi_, 0, sizeof(mi_)); mi_.group = groupIndex[grp];
        const Matrix4x4 &m = trafo.getMatrix(), &inv = trafo.getInverseMatrix();
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { mi_.to_world[i * 4 + j] = m(i, j); mi_.to_object[i * 4 + j] = inv(i, j); }
        fs.instances.push_back(mi_);
    }
    for (const TriMesh *mesh : allMeshes) { fs.anyNormals |= mesh->getVertexNormals() != NULL; fs.anyUV |= mesh->getVertexTexcoords() != NULL; }
    for (size_t mi = 0; mi < allMeshes.size(); ++mi) {
        const TriMesh *mesh = allMeshes[mi];
        mi_shape sh; memset(&sh, 0, sizeof(sh));
        sh.first_tri = (uint32_t) (fs.idx.size() / 3); sh.tri_count = (uint32_t) mesh->getTriangleCount();
        sh.first_vert = (uint32_t) (fs.pos.size() / 3); sh.vert_count = (uint32_t) mesh->getVertexCount();
        const Point *p = mesh->getVertexPositions(); const Normal *n = mesh->getVertexNormals();
        for (size_t v = 0; v < mesh->getVertexCount(); ++v) {
            fs.pos.push_back(p[v].x); fs.pos.push_back(p[v].y); fs.pos.push_back(p[v].z);
            if (fs.anyNormals) { fs.nrm.push_back(n ? n[v].x : 0); fs.nrm.push_back(n ? n[v].y : 0); fs.nrm.push_back(n ? n[v].z : 0); }
            if (fs.anyUV) { const Point2 *tc = mesh->getVertexTexcoords(); fs.uv.push_back(tc ? tc[v].x : 0); fs.uv.push_back(tc ? tc[v].y : 0); }
        }
        const Triangle *t = mesh->getTriangles();
        for (size_t k = 0; k < mesh->getTriangleCount(); ++k) for (int c = 0; c < 3; ++c) fs.idx.push_back(sh.first_vert + t[k].idx[c]);
        sh.flags = (n ? 0u : 1u) | (mesh->getVertexTexcoords() ? 2u : 0u);
        const BSDF *bsdf = mesh->getBSDF();
        if (!bsdfIndex.count(bsdf)) { const mi_material cm = convertBSDF(bsdf); bsdfIndex[bsdf] = (int) fs.materials.size(); fs.materials.push_back(cm); }   // (a mask appends its nested record first)
        sh.bsdf = bsdfIndex[bsdf]; sh.emitter = -1; sh.group = meshGroup[mi];
        fs.shapes.push_back(sh);
    }
    for (size_t ai = 0; ai < fs.analytic.size(); ++ai) {
        const BSDF *bsdf = fs.analyticShapes[ai]->getBSDF();
        if (!bsdf) SLog(EError, "path_hip: analytic shape without a BSDF");
        if (!bsdfIndex.count(bsdf)) { const mi_material cm = convertBSDF(bsdf); bsdfIndex[bsdf] = (int) fs.materials.size(); fs.materials.push_back(cm); }   // (a mask appends its nested record first)
        fs.analytic[ai].bsdf = bsdfIndex[bsdf];
    }
    // participating media: `homogeneous` with an `isotropic` / `hg` phase function, referenced by shapes (interior / exterior) and by the sensor.  Its sampling
    // parameters are private -> read from the serialised form: Medium::serialize (medium.cpp:67-72: phase function instance, sigmaA, sigmaS) followed by
    // HomogeneousMedium::serialize (homogeneous.cpp:259-264: strategy, samplingDensity, mediumSamplingWeight)
    {
        std::map<const Medium *, int32_t> mediumIndex;
        auto mediumOf = [&](const Medium *m) -> int32_t {
            if (!m) return -1;
            if (mediumIndex.count(m)) return mediumIndex[m];
            if (m->getClass()->getName() != "HomogeneousMedium") SLog(EError, "path_hip: medium \"%s\" is not implemented (homogeneous)", m->getClass()->getName().c_str());
            const PhaseFunction *ph = m->getPhaseFunction(); const std::string pcls = ph->getClass()->getName();
            if (pcls != "IsotropicPhaseFunction" && pcls != "HGPhaseFunction") SLog(EError, "path_hip: phase function \"%s\" is not implemented (isotropic, hg)", pcls.c_str());
            ref<MemoryStream> ms = new MemoryStream(); ref<InstanceManager> mgr = new InstanceManager(); mgr->serialize(ms, m); ms->seek(0);
            ms->readUInt(); ms->readString();                                   // the medium's instance id and class name
            ms->readUInt(); ms->readString(); if (pcls == "HGPhaseFunction") ms->readFloat();     // the phase function instance (hg.cpp:62-66: its g)
            mi_medium r; memset(&r, 0, sizeof(r));
            { Spectrum a(ms.get()), s(ms.get()); Float x, y, z; a.toLinearRGB(x, y, z); r.sigma_a[0] = x; r.sigma_a[1] = y; r.sigma_a[2] = z; s.toLinearRGB(x, y, z); r.sigma_s[0] = x; r.sigma_s[1] = y; r.sigma_s[2] = z; }
            const int strategy = ms->readInt(); r.sampling_density = ms->readFloat(); r.medium_sampling_weight = ms->readFloat();
            if (strategy > 2) SLog(EError, "path_hip: the `maximum` sampling strategy of the homogeneous medium is not implemented (balance, single, manual)");
            r.strategy = (uint32_t) strategy; r.phase = pcls == "HGPhaseFunction" ? MI_PHASE_HG : MI_PHASE_ISOTROPIC; r.g = pcls == "HGPhaseFunction" ? (float) ph->getMeanCosine() : 0.0f;
            mediumIndex[m] = (int32_t) fs.media.size(); fs.media.push_back(r); return mediumIndex[m];
        };
        std::vector<int32_t> pairs;
        for (size_t mi = 0; mi < allMeshes.size(); ++mi) { pairs.push_back(mediumOf(allMeshes[mi]->getInteriorMedium())); pairs.push_back(mediumOf(allMeshes[mi]->getExteriorMedium())); }
        for (size_t ai = 0; ai < fs.analyticShapes.size(); ++ai) { pairs.push_back(mediumOf(fs.analyticShapes[ai]->getInteriorMedium())); pairs.push_back(mediumOf(fs.analyticShapes[ai]->getExteriorMedium())); }
        fs.sensorMedium = mediumOf(scene->getSensor()->getMedium());
        if (!fs.media.empty()) fs.shapeMedia = pairs;
    }
    // emitters in Scene::getEmitters() order (the order the emitter PDF is built in, scene.cpp:383-388)
    const ref_vector<Emitter> &emitters = scene->getEmitters();
    for (size_t e = 0; e < emitters.size(); ++e) {
        const Emitter *em = emitters[e].get();
        if (em->isEnvironmentEmitter() && em->getClass()->getName() == "EnvironmentMap") {
            // level-0 texels of the (half precision) MIP map, world transform and scale (src/emitters/envmap.cpp)
            ref<Bitmap> bmp = em->getBitmap(Vector2i(-1));
            fs.envW = (uint32_t) bmp->getWidth(); fs.envH = (uint32_t) bmp->getHeight(); const size_t n = (size_t) fs.envW * fs.envH * 3;
            if (bmp->getPixelFormat() != Bitmap::ERGB) SLog(EError, "path_hip: unexpected environment bitmap format");
            if (bmp->getComponentFormat() == Bitmap::EFloat16) { const half *h = bmp->getFloat16Data(); fs.envRGB.resize(n); for (size_t i = 0; i < n; ++i) fs.envRGB[i] = (float) h[i]; }
            else if (bmp->getComponentFormat() == Bitmap::EFloat32) fs.envRGB.assign(bmp->getFloat32Data(), bmp->getFloat32Data() + n);
            else SLog(EError, "path_hip: unexpected environment bitmap component format");
            Matrix4x4 tw = em->getWorldTransform()->eval(0.0f).getMatrix(); for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) fs.envToWorld[i * 4 + j] = tw(i, j);
            fs.envScale = em->getProperties().getFloat("scale", 1.0f);
            {   // MIP pyramid for the filtered camera-ray lookups (envmap.cpp:398-411), built by the reference's own TMIPMap with the settings of envmap.cpp:144-145,
                // 182-185 from level 0 (the map's own pyramid is a private member; its level 0 is what getBitmap returns, already rounded to half precision)
                ref<Bitmap> f32bmp = new Bitmap(Bitmap::ERGB, Bitmap::EFloat32, Vector2i((int) fs.envW, (int) fs.envH));
                memcpy(f32bmp->getFloat32Data(), fs.envRGB.data(), n * sizeof(float));
                Properties rp("lanczos"); rp.setInteger("lobes", 2);
                ref<ReconstructionFilter> rf = static_cast<ReconstructionFilter *>(PluginManager::getInstance()->createObject(MTS_CLASS(ReconstructionFilter), rp)); rf->configure();
                typedef TSpectrum<Float, 3> Color3; typedef TSpectrum<half, 3> Color3h;
                ref<TMIPMap<Color3, Color3h> > mip = new TMIPMap<Color3, Color3h>(f32bmp, Bitmap::ERGB, Bitmap::EFloat, rf, ReconstructionFilter::ERepeat, ReconstructionFilter::EClamp,
                                                                                 EEWA, 10.0f, fs::pathstr(), 0, std::numeric_limits<Float>::infinity(), Spectrum::EIlluminant);
                mi_texture t; memset(&t, 0, sizeof(t)); t.type = MI_TEXTURE_BITMAP; t.uscale = t.vscale = 1.0f; t.wrap_u = 1; t.wrap_v = 0; t.filter = 3; t.max_anisotropy = 10.0f;
                t.first_level = (uint32_t) (fs.texLevels.size() / 3); t.n_levels = (uint32_t) mip->getLevels();
                for (int l = 0; l < mip->getLevels(); ++l) {
                    ref<Bitmap> lb = mip->toBitmap(l); const half *hp = lb->getFloat16Data(); const size_t ln = (size_t) lb->getWidth() * lb->getHeight() * 3;
                    fs.texLevels.push_back((uint32_t) lb->getWidth()); fs.texLevels.push_back((uint32_t) lb->getHeight()); fs.texLevels.push_back((uint32_t) fs.texTexels.size());
                    for (size_t i = 0; i < ln; ++i) fs.texTexels.push_back((float) hp[i]);
                }
                fs.envTexture = (int32_t) fs.textures.size(); fs.textures.push_back(t);
            }
            mi_emitter me; memset(&me, 0, sizeof(me)); me.type = MI_EMITTER_ENVMAP; me.shape = -1; me.weight = em->getSamplingWeight();
            fs.emitters.push_back(me); continue;
        }
        {   // scene-level emitters of the other implemented kinds; parameters as their constructors read them (constant.cpp:49-52, point.cpp:59-71, spot.cpp:70-81, directional.cpp:54-72)
            const std::string cls = em->getClass()->getName(); const Properties &ep = em->getProperties();
            mi_emitter me; memset(&me, 0, sizeof(me)); me.shape = -1; me.weight = em->getSamplingWeight(); Spectrum value; bool known = true;
            if (cls == "ConstantBackgroundEmitter") { me.type = MI_EMITTER_CONSTANT; value = ep.getSpectrum("radiance", Spectrum::getD65()); }
            else if (cls == "PointEmitter") { me.type = MI_EMITTER_POINT; value = ep.getSpectrum("intensity", Spectrum::getD65()); }
            else if (cls == "SpotEmitter") {
                me.type = MI_EMITTER_SPOT; value = ep.getSpectrum("intensity", Spectrum(1.0f));
                me.cutoff = ep.getFloat("cutoffAngle", 20); me.beam = ep.getFloat("beamWidth", me.cutoff * 3.0f / 4.0f);
                if (ep.hasProperty("texture")) SLog(EError, "path_hip: spot emitters with a projection texture are not implemented");
            } else if (cls == "DirectionalEmitter") { me.type = MI_EMITTER_DIRECTIONAL; value = ep.getSpectrum("irradiance", Spectrum::getD65()); }
            else if (cls == "CollimatedBeamEmitter") { me.type = MI_EMITTER_COLLIMATED; value = ep.getSpectrum("power", Spectrum::getD65()); }      // collimated.cpp:60
            else known = false;
            if (known) {
                if (em->getWorldTransform() && !em->getWorldTransform()->isStatic()) SLog(EError, "path_hip: animated transforms are not implemented");
                Matrix4x4 tw = em->getWorldTransform() ? em->getWorldTransform()->eval(0.0f).getMatrix() : Transform().getMatrix();
                for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) me.to_world[i * 4 + j] = tw(i, j);
                Float r, g, b; value.toLinearRGB(r, g, b); me.radiance[0] = r; me.radiance[1] = g; me.radiance[2] = b;
                fs.emitters.push_back(me); continue;
            }
        }
        if (!em->isOnSurface() || em->isEnvironmentEmitter())
            SLog(EError, "path_hip: emitter \"%s\" is not implemented (area, envmap, constant, point, spot, directional)", em->getClass()->getName().c_str());
        const Shape *shape = em->getShape(); int shapeIdx = -1;
        for (size_t mi = 0; mi < meshes.size(); ++mi) if (meshes[mi] == shape) shapeIdx = (int) mi;
        for (size_t ai = 0; ai < fs.analyticShapes.size(); ++ai) if (fs.analyticShapes[ai] == shape) shapeIdx = (int) (meshes.size() + ai);
        if (shapeIdx < 0) SLog(EError, "path_hip: area emitter without a supported shape");
        Intersection its; its.shFrame.n = Normal(0, 0, 1);
        Spectrum rad = em->eval(its, Vector(0, 0, 1)); Float r, g, b; rad.toLinearRGB(r, g, b);      // AreaLight::eval = radiance on the lit side (area.cpp:106-111)
        mi_emitter me; memset(&me, 0, sizeof(me));
        me.type = MI_EMITTER_AREA; me.shape = shapeIdx; me.radiance[0] = r; me.radiance[1] = g; me.radiance[2] = b; me.weight = em->getSamplingWeight();
        if ((size_t) shapeIdx < meshes.size()) fs.shapes[shapeIdx].emitter = (int32_t) fs.emitters.size();
        else fs.analytic[shapeIdx - meshes.size()].emitter = (int32_t) fs.emitters.size();
        fs.emitters.push_back(me);
    }
}

/// mi_scene built from a live scene; owns the handle
struct GpuScene {
    mi_scene *scene = nullptr; int border = 0; Vector2i size;
    ~GpuScene() { if (scene) mi_scene_destroy(scene); }
    void build(const Scene *s, const Sensor *sensor, uint32_t device) {
        FlatScene fs; flatten(s, fs);
        if (scene) { mi_scene_destroy(scene); scene = nullptr; }
        MI_CHECK(mi_scene_create(&scene));
        MI_CHECK(mi_scene_set_triangles(scene, fs.pos.data(), fs.anyNormals ? fs.nrm.data() : NULL, fs.anyUV ? fs.uv.data() : NULL, fs.idx.data(),
                                        (uint32_t) (fs.pos.size() / 3), (uint32_t) (fs.idx.size() / 3), fs.shapes.data(), (uint32_t) fs.shapes.size()));
        if (!fs.analytic.empty()) MI_CHECK(mi_scene_set_analytic(scene, fs.analytic.data(), (uint32_t) fs.analytic.size()));
        if (!fs.instances.empty()) MI_CHECK(mi_scene_set_instances(scene, fs.instances.data(), (uint32_t) fs.instances.size()));
        if (!fs.media.empty()) MI_CHECK(mi_scene_set_media(scene, fs.media.data(), (uint32_t) fs.media.size(), fs.shapeMedia.data(), (uint32_t) (fs.shapeMedia.size() / 2), fs.sensorMedium));
        MI_CHECK(mi_scene_set_materials(scene, fs.materials.data(), (uint32_t) fs.materials.size()));
        if (!fs.textures.empty()) MI_CHECK(mi_scene_set_textures(scene, fs.textures.data(), (uint32_t) fs.textures.size()));
        if (!fs.texLevels.empty()) MI_CHECK(mi_scene_set_texture_data(scene, fs.texLevels.data(), (uint32_t) (fs.texLevels.size() / 3), fs.texTexels.data(), fs.texTexels.size()));
        if (!fs.materialTables.empty()) MI_CHECK(mi_scene_set_material_tables(scene, fs.materialTables.data(), (uint32_t) fs.materialTables.size()));
        MI_CHECK(mi_scene_set_emitters(scene, fs.emitters.data(), (uint32_t) fs.emitters.size()));
        if (fs.envW) MI_CHECK(mi_scene_set_envmap(scene, fs.envRGB.data(), fs.envW, fs.envH, fs.envToWorld, fs.envScale));
        if (fs.envTexture >= 0) MI_CHECK(mi_scene_set_envmap_filter(scene, fs.envTexture));
        // camera: rebuild m_sampleToCamera exactly as PerspectiveCameraImpl::configure does (perspective.cpp:150-157); it is a protected member
        if (!sensor->getClass()->derivesFrom(MTS_CLASS(PerspectiveCamera))) SLog(EError, "path_hip: only the perspective camera is implemented");
        const PerspectiveCamera *cam = static_cast<const PerspectiveCamera *>(sensor);
        const Film *film = sensor->getFilm();
        const Vector2i &filmSize = film->getSize(), &cropSize = film->getCropSize(); const Point2i &cropOffset = film->getCropOffset();
        // crop window (perspective.cpp:129-136, 150-152): the film of the path is the crop window, the camera maps it onto its part of the full frame
        Vector2 relSize((Float) cropSize.x / (Float) filmSize.x, (Float) cropSize.y / (Float) filmSize.y);
        Point2 relOffset((Float) cropOffset.x / (Float) filmSize.x, (Float) cropOffset.y / (Float) filmSize.y);
        Float aspect = cam->getAspect();
        Transform cameraToSample = Transform::scale(Vector(1.0f / relSize.x, 1.0f / relSize.y, 1.0f)) * Transform::translate(Vector(-relOffset.x, -relOffset.y, 0.0f))
                                 * Transform::scale(Vector(-0.5f, -0.5f * aspect, 1.0f)) * Transform::translate(Vector(-1.0f, -1.0f / aspect, 0.0f))
                                 * Transform::perspective(cam->getXFov(), cam->getNearClip(), cam->getFarClip());
        Matrix4x4 s2c = cameraToSample.inverse().getMatrix(), c2w = cam->getWorldTransform(0.0f).getMatrix();
        float a[16], w[16]; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a[i * 4 + j] = s2c(i, j); w[i * 4 + j] = c2w(i, j); }
        MI_CHECK(mi_scene_set_camera(scene, a, w, cam->getNearClip(), cam->getFarClip()));
        const ReconstructionFilter *rf = film->getReconstructionFilter(); std::string fname = rf->getClass()->getName();
        if (fname == "BoxFilter") MI_CHECK(mi_scene_set_film(scene, cropSize.x, cropSize.y, 0, rf->getRadius() - 1e-5f, 0.5f));
        else if (fname == "GaussianFilter") MI_CHECK(mi_scene_set_film(scene, cropSize.x, cropSize.y, 1, 0.5f, rf->getRadius() / 4.0f));
        else if (fname == "TentFilter") MI_CHECK(mi_scene_set_film(scene, cropSize.x, cropSize.y, 2, 1.0f, 0.5f));
        else if (fname == "MitchellNetravaliFilter") MI_CHECK(mi_scene_set_film(scene, cropSize.x, cropSize.y, 3, rf->getProperties().getFloat("B", 1.0f / 3.0f), rf->getProperties().getFloat("C", 1.0f / 3.0f)));
        else if (fname == "CatmullRomFilter") MI_CHECK(mi_scene_set_film(scene, cropSize.x, cropSize.y, 4, 2.0f, 0.5f));
        else if (fname == "LanczosSincFilter") MI_CHECK(mi_scene_set_film(scene, cropSize.x, cropSize.y, 5, rf->getRadius(), 0.5f));
        else SLog(EError, "path_hip: reconstruction filter \"%s\" is not implemented (box, gaussian, tent, mitchell, catmullrom, lanczos)", fname.c_str());
        MI_CHECK(mi_scene_commit(scene, device));
        border = rf->getBorderSize(); size = cropSize;
    }
};

static mi355::Properties convertProps(const Properties &props, const Sampler *sampler) {
    mi355::Properties p;
    p.maxDepth = props.getInteger("maxDepth", -1); p.rrDepth = props.getInteger("rrDepth", 5);
    p.strictNormals = props.getBoolean("strictNormals", false); p.hideEmitters = props.getBoolean("hideEmitters", false);
    {   // build-specific `integrator` = "path" (default) | "volpath_simple" | "volpath": which of the reference's three path-tracing loops runs on the GPU
        const std::string loop = props.getString("integrator", "path");
        if (loop == "path") p.integrator = MI_INTEGRATOR_PATH; else if (loop == "volpath_simple") p.integrator = MI_INTEGRATOR_VOLPATH_SIMPLE; else if (loop == "volpath") p.integrator = MI_INTEGRATOR_VOLPATH;
        else SLog(EError, "path_hip: integrator \"%s\" is not implemented (path, volpath_simple, volpath)", loop.c_str());
    }
    p.device = (uint32_t) props.getInteger("device", 0); p.planesPerBatch = (uint32_t) props.getInteger("planesPerBatch", 0);
    {   // build-specific `devices` = "0,1,2,...": HIP devices to spread the film rows over (one scene replica + one host thread each); default: `device` alone
        const std::string list = props.getString("devices", "");
        for (size_t i = 0; i < list.size();) {
            size_t j = list.find_first_of(", ;", i); if (j == std::string::npos) j = list.size();
            if (j > i) p.devices.push_back((uint32_t) atoi(list.substr(i, j - i).c_str()));
            i = j + 1;
        }
        if (!p.devices.empty()) p.device = p.devices[0];
    }
    p.sampleCount = (uint32_t) sampler->getSampleCount();
    std::string sname = sampler->getClass()->getName();
    if (sname == "SobolSampler") { p.sampler = MI_SAMPLER_SOBOL; p.seed = (uint64_t) sampler->getProperties().getSize("scramble", 0); }
    else {   // `independent` (time-seeded SFMT in the reference) and anything else map to the seedable counter stream
        p.sampler = MI_SAMPLER_INDEPENDENT; p.seed = (uint64_t) props.getSize("seed", 0);
        if (sname != "IndependentSampler") SLog(EWarn, "path_hip: sampler \"%s\" is replaced by the independent counter stream", sname.c_str());
    }
    return p;
}

}  // namespace

/// Responsive face (Integrator2): the object im-mts / `mitsuba` (responsive mode) drive from their worker threads
class PathTracerHIPResponsive : public ResponsiveIntegrator {
public:
    PathTracerHIPResponsive(const Properties &props) : ResponsiveIntegrator(props) { }
    bool preprocess(const Scene *scene, const Sensor *sensor, const Sampler *sampler) override {
        m_host.reset(new mi355::MIPathTracerHIP(convertProps(getProperties(), sampler)));   // throws on bad rrDepth / maxDepth like the reference
        m_gpu.build(scene, sensor, m_host->getProperties().device);
        return m_host->preprocess(m_gpu.scene);
    }
    bool allocate(const Scene &, Sampler *const *, ImageBlock *const *, int threadCount) override { return m_host && m_host->allocate(threadCount); }
    int render(const Scene &scene, const Sensor &sensor, Sampler &sampler, ImageBlock &target, Controls controls, int threadIdx, int threadCount) override {
        if (threadIdx != 0) return 0;
        struct Bridge : mi355::Interrupt {
            PathTracerHIPResponsive *self; const Scene *scene; const Sensor *sensor; Sampler *sampler; ImageBlock *target; Controls controls;
            int progress(mi355::MIPathTracerHIP *, const float *rgba, double spp, mi355::Controls, int ti, int tc) override {
                self->publish(rgba, *target);
                return controls.interrupt ? controls.interrupt->progress(self, *scene, *sensor, *sampler, *target, spp, controls, ti, tc) : 0;
            }
        } bridge; bridge.self = this; bridge.scene = &scene; bridge.sensor = &sensor; bridge.sampler = &sampler; bridge.target = &target; bridge.controls = controls;
        m_rgba.resize((size_t) (m_gpu.size.x + 2 * m_gpu.border) * (m_gpu.size.y + 2 * m_gpu.border) * 4);
        mi355::Controls c{controls.continu, controls.abort, &bridge};
        int rc;
        try { rc = m_host->render(m_rgba.data(), c, threadIdx, threadCount); }
        catch (const std::exception &e) { Log(EError, "%s", e.what()); return -1; }
        publish(m_rgba.data(), target);
        return rc;
    }
    char const *getRealtimeStatistics() override { return m_host ? m_host->getRealtimeStatistics() : nullptr; }
    /// copy the (H+2b)x(W+2b)x4 sums into the caller's ImageBlock (its own border may differ), plain stores
    void publish(const float *rgba, ImageBlock &target) {
        Bitmap *bmp = target.getBitmap(); const int tb = target.getBorderSize(), ch = bmp->getChannelCount();
        const int W = m_gpu.size.x, H = m_gpu.size.y, gb = m_gpu.border, GW = W + 2 * gb, TW = bmp->getSize().x;
        Float *dst = bmp->getFloatData();
        for (int y = -std::min(gb, tb); y < H + std::min(gb, tb); ++y) for (int x = -std::min(gb, tb); x < W + std::min(gb, tb); ++x) {
            const float *s = rgba + ((size_t) (y + gb) * GW + (x + gb)) * 4; Float *d = dst + ((size_t) (y + tb) * TW + (x + tb)) * ch;
            for (int k = 0; k < std::min(ch, 4); ++k) d[k] = s[k];
        }
    }
    MTS_DECLARE_CLASS()
private:
    std::unique_ptr<mi355::MIPathTracerHIP> m_host; GpuScene m_gpu; std::vector<float> m_rgba;
};

/// Classic face: what Scene::render (scene.cpp:475-479) calls from the RenderJob thread
class PathTracerHIP : public MonteCarloIntegrator {
public:
    PathTracerHIP(const Properties &props) : MonteCarloIntegrator(props) { }
    PathTracerHIP(Stream *stream, InstanceManager *manager) : MonteCarloIntegrator(stream, manager) { }
    Spectrum Li(const RayDifferential &, RadianceQueryRecord &) const {
        Log(EError, "path_hip traces whole sample planes on the GPU; per-ray Li() is not offered -- use render() or makeResponsiveIntegrator()");
        return Spectrum(0.0f);
    }
    bool render(Scene *scene, RenderQueue *queue, const RenderJob *job, int, int, int) {
        ref<Sensor> sensor = scene->getSensor(); ref<Film> film = sensor->getFilm();
        mi355::Properties hp = convertProps(getProperties(), scene->getSampler());
        hp.opacity = film->hasAlpha();                                                  // integrator.cpp:160-161
        m_host.reset(new mi355::MIPathTracerHIP(hp));
        m_gpu.build(scene, sensor, m_host->getProperties().device);
        m_host->preprocess(m_gpu.scene);
        mi355::Controls c{nullptr, nullptr, nullptr};     // no preview, no interrupt: one submission (RenderJob::cancel reaches it through cancel() below)
        int rc;
        try { rc = m_host->render(nullptr, c, 0, 1); } catch (const std::exception &e) { Log(EError, "%s", e.what()); return false; }
        if (rc != 0) return false;                                                      // cancelled
        // deliver the raw ImageBlock sums (R,G,B,alpha,weight incl. border) exactly as BlockedRenderProcess::processResult would: Film::put
        ref<ImageBlock> block = new ImageBlock(Bitmap::ESpectrumAlphaWeight, film->getCropSize(), film->getReconstructionFilter());
        block->setOffset(Point2i(0));
        if (mi_render_read_film(m_host->handle(), 0, block->getBitmap()->getFloatData()) != MI_OK) Log(EError, "path_hip: %s", mi_last_error());
        film->put(block);
        if (queue && job) queue->signalRefresh(job);
        return true;
    }
    void cancel() { if (m_host) m_host->cancel(); }
    ref<ResponsiveIntegrator> makeResponsiveIntegrator() { return new PathTracerHIPResponsive(getProperties()); }
    std::string toString() const { return m_host ? m_host->toString() : std::string("MIPathTracerHIP[]"); }
    MTS_DECLARE_CLASS()
private:
    std::unique_ptr<mi355::MIPathTracerHIP> m_host; GpuScene m_gpu;
};

MTS_IMPLEMENT_CLASS(PathTracerHIPResponsive, false, ResponsiveIntegrator)
MTS_IMPLEMENT_CLASS_S(PathTracerHIP, false, MonteCarloIntegrator)
MTS_EXPORT_PLUGIN(PathTracerHIP, "MI355X wavefront path tracer (HIP)");
MTS_NAMESPACE_END

// im-mts lists integrators by scanning plugin binaries for this marker (src/libcore/plugin.cpp:256-309, src/integrators/mark_integrator.cpp:3)
extern "C" { MTS_EXPORT const char *mitsuba_integrator_plugin = "(: mitsuba_integrator_plugin :)"; }
