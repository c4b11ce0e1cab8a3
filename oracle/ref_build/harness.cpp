// oracle/ref_build/harness.cpp -- TEST INFRASTRUCTURE (checker), never shipped, never on the product path.
//
// Our own driver around the REFERENCE's compiled hot path (libmitsuba-core/-render + plugins built by the
// Makefile next to this file from /root/reference).  It reads a flattened scene ("MISCENE2", written by
// mitsuba-im_amd/scenes.py), rebuilds the same scene inside the reference programmatically (the XML loader
// needs pugixml/Xerces, absent here: SURVEY.md §8c) and dumps golden vectors as .npy files:
//   tables  : Sobol direction matrices / vdc matrices, SFMT known answers, TEA values
//   samples : per-(pixel, sampleIndex) Li through MIPathTracer::Li + the sampler values consumed first
//   image   : full SamplingIntegrator::renderBlock render (raw 5-channel ImageBlock) + wall time + ray counters
//   hits    : Scene::rayIntersect records for camera rays
//   units   : warp / BSDF / emitter / camera / filter unit vectors
// It runs only in the build container (the reference cannot travel); the fixtures it writes are data.
#include <mitsuba/mitsuba.h>
#include <mitsuba/core/plugin.h>
#include <mitsuba/core/statistics.h>
#include <mitsuba/core/fstream.h>
#include <mitsuba/core/bitmap.h>
#include <mitsuba/core/sched.h>
#include <mitsuba/core/random.h>
#include <mitsuba/core/qmc.h>
#include <mitsuba/core/warp.h>
#include <mitsuba/core/timer.h>
#include <mitsuba/render/scene.h>
#include <mitsuba/render/trimesh.h>
#include <mitsuba/render/integrator.h>
#include <mitsuba/render/medium.h>
#include <mitsuba/render/phase.h>
#include <mitsuba/render/sampler.h>
#include <mitsuba/render/sensor.h>
#include <mitsuba/render/film.h>
#include <mitsuba/render/emitter.h>
#include <mitsuba/render/bsdf.h>
#include <mitsuba/render/imageblock.h>
#include <mitsuba/render/triaccel.h>
#include <mitsuba/render/skdtree.h>
#include <sobolseq.h>
#include <mitsuba/core/fresolver.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
#include <thread>
#include <atomic>
#include <mutex>
#include <unistd.h>

using namespace mitsuba;

// ------------------------------------------------------------------------------------------ npy writer
static void save_npy(const std::string &path, const char *descr, const std::vector<size_t> &shape,
                     const void *data, size_t bytes) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path.c_str()); _exit(2); }
    std::string sh = "(";
    for (size_t i = 0; i < shape.size(); ++i) { sh += std::to_string(shape[i]); sh += (shape.size() == 1 || i + 1 < shape.size()) ? "," : ""; }
    sh += ")";
    std::string hdr = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': " + sh + ", }";
    size_t total = 10 + hdr.size() + 1;
    size_t pad = (64 - total % 64) % 64;
    hdr += std::string(pad, ' ') + "\n";
    unsigned char pre[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(hdr.size() & 0xff), (unsigned char)(hdr.size() >> 8)};
    fwrite(pre, 1, 10, f); fwrite(hdr.data(), 1, hdr.size(), f); fwrite(data, 1, bytes, f); fclose(f);
}
template <typename T> static void save(const std::string &p, const char *d, std::vector<size_t> shape, const std::vector<T> &v) {
    save_npy(p, d, shape, v.data(), v.size() * sizeof(T));
}

#include <rtrans.h>   // src/bsdfs/rtrans.h: RoughTransmittance (tables for the `tables` mode)
struct RTAccess : RoughTransmittance {   // its slices are protected members
    RTAccess(MicrofacetDistribution::EType t) : RoughTransmittance(t) { }
    const Float *trans() const { return m_trans; } size_t thetaSamples() const { return m_thetaSamples; }
};

// ------------------------------------------------------------------------------------------ scene file
struct FShape { uint32_t firstTri, triCount, firstVert, vertCount; int32_t bsdf, emitter; uint32_t faceNormals, pad; };
struct FBsdf { uint32_t type, twosided, distr, sampleVisible; float refl[3], alpha, eta[3], k[3], spec[3]; };
struct FEmitter { uint32_t type; int32_t shape; float radiance[3], weight, cutoff, beam, toWorld[16]; };
struct FTexture { uint32_t type; float color0[3], color1[3], lineWidth, uoffset, voffset, uscale, vscale; uint32_t wrapU, wrapV, filter; float maxAnisotropy; uint32_t baseW, baseH; std::vector<float> base; };
struct FInstance { uint32_t group, pad[3]; float toWorld[16], toObject[16]; };
struct FAnalytic { uint32_t type; int32_t bsdf, emitter; uint32_t flags; float toWorld[16], toObject[16], radius, length; };
struct FMedium { float sigmaA[3], sigmaS[3]; uint32_t strategy; float samplingDensity, mediumSamplingWeight; uint32_t phase; float g; uint32_t pad; };
struct FScene {
    std::vector<FMedium> media; std::vector<int32_t> shapeMedia; int32_t integrator = 0, sensorMedium = -1;     // MEDI section: volpath_simple scenes
    std::vector<FAnalytic> analytic; std::vector<FInstance> instances; std::vector<FTexture> textures; std::vector<int32_t> bsdfTexture;
    int32_t crop[4] = {0, 0, 0, 0};       // full film width / height, crop offset x / y (W, H = the crop window); all zero: no crop window
    uint32_t nVerts, nTris, nShapes, nBsdfs, nEmitters, hasN, hasUV, hasEnv;
    std::vector<float> pos, nrm, uv; std::vector<uint32_t> idx;
    std::vector<FShape> shapes; std::vector<FBsdf> bsdfs; std::vector<FEmitter> emitters;
    float camToWorld[16], fov, nearClip, farClip; uint32_t W, H;
    uint32_t filter; float filterRadius, filterStddev;
    int32_t maxDepth, rrDepth; uint32_t strictNormals, hideEmitters;
    uint32_t sampler, spp; uint64_t seed;
    uint32_t envW, envH; float envToWorld[16], envScale; std::vector<float> envRGB;
};
static void rd(FILE *f, void *p, size_t n) { if (fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); _exit(2); } }
static FScene loadScene(const char *path) {
    FScene s; FILE *f = fopen(path, "rb"); if (!f) { fprintf(stderr, "cannot open %s\n", path); _exit(2); }
    char magic[8]; rd(f, magic, 8); if (memcmp(magic, "MISCENE2", 8)) { fprintf(stderr, "bad magic\n"); _exit(2); }
    rd(f, &s.nVerts, 32);
    s.pos.resize(s.nVerts * 3); rd(f, s.pos.data(), s.pos.size() * 4);
    if (s.hasN) { s.nrm.resize(s.nVerts * 3); rd(f, s.nrm.data(), s.nrm.size() * 4); }
    if (s.hasUV) { s.uv.resize(s.nVerts * 2); rd(f, s.uv.data(), s.uv.size() * 4); }
    s.idx.resize(s.nTris * 3); rd(f, s.idx.data(), s.idx.size() * 4);
    s.shapes.resize(s.nShapes); rd(f, s.shapes.data(), s.nShapes * sizeof(FShape));
    s.bsdfs.resize(s.nBsdfs); rd(f, s.bsdfs.data(), s.nBsdfs * sizeof(FBsdf));
    s.emitters.resize(s.nEmitters); rd(f, s.emitters.data(), s.nEmitters * sizeof(FEmitter));
    rd(f, s.camToWorld, 64); rd(f, &s.fov, 12); rd(f, &s.W, 8);
    rd(f, &s.filter, 4); rd(f, &s.filterRadius, 8);
    rd(f, &s.maxDepth, 8); rd(f, &s.strictNormals, 8);
    rd(f, &s.sampler, 8); rd(f, &s.seed, 8);
    if (s.hasEnv) {
        rd(f, &s.envW, 8); rd(f, s.envToWorld, 64); rd(f, &s.envScale, 4);
        s.envRGB.resize((size_t) s.envW * s.envH * 3); rd(f, s.envRGB.data(), s.envRGB.size() * 4);
    }
    char tag[4];
    while (fread(tag, 1, 4, f) == 4) {
        uint32_t n; rd(f, &n, 4);
        if (!memcmp(tag, "ANLY", 4)) { s.analytic.resize(n); for (FAnalytic &a : s.analytic) rd(f, &a, sizeof(FAnalytic)); }
        else if (!memcmp(tag, "TEXR", 4)) {
            s.textures.resize(n);
            for (FTexture &t : s.textures) {
                rd(f, &t.type, 48); rd(f, &t.wrapU, 24);
                if (t.type == 2) { rd(f, &t.baseW, 8); t.base.resize((size_t) t.baseW * t.baseH * 3); rd(f, t.base.data(), t.base.size() * 4); }
            }
            s.bsdfTexture.resize(s.nBsdfs); rd(f, s.bsdfTexture.data(), s.nBsdfs * 4);
        }
        else if (!memcmp(tag, "CROP", 4)) { rd(f, s.crop, 16); }
        else if (!memcmp(tag, "MEDI", 4)) {
            s.media.resize(n); for (FMedium &m : s.media) rd(f, &m, sizeof(FMedium));
            rd(f, &s.integrator, 4); rd(f, &s.sensorMedium, 4); uint32_t pairs; rd(f, &pairs, 4); s.shapeMedia.resize(pairs * 2); rd(f, s.shapeMedia.data(), pairs * 8);
        }
        else if (!memcmp(tag, "INST", 4)) { s.instances.resize(n); for (FInstance &a : s.instances) rd(f, &a, sizeof(FInstance)); }
        else { fprintf(stderr, "unknown section\n"); _exit(2); }
    }
    fclose(f); return s;
}

// ------------------------------------------------------------------------------------------ build-defined "independent" stream
// The reference's `independent` sampler seeds SFMT from time() (src/samplers/independent.cpp:58, src/libcore/random.cpp:477),
// so "identical seeds" has to be defined by the build (SURVEY.md §7 "Determinism contract"): a counter-based stream over
// the reference's own sampleTEA (include/mitsuba/core/qmc.h:146).  This subclass mirrors mitsuba-im_amd's definition
// (DESIGN.md "independent stream") inside the reference so MIPathTracer::Li consumes exactly the same numbers.
class SeededIndependent : public Sampler {
public:
    SeededIndependent(size_t spp, uint32_t seed, int width) : Sampler(Properties()), m_seed(seed), m_width(width) { m_sampleCount = spp; }
    ref<Sampler> clone() { ref<SeededIndependent> s = new SeededIndependent(m_sampleCount, m_seed, m_width); return s.get(); }
    void generate(const Point2i &pos, size_t nextSampleIdx) override {
        m_pixel = (uint32_t) (pos.y * m_width + pos.x);
        setSampleIndex(nextSampleIdx != (size_t) ~0 ? nextSampleIdx : m_sampleIndex);
    }
    void advance() { setSampleIndex(m_sampleIndex + 1); }
    void setSampleIndex(size_t idx) { m_sampleIndex = idx; m_call = 0; }
    inline uint64_t draw() {
        uint32_t v0 = m_pixel ^ (m_seed * 0x9E3779B9u);
        uint32_t v1 = ((uint32_t) m_sampleIndex << 8) | (m_call++ & 0xFFu);
        return sampleTEA(v0, v1, 4);
    }
    static inline float toFloat(uint32_t bits) { union { uint32_t u; float f; } x; x.u = (bits >> 9) | 0x3f800000u; return x.f - 1.0f; }
    Float next1D() { return toFloat((uint32_t) draw()); }
    Point2 next2D() { uint64_t r = draw(); return Point2(toFloat((uint32_t) r), toFloat((uint32_t) (r >> 32))); }
    std::string toString() const { return "SeededIndependent[]"; }
    MTS_DECLARE_CLASS()
private:
    uint32_t m_seed, m_pixel = 0, m_call = 0; int m_width;
};
MTS_IMPLEMENT_CLASS(SeededIndependent, false, Sampler)

// A recording proxy: forwards to the wrapped sampler and logs every value handed to the integrator.
class RecordingSampler : public Sampler {
public:
    RecordingSampler(Sampler *inner) : Sampler(Properties()), m_inner(inner) { m_sampleCount = inner->getSampleCount(); }
    ref<Sampler> clone() { return new RecordingSampler(m_inner->clone()); }
    void setFilmResolution(const Vector2i &res, bool blocked) { m_inner->setFilmResolution(res, blocked); }
    void generate(const Point2i &pos, size_t idx) override { m_inner->generate(pos, idx); m_sampleIndex = m_inner->getSampleIndex(); log.clear(); }
    void advance() { m_inner->advance(); m_sampleIndex = m_inner->getSampleIndex(); log.clear(); }
    void setSampleIndex(size_t i) { m_inner->setSampleIndex(i); m_sampleIndex = i; log.clear(); }
    Float next1D() { Float v = m_inner->next1D(); log.push_back(v); return v; }
    Point2 next2D() { Point2 v = m_inner->next2D(); log.push_back(v.x); log.push_back(v.y); return v; }
    std::string toString() const { return "RecordingSampler[]"; }
    std::vector<float> log;
    MTS_DECLARE_CLASS()
private:
    ref<Sampler> m_inner;
};
MTS_IMPLEMENT_CLASS(RecordingSampler, false, Sampler)

// BSDFs with EUsesSampler (roughdielectric) draw from bRec.sampler: the unit mode hands them this constant stream (0.5)
class ConstSampler : public Sampler {
public:
    ConstSampler() : Sampler(Properties()) { m_sampleCount = 1; }
    ref<Sampler> clone() { return new ConstSampler(); }
    Float next1D() { return 0.5f; }
    Point2 next2D() { return Point2(0.5f); }
    std::string toString() const { return "ConstSampler[]"; }
    MTS_DECLARE_CLASS()
};
MTS_IMPLEMENT_CLASS(ConstSampler, false, Sampler)

// ------------------------------------------------------------------------------------------ scene construction
static ConfigurableObject *create(const Class *cls, const Properties &p) {
    return PluginManager::getInstance()->createObject(cls, p);
}
static Spectrum rgb(const float *c) { Spectrum s; s.fromLinearRGB(c[0], c[1], c[2]); return s; }

struct Built {
    ref<Scene> scene; ref<Sensor> sensor; ref<Sampler> sampler; ref<SamplingIntegrator> integrator; ref<Film> film;
    ref<ReconstructionFilter> filter;
};

static Built buildScene(const FScene &fs) {
    Built b;
    b.scene = new Scene();
    // BSDFs
    std::vector<ref<BSDF> > bsdfs;
    // texture record -> the reference's texture plugin (bitmaps from an in-memory image: the plugin builds its own MIP pyramid); asNormals: the bump / normal
    // map adapters insist on an explicit gamma for bitmaps (bumpmap.cpp:122-126)
    auto makeTexture = [&](const FTexture &ft, bool explicitGamma) -> ref<Texture> {
        Properties tp(ft.type == 0 ? "checkerboard" : ft.type == 1 ? "gridtexture" : "bitmap");
        ref<Bitmap> bmp;
        if (ft.type == 2) {
            static const char *wraps[] = {"clamp", "repeat", "mirror", "zero", "one"}; static const char *filters[] = {"nearest", "bilinear", "trilinear", "ewa"};
            bmp = new Bitmap(Bitmap::ERGB, Bitmap::EFloat32, Vector2i(ft.baseW, ft.baseH)); memcpy(bmp->getFloat32Data(), ft.base.data(), ft.base.size() * 4);
            tp.setData("bitmap", Properties::Data{(uint8_t *) bmp.get(), sizeof(Bitmap)});
            tp.setString("wrapModeU", wraps[ft.wrapU]); tp.setString("wrapModeV", wraps[ft.wrapV]); tp.setString("filterType", filters[ft.filter]);
            tp.setFloat("maxAnisotropy", ft.maxAnisotropy);
            if (explicitGamma) tp.setFloat("gamma", 1.0f);
        } else { tp.setSpectrum("color0", rgb(ft.color0)); tp.setSpectrum("color1", rgb(ft.color1)); }
        if (ft.type == 1) tp.setFloat("lineWidth", ft.lineWidth);
        tp.setFloat("uoffset", ft.uoffset); tp.setFloat("voffset", ft.voffset); tp.setFloat("uscale", ft.uscale); tp.setFloat("vscale", ft.vscale);
        ref<Texture> tex = static_cast<Texture *>(create(MTS_CLASS(Texture), tp)); tex->configure();
        return tex;
    };
    for (const FBsdf &fb : fs.bsdfs) {
        ref<BSDF> bsdf;
        if (fb.type == 10) {          // mixturebsdf: children = EARLIER records (indices in refl[0..2], eta[0]), weights in k[0..2], spec[0]
            std::string w;
            for (uint32_t i = 0; i < fb.distr; ++i) { char buf[64]; snprintf(buf, sizeof(buf), "%s%.9g", i ? ", " : "", (double) (i < 3 ? fb.k[i] : fb.spec[0])); w += buf; }
            Properties p("mixturebsdf"); p.setString("weights", w);
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
            for (uint32_t i = 0; i < fb.distr; ++i) {
                size_t c = (size_t) (i < 3 ? fb.refl[i] : fb.eta[0]);
                if (c >= bsdfs.size()) { fprintf(stderr, "mixturebsdf: the children must precede it\n"); _exit(2); }
                bsdf->addChild(bsdfs[c]); bsdfs[c]->setParent(bsdf);
            }
        } else if (fb.type == 11 || fb.type == 12) {      // bumpmap / normalmap: nested = an earlier record, the bound texture = displacement / normals
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), Properties(fb.type == 11 ? "bumpmap" : "normalmap")));
            if (fb.distr >= bsdfs.size()) { fprintf(stderr, "bumpmap: the nested material must precede it\n"); _exit(2); }
            bsdf->addChild(bsdfs[fb.distr]); bsdfs[fb.distr]->setParent(bsdf);
            ref<Texture> tex = makeTexture(fs.textures[fs.bsdfTexture[bsdfs.size()]], true);
            if (fb.type == 11 && fb.alpha != 1.0f) {        // <texture type="scale"> around the displacement
                Properties sp("scale"); sp.setFloat("scale", fb.alpha);
                ref<Texture> st = static_cast<Texture *>(create(MTS_CLASS(Texture), sp)); st->addChild(tex); tex->setParent(st); st->configure(); tex = st;
            }
            bsdf->addChild(tex); tex->setParent(bsdf);
        } else if (fb.type == 0) {
            Properties p("diffuse"); p.setSpectrum("reflectance", rgb(fb.refl));
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
        } else if (fb.type == 2) {
            Properties p("conductor");
            p.setSpectrum("eta", rgb(fb.eta)); p.setSpectrum("k", rgb(fb.k)); p.setSpectrum("specularReflectance", rgb(fb.spec));
            p.setString("material", "none"); p.setFloat("extEta", 1.0f);
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
        } else if (fb.type == 3) {
            Properties p("dielectric");
            p.setFloat("intIOR", fb.eta[0]); p.setFloat("extIOR", 1.0f);
            p.setSpectrum("specularReflectance", rgb(fb.spec)); p.setSpectrum("specularTransmittance", rgb(fb.refl));
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
        } else if (fb.type == 7) {
            Properties p("roughplastic");
            p.setString("distribution", fb.distr == 0 ? "beckmann" : fb.distr == 1 ? "ggx" : "phong"); p.setFloat("alpha", fb.alpha);
            p.setFloat("intIOR", fb.eta[0]); p.setFloat("extIOR", 1.0f); p.setBoolean("sampleVisible", (fb.sampleVisible & 1u) != 0);
            p.setSpectrum("specularReflectance", rgb(fb.spec)); p.setSpectrum("diffuseReflectance", rgb(fb.refl));
            p.setBoolean("nonlinear", (fb.sampleVisible & 2u) != 0);   // FBsdf::sampleVisible of a roughplastic: bit 0 sampleVisible, bit 1 nonlinear
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
        } else if (fb.type == 5) {
            Properties p("roughdielectric");
            p.setString("distribution", fb.distr == 0 ? "beckmann" : fb.distr == 1 ? "ggx" : "phong");
            if (fb.sampleVisible & 4u) { p.setFloat("alphaU", fb.alpha); p.setFloat("alphaV", fb.k[0]); } else p.setFloat("alpha", fb.alpha);     // anisotropic: alphaV travels in k[0]
            p.setFloat("intIOR", fb.eta[0]); p.setFloat("extIOR", 1.0f); p.setBoolean("sampleVisible", (fb.sampleVisible & 1u) != 0);
            p.setSpectrum("specularReflectance", rgb(fb.spec)); p.setSpectrum("specularTransmittance", rgb(fb.refl));
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
        } else if (fb.type == 17) {          // coating: eta[0] = intIOR / extIOR, alpha = thickness, refl = sigmaA, spec = specularReflectance, nested BSDF = an EARLIER record (index in distr)
            Properties p("coating"); p.setFloat("intIOR", fb.eta[0]); p.setFloat("extIOR", 1.0f); p.setFloat("thickness", fb.alpha);
            p.setSpectrum("sigmaA", rgb(fb.refl)); p.setSpectrum("specularReflectance", rgb(fb.spec));
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
            if (fb.distr >= bsdfs.size()) { fprintf(stderr, "coating: the nested material must precede it\n"); _exit(2); }
            bsdf->addChild(bsdfs[fb.distr]); bsdfs[fb.distr]->setParent(bsdf);
        } else if (fb.type == 19) {          // roughcoating: eta[0] = intIOR / extIOR, eta[1] = thickness, eta[2] = distribution, alpha, refl = sigmaA, spec, nested = an EARLIER record (distr)
            Properties p("roughcoating"); p.setFloat("intIOR", fb.eta[0]); p.setFloat("extIOR", 1.0f); p.setFloat("thickness", fb.eta[1]);
            const int mf = (int) fb.eta[2]; p.setString("distribution", mf == 0 ? "beckmann" : mf == 1 ? "ggx" : "phong"); p.setFloat("alpha", fb.alpha);
            p.setBoolean("sampleVisible", (fb.sampleVisible & 1u) != 0);
            p.setSpectrum("sigmaA", rgb(fb.refl)); p.setSpectrum("specularReflectance", rgb(fb.spec));
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
            if (fb.distr >= bsdfs.size()) { fprintf(stderr, "roughcoating: the nested material must precede it\n"); _exit(2); }
            bsdf->addChild(bsdfs[fb.distr]); bsdfs[fb.distr]->setParent(bsdf);
        } else if (fb.type == 18) {          // blendbsdf: children = EARLIER records (indices in eta[0], eta[1]); weight = refl[0] (constant) or the bound texture
            Properties p("blendbsdf"); p.setFloat("weight", fb.refl[0]);
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
            for (int i = 0; i < 2; ++i) { size_t c = (size_t) fb.eta[i]; if (c >= bsdfs.size()) { fprintf(stderr, "blendbsdf: the children must precede it\n"); _exit(2); }
                                          bsdf->addChild(bsdfs[c]); bsdfs[c]->setParent(bsdf); }
        } else if (fb.type == 9) {           // mask: opacity in refl (or the bound texture), nested BSDF = an EARLIER record (index in distr)
            Properties p("mask"); p.setSpectrum("opacity", rgb(fb.refl));
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
            if (fb.distr >= bsdfs.size()) { fprintf(stderr, "mask: the nested material must precede it\n"); _exit(2); }
            bsdf->addChild(bsdfs[fb.distr]); bsdfs[fb.distr]->setParent(bsdf);
        } else if (fb.type == 13) {
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), Properties("null")));
        } else if (fb.type == 8) {
            Properties p("thindielectric");
            p.setFloat("intIOR", fb.eta[0]); p.setFloat("extIOR", 1.0f);
            p.setSpectrum("specularReflectance", rgb(fb.spec)); p.setSpectrum("specularTransmittance", rgb(fb.refl));
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
        } else if (fb.type == 6) {
            Properties p("difftrans"); p.setSpectrum("transmittance", rgb(fb.refl));
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
        } else if (fb.type == 14) {          // roughdiffuse (Oren-Nayar): distr = useFastApprox
            Properties p("roughdiffuse"); p.setSpectrum("reflectance", rgb(fb.refl)); p.setFloat("alpha", fb.alpha); p.setBoolean("useFastApprox", fb.distr == 1);
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
        } else if (fb.type == 15) {          // phong: alpha = exponent
            Properties p("phong"); p.setSpectrum("diffuseReflectance", rgb(fb.refl)); p.setSpectrum("specularReflectance", rgb(fb.spec)); p.setFloat("exponent", fb.alpha);
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
        } else if (fb.type == 16) {          // ward: alpha = alphaU, k[1] = alphaV, distr = variant
            Properties p("ward"); p.setSpectrum("diffuseReflectance", rgb(fb.refl)); p.setSpectrum("specularReflectance", rgb(fb.spec));
            p.setString("variant", fb.distr == 0 ? "ward" : fb.distr == 1 ? "ward-duer" : "balanced"); p.setFloat("alphaU", fb.alpha); p.setFloat("alphaV", fb.k[1]);
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
        } else if (fb.type == 4) {
            Properties p("plastic");
            p.setFloat("intIOR", fb.eta[0]); p.setFloat("extIOR", 1.0f);
            p.setSpectrum("specularReflectance", rgb(fb.spec)); p.setSpectrum("diffuseReflectance", rgb(fb.refl));
            p.setBoolean("nonlinear", fb.distr == 1);
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
        } else {
            Properties p("roughconductor");
            p.setString("distribution", fb.distr == 0 ? "beckmann" : fb.distr == 1 ? "ggx" : "phong");
            if (fb.sampleVisible & 4u) { p.setFloat("alphaU", fb.alpha); p.setFloat("alphaV", fb.refl[0]); }     // anisotropic: alphaV travels in refl[0]
            else p.setFloat("alpha", fb.alpha);
            p.setSpectrum("eta", rgb(fb.eta)); p.setSpectrum("k", rgb(fb.k));
            p.setSpectrum("specularReflectance", rgb(fb.spec));
            p.setBoolean("sampleVisible", (fb.sampleVisible & 1u) != 0);
            p.setString("material", "none"); p.setFloat("extEta", 1.0f);   // eta / k are given as RGB, already relative to the exterior
            bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
        }
        { size_t bi = bsdfs.size();
          if (bi < fs.bsdfTexture.size() && fs.bsdfTexture[bi] >= 0 && fb.type != 11 && fb.type != 12 && fb.type != 17 && fb.type != 19) {      // texture bound to the record's `reflectance`
              ref<Texture> tex = makeTexture(fs.textures[fs.bsdfTexture[bi]], false);
              // the texture drives diffuse.reflectance, plastic / roughplastic.diffuseReflectance, difftrans.transmittance or mask.opacity (the material record's `reflectance`)
              bsdf->addChild(fb.type == 4 || fb.type == 7 ? "diffuseReflectance" : fb.type == 6 ? "transmittance" : fb.type == 9 ? "opacity" : fb.type == 18 ? "weight" : "reflectance", tex); tex->setParent(bsdf);
          } }
        bsdf->configure();
        if (fb.twosided) {
            ref<BSDF> ts = static_cast<BSDF *>(create(MTS_CLASS(BSDF), Properties("twosided")));
            ts->addChild(bsdf); bsdf->setParent(ts); ts->configure();
            bsdf = ts;
        }
        bsdfs.push_back(bsdf);
    }
    // scene-level emitters first, as the XML loader would add them before shapes are expanded
    for (const FEmitter &fe : fs.emitters) {
        if (fe.type == 6) {      // sunsky (src/emitters/sunsky.cpp): a COMPOUND emitter -- Scene::addChild (scene.cpp:530-539) adds its elements: the rasterised `envmap`, and a
                                 // `directional` sun when sunRadiusScale = 0.  radiance = (turbidity, scale, sunRadiusScale), cutoff = resolution, sun direction = z axis of toWorld
            Properties p("sunsky"); p.setFloat("samplingWeight", fe.weight);
            p.setFloat("turbidity", fe.radiance[0]); p.setFloat("scale", fe.radiance[1]); p.setFloat("sunRadiusScale", fe.radiance[2]); p.setInteger("resolution", (int) fe.cutoff);
            p.setVector("sunDirection", Vector(fe.toWorld[2], fe.toWorld[6], fe.toWorld[10]));
            ref<Emitter> em = static_cast<Emitter *>(create(MTS_CLASS(Emitter), p)); em->configure();
            b.scene->addChild(em);
            continue;
        }
        if (fe.type >= 2) {      // constant / point / spot / directional
            static const char *names[] = {"", "", "constant", "point", "spot", "directional", "", "collimated"};
            Properties p(names[fe.type]);
            p.setFloat("samplingWeight", fe.weight);
            Matrix4x4 m; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m(i, j) = fe.toWorld[i * 4 + j];
            if (fe.type == 2) p.setSpectrum("radiance", rgb(fe.radiance));
            else if (fe.type == 5) { p.setSpectrum("irradiance", rgb(fe.radiance)); p.setTransform("toWorld", Transform(m)); }
            else if (fe.type == 7) { p.setSpectrum("power", rgb(fe.radiance)); p.setTransform("toWorld", Transform(m)); }
            else { p.setSpectrum("intensity", rgb(fe.radiance)); p.setTransform("toWorld", Transform(m)); }
            if (fe.type == 4) { p.setFloat("cutoffAngle", fe.cutoff); p.setFloat("beamWidth", fe.beam); }
            ref<Emitter> em = static_cast<Emitter *>(create(MTS_CLASS(Emitter), p));
            b.scene->addChild(em); em->setParent(b.scene); em->configure();
            continue;
        }
        if (fe.type != 1) continue;
        ref<Bitmap> bmp = new Bitmap(Bitmap::ERGB, Bitmap::EFloat32, Vector2i(fs.envW, fs.envH));
        memcpy(bmp->getFloat32Data(), fs.envRGB.data(), fs.envRGB.size() * 4);
        Properties p("envmap");
        p.setData("bitmap", Properties::Data{(uint8_t *) bmp.get(), sizeof(Bitmap)});
        Matrix4x4 m; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m(i, j) = fs.envToWorld[i * 4 + j];
        p.setTransform("toWorld", Transform(m)); p.setFloat("scale", fs.envScale);
        p.setFloat("samplingWeight", fe.weight);
        ref<Emitter> em = static_cast<Emitter *>(create(MTS_CLASS(Emitter), p));
        b.scene->addChild(em); em->setParent(b.scene);
        // configured after the scene knows its bounds (Scene::initialize -> emitter->createShape)
        em->configure();
    }
    // participating media: `homogeneous` + its phase function (the constructor derives mediumSamplingWeight / the `single` channel itself)
    std::vector<ref<Medium> > media;
    for (const FMedium &fm : fs.media) {
        static const char *strategies[] = {"balance", "single", "manual"};
        Properties p("homogeneous"); p.setSpectrum("sigmaA", rgb(fm.sigmaA)); p.setSpectrum("sigmaS", rgb(fm.sigmaS)); p.setFloat("g", 0.0f);    // g: else the default preset's anisotropy rescales sigmaS (medium.cpp:30-34)
        p.setString("strategy", strategies[fm.strategy]); if (fm.strategy == 2) p.setFloat("samplingDensity", fm.samplingDensity);
        ref<Medium> med = static_cast<Medium *>(create(MTS_CLASS(Medium), p));
        Properties pp(fm.phase == 1 ? "hg" : "isotropic"); if (fm.phase == 1) pp.setFloat("g", fm.g);
        ref<PhaseFunction> ph = static_cast<PhaseFunction *>(create(MTS_CLASS(PhaseFunction), pp)); ph->configure();
        med->addChild(ph); ph->setParent(med); med->configure();
        media.push_back(med);
    }
    auto attachMedia = [&](Shape *shape, size_t index) {
        if (fs.shapeMedia.empty()) return;
        int32_t in = fs.shapeMedia[index * 2], ex = fs.shapeMedia[index * 2 + 1];
        if (in >= 0) shape->addChild("interior", media[in]);
        if (ex >= 0) shape->addChild("exterior", media[ex]);
    };
    // shapes
    std::vector<ref<Shape> > groups, instancesKeep;
    for (uint32_t si = 0; si < fs.nShapes; ++si) {
        const FShape &sh = fs.shapes[si];
        bool useN = fs.hasN && !(sh.faceNormals & 1); const bool useUV = fs.hasUV && (sh.faceNormals & 2);
        ref<TriMesh> mesh = new TriMesh("shape" + std::to_string(si), sh.triCount, sh.vertCount, useN, useUV, false, false,
                                        (sh.faceNormals & 1) != 0);
        for (uint32_t v = 0; v < sh.vertCount; ++v) {
            const float *p = &fs.pos[(sh.firstVert + v) * 3];
            mesh->getVertexPositions()[v] = Point(p[0], p[1], p[2]);
            if (useN) { const float *n = &fs.nrm[(sh.firstVert + v) * 3]; mesh->getVertexNormals()[v] = Normal(n[0], n[1], n[2]); }
            if (useUV) { const float *t = &fs.uv[(sh.firstVert + v) * 2]; mesh->getVertexTexcoords()[v] = Point2(t[0], t[1]); }
        }
        for (uint32_t t = 0; t < sh.triCount; ++t)
            for (int k = 0; k < 3; ++k)
                mesh->getTriangles()[t].idx[k] = fs.idx[(sh.firstTri + t) * 3 + k] - sh.firstVert;
        mesh->addChild(bsdfs[sh.bsdf]); bsdfs[sh.bsdf]->setParent(mesh);
        attachMedia(mesh, si);
        if (sh.emitter >= 0) {
            const FEmitter &fe = fs.emitters[sh.emitter];
            Properties p("area"); p.setSpectrum("radiance", rgb(fe.radiance)); p.setFloat("samplingWeight", fe.weight);
            ref<Emitter> em = static_cast<Emitter *>(create(MTS_CLASS(Emitter), p));
            mesh->addChild(em); em->setParent(mesh); em->configure();
        }
        mesh->configure();
        if (sh.pad) {                      // member of shape group sh.pad - 1 (FShape::pad carries the group id)
            if (groups.size() < sh.pad) groups.resize(sh.pad);
            if (!groups[sh.pad - 1]) groups[sh.pad - 1] = static_cast<Shape *>(create(MTS_CLASS(Shape), Properties("shapegroup")));
            groups[sh.pad - 1]->addChild(mesh); mesh->setParent(groups[sh.pad - 1]);
            continue;
        }
        b.scene->addChild(mesh); mesh->setParent(b.scene);
    }
    for (ref<Shape> &g : groups) g->configure();               // builds the group's kd-tree (shapegroup.cpp:50-58)
    for (const FInstance &fi : fs.instances) {
        Properties p("instance");
        Matrix4x4 m; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m(i, j) = fi.toWorld[i * 4 + j];
        p.setTransform("toWorld", Transform(m));
        ref<Shape> inst = static_cast<Shape *>(create(MTS_CLASS(Shape), p));
        inst->addChild(groups[fi.group]); inst->configure();
        instancesKeep.push_back(inst);
    }
    // analytic shapes (after the meshes: shape index = nShapes + i).  The descriptor holds the post-constructor objectToWorld; sphere and
    // cylinder get their scale back through `toWorld` so that the reference's constructors split it off again (sphere.cpp:113-123, cylinder.cpp:85-106)
    for (size_t ai = 0; ai < fs.analytic.size(); ++ai) {
        const FAnalytic &a = fs.analytic[ai];
        static const char *names[] = {"rectangle", "disk", "sphere", "cylinder"};
        Properties p(names[a.type]);
        Matrix4x4 m; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m(i, j) = a.toWorld[i * 4 + j];
        Transform tw(m);
        if (a.type == 2) tw = tw * Transform::scale(Vector(a.radius));
        if (a.type == 3) tw = tw * Transform::scale(Vector(a.radius, a.radius, a.length));
        p.setTransform("toWorld", tw);
        if (a.type >= 2) p.setBoolean("flipNormals", (a.flags & 1) != 0);
        ref<Shape> shape = static_cast<Shape *>(create(MTS_CLASS(Shape), p));
        shape->addChild(bsdfs[a.bsdf]); bsdfs[a.bsdf]->setParent(shape);
        attachMedia(shape, fs.nShapes + ai);
        if (a.emitter >= 0) {
            const FEmitter &fe = fs.emitters[a.emitter];
            Properties ep("area"); ep.setSpectrum("radiance", rgb(fe.radiance)); ep.setFloat("samplingWeight", fe.weight);
            ref<Emitter> em = static_cast<Emitter *>(create(MTS_CLASS(Emitter), ep));
            shape->addChild(em); em->setParent(shape); em->configure();
        }
        shape->configure();
        b.scene->addChild(shape); shape->setParent(b.scene);
    }
    for (ref<Shape> &inst : instancesKeep) { b.scene->addChild(inst); inst->setParent(b.scene); }
    // film + filter
    {
        static const char *fnames[] = {"box", "gaussian", "tent", "mitchell", "catmullrom", "lanczos"};
        Properties fp(fnames[fs.filter]);
        if (fs.filter == 1) fp.setFloat("stddev", fs.filterStddev);
        if (fs.filter == 3) { fp.setFloat("B", fs.filterRadius); fp.setFloat("C", fs.filterStddev); }
        if (fs.filter == 5) fp.setInteger("lobes", (int) fs.filterRadius);
        b.filter = static_cast<ReconstructionFilter *>(create(MTS_CLASS(ReconstructionFilter), fp));
        b.filter->configure();
        Properties p("hdrfilm"); p.setInteger("width", fs.W); p.setInteger("height", fs.H); p.setBoolean("banner", false);
        if (fs.crop[0]) { p.setInteger("width", fs.crop[0]); p.setInteger("height", fs.crop[1]); p.setInteger("cropOffsetX", fs.crop[2]); p.setInteger("cropOffsetY", fs.crop[3]);
                          p.setInteger("cropWidth", fs.W); p.setInteger("cropHeight", fs.H); }
        b.film = static_cast<Film *>(create(MTS_CLASS(Film), p));
        b.film->addChild(b.filter); b.filter->setParent(b.film);
        b.film->configure();
    }
    // sampler
    if (fs.sampler == 1) {
        Properties p("sobol"); p.setSize("sampleCount", fs.spp); p.setSize("scramble", (size_t) fs.seed);
        b.sampler = static_cast<Sampler *>(create(MTS_CLASS(Sampler), p));
        b.sampler->configure();
    } else {
        b.sampler = new SeededIndependent(fs.spp, (uint32_t) fs.seed, fs.W);
    }
    // sensor
    {
        Properties p("perspective");
        Matrix4x4 m; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m(i, j) = fs.camToWorld[i * 4 + j];
        p.setTransform("toWorld", Transform(m));
        p.setFloat("fov", fs.fov); p.setString("fovAxis", "x");
        p.setFloat("nearClip", fs.nearClip); p.setFloat("farClip", fs.farClip);
        b.sensor = static_cast<Sensor *>(create(MTS_CLASS(Sensor), p));
        b.sensor->addChild(b.film); b.film->setParent(b.sensor);
        b.sensor->addChild(b.sampler); b.sampler->setParent(b.sensor);
        if (fs.sensorMedium >= 0) b.sensor->addChild(media[fs.sensorMedium]);
        b.sensor->configure();
        b.scene->addChild(b.sensor); b.sensor->setParent(b.scene);
    }
    // integrator
    {
        Properties p(fs.integrator == 2 ? "volpath" : fs.integrator == 1 ? "volpath_simple" : "path"); p.setInteger("maxDepth", fs.maxDepth); p.setInteger("rrDepth", fs.rrDepth);
        p.setBoolean("strictNormals", fs.strictNormals != 0); p.setBoolean("hideEmitters", fs.hideEmitters != 0);
        b.integrator = static_cast<SamplingIntegrator *>(create(MTS_CLASS(Integrator), p));
        b.integrator->configure();
        b.scene->addChild(b.integrator); b.integrator->setParent(b.scene);
    }
    b.scene->configure();      // picks sensor/sampler, integrator->configureSampler (Sobol: setFilmResolution)
    b.scene->initialize();     // kd-tree build + emitter PDF
    return b;
}

// ------------------------------------------------------------------------------------------ MIP pyramid of a bitmap texture
// BitmapTexture builds TMIPMap<Color3, Color3h> with a 2-lobed Lanczos filter (src/textures/bitmap.cpp:193-214).  This mode builds the very same
// pyramid from raw RGB floats and dumps every level (half-precision texels widened to float): texture input data for the build's scenes.
#include <mitsuba/render/mipmap.h>
static void modeMipmap(const std::string &in, int w, int h, int bcu, int bcv, const std::string &out, Float maxValue = 1.0f) {
    typedef TSpectrum<Float, 3> Color3; typedef TSpectrum<half, 3> Color3h; typedef TMIPMap<Color3, Color3h> MIPMap3;
    ref<Bitmap> bmp = new Bitmap(Bitmap::ERGB, Bitmap::EFloat32, Vector2i(w, h));
    FILE *f = fopen(in.c_str(), "rb"); if (!f || fread(bmp->getFloat32Data(), 4, (size_t) w * h * 3, f) != (size_t) w * h * 3) { fprintf(stderr, "cannot read %s\n", in.c_str()); _exit(2); } fclose(f);
    Properties rp("lanczos"); rp.setInteger("lobes", 2);
    ref<ReconstructionFilter> rf = static_cast<ReconstructionFilter *>(create(MTS_CLASS(ReconstructionFilter), rp)); rf->configure();
    ref<MIPMap3> mip = new MIPMap3(bmp, Bitmap::ERGB, Bitmap::EFloat, rf, (ReconstructionFilter::EBoundaryCondition) bcu, (ReconstructionFilter::EBoundaryCondition) bcv, EEWA, 20.0f, fs::pathstr(), 0, maxValue);
    std::vector<float> texels; std::vector<int32_t> sizes;
    for (int l = 0; l < mip->getLevels(); ++l) {
        ref<Bitmap> lb = mip->toBitmap(l); const half *hp = lb->getFloat16Data(); const size_t n = (size_t) lb->getWidth() * lb->getHeight() * 3;
        sizes.push_back(lb->getWidth()); sizes.push_back(lb->getHeight());
        for (size_t i = 0; i < n; ++i) texels.push_back((float) hp[i]);
    }
    save(out + "_texels.npy", "<f4", {texels.size()}, texels);
    save(out + "_sizes.npy", "<i4", {sizes.size() / 2, 2}, sizes);
}

// ------------------------------------------------------------------------------------------ modes
static void modeTables(const std::string &out) {
    const uint32_t dims = 128;
    std::vector<uint32_t> m32(sobol::Matrices::matrices32, sobol::Matrices::matrices32 + dims * 52);
    save(out + "/sobol_matrices32.npy", "<u4", {dims, 52}, m32);
    std::vector<uint64_t> vdc, vdci;
    for (int m = 1; m <= 16; ++m) for (int c = 0; c < 52; ++c) {
        vdc.push_back(sobol::Matrices::vdc_sobol_matrices[m - 1][c]);
        vdci.push_back(sobol::Matrices::vdc_sobol_matrices_inv[m - 1][c]);
    }
    save(out + "/sobol_vdc.npy", "<u8", {16, 52}, vdc);
    save(out + "/sobol_vdc_inv.npy", "<u8", {16, 52}, vdci);
    // SFMT known answers (reference src/tests/test_random.cpp:433-508 uses Random(4321))
    ref<Random> rng = new Random((uint64_t) 4321);
    std::vector<uint64_t> kat; for (int i = 0; i < 1000; ++i) kat.push_back(rng->nextULong());
    save(out + "/sfmt_kat_4321.npy", "<u8", {1000}, kat);
    ref<Random> rng2 = new Random((uint64_t) 1234);
    std::vector<float> fl; for (int i = 0; i < 256; ++i) fl.push_back(rng2->nextFloat());
    save(out + "/sfmt_float_1234.npy", "<f4", {256}, fl);
    std::vector<uint64_t> tea;
    for (uint32_t a = 0; a < 16; ++a) for (uint32_t c = 0; c < 16; ++c) tea.push_back(sampleTEA(a * 2654435761u, c * 40503u + a, 4));
    save(out + "/tea_4rounds.npy", "<u8", {16, 16}, tea);
    // Sobol look_up + sampleSingle known answers
    std::vector<uint64_t> lu; std::vector<float> sv;
    for (uint32_t m = 2; m <= 12; m += 5) for (uint32_t fr = 0; fr < 20; fr += 3) for (uint32_t k = 0; k < 8; ++k) {
        uint32_t px = (k * 2654435761u) & ((1u << m) - 1), py = (k * 40503u + fr * 977u) & ((1u << m) - 1);
        uint64_t idx = sobol::look_up(m, fr, px, py, 0);
        lu.push_back(m); lu.push_back(fr); lu.push_back(px); lu.push_back(py); lu.push_back(idx);
        for (uint32_t d = 0; d < 8; ++d) sv.push_back(sobol::sampleSingle(idx, d * 7, 0));
    }
    save(out + "/sobol_lookup.npy", "<u8", {lu.size() / 5, 5}, lu);
    save(out + "/sobol_values.npy", "<f4", {sv.size() / 8, 8}, sv);
    // conductor eta / k as linear RGB, exactly as RoughConductor's constructor derives them (src/bsdfs/roughconductor.cpp:177-189):
    // data/ior/<name>.{eta,k}.spd -> InterpolatedSpectrum -> Spectrum::fromContinuousSpectrum (RGB mode).  Rows: Cu, Al, Au; cols eta rgb, k rgb
    const char *names[3] = {"Cu", "Al", "Au"}; std::vector<float> ior;
    for (int i = 0; i < 3; ++i) for (int part = 0; part < 2; ++part) {
        std::string file = std::string(MI_REF_ROOT "/data/ior/") + names[i] + (part ? ".k.spd" : ".eta.spd");
        Spectrum sp; sp.fromContinuousSpectrum(InterpolatedSpectrum(fs::pathstr(file)));
        Float r, g, b; sp.toLinearRGB(r, g, b); ior.push_back(r); ior.push_back(g); ior.push_back(b);
    }
    save(out + "/conductor_ior_rgb.npy", "<f4", {3, 6}, ior);
    // diffuse Fresnel reflectances as SmoothPlastic::configure computes them (plastic.cpp:199-201): rows (eta, fdrInt = F_dr(1/eta), fdrExt = F_dr(eta))
    std::vector<float> fdr;
    for (float eta : {1.1f, 1.3f, 1.33f, 1.49f, 1.5046f, 1.7f, 2.0f, 2.419f}) { fdr.push_back(eta); fdr.push_back(fresnelDiffuseReflectance(1 / eta, false)); fdr.push_back(fresnelDiffuseReflectance(eta, false)); }
    save(out + "/fresnel_diffuse_reflectance.npy", "<f4", {fdr.size() / 3, 3}, fdr);
    // RoughPlastic::configure (roughplastic.cpp:281-299): the external rough transmittance reduced to a 1-D slice over the incidence angle
    // (setEta(eta), setAlpha(alpha)) and the internal diffuse transmittance (setEta(1/eta), evalDiffuse(alpha)); rows: distr, eta, alpha, Tdiff_int, T[100]
    std::vector<float> rt;
    for (int distr = 0; distr < 3; ++distr) for (float eta : {1.49f, 1.5046f, 1.9f}) for (float alpha : {0.05f, 0.1f, 0.3f}) {
        ref<RTAccess> ext = new RTAccess(distr == 0 ? MicrofacetDistribution::EBeckmann : distr == 1 ? MicrofacetDistribution::EGGX : MicrofacetDistribution::EPhong);
        ext->checkEta(eta); ext->checkAlpha(alpha);
        ref<RoughTransmittance> internal = ext->clone();
        ext->setEta(eta); internal->setEta(1 / eta); ext->setAlpha(alpha);
        rt.push_back((float) distr); rt.push_back(eta); rt.push_back(alpha); rt.push_back(internal->evalDiffuse(alpha));
        if (ext->thetaSamples() != 100) { fprintf(stderr, "unexpected table size\n"); _exit(2); }
        for (size_t i = 0; i < 100; ++i) rt.push_back(ext->trans()[i]);
    }
    save(out + "/rough_transmittance_slices.npy", "<f4", {rt.size() / 104, 104}, rt);
    {   // Spectrum::fromContinuousSpectrum (src/libcore/spectrum.cpp:172-185) applied to hat functions of half-width H nm centred every H nm over
        // 360..830 nm: the linear map from a piecewise-linear spectrum (knots on that grid) to linear RGB that <spectrum value="wavelength:value, ...">
        // goes through in an RGB build.  (Narrower hats are missed by the adaptive quadrature of ContinuousSpectrum::average, spectrum.cpp:546-568.)
        const int H = getenv("MI_HAT") ? atoi(getenv("MI_HAT")) : 5;
        std::vector<float> hat; size_t n = 0;
        for (int c = 360; c <= 830; c += H, ++n) {
            InterpolatedSpectrum sp(3);
            sp.append((Float) (c - H), 0.0f); sp.append((Float) c, 1.0f); sp.append((Float) (c + H), 0.0f);
            Spectrum rgbv; rgbv.fromContinuousSpectrum(sp);
            Float r, g, b; rgbv.toLinearRGB(r, g, b);
            hat.push_back(r); hat.push_back(g); hat.push_back(b);
        }
        save(out + "/spectrum_hat_response.npy", "<f4", {n, 3}, hat);
    }
}

static uint64_t g_rays = 0, g_shadow = 0;

// per-(pixel, sampleIndex) Li; `pairs` = [px, py, sampleIdx]*n
static void modeSamples(Built &b, const FScene &fs, const std::string &pairsPath, const std::string &out) {
    FILE *f = fopen(pairsPath.c_str(), "rb"); if (!f) { fprintf(stderr, "no pairs\n"); _exit(2); }
    fseek(f, 0, SEEK_END); size_t n = ftell(f) / 12; fseek(f, 0, SEEK_SET);
    std::vector<uint32_t> pairs(n * 3); rd(f, pairs.data(), n * 12); fclose(f);
    ref<RecordingSampler> rec = new RecordingSampler(b.sampler->clone());
    b.integrator->configureSampler(b.scene, rec);
    std::vector<float> li(n * 3), pos(n * 2), ray(n * 8), logv(n * 64, -1.0f); std::vector<int32_t> depth(n), nlog(n);
    RadianceQueryRecord rRec(b.scene, rec);
    for (size_t i = 0; i < n; ++i) {
        Point2i px(pairs[i * 3], pairs[i * 3 + 1]);
        rec->generate(px, pairs[i * 3 + 2]);
        rRec.newQuery(RadianceQueryRecord::ESensorRay & ~RadianceQueryRecord::EOpacity, b.sensor->getMedium());
        Point2 samplePos(Point2(px) + Vector2(rRec.nextSample2D()));
        RayDifferential r;
        Spectrum spec = b.sensor->sampleRayDifferential(r, samplePos, Point2(0.5f), 0.5f);
        r.scaleDifferential(1.0f / std::sqrt((Float) fs.spp));
        spec *= b.integrator->Li(r, rRec);
        Float R, G, B; spec.toLinearRGB(R, G, B);
        li[i * 3] = R; li[i * 3 + 1] = G; li[i * 3 + 2] = B;
        pos[i * 2] = samplePos.x; pos[i * 2 + 1] = samplePos.y;
        ray[i * 8 + 0] = r.o.x; ray[i * 8 + 1] = r.o.y; ray[i * 8 + 2] = r.o.z; ray[i * 8 + 3] = r.mint;
        ray[i * 8 + 4] = r.d.x; ray[i * 8 + 5] = r.d.y; ray[i * 8 + 6] = r.d.z; ray[i * 8 + 7] = r.maxt;
        depth[i] = rRec.depth;
        nlog[i] = (int32_t) rec->log.size();
        for (size_t k = 0; k < rec->log.size() && k < 64; ++k) logv[i * 64 + k] = rec->log[k];
    }
    save(out + "_li.npy", "<f4", {n, 3}, li);
    save(out + "_pos.npy", "<f4", {n, 2}, pos);
    save(out + "_ray.npy", "<f4", {n, 8}, ray);
    save(out + "_depth.npy", "<i4", {n}, depth);
    save(out + "_nsamples.npy", "<i4", {n}, nlog);
    save(out + "_svalues.npy", "<f4", {n, 64}, logv);
}

// camera-ray hit records
static void modeHits(Built &b, const FScene &fs, uint32_t step, const std::string &out) {
    std::vector<float> rec; size_t n = 0;
    for (uint32_t y = step / 2; y < fs.H; y += step) for (uint32_t x = step / 2; x < fs.W; x += step) {
        RayDifferential r; b.sensor->sampleRayDifferential(r, Point2(x + 0.25f, y + 0.75f), Point2(0.5f), 0.5f);
        Intersection its; bool hit = b.scene->rayIntersect(r, its);
        float v[24] = {0}; v[0] = (float) x; v[1] = (float) y; v[2] = hit ? 1.f : 0.f;
        if (hit) {
            v[3] = its.t; v[4] = its.p.x; v[5] = its.p.y; v[6] = its.p.z;
            v[7] = its.geoFrame.n.x; v[8] = its.geoFrame.n.y; v[9] = its.geoFrame.n.z;
            v[10] = its.shFrame.n.x; v[11] = its.shFrame.n.y; v[12] = its.shFrame.n.z;
            v[13] = its.shFrame.s.x; v[14] = its.shFrame.s.y; v[15] = its.shFrame.s.z;
            v[16] = its.uv.x; v[17] = its.uv.y; v[18] = its.wi.x; v[19] = its.wi.y; v[20] = its.wi.z;
            v[21] = (float) its.primIndex;
            // global shape index = position in scene->getShapes()
            const ref_vector<Shape> &shapes = b.scene->getShapes();
            for (size_t s = 0; s < shapes.size(); ++s) if (shapes[s].get() == its.shape) v[22] = (float) s;
        }
        rec.insert(rec.end(), v, v + 24); ++n;
    }
    save(out, "<f4", {n, 24}, rec);
}

static void modeImage(Built &b, const FScene &fs, int threads, const std::string &out) {
    // SamplingIntegrator::renderBlock over 32x32 blocks, `threads` workers each with its own sampler clone and ImageBlock;
    // results summed into one film-sized 5-channel block (what BlockedRenderProcess::processResult -> Film::put does).
    const int bs = 32; int bx = (fs.W + bs - 1) / bs, by = (fs.H + bs - 1) / bs;
    int border = b.filter->getBorderSize();
    size_t FW = fs.W + 2 * border, FH = fs.H + 2 * border;
    std::vector<float> film(FW * FH * 5, 0.0f);
    std::atomic<int> next(0); std::mutex mtx;
    ref<Timer> timer = new Timer();
    auto worker = [&](int tid) {
        ref<Sampler> sampler = b.sampler->clone();
        b.integrator->configureSampler(b.scene, sampler);
        ref<ImageBlock> block = new ImageBlock(Bitmap::ESpectrumAlphaWeight, Vector2i(bs, bs), b.filter);
        bool stop = false;
        while (true) {
            int id = next++; if (id >= bx * by) break;
            int ox = (id % bx) * bs, oy = (id / bx) * bs;
            int w = std::min(bs, (int) fs.W - ox), h = std::min(bs, (int) fs.H - oy);
            if (w != bs || h != bs) block = new ImageBlock(Bitmap::ESpectrumAlphaWeight, Vector2i(w, h), b.filter);
            block->setOffset(Point2i(ox, oy));
            std::vector<TPoint2<uint8_t> > pts;
            for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) pts.push_back(TPoint2<uint8_t>(x, y));
            b.integrator->renderBlock(b.scene, b.sensor, sampler, block, stop, pts);
            const float *src = block->getBitmap()->getFloat32Data();
            int sw = w + 2 * border, shh = h + 2 * border;
            std::lock_guard<std::mutex> g(mtx);
            for (int y = 0; y < shh; ++y) for (int x = 0; x < sw; ++x)
                for (int c = 0; c < 5; ++c)
                    film[((size_t) (oy + y) * FW + (ox + x)) * 5 + c] += src[((size_t) y * sw + x) * 5 + c];
            if (w != bs || h != bs) block = new ImageBlock(Bitmap::ESpectrumAlphaWeight, Vector2i(bs, bs), b.filter);
        }
    };
    std::vector<std::thread> pool;
    // mitsuba::Thread registration is needed for TLS-based statistics; run worker 0 inline, others as mitsuba threads is overkill:
    // StatsCounter uses Thread::getID() modulo slots, which works for foreign threads after Thread::registerUnmanagedThread.
    for (int t = 1; t < threads; ++t) pool.emplace_back([&, t] { Thread::registerUnmanagedThread("wrk" + std::to_string(t)); worker(t); });
    worker(0);
    for (auto &t : pool) t.join();
    double sec = timer->getMicroseconds() * 1e-6;
    save(out + "_film.npy", "<f4", {FH, FW, 5}, film);
    double nsamples = (double) fs.W * fs.H * fs.spp;
    std::string stats = Statistics::getInstance()->getStats();
    FILE *f = fopen((out + "_stats.txt").c_str(), "w");
    fprintf(f, "seconds %.6f\nsamples %.0f\nmsamples_per_s %.6f\nthreads %d\n%s\n", sec, nsamples, nsamples / sec * 1e-6, threads, stats.c_str());
    fclose(f);
    printf("render: %.3f s, %.4f Msamples/s (%d threads)\n", sec, nsamples / sec * 1e-6, threads);
}

static void modeCamera(Built &b, const FScene &fs, const std::string &out) {
    std::vector<float> v;
    // sampleToCamera is protected: recover it by probing m_sampleToCamera through sampleRayDifferential is lossy, so dump
    // rays on a grid instead and the filter table; the host-side camera matrix is checked through these rays.
    for (int j = 0; j <= 8; ++j) for (int i = 0; i <= 8; ++i) {
        Point2 s(fs.W * (i / 8.0f), fs.H * (j / 8.0f));
        RayDifferential r; b.sensor->sampleRayDifferential(r, s, Point2(0.5f), 0.5f);
        float a[10] = {s.x, s.y, r.o.x, r.o.y, r.o.z, r.mint, r.d.x, r.d.y, r.d.z, r.maxt};
        v.insert(v.end(), a, a + 10);
    }
    save(out + "_camrays.npy", "<f4", {81, 10}, v);
    std::vector<float> ft;
    for (int i = 0; i <= 320; ++i) ft.push_back(b.filter->evalDiscretized(-b.filter->getRadius() * 1.05f + i * (2.1f * b.filter->getRadius() / 320)));
    ft.push_back(b.filter->getRadius()); ft.push_back((float) b.filter->getBorderSize());
    save(out + "_filter.npy", "<f4", {ft.size()}, ft);
}

// unit vectors: warp functions, TriAccel::load, diffuse BSDF, emitter sampling
static void modeUnits(Built &b, const FScene &fs, const std::string &out) {
    std::vector<float> w;
    for (int j = 0; j < 33; ++j) for (int i = 0; i < 33; ++i) {
        Point2 s(std::min(i / 32.0f, 0.99999994f), std::min(j / 32.0f, 0.99999994f));
        Vector h = warp::squareToCosineHemisphere(s); Point2 t = warp::squareToUniformTriangle(s); Point2 d = warp::squareToUniformDiskConcentric(s);
        float a[9] = {s.x, s.y, h.x, h.y, h.z, t.x, t.y, d.x, d.y}; w.insert(w.end(), a, a + 9);
    }
    save(out + "_warp.npy", "<f4", {33 * 33, 9}, w);
    std::vector<float> ta;
    for (uint32_t t = 0; t < fs.nTris; ++t) {
        const float *p0 = &fs.pos[fs.idx[t * 3] * 3], *p1 = &fs.pos[fs.idx[t * 3 + 1] * 3], *p2 = &fs.pos[fs.idx[t * 3 + 2] * 3];
        TriAccel acc; acc.load(Point(p0[0], p0[1], p0[2]), Point(p1[0], p1[1], p1[2]), Point(p2[0], p2[1], p2[2]));
        float a[10] = {(float) acc.k, acc.n_u, acc.n_v, acc.n_d, acc.a_u, acc.a_v, acc.b_nu, acc.b_nv, acc.c_nu, acc.c_nv};
        ta.insert(ta.end(), a, a + 10);
    }
    save(out + "_triaccel.npy", "<f4", {fs.nTris, 10}, ta);
    // emitter sampling from reference points = camera-ray hits
    std::vector<float> em;
    for (uint32_t y = 40; y < fs.H; y += fs.H / 6) for (uint32_t x = 40; x < fs.W; x += fs.W / 6) {
        RayDifferential r; b.sensor->sampleRayDifferential(r, Point2(x + 0.5f, y + 0.5f), Point2(0.5f), 0.5f);
        Intersection its; if (!b.scene->rayIntersect(r, its)) continue;
        for (int k = 0; k < 4; ++k) {
            Point2 s(0.13f + 0.23f * k, 0.91f - 0.27f * k);
            DirectSamplingRecord dRec(its);
            Spectrum val = b.scene->sampleEmitterDirect(dRec, s, true);
            Float R, G, B; val.toLinearRGB(R, G, B);
            float a[20] = {its.p.x, its.p.y, its.p.z, its.shFrame.n.x, its.shFrame.n.y, its.shFrame.n.z, s.x, s.y, R, G, B,
                           dRec.p.x, dRec.p.y, dRec.p.z, dRec.d.x, dRec.d.y, dRec.d.z, dRec.dist, dRec.pdf, 0};
            if (!val.isZero()) a[19] = b.scene->pdfEmitterDirect(dRec);
            em.insert(em.end(), a, a + 20);
        }
    }
    save(out + "_emitter.npy", "<f4", {em.size() / 20, 20}, em);
    // BSDF eval/pdf/sample for every BSDF at a hit (local frame), wi/u grids
    std::vector<float> bs; ref<Sampler> constSampler = new ConstSampler();
    const ref_vector<Shape> &shapes = b.scene->getShapes();
    for (size_t si = 0; si < shapes.size(); ++si) {
        const BSDF *bsdf = const_cast<Shape *>(shapes[si].get())->getBSDF();
        if (!bsdf) continue;      // instances carry no BSDF of their own
        Intersection its; its.shape = const_cast<Shape *>(shapes[si].get()); its.p = Point(0.0f); its.uv = Point2(0.5f); its.hasUVPartials = false; its.time = 0;
        its.shFrame = Frame(Normal(0, 0, 1)); its.geoFrame = its.shFrame;
        for (int a = 0; a < 5; ++a) for (int k = 0; k < 9; ++k) {
            float th = 0.1f + 0.33f * a, ph = 0.7f * a;
            Vector wi(std::sin(th) * std::cos(ph), std::sin(th) * std::sin(ph), std::cos(th));
            if (a == 4) wi.z = -wi.z;   // back side
            Point2 u(0.07f + 0.11f * k, 0.93f - 0.1f * k);
            its.wi = wi;
            BSDFSamplingRecord bRec(its, constSampler.get(), ERadiance);
            Float pdf = 0; Spectrum wgt = bsdf->sample(bRec, pdf, u);
            Float R, G, B; wgt.toLinearRGB(R, G, B);
            float rowv[24] = {(float) si, wi.x, wi.y, wi.z, u.x, u.y, R, G, B, pdf, bRec.wo.x, bRec.wo.y, bRec.wo.z, bRec.eta, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            // eval/pdf for a fixed outgoing direction
            Vector wo(std::sin(0.5f + 0.1f * k) * std::cos(2.1f), std::sin(0.5f + 0.1f * k) * std::sin(2.1f), std::cos(0.5f + 0.1f * k));
            BSDFSamplingRecord eRec(its, wi, wo, ERadiance);
            Spectrum ev = bsdf->eval(eRec); ev.toLinearRGB(R, G, B);
            rowv[14] = wo.x; rowv[15] = wo.y; rowv[16] = wo.z; rowv[17] = R; rowv[18] = G; rowv[19] = B; rowv[20] = bsdf->pdf(eRec);
            rowv[21] = (float) bsdf->getType();
            bs.insert(bs.end(), rowv, rowv + 24);
        }
    }
    save(out + "_bsdf.npy", "<f4", {bs.size() / 24, 24}, bs);
}

// Drop-in check: drive ANY integrator plugin (the reference's own `path`, or the MI355X adapter `path_hip`) through the responsive
// interface exactly as src/mitsuba/im_render.cpp:103-222 does (preprocess -> allocate -> render(..., controls, threadIdx, threadCount)),
// one thread, and dump the target ImageBlock (un-normalised RGBA sums) + the return code + the spp values progress() was called with.
#include <mitsuba/render/integrator2.h>
struct CountingInterrupt : ResponsiveIntegrator::Interrupt {
    std::vector<double> calls; int stopAfter = -1;
    int progress(ResponsiveIntegrator *, const Scene &, const Sensor &, Sampler &, ImageBlock &, double spp, ResponsiveIntegrator::Controls, int, int) override {
        calls.push_back(spp); return (stopAfter >= 0 && (int) calls.size() > stopAfter) ? 101 : 0;
    }
};
static void modeResponsive(Built &b, const FScene &fs, const std::string &plugin, int stopAfter, const std::string &out) {
    Properties p(plugin); p.setInteger("maxDepth", fs.maxDepth); p.setInteger("rrDepth", fs.rrDepth);
    p.setBoolean("strictNormals", fs.strictNormals != 0); p.setBoolean("hideEmitters", fs.hideEmitters != 0);
    if (fs.sampler == 0) p.setSize("seed", (size_t) fs.seed);
    if (plugin == "path_hip") { p.setInteger("planesPerBatch", 4); if (fs.integrator) p.setString("integrator", fs.integrator == 2 ? "volpath" : "volpath_simple"); }   // several progress() calls on small films
    ref<Integrator> integ = static_cast<Integrator *>(create(MTS_CLASS(Integrator), p));
    integ->configure();
    ref<ResponsiveIntegrator> resp = integ->makeResponsiveIntegrator();
    if (!resp) { fprintf(stderr, "plugin %s has no responsive face\n", plugin.c_str()); _exit(3); }
    ref<Sampler> sampler = b.sampler->clone();
    integ->configureSampler(b.scene, sampler);
    ref<ImageBlock> target = new ImageBlock(Bitmap::ESpectrumAlpha, b.film->getCropSize(), b.filter); target->clear();
    Sampler *sp = sampler.get(); ImageBlock *tp = target.get();
    ref<Timer> timer = new Timer();
    resp->preprocess(b.scene, b.sensor, sampler);
    resp->allocate(*b.scene, &sp, &tp, 1);
    CountingInterrupt in; in.stopAfter = stopAfter; int continu = 1, abortFlag = 0;
    ResponsiveIntegrator::Controls c{&continu, &abortFlag, &in};
    int rc = resp->render(*b.scene, *b.sensor, *sampler, *target, c, 0, 1);
    double sec = timer->getMicroseconds() * 1e-6;
    const Bitmap *bmp = target->getBitmap();
    std::vector<float> data(bmp->getFloatData(), bmp->getFloatData() + (size_t) bmp->getSize().x * bmp->getSize().y * bmp->getChannelCount());
    save(out + "_target.npy", "<f4", {(size_t) bmp->getSize().y, (size_t) bmp->getSize().x, (size_t) bmp->getChannelCount()}, data);
    std::vector<double> meta; meta.push_back(rc); meta.push_back(sec); meta.push_back((double) in.calls.size()); for (double v : in.calls) meta.push_back(v);
    save(out + "_meta.npy", "<f8", {meta.size()}, meta);
    const char *st = resp->getRealtimeStatistics();
    printf("responsive[%s]: rc %d, %.3f s, %zu progress calls%s%s\n", plugin.c_str(), rc, sec, in.calls.size(), st ? ", " : "", st ? st : "");
}

// Drop-in check of the CLASSIC face: Scene::render (src/librender/scene.cpp:475-479) calls Integrator::render(scene, queue, job, ...) from the RenderJob thread
// and the integrator delivers its result with Film::put(const ImageBlock *).  CaptureFilm is a Film that keeps what it is given (HDRFilm's storage is private and
// its develop() needs the format converter this build lacks): the mode swaps it into the sensor, runs <plugin>::render and dumps the accumulated block.
class CaptureFilm : public Film {
public:
    CaptureFilm(const Properties &p) : Film(p) { }
    void configure() { Film::configure(); m_block = new ImageBlock(Bitmap::ESpectrumAlphaWeight, m_cropSize); m_block->setOffset(m_cropOffset); m_block->clear(); }
    void clear() { m_block->clear(); }
    void put(const ImageBlock *block) { m_block->put(block); ++puts; }
    void setBitmap(const Bitmap *, Float) { }
    void addBitmap(const Bitmap *, Float) { }
    void setDestinationFile(const fs::pathstr &, uint32_t) { }
    void develop(const Scene *, Float) { }
    bool develop(const Point2i &, const Vector2i &, const Point2i &, Bitmap *) const { return false; }
    bool destinationExists(const fs::pathstr &) const { return false; }
    bool hasAlpha() const { return true; }
    std::string toString() const { return "CaptureFilm[]"; }
    ref<ImageBlock> m_block; int puts = 0;
    MTS_DECLARE_CLASS()
};
MTS_IMPLEMENT_CLASS(CaptureFilm, false, Film)
#include <mitsuba/render/renderjob.h>
#include <mitsuba/render/renderqueue.h>
#include <mitsuba/core/sched.h>
static void modeClassic(Built &b, const FScene &fs, const std::string &plugin, int threads, const std::string &out) {
    Properties fp("capture"); fp.setInteger("width", fs.W); fp.setInteger("height", fs.H);
    if (fs.crop[0]) { fp.setInteger("width", fs.crop[0]); fp.setInteger("height", fs.crop[1]); fp.setInteger("cropOffsetX", fs.crop[2]); fp.setInteger("cropOffsetY", fs.crop[3]);
                      fp.setInteger("cropWidth", fs.W); fp.setInteger("cropHeight", fs.H); }
    ref<CaptureFilm> film = new CaptureFilm(fp);
    film->addChild(b.filter); film->configure();
    b.sensor->addChild(film); film->setParent(b.sensor); b.sensor->configure();
    Properties p(plugin); p.setInteger("maxDepth", fs.maxDepth); p.setInteger("rrDepth", fs.rrDepth);
    p.setBoolean("strictNormals", fs.strictNormals != 0); p.setBoolean("hideEmitters", fs.hideEmitters != 0);
    if (fs.sampler == 0) p.setSize("seed", (size_t) fs.seed);
    if (plugin == "path_hip" && fs.integrator) p.setString("integrator", fs.integrator == 2 ? "volpath" : "volpath_simple");
    ref<Integrator> integ = static_cast<Integrator *>(create(MTS_CLASS(Integrator), p));
    integ->configure();
    b.scene->setIntegrator(integ);
    // the reference's own driver: RenderJob::run (src/librender/renderjob.cpp:66-120) registers the scene / sensor / sampler with the scheduler, then
    // Scene::preprocess -> Scene::render -> Integrator::render -> Scene::postprocess; `path` renders through the scheduler's local workers, path_hip by itself
    Scheduler *sched = Scheduler::getInstance();
    for (int i = 0; i < threads; ++i) sched->registerWorker(new LocalWorker(i, "wrk" + std::to_string(i)));
    sched->start();
    ref<RenderQueue> queue = new RenderQueue();
    ref<Timer> timer = new Timer();
    ref<RenderJob> job = new RenderJob("rend", b.scene, queue, -1, -1, -1, true, false);
    job->start();
    queue->waitLeft(0); queue->join();
    double sec = timer->getMicroseconds() * 1e-6;
    bool ok = true;
    sched->stop();
    const Bitmap *bmp = film->m_block->getBitmap();
    std::vector<float> data(bmp->getFloatData(), bmp->getFloatData() + (size_t) bmp->getSize().x * bmp->getSize().y * bmp->getChannelCount());
    save(out + "_film.npy", "<f4", {(size_t) bmp->getSize().y, (size_t) bmp->getSize().x, (size_t) bmp->getChannelCount()}, data);
    std::vector<double> meta = {ok ? 1.0 : 0.0, sec, (double) film->puts}; save(out + "_meta.npy", "<f8", {meta.size()}, meta);
    printf("classic[%s]: render() returned %d, %.3f s, %d Film::put call(s)\n", plugin.c_str(), (int) ok, sec, film->puts);
}

// mesh loaders (src/shapes/{obj,ply,serialized,cube}.cpp over TriMesh::configure): dump every mesh the plugin creates, after configure()
static void modeMesh(int argc, char **argv) {
    // mesh <plugin> <file|-> <out> <faceNormals> <flipNormals> <maxSmoothAngle|-1> <shapeIndex|-1> <flipTexCoords> [16 floats toWorld, row major]
    Properties p(argv[2]);
    if (std::string(argv[3]) != "-") p.setString("filename", argv[3]);
    p.setBoolean("faceNormals", atoi(argv[5]) != 0);
    p.setBoolean("flipNormals", atoi(argv[6]) != 0);
    if (atof(argv[7]) >= 0) p.setFloat("maxSmoothAngle", (Float) atof(argv[7]));
    if (atoi(argv[8]) >= 0) p.setInteger("shapeIndex", atoi(argv[8]));
    if (std::string(argv[2]) == "obj") { p.setBoolean("flipTexCoords", atoi(argv[9]) != 0); p.setBoolean("loadMaterials", false); }
    if (argc >= 26) {
        Matrix4x4 m; for (int i = 0; i < 16; ++i) m.m[i / 4][i % 4] = (Float) atof(argv[10 + i]);
        p.setTransform("toWorld", Transform(m));
    }
    ref<Shape> shape;
    if (std::string(argv[2]) == "serialized") {
        // the serialized PLUGIN needs pushSceneCleanupHandler from scenehandler.cpp (xerces, absent): the file format is read through the
        // reader the plugin itself calls, TriMesh::loadCompressed (src/librender/trimesh.cpp:80-87, :176-253); toWorld / flags are ignored
        ref<FileStream> in = new FileStream(fs::pathstr(argv[3]), FileStream::EReadOnly);
        in->setByteOrder(Stream::ELittleEndian);
        shape = new TriMesh(in, std::max(atoi(argv[8]), 0));
    } else
        shape = static_cast<Shape *>(create(MTS_CLASS(Shape), p));
    shape->configure();
    std::vector<ref<TriMesh> > meshes;
    if (shape->isCompound()) {
        for (int i = 0; ; ++i) { Shape *e = shape->getElement(i); if (!e) break; meshes.push_back(static_cast<TriMesh *>(e)); }
    } else meshes.push_back(static_cast<TriMesh *>(shape.get()));
    FILE *f = fopen(argv[4], "wb");
    uint32_t n = (uint32_t) meshes.size(); fwrite(&n, 4, 1, f);
    for (size_t k = 0; k < meshes.size(); ++k) {
        TriMesh *m = meshes[k];
        uint32_t hdr[4] = { (uint32_t) m->getVertexCount(), (uint32_t) m->getTriangleCount(),
                            (uint32_t) ((m->hasVertexNormals() ? 1 : 0) | (m->hasVertexTexcoords() ? 2 : 0)), (uint32_t) m->getName().size() };
        fwrite(hdr, 4, 4, f); fwrite(m->getName().data(), 1, hdr[3], f);
        fwrite(m->getVertexPositions(), 12, hdr[0], f);
        if (m->hasVertexNormals()) fwrite(m->getVertexNormals(), 12, hdr[0], f);
        if (m->hasVertexTexcoords()) fwrite(m->getVertexTexcoords(), 8, hdr[0], f);
        fwrite(m->getTriangles(), 12, hdr[1], f);
    }
    fclose(f);
    {   // the first mesh again, through the reference's own writer (TriMesh::serialize, src/librender/trimesh.cpp:1131)
        ref<FileStream> fsm = new FileStream(fs::pathstr(std::string(argv[4]) + ".serialized"), FileStream::ETruncReadWrite);
        fsm->setByteOrder(Stream::ELittleEndian);
        meshes[0]->serialize(fsm);
        fsm->close();
    }
}

// <spectrum value="wl:value, ..."> as scenehandler.cpp:680-697 converts it (InterpolatedSpectrum, zeroExtend, fromContinuousSpectrum, clampNegative)
static void modeSpectrum(int argc, char **argv) {
    InterpolatedSpectrum sp((size_t) (argc - 2) / 2);
    for (int i = 2; i + 1 < argc; i += 2) sp.append((Float) atof(argv[i]), (Float) atof(argv[i + 1]));
    sp.zeroExtend();
    Spectrum d; d.fromContinuousSpectrum(sp); d.clampNegative();
    Float r, g, b; d.toLinearRGB(r, g, b);
    printf("%.9g %.9g %.9g\n", r, g, b);
}

/// <blackbody temperature=.. /> as the scene loader converts it (scenehandler.cpp:618-631): one "r g b" line per temperature
static void modeBlackbody(int argc, char **argv) {
    for (int i = 2; i < argc; ++i) {
        BlackBodySpectrum bb((Float) atof(argv[i]));
        Spectrum d; d.fromContinuousSpectrum(bb); d.clampNegative();
        Float r, g, b; d.toLinearRGB(r, g, b);
        printf("%.9g %.9g %.9g\n", r, g, b);
    }
}

int main(int argc, char **argv) {
    Class::staticInitialization();
    Object::staticInitialization();
    PluginManager::staticInitialization();
    Statistics::staticInitialization();
    Thread::staticInitialization();
    Logger::staticInitialization();
    FileStream::staticInitialization();
    Spectrum::staticInitialization();
    // Bitmap::staticInitialization() only initialises FormatConverter (fmtconv.cpp needs boost::mpl, absent) -> skipped; nothing here converts bitmaps.
    Scheduler::staticInitialization();
    Thread::getThread()->getLogger()->setLogLevel(EWarn);
    Thread::getThread()->getFileResolver()->appendPath(fs::pathstr(MI_REF_ROOT));   // data/microfacet/*.dat, data/ior/*.spd (roughplastic, named conductors)
    if (argc >= 8 && std::string(argv[1]) == "mipmap") { modeMipmap(argv[2], atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), argv[7], argc >= 9 ? (std::string(argv[8]) == "inf" ? std::numeric_limits<Float>::infinity() : (Float) atof(argv[8])) : 1.0f); fflush(NULL); _exit(0); }
    if (argc >= 3 && std::string(argv[1]) == "blackbody") { modeBlackbody(argc, argv); fflush(NULL); _exit(0); }
    if (argc >= 6 && std::string(argv[1]) == "spectrum") { modeSpectrum(argc, argv); fflush(NULL); _exit(0); }
    if (argc >= 10 && std::string(argv[1]) == "mesh") { modeMesh(argc, argv); fflush(NULL); _exit(0); }
    if (argc < 3) { fprintf(stderr, "usage: harness tables <outdir> | spectrum <wl value>... | blackbody <kelvin>... | mesh <plugin> <file|-> <out> <faceNormals> <flipNormals> <maxSmoothAngle|-1> <shapeIndex|-1> <flipTexCoords> [toWorld x16] | mipmap <rgb.bin> <w> <h> <bcu> <bcv> <out> [maxValue|inf] | <scene> samples <pairs.bin> <out> | <scene> image <threads> <out> | <scene> hits <step> <out> | <scene> camera <out> | <scene> units <out> | <scene> responsive <plugin> <stopAfterProgressCalls|-1> <out>\n"); _exit(1); }
    std::string a1 = argv[1];
    if (a1 == "tables") { modeTables(argv[2]); fflush(stdout); _exit(0); }
    FScene fs = loadScene(argv[1]);
    Built b = buildScene(fs);
    std::string mode = argv[2];
    if (mode == "samples") modeSamples(b, fs, argv[3], argv[4]);
    else if (mode == "image") modeImage(b, fs, atoi(argv[3]), argv[4]);
    else if (mode == "hits") modeHits(b, fs, atoi(argv[3]), argv[4]);
    else if (mode == "probe") {      // debugging aid: one ray through Scene::rayIntersect -- probe ox oy oz dx dy dz
        Ray ray(Point(atof(argv[3]), atof(argv[4]), atof(argv[5])), normalize(Vector(atof(argv[6]), atof(argv[7]), atof(argv[8]))), 0.0f); Intersection its;
        bool hit = b.scene->rayIntersect(ray, its);
        printf("probe: hit %d t %g p %g %g %g shape %s\n", (int) hit, (double) its.t, (double) its.p.x, (double) its.p.y, (double) its.p.z, hit ? its.shape->getName().c_str() : "-");
    }
    else if (mode == "camera") modeCamera(b, fs, argv[3]);
    else if (mode == "units") modeUnits(b, fs, argv[3]);
    else if (mode == "responsive") modeResponsive(b, fs, argv[3], atoi(argv[4]), argv[5]);
    else if (mode == "classic") modeClassic(b, fs, argv[3], atoi(argv[4]), argv[5]);
    else { fprintf(stderr, "unknown mode\n"); _exit(1); }
    fflush(stdout);
    _exit(0);   // skip the static shutdown sequence (the process hangs in thread cleanup otherwise)
}
