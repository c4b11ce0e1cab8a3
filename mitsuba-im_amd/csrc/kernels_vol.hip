// kernels_vol.hip -- the volumetric variant of the path: SimpleVolumetricPathTracer::Li (reference src/integrators/path/volpath_simple.cpp:84-289) over homogeneous
// media (src/medium/homogeneous.cpp, src/phase/isotropic.cpp, hg.cpp) and index-matched boundaries (src/bsdfs/null.cpp), as two stages next to generate / extend / film:
//   k_shade_vol   one iteration of the loop per launch: tail of the previous one (Russian roulette :278-287), distance sampling in the current medium :114, then either
//                 the medium interaction (emitter sampling :127-138, phase-function sampling :149-162) or the surface interaction (:163-275) -- both leave at most one
//                 shadow RECORD and one continuation ray;
//   k_shadow_vol  Scene::sampleAttenuatedEmitterDirect's second half: Scene::evalTransmittance (src/librender/scene.cpp:650-713) walks from the reference point to the
//                 emitter sample through `null` boundaries and media, then `Li += throughput * (value * transmittance / emitterPdf) * (BSDF | phase) value`.
// No MIS in this integrator: emitters seen by continuation rays count only while the radiance-type bits say so (camera ray, chains of delta / null interactions).
// State word st0.w: bits 0..7 sampler dimension, 8..15 depth, 16 EEmittedRadiance, 17 the bits of ERadianceNoEmission (they change together), 18 nullChain, 19 scattered,
// 20..27 current medium + 1.  Shadow record: shO = p1 | medium + 1, maxInteractions (int16), p1OnSurface, p2OnSurface; shD = p2 | path id; shC = emitter value BEFORE the
// division by the emitter-selection probability | that probability; shT = throughput; shX = BSDF value (or the phase value in all three channels).
// Built for: meshes + analytic shapes (media on scene-level shapes), every plain BSDF incl. `null`, textures, area / point / spot / directional emitters.  Refused at
// mi_render_create: adapters (mixturebsdf / bumpmap / normalmap) that nest a `null` / `thindielectric`.  A `mask` attenuates the walks by 1 - opacity (surfaceNullEval).
#include "kernels_common.h"
#include "trace.h"

// EnvironmentMap::evalEnvironment for a ray that carries differentials (the sensor ray; envmap.cpp:398-411 -> TMIPMap::eval over the map's pyramid), as k_env_primary
DEV v3 envEvalSensorRay(const DScene &sc, const RenderConst &rc, v3 d, float2 sp) {
    if (!sc.env_texture) return envEval(sc, d);
    const TextureD tx = sc.textures[sc.env_texture - 1u];
    v3 rxd, ryd; cameraDifferentials(sc, rc.inv_sqrt_spp, sp.x, sp.y, d, rxd, ryd);
    const v3 v = mat3(sc.env_to_local, d);
    const float uvx = atan2f(v.x, -v.z) * MI_INV_TWOPI, uvy = acosf(minf(1.0f, maxf(-1.0f, v.y))) * MI_INV_PI;
    const v3 dvdx = mat3(sc.env_to_local, rxd) - v, dvdy = mat3(sc.env_to_local, ryd) - v;
    const float t1 = MI_INV_TWOPI / (v.x * v.x + v.z * v.z), t2 = -MI_INV_PI / maxf(sqrtf(maxf(0.0f, 1.0f - v.y * v.y)), MI_EPSILON);
    return mipEval(sc, tx, uvx, uvy, t1 * (dvdx.z * v.x - dvdx.x * v.z), t2 * dvdx.y, t1 * (dvdy.z * v.x - dvdy.x * v.z), t2 * dvdy.y) * sc.env_scale;
}

#define VOL_EMITTED (1u << 16)
#define VOL_OTHERS (1u << 17)
#define VOL_NULLCHAIN (1u << 18)
#define VOL_SCATTERED (1u << 19)

template <bool TEX, bool ENV, bool WRAP>      // TEX: textures bound to materials; ENV: an environment map among the emitters; WRAP: mixturebsdf / bumpmap / normalmap records present (with TEX)
__global__ __launch_bounds__(WG) void k_shade_vol(DScene sc, RenderConst rc, Queues q, int buf) {
    extern __shared__ uint32_t s_dyn[];
    uint32_t *s_nib = s_dyn;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = buf ^ 1;
    const SobolTabLds m32{(lds_u32_ptr) s_nib, rc.nib_count, rc.sobol_scramble};
    if (rc.sampler == 1) { const uint32_t nibWords = rc.nib_dims * rc.nib_count * 16u; for (uint32_t i = tid; i < nibWords; i += WG) s_nib[i] = rc.sobol_nib[i]; }
    Tabs<false> tb; tb.shade4 = (AS<false>::p4) sc.shade; tb.materials4 = (AS<false>::p4) sc.materials; tb.emitters4 = (AS<false>::p4) sc.emitters; tb.emitter_cdf = sc.emitter_cdf; tb.area_cdf = sc.area_cdf;
    unsigned long long pathLen = 0;
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t seg = blockIdx.x * (WG / 64) + wave; seg < q.n_seg; seg += gridDim.x * (WG / 64)) {     // a segment is owned by one wave (shade.h)
    const uint32_t n = q.count[buf][seg];
    const uint64_t segBase = (uint64_t) seg * q.cap, shBase = segBase * 2u;      // the shadow queue holds two records per path slot (emitter sampling + the alpha walk of a sensor ray)
    uint32_t outA = 0, outS = 0, outE = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        bool alive = false, wantShadow = false, wantAlpha = false;
        float4 shO, shD, shC, shT, shX, alO, alD, nrO, nrD, nS1; uint4 nS0;
        if (i < n) {
            const uint64_t slot = segBase + i;
            const float4 ro = q.rayO[buf][slot], rd = q.rayD[buf][slot], hr = q.hit[slot]; const uint4 s0 = q.st0[buf][slot]; const float4 s1 = q.st1[buf][slot];
            SamplerState ss; const uint32_t pid = s0.x; ss.a = s0.y; ss.b = s0.z; ss.dim = s0.w & 0xFFu;
            const int depth = (int) ((s0.w >> 8) & 0xFFu); uint32_t fl = s0.w & (VOL_EMITTED | VOL_OTHERS | VOL_NULLCHAIN | VOL_SCATTERED); int medium = (int) ((s0.w >> 20) & 0xFFu) - 1;
            const v3 o = V(ro.x, ro.y, ro.z), d = V(rd.x, rd.y, rd.z); v3 T = V(s1.x, s1.y, s1.z); float eta = s1.w;
            const uint32_t prim = __float_as_uint(hr.w); const float tHit = prim == 0xFFFFFFFFu ? INFINITY : hr.x;
            v3 add = V(0, 0, 0); bool haveAdd = false;
            do {
                if (depth > 1 && depth - 1 >= rc.rr_depth) {                 // volpath_simple.cpp:278-287 (the previous iteration's tail)
                    const float qq = minf(maxf(maxf(T.x, T.y), T.z) * eta * eta, 0.95f);
                    if (next1D(ss, rc.sampler, m32) >= qq) { pathLen += (unsigned) depth; break; }
                    const float r = 1.0f / qq; T = T * r;
                }
                if (!(depth <= rc.max_depth || rc.max_depth < 0)) { pathLen += (unsigned) depth; break; }
                const bool others = (fl & VOL_OTHERS) != 0, emitted = (fl & VOL_EMITTED) != 0, scattered = (fl & VOL_SCATTERED) != 0;
                MediumRec mRec; bool mediumEvent = false; MediumD md;
                if (medium >= 0) { md = sc.media[medium]; mediumEvent = mediumSampleDistance(md, o, d, 0.0f, tHit, ss, rc.sampler, m32, mRec); }
                if (depth == 1 && rc.opacity && prim == 0xFFFFFFFFu) {      // records.inl:131-137: EOpacity on a sensor ray that hits nothing -- what the medium in front of the sensor absorbs / scatters, or 0
                    float al = 0.0f;
                    if (medium >= 0) { const v3 p2 = o + d * rc.alpha_dist, dd = p2 - o; const v3 tr = mediumTransmittance(md, 0.0f, sqrtf(dot(dd, dd))); al = 1 - ((0.0f + tr.x) + tr.y + tr.z) * (1.0f / 3); }
                    float4 a = q.acc[pid]; a.w = al; q.acc[pid] = a;
                }
                if (depth == 1 && rc.opacity && prim != 0xFFFFFFFFu) {      // records.inl:124-130: a sensor ray that hits a medium-transition shape -- alpha = 1 - the transmittance of what lies behind it
                    const uint32_t pmA = sc.prim_media ? sc.prim_media[prim] : 0u;
                    if (pmA) {
                        Hit ha; const int instA = q.hitInst ? q.hitInst[slot] : -1;
                        if (instA >= 0) fillHitInstanced(sc, tb, sc.instances[instA], o, d, hr.x, prim, hr.y, hr.z, ha);
                        else if (prim >= sc.n_tris) fillHitAnalytic(sc.analytic[prim - sc.n_tris], o, d, hr.x, hr.y, hr.z, ha);
                        else fillHit<false, true>(sc, tb, d, hr.x, prim, hr.y, hr.z, ha);
                        const v3 p2 = o + d * rc.alpha_dist; const int mA = targetMedium(pmA, ha.ng, d);
                        wantAlpha = true;
                        alO = make_float4(ha.p.x, ha.p.y, ha.p.z, __uint_as_float((uint32_t) (mA + 1) | (0x7FFFu << 8) | (1u << 24) | (1u << 29)));
                        alD = make_float4(p2.x, p2.y, p2.z, __uint_as_float(pid));
                    }
                }
                const int interactions = rc.max_depth - depth - 1;
                // ---- the interaction: a point in the medium, or the surface at the end of the segment
                bool nee = false; v3 nref = V(0, 0, 0), nrefN = V(0, 0, 0); Hit h; MaterialD bsdf; uint32_t pm = 0; bool bumped = false, masked = false; v3 bps = V(0, 0, 0), bpt = V(0, 0, 0), bpn = V(0, 0, 0), opac = V(1, 1, 1);
                if (mediumEvent) {
                    { const float r = 1.0f / mRec.pdfSuccess; T = T * ((ld3(md.sigma_s) * mRec.transmittance) * r); }
                    nee = others; nref = mRec.p;                             // EDirectMediumRadiance
                } else {
                    if (medium >= 0) { const float r = 1.0f / mRec.pdfFailure; T = T * (mRec.transmittance * r); }
                    if (prim == 0xFFFFFFFFu) {                               // volpath_simple.cpp:172-183: possibly attenuated radiance from the environment
                        if (ENV && emitted && (!rc.hide_emitters || scattered)) {
                            v3 value = T * (depth == 1 ? envEvalSensorRay(sc, rc, d, q.pos[pid]) : envEval(sc, d));
                            if (medium >= 0) value = value * mediumTransmittance(md, ro.w, rd.w);
                            add = value; haveAdd = true;
                        }
                        pathLen += (unsigned) depth; break;
                    }
                    const int inst = q.hitInst ? q.hitInst[slot] : -1;
                    if (inst >= 0) fillHitInstanced(sc, tb, sc.instances[inst], o, d, hr.x, prim, hr.y, hr.z, h);
                    else if (prim >= sc.n_tris) fillHitAnalytic(sc.analytic[prim - sc.n_tris], o, d, hr.x, hr.y, hr.z, h);
                    else fillHit<false, true>(sc, tb, d, hr.x, prim, hr.y, hr.z, h);
                    if (h.emitter >= 0 && emitted && (!rc.hide_emitters || scattered)) { add = T * emitterEval(tb, h.emitter, h.ns, -d); haveAdd = true; }
                    if (rc.strict_normals && (-dot(h.ng, d)) * h.wi.z < 0) { pathLen += (unsigned) depth; break; }
                    bsdf = loadMaterial(tb, h.material);
                    auto applyTexture = [&](MaterialD &mm) {
                    if (TEX) {                                               // a textured reflectance (shade.h applyTexture); only the sensor ray carries differentials
                        const uint32_t tex = (WRAP && (mm.type == MI_BSDF_T_BUMPMAP || mm.type == MI_BSDF_T_NORMALMAP)) ? 0u : (mm.flags >> 8) & 0xFFFFu;      // (an adapter's texture is its displacement / normal map)
                        if (tex) {
                            const TextureD &tx = sc.textures[tex - 1]; v3 c; float huvx = h.uvx, huvy = h.uvy;
                            const bool onAnalytic = inst < 0 && prim >= sc.n_tris;
                            if (onAnalytic) { v3 du, dv; analyticUV(sc.analytic[prim - sc.n_tris], hr.y, hr.z, o + d * hr.x, huvx, huvy, du, dv); }
                            if (tx.type == 2u) {
                                const float uvx = huvx * tx.uscale + tx.uoffset, uvy = huvy * tx.vscale + tx.voffset;
                                if (depth == 1) {
                                    v3 dpdu, dpdv;
                                    if (onAnalytic) { float tu_, tv_; analyticUV(sc.analytic[prim - sc.n_tris], hr.y, hr.z, o + d * hr.x, tu_, tv_, dpdu, dpdv); }
                                    else if (h.flags & 16u) { const TriUV &tu = sc.triuv[prim]; dpdu = ld3(tu.dpdu); dpdv = ld3(tu.dpdv); }
                                    else { AS<false>::p4 rec = tb.shade4 + prim * (uint32_t) MI_SHADE_WORDS; f4 r0 = rec[0], r1 = rec[1], r2 = rec[2]; dpdu = V(r1.x - r0.x, r1.y - r0.y, r1.z - r0.z); dpdv = V(r2.x - r0.x, r2.y - r0.y, r2.z - r0.z); }
                                    if (inst >= 0) { dpdu = xfVector(sc.instances[inst].to_world, dpdu); dpdv = xfVector(sc.instances[inst].to_world, dpdv); }
                                    const float2 sp = q.pos[pid]; v3 rxd, ryd; cameraDifferentials(sc, rc.inv_sqrt_spp, sp.x, sp.y, d, rxd, ryd);
                                    float pa[4]; computePartials(h.p, h.ng, dpdu, dpdv, o, rxd, ryd, pa);
                                    c = mipEval(sc, tx, uvx, uvy, pa[0] * tx.uscale, pa[1] * tx.vscale, pa[2] * tx.uscale, pa[3] * tx.vscale);
                                } else c = tx.filter != 0u ? mipBilinear(sc, tx, 0, uvx, uvy) : mipBox(sc, tx, 0, uvx, uvy);
                            } else c = textureEval(tx, huvx, huvy);
                            mm.reflectance[0] = c.x; mm.reflectance[1] = c.y; mm.reflectance[2] = c.z;
                        }
                    }
                    };
                    applyTexture(bsdf);
                    if (bsdf.type == MI_BSDF_T_MASK) { opac = ld3(bsdf.reflectance); masked = true; bsdf = loadMaterial(tb, (int) bsdf.distr); applyTexture(bsdf); }      // mask.cpp (shade.h)
                    if (WRAP && TEX && (bsdf.type == MI_BSDF_T_BUMPMAP || bsdf.type == MI_BSDF_T_NORMALMAP)) {      // bumpmap.cpp / normalmap.cpp: getFrame(its), then the nested record (shade.h)
                        float huvx = h.uvx, huvy = h.uvy; v3 dpdu, dpdv;
                        if (inst < 0 && prim >= sc.n_tris) analyticUV(sc.analytic[prim - sc.n_tris], hr.y, hr.z, o + d * hr.x, huvx, huvy, dpdu, dpdv);
                        else if (h.flags & 16u) { const TriUV &tu = sc.triuv[prim]; dpdu = ld3(tu.dpdu); dpdv = ld3(tu.dpdv); }
                        else { AS<false>::p4 rec = tb.shade4 + prim * (uint32_t) MI_SHADE_WORDS; f4 r0 = rec[0], r1 = rec[1], r2 = rec[2]; dpdu = V(r1.x - r0.x, r1.y - r0.y, r1.z - r0.z); dpdv = V(r2.x - r0.x, r2.y - r0.y, r2.z - r0.z); }
                        if (inst >= 0) { dpdu = xfVector(sc.instances[inst].to_world, dpdu); dpdv = xfVector(sc.instances[inst].to_world, dpdv); }
                        perturbFrame(sc, bsdf, h, huvx, huvy, dpdu, dpdv, bps, bpt, bpn); bumped = true;
                        bsdf = loadMaterial(tb, (int) bsdf.distr); applyTexture(bsdf);
                    }
                    pm = sc.prim_media ? sc.prim_media[prim] : 0u;           // (interior + 1) | (exterior + 1) << 16 of the shape that was hit; 0: not a medium transition
                    nee = others && !(h.flags & 4u); nref = h.p;            // EDirectSurfaceRadiance, BSDFs with a smooth component
                    if (!(h.flags & 2u)) nrefN = h.ns;                       // records.inl:160-164
                }
                // ---- emitter sampling for either interaction (volpath_simple.cpp:127-138 / :197-216): the record goes to k_shadow_vol, which attenuates and adds it
                if (nee) {
                    float sx, sy; next2D(ss, rc.sampler, m32, sx, sy);
                    Direct dr; const v3 value = sampleEmitterDirect<ENV, true, false, true>(sc, tb, nref, nrefN, sx, sy, dr);
                    if (dr.pdf != 0) {
                        v3 x; int m2 = medium; uint32_t onSurface = 0u;
                        if (mediumEvent) { const float ph = phaseEval(md, -d, dr.d); x = V(ph, ph, ph); }
                        else {
                            const v3 wo = toLocal(h, dr.d);
                            v3 wiQ = h.wi, woQ = wo; bool rejected = false;
                            if (WRAP && bumped) { wiQ = frameToLocal(bps, bpt, bpn, toWorld(h, h.wi)); woQ = frameToLocal(bps, bpt, bpn, toWorld(h, wo)); rejected = wo.z * woQ.z <= 0; }      // bumpmap.cpp:165-180
                            x = (!rejected && (!rc.strict_normals || dot(h.ng, dr.d) * wo.z > 0)) ? mxEval<true, WRAP>(sc, tb, bsdf, wiQ, woQ) : V(0, 0, 0);
                            if (masked) x = x * opac;                            // mask.cpp:117-118
                            if (pm) m2 = targetMedium(pm, h.ng, dr.d);      // scene.cpp:920-921
                            onSurface = 1u << 24;
                        }
                        wantShadow = true;
                        shO = make_float4(nref.x, nref.y, nref.z, __uint_as_float((uint32_t) (m2 + 1) | (((uint32_t) interactions & 0xFFFFu) << 8) | onSurface | (dr.delta ? 0u : (1u << 25))));
                        shD = make_float4(dr.p.x, dr.p.y, dr.p.z, __uint_as_float(pid)); shC = make_float4(value.x, value.y, value.z, dr.em_pdf);
                        shT = make_float4(T.x, T.y, T.z, 0.0f); shX = make_float4(x.x, x.y, x.z, 0.0f);
                    }
                }
                // ---- the continuation ray
                if (mediumEvent) {                                           // phase-function sampling (volpath_simple.cpp:141-162)
                    if ((depth + 1 >= rc.max_depth && rc.max_depth > 0) || !others) { pathLen += (unsigned) depth; break; }
                    float sx, sy; next2D(ss, rc.sampler, m32, sx, sy);
                    const v3 wo = phaseSample(md, -d, sx, sy);
                    fl = (fl & ~VOL_NULLCHAIN) | VOL_SCATTERED;
                    alive = true;
                    nrO = make_float4(mRec.p.x, mRec.p.y, mRec.p.z, 0.0f); nrD = make_float4(wo.x, wo.y, wo.z, INFINITY);
                    break;
                }
                float bPdf = 0, bEta = 1; v3 woL = V(0, 0, 0); bool sampledDelta, sampledNull;
                float sx, sy; next2D(ss, rc.sampler, m32, sx, sy);
                auto drawExtra = [&]() { return next1D(ss, rc.sampler, m32); };
                if (bsdf.type == MI_BSDF_T_THINDIELECTRIC) bsdf.flags |= MI_THIN_SIGNED_COS;      // the pdf-less ThinDielectric::sample overload this integrator calls (pt_device.h)
                v3 bw; bool passThrough = false; float invProb = 1.0f;      // mask.cpp:152-172, the overload WITHOUT a pdf: sample.x *= 1 / prob, result * opacity * (1 / prob)
                if (masked) { const float prob = luminance3(opac); if (sx < prob) { invProb = 1.0f / prob; sx *= invProb; } else passThrough = true; }
                if (passThrough) {
                    const float p = 1 - luminance3(opac);
                    woL = V(-h.wi.x, -h.wi.y, -h.wi.z); bEta = 1.0f; bPdf = p; sampledDelta = true; sampledNull = true;
                    bw = V((1.0f - opac.x) / p, (1.0f - opac.y) / p, (1.0f - opac.z) / p);
                } else
                if (WRAP && bumped) {                                        // bumpmap.cpp:196-216
                    const v3 wiQ = frameToLocal(bps, bpt, bpn, toWorld(h, h.wi)); v3 woQ = V(0, 0, 0);
                    bw = mxSample<true, WRAP>(sc, tb, bsdf, wiQ, sx, sy, drawExtra, woQ, bPdf, bEta, sampledDelta, sampledNull);
                    if (!isZero(bw)) { woL = toLocal(h, frameToWorld(bps, bpt, bpn, woQ)); if (woL.z * woQ.z <= 0) bw = V(0, 0, 0); }
                } else bw = mxSample<true, WRAP>(sc, tb, bsdf, h.wi, sx, sy, drawExtra, woL, bPdf, bEta, sampledDelta, sampledNull);
                if (masked && !passThrough) bw = V(bw.x * opac.x * invProb, bw.y * opac.y * invProb, bw.z * opac.z * invProb);
                if (isZero(bw)) { pathLen += (unsigned) depth; break; }
                // which radiance types the next iteration gathers (volpath_simple.cpp:236-257)
                const bool rtOthers = (depth + 1 < rc.max_depth || rc.max_depth < 0) && others; bool rtEmitted = false; bool nullChain = (fl & VOL_NULLCHAIN) != 0;
                if ((depth < rc.max_depth || rc.max_depth < 0) && others && sampledDelta && (!sampledNull || nullChain)) { rtEmitted = true; nullChain = true; }
                else nullChain = nullChain && sampledNull;
                if (!rtOthers && !rtEmitted) { pathLen += (unsigned) depth; break; }
                const v3 wo = toWorld(h, woL);
                if (dot(h.ng, wo) * woL.z <= 0 && rc.strict_normals) { pathLen += (unsigned) depth; break; }
                T = T * bw; eta *= bEta;
                if (pm) medium = targetMedium(pm, h.ng, wo);                 // its.isMediumTransition(): rRec.medium = its.getTargetMedium(wo)
                fl = (rtOthers ? VOL_OTHERS : 0u) | (rtEmitted ? VOL_EMITTED : 0u) | (nullChain ? VOL_NULLCHAIN : 0u) | ((scattered || !sampledNull) ? VOL_SCATTERED : 0u);
                alive = true;
                nrO = make_float4(h.p.x, h.p.y, h.p.z, MI_EPSILON); nrD = make_float4(wo.x, wo.y, wo.z, INFINITY);
            } while (false);
            if (alive) {
                nS0 = make_uint4(pid, ss.a, ss.b, (ss.dim & 0xFFu) | ((uint32_t) (depth + 1) << 8) | fl | ((uint32_t) (medium + 1) << 20));
                nS1 = make_float4(T.x, T.y, T.z, eta);
            }
            if (haveAdd) { float4 a = q.acc[pid]; a.x += add.x; a.y += add.y; a.z += add.z; q.acc[pid] = a; }
        }
        const unsigned long long mE = __ballot(wantAlpha);
        if (wantAlpha) { const uint64_t w = shBase + 2u * q.cap - 1u - (outE + (uint32_t) __popcll(mE & lt)); q.shO[w] = alO; q.shD[w] = alD; }      // from the back of the segment's area: k_shadow_vol runs them BEFORE the emitter-sampling records
        outE += (uint32_t) __popcll(mE);
        const unsigned long long mS = __ballot(wantShadow);
        if (wantShadow) { const uint64_t w = shBase + outS + (uint32_t) __popcll(mS & lt); q.shO[w] = shO; q.shD[w] = shD; q.shC[w] = shC; q.shT[w] = shT; q.shX[w] = shX; }
        outS += (uint32_t) __popcll(mS);
        const unsigned long long mA = __ballot(alive);
        if (alive) { const uint64_t w = segBase + outA + (uint32_t) __popcll(mA & lt); q.rayO[nb][w] = nrO; q.rayD[nb][w] = nrD; q.st0[nb][w] = nS0; q.st1[nb][w] = nS1; q.st2[nb][w] = 0.0f; }
        outA += (uint32_t) __popcll(mA);
    }
    if (lane == 0) { q.count[nb][seg] = outA; q.shCount[seg] = outS | (outE << 16); }      // (cap <= 65535 in the volumetric modes, api.cpp)
    }
    for (int off = 32; off > 0; off >>= 1) pathLen += __shfl_down(pathLen, off);
    if (lane == 0 && pathLen) atomicAdd(&q.counters[2], pathLen);
}

// Scene::evalTransmittance (scene.cpp:650-713) for every shadow record of the segment, then the deferred `Li += throughput * value * (bsdf | phase)`
template <bool WIDE, bool NX>      // NX: ENull lobes behind a wrapper (mask, mixturebsdf with an index-matched child) are present: surfaceNullEval instead of the plain records' pass-through value
__global__ __launch_bounds__(WG) void k_shadow_vol(DScene sc, Queues q) {
    __shared__ int s_stk[STACK_DEPTH * WG];
    const uint32_t tid = threadIdx.x;
    Tabs<false> tb; tb.shade4 = (AS<false>::p4) sc.shade; tb.materials4 = (AS<false>::p4) sc.materials; tb.emitters4 = (AS<false>::p4) sc.emitters; tb.emitter_cdf = sc.emitter_cdf; tb.area_cdf = sc.area_cdf;
    unsigned long long rays = 0;
    for (uint32_t seg = blockIdx.x; seg < q.n_seg; seg += gridDim.x) {
    // two passes over the segment's records, a barrier in between: first the alpha walks (written from the back of the area), then the emitter-sampling records -- a
    // path can own one of each, and its accumulator is a plain read-modify-write
    const uint32_t cnt = q.shCount[seg], nFront = cnt & 0xFFFFu, nBack = cnt >> 16;
    const uint64_t areaBase = (uint64_t) seg * q.cap * 2u;
    for (int phase = 0; phase < 2; ++phase) {
    if (phase == 1) __syncthreads();
    const uint32_t n = phase == 0 ? nBack : nFront;
    for (uint32_t i0 = tid; i0 < n; i0 += WG) {
        const uint64_t segBase = phase == 0 ? areaBase + 2u * q.cap - 1u - i0 : areaBase + i0; const uint32_t i = 0;
        const float4 so = q.shO[segBase + i], sd = q.shD[segBase + i];
        const uint32_t bits = __float_as_uint(so.w), pid = __float_as_uint(sd.w);
        int medium = (int) (bits & 0xFFu) - 1; const int maxInteractions = (int) (int16_t) ((bits >> 8) & 0xFFFFu); const bool p1OnSurface = (bits >> 24) & 1u, p2OnSurface = (bits >> 25) & 1u;
        const v3 p1 = V(so.x, so.y, so.z), p2 = V(sd.x, sd.y, sd.z);
        v3 d = p2 - p1; float remaining = sqrtf(dot(d, d)); { const float r = 1.0f / remaining; d = d * r; }
        const float lengthFactor = p2OnSurface ? (1 - MI_SHADOW_EPSILON) : 1;
        v3 o = p1; float rmint = p1OnSurface ? MI_EPSILON : 0.0f, rmaxt = remaining * lengthFactor;
        v3 tr = V(1, 1, 1); int interactions = 0; bool blocked = false;
        while (remaining > 0) {
            float mint, maxt, t = INFINITY, u = 0, v = 0; uint32_t prim = 0xFFFFFFFFu; int inst = -1; bool surface = false;
            ++rays;                                                          // skdtree.cpp:152 ++shadowRaysTraced
            if (clipInterval(sc, o, d, rmint, rmaxt, true, mint, maxt)) surface = traverse<false, 3, WIDE>(sc, o, d, mint, maxt, s_stk + tid, t, prim, u, v, inst);
            if (!surface) t = INFINITY;
            int material = -1; v3 n = V(0, 0, 0);
            if (surface) {
                if (prim >= sc.n_tris) { const AnalyticD &a = sc.analytic[prim - sc.n_tris]; material = a.material; Hit h; fillHitAnalytic(a, o, d, t, u, v, h); n = h.ng; }
                else {                                                       // skdtree.cpp:165-171: n = normalize(cross(p1 - p0, p2 - p0)), NOT flipped towards the shading normal
                    AS<false>::p4 rec = tb.shade4 + prim * (uint32_t) MI_SHADE_WORDS; const f4 r0 = rec[0], r1 = rec[1], r2 = rec[2]; material = __float_as_int(r0.w);
                    v3 fn = cross(V(r1.x - r0.x, r1.y - r0.y, r1.z - r0.z), V(r2.x - r0.x, r2.y - r0.y, r2.z - r0.z)); const float len = sqrtf(dot(fn, fn));
                    if (!isZero(fn)) { const float r = 1.0f / len; fn = fn * r; }
                    n = fn;
                }
                if (interactions == maxInteractions || !(NX ? surfaceHasNull(tb, loadMaterial(tb, material)) : materialHasNull(loadMaterial(tb, material).type))) { blocked = true; break; }   // an occluder: zero transmittance
            }
            if (medium >= 0) tr = tr * mediumTransmittance(sc.media[medium], 0.0f, minf(t, remaining));
            if (!surface || isZero(tr)) break;
            tr = tr * (NX ? surfaceNullEval(sc, tb, loadMaterial(tb, material), o, d, t, prim, u, v, inst, -dot(d, n), true) : materialNullEval(loadMaterial(tb, material), -dot(d, n)));      // its.geoFrame = Frame(n): cosTheta(wi) = -dot(d, n)
            const uint32_t pm = sc.prim_media ? sc.prim_media[prim] : 0u;   // `null`: bsdf->eval(bRec, EDiscrete) with typeMask = ENull is 1 (null.cpp:48-50)
            if (pm) {
                if (medium != targetMedium(pm, n, -d)) { blocked = true; break; }      // medium inconsistency (scene.cpp:689-692)
                medium = targetMedium(pm, n, d);
            }
            if (++interactions > 100) break;
            o = o + d * t; remaining -= t; rmaxt = remaining * lengthFactor; rmint = MI_EPSILON;
        }
        if (bits & (1u << 29)) {                                                 // the alpha walk of a sensor ray (records.inl:128-130)
            const v3 tf = blocked ? V(0, 0, 0) : tr; float4 a = q.acc[pid]; a.w = 1 - ((0.0f + tf.x) + tf.y + tf.z) * (1.0f / 3); q.acc[pid] = a;
            continue;
        }
        if (blocked) continue;
        const float4 c = q.shC[segBase + i], tt = q.shT[segBase + i], xx = q.shX[segBase + i];
        const float r = 1.0f / c.w;
        const v3 value = V(c.x, c.y, c.z) * (tr * r);                       // value *= evalTransmittance(...) / emPdf (scene.cpp:897-899)
        if (!isZero(value)) {
            const v3 li = (V(tt.x, tt.y, tt.z) * value) * V(xx.x, xx.y, xx.z);
            float4 a = q.acc[pid]; a.x += li.x; a.y += li.y; a.z += li.z; q.acc[pid] = a;
        }
    }
    }
    }
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off);
    if ((tid & 63u) == 0 && rays) atomicAdd(&q.counters[1], rays);
}

extern "C" {
void mi_launch_shade_vol(const DScene &sc, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, size_t lds, hipStream_t st) {
    const bool env = sc.env_index >= 0;
    if (sc.has_adapters & 1u) { if (env) launchWithLds(k_shade_vol<true, true, true>, grid, lds, st, sc, rc, q, buf); else launchWithLds(k_shade_vol<true, false, true>, grid, lds, st, sc, rc, q, buf); }
    else if (sc.n_textures) { if (env) launchWithLds(k_shade_vol<true, true, false>, grid, lds, st, sc, rc, q, buf); else launchWithLds(k_shade_vol<true, false, false>, grid, lds, st, sc, rc, q, buf); }
    else { if (env) launchWithLds(k_shade_vol<false, true, false>, grid, lds, st, sc, rc, q, buf); else launchWithLds(k_shade_vol<false, false, false>, grid, lds, st, sc, rc, q, buf); }
}
void mi_launch_shadow_vol(const DScene &sc, const Queues &q, uint32_t grid, hipStream_t st) {
    const bool nx = (sc.has_adapters & 2u) != 0;
    if (sc.bvh_wide) { if (nx) hipLaunchKernelGGL((k_shadow_vol<true, true>), dim3(grid), dim3(WG), 0, st, sc, q); else hipLaunchKernelGGL((k_shadow_vol<true, false>), dim3(grid), dim3(WG), 0, st, sc, q); }
    else { if (nx) hipLaunchKernelGGL((k_shadow_vol<false, true>), dim3(grid), dim3(WG), 0, st, sc, q); else hipLaunchKernelGGL((k_shadow_vol<false, false>), dim3(grid), dim3(WG), 0, st, sc, q); }
}
}
