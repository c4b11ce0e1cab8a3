// kernels_volmis.hip -- VolumetricPathTracer::Li (reference src/integrators/path/volpath.cpp:84-343): the volumetric loop of kernels_vol.hip WITH multiple importance
// sampling.  Differences to the simple variant, stage by stage:
//   k_shade_volmis   * emitter sampling (:127-150 / :215-250) carries the power-heuristic weight against the phase-function / BSDF density (known here) in its shadow record;
//                    * the ray spawned by phase-function / BSDF sampling looks for an emitter (rayIntersectAndLookForEmitter, :370-431): its FIRST hit comes from k_extend and
//                      is judged at the start of the next pass -- an emitter: `Li += throughput * (T_medium Le) * w(samplingPdf, emitterPdf)`; an index-matched boundary:
//                      the search goes on BEHIND it, as a record of its own in the shadow queue (the boundary itself is where the path continues, and `null` surfaces
//                      sample no emitters, so a pass leaves at most one record of each kind per path: the shadow queue holds 2 x cap records per segment here);
//                    * a `null` pass-through (:268-276) is not an iteration of its own: no Russian roulette, `scattered` unchanged, radiance type reset.
//   k_shadow_volmis  kind 0: Scene::evalTransmittance as in k_shadow_vol, contribution times the stored weight; kind 1: the emitter search (closest-hit walk with full
//                    intersections through `null` boundaries and media; DirectSamplingRecord::setQuery keeps the LAST segment's length as `dist`, records.inl:170-178).
// State word st0.w: bits 0..7 dimension, 8..15 depth, 16 EEmittedRadiance, 17 the spawning sample was a Dirac delta, 18 the ray looks for an emitter, 19 scattered,
// 20..27 medium + 1, 28 dot(d, refN) >= 0 at the spawning vertex, 29 skip the Russian roulette (the previous pass was a `null` pass-through); st2 = the sampling density.
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
// density of the environment emitter for the direction d as Scene::pdfEmitterDirect reports it: EnvironmentMap::pdfDirect (envmap.cpp:549-560) or
// ConstantBackgroundEmitter::pdfDirect (constant.cpp:219-233: cosine-weighted about the spawning vertex' reference normal -- `cosRef` = dot(d, refN), 2 = none)
template <bool L> DEV float envLumPdf(const DScene &sc, const Tabs<L> &tb, v3 d, float cosRef) {
    const float pdfSA = sc.env_constant ? (cosRef != 2.0f ? MI_INV_PI * maxf(0.0f, cosRef) : MI_INV_FOURPI) : envPdfDirection(sc, mat3(sc.env_to_local, d));
    return pdfSA * (loadEmitter(tb, sc.env_index).weight * sc.emitter_norm);
}

#define VM_EMITTED (1u << 16)
#define VM_DELTA (1u << 17)
#define VM_SEARCH (1u << 18)
#define VM_SCATTERED (1u << 19)
#define VM_FACING (1u << 28)
#define VM_SKIPRR (1u << 29)
// shadow record bits (shO.w): 0..7 medium + 1, 8..23 maxInteractions (int16), 24 p1OnSurface, 25 p2OnSurface, 26 kind = emitter search, 27 facingRef, 28 delta sample

template <bool TEX, bool ENV, bool WRAP>      // WRAP: mixturebsdf / bumpmap / normalmap records present (with TEX), as in kernels_vol.hip
__global__ __launch_bounds__(WG) void k_shade_volmis(DScene sc, RenderConst rc, Queues q, int buf) {
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
    for (uint32_t seg = blockIdx.x * (WG / 64) + wave; seg < q.n_seg; seg += gridDim.x * (WG / 64)) {
    const uint32_t n = q.count[buf][seg];
    const uint64_t segBase = (uint64_t) seg * q.cap, shBase = segBase * 2u;
    uint32_t outA = 0, outS = 0, outE = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        bool alive = false, wantShadow = false, wantSearch = false, wantAlpha = false; float4 alO, alD;
        float4 shO, shD, shC, shT, shX, seO, seD, seC, seT, seX, nrO, nrD, nS1; uint4 nS0; float nS2 = 0, nS3 = 2.0f;
        if (i < n) {
            const uint64_t slot = segBase + i;
            const float4 ro = q.rayO[buf][slot], rd = q.rayD[buf][slot], hr = q.hit[slot]; const uint4 s0 = q.st0[buf][slot]; const float4 s1 = q.st1[buf][slot]; const float prevPdf = q.st2[buf][slot]; const float prevCos = (ENV && sc.env_constant) ? q.st3[buf][slot] : 2.0f;
            SamplerState ss; const uint32_t pid = s0.x; ss.a = s0.y; ss.b = s0.z; ss.dim = s0.w & 0xFFu;
            const int depth = (int) ((s0.w >> 8) & 0xFFu); uint32_t fl = s0.w; int medium = (int) ((s0.w >> 20) & 0xFFu) - 1;
            const v3 o = V(ro.x, ro.y, ro.z), d = V(rd.x, rd.y, rd.z); v3 T = V(s1.x, s1.y, s1.z); float eta = s1.w;
            const uint32_t prim = __float_as_uint(hr.w); const float tHit = prim == 0xFFFFFFFFu ? INFINITY : hr.x;
            v3 add = V(0, 0, 0); bool haveAdd = false;
            MediumD md; if (medium >= 0) md = sc.media[medium];
            Hit h; bool haveHit = false; const int inst = (q.hitInst && prim != 0xFFFFFFFFu) ? q.hitInst[slot] : -1;
            auto fill = [&]() {
                if (haveHit) return; haveHit = true;
                if (inst >= 0) fillHitInstanced(sc, tb, sc.instances[inst], o, d, hr.x, prim, hr.y, hr.z, h);
                else if (prim >= sc.n_tris) fillHitAnalytic(sc.analytic[prim - sc.n_tris], o, d, hr.x, hr.y, hr.z, h);
                else fillHit<false, true>(sc, tb, d, hr.x, prim, hr.y, hr.z, h);
            };
            do {
                // ---- rayIntersectAndLookForEmitter, first intersection (volpath.cpp:370-431) for the ray the previous pass sampled
                if ((fl & VM_SEARCH) && prim != 0xFFFFFFFFu) {
                    const v3 segT = medium >= 0 ? mediumTransmittance(md, 0.0f, tHit) : V(1, 1, 1);
                    fill();
                    const int maxInteractions = rc.max_depth - depth;           // m_maxDepth - rRec.depth - 1 at the spawning vertex (depth has advanced by one since)
                    const MaterialD hm = loadMaterial(tb, h.material); const bool isNull = surfaceHasNull(tb, hm);
                    if (h.emitter >= 0) {                                        // an emitter (also one behind a `null` BSDF, :388-390)
                        const v3 value = segT * emitterEval(tb, h.emitter, h.ns, -d);
                        if (!isZero(value)) {
                            const float lumPdf = (fl & VM_DELTA) ? 0.0f : pdfEmitterDirect<true>(sc, tb, h.emitter, o, d, h.ns, hr.x, (fl & VM_FACING) != 0);
                            add = (T * value) * miWeight(prevPdf, lumPdf); haveAdd = true;
                        }
                    } else if (isNull && maxInteractions != 0 && !isZero(segT)) {   // the search continues behind the boundary: a record for k_shadow_volmis
                        const uint32_t pm = sc.prim_media ? sc.prim_media[prim] : 0u;
                        const int m2 = pm ? targetMedium(pm, h.ng, d) : medium;
                        const v3 p = o + d * hr.x;                              // ray.o = ray(its->t)
                        wantSearch = true;
                        seO = make_float4(p.x, p.y, p.z, __uint_as_float((uint32_t) (m2 + 1) | (((uint32_t) maxInteractions & 0xFFFFu) << 8) | (1u << 26) | ((fl & VM_FACING) ? (1u << 27) : 0u) | ((fl & VM_DELTA) ? (1u << 28) : 0u)));
                        const v3 segN = segT * surfaceNullEval(sc, tb, hm, o, d, hr.x, prim, hr.y, hr.z, inst, -dot(d, h.ns), false);      // wo = shFrame.toLocal(ray.d): cosTheta(wi) = -dot(d, ns) (volpath.cpp:399-402)
                        seD = make_float4(d.x, d.y, d.z, __uint_as_float(pid)); seC = make_float4(segN.x, segN.y, segN.z, prevPdf);
                        seT = make_float4(T.x, T.y, T.z, 0.0f); seX = make_float4(o.x, o.y, o.z, prevCos);     // dRec.ref = the spawning vertex (+ the cosine a `constant` environment's density needs)
                    }
                }
                if (ENV && (fl & VM_SEARCH) && prim == 0xFFFFFFFFu && medium < 0) {      // :421-426: the ray left the scene -> the environment map (inside a medium the unbounded segment has zero transmittance)
                    float nearT, farT;
                    if (bsphereIntersect(sc, o, d, nearT, farT) && !(nearT > 0) && !(farT < 0)) {        // EnvironmentMap::fillDirectSamplingRecord (envmap.cpp:362-378)
                        const v3 value = envEval(sc, d);
                        if (!isZero(value)) {
                            const float lumPdf = (fl & VM_DELTA) ? 0.0f : envLumPdf(sc, tb, d, prevCos);
                            add = (T * value) * miWeight(prevPdf, lumPdf); haveAdd = true;
                        }
                    }
                }
                if (depth > 1 && !(fl & VM_SKIPRR) && depth - 1 >= rc.rr_depth) {     // volpath.cpp:326-337 (the previous iteration's tail)
                    const float qq = minf(maxf(maxf(T.x, T.y), T.z) * eta * eta, 0.95f);
                    if (next1D(ss, rc.sampler, m32) >= qq) { pathLen += (unsigned) depth; break; }
                    const float r = 1.0f / qq; T = T * r;
                }
                if (!(depth <= rc.max_depth || rc.max_depth < 0)) { pathLen += (unsigned) depth; break; }
                const bool emitted = (fl & VM_EMITTED) != 0, scattered = (fl & VM_SCATTERED) != 0;
                MediumRec mRec; bool mediumEvent = false;
                if (medium >= 0) mediumEvent = mediumSampleDistance(md, o, d, 0.0f, tHit, ss, rc.sampler, m32, mRec);
                if (depth == 1 && rc.opacity && prim == 0xFFFFFFFFu) {          // records.inl:131-137
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
                bool nee = false; v3 nref = V(0, 0, 0), nrefN = V(0, 0, 0); MaterialD bsdf; uint32_t pm = 0; bool bumped = false, masked = false; v3 bps = V(0, 0, 0), bpt = V(0, 0, 0), bpn = V(0, 0, 0), opac = V(1, 1, 1);
                if (mediumEvent) {
                    if (depth >= rc.max_depth && rc.max_depth != -1) { pathLen += (unsigned) depth; break; }      // :112-113
                    { const float r = 1.0f / mRec.pdfSuccess; T = T * ((ld3(md.sigma_s) * mRec.transmittance) * r); }
                    nee = true; nref = mRec.p;
                } else {
                    if (medium >= 0) { const float r = 1.0f / mRec.pdfFailure; T = T * (mRec.transmittance * r); }
                    if (prim == 0xFFFFFFFFu) {                               // volpath.cpp:181-192
                        if (ENV && emitted && (!rc.hide_emitters || scattered)) {
                            v3 value = T * (depth == 1 ? envEvalSensorRay(sc, rc, d, q.pos[pid]) : envEval(sc, d));
                            if (medium >= 0) value = value * mediumTransmittance(md, ro.w, rd.w);
                            add = haveAdd ? add + value : value; haveAdd = true;
                        }
                        pathLen += (unsigned) depth; break;
                    }
                    fill();
                    if (h.emitter >= 0 && emitted && (!rc.hide_emitters || scattered)) { const v3 e = T * emitterEval(tb, h.emitter, h.ns, -d); add = haveAdd ? add + e : e; haveAdd = true; }
                    if (depth >= rc.max_depth && rc.max_depth != -1) { pathLen += (unsigned) depth; break; }      // :203-204
                    if ((-dot(h.ng, d)) * h.wi.z < 0 && rc.strict_normals) { pathLen += (unsigned) depth; break; }
                    bsdf = loadMaterial(tb, h.material);
                    auto applyTexture = [&](MaterialD &mm) {
                    if (TEX) {
                        const uint32_t tex = (WRAP && (mm.type == MI_BSDF_T_BUMPMAP || mm.type == MI_BSDF_T_NORMALMAP)) ? 0u : (mm.flags >> 8) & 0xFFFFu;
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
                    pm = sc.prim_media ? sc.prim_media[prim] : 0u;
                    nee = !(h.flags & 4u); nref = h.p;
                    if (!(h.flags & 2u)) nrefN = h.ns;
                }
                // ---- emitter sampling with the power heuristic (volpath.cpp:127-150 / :215-250)
                if (nee) {
                    float sx, sy; next2D(ss, rc.sampler, m32, sx, sy);
                    Direct dr; const v3 value = sampleEmitterDirect<ENV, true, false, true>(sc, tb, nref, nrefN, sx, sy, dr);
                    if (dr.pdf != 0) {
                        v3 x; float w; int m2 = medium; uint32_t onSurface = 0u;
                        if (mediumEvent) { const float ph = phaseEval(md, -d, dr.d); x = V(ph, ph, ph); w = miWeight(dr.pdf, dr.delta ? 0.0f : ph); }      // PhaseFunction::pdf = eval (phase.cpp:21-23)
                        else {
                            const v3 wo = toLocal(h, dr.d);
                            v3 wiQ = h.wi, woQ = wo; bool rejected = false;
                            if (WRAP && bumped) { wiQ = frameToLocal(bps, bpt, bpn, toWorld(h, h.wi)); woQ = frameToLocal(bps, bpt, bpn, toWorld(h, wo)); rejected = wo.z * woQ.z <= 0; }      // bumpmap.cpp:165-180
                            v3 bv = rejected ? V(0, 0, 0) : mxEval<true, WRAP>(sc, tb, bsdf, wiQ, woQ);
                            if (masked) bv = bv * opac;                           // mask.cpp:117-118
                            const bool ok = !isZero(bv) && (!rc.strict_normals || dot(h.ng, dr.d) * wo.z > 0);
                            float bp = 0.0f;
                            if (ok && !dr.delta) { bp = mxPdf<true, WRAP>(sc, tb, bsdf, wiQ, woQ); if (masked) bp *= luminance3(opac); }      // mask.cpp:141-146
                            x = ok ? bv : V(0, 0, 0); w = ok ? miWeight(dr.pdf, bp) : 0.0f;
                            if (pm) m2 = targetMedium(pm, h.ng, dr.d);
                            onSurface = 1u << 24;
                        }
                        wantShadow = true;
                        shO = make_float4(nref.x, nref.y, nref.z, __uint_as_float((uint32_t) (m2 + 1) | (((uint32_t) interactions & 0xFFFFu) << 8) | onSurface | (dr.delta ? 0u : (1u << 25))));
                        shD = make_float4(dr.p.x, dr.p.y, dr.p.z, __uint_as_float(pid)); shC = make_float4(value.x, value.y, value.z, dr.em_pdf);
                        shT = make_float4(T.x, T.y, T.z, 0.0f); shX = make_float4(x.x, x.y, x.z, w);
                    }
                }
                // ---- the continuation ray
                if (mediumEvent) {                                           // phase-function sampling (volpath.cpp:156-166); its density = its value
                    float sx, sy; next2D(ss, rc.sampler, m32, sx, sy);
                    const v3 wo = phaseSample(md, -d, sx, sy);
                    nS2 = phaseEval(md, -d, wo);
                    fl = VM_SEARCH | VM_SCATTERED | VM_FACING;              // type = ERadianceNoEmission; refN = 0 -> dot(d, refN) >= 0 holds
                    nS3 = 2.0f;
                    alive = true;
                    nrO = make_float4(mRec.p.x, mRec.p.y, mRec.p.z, 0.0f); nrD = make_float4(wo.x, wo.y, wo.z, INFINITY);
                    break;
                }
                float bPdf = 0, bEta = 1; v3 woL = V(0, 0, 0); bool sampledDelta, sampledNull;
                float sx, sy; next2D(ss, rc.sampler, m32, sx, sy);
                auto drawExtra = [&]() { return next1D(ss, rc.sampler, m32); };
                v3 bw; bool passThrough = false;                            // mask.cpp:174-214 (shade.h)
                if (masked) { const float prob = luminance3(opac); if (sx < prob) sx /= prob; else passThrough = true; }
                if (passThrough) {
                    const float p = 1 - luminance3(opac);
                    woL = V(-h.wi.x, -h.wi.y, -h.wi.z); bEta = 1.0f; bPdf = p; sampledDelta = true; sampledNull = true;
                    bw = V((1.0f - opac.x) / p, (1.0f - opac.y) / p, (1.0f - opac.z) / p);
                } else
                if (WRAP && bumped) {                                        // bumpmap.cpp:218-240
                    const v3 wiQ = frameToLocal(bps, bpt, bpn, toWorld(h, h.wi)); v3 woQ = V(0, 0, 0);
                    bw = mxSample<true, WRAP>(sc, tb, bsdf, wiQ, sx, sy, drawExtra, woQ, bPdf, bEta, sampledDelta, sampledNull);
                    if (!isZero(bw)) { woL = toLocal(h, frameToWorld(bps, bpt, bpn, woQ)); if (woL.z * woQ.z <= 0) bw = V(0, 0, 0); }
                } else bw = mxSample<true, WRAP>(sc, tb, bsdf, h.wi, sx, sy, drawExtra, woL, bPdf, bEta, sampledDelta, sampledNull);
                if (masked && !passThrough) { const float prob = luminance3(opac); bw = V(bw.x * opac.x / prob, bw.y * opac.y / prob, bw.z * opac.z / prob); bPdf *= prob; }
                if (isZero(bw)) { pathLen += (unsigned) depth; break; }
                const v3 wo = toWorld(h, woL);
                if (dot(h.ng, wo) * woL.z <= 0 && rc.strict_normals) { pathLen += (unsigned) depth; break; }
                T = T * bw; eta *= bEta;
                if (pm) medium = targetMedium(pm, h.ng, wo);
                if (sampledNull) fl = (scattered ? VM_SCATTERED : VM_EMITTED) | VM_SKIPRR;           // :268-276: type = scattered ? ERadianceNoEmission : ERadiance; no emitter search; depth++ without the roulette
                else fl = VM_SEARCH | VM_SCATTERED | (sampledDelta ? VM_DELTA : 0u) | (dot(wo, nrefN) >= 0 ? VM_FACING : 0u);
                nS2 = bPdf; nS3 = (h.flags & 2u) ? 2.0f : dot(wo, nrefN);
                alive = true;
                nrO = make_float4(h.p.x, h.p.y, h.p.z, MI_EPSILON); nrD = make_float4(wo.x, wo.y, wo.z, INFINITY);
            } while (false);
            if (alive) {
                nS0 = make_uint4(pid, ss.a, ss.b, (ss.dim & 0xFFu) | ((uint32_t) (depth + 1) << 8) | fl | ((uint32_t) (medium + 1) << 20));
                nS1 = make_float4(T.x, T.y, T.z, eta);
            }
            if (haveAdd) { float4 a = q.acc[pid]; a.x += add.x; a.y += add.y; a.z += add.z; q.acc[pid] = a; }
        }
        const unsigned long long mE = __ballot(wantSearch | wantAlpha);      // (never both: the alpha walk belongs to the sensor ray, which searches for nothing)
        // search / alpha records from the BACK of the segment's area: k_shadow_volmis runs them before the emitter-sampling records (a path can own one of each; the
        // reference adds the emitter found by the search first, volpath.cpp:166-173 before :127-150 of the next iteration)
        if (wantSearch) { const uint64_t w = shBase + 2u * q.cap - 1u - (outE + (uint32_t) __popcll(mE & lt)); q.shO[w] = seO; q.shD[w] = seD; q.shC[w] = seC; q.shT[w] = seT; q.shX[w] = seX; }
        else if (wantAlpha) { const uint64_t w = shBase + 2u * q.cap - 1u - (outE + (uint32_t) __popcll(mE & lt)); q.shO[w] = alO; q.shD[w] = alD; }
        outE += (uint32_t) __popcll(mE);
        const unsigned long long mS = __ballot(wantShadow);
        if (wantShadow) { const uint64_t w = shBase + outS + (uint32_t) __popcll(mS & lt); q.shO[w] = shO; q.shD[w] = shD; q.shC[w] = shC; q.shT[w] = shT; q.shX[w] = shX; }
        outS += (uint32_t) __popcll(mS);
        const unsigned long long mA = __ballot(alive);
        if (alive) { const uint64_t w = segBase + outA + (uint32_t) __popcll(mA & lt); q.rayO[nb][w] = nrO; q.rayD[nb][w] = nrD; q.st0[nb][w] = nS0; q.st1[nb][w] = nS1; q.st2[nb][w] = nS2; if (ENV && sc.env_constant) q.st3[nb][w] = nS3; }
        outA += (uint32_t) __popcll(mA);
    }
    if (lane == 0) { q.count[nb][seg] = outA; q.shCount[seg] = outS | (outE << 16); }      // (cap <= 65535 in the volumetric modes, api.cpp)
    }
    for (int off = 32; off > 0; off >>= 1) pathLen += __shfl_down(pathLen, off);
    if (lane == 0 && pathLen) atomicAdd(&q.counters[2], pathLen);
}

template <bool WIDE, bool ENV, bool NX>      // NX: as in k_shadow_vol
__global__ __launch_bounds__(WG) void k_shadow_volmis(DScene sc, Queues q) {
    __shared__ int s_stk[STACK_DEPTH * WG];
    const uint32_t tid = threadIdx.x;
    Tabs<false> tb; tb.shade4 = (AS<false>::p4) sc.shade; tb.materials4 = (AS<false>::p4) sc.materials; tb.emitters4 = (AS<false>::p4) sc.emitters; tb.emitter_cdf = sc.emitter_cdf; tb.area_cdf = sc.area_cdf;
    unsigned long long shadowRays = 0, normalRays = 0;
    for (uint32_t seg = blockIdx.x; seg < q.n_seg; seg += gridDim.x) {
    // two passes, a barrier in between: the records written from the back of the area (emitter searches, alpha walks) first, then the emitter-sampling records
    const uint32_t cnt = q.shCount[seg], nFront = cnt & 0xFFFFu, nBack = cnt >> 16;
    const uint64_t areaBase = (uint64_t) seg * q.cap * 2u;
    for (int phase = 0; phase < 2; ++phase) {
    if (phase == 1) __syncthreads();
    const uint32_t n = phase == 0 ? nBack : nFront;
    for (uint32_t i0 = tid; i0 < n; i0 += WG) {
        const uint64_t shBase = phase == 0 ? areaBase + 2u * q.cap - 1u - i0 : areaBase + i0; const uint32_t i = 0;
        const float4 so = q.shO[shBase + i], sd = q.shD[shBase + i];
        const uint32_t bits = __float_as_uint(so.w), pid = __float_as_uint(sd.w);
        int medium = (int) (bits & 0xFFu) - 1; const int maxInteractions = (int) (int16_t) ((bits >> 8) & 0xFFFFu);
        if (bits & (1u << 26)) {
            // ---- kind 1: rayIntersectAndLookForEmitter behind the first index-matched boundary (volpath.cpp:381-419); interactions = 1 so far
            v3 o = V(so.x, so.y, so.z); const v3 d = V(sd.x, sd.y, sd.z);
            const float4 c = q.shC[shBase + i]; v3 tr = V(c.x, c.y, c.z); int interactions = 1; bool surface = false; Hit h; float t = INFINITY;
            while (true) {
                float mint, maxt, u = 0, v = 0; uint32_t prim = 0xFFFFFFFFu; int inst = -1; t = INFINITY; surface = false;
                ++normalRays;                                                    // Scene::rayIntersect(ray, its): a "normal" ray (skdtree.cpp:118)
                if (clipInterval(sc, o, d, MI_EPSILON, INFINITY, false, mint, maxt)) surface = traverse<false, 3, WIDE>(sc, o, d, mint, maxt, s_stk + tid, t, prim, u, v, inst);
                if (!surface) t = INFINITY;
                if (medium >= 0) tr = tr * mediumTransmittance(sc.media[medium], 0.0f, t);
                if (!surface) break;
                if (inst >= 0) fillHitInstanced(sc, tb, sc.instances[inst], o, d, t, prim, u, v, h);
                else if (prim >= sc.n_tris) fillHitAnalytic(sc.analytic[prim - sc.n_tris], o, d, t, u, v, h);
                else fillHit<false, true>(sc, tb, d, t, prim, u, v, h);
                const MaterialD hm = loadMaterial(tb, h.material);
                if (interactions == maxInteractions || !(NX ? surfaceHasNull(tb, hm) : materialHasNull(hm.type)) || h.emitter >= 0) break;
                if (isZero(tr)) { surface = false; break; }
                const uint32_t pm = sc.prim_media ? sc.prim_media[prim] : 0u;
                if (pm) medium = targetMedium(pm, h.ng, d);
                tr = tr * (NX ? surfaceNullEval(sc, tb, hm, o, d, t, prim, u, v, inst, -dot(d, h.ns), false) : materialNullEval(hm, -dot(d, h.ns)));
                o = o + d * t;
                if (++interactions > 100) { surface = false; break; }
            }
            if (ENV && !surface && interactions <= 100 && !isZero(tr)) {        // the search left the scene: the environment map, from the ADVANCED origin (volpath.cpp:421-426)
                float nearT, farT;
                if (bsphereIntersect(sc, o, d, nearT, farT) && !(nearT > 0) && !(farT < 0)) {
                    const v3 value = tr * envEval(sc, d);
                    if (!isZero(value)) {
                        const float4 tt = q.shT[shBase + i];
                        const float lumPdf = (bits & (1u << 28)) ? 0.0f : envLumPdf(sc, tb, d, q.shX[shBase + i].w);
                        const v3 li = (V(tt.x, tt.y, tt.z) * value) * miWeight(c.w, lumPdf);
                        float4 a = q.acc[pid]; a.x += li.x; a.y += li.y; a.z += li.z; q.acc[pid] = a;
                    }
                }
            }
            if (surface && h.emitter >= 0) {
                const v3 value = tr * emitterEval(tb, h.emitter, h.ns, -d);
                if (!isZero(value)) {
                    const float4 tt = q.shT[shBase + i], xx = q.shX[shBase + i];
                    const float lumPdf = (bits & (1u << 28)) ? 0.0f : pdfEmitterDirect<true>(sc, tb, h.emitter, V(xx.x, xx.y, xx.z), d, h.ns, t, (bits & (1u << 27)) != 0);
                    const v3 li = (V(tt.x, tt.y, tt.z) * value) * miWeight(c.w, lumPdf);
                    float4 a = q.acc[pid]; a.x += li.x; a.y += li.y; a.z += li.z; q.acc[pid] = a;
                }
            }
            continue;
        }
        // ---- kind 0: Scene::evalTransmittance (scene.cpp:650-713), as k_shadow_vol
        const bool p1OnSurface = (bits >> 24) & 1u, p2OnSurface = (bits >> 25) & 1u;
        const v3 p1 = V(so.x, so.y, so.z), p2 = V(sd.x, sd.y, sd.z);
        v3 d = p2 - p1; float remaining = sqrtf(dot(d, d)); { const float r = 1.0f / remaining; d = d * r; }
        const float lengthFactor = p2OnSurface ? (1 - MI_SHADOW_EPSILON) : 1;
        v3 o = p1; float rmint = p1OnSurface ? MI_EPSILON : 0.0f, rmaxt = remaining * lengthFactor;
        v3 tr = V(1, 1, 1); int interactions = 0; bool blocked = false;
        while (remaining > 0) {
            float mint, maxt, t = INFINITY, u = 0, v = 0; uint32_t prim = 0xFFFFFFFFu; int inst = -1; bool surface = false;
            ++shadowRays;
            if (clipInterval(sc, o, d, rmint, rmaxt, true, mint, maxt)) surface = traverse<false, 3, WIDE>(sc, o, d, mint, maxt, s_stk + tid, t, prim, u, v, inst);
            if (!surface) t = INFINITY;
            int material = -1; v3 nn = V(0, 0, 0);
            if (surface) {
                if (prim >= sc.n_tris) { const AnalyticD &a = sc.analytic[prim - sc.n_tris]; material = a.material; Hit h; fillHitAnalytic(a, o, d, t, u, v, h); nn = h.ng; }
                else {
                    AS<false>::p4 rec = tb.shade4 + prim * (uint32_t) MI_SHADE_WORDS; const f4 r0 = rec[0], r1 = rec[1], r2 = rec[2]; material = __float_as_int(r0.w);
                    v3 fn = cross(V(r1.x - r0.x, r1.y - r0.y, r1.z - r0.z), V(r2.x - r0.x, r2.y - r0.y, r2.z - r0.z)); const float len = sqrtf(dot(fn, fn));
                    if (!isZero(fn)) { const float r = 1.0f / len; fn = fn * r; }
                    nn = fn;
                }
                if (interactions == maxInteractions || !(NX ? surfaceHasNull(tb, loadMaterial(tb, material)) : materialHasNull(loadMaterial(tb, material).type))) { blocked = true; break; }
            }
            if (medium >= 0) tr = tr * mediumTransmittance(sc.media[medium], 0.0f, minf(t, remaining));
            if (!surface || isZero(tr)) break;
            tr = tr * (NX ? surfaceNullEval(sc, tb, loadMaterial(tb, material), o, d, t, prim, u, v, inst, -dot(d, nn), true) : materialNullEval(loadMaterial(tb, material), -dot(d, nn)));
            const uint32_t pm = sc.prim_media ? sc.prim_media[prim] : 0u;
            if (pm) { if (medium != targetMedium(pm, nn, -d)) { blocked = true; break; } medium = targetMedium(pm, nn, d); }
            if (++interactions > 100) break;
            o = o + d * t; remaining -= t; rmaxt = remaining * lengthFactor; rmint = MI_EPSILON;
        }
        if (bits & (1u << 29)) {                                                 // the alpha walk of a sensor ray (records.inl:128-130)
            const v3 tf = blocked ? V(0, 0, 0) : tr; float4 a = q.acc[pid]; a.w = 1 - ((0.0f + tf.x) + tf.y + tf.z) * (1.0f / 3); q.acc[pid] = a;
            continue;
        }
        if (blocked) continue;
        const float4 c = q.shC[shBase + i], tt = q.shT[shBase + i], xx = q.shX[shBase + i];
        const float r = 1.0f / c.w;
        const v3 value = V(c.x, c.y, c.z) * (tr * r);
        if (!isZero(value) && xx.w != 0.0f) {                                   // (phaseVal != 0 / !bsdfVal.isZero(), volpath.cpp:139, :232)
            const v3 li = ((V(tt.x, tt.y, tt.z) * value) * V(xx.x, xx.y, xx.z)) * xx.w;
            float4 a = q.acc[pid]; a.x += li.x; a.y += li.y; a.z += li.z; q.acc[pid] = a;
        }
    }
    }
    }
    for (int off = 32; off > 0; off >>= 1) { shadowRays += __shfl_down(shadowRays, off); normalRays += __shfl_down(normalRays, off); }
    if ((tid & 63u) == 0) { if (shadowRays) atomicAdd(&q.counters[1], shadowRays); if (normalRays) atomicAdd(&q.counters[0], normalRays); }
}

extern "C" {
void mi_launch_shade_volmis(const DScene &sc, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, size_t lds, hipStream_t st) {
    const bool env = sc.env_index >= 0;
    if (sc.has_adapters & 1u) { if (env) launchWithLds(k_shade_volmis<true, true, true>, grid, lds, st, sc, rc, q, buf); else launchWithLds(k_shade_volmis<true, false, true>, grid, lds, st, sc, rc, q, buf); }
    else if (sc.n_textures) { if (env) launchWithLds(k_shade_volmis<true, true, false>, grid, lds, st, sc, rc, q, buf); else launchWithLds(k_shade_volmis<true, false, false>, grid, lds, st, sc, rc, q, buf); }
    else { if (env) launchWithLds(k_shade_volmis<false, true, false>, grid, lds, st, sc, rc, q, buf); else launchWithLds(k_shade_volmis<false, false, false>, grid, lds, st, sc, rc, q, buf); }
}
void mi_launch_shadow_volmis(const DScene &sc, const Queues &q, uint32_t grid, hipStream_t st) {
    const bool env = sc.env_index >= 0;
#define MI_SHV(W, E) do { if (sc.has_adapters & 2u) hipLaunchKernelGGL((k_shadow_volmis<W, E, true>), dim3(grid), dim3(WG), 0, st, sc, q); else hipLaunchKernelGGL((k_shadow_volmis<W, E, false>), dim3(grid), dim3(WG), 0, st, sc, q); } while (0)
    if (sc.bvh_wide) { if (env) MI_SHV(true, true); else MI_SHV(true, false); }
    else { if (env) MI_SHV(false, true); else MI_SHV(false, false); }
#undef MI_SHV
}
}
