// shade.h -- the shading stage: one bounce of MIPathTracer::Li per launch (see kernels_common.h).  Instantiated by kernels_shade_*.hip, one translation
// unit per (RC, ENV) pair so that `make -j` compiles the variants side by side.
#pragma once
#include "kernels_common.h"
#ifndef MI_SHADE_MIN_WAVES
#define MI_SHADE_MIN_WAVES 1      // waves per SIMD asked of the compiler (A/B builds; DESIGN.md §3: a register cap only moves the live set into scratch)
#endif

// ---------------------------------------------------------------------------------------------- shade
// One bounce of MIPathTracer::Li (src/integrators/path/path.cpp:135-287) for every live path of the segment:
//   tail of the previous iteration (emitter hit by the BSDF ray -> MIS term :257-264, Russian roulette :276-286), then
//   emitted radiance :148-150, depth test :156-165, emitter sampling :172-200 (visibility deferred to the shadow queue),
//   BSDF sampling :207-226.  Survivors are compacted into the other ray/state buffer, shadow rays into the shadow queue.
// WRAP: the BSDF adapters mixturebsdf / bumpmap / normalmap are present (only with RC and AN): scenes without them keep the leaner kernels
template <bool RC, bool ENV, bool SMALL, bool AN, bool TEX, bool WRAP>   // TEX: textures bound to materials (implies AN); RC: rough conductors present; ENV: environment emitter present; SMALL: scene tables staged in LDS; AN ("extended"): analytic shapes or delta emitters (point / spot / directional) present
__global__ __launch_bounds__(WG, MI_SHADE_MIN_WAVES) void k_shade(DScene sc, RenderConst rc, Queues q, int buf) {
    extern __shared__ uint32_t s_dyn[];
    uint32_t *s_nib = s_dyn;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = buf ^ 1;
    const SobolTabLds m32{(lds_u32_ptr) s_nib, rc.nib_count, rc.sobol_scramble};   // always the LDS copy (ds_read lookups); unused by the independent stream
    const uint32_t nibWords = rc.sampler == 1 ? rc.nib_dims * rc.nib_count * 16u : 4u;
    if (rc.sampler == 1) {   // stage the Sobol' nibble tables in LDS (nib_dims x nib_count x 16 words)
        for (uint32_t i = tid; i < nibWords; i += WG) s_nib[i] = rc.sobol_nib[i];
    }
    // Scene tables: shading records, materials, emitters, CDFs.  Small scenes (<= 128 triangles ...) are staged in LDS: the shading
    // stage chases hit -> triangle record -> material -> emitter CDF -> light triangle, and each hop is an L2 round trip otherwise.
    Tabs<SMALL> tb;
    if (SMALL) {
        uint32_t *base = s_dyn + ((nibWords + 3u) & ~3u);
        const uint32_t wShade = sc.n_tris * (4u * MI_SHADE_WORDS), wMat = sc.n_materials * 16u, wEm = sc.n_emitters * 12u, wEc = (sc.n_emitters + 1u + 3u) & ~3u, wAc = sc.area_cdf_len;
        uint32_t *pS = base, *pM = pS + wShade, *pE = pM + wMat, *pEc = pE + wEm, *pAc = pEc + wEc;
        const uint32_t *gS = (const uint32_t *) sc.shade, *gM = (const uint32_t *) sc.materials, *gE = (const uint32_t *) sc.emitters, *gEc = (const uint32_t *) sc.emitter_cdf, *gAc = (const uint32_t *) sc.area_cdf;
        for (uint32_t i = tid; i < wShade; i += WG) pS[i] = gS[i];
        for (uint32_t i = tid; i < wMat; i += WG) pM[i] = gM[i];
        for (uint32_t i = tid; i < wEm; i += WG) pE[i] = gE[i];
        for (uint32_t i = tid; i < sc.n_emitters + 1u; i += WG) pEc[i] = gEc[i];
        for (uint32_t i = tid; i < wAc; i += WG) pAc[i] = gAc[i];
        tb.shade4 = (typename AS<SMALL>::p4) pS; tb.materials4 = (typename AS<SMALL>::p4) pM; tb.emitters4 = (typename AS<SMALL>::p4) pE;
        tb.emitter_cdf = (typename AS<SMALL>::pf) pEc; tb.area_cdf = (typename AS<SMALL>::pf) pAc;
    } else {
        tb.shade4 = (typename AS<SMALL>::p4) sc.shade; tb.materials4 = (typename AS<SMALL>::p4) sc.materials; tb.emitters4 = (typename AS<SMALL>::p4) sc.emitters;
        tb.emitter_cdf = (typename AS<SMALL>::pf) sc.emitter_cdf; tb.area_cdf = (typename AS<SMALL>::pf) sc.area_cdf;
    }
    // Material-sorted shading (scenes that mix BSDF classes): the paths of a segment are first ordered by the class of the surface
    // they hit -- diffuse-like from the front, rough conductors from the back of an LDS index list (wave64 ballots, order preserving) --
    // so a wave runs either the cheap diffuse code or the microfacet code, not both.  The queues are then read through that list.
    // In this stage a SEGMENT IS OWNED BY ONE WAVE: compaction is a pair of wave64 ballots with the running output offsets kept in
    // (uniform) registers -- no LDS exchange and no workgroup barrier anywhere in the loop; the four waves of a workgroup only share the
    // LDS copies of the tables above.
    uint16_t *s_order = reinterpret_cast<uint16_t *>(s_dyn + rc.order_offset_words) + (size_t) wave * q.cap;
    const bool doSort = RC && rc.order_offset_words != 0 && q.cap <= 0xFFFFu;
    unsigned long long pathLen = 0, shadowRays = 0;
    __syncthreads();                                         // tables staged
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t seg = blockIdx.x * (WG / 64) + wave; seg < q.n_seg; seg += gridDim.x * (WG / 64)) {
    const uint32_t n = q.count[buf][seg];
    const uint32_t segBase = seg * q.cap;                    // (a pool has fewer than 2^28 slots: queues.h qat)
    uint32_t outA = 0, outS = 0;                             // survivors / shadow records written so far (uniform)
    uint32_t done0 = 0, done1 = 0;                           // uniform running counts (front / back)
    if (doSort) {
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t i = base + lane; int cls = 2;
            if (i < n) {
                const uint32_t prim = __float_as_uint(qat(q.hit, segBase + i).w);
                if (prim == 0xFFFFFFFFu) cls = 0;
                else if (AN && prim >= sc.n_tris) cls = (sc.analytic[prim - sc.n_tris].flags & 8u) ? 1 : 0;
                else cls = (__float_as_uint(tb.shade4[prim * (uint32_t) MI_SHADE_WORDS + 2u].w) & 8u) ? 1 : 0;
            }
            const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1);
            if (cls == 0) s_order[done0 + (uint32_t) __popcll(m0 & lt)] = (uint16_t) i;
            else if (cls == 1) s_order[n - 1u - (done1 + (uint32_t) __popcll(m1 & lt))] = (uint16_t) i;
            done0 += (uint32_t) __popcll(m0); done1 += (uint32_t) __popcll(m1);
        }
    }
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        // Two phases per chunk, each closed by its own wave64 ballot, so that the 12 registers of a shadow record are written out before
        // the BSDF-sampling code runs (register budget -> one more resident wave per SIMD):
        //   A: tail of the previous bounce, emitted radiance, emitter sampling  -> shadow queue
        //   B: BSDF sampling                                                     -> next ray / state
        bool alive = false, wantShadow = false, toSample = false;
        float4 shO, shD, shC;
        Hit h; MaterialD bsdf; SamplerState ss; v3 T = V(0, 0, 0); float eta = 1.0f; uint32_t pid = 0; int depth = 0; bool unscattered = false; v3 opac = V(1, 1, 1); bool masked = false; bool bumped = false; v3 bps = V(0, 0, 0), bpt = bps, bpn = bps; int coat = -1;   // bumpmap / normalmap: perturbed frame (RC variants)
          // mask wrapper (mask.cpp): opacity in front of `bsdf` (RC variants)
        if (i < n) {
            const uint32_t slot = segBase + (doSort ? (uint32_t) s_order[i] : i);
            float4 rd = qat(q.rayD[buf], slot), hr = qat(q.hit, slot); uint4 s0 = qat(q.st0[buf], slot); float4 s1 = qat(q.st1[buf], slot);
            float prevPdf = qat(q.st2[buf], slot);
            pid = s0.x; ss.a = s0.y; ss.b = s0.z; ss.dim = s0.w & 0xFFu;
            depth = (int) ((s0.w >> 8) & 0xFFu); const bool facingRef = ((s0.w >> 16) & 1u) != 0;
            const bool prevDelta = RC && ((s0.w >> 17) & 1u) != 0;      // the BSDF sample that spawned this ray was a delta component -> lumPdf = 0 (path.cpp:259-260)
            unscattered = RC && ((s0.w >> 18) & 1u) != 0;    // every component sampled so far was ENull (thin dielectric panes): `scattered` is still false (path.cpp:213)
            v3 d = V(rd.x, rd.y, rd.z); T = V(s1.x, s1.y, s1.z); eta = s1.w;
            const uint32_t prim = __float_as_uint(hr.w);
            v3 add = V(0, 0, 0); bool haveAdd = false;
            do {
                if (prim == 0xFFFFFFFFu) {                                     // miss: path.cpp:136-143 / :234-248
                    pathLen += (unsigned) (depth > 1 ? depth - 1 : 1);
                    // the pixel's accumulator is read HERE, next to the ray origin, so that its round trip to HBM overlaps the environment lookup instead of following it
                    const bool clearAlpha = depth == 1 && rc.opacity;                                                 // records.inl:121-137: alpha = 0 on a camera-ray miss
                    float4 accum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (ENV || clearAlpha) accum = qat(q.acc, pid);
                    if (clearAlpha) accum.w = 0.0f;
                    if (ENV) {
                        if (depth == 1) { if (!rc.hide_emitters && !sc.env_texture) { add = T * envEval(sc, d); haveAdd = true; } }     // path.cpp:139-141; with a MIP pyramid k_env_primary has added the filtered lookup
                        else {
                            // BSDF ray left the scene: env->evalEnvironment + fillDirectSamplingRecord (envmap.cpp:362-378), MIS term path.cpp:257-264
                            // (the ray origin is only needed for the bounding-sphere test: it is requested first and consumed after the environment lookup, whose
                            // atan2f / acosf and texel fetches then cover its round trip to HBM)
                            const float4 ro = qat(q.rayO[buf], slot); float nearT, farT;
                            if (!(rc.hide_emitters && unscattered)) {   // path.cpp:238-239: hideEmitters && !scattered
                                v3 value; float pdfSA;
                                if (sc.env_constant) {      // ConstantBackgroundEmitter::pdfDirect (constant.cpp:219-233): needs the reference normal of the previous vertex
                                    value = envEval(sc, d);
                                    const float c = qat(q.st3[buf], slot); pdfSA = c != 2.0f ? MI_INV_PI * maxf(0.0f, c) : MI_INV_FOURPI;
                                } else envEvalAndPdf(sc, d, value, pdfSA);      // one atan2f / acosf and one set of texels for both
                                if (bsphereIntersect(sc, V(ro.x, ro.y, ro.z), d, nearT, farT) && !(nearT > 0) && !(farT < 0)) {
                                float lumPdf = prevDelta ? 0.0f : pdfSA * (loadEmitter(tb, sc.env_index).weight * sc.emitter_norm);
                                add = (T * value) * miWeight(prevPdf, lumPdf); haveAdd = true;
                                }
                            }
                        }
                    }
                    if (haveAdd) { accum.x += add.x; accum.y += add.y; accum.z += add.z; }
                    if (haveAdd || clearAlpha) qat(q.acc, pid) = accum;
                    haveAdd = false;
                    break;
                }
                v3 ro3 = V(0, 0, 0);
                if (AN) { float4 ro = qat(q.rayO[buf], slot); ro3 = V(ro.x, ro.y, ro.z); }      // analytic shapes: hit point = o + t d; sphere lights: reference point of pdfDirect
                const int inst = (AN && q.hitInst) ? qat(q.hitInst, slot) : -1;
                if (AN && inst >= 0) fillHitInstanced(sc, tb, sc.instances[inst], ro3, d, hr.x, prim, hr.y, hr.z, h);
                else if (AN && prim >= sc.n_tris) fillHitAnalytic(sc.analytic[prim - sc.n_tris], ro3, d, hr.x, hr.y, hr.z, h);
                else fillHit<SMALL, AN>(sc, tb, d, hr.x, prim, hr.y, hr.z, h);
                if (depth > 1) {
                    if (h.emitter >= 0) {                                    // path.cpp:229-233, 257-264
                        v3 value = emitterEval(tb, h.emitter, h.ns, -d);
                        float lumPdf = prevDelta ? 0.0f : pdfEmitterDirect<AN>(sc, tb, h.emitter, ro3, d, h.ns, h.dist, facingRef);
                        add = (T * value) * miWeight(prevPdf, lumPdf); haveAdd = true;
                    }
                    const int prevDepth = depth - 1;                         // rRec.depth++ >= m_rrDepth (path.cpp:276)
                    if (prevDepth >= rc.rr_depth) {
                        float qq = minf(maxf(maxf(T.x, T.y), T.z) * eta * eta, 0.95f);
                        if (next1D(ss, rc.sampler, m32) >= qq) { pathLen += (unsigned) depth; break; }
                        float r = 1.0f / qq; T = T * r;
                    }
                }
                if (!(depth <= rc.max_depth || rc.max_depth < 0)) { pathLen += (unsigned) depth; break; }   // loop guard path.cpp:135
                int curMat = h.material; bsdf = loadMaterial(tb, curMat);
                auto applyTexture = [&](MaterialD &mm) {
                if (TEX) {                                                   // a textured parameter: m_reflectance->eval(bRec.its) (diffuse.cpp:112-121) and its siblings
                    const uint32_t tex = (mm.type == MI_BSDF_T_BUMPMAP || mm.type == MI_BSDF_T_NORMALMAP) ? 0u : (mm.flags >> 8) & 0xFFFFu;      // (an adapter's texture is its displacement / normal map)
                    if (tex) {
                        const TextureD &tx = sc.textures[tex - 1]; v3 c;
                        float huvx = h.uvx, huvy = h.uvy;
                        const bool onAnalytic = AN && inst < 0 && prim >= sc.n_tris;         // an analytic shape: its own parameterisation, evaluated on demand
                        if (onAnalytic) { v3 du, dv; analyticUV(sc.analytic[prim - sc.n_tris], hr.y, hr.z, ro3 + d * hr.x, huvx, huvy, du, dv); }   // (tangents: recomputed below where a filtered lookup needs them)
                        if (tx.type == 2u) {                                 // BitmapTexture::eval (src/textures/bitmap.cpp:434-502) under Texture2D::eval (texture.cpp:112-121)
                            const float uvx = huvx * tx.uscale + tx.uoffset, uvy = huvy * tx.vscale + tx.voffset;
                            if (depth == 1) {                                // its.getBSDF(ray) -> computePartials: only the camera ray carries differentials (records.inl:68-75)
                                v3 dpdu, dpdv;
                                if (onAnalytic) { float tu_, tv_; analyticUV(sc.analytic[prim - sc.n_tris], hr.y, hr.z, ro3 + d * hr.x, tu_, tv_, dpdu, dpdv); }
                                else if (h.flags & 16u) { const TriUV &tu = sc.triuv[prim]; dpdu = ld3(tu.dpdu); dpdv = ld3(tu.dpdv); }
                                else { typename AS<SMALL>::p4 rec = tb.shade4 + prim * (uint32_t) MI_SHADE_WORDS; f4 r0 = rec[0], r1 = rec[1], r2 = rec[2]; dpdu = V(r1.x - r0.x, r1.y - r0.y, r1.z - r0.z); dpdv = V(r2.x - r0.x, r2.y - r0.y, r2.z - r0.z); }
                                if (inst >= 0) { dpdu = xfVector(sc.instances[inst].to_world, dpdu); dpdv = xfVector(sc.instances[inst].to_world, dpdv); }
                                const float2 sp = qat(q.pos, pid); v3 rxd, ryd; cameraDifferentials(sc, rc.inv_sqrt_spp, sp.x, sp.y, d, rxd, ryd);
                                float pa[4]; computePartials(h.p, h.ng, dpdu, dpdv, ro3, rxd, ryd, pa);
                                c = mipEval(sc, tx, uvx, uvy, pa[0] * tx.uscale, pa[1] * tx.vscale, pa[2] * tx.uscale, pa[3] * tx.vscale);
                            } else c = tx.filter != 0u ? mipBilinear(sc, tx, 0, uvx, uvy) : mipBox(sc, tx, 0, uvx, uvy);
                        } else c = textureEval(tx, huvx, huvy);
                        mm.reflectance[0] = c.x; mm.reflectance[1] = c.y; mm.reflectance[2] = c.z;
                    }
                }
                };
                applyTexture(bsdf);
                if (RC && bsdf.type == MI_BSDF_T_MASK) {                     // mask.cpp: this record's (textured) `reflectance` is the opacity in front of the nested record `distr`
                    opac = ld3(bsdf.reflectance); masked = true; curMat = (int) bsdf.distr; bsdf = loadMaterial(tb, curMat); applyTexture(bsdf);
                }
                if (WRAP && TEX && (bsdf.type == MI_BSDF_T_BUMPMAP || bsdf.type == MI_BSDF_T_NORMALMAP)) {      // bumpmap.cpp / normalmap.cpp: getFrame(its), then the nested record
                    float huvx = h.uvx, huvy = h.uvy; v3 dpdu, dpdv;
                    if (AN && inst < 0 && prim >= sc.n_tris) analyticUV(sc.analytic[prim - sc.n_tris], hr.y, hr.z, ro3 + d * hr.x, huvx, huvy, dpdu, dpdv);
                    else if (h.flags & 16u) { const TriUV &tu = sc.triuv[prim]; dpdu = ld3(tu.dpdu); dpdv = ld3(tu.dpdv); }
                    else { typename AS<SMALL>::p4 rec = tb.shade4 + prim * (uint32_t) MI_SHADE_WORDS; f4 r0 = rec[0], r1 = rec[1], r2 = rec[2]; dpdu = V(r1.x - r0.x, r1.y - r0.y, r1.z - r0.z); dpdv = V(r2.x - r0.x, r2.y - r0.y, r2.z - r0.z); }
                    if (inst >= 0) { dpdu = xfVector(sc.instances[inst].to_world, dpdu); dpdv = xfVector(sc.instances[inst].to_world, dpdv); }
                    perturbFrame(sc, bsdf, h, huvx, huvy, dpdu, dpdv, bps, bpt, bpn); bumped = true;
                    curMat = (int) bsdf.distr; bsdf = loadMaterial(tb, curMat); applyTexture(bsdf);
                }
                if (WRAP && (bsdf.type == MI_BSDF_T_COATING || bsdf.type == MI_BSDF_T_ROUGHCOATING)) { coat = curMat; bsdf = loadMaterial(tb, (int) bsdf.distr); applyTexture(bsdf); }      // coating.cpp: the layer around the nested record
                if (depth == 1 && h.emitter >= 0 && !rc.hide_emitters) {    // path.cpp:148-150 (EEmittedRadiance only on the camera segment)
                    add = T * emitterEval(tb, h.emitter, h.ns, -d); haveAdd = true;
                }
                if ((depth >= rc.max_depth && rc.max_depth > 0) || (rc.strict_normals && dot(d, h.ng) * h.wi.z >= 0)) { pathLen += (unsigned) depth; break; }
                // emitter sampling (path.cpp:172-200)
                v3 refN = (h.flags & 2u) ? V(0, 0, 0) : h.ns;               // records.inl:160-164
                if (!(h.flags & 4u)) {                                       // bsdf->getType() & BSDF::ESmooth
                    float sx, sy; next2D(ss, rc.sampler, m32, sx, sy);
                    Direct dr; v3 value = sampleEmitterDirect<ENV, AN>(sc, tb, h.p, refN, sx, sy, dr);
                    if (dr.pdf != 0) {
                        ++shadowRays;                                        // scene.cpp:871-875: a shadow ray is cast whenever pdf != 0
                        v3 wo = toLocal(h, dr.d);
                        v3 wiQ = h.wi, woQ = wo; bool rejected = false;
                        if (WRAP && bumped) {                                                // bumpmap.cpp:165-180: the query in the perturbed frame
                            wiQ = frameToLocal(bps, bpt, bpn, toWorld(h, h.wi)); woQ = frameToLocal(bps, bpt, bpn, toWorld(h, wo)); rejected = wo.z * woQ.z <= 0;
                        }
                        v3 bsdfVal = rejected ? V(0, 0, 0) : ctEval<RC, WRAP>(sc, tb, coat, bsdf, wiQ, woQ);
                        if (RC && masked) bsdfVal = bsdfVal * opac;                          // mask.cpp:124-127
                        if (!isZero(value) && !isZero(bsdfVal) && (!rc.strict_normals || dot(h.ng, dr.d) * wo.z > 0)) {
                            float bp = (dr.delta || rejected) ? 0.0f : ctPdf<RC, WRAP>(sc, tb, coat, bsdf, wiQ, woQ);     // emitter->isOnSurface() && measure == ESolidAngle (path.cpp:191-192)
                            if (RC && masked) bp *= luminance3(opac);                          // mask.cpp:141-146
                            float weight = miWeight(dr.pdf, bp);
                            v3 c = ((T * value) * bsdfVal) * weight;
                            wantShadow = true;
                            shO = make_float4(h.p.x, h.p.y, h.p.z, dr.dist * (1 - MI_SHADOW_EPSILON));
                            shD = make_float4(dr.d.x, dr.d.y, dr.d.z, __uint_as_float(pid));
                            shC = make_float4(c.x, c.y, c.z, 0.0f);
                        }
                    }
                }
                toSample = true;
            } while (false);
            if (haveAdd) { float4 a = qat(q.acc, pid); a.x += add.x; a.y += add.y; a.z += add.z; qat(q.acc, pid) = a; }
        }
        // wave64 ballots: order-preserving compaction inside the (wave-owned) segment
        const unsigned long long mS = __ballot(wantShadow);
        if (wantShadow) {
            const uint32_t o = segBase + outS + (uint32_t) __popcll(mS & lt);
            qat(q.shO, o) = shO; qat(q.shD, o) = shD; qat(q.shC, o) = shC;
        }
        outS += (uint32_t) __popcll(mS);
        float4 nrO, nrD, nS1; uint4 nS0; float nS2 = 0, nS3 = 0;
        if (toSample) {
            // BSDF sampling (path.cpp:207-226)
            float bPdf = 0, bEta = 1; v3 woL = V(0, 0, 0);
            float sx, sy; next2D(ss, rc.sampler, m32, sx, sy);
            bool sampledDelta, sampledNull; float extra = 0.0f; v3 bw;
            bool passThrough = false;                                                   // mask.cpp:196-208: the nested BSDF with probability luminance(opacity), else straight through
            if (RC && masked) { const float prob = luminance3(opac); if (sx < prob) sx /= prob; else passThrough = true; }
            if (RC && passThrough) {
                const float p = 1 - luminance3(opac);
                woL = V(-h.wi.x, -h.wi.y, -h.wi.z); bEta = 1.0f; bPdf = p; sampledDelta = true; sampledNull = true;
                bw = V((1.0f - opac.x) / p, (1.0f - opac.y) / p, (1.0f - opac.z) / p);
            } else {
                auto drawExtra = [&]() { return next1D(ss, rc.sampler, m32); };        // bRec.sampler->next1D() inside BSDF::sample (EUsesSampler), only if the sampled BSDF asks
                (void) extra;
                if (WRAP && bumped) {                                                  // bumpmap.cpp:199-222
                    const v3 wiQ = frameToLocal(bps, bpt, bpn, toWorld(h, h.wi)); v3 woQ = V(0, 0, 0);
                    bw = ctSample<RC, WRAP>(sc, tb, coat, bsdf, wiQ, sx, sy, drawExtra, woQ, bPdf, bEta, sampledDelta, sampledNull);
                    if (!isZero(bw)) { woL = toLocal(h, frameToWorld(bps, bpt, bpn, woQ)); if (woL.z * woQ.z <= 0) bw = V(0, 0, 0); }
                } else bw = ctSample<RC, WRAP>(sc, tb, coat, bsdf, h.wi, sx, sy, drawExtra, woL, bPdf, bEta, sampledDelta, sampledNull);
                if (RC && masked) { const float prob = luminance3(opac); bw = V(bw.x * opac.x / prob, bw.y * opac.y / prob, bw.z * opac.z / prob); bPdf *= prob; }
            }
            v3 wo = toWorld(h, woL);
            if (isZero(bw) || (rc.strict_normals && dot(h.ng, wo) * woL.z <= 0)) pathLen += (unsigned) depth;
            else {
                T = T * bw; eta *= bEta;
                alive = true;
                nrO = make_float4(h.p.x, h.p.y, h.p.z, MI_EPSILON);
                nrD = make_float4(wo.x, wo.y, wo.z, INFINITY);
                v3 refN = (h.flags & 2u) ? V(0, 0, 0) : h.ns;           // records.inl:160-164
                const float cosRef = dot(wo, refN);
                uint32_t fl = cosRef >= 0 ? 1u : 0u;
                nS3 = (h.flags & 2u) ? 2.0f : cosRef;
                nS0 = make_uint4(pid, ss.a, ss.b, (ss.dim & 0xFFu) | ((uint32_t) (depth + 1) << 8) | (fl << 16) | ((RC && sampledDelta) ? (1u << 17) : 0u) | ((RC && sampledNull && (depth == 1 || unscattered)) ? (1u << 18) : 0u));
                nS1 = make_float4(T.x, T.y, T.z, eta); nS2 = bPdf;
            }
        }
        const unsigned long long mA = __ballot(alive);
        if (alive) {
            const uint32_t o = segBase + outA + (uint32_t) __popcll(mA & lt);
            qat(q.rayO[nb], o) = nrO; qat(q.rayD[nb], o) = nrD; qat(q.st0[nb], o) = nS0; qat(q.st1[nb], o) = nS1; qat(q.st2[nb], o) = nS2;
            if (ENV && sc.env_constant) qat(q.st3[nb], o) = nS3;
        }
        outA += (uint32_t) __popcll(mA);
    }
    if (lane == 0) { q.count[nb][seg] = outA; q.shCount[seg] = outS; }
    }
    // counters: wave reduction, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) { pathLen += __shfl_down(pathLen, off); shadowRays += __shfl_down(shadowRays, off); }
    if (lane == 0) { if (pathLen) atomicAdd(&q.counters[2], pathLen); if (shadowRays) atomicAdd(&q.counters[1], shadowRays); }
}



// launch the (AN, TEX) variant of one (RC, ENV, WRAP, SMALL) combination that fits the scene; WRAP variants exist for AN = true only
template <bool RC, bool ENV, bool WRAP, bool SM>
static void launchShadeVariantSM(const DScene &sc, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, size_t lds, hipStream_t st) {
    if ((sc.ext || WRAP) && sc.n_textures) launchWithLds(k_shade<RC, ENV, SM, true, true, WRAP>, grid, lds, st, sc, rc, q, buf);
    else if (sc.ext || WRAP) launchWithLds(k_shade<RC, ENV, SM, true, false, WRAP>, grid, lds, st, sc, rc, q, buf);
    else launchWithLds(k_shade<RC, ENV, SM, WRAP, false, WRAP>, grid, lds, st, sc, rc, q, buf);
}
// both SMALL halves from one translation unit (the WRAP combinations compile theirs in two: kernels_shade_rcw*_small.hip, for the build time)
template <bool RC, bool ENV, bool WRAP>
static void launchShadeVariant(const DScene &sc, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, size_t lds, hipStream_t st) {
    if (sc.small_tables != 0) launchShadeVariantSM<RC, ENV, WRAP, true>(sc, rc, q, buf, grid, lds, st);
    else launchShadeVariantSM<RC, ENV, WRAP, false>(sc, rc, q, buf, grid, lds, st);
}
