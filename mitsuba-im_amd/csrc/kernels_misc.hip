// kernels_misc.hip -- generate, filtered environment lookups of camera rays, film accumulation / read-back, unit-level entry points (see kernels_common.h)
#include "kernels_common.h"

// ---------------------------------------------------------------------------------------------- generate
// One camera sample per path: src/librender/integrator.cpp:166-181 (pixel offset + sensor ray), sampler set-up
// src/samplers/sobol.cpp:171-216.  Path q of the batch = (plane q / npix, tile pixel q % npix).
__global__ __launch_bounds__(WG) void k_generate(DScene sc, RenderConst rc, Queues q, BatchDesc bd) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t seg = blockIdx.x; seg < q.n_seg; seg += gridDim.x) {
    const uint64_t segBase = (uint64_t) seg * q.cap;
    uint64_t remaining = bd.n_paths > segBase ? bd.n_paths - segBase : 0;
    const uint32_t n = remaining > q.cap ? q.cap : (uint32_t) remaining;
    const uint32_t tw = bd.tile.x1 - bd.tile.x0;
    for (uint32_t i = tid; i < n; i += WG) {
        const uint64_t pid = segBase + i;
        uint32_t plane = (uint32_t) (pid / bd.n_pix), pl = (uint32_t) (pid % bd.n_pix);
        uint32_t px = bd.tile.x0 + pl % tw, py = bd.tile.y0 + (pl / tw) * bd.row_stride, sidx = bd.sample_begin + plane;
        if (bd.list) { px = bd.list[pid * 3]; py = bd.list[pid * 3 + 1]; sidx = bd.list[pid * 3 + 2]; }
        SamplerState ss; float jx, jy;
        if (rc.sampler == 1) {
            uint32_t r0, r1;
            if (rc.sobol_frame && sidx < rc.sobol_nframes) {      // look_up + dimensions 0 / 1 through the XOR-linear tables (RenderConst; bit-identical to sobolLookUp + sampleSingle)
                const uint32_t scr = rc.sobol_scramble >> (32u - sc.log_res);
                const uint4 a = rc.sobol_frame[sidx], b = rc.sobol_px[px ^ scr], c = rc.sobol_py[py ^ scr];
                ss.a = a.x ^ b.x ^ c.x; ss.b = a.y ^ b.y ^ c.y; r0 = rc.sobol_scramble ^ a.z ^ b.z ^ c.z; r1 = rc.sobol_scramble ^ a.w ^ b.w ^ c.w;
                jx = minf((float) r0 * (1.0f / 4294967296.0f), MI_ONE_MINUS_EPS); jy = minf((float) r1 * (1.0f / 4294967296.0f), MI_ONE_MINUS_EPS);
            } else {
                const uint64_t idx = sc.log_res > 1 ? sobolLookUp(sc.sobol_vdc, sc.sobol_vdc_inv, sc.log_res, sidx, px, py, rc.sobol_scramble) : (uint64_t) sidx;
                ss.a = (uint32_t) idx; ss.b = (uint32_t) (idx >> 32);
                const SobolTab gt{rc.sobol_nib, rc.nib_count, rc.sobol_scramble};
                jx = sobolSampleNib(gt, ss.a, ss.b, 0); jy = sobolSampleNib(gt, ss.a, ss.b, 1);
            }
            ss.dim = 2;
            if (ss.a != sidx || ss.b != 0u) {      // sobol.cpp:239-245: rescale the first two dimensions to a pixel-relative offset
                jx = jx * sc.resolution - (float) (int) px; jy = jy * sc.resolution - (float) (int) py;
            }
        } else {
            ss.a = (py * sc.width + px) ^ rc.seed_mix; ss.b = sidx; ss.dim = 0;
            next2D(ss, 0, SobolTab{nullptr, 0, 0}, jx, jy);
        }
        float sx = (float) (int) px + jx, sy = (float) (int) py + jy;
        v3 o, d; float mint, maxt; cameraRay(sc, sx, sy, o, d, mint, maxt);
        const uint64_t slot = segBase + i;
        q.rayO[0][slot] = make_float4(o.x, o.y, o.z, mint);
        q.rayD[0][slot] = make_float4(d.x, d.y, d.z, maxt);
        // packed: dim | depth << 8 | flags << 16   (flags bit0: facingRef of the previous vertex)
        q.st0[0][slot] = make_uint4((uint32_t) pid, ss.a, ss.b, ss.dim | (1u << 8) | rc.state_init);
        q.st1[0][slot] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);      // throughput rgb, eta
        q.st2[0][slot] = 0.0f;                                      // bsdfPdf of the segment that produced this ray
        q.pos[pid] = make_float2(sx, sy);
        q.acc[pid] = make_float4(0, 0, 0, 1.0f);        // Li rgb, alpha (newQuery: alpha = 1, integrator.h:223-229)
    }
    if (tid == 0) q.count[0][seg] = n;
    }
}

// Camera rays that leave the scene: EnvironmentMap::evalEnvironment WITH ray differentials (src/emitters/envmap.cpp:384-416) -- texture-space partials
// of the sensor ray's rx / ry directions, then TMIPMap::eval (EWA, anisotropy <= 10; u repeats, v clamps) over the map's MIP pyramid (input data).
// Runs once per batch between the first extend and the first shade; throughput is 1 at depth 1.
__global__ __launch_bounds__(WG) void k_env_primary(DScene sc, RenderConst rc, Queues q, int buf) {
    const TextureD tx = sc.textures[sc.env_texture - 1u];
    for (uint32_t seg = blockIdx.x; seg < q.n_seg; seg += gridDim.x) {
        const uint32_t n = q.count[buf][seg]; const uint64_t segBase = (uint64_t) seg * q.cap;
        for (uint32_t i = threadIdx.x; i < n; i += WG) {
            if (__float_as_uint(q.hit[segBase + i].w) != 0xFFFFFFFFu) continue;
            const float4 rd = q.rayD[buf][segBase + i]; const uint32_t pid = q.st0[buf][segBase + i].x;
            const v3 d = V(rd.x, rd.y, rd.z); const float2 sp = q.pos[pid];
            v3 rxd, ryd; cameraDifferentials(sc, rc.inv_sqrt_spp, sp.x, sp.y, d, rxd, ryd);
            const v3 v = mat3(sc.env_to_local, d);
            const float uvx = atan2f(v.x, -v.z) * MI_INV_TWOPI, uvy = acosf(minf(1.0f, maxf(-1.0f, v.y))) * MI_INV_PI;
            const v3 dvdx = mat3(sc.env_to_local, rxd) - v, dvdy = mat3(sc.env_to_local, ryd) - v;
            const float t1 = MI_INV_TWOPI / (v.x * v.x + v.z * v.z), t2 = -MI_INV_PI / maxf(sqrtf(maxf(0.0f, 1.0f - v.y * v.y)), MI_EPSILON);
            const v3 value = mipEval(sc, tx, uvx, uvy, t1 * (dvdx.z * v.x - dvdx.x * v.z), t2 * dvdx.y, t1 * (dvdy.z * v.x - dvdy.x * v.z), t2 * dvdy.y) * sc.env_scale;
            float4 a = q.acc[pid]; a.x += value.x; a.y += value.y; a.z += value.z; q.acc[pid] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------- film
// ImageBlock::put (include/mitsuba/render/imageblock.h:161-221) of every sample of the batch, 5 channels (R,G,B,alpha,weight),
// film planes are SoA.  One thread per tile pixel walks its planes in sample order.  The part of a footprint that lands on
// the thread's own pixel is added to `film` with plain loads/stores, one sample after the other -- the same order of float
// additions as the reference's per-pixel loop, independent of batch size and tiling.  Anything that spills into another pixel
// (box filter: only samples within 1e-5 of a pixel edge; wider filters: most of the footprint) goes to the separate `spill`
// planes with float atomics; read-back returns film + spill.
__global__ __launch_bounds__(WG) void k_film(DScene sc, Queues q, BatchDesc bd, float *film, float *spill) {
    const uint32_t pl = blockIdx.x * WG + threadIdx.x;
    if (pl >= bd.n_pix) return;
    const uint32_t tw = bd.tile.x1 - bd.tile.x0;
    const int px = (int) (bd.tile.x0 + pl % tw), py = (int) (bd.tile.y0 + (pl / tw) * bd.row_stride);
    const int W = (int) sc.width + 2 * sc.border, H = (int) sc.height + 2 * sc.border;
    const size_t plane = (size_t) W * H;
    const int ownX = px + sc.border, ownY = py + sc.border;
    const size_t ownIdx = (size_t) ownY * W + ownX;
    float own[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) own[k] = film[k * plane + ownIdx];
    const float r = sc.filter_radius;
    for (uint32_t s = 0; s < bd.n_planes; ++s) {
        const uint64_t pid = (uint64_t) s * bd.n_pix + pl;
        float4 li = q.acc[pid]; float2 sp = q.pos[pid];
        float vals[5] = {li.x, li.y, li.z, li.w, 1.0f};
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 5; ++k) bad |= (!isfinite(vals[k]) || vals[k] < 0);
        if (bad) continue;
        float posx = sp.x - 0.5f - (float) (0 - sc.border), posy = sp.y - 0.5f - (float) (0 - sc.border);
        int minx = (int) ceilf(posx - r), miny = (int) ceilf(posy - r), maxx = (int) floorf(posx + r), maxy = (int) floorf(posy + r);
        minx = max(minx, 0); miny = max(miny, 0); maxx = min(maxx, W - 1); maxy = min(maxy, H - 1);
        for (int y = miny; y <= maxy; ++y) {
            float wy = filterEvalDiscretized(sc, (float) y - posy);
            for (int x = minx; x <= maxx; ++x) {
                float w = filterEvalDiscretized(sc, (float) x - posx) * wy;
                if (x == ownX && y == ownY) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) own[k] += w * vals[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 5; ++k) atomicAdd(&spill[k * plane + (size_t) y * W + x], w * vals[k]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) film[k * plane + ownIdx] = own[k];
}

// merge of two replicas' films (multi-device renders: mi_render_merge_film): plain sums, element by element
__global__ void k_film_add(float *dst, const float *src, size_t n) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// film read-back helpers: SoA planes -> interleaved layouts of mi_render_read_film
__global__ void k_film_layout(const float *film, const float *spill, float *out, int W, int H, int border, int layout) {
    const size_t plane = (size_t) W * H;
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (layout == 0) { if (i < plane) for (int k = 0; k < 5; ++k) out[i * 5 + k] = film[k * plane + i] + spill[k * plane + i]; }
    else if (layout == 1) { if (i < plane) for (int k = 0; k < 4; ++k) out[i * 4 + k] = film[k * plane + i] + spill[k * plane + i]; }
    else {
        const int w = W - 2 * border, h = H - 2 * border;
        if (i < (size_t) w * h) {
            const int x = (int) (i % w), y = (int) (i / w); const size_t src = (size_t) (y + border) * W + (x + border);
            const float wgt = film[4 * plane + src] + spill[4 * plane + src], inv = wgt != 0 ? 1.0f / wgt : 0.0f;
            for (int k = 0; k < 3; ++k) out[i * 3 + k] = (film[k * plane + src] + spill[k * plane + src]) * inv;
        }
    }
}

// ---------------------------------------------------------------------------------------------- parity / unit kernels
__global__ void k_gather_samples(Queues q, const uint32_t *slots, uint64_t n, float *out) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float4 a = q.acc[slots[i]]; out[i * 3] = a.x; out[i * 3 + 1] = a.y; out[i * 3 + 2] = a.z; }
}
__global__ void k_debug_sobol(DScene sc, const uint32_t *in, uint64_t n, uint32_t ndims, unsigned long long *outIdx, float *outVals) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t idx = sc.log_res > 1 ? sobolLookUp(sc.sobol_vdc, sc.sobol_vdc_inv, sc.log_res, in[i * 3 + 2], in[i * 3], in[i * 3 + 1]) : in[i * 3 + 2];
    outIdx[i] = idx;
    for (uint32_t dmn = 0; dmn < ndims; ++dmn) outVals[i * ndims + dmn] = sobolSample(sc.sobol_m32, idx, dmn);
}
__global__ void k_debug_sincosf(const float *in, uint64_t n, float *out) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float s, c; glibcSincosf(in[i], s, c); out[i * 2] = s; out[i * 2 + 1] = c; }
}
__global__ void k_debug_libm(int fn, const float *x, const float *y, uint64_t n, float *out) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[i], b = y ? y[i] : 0.0f; float r;
    switch (fn) { case 0: r = expf(a); break; case 1: r = logf(a); break; case 2: r = powf(a, b); break; case 3: r = tanf(a); break; case 4: r = atanf(a); break; case 5: r = atan2f(a, b); break; default: r = acosf(a); break; }
    out[i] = r;
}
__global__ void k_debug_camera(DScene sc, const float *pos, uint64_t n, float *out) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    v3 o, d; float mint, maxt; cameraRay(sc, pos[i * 2], pos[i * 2 + 1], o, d, mint, maxt);
    float *r = out + i * 8; r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = mint; r[4] = d.x; r[5] = d.y; r[6] = d.z; r[7] = maxt;
}


// ---------------------------------------------------------------------------------------------- launch wrappers (used by api.cpp)
extern "C" {
void mi_launch_generate(const DScene &sc, const RenderConst &rc, const Queues &q, const BatchDesc &bd, uint32_t grid, hipStream_t st) { hipLaunchKernelGGL(k_generate, dim3(grid), dim3(WG), 0, st, sc, rc, q, bd); }
void mi_launch_env_primary(const DScene &sc, const RenderConst &rc, const Queues &q, int buf, uint32_t grid, hipStream_t st) { hipLaunchKernelGGL(k_env_primary, dim3(grid), dim3(WG), 0, st, sc, rc, q, buf); }
void mi_launch_film(const DScene &sc, const Queues &q, const BatchDesc &bd, float *film, float *spill, hipStream_t st) { hipLaunchKernelGGL(k_film, dim3((bd.n_pix + WG - 1) / WG), dim3(WG), 0, st, sc, q, bd, film, spill); }
void mi_launch_film_layout(const float *film, const float *spill, float *out, int W, int H, int border, int layout, hipStream_t st) {
    size_t n = (size_t) W * H; hipLaunchKernelGGL(k_film_layout, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, film, spill, out, W, H, border, layout);
}
void mi_launch_film_add(float *dst, const float *src, size_t n, hipStream_t st) { hipLaunchKernelGGL(k_film_add, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, dst, src, n); }
void mi_launch_gather_samples(const Queues &q, const uint32_t *slots, uint64_t n, float *out, hipStream_t st) { hipLaunchKernelGGL(k_gather_samples, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, q, slots, n, out); }
void mi_launch_debug_sobol(const DScene &sc, const uint32_t *in, uint64_t n, uint32_t ndims, unsigned long long *oi, float *ov, hipStream_t st) { hipLaunchKernelGGL(k_debug_sobol, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, sc, in, n, ndims, oi, ov); }
void mi_launch_debug_sincosf(const float *in, uint64_t n, float *out, hipStream_t st) { hipLaunchKernelGGL(k_debug_sincosf, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, in, n, out); }
void mi_launch_debug_libm(int fn, const float *x, const float *y, uint64_t n, float *out, hipStream_t st) { hipLaunchKernelGGL(k_debug_libm, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, fn, x, y, n, out); }
void mi_launch_debug_camera(const DScene &sc, const float *pos, uint64_t n, float *out, hipStream_t st) { hipLaunchKernelGGL(k_debug_camera, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, sc, pos, n, out); }
}
