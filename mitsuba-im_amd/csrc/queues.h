// queues.h -- SoA wavefront queues in HBM (device pointers), shared by api.cpp and the kernel translation units (kernels_*.hip).
// Layout per path slot (algorithmic bytes, DESIGN.md §"bytes per segment"):
//   extension ray 32 B (rayO: o.xyz,mint | rayD: d.xyz,maxt), hit 16 B (t,u,v,prim), path state 36 B
//   (st0: path id, sampler word a, sampler word b, dim|depth|flags; st1: throughput rgb, eta; st2: bsdfPdf),
//   shadow record 48 B (shO: o.xyz,maxt | shD: d.xyz,path id | shC: contribution rgb), accumulator 16 B, film position 8 B.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mi355pt.h"

#define MI_TICKETS 1024
struct Queues {
    float4 *rayO[2], *rayD[2];
    uint4 *st0[2]; float4 *st1[2]; float *st2[2];
    float *st3[2];                              // only with a `constant` environment emitter: cos(wo, refN) of the vertex that spawned the ray (2 = no reference normal)
    float4 *hit;
    int32_t *hitInst;                           // only in scenes with instances: instance index of the hit (-1: a scene-level primitive)
    float4 *shO, *shD, *shC;
    float4 *shT, *shX;                          // volumetric integrators only: throughput and BSDF / phase value of a shadow record (the transmittance between the two enters before them, shade_vol.h)
    float4 *acc; float2 *pos;
    uint32_t *count[2]; uint32_t *shCount;      // per segment
    uint32_t *ticket;                           // [MI_TICKETS] segment tickets of the fused traversal launches of a batch (trace_fused.h), zeroed when the batch starts
    int32_t *stkSpill;                          // fused walk: stack entries beyond FZ_LDS_STACK, one column per lane of the persistent grid (entry e of lane l at [e * lanes + l])
    unsigned long long *counters;               // [0] closest-hit rays, [1] shadow rays, [2] sum of path depths
    uint32_t cap;                               // slots per segment (multiple of 64)
    uint32_t n_seg;                             // number of segments
};

// Element i of a queue array through a 32-bit BYTE offset: the address is base (a uniform pointer: SGPR pair) + a 32-bit vector offset, one shift instead of the
// 64-bit shift-and-add per access that `base[i]` costs with a 64-bit index -- integer vector operations issue at half the rate of fp32 multiply-adds on gfx950
// (scripts/ubench/valu_rates.hip), and the shade stage touches ~20 arrays per path.  Holds while a pool has fewer than 2^28 slots (allocPoolQ enforces it).
template <typename T> __device__ __forceinline__ T &qat(T *base, uint32_t i) { return *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + (uint32_t) (i * (uint32_t) sizeof(T))); }
template <typename T> __device__ __forceinline__ const T &qat(const T *base, uint32_t i) { return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (uint32_t) (i * (uint32_t) sizeof(T))); }

struct BatchDesc {
    mi_tile tile; uint32_t n_pix; uint32_t n_planes; uint32_t sample_begin; uint64_t n_paths;
    uint32_t row_stride;    // film rows of the tile: y0, y0 + row_stride, ... (1 = contiguous rectangle; N = rows interleaved over N ranks)
    const uint32_t *list;   // optional explicit (px, py, sampleIndex) triples, one per path (parity entry point mi_render_samples)
};
