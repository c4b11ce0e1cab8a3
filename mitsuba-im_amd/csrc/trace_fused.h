// trace_fused.h -- the tree ray cast as ONE fused loop over persistent waves (triangle-only trees: no analytic shapes, no instances).
//
// Why (scripts/sim/bvh_sim.cpp, the lock-step model of the walk on the 251 k-triangle atrium): a ray visits ~12 four-wide nodes and ~4 triangles, but the
// "while-while" walk of trace.h (all lanes descend until each holds a leaf, then all test their leaves) keeps only 24 % of a wave64's lanes busy on incoherent
// rays -- a wave issues ~49 node steps and ~11 triangle steps for rays that need 12 and 4, and every step is a dependent memory round trip.  Sorting the rays of a
// segment (octant + Morton) moves that by 5 %.  What does move it:
//   * ONE loop in which every busy lane takes one step of whatever it needs next -- an inner node (64 B) or a triangle (48 B) -- so nobody waits for the other
//     kind; the loads of both kinds are issued together at the top of the iteration (one round trip per iteration instead of one per kind);
//   * idle lanes are REFILLED from the wave's ray stream as soon as fewer than `thr` lanes are busy (a wave owns whole segments, fetched through a ticket
//     counter, so there is no workgroup barrier and no LDS exchange in the loop);
//   * the stack holds child codes themselves (nearest child is entered, the others are pushed farthest first), so a pop is one LDS read -- no re-read of the
//     parent's child pointers in the dependent chain.
// Model: 101 -> 57 issued vector instructions per ray on bounces >= 2 at 83 % busy lanes, a third of the dependent round trips.
// What an instruction costs (scripts/ubench/valu_rates.hip, 8 waves per SIMD): v_fma / v_add / v_mul / v_mov issue every ~2.4 cycles, EVERYTHING else --
// compares, selects, conversions, min / max, integer and bit operations -- every ~4.2 cycles on a second pipe that runs beside the first, v_rcp 8, an IEEE
// division ~40.  The walk is bound by that second pipe (measured 40 ms per 328 M rays = 200 such operations per wave iteration x 4.2 cycles), so the code below
// counts those: the children behind the nearest one are pushed in slot order (the model: 12.1 instead of 12.0 node visits per ray, no sort network, no 64-bit
// compares), the Wald test's axis rotation reads the ray from an LDS copy (three ds_read2st64 instead of twelve selects).
// The arithmetic is trace.h's: same conservative box tests, same exact Wald test, closest hit = minimum t with ties to the lower triangle index -- the result is
// independent of the visiting order, so (t, u, v, prim) stay bit-identical (tests: test_intersection_bit_exact, test_both_tree_node_kinds, test_fused_walk_*).
#pragma once
#include "trace.h"

#define FZ_IDLE 0x7FFFFFFF
DEV float fastInv(float d) { const float a = fabsf(d) < 1e-30f ? copysignf(1e-30f, d) : d; return __builtin_amdgcn_rcpf(a); }

// ANY = false: closest hit of the extension rays of buffer `buf` -> Queues::hit.  ANY = true: visibility of the shadow records -> deferred `Li +=` into Queues::acc.
// WIDE: Bvh4Node / BvhNode records.  `ticket`: zeroed counter handing out segments to waves.
// LDS per workgroup: FZ_LDS_STACK stack entries per lane (deeper entries -- the model sees 8 entries or fewer for 99.97 % of the rays, 11 at most on the atrium,
// while the builder's bound DScene::bvh_stack_direct is 35 there -- spill to Queues::stkSpill, one column per lane of the persistent grid) and ten words per
// lane for the ray (o.x o.y o.z o.x o.y | d.x d.y d.z d.x d.y: the components (k, u, v) of a triangle's projection axis k are three consecutive words).
#define FZ_LDS_STACK 10
#define FZ_LDS_WORDS ((FZ_LDS_STACK + 10) * WG)
template <bool ANY, bool WIDE>
DEV void fusedStage(const DScene &sc, const Queues &q, const int buf, uint32_t *ticket, const uint32_t thr, int *s_lds) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int *stk = s_lds + tid;
    float *rayL = reinterpret_cast<float *>(s_lds + FZ_LDS_STACK * WG) + tid;      // word j of this lane's ray at rayL[j * WG]
    int *spill = q.stkSpill + ((size_t) blockIdx.x * WG + tid); const size_t spillStride = (size_t) gridDim.x * WG;
    const char *geo = reinterpret_cast<const char *>(sc.nodes);
    const uint32_t triOff = (uint32_t) (reinterpret_cast<const char *>(sc.tris) - geo);      // mi_fused_walk: nodes + leaf records are one allocation smaller than 4 GB
    const uint32_t *segCount = ANY ? q.shCount : q.count[buf];
    const float4 *rO = ANY ? q.shO : q.rayO[buf], *rD = ANY ? q.shD : q.rayD[buf];
    // per-lane ray (o and d live in LDS for the triangle test, 1 / d and -o / d in registers for the box tests)
    v3 inv = V(0, 0, 0), oi = V(0, 0, 0);
    float mint = 0, best = 0, bu = 0, bv = 0; uint32_t bprim = 0xFFFFFFFFu, pid = 0; uint64_t slot = 0;
    int cur = FZ_IDLE, sp = 0;
    // wave-uniform stream cursor
    uint32_t seg = 0, n = 0, nxt = 0; bool more = true; unsigned long long rays = 0;
    while (true) {
        unsigned long long busy = __ballot(cur != FZ_IDLE);
        if (more && (uint32_t) __popcll(busy) < thr) {
            unsigned long long idle = ~busy;
            while (idle) {
                if (nxt >= n) {                      // next segment of the pool
                    uint32_t s = 0; if (lane == 0) s = atomicAdd(ticket, 1u);
                    seg = (uint32_t) __builtin_amdgcn_readfirstlane((int) s);
                    if (seg >= q.n_seg) { more = false; break; }
                    n = (uint32_t) __builtin_amdgcn_readfirstlane((int) segCount[seg]); nxt = 0; rays += n;
                    continue;
                }
                const uint32_t want = (uint32_t) __popcll(idle), avail = n - nxt, take = want < avail ? want : avail;
                const uint32_t rank = (uint32_t) __popcll(idle & lt);
                if (((idle >> lane) & 1ull) && rank < take) {
                    slot = (uint64_t) seg * q.cap + nxt + rank;
                    const float4 ro = rO[slot], rd = rD[slot];
                    const v3 o = V(ro.x, ro.y, ro.z), d = V(rd.x, rd.y, rd.z);
                    float maxt;
                    const bool inside = ANY ? clipInterval(sc, o, d, MI_EPSILON, ro.w, true, mint, maxt) : clipInterval(sc, o, d, ro.w, rd.w, false, mint, maxt);
                    if (ANY) pid = __float_as_uint(rd.w);
                    if (inside) {
                        inv = V(fastInv(d.x), fastInv(d.y), fastInv(d.z)); oi = V(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);      // 1-ulp reciprocals: the box tests are conservative culling only (2e-6 slack)
                        rayL[0] = o.x; rayL[WG] = o.y; rayL[2 * WG] = o.z; rayL[3 * WG] = o.x; rayL[4 * WG] = o.y;
                        rayL[5 * WG] = d.x; rayL[6 * WG] = d.y; rayL[7 * WG] = d.z; rayL[8 * WG] = d.x; rayL[9 * WG] = d.y;
                        best = maxt; bprim = 0xFFFFFFFFu; bu = 0; bv = 0; cur = 0; sp = 0;
                    } else if (ANY) {                // the segment misses the scene box: unoccluded
                        const float4 c = q.shC[slot]; float4 a = q.acc[pid]; a.x += c.x; a.y += c.y; a.z += c.z; q.acc[pid] = a;
                    } else q.hit[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
                }
                nxt += take;
                idle = __ballot(cur == FZ_IDLE);
            }
            busy = __ballot(cur != FZ_IDLE);
        }
        if (!busy) { if (!more) break; continue; }
        // ---- one step per busy lane: an inner node (cur >= 0) or the next triangle of a leaf (cur < 0: ~cur = first * 8 + (remaining - 1)).  Both kinds load
        //      together; an idle lane re-reads the root (no exec juggling around the loads)
        const bool live = cur != FZ_IDLE, isNode = cur >= 0;
        const uint32_t code = (uint32_t) ~cur;
        const uint32_t off = isNode ? (live ? (uint32_t) cur << 6 : 0u) : triOff + (code >> 3) * 48u;      // byte offset into the one nodes + leaf-records allocation
        const f4 *p = reinterpret_cast<const f4 *>(geo + off);
        const f4 r0 = p[0], r1 = p[1], r2 = p[2], r3 = *reinterpret_cast<const f4 *>(geo + (off + (isNode ? 48u : 32u)));      // (a triangle lane re-reads its third word)
        bool pop = false, finished = false;
        if (live & isNode) {
            if (WIDE) {
                const float bx = r0.w * inv.x, by = r2.z * inv.y, bz = r2.w * inv.z;      // quantisation steps (floats in the node) x 1 / d
                const float ax = __builtin_fmaf(r0.x, inv.x, oi.x), ay = __builtin_fmaf(r0.y, inv.y, oi.y), az = __builtin_fmaf(r0.z, inv.z, oi.z);
                const uint32_t lx = __float_as_uint(r1.x), ly = __float_as_uint(r1.y), lz = __float_as_uint(r1.z), hx = __float_as_uint(r1.w), hy = __float_as_uint(r2.x), hz = __float_as_uint(r2.y);
                const uint32_t nxq = inv.x >= 0 ? lx : hx, fxq = inv.x >= 0 ? hx : lx, nyq = inv.y >= 0 ? ly : hy, fyq = inv.y >= 0 ? hy : ly, nzq = inv.z >= 0 ? lz : hz, fzq = inv.z >= 0 ? hz : lz;
                uint32_t key[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float tn = fmaxf(fmaxf(__builtin_fmaf((float) ((nxq >> (8 * c)) & 0xFFu), bx, ax), __builtin_fmaf((float) ((nyq >> (8 * c)) & 0xFFu), by, ay)),
                                           fmaxf(__builtin_fmaf((float) ((nzq >> (8 * c)) & 0xFFu), bz, az), mint));
                    const float tf = fminf(fminf(__builtin_fmaf((float) ((fxq >> (8 * c)) & 0xFFu), bx, ax), __builtin_fmaf((float) ((fyq >> (8 * c)) & 0xFFu), by, ay)),
                                           fminf(__builtin_fmaf((float) ((fzq >> (8 * c)) & 0xFFu), bz, az), best));
                    key[c] = (tn <= __builtin_fmaf(tf, 1.000002f, 1e-30f)) ? ((__float_as_uint(tn) & ~3u) | (uint32_t) c) : 0xFFFFFFFFu;      // tn >= mint > 0: its bit pattern orders like the value
                }
                // nearest child: the smallest key (its low two bits name the slot); the other children that are hit wait on the stack in slot order -- plain
                // stores, the stack pointer moves only past the real ones
                const uint32_t m = min(min(key[0], key[1]), min(key[2], key[3]));
                const int c0 = __float_as_int(r3.x), c1 = __float_as_int(r3.y), c2 = __float_as_int(r3.z), c3 = __float_as_int(r3.w);
                const int b0 = __builtin_amdgcn_sbfe((int) m, 0, 1), b1 = __builtin_amdgcn_sbfe((int) m, 1, 1);
                const int lo = (c1 & b0) | (c0 & ~b0), hi = (c3 & b0) | (c2 & ~b0), nearest = (hi & b1) | (lo & ~b1);
                const int w0 = (key[0] != m) & (key[0] != 0xFFFFFFFFu), w1 = (key[1] != m) & (key[1] != 0xFFFFFFFFu), w2 = (key[2] != m) & (key[2] != 0xFFFFFFFFu), w3 = (key[3] != m) & (key[3] != 0xFFFFFFFFu);
                if (sp <= FZ_LDS_STACK - 4) {
                    stk[sp * WG] = c0; sp += w0; stk[sp * WG] = c1; sp += w1; stk[sp * WG] = c2; sp += w2; stk[sp * WG] = c3; sp += w3;
                } else {
                    if (w0) { if (sp < FZ_LDS_STACK) stk[sp * WG] = c0; else spill[(size_t) (sp - FZ_LDS_STACK) * spillStride] = c0; ++sp; }
                    if (w1) { if (sp < FZ_LDS_STACK) stk[sp * WG] = c1; else spill[(size_t) (sp - FZ_LDS_STACK) * spillStride] = c1; ++sp; }
                    if (w2) { if (sp < FZ_LDS_STACK) stk[sp * WG] = c2; else spill[(size_t) (sp - FZ_LDS_STACK) * spillStride] = c2; ++sp; }
                    if (w3) { if (sp < FZ_LDS_STACK) stk[sp * WG] = c3; else spill[(size_t) (sp - FZ_LDS_STACK) * spillStride] = c3; ++sp; }
                }
                pop = m == 0xFFFFFFFFu; cur = pop ? cur : nearest;
            } else {
                const int c0 = __float_as_int(r0.w), c1 = __float_as_int(r1.w);
                float t0, t1;
                const bool h0 = slab(r0, r1, inv, oi, mint, best, t0), h1 = slab(r2, r3, inv, oi, mint, best, t1);
                const bool both = h0 & h1, sw = t1 < t0;
                if (both) { const int far = sw ? c0 : c1; if (sp < FZ_LDS_STACK) stk[sp * WG] = far; else spill[(size_t) (sp - FZ_LDS_STACK) * spillStride] = far; ++sp; }
                pop = !(h0 | h1); cur = both ? (sw ? c1 : c0) : (h0 ? c0 : (h1 ? c1 : cur));
            }
        } else if (live) {
            // TriAccel::rayIntersect (triaccel.h:96-158), written without branches: r0 = k n_u n_v n_d | r1 = a_u a_v b_nu b_nv | r2 = c_nu c_nv prim pad.
            // (o_k, o_u, o_v) and (d_k, d_u, d_v) are words k, k + 1, k + 2 of the LDS copy of the ray
            const uint32_t k = __float_as_uint(r0.x), kk = k < 3u ? k : 0u;
            const float *rk = rayL + kk * WG;
            const float o_k = rk[0], o_u = rk[WG], o_v = rk[2 * WG], d_k = rk[5 * WG], d_u = rk[6 * WG], d_v = rk[7 * WG];
            const float tt = (r0.w - o_u * r0.y - o_v * r0.z - o_k) / (d_u * r0.y + d_v * r0.z + d_k);
            const float hu = o_u + tt * d_u - r1.x, hv = o_v + tt * d_v - r1.y;
            const float uu = hv * r1.z + hu * r1.w, vv = hu * r2.x + hv * r2.y;
            const bool ok = (k <= 2u) & !(tt < mint) & !(tt > best) & (uu >= 0) & (vv >= 0) & (uu + vv <= 1.0f);
            if (ANY) { finished = ok; bprim = ok ? 0u : bprim; }
            else {
                const uint32_t prim = __float_as_uint(r2.z);
                const bool better = ok & ((bprim == 0xFFFFFFFFu) | (tt < best) | ((tt == best) & (prim < bprim)));
                best = better ? tt : best; bprim = better ? prim : bprim; bu = better ? uu : bu; bv = better ? vv : bv;
            }
            pop = (code & 7u) == 0u; cur = pop ? cur : cur - 7;                                  // next triangle: first + 1, remaining - 1
        }
        {   // pop: one LDS read (or, beyond FZ_LDS_STACK entries, one read of the spill column); an empty stack retires the ray
            const bool doPop = pop & !finished;
            const int spm = sp > 0 ? sp - 1 : 0;
            int top = stk[(spm < FZ_LDS_STACK ? spm : FZ_LDS_STACK - 1) * WG];
            if (doPop && spm >= FZ_LDS_STACK) top = spill[(size_t) (spm - FZ_LDS_STACK) * spillStride];
            finished |= doPop & (sp == 0);
            cur = (doPop & (sp > 0)) ? top : cur; sp = doPop ? spm : sp;
        }
        if (finished) {
            cur = FZ_IDLE;
            if (ANY) {
                if (bprim == 0xFFFFFFFFu) { const float4 c = q.shC[slot]; float4 a = q.acc[pid]; a.x += c.x; a.y += c.y; a.z += c.z; q.acc[pid] = a; }
            } else q.hit[slot] = make_float4(best, bu, bv, __uint_as_float(bprim));      // a miss keeps t = the far end of the clipped interval, like traverse()
        }
    }
    if (!ANY && lane == 0 && rays) atomicAdd(&q.counters[0], rays);
}
