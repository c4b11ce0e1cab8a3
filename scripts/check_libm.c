// check_libm.c -- restatements of the glibc 2.35 single-precision routines the path calls (expf, logf, powf: ARM optimized-routines, binary64 inside; atanf, atan2f,
// acosf, tanf: fdlibm's float code), compared bit for bit against the host's libm: every float for the one-argument routines, 2^31 pseudo-random pairs for the others.
// The SAME code (mitsuba-im_amd/csrc/libm_glibc.h, included below) is compiled into the gfx950 kernels, so that radiance samples that pass through these routines
// can equal the oracle's (which calls libm) bit for bit.  Test infrastructure:  gcc -O2 -fopenmp -ffp-contract=off scripts/check_libm.c -lm -o /tmp/check_libm
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#define MI_LIBM_HOST 1
#include "../mitsuba-im_amd/csrc/libm_glibc.h"

static inline uint32_t bitsf(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float fromBits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static int same(float a, float b) { return bitsf(a) == bitsf(b) || (a != a && b != b); }

#define CHECK1(NAME, MINE, REF, LO, HI) do { \
    unsigned long long bad = 0, tot = 0; \
    _Pragma("omp parallel for reduction(+:bad,tot) schedule(dynamic, 1 << 20)") \
    for (long long u = 0; u < (1ll << 32); ++u) { const float x = fromBits((uint32_t) u); if (!(x >= (LO) && x <= (HI))) continue; ++tot; \
        const float a = MINE(x), b = REF(x); if (!same(a, b)) { if (bad < 5) printf("  " NAME "(%a): libm %a mine %a\n", x, b, a); ++bad; } } \
    printf("%-8s on [%g, %g]: %llu arguments, %llu mismatches\n", NAME, (double) (LO), (double) (HI), tot, bad); total_bad += bad; } while (0)

static uint64_t rngs(uint64_t *s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }

int main(int argc, char **argv) {
    unsigned long long total_bad = 0;
    const int quick = argc > 1;
    (void) quick;
    CHECK1("expf", mi_expf, expf, -110.0f, 90.0f);
    CHECK1("logf", mi_logf, logf, 0.0f, 3.4e38f);
    CHECK1("atanf", mi_atanf, atanf, -3.4e38f, 3.4e38f);
    CHECK1("acosf", mi_acosf, acosf, -1.0f, 1.0f);
    CHECK1("tanf", mi_tanf, tanf, -64.0f, 64.0f);
    {   // atan2f: all sign combinations, magnitudes log-uniform over 2^-40 .. 2^40 plus exact zeros
        unsigned long long bad = 0, tot = 0;
        #pragma omp parallel for reduction(+:bad,tot)
        for (int t = 0; t < 64; ++t) { uint64_t s = 0x9E3779B97F4A7C15ull * (t + 1);
            for (long long i = 0; i < (1ll << 25); ++i) { const uint64_t r = rngs(&s);
                float y = fromBits((uint32_t) (((r >> 1) & 0x807FFFFFu) | ((uint32_t) (87 + ((r >> 40) % 81)) << 23))), x = fromBits((uint32_t) (((r >> 33) & 0x7FFFFFu) | (((r >> 60) & 1) << 31) | ((uint32_t) (87 + ((r >> 50) % 81)) << 23)));
                if ((r & 0xFFF) == 0) y = 0.0f; if ((r & 0xFFF000) == 0) x = (r & 1) ? 1.0f : 0.0f;
                ++tot; const float a = mi_atan2f(y, x), b = atan2f(y, x); if (!same(a, b)) { if (bad < 5) printf("  atan2f(%a, %a): libm %a mine %a\n", y, x, b, a); ++bad; } } }
        printf("%-8s: %llu pairs, %llu mismatches\n", "atan2f", tot, bad); total_bad += bad;
    }
    {   // powf: bases in (0, 4] and [1e-6, 1e6], exponents in [-64, 64] -- the path raises cosines / (1 - u) to roughness-derived powers
        unsigned long long bad = 0, tot = 0;
        #pragma omp parallel for reduction(+:bad,tot)
        for (int t = 0; t < 64; ++t) { uint64_t s = 0xD1B54A32D192ED03ull * (t + 1);
            for (long long i = 0; i < (1ll << 25); ++i) { const uint64_t r = rngs(&s);
                const float x = (r & 1) ? fromBits((uint32_t) (((r >> 8) & 0x7FFFFFu) | ((uint32_t) (107 + ((r >> 32) % 41)) << 23))) : (float) ((r >> 8) & 0xFFFFFF) * (4.0f / 16777216.0f);
                const float y = ((float) ((r >> 36) & 0xFFFFFF) * (128.0f / 16777216.0f) - 64.0f) * (((r >> 61) & 1) ? 1.0f : 0.03125f);
                ++tot; const float a = mi_powf(x, y), b = powf(x, y); if (!same(a, b)) { if (bad < 5) printf("  powf(%a, %a): libm %a mine %a\n", x, y, b, a); ++bad; } } }
        printf("%-8s: %llu pairs, %llu mismatches\n", "powf", tot, bad); total_bad += bad;
    }
    printf("total mismatches %llu\n", total_bad);
    return total_bad ? 1 : 0;
}
