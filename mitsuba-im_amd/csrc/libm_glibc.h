// libm_glibc.h -- glibc 2.35's single-precision routines, restated operation for operation, for BOTH sides of the parity contract:
//   * the gfx950 kernels (pt_device.h) call these instead of the device math library, so that a radiance sample which passes through exp / log / pow / atan2 /
//     acos / tan can equal the oracle's -- the oracle calls the host's libm, as the reference does -- bit for bit (same idea as glibcSincosf);
//   * scripts/check_libm.c compiles this very file on the host and compares every routine against the host's libm: all 2^32 arguments of the one-argument
//     routines, 2^31 pseudo-random pairs of the others.
// expf / logf / powf: the ARM optimized-routines code glibc adopted in 2.27-2.28 (sysdeps/ieee754/flt-32/e_expf.c, e_logf.c, e_powf.c: binary64 arithmetic
// inside, table + short polynomial).  atanf / atan2f / acosf / tanf: fdlibm's float routines (s_atanf.c, e_atan2f.c, e_acosf.c, k_tanf.c + the |x| < 3 pi / 4
// branch of e_rem_pio2f.c).  No contraction anywhere (-ffp-contract=off on both sides).  Arguments outside the ranges the path produces fall back to the
// platform's routine (MI_LIBM_FALLBACK).
#pragma once
#include <stdint.h>
#ifdef MI_LIBM_HOST
#include <math.h>
#include <string.h>
#define MI_LIBM_FN static inline
#define MI_LIBM_TAB static const
static inline uint32_t mi_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float mi_asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t mi_asuint64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double mi_asdouble(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
#define MI_LIBM_FALLBACK(fn, ...) fn(__VA_ARGS__)
#define MI_FMA(a, b, c) fma(a, b, c)
#else
#define MI_LIBM_FN __device__ static inline
#define MI_LIBM_TAB __device__ static const
__device__ static inline uint32_t mi_asuint(float f) { return __float_as_uint(f); }
__device__ static inline float mi_asfloat(uint32_t u) { return __uint_as_float(u); }
__device__ static inline uint64_t mi_asuint64(double f) { return (uint64_t) __double_as_longlong(f); }
__device__ static inline double mi_asdouble(uint64_t u) { return __longlong_as_double((long long) u); }
#define MI_LIBM_FALLBACK(fn, ...) fn(__VA_ARGS__)
#define MI_FMA(a, b, c) __builtin_fma(a, b, c)
#endif

// ---------------------------------------------------------------------------------------------- exp2f_data.c (N = 32), shared by expf and powf
MI_LIBM_TAB uint64_t mi_exp2f_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

// sysdeps/ieee754/flt-32/e_expf.c
MI_LIBM_FN float mi_expf(float x) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    const uint32_t abstop = (mi_asuint(x) >> 20) & 0x7ffu;
    if (abstop >= (0x42b00000u >> 20)) {      /* |x| >= 88 or NaN */
        if (mi_asuint(x) == 0xff800000u) return 0.0f;
        if (abstop >= (0x7f800000u >> 20)) return x + x;
        if (x > 0x1.62e42ep6f) return mi_asfloat(0x7f800000u);      /* overflow */
        if (x < -0x1.9fe368p6f) return 0.0f;                        /* underflow */
    }
    const double xd = (double) x;
    /* the host's libm runs glibc's FMA build of this routine (sysdeps/x86_64/fpu/multiarch/e_expf.c = the same C under -mfma): every product that feeds only
       additions is contracted, the scaled argument z = InvLn2N * xd included (it feeds z + SHIFT and z - kd) -- scripts/check_libm.c pins this against the host */
    double kd = MI_FMA(InvLn2N, xd, SHIFT); const uint64_t ki = mi_asuint64(kd); kd -= SHIFT;
    const double r = MI_FMA(InvLn2N, xd, -kd);
    uint64_t t = mi_exp2f_tab[ki % 32]; t += ki << (52 - 5);
    const double s = mi_asdouble(t);
    double z = MI_FMA(C0, r, C1); const double r2 = r * r; double y = MI_FMA(C2, r, 1.0); y = MI_FMA(z, r2, y); y = y * s;
    return (float) y;
}

// sysdeps/ieee754/flt-32/e_logf.c (logf_data.c: N = 16)
MI_LIBM_TAB double mi_logf_tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2}, {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3}, {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0}, {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5}, {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3}, {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2}, {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
MI_LIBM_FN float mi_logf(float x) {
    const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = mi_asuint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2 == 0) return mi_asfloat(0xff800000u);            /* log(+-0) = -inf */
        if (ix == 0x7f800000u) return x;                            /* log(inf) = inf */
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return (x - x) / (x - x);      /* negative or NaN */
        ix = mi_asuint(x * 0x1p23f); ix -= 23u << 23;               /* subnormal: normalise */
    }
    const uint32_t tmp = ix - 0x3f330000u; const int i = (int) ((tmp >> (23 - 4)) % 16); const int k = (int32_t) tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double invc = mi_logf_tab[i][0], logc = mi_logf_tab[i][1], z = (double) mi_asfloat(iz);
    const double r = MI_FMA(z, invc, -1.0), y0 = MI_FMA((double) k, Ln2, logc), r2 = r * r;      /* (FMA build, as expf) */
    double y = MI_FMA(A1, r, A2); y = MI_FMA(A0, r2, y); y = MI_FMA(y, r2, y0 + r);
    return (float) y;
}

// sysdeps/ieee754/flt-32/e_powf.c (powf_log2_data.c: N = 16, POWF_SCALE = 1), positive finite bases; anything else: the platform's powf
MI_LIBM_TAB double mi_powf_log2_tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2}, {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2}, {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3}, {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0}, {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4}, {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2}, {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2}, {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
MI_LIBM_FN float mi_powf(float x, float y) {
    const uint32_t ix = mi_asuint(x), iy = mi_asuint(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || 2 * iy - 1 >= 2u * 0x7f800000u - 1) return MI_LIBM_FALLBACK(powf, x, y);      /* x <= 0, subnormal, inf, NaN; y zero, inf, NaN */
    /* log2_inline */
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
    const uint32_t tmp = ix - 0x3f330000u; const int i = (int) ((tmp >> (23 - 4)) % 16); const uint32_t top = tmp & 0xff800000u, iz = ix - top; const int k = (int32_t) top >> 23;
    const double invc = mi_powf_log2_tab[i][0], logc = mi_powf_log2_tab[i][1], z = (double) mi_asfloat(iz);
    const double r = MI_FMA(z, invc, -1.0), y0 = logc + (double) k;      /* (FMA build, as expf; y * log2(x) itself also feeds comparisons and stays a product) */
    const double r2 = r * r; double yy = MI_FMA(A0, r, A1); const double p = MI_FMA(A2, r, A3), r4 = r2 * r2; double q = MI_FMA(A4, r, y0); q = MI_FMA(p, r2, q); yy = MI_FMA(yy, r4, q);
    const double ylogx = (double) y * yy;
    if (((mi_asuint64(ylogx) >> 47) & 0xffff) >= (mi_asuint64(126.0) >> 47)) {      /* |y log2 x| >= 126 */
        if (ylogx > 0x1.fffffffd1d571p+6) return mi_asfloat(0x7f800000u);
        if (ylogx <= -150.0) return 0.0f;
    }
    /* exp2_inline, sign_bias = 0 */
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1, SHIFT = 0x1.8p+52 / 32;
    double kd = ylogx + SHIFT; const uint64_t ki = mi_asuint64(kd); kd -= SHIFT;
    const double rr = ylogx - kd;
    uint64_t t = mi_exp2f_tab[ki % 32]; t += ki << (52 - 5);
    const double s = mi_asdouble(t);
    double zz = MI_FMA(C0, rr, C1); const double rr2 = rr * rr; double out = MI_FMA(C2, rr, 1.0); out = MI_FMA(zz, rr2, out); out = out * s;
    return (float) out;
}

// ---------------------------------------------------------------------------------------------- fdlibm float routines
// sysdeps/ieee754/flt-32/s_atanf.c
MI_LIBM_FN float mi_atanf(float x) {
    /* the decimal literals of s_atanf.c (fdlibm's hex comments are not always the literal's value: 3.3333334327e-01 is 0x3eaaaaab, not 0x3eaaaaaa) */
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
                aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const float one = 1.0f;
    const int32_t hx = (int32_t) mi_asuint(x), ix = hx & 0x7fffffff; int id;
    if (ix >= 0x4c000000) {      /* |x| >= 2^25 */
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {       /* |x| < 0.4375 */
        if (ix < 0x31000000) return x;      /* |x| < 2^-29 */
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {   /* |x| < 1.1875 */
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - one) / (2.0f + x); }
            else { id = 1; x = (x - one) / (x + one); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    float z = x * x; const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -z : z;
}
// sysdeps/ieee754/flt-32/e_atan2f.c
MI_LIBM_FN float mi_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = mi_asfloat(0x3f490fdbu), pi_o_2 = mi_asfloat(0x3fc90fdbu), pi = mi_asfloat(0x40490fdbu), pi_lo = mi_asfloat(0xb3bbbd2eu);
    const int32_t hx = (int32_t) mi_asuint(x), hy = (int32_t) mi_asuint(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return mi_atanf(y);
    const int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) { switch (m) { case 0: case 1: return y; case 2: return pi + tiny; default: return -pi - tiny; } }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) { switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny; case 2: return 3.0f * pi_o_4 + tiny; default: return -3.0f * pi_o_4 - tiny; } }
        else { switch (m) { case 0: return 0.0f; case 1: return -0.0f; case 2: return pi + tiny; default: return -pi - tiny; } }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23; float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = mi_atanf(fabsf(y / x));
    switch (m) {
        case 0: return z;
        case 1: return mi_asfloat(mi_asuint(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}
// sysdeps/ieee754/flt-32/e_acosf.c
MI_LIBM_FN float mi_acosf(float x) {
    const float one = 1.0f, pi = mi_asfloat(0x40490fdau), pio2_hi = mi_asfloat(0x3fc90fdau), pio2_lo = mi_asfloat(0x33a22168u);
    const float pS0 = mi_asfloat(0x3e2aaaabu), pS1 = mi_asfloat(0xbea6b090u), pS2 = mi_asfloat(0x3e4e0aa8u), pS3 = mi_asfloat(0xbd241146u), pS4 = mi_asfloat(0x3a4f7f04u), pS5 = mi_asfloat(0x3811ef08u);
    const float qS1 = mi_asfloat(0xc019d139u), qS2 = mi_asfloat(0x4001572du), qS3 = mi_asfloat(0xbf303361u), qS4 = mi_asfloat(0x3d9dc62eu);
    const int32_t hx = (int32_t) mi_asuint(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;      /* |x| == 1 */
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {      /* |x| < 0.5 */
        if (ix <= 0x32800000) return pio2_hi + pio2_lo;      /* |x| < 2^-26 */
        const float z = x * x;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (hx < 0) {        /* x < -0.5 */
        const float z = (one + x) * 0.5f;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float s = sqrtf(z), r = p / q, w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    } else {                    /* x > 0.5 */
        const float z = (one - x) * 0.5f, s = sqrtf(z);
        const float df = mi_asfloat(mi_asuint(s) & 0xfffff000u);
        const float c = (z - df * df) / (s + df);
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float r = p / q, w = r * s + c;
        return 2.0f * (df + w);
    }
}
// sysdeps/ieee754/flt-32/k_tanf.c
MI_LIBM_FN float mi_kernel_tanf(float x, float y, int iy) {
    const float one = 1.0f, pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
    const float T0 = 3.3333334327e-01f, T1 = 1.3333334029e-01f, T2 = 5.3968254477e-02f, T3 = 2.1869488060e-02f, T4 = 8.8632395491e-03f, T5 = 3.5920790397e-03f,
                T6 = 1.4562094584e-03f, T7 = 5.8804126456e-04f, T8 = 2.4646313977e-04f, T9 = 7.8179444245e-05f, T10 = 7.1407252108e-05f, T11 = -1.8558637748e-05f, T12 = 2.5907305826e-05f;
    const int32_t hx = (int32_t) mi_asuint(x), ix = hx & 0x7fffffff;
    if (ix < 0x39000000) {      /* |x| < 2^-13 */
        if ((int) x == 0) {
            if ((ix | (iy + 1)) == 0) return one / fabsf(x);
            else if (iy == 1) return x;
            else return -one / x;
        }
    }
    if (ix >= 0x3f2ca140) {     /* |x| >= 0.6744 */
        if (hx < 0) { x = -x; y = -y; }
        const float z0 = pio4 - x, w0 = pio4lo - y;
        x = z0 + w0; y = 0.0f;
        if (fabsf(x) < 0x1p-13f) return (float) (1 - ((hx >> 30) & 2)) * (float) iy * (1.0f - 2.0f * (float) iy * x);
    }
    float z = x * x, w = z * z;
    float r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
    float v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
    float s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T0 * s;
    w = x + r;
    if (ix >= 0x3f2ca140) { v = (float) iy; return (float) (1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r))); }
    if (iy == 1) return w;
    /* -1 / (x + r), accurately */
    z = mi_asfloat(mi_asuint(w) & 0xfffff000u);
    v = r - (z - x);
    const float a = -1.0f / w; const float t = mi_asfloat(mi_asuint(a) & 0xfffff000u);
    s = 1.0f + t * z;
    return t + a * (s + t * v);
}
// sysdeps/ieee754/flt-32/s_tanf.c over __ieee754_rem_pio2f for |x| <= 64 (n = rint(|x| 2 / pi) <= 41); larger arguments: the platform's tanf.
// The reduction |x| - n pi / 2 is done in binary64 with the one 53-bit constant, head and tail rounded to float: against the host's libm (glibc 2.35) this form differs in
// none of the 2.2e9 arguments of the range (scripts/check_libm.c), while the 24 + 24-bit Cody-Waite steps of fdlibm's float version differ in 277 arguments below 3 pi / 4
// alone.  (First mismatches of this form: four arguments next to 77 pi / 2 = 120.95.)
MI_LIBM_FN float mi_tanf(float x) {
    const int32_t hx = (int32_t) mi_asuint(x), ix = hx & 0x7fffffff;
    if (ix <= 0x3f490fda) return mi_kernel_tanf(x, 0.0f, 1);      /* |x| ~<= pi / 4 */
    if (ix > 0x42800000) return MI_LIBM_FALLBACK(tanf, x);         /* |x| > 64 (also inf / NaN) */
    const double t = fabs((double) x);
    int n = (int) (t * 6.36619772367581382433e-01 + 0.5);
    const double r = t - (double) n * 1.57079632679489661923;
    float y0 = (float) r, y1 = (float) (r - (double) y0);
    if (hx < 0) { y0 = -y0; y1 = -y1; n = -n; }
    return mi_kernel_tanf(y0, y1, 1 - ((n & 1) << 1));
}
