/*
 * wm_exact.h -- bit-exact scalar arithmetic shared by the HIP kernels and host-side checks.
 *
 * The reference binary is x86-64 baseline code: every float multiply/add is rounded on its own
 * (no FMA), and it calls glibc 2.35 libm for atan2f (via cargf, /root/reference/atan2.h:7-10)
 * and sqrtf.  To produce the same soft symbols on a GPU we
 *   - compile device code with -ffp-contract=off and use explicitly rounded helpers here,
 *   - restate glibc 2.35's atan2f/atanf (the fdlibm float algorithm, sysdeps/ieee754/flt-32/
 *     e_atan2f.c and s_atanf.c; glibc is NOT under /root/reference, it is the pinned third-party
 *     dependency of the reference's discriminator) operation by operation.
 * tests/test_exact_math.py compiles this header for the host and checks wm_atan2f against the
 * libm of this image bit-for-bit on the discriminator's input domain.
 *
 * WM_HD expands to __host__ __device__ under hipcc and to nothing under gcc.
 */
#ifndef WM_EXACT_H
#define WM_EXACT_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define WM_HD __host__ __device__ __forceinline__
#else
#define WM_HD static inline
#endif

WM_HD uint32_t wm_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
WM_HD float wm_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Individually rounded IEEE operations.  On the device these map to the *_rn intrinsics, which
 * the compiler never contracts into FMAs; on the host the file is built with -ffp-contract=off. */
#if defined(__HIP_DEVICE_COMPILE__)
WM_HD float wm_mul(float a, float b) { return __fmul_rn(a, b); }
WM_HD float wm_add(float a, float b) { return __fadd_rn(a, b); }
WM_HD float wm_sub(float a, float b) { return __fsub_rn(a, b); }
WM_HD float wm_div(float a, float b) { return __fdiv_rn(a, b); }
WM_HD float wm_sqrt(float a) { return __fsqrt_rn(a); }
#else
#include <math.h>
WM_HD float wm_mul(float a, float b) { return a * b; }
WM_HD float wm_add(float a, float b) { return a + b; }
WM_HD float wm_sub(float a, float b) { return a - b; }
WM_HD float wm_div(float a, float b) { return a / b; }
WM_HD float wm_sqrt(float a) { return sqrtf(a); }
#endif

/* fdlibm atanf for x >= 0 (the only case atan2f needs), constants given by bit pattern. */
WM_HD float wm_atanf_pos(float x)
{
    const uint32_t ix = wm_f2u(x) & 0x7fffffffu;
    const float atanhi3 = wm_u2f(0x3fc90fdau), atanlo3 = wm_u2f(0x33a22168u);
    if (ix >= 0x4c000000u) return wm_add(atanhi3, atanlo3);      /* |x| >= 2^25 */
    int id;
    if (ix < 0x3ee00000u) {                                      /* |x| < 0.4375 */
        if (ix < 0x31000000u) return x;                          /* |x| < 2^-29 */
        id = -1;
    } else if (ix < 0x3f980000u) {                               /* |x| < 1.1875 */
        if (ix < 0x3f300000u) { id = 0; x = wm_div(wm_sub(wm_mul(2.0f, x), 1.0f), wm_add(2.0f, x)); }
        else                  { id = 1; x = wm_div(wm_sub(x, 1.0f), wm_add(x, 1.0f)); }
    } else {
        if (ix < 0x401c0000u) { id = 2; x = wm_div(wm_sub(x, 1.5f), wm_add(1.0f, wm_mul(1.5f, x))); }
        else                  { id = 3; x = wm_div(-1.0f, x); }
    }
    const float aT0 = wm_u2f(0x3eaaaaabu), aT1 = wm_u2f(0xbe4ccccdu), aT2 = wm_u2f(0x3e124925u),
                aT3 = wm_u2f(0xbde38e38u), aT4 = wm_u2f(0x3dba2e6eu), aT5 = wm_u2f(0xbd9d8795u),
                aT6 = wm_u2f(0x3d886b35u), aT7 = wm_u2f(0xbd6ef16bu), aT8 = wm_u2f(0x3d4bda59u),
                aT9 = wm_u2f(0xbd15a221u), aT10 = wm_u2f(0x3c8569d7u);
    const float z = wm_mul(x, x);
    const float w = wm_mul(z, z);
    float s1 = wm_add(aT8, wm_mul(w, aT10));
    s1 = wm_add(aT6, wm_mul(w, s1));
    s1 = wm_add(aT4, wm_mul(w, s1));
    s1 = wm_add(aT2, wm_mul(w, s1));
    s1 = wm_add(aT0, wm_mul(w, s1));
    s1 = wm_mul(z, s1);
    float s2 = wm_add(aT7, wm_mul(w, aT9));
    s2 = wm_add(aT5, wm_mul(w, s2));
    s2 = wm_add(aT3, wm_mul(w, s2));
    s2 = wm_add(aT1, wm_mul(w, s2));
    s2 = wm_mul(w, s2);
    const float p = wm_mul(x, wm_add(s1, s2));
    if (id < 0) return wm_sub(x, p);
    float hi, lo;
    switch (id) {
    case 0:  hi = wm_u2f(0x3eed6338u); lo = wm_u2f(0x31ac3769u); break;
    case 1:  hi = wm_u2f(0x3f490fdau); lo = wm_u2f(0x33222168u); break;
    case 2:  hi = wm_u2f(0x3f7b985eu); lo = wm_u2f(0x33140fb4u); break;
    default: hi = atanhi3;             lo = atanlo3;             break;
    }
    return wm_sub(hi, wm_sub(wm_sub(p, lo), x));
}

/* fdlibm atan2f for finite arguments (NaN/Inf cannot occur: the discriminator's operands are
 * small exact rationals). */
WM_HD float wm_atan2f(float y, float x)
{
    const uint32_t hx = wm_f2u(x), hy = wm_f2u(y);
    const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    const float pi = wm_u2f(0x40490fdbu), pi_o_2 = wm_u2f(0x3fc90fdbu), pi_lo = wm_u2f(0xb3bbbd2eu);
    const uint32_t m = ((hy >> 31) & 1u) | ((hx >> 30) & 2u);
    if (hx == 0x3f800000u) {                                      /* x == 1.0: atanf(y) */
        const float a = wm_atanf_pos(wm_u2f(iy));
        return (hy >> 31) ? -a : a;
    }
    if (iy == 0u) {                                               /* y == +-0 */
        if (m < 2u) return y;
        return m == 2u ? pi : -pi;                                /* pi +- tiny rounds to pi */
    }
    if (ix == 0u) return (hy >> 31) ? -pi_o_2 : pi_o_2;           /* x == +-0 */
    const int k = ((int)iy - (int)ix) >> 23;
    float z;
    if (k > 60) z = wm_add(pi_o_2, wm_mul(0.5f, pi_lo));
    else if ((hx >> 31) && k < -60) z = 0.0f;
    else z = wm_atanf_pos(wm_u2f(wm_f2u(wm_div(y, x)) & 0x7fffffffu));
    switch (m) {
    case 0:  return z;
    case 1:  return wm_u2f(wm_f2u(z) ^ 0x80000000u);
    case 2:  return wm_sub(pi, wm_sub(z, pi_lo));
    default: return wm_sub(wm_sub(z, pi_lo), pi);
    }
}

/* Polar discriminator (rtl_wmbus.c:517-534 / 553-570): y = s * conj(s_prev), cargf(y)/pi.
 * gcc expands the complex product as (a*c - b*d) + j(a*d + b*c) with (c,d) = (i', -q'). */
WM_HD float wm_discriminator(float i, float q, float pi_, float pq_)
{
    const float c = pi_, d = -pq_;
    const float re = wm_sub(wm_mul(i, c), wm_mul(q, d));
    const float im = wm_add(wm_mul(i, d), wm_mul(q, c));
    return wm_mul(wm_atan2f(im, re), wm_u2f(0x3ea2f983u));        /* (float)M_1_PI */
}

/* -a variant (rtl_wmbus.c:536-551 / 572-586). */
WM_HD float wm_discriminator_fast(float i, float q, float pi_, float pq_)
{
    return wm_sub(wm_mul(pi_, q), wm_mul(i, pq_));
}

/* cu8 sample -> boxcar input (rtl_wmbus.c:1312-1313 then the int parameter of mavgi,
 * moving_average_filter.h:47): (int)((float)u8 - 127.5f), truncation toward zero. */
WM_HD int wm_quantise(unsigned u8) { return u8 >= 128u ? (int)u8 - 128 : (int)u8 - 127; }

#endif /* WM_EXACT_H */
