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
/* hipcc rounds `/` and sqrtf correctly by default (-fhip-fp32-correctly-rounded-divide-sqrt);
 * __fsqrt_rn does NOT (1 ulp off on 15 % of the RSSI operands, measured on MI355X), so it is not used. */
WM_HD float wm_div(float a, float b) { return a / b; }
WM_HD float wm_sqrt(float a) { return __builtin_sqrtf(a); }
#else
#include <math.h>
WM_HD float wm_mul(float a, float b) { return a * b; }
WM_HD float wm_add(float a, float b) { return a + b; }
WM_HD float wm_sub(float a, float b) { return a - b; }
WM_HD float wm_div(float a, float b) { return a / b; }
WM_HD float wm_sqrt(float a) { return sqrtf(a); }
#endif

/* a*b + c and a*b - c*d for products that are EXACT (integers below 2^24, powers of two times a float): one rounding either
 * way, so the fused form returns the very bits of the separately rounded one -- zero signs included: an exactly-zero sum of
 * opposite-signed terms is +0 under round-to-nearest fused or not, and a zero product carries the sign of its factors in both.
 * One instruction instead of two (round 5: the discriminator's complex product and the arctangent's numerator). */
#if defined(__HIP_DEVICE_COMPILE__)
WM_HD float wm_fma_exact(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#else
WM_HD float wm_fma_exact(float a, float b, float c) { return wm_add(wm_mul(a, b), c); }      /* the host states the claim; the device tests compare */
#endif

/* Correctly rounded divide and square root for TAME operands (no subnormal, infinite or NaN
 * operand or result, exponents far from the limits): the same Newton/FMA refinement the compiler
 * emits for `/` and sqrtf under -fhip-fp32-correctly-rounded-divide-sqrt, without its range
 * scaling (v_div_scale / v_div_fmas scaling, the 2^32 pre-scale of tiny sqrt arguments) and
 * special-case fix-up (v_div_fixup, the class test), which can only act outside that domain.
 * 11 -> 6 and ~18 -> 9 instructions.  The discriminator divides integers below 2^24 (and range
 * reduced values in [2^-24, 2^24]); the RSSI takes the root of an integer below 2^24.  A zero
 * DIVISOR gives NaN here instead of +-Inf: wm_atan2f_tab overrides x == 0 anyway.
 * tests/test_gpu_parity.py checks both on the device against the host's IEEE results:
 * the square root exhaustively on [0, 2^24), the quotient on 10^8 operand pairs of the domain. */
#if defined(__HIP_DEVICE_COMPILE__)
WM_HD float wm_div_dom(float a, float b)
{
    /* Markstein: with y = RN(1/b) and q a faithful a/b, ONE correction q + (a - b q) y rounds to RN(a/b).  gfx950's v_rcp_f32 is
     * good to 1 ulp, and one Newton step from it IS the correctly rounded reciprocal for every one of the 2^23 significands
     * (tools/recip_check.hip, exhaustive on the device; the quotient itself against `/` on 8 x 10^8 operand pairs) -- so the
     * second correction step the compiler's expansion carries (it has to cope with scaled and subnormal operands) is not
     * needed on this domain: 6 instructions instead of 8 (round 5).  The sign of a zero quotient is not the IEEE one for
     * a = -0 (+0 comes out): wm_atan2f_tab uses |q| only. */
    float y = __builtin_amdgcn_rcpf(b);                      /* 1 ulp */
    const float e = __builtin_fmaf(-b, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    const float q = __fmul_rn(a, y);
    const float r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, y, q);
}
WM_HD float wm_sqrt_dom(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);                     /* 1 ulp */
    const float sm = wm_u2f(wm_f2u(s) - 1u), sp = wm_u2f(wm_f2u(s) + 1u);
    const float em = __builtin_fmaf(-sm, s, x), ep = __builtin_fmaf(-sp, s, x);
    s = em <= 0.0f ? sm : s;
    s = ep > 0.0f ? sp : s;
    return s;
}
#else
WM_HD float wm_div_dom(float a, float b) { return a / b; }
WM_HD float wm_sqrt_dom(float x) { return sqrtf(x); }
#endif

/* fdlibm atan2f (glibc 2.35 e_atan2f.c + s_atanf.c) for finite arguments, restated WITHOUT
 * branches: on a 64-lane wavefront the five argument-reduction ranges and the special cases
 * would otherwise all execute serially.  Every range's (numerator, denominator) pair is formed
 * and one division is selected in; the operations that reach the result are, value for value,
 * the ones the branchy original performs for that range:
 *   |t| <  7/16        r = t                (t/1 is exact)             atan = r - r*(s1+s2)
 *   7/16 .. 11/16      r = (2t-1)/(2+t)     hi/lo = atan(0.5)          atan = hi - ((r*(s1+s2) - lo) - r)
 *   11/16 .. 19/16     r = (t-1)/(t+1)      atan(1.0)
 *   19/16 .. 39/16     r = (t-1.5)/(1+1.5t) atan(1.5)
 *   >= 39/16           r = -1/t             atan(inf)
 * NaN/Inf cannot occur (the discriminator's operands are small exact rationals).
 * The x == 1.0 shortcut of e_atan2f.c (atanf(y)) yields the same bits as the general path
 * (y/1 is exact and atanf is odd), so it needs no case of its own. */
WM_HD float wm_sel(int c, float a, float b) { return c ? a : b; }

WM_HD float wm_atan2f(float y, float x)
{
    const uint32_t hx = wm_f2u(x), hy = wm_f2u(y);
    const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    const float pi = wm_u2f(0x40490fdbu), pi_o_2 = wm_u2f(0x3fc90fdbu), pi_lo = wm_u2f(0xb3bbbd2eu);
    const uint32_t m = ((hy >> 31) & 1u) | ((hx >> 30) & 2u);

    const float t = wm_u2f(wm_f2u(wm_div(y, x)) & 0x7fffffffu);       /* fabsf(y/x) */
    const uint32_t it = wm_f2u(t);
    const int r0 = it < 0x3ee00000u, r1 = it < 0x3f300000u, r2 = it < 0x3f980000u, r3 = it < 0x401c0000u;
    const float num = wm_sel(r0, t, wm_sel(r1, wm_sub(wm_mul(2.0f, t), 1.0f),
                      wm_sel(r2, wm_sub(t, 1.0f), wm_sel(r3, wm_sub(t, 1.5f), -1.0f))));
    const float den = wm_sel(r0, 1.0f, wm_sel(r1, wm_add(2.0f, t),
                      wm_sel(r2, wm_add(t, 1.0f), wm_sel(r3, wm_add(1.0f, wm_mul(1.5f, t)), t))));
    const float hi = wm_sel(r1, wm_u2f(0x3eed6338u), wm_sel(r2, wm_u2f(0x3f490fdau), wm_sel(r3, wm_u2f(0x3f7b985eu), wm_u2f(0x3fc90fdau))));
    const float lo = wm_sel(r1, wm_u2f(0x31ac3769u), wm_sel(r2, wm_u2f(0x33222168u), wm_sel(r3, wm_u2f(0x33140fb4u), wm_u2f(0x33a22168u))));
    const float r = wm_div(num, den);

    const float aT0 = wm_u2f(0x3eaaaaabu), aT1 = wm_u2f(0xbe4ccccdu), aT2 = wm_u2f(0x3e124925u),
                aT3 = wm_u2f(0xbde38e38u), aT4 = wm_u2f(0x3dba2e6eu), aT5 = wm_u2f(0xbd9d8795u),
                aT6 = wm_u2f(0x3d886b35u), aT7 = wm_u2f(0xbd6ef16bu), aT8 = wm_u2f(0x3d4bda59u),
                aT9 = wm_u2f(0xbd15a221u), aT10 = wm_u2f(0x3c8569d7u);
    const float z2 = wm_mul(r, r);
    const float w = wm_mul(z2, z2);
    float s1 = wm_add(aT8, wm_mul(w, aT10));
    s1 = wm_add(aT6, wm_mul(w, s1));
    s1 = wm_add(aT4, wm_mul(w, s1));
    s1 = wm_add(aT2, wm_mul(w, s1));
    s1 = wm_add(aT0, wm_mul(w, s1));
    s1 = wm_mul(z2, s1);
    float s2 = wm_add(aT7, wm_mul(w, aT9));
    s2 = wm_add(aT5, wm_mul(w, s2));
    s2 = wm_add(aT3, wm_mul(w, s2));
    s2 = wm_add(aT1, wm_mul(w, s2));
    s2 = wm_mul(w, s2);
    const float p = wm_mul(r, wm_add(s1, s2));
    float z = wm_sel(r0, wm_sub(r, p), wm_sub(hi, wm_sub(wm_sub(p, lo), r)));
    z = wm_sel(it < 0x31000000u, t, z);                                   /* |t| < 2^-29: atanf(t) = t   */
    z = wm_sel(it >= 0x4c000000u, wm_add(wm_u2f(0x3fc90fdau), wm_u2f(0x33a22168u)), z);   /* |t| >= 2^25 */
    /* e_atan2f.c: exponent gap shortcuts (unreachable for our operands, kept for completeness) */
    const int k = ((int)iy - (int)ix) >> 23;
    z = wm_sel(k > 60, wm_add(pi_o_2, wm_mul(0.5f, pi_lo)), wm_sel((hx >> 31) && k < -60, 0.0f, z));

    const float zl = wm_sub(z, pi_lo);
    float res = wm_sel(m == 0u, z, wm_sel(m == 1u, wm_u2f(wm_f2u(z) ^ 0x80000000u),
                wm_sel(m == 2u, wm_sub(pi, zl), wm_sub(zl, pi))));
    res = wm_sel(ix == 0u, (hy >> 31) ? -pi_o_2 : pi_o_2, res);             /* x == +-0            */
    res = wm_sel(iy == 0u, m < 2u ? y : (m == 2u ? pi : -pi), res);         /* y == +-0 comes first */
    return res;
}

/* ---------------------------------------------------------------------------------------------
 * Table-driven form of the same function for the kernels' hot loop.
 *
 * The five argument-reduction ranges of s_atanf.c differ only in constants:
 *     num = A*t + B,  den = C*t + D,  r = num/den,  atan = hi - ((r*(s1+s2) - lo) - r)
 *   range            A   B     C    D    hi/lo
 *   t <  7/16        1   0     0    1    0 / 0        (hi - ((p - 0) - r) == r - p bit for bit)
 *   7/16 .. 11/16    2  -1     1    2    atan(0.5)
 *   11/16 .. 19/16   1  -1     1    1    atan(1.0)
 *   19/16 .. 39/16   1  -1.5   1.5  1    atan(1.5)
 *   >= 39/16         0  -1     1    0    atan(inf)
 * A*t is exact for A in {0,1,2} and C*t is the product the original forms (1.5*t) or exact, so
 * num and den carry exactly the roundings of the branchy code.  All four range limits are
 * multiples of 2^18 in the float's bit pattern, so `bits(t) >> 18` indexes a row table directly
 * (79 distinct prefixes between 7/16 and 39/16 plus everything below and everything above): a
 * byte LUT gives the range, one 32-byte row fetch gives its constants; together they replace 4
 * compares, 14 selects and the four (num, den) pairs.
 *
 * Reachable-domain pruning (documented, checked by tests/exact_math_check.c on the discriminator's
 * operand domain and on the device by wmbus_selftest_math): the discriminator's operands are
 * integers (sums of <= 16 samples of magnitude <= 180, products of two such) times a common
 * power of two, so for non-zero y and x:  2^-24 < |y/x| < 2^24.  The |t| < 2^-29, |t| >= 2^25 and
 * exponent-gap (> 60) shortcuts of the original can therefore not trigger and are dropped; zero
 * operands keep their exact special cases (signed zeros matter: atan2f(-0, x<0) = -pi).
 * ------------------------------------------------------------------------------------------- */
#define WM_ATAN_RANGES    5
#define WM_ATAN_ROW_WORDS 8
#define WM_ATAN_LUT_BYTES 81
#define WM_ATAN_U0        0xFB8u      /* bits(7/16) >> 18 */
#define WM_ATAN_TAB_WORDS (WM_ATAN_RANGES * WM_ATAN_ROW_WORDS + 24)   /* 5 rows + 81-byte range LUT, padded */

/* Table = 5 rows {A, B, C, D, hi, lo, 0, 0} followed by an 81-byte LUT: range of prefix
 * j = clamp((bits(t) >> 18) - (U0 - 1), 0, 80).  On the device it lives in LDS: the LUT's bytes
 * share 21 dwords in distinct banks and the 5 rows sit 32 bytes apart, so neither fetch can
 * conflict whatever the lanes' arguments are (lanes reading the same row are a broadcast). */
WM_HD void wm_atan_tab_word(int k, float *tab)        /* fills word k of the table, k < WM_ATAN_TAB_WORDS */
{
    const float A[5] = {1.0f, 2.0f, 1.0f, 1.0f, 0.0f}, B[5] = {0.0f, -1.0f, -1.0f, -1.5f, -1.0f};
    const float C[5] = {0.0f, 1.0f, 1.0f, 1.5f, 1.0f}, D[5] = {1.0f, 2.0f, 1.0f, 1.0f, 0.0f};
    const uint32_t hi[5] = {0u, 0x3eed6338u, 0x3f490fdau, 0x3f7b985eu, 0x3fc90fdau};
    const uint32_t lo[5] = {0u, 0x31ac3769u, 0x33222168u, 0x33140fb4u, 0x33a22168u};
    if (k < WM_ATAN_RANGES * WM_ATAN_ROW_WORDS) {
        const int r = k >> 3, c = k & 7;
        tab[k] = c == 0 ? A[r] : c == 1 ? B[r] : c == 2 ? C[r] : c == 3 ? D[r] : c == 4 ? wm_u2f(hi[r]) : c == 5 ? wm_u2f(lo[r]) : 0.0f;
    } else {
        uint32_t w = 0;
        for (int b = 0; b < 4; b++) {
            const int j = 4 * (k - WM_ATAN_RANGES * WM_ATAN_ROW_WORDS) + b;
            const uint32_t u = WM_ATAN_U0 - 1u + (uint32_t)j;
            const uint32_t idx = j == 0 ? 0u : j > 80 ? 4u
                : 1u + (u >= (0x3f300000u >> 18)) + (u >= (0x3f980000u >> 18)) + (u >= (0x401c0000u >> 18));
            w |= idx << (8 * b);
        }
        tab[k] = wm_u2f(w);
    }
}

/* The same table as bit patterns, for a kernel that wants it with ONE coalesced load (word k by lane k) instead of
 * computing it: the branchy generator above, inlined at the head of the demodulation kernel, cost the first wave of every
 * block six global loads one after the other (the constant arrays A .. lo), each behind an s_waitcnt vmcnt(0), before
 * the block's input loads were even issued (round 5, read off the ISA).  tests/exact_math_check.c holds the two against
 * each other word for word. */
#if defined(__HIPCC__)
__device__
#endif
static const uint32_t WM_ATAN_TAB_BITS[WM_ATAN_TAB_WORDS] = {
    0x3f800000u, 0x00000000u, 0x00000000u, 0x3f800000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u,
    0x40000000u, 0xbf800000u, 0x3f800000u, 0x40000000u, 0x3eed6338u, 0x31ac3769u, 0x00000000u, 0x00000000u,
    0x3f800000u, 0xbf800000u, 0x3f800000u, 0x3f800000u, 0x3f490fdau, 0x33222168u, 0x00000000u, 0x00000000u,
    0x3f800000u, 0xbfc00000u, 0x3fc00000u, 0x3f800000u, 0x3f7b985eu, 0x33140fb4u, 0x00000000u, 0x00000000u,
    0x00000000u, 0xbf800000u, 0x3f800000u, 0x00000000u, 0x3fc90fdau, 0x33a22168u, 0x00000000u, 0x00000000u,
    0x01010100u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x02020201u, 0x02020202u, 0x02020202u,
    0x02020202u, 0x02020202u, 0x02020202u, 0x03020202u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u,
    0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x04040404u, 0x04040404u, 0x04040404u, 0x04040404u};

WM_HD float wm_copysign_bits(float mag, uint32_t sign_src) { return wm_u2f((wm_f2u(mag) & 0x7fffffffu) | (sign_src & 0x80000000u)); }

WM_HD float wm_atan2f_tab(float y, float x, const float *tab)
{
    const uint32_t hx = wm_f2u(x), hy = wm_f2u(y);
    const float pi = wm_u2f(0x40490fdbu), pi_o_2 = wm_u2f(0x3fc90fdbu), pi_lo = wm_u2f(0xb3bbbd2eu);
    const float qt = wm_div_dom(y, x);
    const float t = wm_u2f(wm_f2u(qt) & 0x7fffffffu);                     /* fabsf(y/x) */
    int j = (int)((wm_f2u(qt) >> 18) & 0x1FFFu) - (int)(WM_ATAN_U0 - 1u);  /* exponent and five mantissa bits in one bit-field extract */
    j = j < 0 ? 0 : (j > WM_ATAN_LUT_BYTES - 1 ? WM_ATAN_LUT_BYTES - 1 : j);
    const uint32_t idx = ((const uint8_t *)(tab + WM_ATAN_RANGES * WM_ATAN_ROW_WORDS))[j];
    const float *e = tab + WM_ATAN_ROW_WORDS * idx;
    const float num = wm_fma_exact(e[0], t, e[1]);                        /* A in {0, 1, 2}: A*t is exact */
    const float den = wm_add(wm_mul(e[2], t), e[3]);
    const float r = wm_div_dom(num, den);

    const float aT0 = wm_u2f(0x3eaaaaabu), aT1 = wm_u2f(0xbe4ccccdu), aT2 = wm_u2f(0x3e124925u),
                aT3 = wm_u2f(0xbde38e38u), aT4 = wm_u2f(0x3dba2e6eu), aT5 = wm_u2f(0xbd9d8795u),
                aT6 = wm_u2f(0x3d886b35u), aT7 = wm_u2f(0xbd6ef16bu), aT8 = wm_u2f(0x3d4bda59u),
                aT9 = wm_u2f(0xbd15a221u), aT10 = wm_u2f(0x3c8569d7u);
    const float z2 = wm_mul(r, r);
    const float w = wm_mul(z2, z2);
    float s1 = wm_add(aT8, wm_mul(w, aT10));
    s1 = wm_add(aT6, wm_mul(w, s1));
    s1 = wm_add(aT4, wm_mul(w, s1));
    s1 = wm_add(aT2, wm_mul(w, s1));
    s1 = wm_add(aT0, wm_mul(w, s1));
    s1 = wm_mul(z2, s1);
    float s2 = wm_add(aT7, wm_mul(w, aT9));
    s2 = wm_add(aT5, wm_mul(w, s2));
    s2 = wm_add(aT3, wm_mul(w, s2));
    s2 = wm_add(aT1, wm_mul(w, s2));
    s2 = wm_mul(w, s2);
    const float p = wm_mul(r, wm_add(s1, s2));
    const float z = wm_sub(e[4], wm_sub(wm_sub(p, e[5]), r));             /* atanf(|y/x|) >= 0 */
    /* quadrant (e_atan2f.c): x >= 0: +-z ; x < 0: +-(pi - (z - pi_lo)) ; sign of y */
    const float zq = (hx >> 31) ? wm_sub(pi, wm_sub(z, pi_lo)) : z;
    float res = wm_copysign_bits(zq, hy);
    /* Special cases of e_atan2f.c.  y == +-0 with x != 0 needs none here: t = 0 takes the first range's row (A = 1, B = 0, C = 0,
     * D = 1, hi = lo = 0) to z = +0 exactly, so the quadrant step returns +-0 for x > 0 and +-fl(pi - 8.74e-8) = +-pi for x < 0 --
     * what the original's `case 0 / 1: return y; case 2: return pi + tiny; case 3: return -pi - tiny` returns.  Only x == +-0
     * (the quotient is not a number then) has to be overridden: +-pi/2, or for y == +-0 too what the sign of x says. */
    const float at0 = (hy << 1) != 0u ? pi_o_2 : ((hx >> 31) ? pi : 0.0f);
    res = (hx << 1) == 0u ? wm_copysign_bits(at0, hy) : res;
    return res;
}

/* Polar discriminator on the table form.  The operands may carry any common power-of-two scale
 * (the kernels pass the boxcar SUMS, not sums/8): products and sums stay exact integers, the
 * signs of zero products are those of the reference's operands, and y/x is scale-free. */
WM_HD float wm_discriminator_tab(float i, float q, float pi_, float pq_, const float *tab)
{
    /* re = i c - q d, im = i d + q c with (c, d) = (i', -q'): every product an exact integer (sums of at most sixteen samples
     * of magnitude <= 180: below 2^24), so each component is one product and one fused multiply-add */
    const float c = pi_, d = -pq_;
    const float re = wm_fma_exact(i, c, -wm_mul(q, d));
    const float im = wm_fma_exact(i, d, wm_mul(q, c));
    return wm_mul(wm_atan2f_tab(im, re, tab), wm_u2f(0x3ea2f983u));    /* (float)M_1_PI */
}

/* TOLERANCE MODE (wmbus_cfg.tolerance_mode = 1, never the default; BASELINE north_star: "demodulated soft symbols within a
 * stated float tolerance").  The same discriminator with a polynomial arctangent: atan(r) / pi for r = min / max in [0, 1]
 * as r * P(r^2), P of degree 6 fitted to 1.1e-7 (absolute, in the discriminator's units of pi radians; 2.4e-7 measured
 * with the reciprocal and the float evaluation, tests/exact_math_check.c), one hardware reciprocal instead of two correctly
 * rounded divisions, fused multiply-adds.  Octant and sign
 * handling mirrors atan2f's (signed zeros included: atan2f(-0, x < 0) = -pi flips the symbol from +1 to -1, so it must
 * not be "within tolerance" of the other sign).  About 28 instructions against 71. */
WM_HD float wm_discriminator_tol(float i, float q, float pi_, float pq_)
{
    const float c = pi_, d = -pq_;
    const float re = wm_fma_exact(i, c, -wm_mul(q, d));                 /* exact integers either way */
    const float im = wm_fma_exact(i, d, wm_mul(q, c));
    const uint32_t hx = wm_f2u(re), hy = wm_f2u(im);
    const float ax = wm_u2f(hx & 0x7fffffffu), ay = wm_u2f(hy & 0x7fffffffu);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = mn * __builtin_amdgcn_rcpf(mx > 1e-30f ? mx : 1e-30f);
    float p = __builtin_fmaf(0.002168180188164115f, r * r, -0.010696296580135822f);
    const float s = r * r;
    p = __builtin_fmaf(p, s, 0.02534468285739422f);
    p = __builtin_fmaf(p, s, -0.04212284833192825f);
    p = __builtin_fmaf(p, s, 0.06305018067359924f);
    p = __builtin_fmaf(p, s, -0.1060524731874466f);
    p = __builtin_fmaf(p, s, 0.31830865144729614f);
#else
    const float r = mn / (mx > 1e-30f ? mx : 1e-30f);
    const float s = r * r;
    float p = fmaf(0.002168180188164115f, s, -0.010696296580135822f);
    p = fmaf(p, s, 0.02534468285739422f);
    p = fmaf(p, s, -0.04212284833192825f);
    p = fmaf(p, s, 0.06305018067359924f);
    p = fmaf(p, s, -0.1060524731874466f);
    p = fmaf(p, s, 0.31830865144729614f);
#endif
    float z = p * r;                                                    /* atan(min / max) / pi in [0, 1/4] */
    z = ay > ax ? 0.5f - z : z;
    z = (hx >> 31) ? 1.0f - z : z;
    return wm_copysign_bits(z, hy);
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

/* The two approximations the reference keeps behind `#elif 0` / `#else` (atan2.h:14-74), operation by
 * operation; y = imaginary, x = real part of s * conj(s_prev).  Both only use ratios of x and y (and
 * comparisons with zero), so they may be fed the unscaled boxcar sums like the exact version: every
 * product and sum of those is an exactly representable integer, and the 1e-10 "kludge" of the first one
 * only matters at y = 0.  Options (wmbus_cfg.atan_mode), never the default. */
WM_HD float wm_atan2_approx1(float y, float x)
{
    const float oneqtr_pi = (float)(3.14159265358979323846 / 4.0), thrqtr_pi = (float)(3.0 * 3.14159265358979323846 / 4.0);
    const float m_1_pi = wm_u2f(0x3ea2f983u);                                  /* (float)M_1_PI */
    const float abs_y = (float)((double)wm_u2f(wm_f2u(y) & 0x7fffffffu) + (double)1e-10f);   /* fabs() is the double function there */
    const int neg = x < 0.0f;
    const float r = neg ? wm_div(wm_add(x, abs_y), wm_sub(abs_y, x)) : wm_div(wm_sub(x, abs_y), wm_add(x, abs_y));
    float angle = neg ? thrqtr_pi : oneqtr_pi;
    const float p = wm_sub(wm_mul(wm_mul(wm_mul(0.1963f, m_1_pi), r), r), wm_mul(0.9817f, m_1_pi));
    angle = wm_add(angle, wm_mul(p, r));
    return y < 0.0f ? -angle : angle;
}

WM_HD float wm_atan2_approx2(float y, float x)
{
    const float m_pi = (float)3.14159265358979323846, m_1_pi = wm_u2f(0x3ea2f983u);
    if (x == 0.0f) return y > 0.0f ? 0.5f : y == 0.0f ? 0.0f : -0.5f;
    const float z = wm_div(y, x);
    if (wm_u2f(wm_f2u(z) & 0x7fffffffu) < 1.0f) {
        const float at = wm_div(z, wm_add(wm_mul(1.0f, m_pi), wm_mul(wm_mul(wm_mul(0.28086f, m_pi), z), z)));
        return x < 0.0f ? (y < 0.0f ? wm_sub(at, 1.0f) : wm_add(at, 1.0f)) : at;
    }
    const float at = wm_sub(0.5f, wm_mul(wm_div(z, wm_add(wm_mul(z, z), 0.28086f)), m_1_pi));
    return y < 0.0f ? wm_sub(at, 1.0f) : at;
}

WM_HD float wm_discriminator_approx(float i, float q, float pi_, float pq_, int which)
{
    const float c = pi_, d = -pq_;
    const float re = wm_sub(wm_mul(i, c), wm_mul(q, d));
    const float im = wm_add(wm_mul(i, d), wm_mul(q, c));
    return which == 1 ? wm_atan2_approx1(im, re) : wm_atan2_approx2(im, re);
}

/* -a variant (rtl_wmbus.c:536-551 / 572-586). */
WM_HD float wm_discriminator_fast(float i, float q, float pi_, float pq_)
{
    return wm_sub(wm_mul(pi_, q), wm_mul(i, pq_));
}

/* Level of the recovered clock = (y * gain >= 0) with gain = 1.874981046e-06f (iir.h:74, rtl_wmbus.c:338,353,
 * 1089).  The product is only ever compared with zero, and it is >= 0 exactly when y is not below
 * -266669 * 2^-149 (the largest negative y whose product rounds to -0; found by running the multiply
 * over every subnormal y on the host, tests/test_exact_math.py).  On the bit pattern this is one
 * carry: bits(y) + WM_LEVEL_CARRY overflows 32 bits <=> level low. */
#define WM_LEVEL_CARRY 0x7FFBEE52u      /* 0xFFFFFFFF - (0x80000000 + 266669) */
WM_HD int wm_level_high(float y) { return (uint64_t)wm_f2u(y) + WM_LEVEL_CARRY <= 0xFFFFFFFFull; }
/* acc = (acc << 1) | (level LOW), for a word that collects the clock levels of a block: the carry of
 * the add goes straight into an add-with-carry on the device. */
#if defined(__HIP_DEVICE_COMPILE__)
WM_HD uint32_t wm_shift_in_level_low(uint32_t acc, uint32_t ybits)
{
    uint32_t tmp;
    asm("v_add_co_u32 %1, vcc, %3, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
        : "+v"(acc), "=&v"(tmp) : "v"(ybits), "s"(WM_LEVEL_CARRY) : "vcc");
    return acc;
}
#else
WM_HD uint32_t wm_shift_in_level_low(uint32_t acc, uint32_t ybits)
{
    return (acc << 1) | (uint32_t)(((uint64_t)ybits + WM_LEVEL_CARRY) >> 32);
}
#endif

/* cu8 sample -> boxcar input (rtl_wmbus.c:1312-1313 then the int parameter of mavgi,
 * moving_average_filter.h:47): (int)((float)u8 - 127.5f), truncation toward zero. */
WM_HD int wm_quantise(unsigned u8) { return u8 >= 128u ? (int)u8 - 128 : (int)u8 - 127; }

#endif /* WM_EXACT_H */
