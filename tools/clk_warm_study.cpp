/* clk_warm_study.cpp -- host-only study (no GPU): how often does a clock-recovery lane's speculative start certify,
 * and where does a re-run meet the speculative trajectory again, as a function of the warm-up SCHEME.
 *
 * The clock kernel (rtl-wmbus_amd/csrc/wm_k2_clock.h) starts a segment's lane W samples early from the all-zero
 * state and runs the reference's recurrence (iir.h:49-77 driven from rtl_wmbus.c:1089-1111) with the reference's
 * roundings; the hand-off is certified bitwise against the predecessor's end state and re-run if it differs, so the
 * RESULT never depends on the warm-up -- only the number of re-runs does.  Scheme "fma": the first W - E samples of the
 * warm-up run the same recurrence with fused multiply-adds (13 instead of 31 operations per sample, other roundings),
 * the last E samples exactly.  This file measures the certification rate and the re-run lengths of both on soft symbols
 * handed in by tools/clk_warm_study.py (oracle taps of the bench workload's captures). */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

struct Coef { float a1[3], a2[3], b1[3], b2[3]; };
static Coef coef(int ch)
{
    Coef c;
    if (ch == 0) {
        c.b1[0] = 1.999994649f; c.b2[0] = 0.9999946492f; c.b1[1] = -1.99999482f; c.b2[1] = 0.9999948196f;
        c.b1[2] = 1.703868036e-07f; c.b2[2] = -1.000010531f;
        c.a1[0] = -1.387139203f; c.a2[0] = 0.9921518712f; c.a1[1] = -1.403492665f; c.a2[1] = 0.9845934971f;
        c.a1[2] = -1.430055639f; c.a2[2] = 0.9923856172f;
    } else {
        c.b1[0] = 1.999994187f; c.b2[0] = 0.9999941867f; c.b1[1] = -1.999994026f; c.b2[1] = 0.9999940262f;
        c.b1[2] = -1.605750097e-07f; c.b2[2] = -1.000011787f;
        c.a1[0] = -1.92151475f; c.a2[0] = 0.9918135499f; c.a1[1] = -1.922481015f; c.a2[1] = 0.984593497f;
        c.a1[2] = -1.937432099f; c.a2[2] = 0.9927241336f;
    }
    return c;
}

struct St { float h[6]; uint32_t clk; };

static inline bool step_exact(St &s, const Coef &c, float x)
{
    float v = x * x;
    for (int k = 0; k < 3; k++) {
        const float h1 = s.h[2 * k], h2 = s.h[2 * k + 1];
        const float m1 = c.a1[k] * h1, m2 = c.a2[k] * h2, p1 = c.b1[k] * h1, p2 = c.b2[k] * h2;
        const float t = m1 + m2;
        const float h0 = v - t;
        const float u = h0 + p1;
        v = u + p2;
        s.h[2 * k + 1] = h1; s.h[2 * k] = h0;
    }
    const float y = v * 1.874981046e-06f;
    return y >= 0.0f;
}

static inline void step_fma(St &s, const Coef &c, float x)
{
    float v = x * x;
    for (int k = 0; k < 3; k++) {
        const float h1 = s.h[2 * k], h2 = s.h[2 * k + 1];
        const float h0 = fmaf(-c.a2[k], h2, fmaf(-c.a1[k], h1, v));
        v = fmaf(c.b2[k], h2, fmaf(c.b1[k], h1, h0));
        s.h[2 * k + 1] = h1; s.h[2 * k] = h0;
    }
}

/* the first nsec sections only (the others keep their zero state): a STAGGERED warm-up switches section 1 on A samples and
 * section 2 A + B samples behind section 0 -- a later section's input is only true once the sections before it have settled */
static inline void step_sections(St &s, const Coef &c, float x, int nsec)
{
    float v = x * x;
    for (int k = 0; k < nsec; k++) {
        const float h1 = s.h[2 * k], h2 = s.h[2 * k + 1];
        const float m1 = c.a1[k] * h1, m2 = c.a2[k] * h2, p1 = c.b1[k] * h1, p2 = c.b2[k] * h2;
        const float t = m1 + m2;
        const float h0 = v - t;
        const float u = h0 + p1;
        v = u + p2;
        s.h[2 * k + 1] = h1; s.h[2 * k] = h0;
    }
}

static inline bool same(const St &a, const St &b) { return std::memcmp(a.h, b.h, sizeof a.h) == 0 && a.clk == b.clk; }

extern "C" {

/* x: M soft symbols of one (capture, chain).  For every segment boundary mb = j * seg (j >= 1, mb > W): the speculative state at
 * mb under the scheme (E = exact tail; E >= W: the plain exact warm-up) against the exact one.
 * leave[]: histogram over ck-sample checkpoints behind mb at which a re-run from the exact state meets the speculative
 * trajectory again (index 0 = certified at mb, i = after i checkpoints, n_leave - 1 = not within the segment).
 * Returns the number of boundaries looked at. */
int clk_warm_study(const float *x, uint32_t M, int ch, uint32_t seg, uint32_t W, uint32_t E, uint32_t ck, uint32_t *leave, uint32_t n_leave)
{
    const Coef c = coef(ch);
    /* the exact trajectory, its state kept at every boundary and checkpoint */
    const uint32_t nck = M / ck + 1;
    std::vector<St> ex(nck);
    St s{};
    for (uint32_t m = 0; m < M; m++) {
        if (m % ck == 0) ex[m / ck] = s;
        const bool hi = step_exact(s, c, x[m]);
        s.clk = ((s.clk << 1) | (hi ? 1u : 0u)) & 7u;
    }
    int n = 0;
    for (uint32_t mb = seg; mb + seg <= M; mb += seg) {
        if (mb <= W) continue;
        St q{};
        uint32_t m = mb - W;
        const uint32_t m_ex = E >= (1u << 30) ? m : E >= W ? m : mb - E;
        if (E >= (1u << 30)) {                                   /* staggered: E = 2^30 | A << 15 | B (A, B in units of 1 sample, < 32768) */
            const uint32_t A = (E >> 15) & 0x7FFFu, B = E & 0x7FFFu;
            for (; m < mb - W + A; m++) step_sections(q, c, x[m], 1);
            for (; m < mb - W + A + B; m++) step_sections(q, c, x[m], 2);
        } else
        for (; m < m_ex; m++) step_fma(q, c, x[m]);
        for (; m < mb; m++) { const bool hi = step_exact(q, c, x[m]); q.clk = ((q.clk << 1) | (hi ? 1u : 0u)) & 7u; }
        uint32_t i = 0;
        for (;; i++) {
            if (same(q, ex[m / ck])) break;
            if (i + 1 >= n_leave - 1 || m + ck > mb + seg) { i = n_leave - 1; break; }
            for (uint32_t e = m + ck; m < e; m++) { const bool hi = step_exact(q, c, x[m]); q.clk = ((q.clk << 1) | (hi ? 1u : 0u)) & 7u; }
        }
        leave[i]++;
        n++;
    }
    return n;
}

}
