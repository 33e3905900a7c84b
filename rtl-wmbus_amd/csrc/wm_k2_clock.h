/* wm_k2_clock.h -- K2 clock recovery + time2 framer lanes.
 * Device code, included by wm_kernels.hip (one translation unit, see the overview there). */
#ifndef WM_K2_CLOCK_H
#define WM_K2_CLOCK_H

/* the lambdas of a lane take the lane's register sets by reference: one of them left out of line (or inlined late) and the
 * sets live in scratch memory -- in the re-run kernel that cost every 32-sample block sixteen scratch accesses AND the prefetch
 * (a block of loads had to arrive before it could be stored away; round 5, read off the ISA) */
#if defined(__clang__)
#define WM_LAMBDA_INLINE __attribute__((always_inline))
#else
#define WM_LAMBDA_INLINE
#endif

/* the lane's register sets are NATIVE vectors: a float4 (HIP's struct type) is assigned by a 16-byte memcpy, and in the re-run
 * kernel -- one load path, no cooperative alternative -- those memcpys survived the optimiser as they were: global -> private
 * memory -> LDS, i.e. the sets lived in scratch (272 bytes per lane; r04: 576 with the states) and every block of loads had to
 * ARRIVE before it could be put away, which is the opposite of a prefetch */
typedef float wm_f4 __attribute__((vector_size(16)));

struct IirCoef { float a1[3], a2[3], b1[3], b2[3]; };

__device__ __forceinline__ IirCoef iir_coef(uint32_t ch)
{
    IirCoef c;
    if (ch == 0) { /* rtl_wmbus.c:340-341 */
        c.b1[0] = 1.999994649f; c.b2[0] = 0.9999946492f; c.b1[1] = -1.99999482f; c.b2[1] = 0.9999948196f;
        c.b1[2] = 1.703868036e-07f; c.b2[2] = -1.000010531f;
        c.a1[0] = -1.387139203f; c.a2[0] = 0.9921518712f; c.a1[1] = -1.403492665f; c.a2[1] = 0.9845934971f;
        c.a1[2] = -1.430055639f; c.a2[2] = 0.9923856172f;
    } else {       /* rtl_wmbus.c:355-356 */
        c.b1[0] = 1.999994187f; c.b2[0] = 0.9999941867f; c.b1[1] = -1.999994026f; c.b2[1] = 0.9999940262f;
        c.b1[2] = -1.605750097e-07f; c.b2[2] = -1.000011787f;
        c.a1[0] = -1.92151475f; c.a2[0] = 0.9918135499f; c.a1[1] = -1.922481015f; c.a2[1] = 0.984593497f;
        c.a1[2] = -1.937432099f; c.a2[2] = 0.9927241336f;
    }
    return c;
}

/* One sample through DC remover + squarer + 3 biquads; returns the clock level (iir.h:57-74). */
__device__ __forceinline__ bool clk_step(WmClkState &s, const IirCoef &c, bool dc, float x, float &soft)
{
    if (dc) { /* rtl_wmbus.c:501/511: (1+a)/2 * (x - x_old) + a * y_old, a = 0.999f */
        const float al = 0.999f, k = wm_div(wm_add(1.0f, al), 2.0f);
        const float y = wm_add(wm_mul(k, wm_sub(x, s.dc_x)), wm_mul(al, s.dc_y));
        s.dc_x = x; s.dc_y = y; x = y;
    }
    soft = x;
    float v = wm_mul(x, x);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float h1 = s.h[2 * k], h2 = s.h[2 * k + 1];
        const float h0 = wm_sub(v, wm_add(wm_mul(c.a1[k], h1), wm_mul(c.a2[k], h2)));
        v = wm_add(wm_add(h0, wm_mul(c.b1[k], h1)), wm_mul(c.b2[k], h2));   /* b0 == 1 */
        s.h[2 * k + 1] = h1; s.h[2 * k] = h0;
    }
    return wm_mul(v, 1.874981046e-06f) >= 0.0f;
}

/* A lane state as its twelve words, and two states compared bit for bit -- member by member: viewing the struct through a
 * uint32_t pointer makes the compiler keep it in memory (scratch) in the re-run kernel, whose chain walk carries a state from
 * one segment into the next (round 4: 576 bytes of scratch per lane, sixteen scratch accesses inside the 32-sample block loop). */
__device__ __forceinline__ void clk_state_words(const WmClkState &s, uint32_t (&w)[12])
{
#pragma unroll
    for (int i = 0; i < 6; i++) w[i] = wm_f2u(s.h[i]);
    w[6] = wm_f2u(s.dc_x); w[7] = wm_f2u(s.dc_y); w[8] = s.clk; w[9] = s.sr; w[10] = s.pad[0]; w[11] = s.pad[1];
}
__device__ __forceinline__ bool clk_state_same(const WmClkState &a, const WmClkState &b)
{
    uint32_t x[12], y[12];
    clk_state_words(a, x); clk_state_words(b, y);
    bool same = true;
#pragma unroll
    for (int i = 0; i < 12; i++) same &= x[i] == y[i];
    return same;
}

#define WM_CLK_XROW 36           /* words per lane in the clock kernel's soft-symbol buffer: 32 + 4 (rows stay 16-byte
                                    aligned; a lane's 8 ds_read_b128 are bank-conflict free: 9 L mod 16 is a permutation) */
#define WM_CLK_CROW 17           /* words per lane in its chip staging (16 + 1) */
#define WM_CLK_BROW 9            /* words per lane in its slicer-word staging (8 + 1) */

/* 32 samples through [DC remover] -> x^2 -> 3 biquads -> clock level, SOFTWARE-PIPELINED across the
 * filter sections: at tick t section k works on sample t - k, so the three (four with -o) recurrences
 * of a tick are independent instruction streams; a lone wave issues a dependent VALU operation only
 * every ~8.5 cycles on gfx950, and the straight per-sample order is one 36-deep dependent chain.
 * The compiler's scheduler would undo the interleaving (it sinks each section's recurrence into one
 * serial run over the block), so the levels of a tick are fenced with sched_barrier.  The pipeline
 * drains at the end of the block: the lane state at block boundaries is the plain sequential
 * state.  Every value is produced by exactly the operations of iir.h:57-74 / rtl_wmbus.c:497-515.
 *
 * Bits: the slicer output (soft >= 0, rtl_wmbus.c:1059) is the inverted sign bit -- a soft symbol
 * is never -0 (the FIR accumulates from +0, and +0 + -0 = +0; the DC remover's x - x_old is never
 * -0 either) -- shifted into a word with one v_alignbit; clock levels via WM_LEVEL_CARRY. */
/* WARM (round 5): a warm-up block whose chips nobody looks at (it ends more than WM_CLK_SR_WINDOW samples before the segment) --
 * the recurrences of all three sections run as ever, but what only FEEDS THE OUTPUT is left out: the last section's feed-forward
 * half and the level (6 instructions), the slicer bit (1): 20 instead of 27 per sample.  s.clk is not touched; the full blocks
 * behind it (at least WM_CLK_SR_WINDOW / 32 + 1 of them) set it. */
template <bool DC, bool WARM = false>
__device__ __forceinline__ void clk_block32(WmClkState &s, const IirCoef &c, const float *xrow, uint32_t &bitw, uint32_t &smask)
{
    /* xrow: this lane's 32 soft symbols in LDS; four are fetched every fourth tick, so the block in
     * flight and the one after it can stay in registers (two blocks of loads outstanding per lane) */
    float4 xq = {0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int P = DC ? 1 : 0;                          /* pipeline depth before the first biquad */
    float h1[3] = {s.h[0], s.h[2], s.h[4]}, h2[3] = {s.h[1], s.h[3], s.h[5]};
    float dcx = s.dc_x, dcy = s.dc_y;
    float in[3] = {0.0f, 0.0f, 0.0f};                      /* input of section k at the coming tick */
    float soft = 0.0f;                                     /* DC stage output waiting for section 0 */
    uint32_t sgn = 0, low = 0;                             /* MSB-first: sample n ends up in bit 31 - n */
    const float al = 0.999f, kk = wm_div(wm_add(1.0f, al), 2.0f);
#pragma unroll
    for (int t = 0; t < 32 + P + 2; t++) {
        float m1[3], m2[3], p1[3], p2[3], tt[3], h0[3], u[3], o[3];
        float d1 = 0.0f, d2 = 0.0f, d3 = 0.0f;
        if (t < 32 && (t & 3) == 0) xq = *(const float4 *)(xrow + t);
        const float xt = (t & 3) == 0 ? xq.x : (t & 3) == 1 ? xq.y : (t & 3) == 2 ? xq.z : xq.w;   /* sample t (t < 32) */
        /* level 1: every product that only needs last tick's state */
        if (DC && t < 32) { d1 = wm_sub(xt, dcx); d2 = wm_mul(al, dcy); }
        {   /* section 0's input: the (DC-filtered) soft symbol, squared */
            const int n0 = t - P;
            if (n0 >= 0 && n0 < 32) {
                const float sf = DC ? soft : xt;
                if (!WARM) sgn = __builtin_amdgcn_alignbit(sgn, wm_f2u(sf), 31);      /* (sgn << 1) | signbit */
                in[0] = wm_mul(sf, sf);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int n = t - P - k;
            if (n >= 0 && n < 32) {
                m1[k] = wm_mul(c.a1[k], h1[k]); m2[k] = wm_mul(c.a2[k], h2[k]);
                if (!(WARM && k == 2)) { p1[k] = wm_mul(c.b1[k], h1[k]); p2[k] = wm_mul(c.b2[k], h2[k]); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        /* level 2 */
        if (DC && t < 32) d3 = wm_mul(kk, d1);
#pragma unroll
        for (int k = 0; k < 3; k++) { const int n = t - P - k; if (n >= 0 && n < 32) tt[k] = wm_add(m1[k], m2[k]); }
        __builtin_amdgcn_sched_barrier(0);
        /* level 3 */
        if (DC && t < 32) { const float y = wm_add(d3, d2); dcx = xt; dcy = y; soft = y; }
#pragma unroll
        for (int k = 0; k < 3; k++) { const int n = t - P - k; if (n >= 0 && n < 32) h0[k] = wm_sub(in[k], tt[k]); }
        __builtin_amdgcn_sched_barrier(0);
        /* level 4, 5 */
#pragma unroll
        for (int k = 0; k < 3; k++) { const int n = t - P - k; if (n >= 0 && n < 32 && !(WARM && k == 2)) u[k] = wm_add(h0[k], p1[k]); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 3; k++) { const int n = t - P - k; if (n >= 0 && n < 32 && !(WARM && k == 2)) o[k] = wm_add(u[k], p2[k]); }
        /* hand over: section k's output is section k+1's input at the next tick */
#pragma unroll
        for (int k = 2; k >= 0; k--) {
            const int n = t - P - k;
            if (n >= 0 && n < 32) {
                h2[k] = h1[k]; h1[k] = h0[k];
                if (k < 2) in[k + 1] = o[k];
                else if (!WARM) low = wm_shift_in_level_low(low, wm_f2u(o[2]));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    s.h[0] = h1[0]; s.h[1] = h2[0]; s.h[2] = h1[1]; s.h[3] = h2[1]; s.h[4] = h1[2]; s.h[5] = h2[2];
    s.dc_x = dcx; s.dc_y = dcy;
    if (WARM) { bitw = 0u; smask = 0u; return; }
    bitw = ~__builtin_bitreverse32(sgn);
    /* clock lock (rtl_wmbus.c:1092-1111): take the bit at n iff the levels at n-3..n are L,H,H,H */
    /* WmClkState.clk keeps the last three levels with the NEWEST in bit 0; here time runs upwards */
    const uint32_t prev3 = ((s.clk & 1u) << 2) | (s.clk & 2u) | ((s.clk >> 2) & 1u);
    const uint64_t H = ((uint64_t)(~__builtin_bitreverse32(low)) << 3) | prev3;           /* bit n+3 = level at n */
    smask = (uint32_t)((~H) & (H >> 1) & (H >> 2) & (H >> 3));
    const uint32_t last3 = (uint32_t)(H >> 32) & 7u;                                        /* levels at 29, 30, 31 */
    s.clk = ((last3 & 1u) << 2) | (last3 & 2u) | ((last3 >> 2) & 1u);
}

/* Clock-recovery lane.  The reference's lock counter (rtl_wmbus.c:1092-1111: rising edge -> 1,
 * still high -> 2, third high sample -> take the bit) is equivalent to "sample at n iff the clock
 * levels at n-3..n are L,H,H,H" (checked exhaustively over all level sequences);
 * the lane state keeps the last three levels.
 *
 * Memory: a lane walks its own row (stream, chain) of soft symbols, 128 bytes per 32-sample block.
 * When the 64 lanes of the wave are 64 consecutive streams of one (chain, segment) -- n_streams a
 * multiple of 64, first pass -- the wave fetches the 64 rows' blocks COOPERATIVELY: 8 lanes per
 * row read one whole 128-byte line, and the block is transposed through LDS (conflict-free, see
 * WM_CLK_XROW).  Lane-private 16-byte loads of the same data touch 64 lines per instruction and
 * re-fetch each line from L2 several times.  Re-run launches and odd stream counts take the
 * lane-private path. */
template <int W> struct ClkLds {         /* per block: W independent waves */
    float x[W][64 * WM_CLK_XROW];
    uint32_t chip[W][64 * WM_CLK_CROW];
    uint32_t bits[W][64 * WM_CLK_BROW];
};

/* WM_CLK_WPB independent waves per block (no block-wide barrier anywhere): a block's waves land on
 * the CU's four SIMDs, so the framer loads every SIMD of the CUs it is on equally.  A lone
 * long-running wave on ONE SIMD slows every 4-wave K1 block of that CU down to the pace of the K1
 * wave that shares the SIMD with it (measured: two clock launches in flight, one wave per CU, cost
 * K1 60 %). */
/* PASS: 0 = the speculative first pass only (a.list == nullptr), 1 = a re-run list only, 2 = either (host emulation): each kind of launch has its own
 * kernel, so the first pass carries neither the list walk nor the checkpoint comparison (with both in one kernel behind
 * a grid-stride loop the first pass needed 254 VGPRs + 16 AGPRs and ran at one wave per SIMD, round 2). */
/* One segment of one (chain, capture): returns 0 when the lane ran to the segment's end (`fin` = its end state, also written to
 * st_final), 1 when a re-run left early at a checkpoint it reproduced (the end state in st_final was exact already), 2 when
 * there was nothing to do.  `from` (with have_from): the exact state a re-run starts from when the caller has it at hand (else: the
 * predecessor's record / the carried state).  By value, not by pointer: a pointer that may name the caller's `fin` kept both in scratch. */
template <bool DC, int W, int PASS>
__device__ __forceinline__ int clock_segment(const K2Args &a, ClkLds<W> &lds, const uint32_t wv, const uint32_t ln, const bool rerun,
                                             const uint32_t ch, const uint32_t stream, const uint32_t seg, const bool have_from, const WmClkState &from, WmClkState &fin)
{
    float *s_x = lds.x[wv];
    uint32_t *s_chip = lds.chip[wv], *s_bits = lds.bits[wv];
    const WmPush &g = a.g;
    const bool coop = !rerun && (g.S % 64u) == 0u;         /* wave = 64 consecutive streams, lock step */

    /* S1 lanes may span two segments (WmPush.s1_span): the odd segment rides with its even predecessor.  The lane then
     * owns both segments' chip regions, checkpoint slots and hand-off records (they are adjacent): chips and count go to
     * the even one (the odd one's count is 0), the end state to the odd one's record, and the pair (final[even],
     * start[odd]) is set to one constant so that the verifier, which knows nothing of this, sees a certified hand-off. */
    const uint32_t span = (ch == 1u && g.s1_span == 2u) ? 2u : 1u;
    if (seg % span) return 2;
    const uint64_t row = (uint64_t)ch * g.S + stream;
    const uint64_t sidx = row * g.nseg_cap[1] + seg;
    const uint32_t mb = seg * g.seg_len[1], me = min(g.M, mb + span * g.seg_len[1]);
    const uint32_t covered = (me - mb + g.seg_len[1] - 1u) / g.seg_len[1];          /* segments this lane really covers: 1 or 2 */
    const uint32_t cap_t2 = covered * g.cap[1];
    const uint32_t nck = covered == 2u ? 2u * a.nck : a.nck;                          /* checkpoint slots (one interior point of a pair has none) */
    const uint64_t sidxF = sidx + covered - 1u;                                       /* where the end state goes */
    WmClkState *stS = (WmClkState *)a.st_start, *stF = (WmClkState *)a.st_final, *stC = (WmClkState *)a.st_carry;

    WmClkState s;
    uint32_t m;
    if (rerun) { if (have_from) s = from; else s = seg ? stF[sidx - 1] : stC[row]; m = mb; }
    else {
        const uint32_t w = g.warm[ch];
        if (mb <= w) { s = stC[row]; m = 0; }            /* exact: run from the push start  */
        else { s = WmClkState{}; m = mb - w; }           /* speculative cold start          */
    }
    const IirCoef c = iir_coef(ch);
    const bool t2a = g.flags & WM_F_T2A;
    const float *x = a.dphi + row * g.Mcap;
    /* cooperative view: lane ln fetches piece ln%8 of row (8 i + ln/8), i = 0..7; rows of the wave
     * are consecutive */
    const uint64_t row0 = row - ln;
    const float *xc = a.dphi + (row0 + (ln >> 3)) * g.Mcap + 4u * (ln & 7u);
    const uint64_t xc_step = 8ull * g.Mcap;
    const uint32_t syncw = ch ? WM_SYNC_S1 : WM_SYNC_T1C1, syncm = ch ? WM_SYNC_S1_MASK : WM_SYNC_T1C1_MASK;
    uint32_t *out = a.chips + sidx * g.cap[1];            /* region pitch (cap_t2 may be two regions) */
    uint32_t *bw = a.bits + row * (g.Mcap / 32);
    uint32_t n_out = 0, saw_sync = 0;

    /* chips of one 32-sample block: walk the set bits of the sample mask (ragged tail, shift
     * register upkeep during warm-up) */
    auto emit_block = [&](uint32_t m0, uint32_t smask, uint32_t bitw, bool emit) WM_LAMBDA_INLINE {
        while (smask) {
            const uint32_t k = (uint32_t)__ffs((int)smask) - 1u;
            smask &= smask - 1u;
            const uint32_t bit = (bitw >> k) & 1u;
            s.sr = ((s.sr << 1) | bit) & syncm;                       /* rtl_wmbus.c:818-828 */
            if (emit && t2a) {
                const uint32_t val = bit | (s.sr == syncw ? 2u : 0u);
                saw_sync |= val & 2u;
                if (n_out < cap_t2) out[n_out] = WM_CHIP_WORD(m0 + k - mb, val);
                n_out++;
            }
        }
    };

    const uint32_t me_full = mb + ((me - mb) & ~31u);
    /* Two blocks of loads are kept in flight per lane (register sets A and B, used alternately):
     * with one, the kernel ran at the latency of a single 10 KB request per wave (2.8 TB/s). */
#ifndef WM_CLK_EARLY_EXIT
#define WM_CLK_EARLY_EXIT 1        /* 0: always eight trips through the chip loops of a block (the round-1 form; A/B) */
#endif
#ifndef WM_CLK_SR_WINDOW
#define WM_CLK_SR_WINDOW 1024     /* 0: shift-register upkeep over the whole warm-up (the r03 form; A/B) */
#endif
#ifndef WM_CLK_WARM_SHORT
#define WM_CLK_WARM_SHORT 1        /* 0: warm-up blocks compute their (unread) outputs too (A/B) */
#endif
#ifndef WM_CLK_PREFETCH
#define WM_CLK_PREFETCH 2          /* blocks of loads in flight per lane; 1 = build-time experiment (32 VGPRs fewer) */
#endif
    constexpr uint32_t AHEAD = 32u * WM_CLK_PREFETCH;
    wm_f4 gxA[8], gxB[8];
    auto fetch_x = [&](wm_f4 (&gx)[8], uint32_t mm) WM_LAMBDA_INLINE {
        if (coop) {
#pragma unroll
            for (int i = 0; i < 8; i++) gx[i] = *(const wm_f4 *)(xc + i * xc_step + mm);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) gx[i] = *(const wm_f4 *)(x + mm + 4 * i);
        }
    };
    /* registers -> LDS rows (coop: the pieces I fetched for other lanes' rows; else my own row) */
    const uint32_t xw = coop ? (ln >> 3) * WM_CLK_XROW + 4u * (ln & 7u) : ln * WM_CLK_XROW;
    const uint32_t xw_step = coop ? 8u * WM_CLK_XROW : 4u;
    const float *xrow = s_x + ln * WM_CLK_XROW;
    auto put_x = [&](const wm_f4 (&gx)[8]) WM_LAMBDA_INLINE {
        __builtin_amdgcn_wave_barrier();                     /* the previous block's reads are done */
#pragma unroll
        for (int i = 0; i < 8; i++) *(wm_f4 *)(s_x + xw + i * xw_step) = gx[i];
        __builtin_amdgcn_wave_barrier();
    };
    const uint32_t m_last = me_full >= 32u ? me_full - 32u : 0u;      /* clamp for prefetches past the end */

    /* ---- phase 1: warm-up blocks [m, mb): soft symbols only; no store is issued in this loop, so
     * waiting for a block in flight never waits for anything else (gfx950's vmcnt counts loads
     * and stores in one in-order queue) --------------------------------------------------------- */
    constexpr bool warm_short = WM_CLK_WARM_SHORT != 0 && WM_CLK_SR_WINDOW != 0;
    auto warm_block = [&](wm_f4 (&gx)[8]) WM_LAMBDA_INLINE {
        put_x(gx);
        fetch_x(gx, min(m + AHEAD, m_last));
        uint32_t bitw, smask;
        if (warm_short && mb - m > (uint32_t)WM_CLK_SR_WINDOW + 32u) clk_block32<DC, true>(s, c, xrow, bitw, smask);
        else clk_block32<DC>(s, c, xrow, bitw, smask);
        /* shift-register upkeep: at most 8 chips per block, oldest first; the wave stops as soon as none of its lanes
         * has a chip left (T1/C1 lanes meet 4 per block, S1 lanes 1.3: half the trips of the fixed eight).  The register
         * is a function of the last 16 / 24 chips only, so the upkeep starts WM_CLK_SR_WINDOW samples before the segment
         * (>= 40 chips of either chain at their nominal rates; if a stretch of silence leaves fewer, the hand-off does not
         * certify and the segment is re-run, as after any other uncertified start) */
        if (WM_CLK_SR_WINDOW && mb - m > (uint32_t)WM_CLK_SR_WINDOW) smask = 0u;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const bool has = smask != 0u;
            if (WM_CLK_EARLY_EXIT && __ballot(has) == 0ull) break;
            const uint32_t k = has ? (uint32_t)__ffs((int)smask) - 1u : 0u;
            smask &= smask - 1u;
            const uint32_t sr_new = ((s.sr << 1) | ((bitw >> k) & 1u)) & syncm;
            s.sr = has ? sr_new : s.sr;
        }
        m += 32;
    };
    if (m < me_full) { fetch_x(gxA, m); if (WM_CLK_PREFETCH == 2) fetch_x(gxB, min(m + 32u, m_last)); }
    while (m < mb) {
        warm_block(gxA);
        if (WM_CLK_PREFETCH == 1) continue;
        if (m < mb) warm_block(gxB);
        else {                                               /* keep "A = next block" for phase 2 */
#pragma unroll
            for (int i = 0; i < 8; i++) { const wm_f4 t = gxA[i]; gxA[i] = gxB[i]; gxB[i] = t; }
        }
    }
    stS[sidx] = s;                                       /* state the main loop starts from */
    /* ---- phase 2: blocks of the segment proper.  Exactly three stores per block (slicer word and
     * two 16-byte chip stores; a block holds at most 8 chips because the lock pattern L,H,H,H needs 4
     * samples, and slots beyond the block's chips are overwritten by the next block), so the
     * compiler can wait for a prefetched block with a counted vmcnt instead of draining the stores. */
    /* chips leave in whole, 32-byte aligned groups of 8 (see k2_rla: partial-sector stores from
     * 131 072 lanes with private output regions become read-modify-write traffic) */
    uint32_t *my_chip = s_chip + ln * WM_CLK_CROW;
    uint32_t pend = 0, n_fl = 0;
    auto flush8 = [&]() WM_LAMBDA_INLINE {
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = my_chip[i];
        *(uint4 *)(out + n_fl) = make_uint4(w[0], w[1], w[2], w[3]);
        *(uint4 *)(out + n_fl + 4) = make_uint4(w[4], w[5], w[6], w[7]);
#pragma unroll
        for (int i = 0; i < 8; i++) { const uint32_t v = my_chip[8 + i]; if (8u + i < pend) my_chip[i] = v; }
        n_fl += 8u; pend = pend > 8u ? pend - 8u : 0u;
    };
    /* slicer words leave in aligned groups of 8 as well (one word per 32 samples and lane) */
    uint32_t *my_bits = s_bits + ln * WM_CLK_BROW;
    auto main_block = [&](wm_f4 (&gx)[8]) WM_LAMBDA_INLINE {
        put_x(gx);
        fetch_x(gx, min(m + AHEAD, m_last));
        uint32_t bitw, smask;
        clk_block32<DC>(s, c, xrow, bitw, smask);
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const bool has = smask != 0u;
            if (WM_CLK_EARLY_EXIT && __ballot(has) == 0ull) break;              /* no lane of the wave has a chip left in this block */
            const uint32_t k = has ? (uint32_t)__ffs((int)smask) - 1u : 0u;
            smask &= smask - 1u;
            const uint32_t bit = (bitw >> k) & 1u;
            const uint32_t sr_new = ((s.sr << 1) | bit) & syncm;          /* rtl_wmbus.c:818-828 */
            s.sr = has ? sr_new : s.sr;
            const uint32_t val = bit | (sr_new == syncw ? 2u : 0u);
            saw_sync |= has ? (val & 2u) : 0u;
            my_chip[pend + i] = WM_CHIP_WORD(m + k - mb, val);                /* slots beyond the block's chips are rewritten */
            cnt += has;
        }
        pend += t2a ? cnt : 0u;
        const uint32_t bi = m >> 5;
        my_bits[bi & 7u] = bitw;
        if ((bi & 7u) == 7u) {
            uint32_t w[8];
#pragma unroll
            for (int i = 0; i < 8; i++) w[i] = my_bits[i];
            *(uint4 *)(bw + (bi - 7u)) = make_uint4(w[0], w[1], w[2], w[3]);
            *(uint4 *)(bw + (bi - 3u)) = make_uint4(w[4], w[5], w[6], w[7]);
        }
        if (pend >= 8u) flush8();
        m += 32;
    };
    uint32_t *ck = a.ckpt + sidx * (uint64_t)a.nck * 16u;
    for (uint32_t j = 0; m < me_full; j++) {
        const uint32_t stop = min(me_full, m + (uint32_t)WM_CK_SAMPLES);     /* an even number of blocks, or the end */
        while (m < stop) {
            main_block(gxA);
            if (WM_CLK_PREFETCH == 2 && m < stop) main_block(gxB);
        }
        if (m < me_full && j < nck) {                    /* interior checkpoint j */
            uint32_t *q = ck + 16u * j;
            uint32_t sw[12];
            clk_state_words(s, sw);
            if (!rerun) {
                *(uint4 *)(q) = make_uint4(sw[0], sw[1], sw[2], sw[3]);
                *(uint4 *)(q + 4) = make_uint4(sw[4], sw[5], sw[6], sw[7]);
                *(uint4 *)(q + 8) = make_uint4(sw[8], sw[9], sw[10], sw[11]);
                q[12] = n_fl + pend;
            } else {
                bool same = true;
#pragma unroll
                for (int i = 0; i < 12; i++) same &= q[i] == sw[i];
                const uint32_t n1 = n_fl + pend, n0 = q[12];
                if (same && n1 <= n0) {
                    /* Back on the speculative pass's trajectory: everything it produced from here on is
                     * exact already.  My chips replace its first n0; if they are fewer, its tail moves
                     * down.  (More chips
                     * than it had: its tail is partly overwritten -- run on to the segment's end.) */
                    for (uint32_t i = 0; i < pend; i++) out[n_fl + i] = my_chip[i];
                    if (n1 < n0) {
                        const uint32_t total0 = a.counts[sidx];
                        for (uint32_t i = n0; i < total0; i++) {
                            const uint32_t w = out[i];
                            out[n1 + (i - n0)] = w;
                        }
                        a.counts[sidx] = n1 + (total0 - n0);
                        /* this and the later checkpoints describe the tail, which has moved: a later
                         * round may re-run this segment again and meet them */
                        for (uint32_t jj = j; jj < nck; jj++) ck[16u * jj + 12u] -= n0 - n1;
                    }
                    if (saw_sync) a.sync_seen[sidx] = 1u;       /* the tail's flag, if any, is already set */
                    return 1;
                }
                /* Not on the recorded trajectory: from here on the region holds MY chips (and all of it
                 * if I run to the end), so the checkpoint must describe me -- a later round that re-runs
                 * this segment once more compares against what is in memory, not against the
                 * speculative pass.  (Found by the randomised tests: two chips lost after a second
                 * round met a checkpoint whose chip count predated the first round's move.) */
                *(uint4 *)(q) = make_uint4(sw[0], sw[1], sw[2], sw[3]);
                *(uint4 *)(q + 4) = make_uint4(sw[4], sw[5], sw[6], sw[7]);
                *(uint4 *)(q + 8) = make_uint4(sw[8], sw[9], sw[10], sw[11]);
                q[12] = n1;
            }
        }
    }
    for (uint32_t bi = (m >> 5) & ~7u; bi < (m >> 5); bi++) bw[bi] = my_bits[bi & 7u];   /* incomplete last group */
    n_out = n_fl + pend;
    if (pend) flush8();                                  /* last group; slots beyond n_out are never read */
    if (m < me) {                                        /* ragged tail of the last segment */
        uint32_t bitw = 0, smask = 0, hist = s.clk;
        for (uint32_t k = 0; m + k < me; k++) {
            float soft;
            const uint32_t high = clk_step(s, c, DC, x[m + k], soft);
            hist = ((hist << 1) | high) & 0xFu;
            bitw |= (uint32_t)(soft >= 0.0f) << k;
            smask |= (uint32_t)(hist == 7u) << k;
        }
        s.clk = hist & 7u;
        bw[m >> 5] = bitw;
        emit_block(m, smask, bitw, true);
    }
    stF[sidxF] = s;
    if (covered == 2u) { const WmClkState none = {}; stF[sidx] = none; stS[sidx + 1u] = none; a.counts[sidx + 1u] = 0u; }
    a.counts[sidx] = min(n_out, cap_t2);
    if (saw_sync) a.sync_seen[sidx] = 1u;
    if (n_out > cap_t2) atomicOr(a.err, WM_ERR_CHIP_OVERFLOW);       /* cannot happen: the lock pattern takes >= 4 samples per chip */
    fin = s;
    return 0;
}

/* The lanes of one launch.  First pass: lane = (chain, segment, capture), every lane one segment.
 * Re-run list, from the SECOND list round on (round 4; K2Args.bad set): a listed lane walks its CHAIN.  A segment is listed
 * because its start did not match its predecessor's end; k2_verify also leaves that verdict per segment in `a.bad`.  The first
 * list round re-runs every listed segment on its own, in parallel, from the predecessor's end state as recorded -- right
 * unless that predecessor is itself re-run and comes out different, which is rare with whole-wave batches (fewer than ten
 * lanes of 16 384) and the rule with the short segments of a small batch, where a slowly converging stretch covers several
 * segments and every round settled one more of them (a single capture of configs[1] fell to the host-driven path on every
 * push).  In a chain walk the FIRST listed segment of a run of consecutive listed ones does them all, one after the other,
 * each from the exact end state of the one before (the others return at once), and goes on into the segment behind the run
 * as long as the end state it arrives with differs from that segment's recorded start -- unless that segment has a lane of
 * its own in this launch (listed behind an unlisted one), which the next round sorts out.  (Walking chains already in the
 * first list round made it 2.8 ms longer on the bench workload: neighbours that are both listed usually both leave at an
 * early checkpoint, and serialising them doubles the longest lane.) */
template <bool DC, int W, int PASS = 2>
__device__ __forceinline__ void clock_lanes(const K2Args &a, const uint32_t block, ClkLds<W> &lds)
{
    const uint32_t ln = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (wv >= (uint32_t)W) return;
    uint32_t lane = (block * W + wv) * 64 + ln;
    const bool rerun = PASS == 2 ? a.list != nullptr : PASS == 1;
    const WmPush &g = a.g;
    if (lane >= k2_lane_count(a)) return;
    if (rerun) lane = a.list[lane];
    uint32_t ch, stream, seg;
    lane_decode(g, 1, lane, ch, stream, seg);
    if (!(g.flags & (ch ? WM_F_S1 : WM_F_T1C1))) return;
    WmClkState fin, from{};
    const bool chains = rerun && a.bad != nullptr && !(ch == 1u && g.s1_span == 2u);
    if (!chains) { clock_segment<DC, W, PASS>(a, lds, wv, ln, rerun, ch, stream, seg, false, from, fin); return; }
    const uint64_t row = (uint64_t)ch * g.S + stream;
    const uint32_t *bad = a.bad + (uint64_t)ch * g.nseg_cap[1] * g.S + stream;       /* verdict of segment j at bad[j * S] */
    const WmClkState *stS = (const WmClkState *)a.st_start, *stF = (const WmClkState *)a.st_final;
    if (seg > 0u && bad[(uint64_t)(seg - 1u) * g.S]) return;            /* the head of my run covers me */
    bool have_from = false;
    for (;;) {
        const int how = clock_segment<DC, W, PASS>(a, lds, wv, ln, true, ch, stream, seg, have_from, from, fin);
        const uint64_t sidx = row * g.nseg_cap[1] + seg;
        if (how == 1) fin = stF[sidx];                     /* left at a checkpoint: the recorded end state was exact */
        if (how == 2 || seg + 1u >= g.nseg[1]) return;
        const WmClkState next = stS[sidx + 1u];
        if (clk_state_same(fin, next)) return;             /* the next segment started from exactly this state */
        if (bad[(uint64_t)(seg + 1u) * g.S] && !bad[(uint64_t)seg * g.S]) return;      /* it is listed and has a lane of its own in this launch: next round */
        seg++;
        from = fin; have_from = true;
    }
}

template <bool DC>
__global__ __launch_bounds__(64 * WM_CLK_WPB) void k2_clock(K2Args a)                 /* first pass: one block per 64 * WM_CLK_WPB lanes */
{
    wm_framer_prio();
    __shared__ __attribute__((aligned(16))) ClkLds<WM_CLK_WPB> lds;
    clock_lanes<DC, WM_CLK_WPB, 0>(a, blockIdx.x, lds);
}

template <bool DC>
__global__ __launch_bounds__(64 * WM_CLK_WPB) void k2_clock_list(K2Args a)            /* re-run list: a fixed grid whose blocks walk the list */
{
    wm_framer_prio();
    __shared__ __attribute__((aligned(16))) ClkLds<WM_CLK_WPB> lds;
    const uint32_t n = k2_lane_count(a);
    for (uint32_t b = blockIdx.x; (uint64_t)b * (64u * WM_CLK_WPB) < n; b += gridDim.x) clock_lanes<DC, WM_CLK_WPB, 1>(a, b, lds);
}

#endif /* WM_K2_CLOCK_H */
