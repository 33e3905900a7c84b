/* wm_k2_clock_sys.h -- K2 clock recovery + time2 framer, SYSTOLIC form (round 6): a lane group's cascade on the four waves of a block.
 * Device code, included by wm_kernels.hip (one translation unit, see the overview there).
 *
 * What a lane computes, what it reads and what it leaves in memory is wm_k2_clock.h's clock_segment / clock_lanes to the word
 * (start / end state records, checkpoints, chips, counts, slicer words, flags): the two forms are interchangeable launch by launch,
 * cfg.clock_waves picks one, and the host emulation runs them against each other.  What changes is WHO computes: lane l of a block
 * is four threads, one in each wave, and wave r carries role r of wm_k2_sys_blocks.h for all 64 lanes.
 *
 * Step s of a block, all four waves:
 *     control   every thread reads its lane's control word of step s-1 (start a segment / lane finished)
 *     phase 1   role r reads the 32 values its predecessor left for block s - b0 - r (b0: the step the lane's segment started at)
 *     barrier
 *     phase 2   role r computes, writes its 32 values (role 0 stages the next block's soft symbols, role 1 moves slicer words to
 *               memory, role 3 does the clock lock, the chips, every record in memory and the lane's control word of step s)
 *     barrier
 * A lane's blocks are 32 samples; a ragged tail (< 32 samples at the end of a row) is role 3's alone, sample by sample, from the
 * lane state it has assembled anyway for the end record.  Lane state travels through LDS snapshots: after a block that ends a warm-up,
 * a checkpoint interval or the segment, roles 0 .. 2 leave their words for role 3, which meets them three steps later.
 *
 * Lanes of a block need not march together (a re-run list mixes segments, lanes leave at checkpoints, walk chains): everything above is
 * per lane (b0, the segment's geometry, "has a block at this step"), only the two barriers and the loop's exit are the block's.
 * The cooperative load of the first pass (wave = 64 consecutive captures of one segment) is role 0's.
 *
 * Resources: 47.9 KB LDS and at most 128 VGPRs for 256 threads -- a block takes the place of ONE 512-thread block of the
 * demodulation kernel's first pass (35 KB, 64 VGPRs) plus the 20 KB four of those leave free on a CU.  (Round 6 measured the
 * alternative with double-buffered hops, one barrier per step, 72 KB: 15 % faster alone and no faster than the one-wave form beside a
 * demodulation-shaped background, because its blocks wait for two neighbouring holes: tools/clkbench.hip.) */
#ifndef WM_K2_CLOCK_SYS_H
#define WM_K2_CLOCK_SYS_H

#include "wm_k2_sys_blocks.h"

struct ClkSysLds {
    float x[64 * WM_CLK_XROW];               /* role 0: a block of soft symbols, one row per lane (transposed here when loaded cooperatively) */
    float hop[3][WM_SYS_HOP_WORDS];          /* role r -> role r + 1 */
    uint32_t chip[64 * WM_CLK_CROW];         /* role 3: chips waiting for a whole 32-byte group -- a ring of 16 per lane, chip n of a segment at [lane][n & 15]
                                                (rows of 17 words: the lanes' 4-byte accesses are bank-conflict free); a half that fills up leaves as it lies */
    uint32_t bits[8][64];                    /* role 1: slicer words waiting for a whole 32-byte group */
    uint32_t bitw[4][64];                    /* role 0 -> roles 1, 3: the slicer word of block b in slot b & 3 */
    uint32_t snap[2][8][64];                 /* roles 0 .. 2 -> role 3: state words after a block (slot 1: the segment's last block) */
    uint32_t start[8][64];                   /* role 3 -> roles 0 .. 2: state words a segment starts from */
    uint32_t ctl[2][64];                     /* role 3 -> all: control word of step s in slot s & 1 */
};
/* word order of snap / start: role 0: h1, h2 of section 0, dc_x, dc_y; role 1: h1, h2 of section 1; role 2: h1, h2 of section 2 */

enum { WM_SYS_NONE = 0, WM_SYS_START = 1, WM_SYS_NEXT = 2, WM_SYS_DONE = 3 };

/* -DWM_SYS_STAMPS (tools/sysbench.hip only): where a wave's cycles go -- control word, input reads + first barrier, work, second
 * barrier -- summed per wave into wm_sys_stamps[(block * 4 + role) * 4 ..] */
#if defined(WM_SYS_STAMPS)
__device__ unsigned long long *wm_sys_stamps;
#endif
#if defined(WM_SYS_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define WM_SYS_T0() unsigned long long st_t = __builtin_readcyclecounter(), st_acc[4] = {0, 0, 0, 0}
#define WM_SYS_MARK(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); st_acc[i] += t_ - st_t; st_t = t_; } while (0)
#define WM_SYS_DUMP() do { if (ln == 0 && wm_sys_stamps) for (int i_ = 0; i_ < 4; i_++) wm_sys_stamps[((size_t)blockIdx.x * 4 + role) * 4 + i_] = st_acc[i_]; } while (0)
#else
#define WM_SYS_T0() do {} while (0)
#define WM_SYS_MARK(i) do {} while (0)
#define WM_SYS_DUMP() do {} while (0)
#endif

/* a value every lane of the wave holds alike, moved to scalar registers */
__device__ __forceinline__ uint32_t wm_uniform(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#else
    return v;
#endif
}
__device__ __forceinline__ uint64_t wm_uniform64(uint64_t v) { return ((uint64_t)wm_uniform((uint32_t)(v >> 32)) << 32) | wm_uniform((uint32_t)v); }

#if defined(__HIP_DEVICE_COMPILE__)
#define WM_SYS_ANY(p) (__ballot(p) != 0ull)      /* "some lane of the wave still has a chip in this block": an early exit, never a decision */
#else
#define WM_SYS_ANY(p) true
#endif

struct SysGeo {
    uint64_t row, sidx;
    uint32_t mb, me, me_full;        /* the segment [mb, me); whole blocks end at me_full */
    uint32_t m0, nb;                 /* first sample of the lane's walk (warm-up start), number of whole blocks */
};

__device__ __forceinline__ SysGeo sys_geo(const K2Args &a, const bool rerun, const uint32_t ch, const uint32_t stream, const uint32_t seg)
{
    const WmPush &g = a.g;
    SysGeo G;
    G.row = (uint64_t)ch * g.S + stream; G.sidx = G.row * g.nseg_cap[1] + seg;
    G.mb = seg * g.seg_len[1]; G.me = min(g.M, G.mb + g.seg_len[1]);
    G.me_full = G.mb + ((G.me - G.mb) & ~31u);
    const uint32_t w = g.warm[ch];
    G.m0 = rerun ? G.mb : (G.mb <= w ? 0u : G.mb - w);          /* as clock_segment: exact from the push start, or a cold start w samples early */
    G.nb = (G.me_full - G.m0) >> 5;
    return G;
}

/* kinds of block m .. m + 32 of a walk (clock_segment's warm_block / main_block) */
__device__ __forceinline__ bool sys_warm_short(const SysGeo &G, uint32_t m) { return WM_CLK_WARM_SHORT != 0 && WM_CLK_SR_WINDOW != 0 && m < G.mb && G.mb - m > (uint32_t)WM_CLK_SR_WINDOW + 32u; }
/* after which blocks the lane state is recorded: the end of the warm-up (-> start record), interior checkpoints; slot 1: the last whole block */
__device__ __forceinline__ bool sys_snap0(const SysGeo &G, uint32_t m, uint32_t nck)
{
    const uint32_t mn = m + 32u;
    if (m < G.mb) return mn == G.mb;
    return mn < G.me_full && (mn - G.mb) % (uint32_t)WM_CK_SAMPLES == 0u && (mn - G.mb) / (uint32_t)WM_CK_SAMPLES - 1u < nck;
}

/* The control word a lane meets at the top of a step.  A re-run lane's comes from role 3 through LDS (it leaves at checkpoints, walks
 * chains).  In the first pass a lane's life is known in advance -- started before step 0, finished when role 3 has done its last block,
 * b0 + nb + 2 -- so after step 0 nobody reads or writes control words (an LDS round trip at the top of every step of every wave: 280 of
 * 2 900 cycles per step, tools/sysbench.hip). */
template <int PASS>
__device__ __forceinline__ uint32_t sys_control(const ClkSysLds &lds, uint32_t step, uint32_t ln, bool valid, uint32_t b0, uint32_t nb)
{
    if (PASS == 0 && step != 0u) return valid && step - b0 == (nb ? nb + 3u : 1u) ? (uint32_t)WM_SYS_DONE : (uint32_t)WM_SYS_NONE;
    return lds.ctl[(step + 1u) & 1u][ln];
}

/* PASS 0: the speculative first pass (every lane one segment), 1: a re-run list.  The lanes of chunk `group` (64 list entries / lane ids).
 * COOP (first pass of a batch of whole waves only): a wave is 64 consecutive captures of one (chain, segment) in lock step and loads
 * their soft symbols cooperatively -- a compile-time choice: the two load paths side by side cost role 0 the registers it does not have. */
template <bool DC, int PASS, bool COOP>
__device__ __forceinline__ void clock_sys_group(const K2Args &a, const uint32_t group, ClkSysLds &lds)
{
    const uint32_t ln = threadIdx.x & 63u;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#else
    const uint32_t role = threadIdx.x >> 6;
#endif
    const WmPush &g = a.g;
    constexpr bool rerun = PASS == 1;
    constexpr bool coop = COOP;
    static_assert(!(COOP && PASS == 1), "re-run lanes are not neighbours");

    /* ---- which segment is this lane's?  (every role works it out for itself) ---- */
    uint32_t lane = group * 64u + ln;
    bool valid = lane < k2_lane_count(a);
    if (valid && rerun) lane = a.list[lane];
    uint32_t ch = 0, stream = 0, seg = 0;
    if (valid) { lane_decode(g, 1, lane, ch, stream, seg); valid = (g.flags & (ch ? WM_F_S1 : WM_F_T1C1)) != 0u; }
    if (COOP) {
        /* the 64 lanes are 64 captures of ONE (chain, segment): said so, the segment's geometry, the kinds of its blocks, "has a block at
         * this step" and the filter coefficients live in scalar registers and the steps' branches are scalar branches -- as per-lane
         * values they cost every role some 50 vector instructions per step beside the 190-260 of its arithmetic */
        ch = wm_uniform(ch); seg = wm_uniform(seg); valid = wm_uniform((uint32_t)valid) != 0u;
    }
    const bool chains = rerun && a.bad != nullptr;
    const uint32_t *bad = a.bad + (uint64_t)ch * g.nseg_cap[1] * g.S + stream;       /* verdict of segment j at bad[j * S] (chains only) */
    if (valid && chains && seg > 0u && bad[(uint64_t)(seg - 1u) * g.S]) valid = false;     /* the head of my run of listed segments covers me */
    const IirCoef c = iir_coef(ch);
    const uint32_t nck = a.nck;
    WmClkState *stS = (WmClkState *)a.st_start, *stF = (WmClkState *)a.st_final, *stC = (WmClkState *)a.st_carry;

    SysGeo G = sys_geo(a, rerun, ch, stream, seg);
    uint32_t b0 = 0;
    bool active = false, finished = false;
    float *hop_in = lds.hop[role ? role - 1u : 0u] + 4u * ln, *hop_out = lds.hop[role < 3u ? role : 2u] + 4u * ln;

    wm_sys_barrier();                                     /* the previous chunk's last control words have been read */

    if (role == 0u) {
        /* ================= role 0: soft symbols -> [DC remover] -> slicer bits -> square -> feedback half of section 0 ================= */
        float h1 = 0.0f, h2 = 0.0f, dcx = 0.0f, dcy = 0.0f;
        wm_f4 gx0[8], gx1[8];                               /* the next two blocks as fetched (cooperatively: pieces of other lanes' rows); the block to compute lies in LDS */
        /* cooperative view: lane ln fetches piece ln % 8 of row (row0 + 8 i + ln / 8), i = 0 .. 7, rows of the wave consecutive: the
         * eight addresses of a block are a UNIFORM base (row0 + 8 i, the sample index: scalar registers, scalar adds) plus one
         * 32-bit lane offset -- eight hoisted 64-bit lane pointers were 16 VGPRs of a wave that has 128 */
        const float *xown = a.dphi;
        uint64_t crow0 = 0;                                 /* first row of the wave (uniform) */
        const uint32_t coff = (ln >> 3) * g.Mcap + 4u * (ln & 7u);
        uint32_t m_last = 0;
        const uint32_t xw = coop ? (ln >> 3) * WM_CLK_XROW + 4u * (ln & 7u) : ln * WM_CLK_XROW, xw_step = coop ? 8u * WM_CLK_XROW : 4u;
        auto fetch = [&](wm_f4 (&gx)[8], uint32_t mm) WM_LAMBDA_INLINE {
            mm = min(mm, m_last);
            if (coop) {
                const uint32_t mu = wm_uniform(mm);
#pragma unroll
                for (int i = 0; i < 8; i++) gx[i] = *(const wm_f4 *)(a.dphi + ((crow0 + 8u * i) * g.Mcap + mu) + coff);
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) gx[i] = *(const wm_f4 *)(xown + mm + 4 * i);
            }
        };
        wm_sys_barrier();                                   /* role 3 has set up every lane's first segment */
        const float *xrow = lds.x + ln * WM_CLK_XROW;
        auto stage = [&](const wm_f4 (&gx)[8]) WM_LAMBDA_INLINE {          /* fetched block -> the lanes' rows in LDS (the rows' previous block has been read) */
            if (coop) __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; i++) *(wm_f4 *)(lds.x + xw + i * xw_step) = gx[i];
            if (coop) __builtin_amdgcn_wave_barrier();
        };
        /* Two blocks of loads in flight, in two register sets used by STEP parity (sets that alternate by copying -- gx0 = gx1 --
         * make the copy wait for the load it moves: one block in flight, a step as long as a load's latency; measured: 1.8 instead of
         * 0.9 us per step).  At the end of step s set[s & 1] holds the lane's block s + 1 - b0 (asked for at step s - 2), is staged
         * for step s + 1 and refilled with block s + 3 - b0.  Returns true when every lane of the block has finished. */
        uint32_t step = 0;
        WM_SYS_T0();
        auto one_step = [&](wm_f4 (&mine)[8], wm_f4 (&other)[8]) WM_LAMBDA_INLINE -> bool {
            uint32_t cw = sys_control<PASS>(lds, step, ln, valid, b0, G.nb);
            if (COOP) cw = wm_uniform(cw);
            if (cw == WM_SYS_START || cw == WM_SYS_NEXT) {
                if (cw == WM_SYS_NEXT) { seg++; G = sys_geo(a, rerun, ch, stream, seg); }
                h1 = wm_u2f(lds.start[0][ln]); h2 = wm_u2f(lds.start[1][ln]); dcx = wm_u2f(lds.start[2][ln]); dcy = wm_u2f(lds.start[3][ln]);
                b0 = step; active = true;
                if (G.nb) {
                    xown = a.dphi + G.row * g.Mcap;
                    crow0 = wm_uniform64(G.row - ln);
                    m_last = G.me_full - 32u;
                    fetch(mine, G.m0); stage(mine);          /* block 0: computed in this very step */
                    fetch(mine, G.m0 + 32u); fetch(other, G.m0 + 64u);
                }
            } else if (cw == WM_SYS_DONE) { active = false; finished = true; }
            if (__ballot(!finished) == 0ull) return true;       /* the same answer in all four waves: the lanes' flags come from the control words */
            const uint32_t b = step - b0;
            const bool has = active && b < G.nb;
            /* the first half of my row (staged a step ago) into registers while the other roles read their blocks; behind the barrier:
             * first half's arithmetic, second half into registers -- the row is free --, the next block takes its place and the one
             * after the next is asked for, then the second half's arithmetic hides both */
            wm_f4 xa[4], xb[4];
#ifndef WM_SYS_R0_TAIL
#define WM_SYS_R0_TAIL 0        /* 1 (A/B): both halves and the staging behind barrier A, staging last */
#endif
            if (has && !WM_SYS_R0_TAIL) {
#pragma unroll
                for (int q = 0; q < 4; q++) xa[q] = *(const wm_f4 *)(xrow + 4 * q);
            }
            wm_sys_barrier();
            WM_SYS_MARK(0);                                  /* (role 0's accounting: control + reads + barrier A | first half | staging, loads, second half | barrier B) */
            if (has) {
                const uint32_t m = G.m0 + 32u * b;
                const bool last = b + 1u == G.nb, warm = sys_warm_short(G, m);
                uint32_t sgn = 0;
                if (WM_SYS_R0_TAIL) {
#pragma unroll
                    for (int q = 0; q < 4; q++) xa[q] = *(const wm_f4 *)(xrow + 4 * q);
                }
                if (warm) sys_r0_half16<DC, true, 0>(h1, h2, dcx, dcy, c, xa, hop_out, sgn); else sys_r0_half16<DC, false, 0>(h1, h2, dcx, dcy, c, xa, hop_out, sgn);
                if (WM_SYS_R0_TAIL) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; q++) xb[q] = *(const wm_f4 *)(xrow + 16 + 4 * q);
                WM_SYS_MARK(1);
                if (!last && !WM_SYS_R0_TAIL) { stage(mine); fetch(mine, m + 96u); }
                if (warm) sys_r0_half16<DC, true, 1>(h1, h2, dcx, dcy, c, xb, hop_out, sgn); else sys_r0_half16<DC, false, 1>(h1, h2, dcx, dcy, c, xb, hop_out, sgn);
                if (!last && WM_SYS_R0_TAIL) { __builtin_amdgcn_sched_barrier(0); stage(mine); fetch(mine, m + 96u); }
                lds.bitw[b & 3u][ln] = warm ? 0u : ~__builtin_bitreverse32(sgn);
                if (last || sys_snap0(G, m, nck)) {
                    uint32_t (*sn)[64] = lds.snap[last ? 1 : 0];
                    sn[0][ln] = wm_f2u(h1); sn[1][ln] = wm_f2u(h2); sn[2][ln] = wm_f2u(dcx); sn[3][ln] = wm_f2u(dcy);
                }
            }
            WM_SYS_MARK(2);
            wm_sys_barrier();
            WM_SYS_MARK(3);
            step++;
            return false;
        };
        for (;;) { if (one_step(gx0, gx1)) break; if (one_step(gx1, gx0)) break; }
        WM_SYS_DUMP();
    } else if (role == 1u || role == 2u) {
        /* ================= roles 1, 2: feed-forward half of section role - 1, feedback half of section role ================= */
        float g1 = 0.0f, g2 = 0.0f, h1 = 0.0f, h2 = 0.0f;
        uint32_t *bw = a.bits;
        const uint32_t so = role == 1u ? 0u : 4u, sh = role == 1u ? 4u : 6u;        /* my predecessor's section and mine in snap / start */
        wm_sys_barrier();                                   /* role 3 has set up every lane's first segment */
        WM_SYS_T0();
        for (uint32_t step = 0;; step++) {
            /* the input block is asked for before the control word is looked at (one LDS round trip, not two): a lane that is told to
             * start or to stop has no block of its own in this step */
            const uint32_t b = step - b0 - role;
            const bool has0 = active && b < G.nb;
            wm_f4 in[8];
            if (has0) sys_hop_read(hop_in, in);
            uint32_t cw = sys_control<PASS>(lds, step, ln, valid, b0, G.nb);
            if (COOP) cw = wm_uniform(cw);
            if (cw == WM_SYS_START || cw == WM_SYS_NEXT) {
                if (cw == WM_SYS_NEXT) { seg++; G = sys_geo(a, rerun, ch, stream, seg); }
                g1 = wm_u2f(lds.start[so][ln]); g2 = wm_u2f(lds.start[so + 1u][ln]); h1 = wm_u2f(lds.start[sh][ln]); h2 = wm_u2f(lds.start[sh + 1u][ln]);
                b0 = step; active = true;
                bw = a.bits + G.row * (g.Mcap / 32);
            } else if (cw == WM_SYS_DONE) { active = false; finished = true; }
            if (__ballot(!finished) == 0ull) break;             /* the same answer in all four waves: the lanes' flags come from the control words */
            WM_SYS_MARK(0);
            const bool has = has0 && cw == WM_SYS_NONE;
            wm_sys_barrier();
            WM_SYS_MARK(1);
            if (has) {
                const uint32_t m = G.m0 + 32u * b;
                if (role == 1u) sys_mid_block32<1>(g1, g2, h1, h2, c, in, hop_out); else sys_mid_block32<2>(g1, g2, h1, h2, c, in, hop_out);
                const bool last = b + 1u == G.nb;
                if (last || sys_snap0(G, m, nck)) {
                    uint32_t (*sn)[64] = lds.snap[last ? 1 : 0];
                    sn[sh][ln] = wm_f2u(h1); sn[sh + 1u][ln] = wm_f2u(h2);
                }
                if (role == 1u && m >= G.mb) {
                    /* slicer words leave in aligned groups of 8 (one word per 32 samples and lane), as in clock_segment */
                    const uint32_t bi = m >> 5;
                    lds.bits[bi & 7u][ln] = lds.bitw[b & 3u][ln];
                    if ((bi & 7u) == 7u) {
                        uint32_t w[8];
#pragma unroll
                        for (int i = 0; i < 8; i++) w[i] = lds.bits[i][ln];
                        *(uint4 *)(bw + (bi - 7u)) = make_uint4(w[0], w[1], w[2], w[3]);
                        *(uint4 *)(bw + (bi - 3u)) = make_uint4(w[4], w[5], w[6], w[7]);
                    } else if (last) {
                        for (uint32_t k = bi & ~7u; k <= bi; k++) bw[k] = lds.bits[k & 7u][ln];           /* incomplete last group */
                    }
                }
            }
            WM_SYS_MARK(2);
            wm_sys_barrier();
            WM_SYS_MARK(3);
        }
        WM_SYS_DUMP();
    } else {
        /* ================= role 3: feed-forward half of section 2, level, clock lock, time2 chips, every record in memory ================= */
        float g1 = 0.0f, g2 = 0.0f;
        WmClkState s = {};                                  /* the assembled lane state (clk, sr, pad live here all the time) */
        const bool t2a = g.flags & WM_F_T2A;
        const uint32_t syncw = ch ? WM_SYNC_S1 : WM_SYNC_T1C1, syncm = ch ? WM_SYNC_S1_MASK : WM_SYNC_T1C1_MASK;
        uint32_t *my_chip = lds.chip + ln * WM_CLK_CROW;
        auto ring = [&](uint32_t n) WM_LAMBDA_INLINE -> uint32_t & { return my_chip[n & 15u]; };
        uint32_t *out = a.chips, *ck = a.ckpt;
        uint32_t n_fl = 0, pend = 0, saw_sync = 0;
        auto post_start = [&]() WM_LAMBDA_INLINE {            /* s -> the other roles' start words and my own registers */
            lds.start[0][ln] = wm_f2u(s.h[0]); lds.start[1][ln] = wm_f2u(s.h[1]); lds.start[2][ln] = wm_f2u(s.dc_x); lds.start[3][ln] = wm_f2u(s.dc_y);
            lds.start[4][ln] = wm_f2u(s.h[2]); lds.start[5][ln] = wm_f2u(s.h[3]); lds.start[6][ln] = wm_f2u(s.h[4]); lds.start[7][ln] = wm_f2u(s.h[5]);
            g1 = s.h[4]; g2 = s.h[5];
        };
        auto gather = [&](int slot) WM_LAMBDA_INLINE {        /* the other roles' words after the block I have just done -> s */
            const uint32_t (*sn)[64] = lds.snap[slot];
            s.h[0] = wm_u2f(sn[0][ln]); s.h[1] = wm_u2f(sn[1][ln]); s.dc_x = wm_u2f(sn[2][ln]); s.dc_y = wm_u2f(sn[3][ln]);
            s.h[2] = wm_u2f(sn[4][ln]); s.h[3] = wm_u2f(sn[5][ln]); s.h[4] = g1; s.h[5] = g2;
        };
        auto begin_segment = [&]() WM_LAMBDA_INLINE {         /* geometry, output pointers, the start record of a walk without warm-up */
            G = sys_geo(a, rerun, ch, stream, seg);
            out = a.chips + G.sidx * g.cap[1];
            ck = a.ckpt + G.sidx * (uint64_t)nck * 16u;
            n_fl = 0; pend = 0; saw_sync = 0;
            if (G.m0 == G.mb) stS[G.sidx] = s;
        };
        /* Chips leave in whole, 32-byte aligned groups of 8 (clock_segment); n_fl is a multiple of 8.  A group that fills up in the middle of
         * a segment is only READ here (fl_w); its two stores are issued at the top of the lane's next block (put_group), when the words have
         * long arrived -- read-then-store on the spot was a full LDS round trip in every block of a T1/C1 wave, whose 64 lanes between
         * them fill a group nearly every block. */
        uint32_t fl_w[8], fl_at = 0xFFFFFFFFu;
        auto put_group = [&]() WM_LAMBDA_INLINE {
            if (fl_at != 0xFFFFFFFFu) {
                *(uint4 *)(out + fl_at) = make_uint4(fl_w[0], fl_w[1], fl_w[2], fl_w[3]);
                *(uint4 *)(out + fl_at + 4) = make_uint4(fl_w[4], fl_w[5], fl_w[6], fl_w[7]);
                fl_at = 0xFFFFFFFFu;
            }
        };
        auto take_group = [&]() WM_LAMBDA_INLINE {
            const uint32_t *h = my_chip + (n_fl & 8u);
#pragma unroll
            for (int i = 0; i < 8; i++) fl_w[i] = h[i];
            fl_at = n_fl;
            n_fl += 8u; pend = pend > 8u ? pend - 8u : 0u;
        };
        auto flush8 = [&]() WM_LAMBDA_INLINE { put_group(); take_group(); put_group(); };
        /* the end of a segment (its last whole block done, or none to do): ragged tail, end record, count -- clock_segment's epilogue */
        auto end_segment = [&]() WM_LAMBDA_INLINE {
            uint32_t n_out = n_fl + pend;
            const uint32_t cap_t2 = g.cap[1];
            put_group();
            if (pend) flush8();                              /* last group; slots beyond n_out are never read */
            if (G.me_full < G.me) {
                const float *x = a.dphi + G.row * g.Mcap;
                const uint32_t m = G.me_full;
                uint32_t bitw = 0, smask = 0, hist = s.clk;
                for (uint32_t k = 0; m + k < G.me; k++) {
                    float soft;
                    const uint32_t high = clk_step(s, c, DC, x[m + k], soft);
                    hist = ((hist << 1) | high) & 0xFu;
                    bitw |= (uint32_t)(soft >= 0.0f) << k;
                    smask |= (uint32_t)(hist == 7u) << k;
                }
                s.clk = hist & 7u;
                a.bits[G.row * (g.Mcap / 32) + (m >> 5)] = bitw;
                while (smask) {                              /* rtl_wmbus.c:818-828 */
                    const uint32_t k = (uint32_t)__ffs((int)smask) - 1u;
                    smask &= smask - 1u;
                    const uint32_t bit = (bitw >> k) & 1u;
                    s.sr = ((s.sr << 1) | bit) & syncm;
                    if (t2a) {
                        const uint32_t val = bit | (s.sr == syncw ? 2u : 0u);
                        saw_sync |= val & 2u;
                        if (n_out < cap_t2) out[n_out] = WM_CHIP_WORD(m + k - G.mb, val);
                        n_out++;
                    }
                }
            }
            stF[G.sidx] = s;
            a.counts[G.sidx] = min(n_out, cap_t2);
            if (saw_sync) a.sync_seen[G.sidx] = 1u;
            if (n_out > cap_t2) atomicOr(a.err, WM_ERR_CHIP_OVERFLOW);       /* cannot happen: the lock pattern takes >= 4 samples per chip */
        };
        /* what comes after a segment of this lane (clock_lanes): nothing, or -- a re-run lane walking its chain -- the next segment from
         * the exact end state in s.  `early`: the lane left at a checkpoint, the recorded end state was exact already. */
        auto after_segment = [&](bool early) WM_LAMBDA_INLINE -> uint32_t {
            if (!chains) return WM_SYS_DONE;
            if (early) s = stF[G.sidx];
            if (seg + 1u >= g.nseg[1]) return WM_SYS_DONE;
            const WmClkState next = stS[G.sidx + 1u];
            if (clk_state_same(s, next)) return WM_SYS_DONE;              /* the next segment started from exactly this state */
            if (bad[(uint64_t)(seg + 1u) * g.S] && !bad[(uint64_t)seg * g.S]) return WM_SYS_DONE;      /* it is listed and has a lane of its own in this launch: next round */
            seg++;
            begin_segment();
            post_start();
            return WM_SYS_NEXT;
        };

        /* ---- before step 0: the lane's first segment ---- */
        uint32_t cmd = WM_SYS_DONE;
        if (valid) {
            if (rerun) s = seg ? stF[G.sidx - 1u] : stC[G.row];          /* the predecessor's end state as recorded / the carried state */
            else if (G.mb <= g.warm[ch]) s = stC[G.row];                /* exact: the walk starts at the push start */
            begin_segment();
            post_start();
            cmd = WM_SYS_START;
        }
        lds.ctl[1][ln] = cmd;
        wm_sys_barrier();

        WM_SYS_T0();
        for (uint32_t step = 0;; step++) {
            const uint32_t b = step - b0 - 3u;
            const bool has0 = active && b < G.nb;
            wm_f4 in[8];
            uint32_t bitw = 0;                               /* the block's slicer word comes with its input: nothing in the block waits for LDS */
            if (has0) { sys_hop_read(hop_in, in); bitw = lds.bitw[b & 3u][ln]; }
            uint32_t cw = sys_control<PASS>(lds, step, ln, valid, b0, G.nb);
            if (COOP) cw = wm_uniform(cw);
            cmd = WM_SYS_NONE;
            if (cw == WM_SYS_START || cw == WM_SYS_NEXT) {
                b0 = step; active = true;
                if (G.nb == 0u) { end_segment(); cmd = after_segment(false); active = false; }       /* fewer than 32 samples: all of it is the tail */
            } else if (cw == WM_SYS_DONE) { active = false; finished = true; }
            if (__ballot(!finished) == 0ull) break;             /* the same answer in all four waves: the lanes' flags come from the control words */
            WM_SYS_MARK(0);
            const bool has = has0 && cw == WM_SYS_NONE;
            wm_sys_barrier();
            WM_SYS_MARK(1);
            if (has) {
                const uint32_t m = G.m0 + 32u * b;
                const bool last = b + 1u == G.nb;
                if (m < G.mb) {
                    /* ---- warm-up block: the shift register is kept up over the last WM_CLK_SR_WINDOW samples only (clock_segment) ---- */
                    if (sys_warm_short(G, m)) { g2 = in[7][2]; g1 = in[7][3]; }
                    else {
                        uint32_t smask;
                        sys_r3_block32(g1, g2, s.clk, c, in, smask);
                        if (WM_CLK_SR_WINDOW && G.mb - m > (uint32_t)WM_CLK_SR_WINDOW) smask = 0u;
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const bool hs = smask != 0u;
                            if (!WM_SYS_ANY(hs)) break;
                            const uint32_t k = hs ? (uint32_t)__ffs((int)smask) - 1u : 0u;
                            smask &= smask - 1u;
                            const uint32_t sr_new = ((s.sr << 1) | ((bitw >> k) & 1u)) & syncm;
                            s.sr = hs ? sr_new : s.sr;
                        }
                    }
                    if (m + 32u == G.mb) { gather(0); stS[G.sidx] = s; }           /* state the segment proper starts from */
                } else {
                    /* ---- block of the segment proper: chips into the staging row, whole groups to memory ---- */
                    uint32_t smask;
                    sys_r3_block32(g1, g2, s.clk, c, in, smask);
                    uint32_t cnt = 0;
                    put_group();                             /* the group that filled up a block ago */
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const bool hs = smask != 0u;
                        if (!WM_SYS_ANY(hs)) break;                                   /* no lane of the wave has a chip left in this block */
                        const uint32_t k = hs ? (uint32_t)__ffs((int)smask) - 1u : 0u;
                        smask &= smask - 1u;
                        const uint32_t bit = (bitw >> k) & 1u;
                        const uint32_t sr_new = ((s.sr << 1) | bit) & syncm;          /* rtl_wmbus.c:818-828 */
                        s.sr = hs ? sr_new : s.sr;
                        const uint32_t val = bit | (sr_new == syncw ? 2u : 0u);
                        saw_sync |= hs ? (val & 2u) : 0u;
                        ring(n_fl + pend + i) = WM_CHIP_WORD(m + k - G.mb, val);       /* slots beyond the block's chips are rewritten (at most 7 + 7 ahead of n_fl: never a waiting chip) */
                        cnt += hs;
                    }
                    pend += t2a ? cnt : 0u;
                    if (pend >= 8u) take_group();
                    if (last) { gather(1); end_segment(); cmd = after_segment(false); active = false; }
                    else if (sys_snap0(G, m, nck)) {
                        /* ---- interior checkpoint j: recorded by the first pass, met again by a re-run (clock_segment) ---- */
                        const uint32_t j = (m + 32u - G.mb) / (uint32_t)WM_CK_SAMPLES - 1u;
                        gather(0);
                        uint32_t *q = ck + 16u * j;
                        uint32_t sw[12];
                        clk_state_words(s, sw);
                        bool record = true;
                        if (rerun) {
                            bool same = true;
#pragma unroll
                            for (int i = 0; i < 12; i++) same &= q[i] == sw[i];
                            const uint32_t n1 = n_fl + pend, n0 = q[12];
                            if (same && n1 <= n0) {
                                /* back on the speculative pass's trajectory: everything it produced from here on is exact already.  My chips
                                 * replace its first n0; if they are fewer, its tail moves down. */
                                put_group();
                                for (uint32_t i = 0; i < pend; i++) out[n_fl + i] = ring(n_fl + i);
                                if (n1 < n0) {
                                    const uint32_t total0 = a.counts[G.sidx];
                                    for (uint32_t i = n0; i < total0; i++) { const uint32_t w = out[i]; out[n1 + (i - n0)] = w; }
                                    a.counts[G.sidx] = n1 + (total0 - n0);
                                    for (uint32_t jj = j; jj < nck; jj++) ck[16u * jj + 12u] -= n0 - n1;      /* these checkpoints describe the tail, which has moved */
                                }
                                if (saw_sync) a.sync_seen[G.sidx] = 1u;       /* the tail's flag, if any, is already set */
                                record = false;
                                cmd = after_segment(true); active = false;
                            }
                        }
                        if (record) {                        /* the first pass -- or a re-run off the recorded trajectory: from here on the region holds ITS chips */
                            *(uint4 *)(q) = make_uint4(sw[0], sw[1], sw[2], sw[3]);
                            *(uint4 *)(q + 4) = make_uint4(sw[4], sw[5], sw[6], sw[7]);
                            *(uint4 *)(q + 8) = make_uint4(sw[8], sw[9], sw[10], sw[11]);
                            q[12] = n_fl + pend;
                        }
                    }
                }
            }
            if (PASS != 0) lds.ctl[step & 1u][ln] = cmd;
            WM_SYS_MARK(2);
            wm_sys_barrier();
            WM_SYS_MARK(3);
        }
        WM_SYS_DUMP();
    }
}

#if defined(__HIPCC__)
/* The block's LDS is DYNAMIC (sizeof(ClkSysLds) at the launch): with a static 47.9 KB the compiler works out that at most three blocks
 * fit a CU, drops the request for four waves per SIMD as unachievable and allocates 200+ VGPRs -- and a wave that wide needs the
 * registers of TWO demodulation blocks to leave before it can start.  What matters is not how many clock blocks fit a CU (one,
 * rarely two) but that one fits wherever a demodulation block has just left: 128 VGPRs. */
extern __shared__ __attribute__((aligned(16))) unsigned char wm_sys_lds[];

template <bool DC, bool COOP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k2_clock_sys(K2Args a)                 /* first pass: one block per 64 lanes; COOP: a.g.S % 64 == 0 */
{
    wm_framer_prio();
    clock_sys_group<DC, 0, COOP>(a, blockIdx.x, *(ClkSysLds *)wm_sys_lds);
}

template <bool DC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k2_clock_sys_list(K2Args a)            /* re-run list: a fixed grid whose blocks walk the list, 64 entries at a time */
{
    wm_framer_prio();
    const uint32_t n = k2_lane_count(a);
    for (uint32_t b = blockIdx.x; (uint64_t)b * 64u < n; b += gridDim.x) clock_sys_group<DC, 1, false>(a, b, *(ClkSysLds *)wm_sys_lds);
}

#endif /* __HIPCC__ */

#endif /* WM_K2_CLOCK_SYS_H */
