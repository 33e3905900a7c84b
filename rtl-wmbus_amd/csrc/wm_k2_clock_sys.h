/* wm_k2_clock_sys.h -- K2 clock recovery + time2 framer, SYSTOLIC form (round 6): a lane group's cascade on the four waves of a block.
 * Device code, included by wm_kernels.hip (one translation unit, see the overview there).
 *
 * What a lane computes, what it reads and what it leaves in memory is wm_k2_clock.h's clock_segment / clock_lanes to the word
 * (start / end state records, checkpoints, chips, counts, slicer words, flags): the two forms are interchangeable launch by launch,
 * cfg.clock_waves picks one, and the host emulation runs them against each other.  What changes is WHO computes: lane l of a block
 * is four threads, one in each wave, and wave r carries role r of wm_k2_sys_blocks.h for all 64 lanes.
 *
 * Step s of a block, all four waves (b0: the step at which role 0 takes block 0 of the lane's segment):
 *     control   every thread learns its lane's control word of step s-1 (start a segment / lane finished)
 *     phase 1   role 0 reads the soft symbols of block s - b0 from the block's LDS rows, roles 1 and 2 the 32 values their
 *               predecessors left for blocks s - b0 - 1 / - 2, role 3 the sample mask and slicer word of block s - b0 - 3
 *     barrier
 *     phase 2   roles 0 .. 2 compute and write their 32 values (role 2: the block's sample mask); role 3 stages the soft symbols of
 *               block s - b0 + 1 (loaded two steps ago) into the rows, asks memory for block s - b0 + 3, and does the chips,
 *               slicer words, every record in memory and the lane's control word of block s - b0 - 3
 *     barrier
 * A lane's blocks are 32 samples; a ragged tail (< 32 samples at the end of a row) is role 3's alone, sample by sample, from the
 * lane state it has assembled anyway for the end record.  Lane state travels through LDS snapshots: after a block that ends a warm-up,
 * a checkpoint interval or the segment, roles 0 .. 2 leave their words for role 3, which meets them at its own step for that block.
 *
 * Lanes of a block need not march together (a re-run list mixes segments, lanes leave at checkpoints, walk chains): everything above is
 * per lane (b0, the segment's geometry, "has a block at this step"), only the two barriers and the loop's exit are the block's.
 * The cooperative load of the first pass (wave = 64 consecutive captures of one segment) is role 3's.
 *
 * Resources: 37.4 KB LDS and at most 128 VGPRs for 256 threads -- a block takes the place of ONE 512-thread block of the
 * demodulation kernel's first pass (35.2 KB, 64 VGPRs) plus part of the 22.6 KB four of those leave free on a CU.  (Round 6 measured the
 * alternative with double-buffered hops, one barrier per step, 72 KB: 15 % faster alone and no faster than the one-wave form beside a
 * demodulation-shaped background, because its blocks wait for two neighbouring holes: tools/clkbench.hip.) */
#ifndef WM_K2_CLOCK_SYS_H
#define WM_K2_CLOCK_SYS_H

#include "wm_k2_sys_blocks.h"

struct ClkSysLds {
    float x[64 * WM_CLK_XROW];               /* role 3 -> role 0: a block of soft symbols, one row per lane (transposed here when loaded cooperatively) */
    float hop[2][WM_SYS_HOP_WORDS];          /* role 0 -> role 1 -> role 2 */
    uint32_t chip[64 * WM_CLK_CROW];         /* role 3: chips waiting for a whole 32-byte group -- a ring of 16 per lane, chip n of a segment at [lane][n & 15]
                                                (rows of 17 words: the lanes' 4-byte accesses are bank-conflict free); a half that fills up leaves as it lies */
    uint32_t bits[8][64];                    /* role 3: slicer words waiting for a whole 32-byte group */
    uint32_t bitw[4][64];                    /* role 0 -> role 3: the slicer word of block b in slot b & 3 */
    uint32_t smask[64];                      /* role 2 -> role 3: the sample mask of the block role 2 has just done */
    uint32_t snap[2][9][64];                 /* roles 0 .. 2 -> role 3: state words after a block (slot 1: the segment's last block) */
    uint32_t start[9][64];                   /* role 3 -> roles 0 .. 2: state words a segment starts from */
    uint32_t ctl[2][64];                     /* role 3 -> all: control word of step s in slot s & 1 */
};
/* word order of snap / start: role 0: h1, h2 of section 0, dc_x, dc_y; role 1: h1, h2 of section 1; role 2: h1, h2 of section 2, clk */

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
 * b0 + nb + 5 -- so after step 0 nobody reads or writes control words (an LDS round trip at the top of every step of every wave: 280 of
 * 2 900 cycles per step, tools/sysbench.hip). */
template <int PASS>
__device__ __forceinline__ uint32_t sys_control(const ClkSysLds &lds, uint32_t step, uint32_t ln, bool valid, uint32_t b0, uint32_t nb)
{
    if (PASS == 0 && step != 0u) return valid && step - b0 == (nb ? nb + 6u : 1u) ? (uint32_t)WM_SYS_DONE : (uint32_t)WM_SYS_NONE;
    return lds.ctl[(step + 1u) & 1u][ln];
}

/* PASS 0: the speculative first pass (every lane one segment), 1: a re-run list.  The lanes of chunk `group` (64 list entries / lane ids).
 * COOP (first pass of a batch of whole waves only): a wave is 64 consecutive captures of one (chain, segment) in lock step and loads
 * their soft symbols cooperatively -- a compile-time choice: as per-lane values the segment's geometry and the filter coefficients cost
 * every role some 50 vector instructions per step. */
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
         * this step" and the filter coefficients live in scalar registers and the steps' branches are scalar branches */
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

    wm_sys_barrier();                                     /* the previous chunk's last control words have been read */

    if (role == 0u) {
        /* ================= role 0: soft symbols -> [DC remover] -> slicer bits -> square -> section 0 ================= */
        float h1 = 0.0f, h2 = 0.0f, dcx = 0.0f, dcy = 0.0f;
        const float *xrow = lds.x + ln * WM_CLK_XROW;
        float *hop_out = lds.hop[0] + 4u * ln;
        wm_sys_barrier();                                   /* role 3 has set up every lane's first segment */
        WM_SYS_T0();
        for (uint32_t step = 0;; step++) {
            uint32_t cw = sys_control<PASS>(lds, step, ln, valid, b0, G.nb);
            if (COOP) cw = wm_uniform(cw);
            if (cw == WM_SYS_START || cw == WM_SYS_NEXT) {
                if (cw == WM_SYS_NEXT) { seg++; G = sys_geo(a, rerun, ch, stream, seg); }
                h1 = wm_u2f(lds.start[0][ln]); h2 = wm_u2f(lds.start[1][ln]); dcx = wm_u2f(lds.start[2][ln]); dcy = wm_u2f(lds.start[3][ln]);
                b0 = step; active = true;
            } else if (cw == WM_SYS_DONE) { active = false; finished = true; }
            if (__ballot(!finished) == 0ull) break;             /* the same answer in all four waves: the lanes' flags come from the control words */
            WM_SYS_MARK(0);
            const uint32_t b = step - b0 - 3u;               /* (role 1 asks for block 0 in the step the start word arrives, and stages it two steps on) */
            const bool has = active && b < G.nb;
            wm_f4 x[8];
            if (has) {
#pragma unroll
                for (int q = 0; q < 8; q++) x[q] = *(const wm_f4 *)(xrow + 4 * q);       /* staged by role 1 a step ago */
            }
            wm_sys_barrier();
            WM_SYS_MARK(1);
            if (has) {
                const uint32_t m = G.m0 + 32u * b;
                uint32_t bitw;
                if (sys_warm_short(G, m)) sys_r0_block32<DC, true>(h1, h2, dcx, dcy, c, x, hop_out, bitw);
                else sys_r0_block32<DC, false>(h1, h2, dcx, dcy, c, x, hop_out, bitw);
                lds.bitw[b & 3u][ln] = bitw;
                const bool last = b + 1u == G.nb;
                if (last || sys_snap0(G, m, nck)) {
                    uint32_t (*sn)[64] = lds.snap[last ? 1 : 0];
                    sn[0][ln] = wm_f2u(h1); sn[1][ln] = wm_f2u(h2); sn[2][ln] = wm_f2u(dcx); sn[3][ln] = wm_f2u(dcy);
                }
            }
            WM_SYS_MARK(2);
            wm_sys_barrier();
            WM_SYS_MARK(3);
        }
        WM_SYS_DUMP();
    } else if (role == 1u) {
        /* ================= role 1: section 1 + the soft symbols' way from memory into the rows ================= */
        float h1 = 0.0f, h2 = 0.0f;
        const float *hop_in = lds.hop[0] + 4u * ln;
        float *hop_out = lds.hop[1] + 4u * ln;
        /* ---- the LOADS.  Roles 1 and 3 take turns: role 1 at even steps, role 3 at odd ones.  At its step s a loader wave puts its
         * register set into the rows (the block it asked for at step s - 2: block s - 2 - b0 of the lane's walk, clamped into it) and asks
         * for block s - b0; role 0 takes a block the step after it reached the rows, i.e. three steps after it was asked for.  So a wave
         * has ONE set in flight, asked for two steps before it is used, and its wait for the set is a wait for everything the wave has
         * in flight -- which is what the compiler makes of any wait in these loops anyway (with both sets in one wave it drained the
         * set asked for a step ago as well: one block of lookahead, a step as long as a load's latency).  Unconditional for every lane:
         * lanes without a segment load row 0.  Cooperative view (COOP): lane ln fetches piece ln % 8 of row (row0 + 8 i + ln / 8),
         * i = 0 .. 7, rows of the wave consecutive: eight addresses = a UNIFORM base (row0 + 8 i, the sample index: scalar registers)
         * + one 32-bit lane offset. */
        wm_f4 gx[8];
        const float *xown = a.dphi + G.row * g.Mcap;        /* (lanes without a segment load too: row 0) */
        uint64_t crow0 = wm_uniform64(G.row - ln);          /* first row of the wave (uniform) */
        const uint32_t coff = (ln >> 3) * g.Mcap + 4u * (ln & 7u);
        uint32_t m_last = G.me_full >= 32u ? G.me_full - 32u : 0u;
        const uint32_t xw = coop ? (ln >> 3) * WM_CLK_XROW + 4u * (ln & 7u) : ln * WM_CLK_XROW, xw_step = coop ? 8u * WM_CLK_XROW : 4u;
        auto fetch = [&](uint32_t mm) WM_LAMBDA_INLINE {
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
        auto stage = [&]() WM_LAMBDA_INLINE {                 /* fetched block -> the lanes' rows (role 0 took the rows' previous block before the barrier) */
            if (coop) __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; i++) *(wm_f4 *)(lds.x + xw + i * xw_step) = gx[i];
            if (coop) __builtin_amdgcn_wave_barrier();
        };
        auto loader_new_segment = [&]() WM_LAMBDA_INLINE {   /* G has changed */
            xown = a.dphi + G.row * g.Mcap;
            crow0 = wm_uniform64(G.row - ln);
            m_last = G.me_full >= 32u ? G.me_full - 32u : 0u;
        };
        auto loader_step = [&](uint32_t step) WM_LAMBDA_INLINE { stage(); fetch(G.m0 + 32u * min(step - b0, G.nb ? G.nb - 1u : 0u)); };
        wm_sys_barrier();                                   /* role 3 has set up every lane's first segment */
        fetch(0u);                                          /* the set starts out defined */
        WM_SYS_T0();
        for (uint32_t step = 0;; step++) {
            /* the input block is asked for before the control word is looked at (one LDS round trip, not two): a lane that is told to
             * start or to stop has no block of its own in this step */
            const uint32_t b = step - b0 - 4u;
            const bool has0 = active && b < G.nb;
            wm_f4 in[8];
            if (has0) sys_hop_read(hop_in, in);
            uint32_t cw = sys_control<PASS>(lds, step, ln, valid, b0, G.nb);
            if (COOP) cw = wm_uniform(cw);
            if (cw == WM_SYS_START || cw == WM_SYS_NEXT) {
                if (cw == WM_SYS_NEXT) { seg++; G = sys_geo(a, rerun, ch, stream, seg); }
                h1 = wm_u2f(lds.start[4][ln]); h2 = wm_u2f(lds.start[5][ln]);
                b0 = step; active = true;
                loader_new_segment();
            } else if (cw == WM_SYS_DONE) { active = false; finished = true; }
            if (__ballot(!finished) == 0ull) break;             /* the same answer in all four waves: the lanes' flags come from the control words */
            WM_SYS_MARK(0);
            const bool has = has0 && cw == WM_SYS_NONE;
            wm_sys_barrier();
            WM_SYS_MARK(1);
            if ((step & 1u) == 0u) loader_step(step);
            if (has) {
                const uint32_t m = G.m0 + 32u * b;
                sys_r1_block32(h1, h2, c, in, hop_out);
                const bool last = b + 1u == G.nb;
                if (last || sys_snap0(G, m, nck)) {
                    uint32_t (*sn)[64] = lds.snap[last ? 1 : 0];
                    sn[4][ln] = wm_f2u(h1); sn[5][ln] = wm_f2u(h2);
                }
            }
            WM_SYS_MARK(2);
            wm_sys_barrier();
            WM_SYS_MARK(3);
        }
        WM_SYS_DUMP();
    } else if (role == 2u) {
        /* ================= role 2: section 2, level, clock lock ================= */
        float h1 = 0.0f, h2 = 0.0f;
        uint32_t clk = 0;
        const float *hop_in = lds.hop[1] + 4u * ln;
        wm_sys_barrier();                                   /* role 3 has set up every lane's first segment */
        WM_SYS_T0();
        for (uint32_t step = 0;; step++) {
            const uint32_t b = step - b0 - 5u;
            const bool has0 = active && b < G.nb;
            wm_f4 in[8];
            if (has0) sys_hop_read(hop_in, in);
            uint32_t cw = sys_control<PASS>(lds, step, ln, valid, b0, G.nb);
            if (COOP) cw = wm_uniform(cw);
            if (cw == WM_SYS_START || cw == WM_SYS_NEXT) {
                if (cw == WM_SYS_NEXT) { seg++; G = sys_geo(a, rerun, ch, stream, seg); }
                h1 = wm_u2f(lds.start[6][ln]); h2 = wm_u2f(lds.start[7][ln]); clk = lds.start[8][ln];
                b0 = step; active = true;
            } else if (cw == WM_SYS_DONE) { active = false; finished = true; }
            if (__ballot(!finished) == 0ull) break;
            WM_SYS_MARK(0);
            const bool has = has0 && cw == WM_SYS_NONE;
            wm_sys_barrier();
            WM_SYS_MARK(1);
            if (has) {
                const uint32_t m = G.m0 + 32u * b;
                uint32_t smask;
                if (sys_warm_short(G, m)) sys_r2_block32<true>(h1, h2, clk, c, in, smask); else sys_r2_block32<false>(h1, h2, clk, c, in, smask);
                lds.smask[ln] = smask;
                const bool last = b + 1u == G.nb;
                if (last || sys_snap0(G, m, nck)) {
                    uint32_t (*sn)[64] = lds.snap[last ? 1 : 0];
                    sn[6][ln] = wm_f2u(h1); sn[7][ln] = wm_f2u(h2); sn[8][ln] = clk;
                }
            }
            WM_SYS_MARK(2);
            wm_sys_barrier();
            WM_SYS_MARK(3);
        }
        WM_SYS_DUMP();
    } else {
        /* ================= role 3: time2 chips, slicer words, every record in memory ================= */
        /* of the lane state only the time2 shift register (and the padding words, as loaded) lives here between records: the rest is
         * gathered from the other roles' snapshots where a record is written -- a whole WmClkState across the loop is 12 of 128 VGPRs */
        uint32_t sr = 0, pad0 = 0, pad1 = 0;
        const bool t2a = g.flags & WM_F_T2A;
        const uint32_t syncw = ch ? WM_SYNC_S1 : WM_SYNC_T1C1, syncm = ch ? WM_SYNC_S1_MASK : WM_SYNC_T1C1_MASK;
        uint32_t *my_chip = lds.chip + ln * WM_CLK_CROW;
        auto ring = [&](uint32_t n) WM_LAMBDA_INLINE -> uint32_t & { return my_chip[n & 15u]; };
        uint32_t *out = a.chips, *ck = a.ckpt, *bw = a.bits;
        uint32_t n_fl = 0, pend = 0, saw_sync = 0;
        /* ---- the LOADS (as in role 1, which see: role 3 at odd steps).  */
        wm_f4 gx[8];
        const float *xown = a.dphi + G.row * g.Mcap;        /* (lanes without a segment load too: row 0) */
        uint64_t crow0 = wm_uniform64(G.row - ln);          /* first row of the wave (uniform) */
        const uint32_t coff = (ln >> 3) * g.Mcap + 4u * (ln & 7u);
        uint32_t m_last = G.me_full >= 32u ? G.me_full - 32u : 0u;
        const uint32_t xw = coop ? (ln >> 3) * WM_CLK_XROW + 4u * (ln & 7u) : ln * WM_CLK_XROW, xw_step = coop ? 8u * WM_CLK_XROW : 4u;
        auto fetch = [&](uint32_t mm) WM_LAMBDA_INLINE {
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
        auto stage = [&]() WM_LAMBDA_INLINE {                 /* fetched block -> the lanes' rows (role 0 took the rows' previous block before the barrier) */
            if (coop) __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; i++) *(wm_f4 *)(lds.x + xw + i * xw_step) = gx[i];
            if (coop) __builtin_amdgcn_wave_barrier();
        };
        auto loader_new_segment = [&]() WM_LAMBDA_INLINE {   /* G has changed */
            xown = a.dphi + G.row * g.Mcap;
            crow0 = wm_uniform64(G.row - ln);
            m_last = G.me_full >= 32u ? G.me_full - 32u : 0u;
        };
        auto loader_step = [&](uint32_t step) WM_LAMBDA_INLINE { stage(); fetch(G.m0 + 32u * min(step - b0, G.nb ? G.nb - 1u : 0u)); };

        auto gather = [&](int slot) WM_LAMBDA_INLINE -> WmClkState {      /* the other roles' words after the block I am at + my own */
            const uint32_t (*sn)[64] = lds.snap[slot];
            WmClkState t;
            t.h[0] = wm_u2f(sn[0][ln]); t.h[1] = wm_u2f(sn[1][ln]); t.dc_x = wm_u2f(sn[2][ln]); t.dc_y = wm_u2f(sn[3][ln]);
            t.h[2] = wm_u2f(sn[4][ln]); t.h[3] = wm_u2f(sn[5][ln]); t.h[4] = wm_u2f(sn[6][ln]); t.h[5] = wm_u2f(sn[7][ln]);
            t.clk = sn[8][ln]; t.sr = sr; t.pad[0] = pad0; t.pad[1] = pad1;
            return t;
        };
        /* a segment is made ready a step before the other roles learn of it: geometry, output pointers, the start record of a walk
         * without warm-up, the start words */
        auto launch_segment = [&](const WmClkState &st) WM_LAMBDA_INLINE {
            G = sys_geo(a, rerun, ch, stream, seg);
            out = a.chips + G.sidx * g.cap[1];
            ck = a.ckpt + G.sidx * (uint64_t)nck * 16u;
            bw = a.bits + G.row * (g.Mcap / 32);
            n_fl = 0; pend = 0; saw_sync = 0;
            if (G.m0 == G.mb) stS[G.sidx] = st;
            lds.start[0][ln] = wm_f2u(st.h[0]); lds.start[1][ln] = wm_f2u(st.h[1]); lds.start[2][ln] = wm_f2u(st.dc_x); lds.start[3][ln] = wm_f2u(st.dc_y);
            lds.start[4][ln] = wm_f2u(st.h[2]); lds.start[5][ln] = wm_f2u(st.h[3]); lds.start[6][ln] = wm_f2u(st.h[4]); lds.start[7][ln] = wm_f2u(st.h[5]);
            lds.start[8][ln] = st.clk;
            sr = st.sr; pad0 = st.pad[0]; pad1 = st.pad[1];
            loader_new_segment();
        };
        /* Chips leave in whole, 32-byte aligned groups of 8 (clock_segment); n_fl is a multiple of 8: the ring's half (n_fl & 8) is the group
         * and leaves as it lies.  (Role 3 has no arithmetic to hide the read behind and no need to: it is the wave with time to spare.) */
        auto flush8 = [&]() WM_LAMBDA_INLINE {
            const uint32_t *h = my_chip + (n_fl & 8u);
            uint32_t w[8];
#pragma unroll
            for (int i = 0; i < 8; i++) w[i] = h[i];
            *(uint4 *)(out + n_fl) = make_uint4(w[0], w[1], w[2], w[3]);
            *(uint4 *)(out + n_fl + 4) = make_uint4(w[4], w[5], w[6], w[7]);
            n_fl += 8u; pend = pend > 8u ? pend - 8u : 0u;
        };
        /* The end of a lane's segment and what comes after it, in ONE place of a step (three places reach it: the last whole block, a
         * checkpoint a re-run reproduces, a segment without a whole block): clock_segment's epilogue -- ragged tail, end record, count --,
         * then clock_lanes' decision: nothing more, or (a re-run lane walking its chain) the next segment from the exact end state.
         * how: 1 = the segment ran to its end, `fin` is the state after its last whole block; 2 = it left at a checkpoint. */
        auto finish_lane = [&](uint32_t how, WmClkState &fin) WM_LAMBDA_INLINE -> uint32_t {
            if (how == 1u) {
                uint32_t n_out = n_fl + pend;
                const uint32_t cap_t2 = g.cap[1];
                if (pend) flush8();                          /* last group; slots beyond n_out are never read */
                if (G.me_full < G.me) {
                    const float *x = a.dphi + G.row * g.Mcap;
                    const uint32_t m = G.me_full;
                    uint32_t bitw = 0, smask = 0, hist = fin.clk;
                    for (uint32_t k = 0; m + k < G.me; k++) {
                        float soft;
                        const uint32_t high = clk_step(fin, c, DC, x[m + k], soft);
                        hist = ((hist << 1) | high) & 0xFu;
                        bitw |= (uint32_t)(soft >= 0.0f) << k;
                        smask |= (uint32_t)(hist == 7u) << k;
                    }
                    fin.clk = hist & 7u;
                    bw[m >> 5] = bitw;
                    while (smask) {                          /* rtl_wmbus.c:818-828 */
                        const uint32_t k = (uint32_t)__ffs((int)smask) - 1u;
                        smask &= smask - 1u;
                        const uint32_t bit = (bitw >> k) & 1u;
                        fin.sr = ((fin.sr << 1) | bit) & syncm;
                        if (t2a) {
                            const uint32_t val = bit | (fin.sr == syncw ? 2u : 0u);
                            saw_sync |= val & 2u;
                            if (n_out < cap_t2) out[n_out] = WM_CHIP_WORD(m + k - G.mb, val);
                            n_out++;
                        }
                    }
                }
                stF[G.sidx] = fin;
                a.counts[G.sidx] = min(n_out, cap_t2);
                if (saw_sync) a.sync_seen[G.sidx] = 1u;
                if (n_out > cap_t2) atomicOr(a.err, WM_ERR_CHIP_OVERFLOW);       /* cannot happen: the lock pattern takes >= 4 samples per chip */
            }
            if (!chains) return WM_SYS_DONE;
            if (how == 2u) fin = stF[G.sidx];                /* left at a checkpoint: the recorded end state was exact already */
            if (seg + 1u >= g.nseg[1]) return WM_SYS_DONE;
            {
                uint32_t fw[12];
                clk_state_words(fin, fw);
                const uint32_t *nx = (const uint32_t *)(stS + G.sidx + 1u);
                bool same = true;
#pragma unroll
                for (int i = 0; i < 12; i++) same &= nx[i] == fw[i];
                if (same) return WM_SYS_DONE;                /* the next segment started from exactly this state */
            }
            if (bad[(uint64_t)(seg + 1u) * g.S] && !bad[(uint64_t)seg * g.S]) return WM_SYS_DONE;      /* it is listed and has a lane of its own in this launch: next round */
            seg++;
            launch_segment(fin);
            return WM_SYS_NEXT;
        };

        /* ---- before step 0: the lane's first segment ---- */
        uint32_t cmd = WM_SYS_DONE;
        if (valid) {
            WmClkState st = {};
            if (rerun) st = seg ? stF[G.sidx - 1u] : stC[G.row];          /* the predecessor's end state as recorded / the carried state */
            else if (G.mb <= g.warm[ch]) st = stC[G.row];               /* exact: the walk starts at the push start */
            launch_segment(st);
            cmd = WM_SYS_START;
        }
        lds.ctl[1][ln] = cmd;
        wm_sys_barrier();

        fetch(0u);                                          /* the set starts out defined */
        WM_SYS_T0();
        for (uint32_t step = 0;; step++) {
            const uint32_t bc = step - b0 - 6u;              /* the block whose chips are due */
            const bool has0 = active && bc < G.nb;
            uint32_t smask = 0, bitw = 0;                    /* what roles 2 and 0 left for that block: nothing in the block waits for LDS */
            if (has0) { smask = lds.smask[ln]; bitw = lds.bitw[bc & 3u][ln]; }
            uint32_t cw = sys_control<PASS>(lds, step, ln, valid, b0, G.nb);
            if (COOP) cw = wm_uniform(cw);
            cmd = WM_SYS_NONE;
            uint32_t how = 0;                                /* the lane's segment ends in this step: 1 at its end, 2 at a checkpoint */
            if (cw == WM_SYS_START || cw == WM_SYS_NEXT) {
                b0 = step; active = true;
                if (G.nb == 0u) how = 1u;                     /* fewer than 32 samples: all of it is the tail, from the start state */
            } else if (cw == WM_SYS_DONE) { active = false; finished = true; }
            if (__ballot(!finished) == 0ull) break;
            WM_SYS_MARK(0);
            const bool has = has0 && cw == WM_SYS_NONE;
            wm_sys_barrier();
            WM_SYS_MARK(1);
            if ((step & 1u) == 1u) loader_step(step);
            if (has) {
                const uint32_t m = G.m0 + 32u * bc;
                const bool last = bc + 1u == G.nb;
                if (m < G.mb) {
                    /* ---- warm-up block: the shift register is kept up over the last WM_CLK_SR_WINDOW samples only (clock_segment) ---- */
                    if (!sys_warm_short(G, m)) {
                        if (WM_CLK_SR_WINDOW && G.mb - m > (uint32_t)WM_CLK_SR_WINDOW) smask = 0u;
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const bool hs = smask != 0u;
                            if (!WM_SYS_ANY(hs)) break;
                            const uint32_t k = hs ? (uint32_t)__ffs((int)smask) - 1u : 0u;
                            smask &= smask - 1u;
                            const uint32_t sr_new = ((sr << 1) | ((bitw >> k) & 1u)) & syncm;
                            sr = hs ? sr_new : sr;
                        }
                    }
                    if (m + 32u == G.mb) stS[G.sidx] = gather(0);                  /* state the segment proper starts from */
                } else {
                    /* ---- block of the segment proper: chips into the staging ring, whole groups to memory ---- */
                    uint32_t cnt = 0;
                    const uint32_t bitw0 = bitw;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const bool hs = smask != 0u;
                        if (!WM_SYS_ANY(hs)) break;                                   /* no lane of the wave has a chip left in this block */
                        const uint32_t k = hs ? (uint32_t)__ffs((int)smask) - 1u : 0u;
                        smask &= smask - 1u;
                        const uint32_t bit = (bitw0 >> k) & 1u;
                        const uint32_t sr_new = ((sr << 1) | bit) & syncm;            /* rtl_wmbus.c:818-828 */
                        sr = hs ? sr_new : sr;
                        const uint32_t val = bit | (sr_new == syncw ? 2u : 0u);
                        saw_sync |= hs ? (val & 2u) : 0u;
                        ring(n_fl + pend + i) = WM_CHIP_WORD(m + k - G.mb, val);       /* slots beyond the block's chips are rewritten (at most 7 + 7 ahead of n_fl: never a waiting chip) */
                        cnt += hs;
                    }
                    pend += t2a ? cnt : 0u;
                    if (pend >= 8u) flush8();
                    {   /* slicer words leave in aligned groups of 8 (one word per 32 samples and lane), as in clock_segment */
                        const uint32_t bi = m >> 5;
                        lds.bits[bi & 7u][ln] = bitw0;
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
                    if (last) how = 1u;
                    else if (sys_snap0(G, m, nck)) {
                        /* ---- interior checkpoint j: recorded by the first pass, met again by a re-run (clock_segment) ---- */
                        const uint32_t j = (m + 32u - G.mb) / (uint32_t)WM_CK_SAMPLES - 1u;
                        uint32_t *q = ck + 16u * j;
                        uint32_t sw[12];
                        clk_state_words(gather(0), sw);
                        bool record = true;
                        if (rerun) {
                            bool same = true;
#pragma unroll
                            for (int i = 0; i < 12; i++) same &= q[i] == sw[i];
                            const uint32_t n1 = n_fl + pend, n0 = q[12];
                            if (same && n1 <= n0) {
                                /* back on the speculative pass's trajectory: everything it produced from here on is exact already.  My chips
                                 * replace its first n0; if they are fewer, its tail moves down. */
                                for (uint32_t i = 0; i < pend; i++) out[n_fl + i] = ring(n_fl + i);
                                if (n1 < n0) {
                                    const uint32_t total0 = a.counts[G.sidx];
                                    for (uint32_t i = n0; i < total0; i++) { const uint32_t w = out[i]; out[n1 + (i - n0)] = w; }
                                    a.counts[G.sidx] = n1 + (total0 - n0);
                                    for (uint32_t jj = j; jj < nck; jj++) ck[16u * jj + 12u] -= n0 - n1;      /* these checkpoints describe the tail, which has moved */
                                }
                                if (saw_sync) a.sync_seen[G.sidx] = 1u;       /* the tail's flag, if any, is already set */
                                record = false;
                                how = 2u;
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
            if (how) {
                WmClkState fin;
                if (G.nb == 0u) {                            /* the start state, as launch_segment posted it */
                    fin.h[0] = wm_u2f(lds.start[0][ln]); fin.h[1] = wm_u2f(lds.start[1][ln]); fin.dc_x = wm_u2f(lds.start[2][ln]); fin.dc_y = wm_u2f(lds.start[3][ln]);
                    fin.h[2] = wm_u2f(lds.start[4][ln]); fin.h[3] = wm_u2f(lds.start[5][ln]); fin.h[4] = wm_u2f(lds.start[6][ln]); fin.h[5] = wm_u2f(lds.start[7][ln]);
                    fin.clk = lds.start[8][ln]; fin.sr = sr; fin.pad[0] = pad0; fin.pad[1] = pad1;
                } else fin = gather(1);                      /* (how == 2: replaced by the recorded end state) */
                cmd = finish_lane(how, fin);
                active = false;
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
