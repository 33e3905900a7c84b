/* wm_k2_rla.h -- K2 run-length framer lanes.  Uses no wave-level or float intrinsics beyond __ffs and one reciprocal, so
 * tests/emu compiles this file for the host and runs it lane by lane against the oracle.
 * Device code, included by wm_kernels.hip (one translation unit, see the overview there). */
#ifndef WM_K2_RLA_H
#define WM_K2_RLA_H

/* Exact truncating division for |a| < 2^24, 4 <= b < 2^12 via the hardware's APPROXIMATE reciprocal and a +-1 fix-up (the
 * integer divide is ~40 instructions and sits on the serial path of every edge; round 3 used the correctly rounded reciprocal,
 * itself a ten-instruction sequence -- three divisions were 130 of the ~310 instructions of an edge trip).
 * Why +-1 suffices: v_rcp_f32 is within 1 ulp, so rcp = (1 / b)(1 + e1), |e1| <= 2^-23; the product rounds once more,
 * |e2| <= 2^-24; ua converts exactly.  |estimate - ua / b| <= (ua / b) * 1.5 * 2^-23 < (2^24 / b) * 1.5 * 2^-23 = 3 / b <= 0.75
 * for b >= 4: the truncated estimate is q - 1, q or q + 1, and the remainder test repairs it.  Divisors below 4 (a bit-length
 * tracker dragged to nothing by an interferer) and operands beyond the bounds take the integer divide.  The result is an exact
 * integer either way, so the host emulation (1.0f / x) and the device agree whatever the reciprocals' last bits. */
#if defined(__HIP_DEVICE_COMPILE__)
#define WM_RCP_APPROX(x) __builtin_amdgcn_rcpf(x)
#else
#define WM_RCP_APPROX(x) (1.0f / (x))
#endif
__device__ __forceinline__ unsigned wm_udiv(unsigned ua, unsigned b)          /* ua / b for non-negative operands */
{
    if (ua >= (1u << 24) || b - 4u >= (1u << 12) - 4u) return ua / b;
    unsigned q = (unsigned)((float)ua * WM_RCP_APPROX((float)b));
    const int r = (int)ua - (int)(q * b);
    q = r < 0 ? q - 1u : r >= (int)b ? q + 1u : q;
    return q;
}
__device__ __forceinline__ int wm_sdiv(int a, int b)                          /* C's truncating a / b, b > 0 */
{
    const unsigned q = wm_udiv((unsigned)(a < 0 ? -a : a), (unsigned)b);
    return a < 0 ? -(int)q : (int)q;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define WM_WAVE_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)    /* over the lanes active here */
#else
#define WM_WAVE_ANY(c) (c)                                         /* host emulation: lane by lane; only the moment of a flush depends on it */
#endif
#ifndef WM_RLA_TRIP_HOOK
#define WM_RLA_TRIP_HOOK(stream, sample)                           /* host measurements (tests/emu) count a lane's edge trips per step */
#endif
#define WM_RLA_CROW 17           /* words per lane in the run-length kernel's chip staging (16 + 1: conflict-free) */
/* Deglitch filter for a whole 32-sample block, bit-parallel.  W holds raw slicer bits in time
 * order: bit 5+k = sample k of the block, bits 0..4 = the five samples before it.
 *   T1/C1 (rtl_wmbus.c:126-144,733): level = popcount(last 6 raw bits) >= 3, by a bit-sliced adder;
 *   S1    (rtl_wmbus.c:149-154,644): LUT 0101011101111111 = newest | majority(previous three).
 * Returns bit k = deglitched level at sample k. */
__device__ __forceinline__ uint32_t deglitch_block(uint64_t W, bool s1)
{
    const uint64_t a0 = W, a1 = W << 1, a2 = W << 2, a3 = W << 3;
    uint64_t D;
    if (s1) D = a0 | (a1 & a2) | (a1 & a3) | (a2 & a3);
    else {
        const uint64_t a4 = W << 4, a5 = W << 5;
        const uint64_t x1 = a0 ^ a1, s_1 = x1 ^ a2, c_1 = (a0 & a1) | (a2 & x1);
        const uint64_t x2 = a3 ^ a4, s_2 = x2 ^ a5, c_2 = (a3 & a4) | (a5 & x2);
        D = (c_1 & c_2) | ((c_1 ^ c_2) & (s_1 | s_2));          /* s1+s2+2(c1+c2) >= 3 */
    }
    return (uint32_t)(D >> 5);
}

/* The same filter on a short word (history in bits 0..4, at most 27 samples): 32-bit operations. */
__device__ __forceinline__ uint32_t deglitch_word(uint32_t W, bool s1)
{
    const uint32_t a0 = W, a1 = W << 1, a2 = W << 2, a3 = W << 3;
    uint32_t D;
    if (s1) D = a0 | (a1 & a2) | (a1 & a3) | (a2 & a3);
    else {
        const uint32_t a4 = W << 4, a5 = W << 5;
        const uint32_t x1 = a0 ^ a1, s_1 = x1 ^ a2, c_1 = (a0 & a1) | (a2 & x1);
        const uint32_t x2 = a3 ^ a4, s_2 = x2 ^ a5, c_2 = (a3 & a4) | (a5 & x2);
        D = (c_1 & c_2) | ((c_1 ^ c_2) & (s_1 | s_2));
    }
    return D >> 5;
}

/* Run-length framer lane (rtl_wmbus.c:640-702 S1, :729-803 T1/C1), edge-driven: the per-sample
 * work (shift, deglitch, compare, count) is done for 64 samples at once with bit operations and
 * the lane only iterates over the EDGES of the deglitched signal.  A framer reset clears the raw
 * history (rtl_wmbus.c:632,723), so after one the remaining levels of the block are recomputed
 * from the masked history.  WmRlaState.raw keeps the last five raw bits in time order. */
struct RlaLds {                  /* lane-private; the block's waves are independent */
    uint64_t dw[3 * 64 * WM_RLA_WPB];                /* deglitched words 1-3 of the 256-sample step, [word][lane] */
    uint32_t chip[64 * WM_RLA_WPB * WM_RLA_CROW];    /* chip staging */
    uint8_t lut[2][1024];                            /* [S1][new five raw bits << 5 | the five before] -> the five levels (deglitch_word) */
};
/* by all threads of the block before its first lane runs, a barrier behind it */
__device__ __forceinline__ void rla_lds_init(RlaLds &lds, const uint32_t tid, const uint32_t nthreads)
{
    for (uint32_t i = tid; i < 2048u; i += nthreads) lds.lut[i >> 10][i & 1023u] = (uint8_t)(deglitch_word(i & 1023u, i >= 1024u) & 0x1Fu);
}

/* PASS: 0 = first pass only (a.list == nullptr), 1 = re-run list only, 2 = either (the fused launch): like the clock kernel,
 * each kind of launch has its own kernel (the main pass then carries no list walk: 78 instead of 97 VGPRs). */
/* One segment of one (chain, capture).  A re-run starts from the predecessor's recorded end state (the carried state for
 * segment 0) and leaves its own in st_final. */
template <int PASS, bool s1>
__device__ __forceinline__ void rla_segment_of(const K2Args &a, RlaLds &lds, const bool rerun, const uint32_t ch, const uint32_t stream, const uint32_t seg)
{
    uint32_t *s_chip = lds.chip;
    const WmPush &g = a.g;
    const uint64_t row = (uint64_t)ch * g.S + stream;
    const uint64_t sidx = row * g.nseg_cap[0] + seg;
    const uint32_t mb = seg * g.seg_len[0], me = min(g.M, mb + g.seg_len[0]);
    const uint32_t cap_rl = g.cap[0];
    WmRlaState *stS = (WmRlaState *)a.st_start, *stF = (WmRlaState *)a.st_final, *stC = (WmRlaState *)a.st_carry;
    const WmRlaState reset = {0, 8 * 256, 0, 2u, 0u, 0u, 24, 24};   /* :628-637 / :717-726, reset pending */

    WmRlaState s;
    uint32_t m;
    if (rerun) { s = seg ? stF[sidx - 1] : stC[row]; m = mb; }
    else if (mb <= g.lookback) { s = stC[row]; m = 0; }
    else { s = reset; m = (mb - g.lookback) & ~255u; }    /* whole 256-sample steps (segments are multiples of 1024) */

    const uint32_t *bw = a.bits + row * (g.Mcap / 32);
    uint32_t *out = a.chips + sidx * cap_rl;
    const uint32_t syncw = s1 ? WM_SYNC_S1 : WM_SYNC_T1C1, syncm = s1 ? WM_SYNC_S1_MASK : WM_SYNC_T1C1_MASK;
    const uint32_t hist_mask = s1 ? 0x1Cu : 0x1Fu;       /* S1 looks back 3 samples, T1/C1 5      */

    /* Chips are staged in LDS (16 words per lane) and leave in whole, 32-byte aligned groups of 8:
     * every lane appends to its own region of HBM, so with half a million lanes in flight the
     * partially written lines do not stay in L2; 4-byte stores (or unaligned 16-byte ones) turn
     * into read-modify-write traffic at the memory side and cost 3 of the kernel's 8.4 ms. */
    uint32_t *my_chip = s_chip + threadIdx.x * WM_RLA_CROW;
    uint32_t pend = 0, n_fl = 0, saw_sync = 0;               /* staged chips; chips already in HBM (multiple of 8) */
    /* beyond the primary region: the segment's spill chain (WmSpill).  cur_lvl/cur_base cache the chunk in use. */
    uint32_t n_chain = 0xFFFFFFFFu, cur_lvl = 0xFFFFFFFFu, cur_base = 0, n_stored = 0xFFFFFFFFu;   /* n_stored: chips kept, once some were dropped */
    auto spill_slot = [&](uint32_t j) -> uint32_t * {        /* 8 words at offset j (multiple of 8) behind the primary region */
        const uint32_t lvl = j / WM_SPILL_CHUNK;
        if (g.sp.arena == nullptr || lvl >= WM_SPILL_LEVELS) return nullptr;
        if (lvl != cur_lvl) {
            if (n_chain == 0xFFFFFFFFu) n_chain = g.sp.nchain[sidx];       /* chunks an earlier pass over this segment took */
            uint32_t base;
            if (lvl < n_chain) base = g.sp.chain[sidx * WM_SPILL_LEVELS + lvl];
            else {
                base = atomicAdd(g.sp.used, WM_SPILL_CHUNK);
                if (base > g.sp.arena_words || g.sp.arena_words - base < WM_SPILL_CHUNK) return nullptr;    /* arena exhausted */
                g.sp.chain[sidx * WM_SPILL_LEVELS + lvl] = base;
                n_chain = lvl + 1u; g.sp.nchain[sidx] = n_chain;
            }
            cur_lvl = lvl; cur_base = base;
        }
        return g.sp.arena + cur_base + j % WM_SPILL_CHUNK;
    };
    auto flush8 = [&]() {                                    /* the oldest 8 staged words -> HBM */
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = my_chip[i];
        /* once storage is exhausted the lane stops asking: every failed request bumps the 32-bit arena counter, and 2^21
         * of them in one push would wrap it back into chunks other segments own */
        uint32_t *dst = n_fl + 8u <= cap_rl ? out + n_fl : n_stored == 0xFFFFFFFFu ? spill_slot(n_fl - cap_rl) : nullptr;
        if (dst != nullptr && n_stored == 0xFFFFFFFFu) {
            *(uint4 *)(dst) = make_uint4(w[0], w[1], w[2], w[3]);
            *(uint4 *)(dst + 4) = make_uint4(w[4], w[5], w[6], w[7]);
        } else if (n_stored == 0xFFFFFFFFu) n_stored = n_fl;   /* storage exhausted: this and every later chip of the segment is dropped */
        for (uint32_t i = 8; i < pend; i++) my_chip[i - 8] = my_chip[i];
        n_fl += 8u; pend = pend > 8u ? pend - 8u : 0u;
    };

    /* slicer words arrive 8 at a time (one aligned 32-byte sector per lane and 256 samples; single
     * words cost a sector of HBM traffic each); the next group is in flight while this one is used */
    uint32_t grp = m >> 8;                                   /* group (256 samples) the lane is in */
    uint4 wq0 = *(const uint4 *)(bw + 8u * grp), wq1 = *(const uint4 *)(bw + 8u * grp + 4), nq0 = {}, nq1 = {};
    auto fetch_group = [&](uint32_t gq) {                    /* rows hold whole groups (Mcap is a multiple of 256) */
        if (gq * 256u < g.Mcap) { nq0 = *(const uint4 *)(bw + 8u * gq); nq1 = *(const uint4 *)(bw + 8u * gq + 4); }
    };
    fetch_group(grp + 1u);
    /* One step = 256 samples = one group of slicer words.  The wave walks its 64 lanes' edges in lock step, so a step costs
     * the wave the LARGEST edge count among its lanes, and that maximum is relatively smaller the longer the step: on the
     * bench workload (tools/rla_trips.py: the device source on the host, 64 captures as the 64 lanes) a T1/C1 lane has 7.5
     * edges per 64 samples on average, and the unluckiest of 64 lanes 18.1 per 64-sample step (round 3), 14.8 per 64 samples
     * of a 128-sample step, 12.6 of a 256-sample step (S1 chain: 2.8 on average; 6.0, 5.4, 5.0).  The step's four 64-bit words are deglitched in lock step up
     * front (words 1-3 wait in LDS); a lane that runs out of edges in its word takes the next one inside the same trip.  The
     * first five levels of a word look at raw history of the word before, which a framer reset in that word has cleared
     * (rtl_wmbus.c:632,723): they are redone at the hand-over from the history as it then is. */
    const uint8_t *lut = lds.lut[s1 ? 1 : 0];               /* levels of five samples from those and the five before (rla_lds_init) */
    uint64_t *my_dw = lds.dw + threadIdx.x;                  /* word i of the step at my_dw[(i - 1) * lanes per block] */
    auto block = [&](const bool emit) {
        const uint32_t kstep = min(256u, me - m);            /* ragged only at the end of the push */
        const uint32_t lo4[4] = {wq0.x, wq0.z, wq1.x, wq1.z}, hi4[4] = {wq0.y, wq0.w, wq1.y, wq1.w};
        uint64_t Rw[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t n = kstep > 64u * i ? min(64u, kstep - 64u * i) : 0u;
            Rw[i] = (((uint64_t)hi4[i] << 32) | lo4[i]) & (n == 64u ? ~0ull : ((1ull << n) - 1ull));   /* raw slicer bits, bit j = sample j of the word */
        }
        auto deglitch64 = [&](const uint64_t r, const uint32_t h) {
            return (uint64_t)deglitch_block(((uint64_t)(uint32_t)r << 5) | h, s1) | ((uint64_t)deglitch_block(((r >> 32) << 5) | ((uint32_t)r >> 27), s1) << 32);
        };
        uint32_t hist = s.raw & hist_mask;                   /* the five samples before the word, bit 4 = newest */
        uint64_t R = Rw[0], D = deglitch64(R, hist);
#pragma unroll
        for (int i = 1; i < 4; i++) my_dw[(i - 1) * (64 * WM_RLA_WPB)] = deglitch64(Rw[i], 0u);
        uint32_t w = 0, k0 = 0, kend = min(64u, kstep);
        uint32_t pos = m - mb;                               /* sample 0 of the word, relative to the segment (look-back: unused) */
        uint64_t rem = kend == 64u ? ~0ull : ((1ull << kend) - 1ull);     /* samples of the word still to look at */
        for (;;) {
            const uint32_t level = s.state & 1u;
            uint64_t x = (level ? ~D : D) & rem;
            bool more = true;
            /* The hand-over is in nearly every trip of the wave (64 lanes, three each per step, spread over fifty trips), so
             * it is kept to a few instructions: the deglitched word from LDS, the raw one picked from registers, the first
             * five levels from the table. */
            while (!x) {                                     /* no edge left in this word: on into the next one */
                s.run += (int)(kend - k0);
                if (64u * (w + 1u) >= kstep) { more = false; break; }
                hist = (uint32_t)(R >> 59) & hist_mask;      /* a whole word lies behind (only the last one can be ragged) */
                w++;
                R = w == 1u ? Rw[1] : w == 2u ? Rw[2] : Rw[3];
                D = my_dw[(w - 1u) * (64 * WM_RLA_WPB)];
                D = (D & ~0x1Full) | lut[(((uint32_t)R & 0x1Fu) << 5) | hist];
                kend = min(64u, kstep - 64u * w); k0 = 0; pos += 64u;
                rem = kend == 64u ? ~0ull : ((1ull << kend) - 1ull);
                x = (level ? ~D : D) & rem;
            }
            if (!more) break;
            const uint32_t k = (uint32_t)__ffsll((long long)x) - 1u;  /* first sample whose level differs */
            s.run += (int)(k - k0);
            WM_RLA_TRIP_HOOK(stream, m + 64u * w);
            /* Everything below is written for a wave whose 64 lanes take DIFFERENT paths at almost every
             * edge (a path one lane in fifty takes is taken by the wave nearly every time): no
             * data-dependent loop for the common cases, and the rare ones kept short. */
            const int run0 = s.run;
            int unit, half, runq;                        /* runq: the run in the units the chip clock counts in */
            bool rst;
            if (!s1) { unit = s.bitlen; half = unit / 2; runq = run0 * 256; rst = run0 < 5 || runq <= half; }   /* :742, :752-756 */
            else { unit = (s.spb0 + s.spb1) / 2; half = unit / 2; runq = run0; rst = unit <= 12 || unit >= 36 || run0 <= half; }   /* :659, :671 */
            if (rst) {
                s = reset;
                R &= ~((2ull << k) - 1ull); hist = 0;    /* raw history cleared, incl. sample k (k = 63: everything) */
                /* only the next five levels still look at cleared history: patch those instead of
                 * deglitching the whole step again (shifts by k + 1 <= 64 in two parts) */
                const uint32_t v5 = (uint32_t)((R >> k) >> 1) & 0x1Fu;                  /* raw samples k+1 .. k+5 */
                const uint64_t pm = (0x1Full << k) << 1;
                D = (D & ~pm) | (((uint64_t)lut[v5 << 5] << k) << 1);
            } else {
                /* chips of this run: the reference counts the run down chip by chip (:765-779 / :680-694),
                 * i.e. n = ceil((run - half) / unit) >= 1 */
                const int n = wm_sdiv(runq - half + unit - 1, unit);
                const uint32_t n_emit = (uint32_t)min(n, (int)WM_RLA_RUN_LIMIT);
                /* A run of more than WM_RLA_RUN_LIMIT chips (exact silence, then an edge): a packet
                 * decoder consumes at most 16*290 chips after an access code, and identical chips
                 * cannot complete one, so the rest of the run is not materialised -- only counted, as
                 * the reference's loop would. */
                /* Only the FIRST chip of a run can complete an access code (both end in two different chips, :97-103; the
                 * chips of a run are equal), and only it carries the reset marker: the others are one constant word, and
                 * the shift register takes them in one shift (what the wave pays per edge is the LONGEST run among its lanes). */
                s.sr = ((s.sr << 1) | level) & syncm;
                if (emit) {
                    const uint32_t word = WM_CHIP_WORD(pos + k, level);
                    const uint32_t hit = s.sr == syncw ? 2u : 0u;
                    saw_sync |= hit;
                    my_chip[pend] = word | hit | ((s.state & 2u) << 1);                      /* reset marker travels with the first chip */
                    if (++pend == 16u) flush8();             /* a long run can emit many chips at one edge */
                    for (uint32_t i = 1; i < n_emit; i++) {
                        my_chip[pend] = word;
                        if (++pend == 16u) flush8();
                    }
                    /* chips leave at the end of a step, all lanes together; a 256-sample step fills the 16 staged words several
                     * times over, and lanes flushing one by one would put the flush into nearly every trip of the wave:
                     * when one lane is about to run full, every lane that has a whole group sends it */
                    if (WM_WAVE_ANY(pend >= 12u) && pend >= 8u) flush8();
                }
                const uint32_t sh = n_emit - 1u < 24u ? n_emit - 1u : 24u;                   /* n >= 1 */
                s.sr = ((s.sr << sh) | (level ? (1u << sh) - 1u : 0u)) & syncm;
                s.state &= ~2u;
                const int rest = runq - n * unit;            /* what the count-down leaves: (half - unit, half] */
                if (!s1) {
                    s.cum += rest;
                    s.bitlen += wm_sdiv(rest + s.cum / 16, 32 * n);                          /* :792-796 */
                } else {
                    const int v = (int)wm_udiv((unsigned)run0, (unsigned)n);                  /* :698; run0 > half >= 0, n >= 1 */
                    if (level) s.spb1 = v; else s.spb0 = v;
                }
            }
            s.state = (s.state & 2u) | (level ^ 1u);
            s.run = 1;
            k0 = k + 1u;
            rem &= ~1ull << k;                               /* k = 63: nothing left */
        }
        /* the five newest raw bits, time order (a ragged last word may be shorter than five samples) */
        s.raw = (kend >= 5u ? (uint32_t)(R >> (kend - 5u)) : (((uint32_t)R << (5u - kend)) | (hist >> kend))) & 0x1Fu & hist_mask;
        wq0 = nq0; wq1 = nq1; grp++; fetch_group(grp + 1u);
        if (emit && pend >= 8u) flush8();
        m += 256;
    };
    while (m < mb) block(false);                         /* speculative look-back: no stores at all */
    stS[sidx] = s;
    while (m < me) block(true);
    const uint32_t n_out = n_fl + pend;
    while (pend) flush8();                               /* last group: the slots beyond n_out are never read */
    stF[sidx] = s;
    a.counts[sidx] = n_stored < n_out ? n_stored : n_out;        /* chips that can be read back */
    if (saw_sync) a.sync_seen[sidx] = 1u;
    if (n_stored < n_out) atomicOr(a.err, WM_ERR_CHIP_TRUNC);     /* a warning: the framer state is exact, some chips of this segment are lost */
}

/* The chain is a compile-time constant inside the segment (a wave of the first pass holds one chain; as a per-lane value every
 * `if (s1)` of the edge loop was a masked if/else -- seven scalar instructions each, three per edge -- and the access code, its
 * mask and the history mask were registers). */
template <int PASS>
__device__ __forceinline__ void rla_segment(const K2Args &a, RlaLds &lds, const bool rerun, const uint32_t ch, const uint32_t stream, const uint32_t seg)
{
    if (ch) rla_segment_of<PASS, true>(a, lds, rerun, ch, stream, seg);
    else rla_segment_of<PASS, false>(a, lds, rerun, ch, stream, seg);
}

/* The lanes of one launch.  First pass: lane = (chain, segment, capture), every lane one segment.
 * Re-run list (round 4): a listed lane walks its CHAIN.  A segment is listed because its start did not match its predecessor's
 * end (k2_verify also leaves that verdict per segment in `a.bad`); a re-run starts from the predecessor's end state -- which is
 * stale when the predecessor is re-run in the same launch.  A burst longer than a segment (an S1 telegram is 30-100 ms, a
 * segment 10 ms; configs[2] puts those into the T1/C1 chain's band as well) makes a run of consecutive listed segments, and
 * round 3 needed one round per segment of the run: verify, list, launch each time, and beyond the rounds enqueued the
 * host-driven path -- on EVERY push of configs[2] (64 ms per step).  Now the FIRST listed segment of a run does them all, one
 * after the other, each from the exact end state of the one before (the others return at once), and goes on into the segment
 * behind the run as long as the end state it arrives with differs from that segment's recorded start -- unless that segment
 * has a lane of its own in this launch (listed behind an unlisted one), which the next round sorts out. */
#ifdef WM_DBG_WALK_STATS
__device__ unsigned wm_walk_hist[2][32];
#endif
template <int PASS = 2>
__device__ __forceinline__ void rla_lanes(const K2Args &a, const uint32_t block_id, RlaLds &lds)
{
    uint32_t lane = block_id * (64 * WM_RLA_WPB) + threadIdx.x;
    if (lane >= k2_lane_count(a)) return;
    const bool rerun = PASS == 2 ? a.list != nullptr : PASS == 1;
    if (rerun) lane = a.list[lane];
    const WmPush &g = a.g;
    uint32_t ch, stream, seg;
    lane_decode(g, 0, lane, ch, stream, seg);
    if (!(g.flags & (ch ? WM_F_S1 : WM_F_T1C1))) return;
    if (!rerun || a.bad == nullptr) { rla_segment<PASS>(a, lds, rerun, ch, stream, seg); return; }
    /* nothing but (chain, capture, segment) lives across a segment: the end state a walk goes on from is the record the
     * segment has just written (kept in registers across the segment's loops it cost 13 VGPRs and spills: 2 ms per stage) */
    const uint64_t row = (uint64_t)ch * g.S + stream;
    const uint32_t *bad = a.bad + (uint64_t)ch * g.nseg_cap[0] * g.S + stream;       /* verdict of segment j at bad[j * S] */
    if (seg > 0u && bad[(uint64_t)(seg - 1u) * g.S]) return;            /* the head of my run covers me */
    /* (A walker that runs past the end of its run into an unlisted segment X rewrites X's end-state record while the lane of a
     * listed segment X + 1 -- listed behind an unlisted one, so it has a lane of its own in this launch -- may be reading that
     * record as its start state: a torn read can drive that lane's whole segment from a mixed state.  Benign: the next k2_verify
     * sees the mismatch and lists it again, so the result stays exact; it costs a round where it happens.  ADVICE r4.) */
#ifdef WM_DBG_WALK_STATS                                  /* tools/build_variant.sh walk -DWM_DBG_WALK_STATS: how long are the walks?  histogram per chain, read in wmbus_collect */
    uint32_t walked = 0;
#define WM_WALK_DONE() atomicAdd(&wm_walk_hist[ch][walked < 31u ? walked : 31u], 1u)
#else
#define WM_WALK_DONE() do {} while (0)
#endif
    for (;;) {
        rla_segment<PASS>(a, lds, true, ch, stream, seg);
#ifdef WM_DBG_WALK_STATS
        walked++;
#endif
        if (seg + 1u >= g.nseg[0]) { WM_WALK_DONE(); return; }
        const uint64_t sidx = row * g.nseg_cap[0] + seg;
        const uint32_t *x = (const uint32_t *)((const WmRlaState *)a.st_final + sidx), *y = (const uint32_t *)((const WmRlaState *)a.st_start + sidx + 1u);
        bool same = true;
#pragma unroll
        for (int k = 0; k < (int)(sizeof(WmRlaState) / 4); k++) same &= x[k] == y[k];
        if (same) { WM_WALK_DONE(); return; }                /* the next segment started from exactly this state */
        if (bad[(uint64_t)(seg + 1u) * g.S] && !bad[(uint64_t)seg * g.S]) { WM_WALK_DONE(); return; }      /* it is listed and has a lane of its own in this launch: next round */
        seg++;
    }
}

#ifndef WM_RLA_WAVES_PER_SIMD
#define WM_RLA_WAVES_PER_SIMD 1        /* 8: at most 64 VGPRs (build-time experiment, DESIGN_HISTORY.md section 10) */
#endif
__global__ __launch_bounds__(64 * WM_RLA_WPB, WM_RLA_WAVES_PER_SIMD) void k2_rla(K2Args a)           /* main pass: one block per 64 * WM_RLA_WPB lanes */
{
    wm_framer_prio();
    __shared__ RlaLds lds;
    rla_lds_init(lds, threadIdx.x, 64 * WM_RLA_WPB);
    __syncthreads();
    rla_lanes<0>(a, blockIdx.x, lds);
}

__global__ __launch_bounds__(64 * WM_RLA_WPB, WM_RLA_WAVES_PER_SIMD) void k2_rla_list(K2Args a)      /* re-run list: a fixed grid walks it */
{
    wm_framer_prio();
    __shared__ RlaLds lds;
    rla_lds_init(lds, threadIdx.x, 64 * WM_RLA_WPB);
    __syncthreads();
    const uint32_t n = k2_lane_count(a);
    for (uint32_t b = blockIdx.x; (uint64_t)b * (64u * WM_RLA_WPB) < n; b += gridDim.x) rla_lanes<1>(a, b, lds);
}

#endif /* WM_K2_RLA_H */
