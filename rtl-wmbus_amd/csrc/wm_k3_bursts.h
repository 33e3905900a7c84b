/* wm_k3_bursts.h -- K3: access-code hits -> bursts for the host packet decoders; K4: debug view of a chip stream.
 * Device code, included by wm_kernels.hip (one translation unit, see the overview there). */
#ifndef WM_K3_BURSTS_H
#define WM_K3_BURSTS_H

/* =============================================================================================
 * K3: access-code hits -> bursts for the host packet decoders
 * ===========================================================================================*/
struct K3Args {
    WmPush g;
    const uint8_t *rssi;         /* [2][S][Mcap]: the framers no longer copy the RSSI byte into every chip */
    const uint32_t *chips[2];    /* per algo: [2][S][nseg_cap][cap]                          */
    const uint32_t *counts[2];   /* per algo: [2][S][nseg_cap]                               */
    const uint2 *hits; const uint32_t *n_hits; uint32_t hits_cap;
    const uint32_t *pending;     /* [2 algo][2 chain][S]: chips still owed to a busy decoder  */
    WmBurstHdr *hdr; uint32_t hdr_cap;     /* bursts that go to the host decoders as chips: cut by the end of the push, or continuing */
    uint32_t *words; uint32_t words_cap;
    uint32_t *n_hdr, *n_words;
    WmPkt *pkts; uint32_t pkts_cap;        /* bursts decoded here (nullptr: every burst goes to the host as chips) */
    uint8_t *bytes; uint32_t bytes_cap;
    uint32_t *n_pkts, *n_bytes;
    uint32_t *err;
    struct WmItemRec *plans;     /* [4 S + hits_cap] what k3_spans has found out about every item (RSSI on demand: it runs before k3_bursts
                                    and needs the same facts); nullptr: k3_bursts finds out itself */
};

__device__ static const uint8_t D3OF6[64] = {
    255,255,255,255,255,255,255,255,255,255,255,3,255,1,2,255,255,255,255,7,255,255,0,255,255,5,6,255,4,255,255,255,
    255,255,255,11,255,9,10,255,255,15,255,255,8,255,255,255,255,13,14,255,12,255,255,255,255,255,255,255,255,255,255,255};

__device__ __forceinline__ uint32_t full_len_a(uint32_t L) { return 1u + L + 2u * (1u + (L > 9u ? (L - 9u + 15u) / 16u : 0u)); }

/* What a decoder does after an access code, read off the next 24 chips (hb: first chip in bit 23; nb: how many of
 * them exist), ignoring RSSI aborts, framer resets and Manchester errors in the data (those only shorten it):
 * `need` chips after the access-code chip, then either the telegram is complete (`done`) or the decoder has given up.
 * Mirrors the length logic of t1_c1_packet_decoder.h:298-349,399-438 and s1_packet_decoder.h:152-197. */
struct WmPlan {
    uint32_t need;
    uint8_t done;            /* 1: `need` chips complete a telegram; 0: they end in an abort (or the length is not known yet) */
    uint8_t mode;            /* 0 T1 (3-out-of-6), 1 C1 (NRZ), 2 S1 (Manchester) */
    uint8_t frame_b;
    uint16_t L;              /* expected length with CRC bytes (the decoder's L) */
    uint16_t nbytes;         /* bytes the decoder stores, L-field included */
};

/* An item as k3_spans leaves it for k3_bursts: where its chips start in the push's chip stream, how many are left, how many a
 * decoder takes (0: nothing to do -- an empty continuation slot, a stale hit) and the plan. */
struct WmItemRec { uint32_t chip0, avail, n; WmPlan plan; };

__device__ WmPlan burst_plan(uint32_t chain, uint32_t hb, uint32_t nb)
{
    WmPlan p = {};
    if (chain == 0) {
        p.need = WM_MAXCHIPS_T1C1;
        if (nb < 12) return p;
        const uint32_t hi = D3OF6[(hb >> 18) & 63u], lo = D3OF6[(hb >> 12) & 63u];
        if (hi != 255u && lo != 255u) {
            p.L = (uint16_t)full_len_a((hi << 4) | lo); p.nbytes = p.L; p.need = 12u * p.L; p.done = 1;
            return p;
        }
        const uint32_t mode = hb >> 12;
        if (mode != 0x54Cu && mode != 0x543u) { p.need = 12u; return p; }
        p.mode = 1; p.frame_b = mode == 0x543u;
        if (nb < 24) return p;
        if (((hb >> 8) & 15u) != 0xDu) { p.need = 16u; return p; }
        const uint32_t L = hb & 255u, total = p.frame_b ? 1u + L : full_len_a(L);
        /* the decoder looks at the length only after storing a byte: a frame B that claims L = 0 still
         * takes one byte after its L-field (found by tests/test_burst_need.py) */
        p.L = (uint16_t)total; p.nbytes = (uint16_t)(total > 2u ? total : 2u);
        p.need = 24u + 8u * (p.nbytes - 1u); p.done = 1;
        return p;
    }
    p.mode = 2; p.need = WM_MAXCHIPS_S1;
    if (nb < 16) return p;
    uint32_t L = 0;
    for (int j = 0; j < 8; j++) {
        const uint32_t pair = (hb >> (22 - 2 * j)) & 3u;
        if (pair == 0u || pair == 3u) { p.need = 2u * (uint32_t)j + 2u; return p; }
        L = (L << 1) | (pair == 1u ? 1u : 0u);
    }
    p.L = (uint16_t)full_len_a(L); p.nbytes = p.L; p.need = 16u * p.L; p.done = 1;
    return p;
}

__device__ uint32_t burst_need(uint32_t chain, uint32_t hb, uint32_t nb) { return burst_plan(chain, hb, nb).need; }

/* Access-code hits = chips with the sync flag, collected AFTER both framers have settled (re-runs
 * included).  (The framer kernels used to append hits as they went; every re-run then left stale
 * and duplicate records behind, each of which cost a burst copy.)  The framers only leave a
 * per-region flag "an access-code chip was emitted here by some pass"; each lane of a wave looks at
 * the flag of one (framer, chain, capture, segment) region, and the wave then scans the flagged
 * regions (a minority) together, appending {lane | algo << 31, chip index}. */
#define WM_K3_SCAN_PART 4096       /* a multiple of 1024 (a trip of the wave) */
__global__ __launch_bounds__(256) void k3_scan(WmPush g, const uint32_t *chips0, const uint32_t *chips1, const uint32_t *counts0,
                                               const uint32_t *counts1, const uint32_t *seen0, const uint32_t *seen1,
                                               uint2 *hits, uint32_t *n_hits, uint32_t hits_cap, uint32_t *err)
{
    wm_framer_prio();
    const uint32_t n0 = 2u * g.nseg[0] * g.S;                /* run-length lanes first */
    const uint32_t ln = threadIdx.x & 63u;
    uint32_t lane = blockIdx.x * 256u + threadIdx.x, algo = 0;
    if (lane >= n0) { lane -= n0; algo = 1; }
    uint32_t my_cnt = 0, my_sidx = 0;
    if (lane < 2u * g.nseg[algo] * g.S) {
        uint32_t ch, stream, seg;
        lane_decode(g, algo, lane, ch, stream, seg);
        const uint32_t sidx = (ch * g.S + stream) * g.nseg_cap[algo] + seg;
        if ((g.flags & (ch ? WM_F_S1 : WM_F_T1C1)) && (g.flags & (algo ? WM_F_T2A : WM_F_RLA)) && (algo ? seen1 : seen0)[sidx]) {
            my_sidx = sidx;
            my_cnt = (algo ? counts1 : counts0)[sidx];           /* chips that can be read back (time2: <= cap by construction) */
        }
    }
    for (uint64_t todo = __ballot(my_cnt != 0u); todo; todo &= todo - 1ull) {
        const int src = __ffsll((long long)todo) - 1;
        const uint32_t cnt = __shfl(my_cnt, src), sidx = __shfl(my_sidx, src), r_lane = __shfl(lane, src), r_algo = __shfl(algo, src);
        const uint32_t *prim = r_algo ? chips1 : chips0;
        /* four 16-byte loads in flight per lane (a region of the clock framer is 8 k chips: 32 dependent
         * trips of one load each were most of this kernel's time) */
        /* a region is scanned in parts of WM_K3_SCAN_PART chips, one wave (blockIdx.y) per part: the scan of a flagged region is a chain
         * of dependent trips as long as the region, and the clock framer's regions grew with its segments (65 536 samples: 16 392 chips,
         * round 6 -- the launch went from 1.3 to 2.1 ms on its own) */
        const uint32_t k_lo = blockIdx.y * (uint32_t)WM_K3_SCAN_PART, k_hi = min(cnt, k_lo + (uint32_t)WM_K3_SCAN_PART);
        for (uint32_t kb = k_lo + 4u * ln; kb < k_hi; kb += 1024u) {    /* regions and spill chunks are 32-byte aligned, cap % 8 == 0 */
            uint4 v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++)
                v[u] = kb + 256u * u < k_hi ? *(const uint4 *)wm_chip_ptr(g, prim, r_algo, sidx, kb + 256u * u) : uint4{0u, 0u, 0u, 0u};
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t k4 = kb + 256u * u;
                const uint32_t q[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (uint32_t j = 0; j < 4; j++)
                    if (k4 + j < k_hi && (q[j] & 2u)) {
                        const uint32_t i = atomicAdd(n_hits, 1u);
                        if (i < hits_cap) hits[i] = make_uint2(r_lane | (r_algo << 31), k4 + j);
                        else atomicOr(err, WM_ERR_BURST_OVERFLOW);
                    }
            }
        }
    }
}

/* Per-wave scratch of k3_bursts: the burst's chips as a bit string (chip j = bit j % 64 of word j / 64) and the
 * telegram's bytes. */
#define WM_K3_BITWORDS 76          /* >= (WM_MAXCHIPS_S1 + 1 + 63) / 64 + 1 */
struct K3Lds { unsigned long long bits[4][WM_K3_BITWORDS]; uint8_t bytes[4][WM_PKT_MAXBYTES + 4]; };

#ifndef WM_PEEK
#define WM_PEEK(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)     /* a counter other waves are adding to */
#endif
#ifndef WM_WAVE_SYNC
#define WM_WAVE_SYNC() __builtin_amdgcn_wave_barrier()       /* LDS hand-over inside one wave (DS operations of a wave are in order) */
#endif

/* `n` chips (n <= 32) of the burst starting at chip j0, first chip in the most significant bit. */
__device__ __forceinline__ uint32_t k3_field(const unsigned long long *bits, uint32_t j0, uint32_t n)
{
    const uint32_t w = j0 >> 6, o = j0 & 63u;
    unsigned long long v = bits[w] >> o;
    if (o) v |= bits[w + 1] << (64u - o);
    return __brev((uint32_t)v) >> (32u - n);
}

/* EN 13757 CRC-16 of `n` bytes (poly 0x3D65, init 0, complemented), t1_c1_packet_decoder.h:463-469. */
__device__ __forceinline__ uint32_t k3_crc16(const uint8_t *p, uint32_t n)
{
    uint32_t crc = 0;
    for (uint32_t i = 0; i < n; i++) {
        crc ^= (uint32_t)p[i] << 8;
#pragma unroll
        for (int k = 0; k < 8; k++) crc = (crc & 0x8000u) ? ((crc << 1) ^ 0x3D65u) : (crc << 1);
        crc &= 0xFFFFu;
    }
    return (~crc) & 0xFFFFu;
}

/* One access-code hit (or pending continuation), handled by one wave.
 *
 * A burst that ends inside the push is DECODED here, chip for chip what the reference's state machines do
 * (t1_c1_packet_decoder.h:649-712, s1_packet_decoder.h:233-282; the host's wm_decoder.c is the same logic and the
 * tests hold the two against each other): the chips the decoder would take are loaded once (value, framer-reset
 * marker, RSSI byte), the first chip at which it would stop early is found by a wave-wide minimum --
 *     the access-code chip itself with RSSI < 5: not armed, 1 chip consumed              (:705-710 gate at idle)
 *     a framer-reset marker on chip j >= 1: aborted BEFORE chip j, j consumed           (rtl_wmbus.c:636,725)
 *     RSSI < 5 on a chip that does not complete the telegram: aborted, j + 1 consumed   (:705-710)
 *     an invalid Manchester pair completed by chip j (S1): j + 1 consumed              (s1_packet_decoder.h:152-168)
 *     an invalid L-field / mode / trailer: the plan's own end                          (:313-349, 399-415)
 * -- and if none precedes the last chip, lanes assemble the bytes (3-out-of-6 with the error flag, NRZ, Manchester)
 * and check the block CRCs.  The host gets {consumed, flags, RSSI pair, completing sample} and the bytes.  A burst
 * cut by the end of the push, and its continuation in the next one, still travel as chips to the host decoder. */
/* What k3_bursts and k3_spans both need to know about an item: where its chips are and how many of them a decoder takes. */
struct K3Item {
    uint32_t algo, ch, stream, seg, k, cont, nseg, seg_len, chip0, avail, n;
    uint64_t row, sidx0;
    const uint32_t *cnt;         /* chips per segment of the (framer, chain, capture) */
    WmPlan plan;
};
__device__ __forceinline__ uint32_t k3_chip(const K3Args &a, const K3Item &it, uint32_t sg, uint32_t kk)
{
    return *wm_chip_ptr(a.g, a.chips[it.algo], it.algo, it.sidx0 + sg, kk);
}
__device__ __forceinline__ void k3_locate(const K3Item &it, uint32_t j, uint32_t &sg, uint32_t &kk)   /* chip0 + j -> (segment, index) */
{
    sg = it.seg; kk = it.k + j;
    while (sg < it.nseg) { const uint32_t c = it.cnt[sg]; if (kk < c) break; kk -= c; sg++; }
}
/* Who the item is: framer, chain, capture, the segment and index of its first chip.  false: an empty continuation slot or an
 * index beyond the hits.  `want` = chips a continuation still owes. */
__device__ __forceinline__ bool k3_item_id(const K3Args &a, const uint32_t item, K3Item &it, uint32_t &want)
{
    const WmPush &g = a.g;
    const uint32_t n_hits = min(*a.n_hits, a.hits_cap);
    it.cont = 0; want = 0;
    if (item < 4u * g.S) {                       /* continuation slots come first            */
        it.algo = item / (2u * g.S); it.ch = (item / g.S) & 1u; it.stream = item % g.S;
        want = a.pending[item];
        if (want == 0u) return false;
        it.seg = 0; it.k = 0; it.cont = 1;
    } else {
        if (item - 4u * g.S >= n_hits) return false;
        const uint2 h = a.hits[item - 4u * g.S];
        it.algo = h.x >> 31;
        lane_decode(g, it.algo, h.x & 0x7FFFFFFFu, it.ch, it.stream, it.seg);
        it.k = h.y;
    }
    it.nseg = g.nseg[it.algo]; it.seg_len = g.seg_len[it.algo];
    it.row = (uint64_t)it.ch * g.S + it.stream;
    it.cnt = a.counts[it.algo] + it.row * g.nseg_cap[it.algo];
    it.sidx0 = it.row * g.nseg_cap[it.algo];
    return true;
}
/* false: nothing to do for this item (an empty continuation slot, a stale hit of a re-run segment).  By one wave. */
__device__ __forceinline__ bool k3_item(const K3Args &a, const uint32_t item, const uint32_t ln, K3Item &it)
{
    uint32_t want;
    if (!k3_item_id(a, item, it, want)) return false;
    if (!it.cont) {                              /* stale record of a re-run segment?         */
        if (it.k >= it.cnt[it.seg] || !(k3_chip(a, it, it.seg, it.k) & 2u)) return false;
    }
    /* chips before / from the hit in this push's chip stream: the wave sums the segment counts in
     * parallel (a serial scan of up to 256 dependent loads per wave was most of this kernel's time) */
    uint32_t before = 0, total = 0;
    for (uint32_t s = ln; s < it.nseg; s += 64u) { const uint32_t c = it.cnt[s]; if (s < it.seg) before += c; total += c; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { before += __shfl_xor(before, off); total += __shfl_xor(total, off); }
    it.chip0 = before + it.k;
    if (it.chip0 >= total) return false;
    it.avail = total - it.chip0;                 /* chips from the hit to the end of the push */
    it.plan = WmPlan{};
    if (it.cont) it.n = min(want, it.avail);
    else {
        uint32_t bit = 0;
        if (ln < 24u && 1u + ln < it.avail) { uint32_t sg, kk; k3_locate(it, 1u + ln, sg, kk); bit = k3_chip(a, it, sg, kk) & 1u; }
        const unsigned long long m = __ballot(bit);
        uint32_t hb = 0;
        for (int j = 0; j < 24; j++) hb |= (uint32_t)((m >> j) & 1ull) << (23 - j);
        it.plan = burst_plan(it.ch, hb, min(24u, it.avail - 1u));
        it.n = min(it.plan.need + 1u, it.avail);
    }
    return true;
}
/* the same from the record k3_spans has left (K3Args.plans) */
__device__ __forceinline__ bool k3_item_recorded(const K3Args &a, const uint32_t item, K3Item &it)
{
    const WmItemRec r = a.plans[item];
    uint32_t want;
    if (r.n == 0u || !k3_item_id(a, item, it, want)) return false;
    it.chip0 = r.chip0; it.avail = r.avail; it.n = r.n; it.plan = r.plan;
    return true;
}

__device__ void burst_item(const K3Args &a, const uint32_t item, const uint32_t ln, unsigned long long *s_bits, uint8_t *s_bytes)
{
    const WmPush &g = a.g;
    K3Item it;
    if (!(a.plans ? k3_item_recorded(a, item, it) : k3_item(a, item, ln, it))) return;
    const uint32_t algo = it.algo, ch = it.ch, stream = it.stream, cont = it.cont, seg_len = it.seg_len, chip0 = it.chip0, avail = it.avail, n = it.n;
    const uint64_t row = it.row;
    const WmPlan plan = it.plan;
    auto chip = [&](uint32_t sg, uint32_t kk) { return k3_chip(a, it, sg, kk); };
    auto locate = [&](uint32_t j, uint32_t &sg, uint32_t &kk) { k3_locate(it, j, sg, kk); };

    if (!cont && a.pkts != nullptr && plan.need + 1u <= avail) {
        /* ---- the whole burst is here: decode it -------------------------------------------------- */
        uint32_t ev = n;                         /* chips consumed if nothing stops the decoder early */
        uint32_t r_first = 0, r_last = 0, pm_last = 0;
        for (uint32_t j0 = 0; j0 < n; j0 += 64u) {
            const uint32_t j = j0 + ln;
            uint32_t w = 0, pm = 0, rs = 255u;
            if (j < n) {
                uint32_t sg, kk; locate(j, sg, kk);
                w = chip(sg, kk);
                pm = sg * seg_len + WM_CHIP_POS(w);
                rs = a.rssi[row * g.Mcap + pm];
                if (j >= 1u && (w & 4u)) ev = min(ev, j);                                  /* framer reset before this chip */
                else if (rs < 5u && (j == 0u || j + 1u < n || !plan.done)) ev = min(ev, j + 1u);   /* RSSI gate */
            }
            const unsigned long long m = __ballot(w & 1u);
            if (ln == 0) s_bits[j0 >> 6] = m;
            if (j == 1u) r_first = rs;
            if (j + 1u == n) { r_last = rs; pm_last = pm; }
        }
        if (ln == 0) s_bits[(n + 63u) >> 6] = 0ull;          /* k3_field may look one word ahead */
        WM_WAVE_SYNC();
        const uint32_t nb = plan.done ? plan.nbytes : 0u;
        uint32_t err36 = 0, bad_pair = 0;
        /* bytes: lane b assembles byte b, b + 64, ... */
        for (uint32_t b0 = 0; b0 < nb; b0 += 64u) {
            const uint32_t b = b0 + ln;
            if (b < nb) {
                uint32_t v;
                if (plan.mode == 0u) {
                    const uint32_t hi = D3OF6[k3_field(s_bits, 1u + 12u * b, 6)], lo = D3OF6[k3_field(s_bits, 7u + 12u * b, 6)];
                    err36 |= (hi == 255u || lo == 255u);
                    v = ((hi == 255u ? 255u : hi << 4) | lo) & 255u;
                } else if (plan.mode == 1u) v = k3_field(s_bits, 17u + 8u * b, 8);
                else {
                    const uint32_t sym = k3_field(s_bits, 1u + 16u * b, 16);
                    v = 0;
#pragma unroll
                    for (int q = 7; q >= 0; q--) {           /* pair q: 01 -> 1, 10 -> 0 (s1_packet_decoder.h:35-37) */
                        const uint32_t pair = (sym >> (2 * q)) & 3u;
                        /* chip completing the pair, + 1.  For the telegram's LAST pair that is n itself, which must not read as
                         * "nothing stopped the decoder": s1_rx_last_data_bit resets on it like on any other pair
                         * (s1_packet_decoder.h:204-215), so the error travels in a flag of its own */
                        if (pair == 0u || pair == 3u) { ev = min(ev, 1u + 16u * b + 2u * (7u - (uint32_t)q) + 2u); bad_pair = 1u; }
                        v = (v << 1) | (pair == 1u);
                    }
                }
                s_bytes[b] = (uint8_t)v;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ev = min(ev, __shfl_xor(ev, off));
        err36 = __ballot(err36) != 0ull;
        bad_pair = __ballot(bad_pair) != 0ull;
        r_first = __shfl(r_first, 1); r_last = __shfl(r_last, (int)((n - 1u) & 63u)); pm_last = __shfl(pm_last, (int)((n - 1u) & 63u));
        const bool done = plan.done && ev == n && !bad_pair;
        WM_WAVE_SYNC();
        /* block CRCs (t1_c1_packet_decoder.h:471-536): frame A 12 bytes then 18s, frame B 128s, the last block may be short */
        uint32_t crc_bad = 0;
        if (done) {
            const uint32_t L = plan.L, first = plan.frame_b ? 128u : 12u, next = plan.frame_b ? 128u : 18u;
            const uint32_t start = ln == 0 ? 0u : first + (ln - 1u) * next;
            if (L < 12u) crc_bad = 1;
            else if (start < L) {
                const uint32_t blk = min(ln == 0 ? first : next, L - start);
                crc_bad = blk < 2u || k3_crc16(s_bytes + start, blk - 2u) != (((uint32_t)s_bytes[start + blk - 2u] << 8) | s_bytes[start + blk - 1u]);
            }
        }
        crc_bad = __ballot(crc_bad) != 0ull;
        /* storage first, then the slot: every slot below the capacity that the counter hands out is written in full (the
         * host walks min(counter, capacity) entries of pinned memory and must never meet one that nobody wrote) */
        uint32_t slot = 0xFFFFFFFFu, boff = 0;
        if (ln == 0) {
            /* a full arena is not asked again (the counters are 32 bits wide; 2^24 hits could wrap them) */
            const bool room = !done || WM_PEEK(a.n_bytes) <= a.bytes_cap;
            boff = done && room ? atomicAdd(a.n_bytes, (nb + 3u) & ~3u) : 0u;
            if (room && (!done || (boff <= a.bytes_cap && a.bytes_cap - boff >= nb))) slot = atomicAdd(a.n_pkts, 1u);
        }
        slot = __shfl(slot, 0); boff = __shfl(boff, 0);
        if (slot >= a.pkts_cap) { if (ln == 0) atomicOr(a.err, WM_ERR_BURST_OVERFLOW); WM_WAVE_SYNC(); return; }
        if (done) for (uint32_t b = ln; b < nb; b += 64u) a.bytes[boff + b] = s_bytes[b];
        if (ln == 0) {
            WmPkt p;
            p.stream = stream; p.chain = (uint8_t)ch; p.algo = (uint8_t)algo;
            p.status = done ? WM_PKT_DONE : WM_PKT_ABORT;
            p.flags = (uint8_t)((plan.mode == 1u ? WM_PKTF_C1 : 0u) | (plan.frame_b ? WM_PKTF_FRAME_B : 0u) | (err36 ? WM_PKTF_ERR3OF6 : 0u) |
                                (done && !crc_bad ? WM_PKTF_CRC_OK : 0u));
            p.chip0 = chip0; p.consumed = ev; p.sample = g.m0 + pm_last; p.off = boff; p.L = plan.L;
            p.pkt_rssi = (uint8_t)r_first; p.rssi_now = (uint8_t)r_last;
            a.pkts[slot] = p;
        }
        WM_WAVE_SYNC();                                      /* the scratch is reused by the wave's next item */
        return;
    }

    /* ---- cut by the end of the push, or the rest of such a burst: chips for the host decoder ---- */
    uint32_t hslot = 0xFFFFFFFFu, woff = 0;
    if (ln == 0) {                                           /* words first, then the slot (see above) */
        if (WM_PEEK(a.n_words) <= a.words_cap) {
            woff = atomicAdd(a.n_words, n);
            if (woff <= a.words_cap && a.words_cap - woff >= n) hslot = atomicAdd(a.n_hdr, 1u);
        }
    }
    hslot = __shfl(hslot, 0); woff = __shfl(woff, 0);
    if (hslot >= a.hdr_cap) { if (ln == 0) atomicOr(a.err, WM_ERR_BURST_OVERFLOW); return; }
    uint32_t sg0, k0; locate(0, sg0, k0);
    const uint64_t pos0 = g.m0 + (uint64_t)sg0 * seg_len + WM_CHIP_POS(chip(sg0, k0));
    /* one chip per lane and trip.  (Four independent chip -> RSSI load chains per lane were measured at -3 % for the whole
     * job, round 2 bisect: the kernel got shorter, its register and issue footprint beside the demodulation kernel larger.) */
    for (uint32_t j = ln; j < n; j += 64u) {
        uint32_t sg, kk; locate(j, sg, kk);
        const uint32_t w = chip(sg, kk);
        const uint32_t pm = sg * seg_len + WM_CHIP_POS(w);               /* push-relative decimated sample */
        const uint64_t pos = g.m0 + pm;
        const uint32_t rssi = a.rssi[row * g.Mcap + pm];                 /* (unsigned)EMA at the chip's sample */
        a.words[woff + j] = ((uint32_t)(pos - pos0) << 11) | (rssi << 3) | (WM_CHIP_VAL(w) & 7u);
    }
    if (ln == 0) {
        WmBurstHdr h;
        h.stream = stream; h.chain = (uint8_t)ch; h.algo = (uint8_t)algo; h.flags = (uint16_t)cont;
        h.chip0 = chip0; h.n_chips = n; h.pos0 = pos0; h.word_off = woff; h.avail = avail;
        a.hdr[hslot] = h;
    }
}

/* A bounded number of waves walks the items (continuation slots, then hits).  One wave per item
 * -- 14 000 single-wave blocks per 128 captures, each a chain of dependent loads -- took every wave
 * slot of the chip for the kernel's duration and stalled the demodulation kernel of the next
 * context (measured: K1 ran at a quarter of its speed while this kernel was resident). */
__global__ __launch_bounds__(256) void k3_bursts(K3Args a, uint32_t n_items_host)
{
    wm_framer_prio();
    __shared__ K3Lds lds;
    const uint32_t ln = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    /* the host may not know the number of hits yet (no round trip between k3_scan and this kernel): ~0 = read it here */
    const uint32_t n_items = n_items_host != 0xFFFFFFFFu ? n_items_host : 4u * a.g.S + min(*a.n_hits, a.hits_cap);
    for (uint32_t item = blockIdx.x * 4u + wv; item < n_items; item += gridDim.x * 4u) burst_item(a, item, ln, lds.bits[wv], lds.bytes[wv]);
}

/* RSSI on demand (wm_k1_demod.h): the demodulation tiles whose RSSI bytes k3_bursts is going to read -- for every item
 * the samples from its first chip to the last one a decoder takes -- are flagged per chain and listed once; the last tile
 * of every capture is always listed (it carries the filter's state into the next push). */
__device__ __forceinline__ void k3_mark(uint32_t *flags, uint32_t *list, uint32_t *n_list, uint32_t S, uint32_t ntiles, uint32_t tile, uint32_t stream, uint32_t bits)
{
    if (atomicOr(flags + (uint64_t)tile * S + stream, bits) == 0u) list[atomicAdd(n_list, 1u)] = stream * ntiles + tile;   /* at most one entry per (tile, capture) */
}
__global__ __launch_bounds__(256) void k3_spans(K3Args a, uint32_t tile_len, uint32_t ntiles, uint32_t *flags, uint32_t *list, uint32_t *n_list)
{
    wm_framer_prio();
    const WmPush &g = a.g;
    const uint32_t ln = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t s = blockIdx.x * 256u + threadIdx.x; s < g.S; s += gridDim.x * 256u) k3_mark(flags, list, n_list, g.S, ntiles, ntiles - 1u, s, 3u);
    const uint32_t n_items = 4u * g.S + min(*a.n_hits, a.hits_cap);
    for (uint32_t item = blockIdx.x * 4u + wv; item < n_items; item += gridDim.x * 4u) {
        K3Item it;
        const bool todo = k3_item(a, item, ln, it);
        if (a.plans && ln == 0) a.plans[item] = todo ? WmItemRec{it.chip0, it.avail, it.n, it.plan} : WmItemRec{};
        if (!todo) continue;
        uint32_t pm = 0;
        if (ln < 2u) { uint32_t sg, kk; k3_locate(it, ln ? it.n - 1u : 0u, sg, kk); pm = sg * it.seg_len + WM_CHIP_POS(k3_chip(a, it, sg, kk)); }
        const uint32_t t0 = __shfl(pm, 0) / tile_len, t1 = min(__shfl(pm, 1) / tile_len, ntiles - 1u);
        for (uint32_t tl = t0 + ln; tl <= t1; tl += 64u) k3_mark(flags, list, n_list, g.S, ntiles, tl, it.stream, 1u << it.ch);
    }
}

/* Debug/parity helper: flatten one (chain, algo, stream) chip stream. */
__global__ void k4_flatten(WmPush g, uint32_t algo, const uint32_t *chips, const uint32_t *counts, const uint8_t *rssi, uint32_t cap,
                           uint32_t ch, uint32_t stream, uint32_t *dst, uint64_t *pos, uint32_t max_out, uint32_t *n_out)
{
    if (blockIdx.x || threadIdx.x) return;
    const uint64_t row = (uint64_t)ch * g.S + stream;
    uint32_t n = 0;
    for (uint32_t s = 0; s < g.nseg[algo]; s++) {
        const uint32_t c = counts[row * g.nseg_cap[algo] + s];
        for (uint32_t k = 0; k < c; k++, n++)
            if (n < max_out) {
                const uint32_t w = *wm_chip_ptr(g, chips, algo, row * g.nseg_cap[algo] + s, k);
                dst[n] = WM_CHIP_VAL(w) | (rssi ? (uint32_t)rssi[row * g.Mcap + s * g.seg_len[algo] + WM_CHIP_POS(w)] << 8 : 0u);   /* rssi == nullptr: no RSSI rows to show */
                if (pos) pos[n] = g.m0 + (uint64_t)s * g.seg_len[algo] + WM_CHIP_POS(w);
            }
    }
    *n_out = n;
}

#endif /* WM_K3_BURSTS_H */
