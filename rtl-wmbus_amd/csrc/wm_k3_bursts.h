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
    WmBurstHdr *hdr; uint32_t hdr_cap;
    uint32_t *words; uint32_t words_cap;
    uint32_t *n_hdr, *n_words;
    uint32_t *err;
};

__device__ static const uint8_t D3OF6[64] = {
    255,255,255,255,255,255,255,255,255,255,255,3,255,1,2,255,255,255,255,7,255,255,0,255,255,5,6,255,4,255,255,255,
    255,255,255,11,255,9,10,255,255,15,255,255,8,255,255,255,255,13,14,255,12,255,255,255,255,255,255,255,255,255,255,255};

__device__ __forceinline__ uint32_t full_len_a(uint32_t L) { return 1u + L + 2u * (1u + (L > 9u ? (L - 9u + 15u) / 16u : 0u)); }

/* Chips after the access-code chip that a decoder consumes before it returns to idle, ignoring
 * RSSI aborts and framer resets (those only shorten it).  hb = the next 24 chips, first chip in
 * bit 23; nb = how many of them exist.  Mirrors the length logic of
 * t1_c1_packet_decoder.h:298-349,399-438 and s1_packet_decoder.h:152-197. */
__device__ uint32_t burst_need(uint32_t chain, uint32_t hb, uint32_t nb)
{
    if (chain == 0) {
        if (nb < 12) return WM_MAXCHIPS_T1C1;
        const uint32_t hi = D3OF6[(hb >> 18) & 63u], lo = D3OF6[(hb >> 12) & 63u];
        if (hi != 255u && lo != 255u) return 12u * full_len_a((hi << 4) | lo);
        const uint32_t mode = hb >> 12;
        if (mode != 0x54Cu && mode != 0x543u) return 12u;
        if (nb < 24) return WM_MAXCHIPS_T1C1;
        if (((hb >> 8) & 15u) != 0xDu) return 16u;
        const uint32_t L = hb & 255u, total = mode == 0x543u ? 1u + L : full_len_a(L);
        /* the decoder looks at the length only after storing a byte: a frame B that claims L = 0 still
         * takes one byte after its L-field (found by tests/test_burst_need.py) */
        return 24u + 8u * (total > 2u ? total - 1u : 1u);
    }
    if (nb < 16) return WM_MAXCHIPS_S1;
    uint32_t L = 0;
    for (int j = 0; j < 8; j++) {
        const uint32_t pair = (hb >> (22 - 2 * j)) & 3u;
        if (pair == 0u || pair == 3u) return 2u * (uint32_t)j + 2u;
        L = (L << 1) | (pair == 1u ? 1u : 0u);
    }
    return 16u * full_len_a(L);
}

/* Access-code hits = chips with the sync flag, collected AFTER both framers have settled (re-runs
 * included).  (The framer kernels used to append hits as they went; every re-run then left stale
 * and duplicate records behind, each of which cost a burst copy.)  The framers only leave a
 * per-region flag "an access-code chip was emitted here by some pass"; each lane of a wave looks at
 * the flag of one (framer, chain, capture, segment) region, and the wave then scans the flagged
 * regions (a minority) together, appending {lane | algo << 31, chip index}. */
__global__ __launch_bounds__(256) void k3_scan(WmPush g, const uint32_t *chips0, const uint32_t *chips1, const uint32_t *counts0,
                                               const uint32_t *counts1, const uint32_t *seen0, const uint32_t *seen1,
                                               uint2 *hits, uint32_t *n_hits, uint32_t hits_cap, uint32_t *err)
{
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
        for (uint32_t kb = 4u * ln; kb < cnt; kb += 1024u) {            /* regions and spill chunks are 32-byte aligned, cap % 8 == 0 */
            uint4 v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++)
                v[u] = kb + 256u * u < cnt ? *(const uint4 *)wm_chip_ptr(g, prim, r_algo, sidx, kb + 256u * u) : uint4{0u, 0u, 0u, 0u};
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t k4 = kb + 256u * u;
                const uint32_t q[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (uint32_t j = 0; j < 4; j++)
                    if (k4 + j < cnt && (q[j] & 2u)) {
                        const uint32_t i = atomicAdd(n_hits, 1u);
                        if (i < hits_cap) hits[i] = make_uint2(r_lane | (r_algo << 31), k4 + j);
                        else atomicOr(err, WM_ERR_BURST_OVERFLOW);
                    }
            }
        }
    }
}

/* One access-code hit (or pending continuation), handled by one wave. */
__device__ void burst_item(const K3Args &a, const uint32_t item, const uint32_t ln)
{
    const WmPush &g = a.g;
    const uint32_t n_hits = min(*a.n_hits, a.hits_cap);
    uint32_t algo, ch, stream, seg, k, cont = 0, want;
    if (item < 4u * g.S) {                       /* continuation slots come first            */
        algo = item / (2u * g.S); ch = (item / g.S) & 1u; stream = item % g.S;
        want = a.pending[item];
        if (want == 0u) return;
        seg = 0; k = 0; cont = 1;
    } else {
        if (item - 4u * g.S >= n_hits) return;
        const uint2 h = a.hits[item - 4u * g.S];
        algo = h.x >> 31;
        lane_decode(g, algo, h.x & 0x7FFFFFFFu, ch, stream, seg);
        k = h.y; want = 0;
    }
    const uint32_t nseg = g.nseg[algo], seg_len = g.seg_len[algo];
    const uint64_t row = (uint64_t)ch * g.S + stream;
    const uint32_t *cnt = a.counts[algo] + row * g.nseg_cap[algo];
    const uint64_t sidx0 = row * g.nseg_cap[algo];
    auto chip = [&](uint32_t sg, uint32_t kk) { return *wm_chip_ptr(g, a.chips[algo], algo, sidx0 + sg, kk); };
    if (!cont) {                                 /* stale record of a re-run segment?         */
        if (k >= cnt[seg] || !(chip(seg, k) & 2u)) return;
    }
    /* chips before / from the hit in this push's chip stream: the wave sums the segment counts in
     * parallel (a serial scan of up to 256 dependent loads per wave was most of this kernel's time) */
    uint32_t before = 0, total = 0;
    for (uint32_t s = ln; s < nseg; s += 64u) { const uint32_t c = cnt[s]; if (s < seg) before += c; total += c; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { before += __shfl_xor(before, off); total += __shfl_xor(total, off); }
    const uint32_t chip0 = before + k;
    if (chip0 >= total) return;
    const uint32_t avail = total - chip0;        /* chips from the hit to the end of the push */

    auto locate = [&](uint32_t j, uint32_t &sg, uint32_t &kk) {   /* chip0 + j -> (segment, index) */
        sg = seg; kk = k + j;
        while (sg < nseg) { const uint32_t c = cnt[sg]; if (kk < c) break; kk -= c; sg++; }
    };

    uint32_t n;
    if (cont) n = min(want, avail);
    else {
        uint32_t bit = 0;
        if (ln < 24u && 1u + ln < avail) { uint32_t sg, kk; locate(1u + ln, sg, kk); bit = chip(sg, kk) & 1u; }
        const unsigned long long m = __ballot(bit);
        uint32_t hb = 0;
        for (int j = 0; j < 24; j++) hb |= (uint32_t)((m >> j) & 1ull) << (23 - j);
        n = min(burst_need(ch, hb, min(24u, avail - 1u)) + 1u, avail);
    }
    uint32_t hslot = 0, woff = 0;
    if (ln == 0) { hslot = atomicAdd(a.n_hdr, 1u); woff = atomicAdd(a.n_words, n); }
    hslot = __shfl(hslot, 0); woff = __shfl(woff, 0);
    if (hslot >= a.hdr_cap || woff + n > a.words_cap) { if (ln == 0) atomicOr(a.err, WM_ERR_BURST_OVERFLOW); return; }
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
__global__ __launch_bounds__(256) void k3_bursts(K3Args a, uint32_t n_items)
{
    const uint32_t ln = threadIdx.x & 63u;
    for (uint32_t item = blockIdx.x * 4u + (threadIdx.x >> 6); item < n_items; item += gridDim.x * 4u) burst_item(a, item, ln);
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
                dst[n] = WM_CHIP_VAL(w) | ((uint32_t)rssi[row * g.Mcap + s * g.seg_len[algo] + WM_CHIP_POS(w)] << 8);
                if (pos) pos[n] = g.m0 + (uint64_t)s * g.seg_len[algo] + WM_CHIP_POS(w);
            }
    }
    *n_out = n;
}

#endif /* WM_K3_BURSTS_H */
