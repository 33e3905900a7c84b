/* k3_emu.cpp -- TEST INFRASTRUCTURE: k3_scan and k3_bursts (device source wm_k3_bursts.h) on the coroutine
 * block emulator: access-code hits of settled chip regions, then the bursts the host decoders get. */
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "block_emu.h"

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static                  /* one block runs at a time: block-shared = static */
#define WM_WAVE_SYNC() emu_wave_barrier()
#define WM_PEEK(p) (*(p))
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
using std::max;
using std::min;
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline uint32_t __brev(uint32_t v) { uint32_t r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p += v; return o; }
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p |= v; return o; }

#include "wm_dev.h"
#include "wm_k2_common.h"
#include "wm_k3_bursts.h"

extern "C" {

unsigned wm_emu_hdr_bytes(void) { return sizeof(WmBurstHdr); }

static WmSpill emu_spill = {};            /* run-length chips beyond the primary regions, as the framer emulation left them */
void wm_emu_k3_set_spill(uint32_t *arena, uint32_t arena_words, uint32_t *chain, uint32_t *nchain, uint32_t *used)
{
    emu_spill.arena = arena; emu_spill.arena_words = arena_words; emu_spill.chain = chain; emu_spill.nchain = nchain; emu_spill.used = used;
}

/* geo: M, Mcap, flags, m0, seg_len[2], nseg[2], cap[2] (S = 1).  Returns the number of bursts, or -1 on overflow. */
/* optional: where the bursts decoded by the kernel go (WmPkt array, byte arena); unset = every burst as chips */
static WmPkt *emu_pkts = nullptr; static uint32_t emu_pkts_cap = 0, *emu_n_pkts = nullptr; static uint8_t *emu_bytes = nullptr; static uint32_t emu_bytes_cap = 0;
void wm_emu_k3_set_decode(void *pkts, uint32_t pkts_cap, uint32_t *n_pkts, uint8_t *bytes, uint32_t bytes_cap)
{
    emu_pkts = (WmPkt *)pkts; emu_pkts_cap = pkts_cap; emu_n_pkts = n_pkts; emu_bytes = bytes; emu_bytes_cap = bytes_cap;
}
unsigned wm_emu_pkt_bytes(void) { return sizeof(WmPkt); }

/* optional: RSSI on demand.  k3_spans flags the demodulation tiles whose RSSI the bursts read (flags [ntiles], S = 1);
 * k3_bursts then gets a copy of the RSSI rows in which every OTHER tile reads 0 -- below the decoders' capture threshold,
 * so a read that k3_spans did not announce changes what comes out. */
static uint32_t *emu_span_flags = nullptr; static uint32_t emu_span_tiles = 0, emu_span_listed = 0;
void wm_emu_k3_set_spans(uint32_t *flags, uint32_t ntiles) { emu_span_flags = flags; emu_span_tiles = ntiles; }
unsigned wm_emu_k3_spans_listed(void) { return emu_span_listed; }

long wm_emu_k3(const uint64_t *geo, const uint32_t *chips0, const uint32_t *chips1, const uint32_t *counts0, const uint32_t *counts1,
               const uint32_t *seen0, const uint32_t *seen1, const uint8_t *rssi, const uint32_t *pending, void *hdr_out, uint32_t hdr_cap,
               uint32_t *words_out, uint32_t words_cap, uint32_t *n_words_out, uint32_t max_blocks)
{
    WmPush g{};
    g.S = 1; g.d = 2; g.M = (uint32_t)geo[0]; g.Mcap = (uint32_t)geo[1]; g.flags = (uint32_t)geo[2]; g.m0 = geo[3];
    for (int a = 0; a < 2; a++) {
        g.seg_len[a] = (uint32_t)geo[4 + a]; g.nseg[a] = (uint32_t)geo[6 + a]; g.nseg_cap[a] = g.nseg[a]; g.cap[a] = (uint32_t)geo[8 + a];
    }
    g.sp = emu_spill;
    std::vector<uint2> hits(hdr_cap);
    uint32_t n_hits = 0, err = 0, n_hdr = 0, n_words = 0;
    const uint32_t lanes = 2u * (g.nseg[0] + g.nseg[1]) * g.S;
    const uint32_t scan_parts = (std::max(g.cap[0] + WM_SPILL_LEVELS * WM_SPILL_CHUNK, g.cap[1]) + WM_K3_SCAN_PART - 1u) / WM_K3_SCAN_PART;   /* as launch_k3 */
    gridDim = {(lanes + 255u) / 256u, scan_parts, 1};
    for (uint32_t part = 0; part < scan_parts; part++)
        for (uint32_t b = 0; b < gridDim.x; b++) {
            blockIdx = {b, part, 0};
            block_emu::run_block(256, [&] { k3_scan(g, chips0, chips1, counts0, counts1, seen0, seen1, hits.data(), &n_hits, hdr_cap, &err); });
        }
    blockIdx = {0, 0, 0};
    K3Args k3{};
    k3.g = g; k3.rssi = rssi; k3.chips[0] = chips0; k3.chips[1] = chips1; k3.counts[0] = counts0; k3.counts[1] = counts1;
    k3.hits = hits.data(); k3.n_hits = &n_hits; k3.hits_cap = hdr_cap; k3.pending = pending;
    k3.hdr = (WmBurstHdr *)hdr_out; k3.hdr_cap = hdr_cap; k3.words = words_out; k3.words_cap = words_cap;
    k3.n_hdr = &n_hdr; k3.n_words = &n_words; k3.err = &err;
    uint32_t n_bytes = 0;
    if (emu_pkts) { *emu_n_pkts = 0; k3.pkts = emu_pkts; k3.pkts_cap = emu_pkts_cap; k3.n_pkts = emu_n_pkts; k3.bytes = emu_bytes; k3.bytes_cap = emu_bytes_cap; k3.n_bytes = &n_bytes; }
    const uint32_t n_items = 0xFFFFFFFFu;                  /* the kernel reads the hit count itself, as in the product */
    const uint32_t n_items_host = 4 * g.S + std::min(n_hits, hdr_cap);
    gridDim = {std::max(1u, std::min((n_items_host + 3u) / 4u, max_blocks)), 1, 1};
    std::vector<uint8_t> masked;
    if (emu_span_flags) {
        const uint32_t T = WM_K1_TILE2, nt = emu_span_tiles;
        std::vector<uint32_t> list(nt);
        uint32_t n_list = 0;
        static std::vector<WmItemRec> plans;                 /* k3_spans leaves its findings for k3_bursts, as in the product */
        plans.assign((size_t)4 * g.S + hdr_cap, WmItemRec{0xDEADu, 0xDEADu, 0xDEADu, WmPlan{}});
        k3.plans = plans.data();
        std::fill(emu_span_flags, emu_span_flags + nt, 0u);
        for (uint32_t b = 0; b < gridDim.x; b++) {
            blockIdx = {b, 0, 0};
            block_emu::run_block(256, [&] { k3_spans(k3, T, nt, emu_span_flags, list.data(), &n_list); });
        }
        emu_span_listed = n_list;
        for (uint32_t i = 0; i < n_list; i++) if (list[i] >= nt || !emu_span_flags[list[i]]) return -2;       /* listed = flagged, once each */
        uint32_t flagged = 0;
        for (uint32_t tl = 0; tl < nt; tl++) flagged += emu_span_flags[tl] != 0;
        if (flagged != n_list) return -3;
        masked.assign(rssi, rssi + 2 * (size_t)g.Mcap);
        for (uint32_t ch = 0; ch < 2; ch++)
            for (uint32_t tl = 0; tl < nt; tl++)
                if (!((emu_span_flags[tl] >> ch) & 1u))
                    for (uint32_t m = tl * T; m < std::min((tl + 1) * T, g.Mcap); m++) masked[(size_t)ch * g.Mcap + m] = 0;
        k3.rssi = masked.data();
    }
    for (uint32_t b = 0; b < gridDim.x; b++) {
        blockIdx = {b, 0, 0};
        block_emu::run_block(256, [&] { k3_bursts(k3, n_items); });
    }
    if (n_words_out) *n_words_out = n_words;
    return err ? -1 : (long)n_hdr;
}

}
