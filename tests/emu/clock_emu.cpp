/* clock_emu.cpp -- TEST INFRASTRUCTURE: the clock-recovery / time2 framer lanes (device source
 * rtl-wmbus_amd/csrc/wm_k2_clock.h) compiled for the host and executed lane by lane, with the
 * speculative-start / verify / re-run rounds of wm_api.hip around them, checkpoints and early exit
 * included.  Only the lane-private load path is emulated (what the kernel takes for batches that are
 * not a multiple of 64 captures, and for every re-run); the cooperative path differs in how a block
 * of soft symbols reaches LDS, not in what is computed from it. */
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "block_emu.h"

#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
struct uint4 { uint32_t x, y, z, w; };
struct float4 { float x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p |= v; return o; }
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31u)); }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_wave_barrier() { emu_wave_barrier(); }   /* a real meeting point on the block emulator */
#if !defined(__clang__)
static inline uint32_t __builtin_bitreverse32(uint32_t v)
{
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(v);
}
#endif
using std::min;

#include "wm_dev.h"
#include "wm_exact.h"
#include "wm_k2_common.h"
#include "wm_k2_clock.h"
#include "wm_k2_clock_sys.h"

extern "C" {

int wm_emu_sys = 0;                       /* 1: the systolic form (wm_k2_clock_sys.h): every 64 lanes a block of four waves on the block emulator */
void *wm_emu_states_out = nullptr;        /* optional: receives st_start then st_final ([2][S][nseg] WmClkState each) after the last round */

int wm_emu_descending = 1;
int wm_emu_s1_span = 0;                   /* WmPush.s1_span of the next calls */
uint32_t *wm_emu_seen_out = nullptr;      /* optional: receives the per-region "access-code chip seen" flags */
int wm_emu_chains = 1;                    /* K2Args.bad: a listed lane walks its chain (round 4); 0: every listed segment on its own */

/* One push of M decimated samples for S captures.  dphi: [2][S][Mcap] soft symbols; carry: [2][S]
 * WmClkState in/out; bits: [2][S][Mcap/32] out; chips: [2][S][nseg][cap]; counts: [2][S][nseg].
 * Returns the number of re-run lanes over all rounds, -1 if verification did not converge. */
long wm_emu_clock(const float *dphi, uint32_t S, uint32_t M, uint32_t Mcap, uint32_t flags, uint32_t seg_len, uint32_t warm0,
                  uint32_t warm1, uint32_t cap, void *carry, uint32_t *bits, uint32_t *chips, uint32_t *counts, uint32_t *err_out,
                  uint32_t *rounds_out)
{
    WmPush g{};
    g.M = M; g.Mcap = Mcap; g.S = S; g.flags = flags; g.d = 2;
    g.seg_len[1] = seg_len; g.nseg[1] = (M + seg_len - 1) / seg_len; g.nseg_cap[1] = g.nseg[1]; g.cap[1] = cap;
    g.warm[0] = warm0; g.warm[1] = warm1; g.s1_span = (uint32_t)wm_emu_s1_span;
    const uint32_t rows = 2 * S, nseg = g.nseg[1], lanes = rows * nseg;
    const uint32_t nck = seg_len / WM_CK_SAMPLES ? seg_len / WM_CK_SAMPLES - 1 : 0;
    std::vector<WmClkState> st_start((size_t)rows * nseg), st_final((size_t)rows * nseg);
    std::vector<uint32_t> bad((size_t)rows * nseg, 0);
    std::vector<uint32_t> seen((size_t)rows * nseg, 0), ckpt(std::max<size_t>(16, (size_t)rows * nseg * nck * 16), 0xDEADBEEFu), list;
    uint32_t err = 0;
    K2Args a{};
    a.g = g; a.dphi = dphi; a.bits = bits; a.chips = chips; a.counts = counts;
    a.st_start = st_start.data(); a.st_final = st_final.data(); a.st_carry = carry;
    a.algo = 1; a.err = &err; a.sync_seen = seen.data(); a.ckpt = ckpt.data(); a.nck = nck;
    a.bad = wm_emu_chains ? bad.data() : nullptr;
    static ClkLds<1> lds;
    static ClkSysLds sys_lds;
    const bool dc = flags & WM_F_DC;
    auto launch = [&](const uint32_t *lst, uint32_t n) {
        a.list = lst; a.n_lanes = n;
        if (wm_emu_sys == 1) {
            /* four coroutines per lane; the chunks of a launch in descending or ascending order (see below) */
            const uint32_t groups = (n + 63u) / 64u;
            for (uint32_t i = 0; i < groups; i++) {
                const uint32_t b = wm_emu_descending ? groups - 1 - i : i;
                block_emu::run_block(256, [&] {
                    if (lst) { if (dc) clock_sys_group<true, 1, false>(a, b, sys_lds); else clock_sys_group<false, 1, false>(a, b, sys_lds); }
                    else if (S % 64u == 0u) { if (dc) clock_sys_group<true, 0, true>(a, b, sys_lds); else clock_sys_group<false, 0, true>(a, b, sys_lds); }
                    else { if (dc) clock_sys_group<true, 0, false>(a, b, sys_lds); else clock_sys_group<false, 0, false>(a, b, sys_lds); }
                });
            }
            return;
        }
        /* On the GPU all lanes of a launch start together: a re-run lane usually reads its predecessor's
         * end state BEFORE that predecessor's own re-run (same launch) has replaced it -- which is what
         * makes cascading rounds.  Lanes in descending order reproduce that; ascending order is the
         * other extreme (every lane already sees its predecessor's new state). */
        if (lst == nullptr && S % 64u == 0u) {
            /* first pass of a batch of whole waves: the kernel loads COOPERATIVELY (8 lanes fetch one row's
             * line, the block is transposed through LDS between wave barriers) -- the 64 lanes of a wave
             * really have to run together: one coroutine each on the block emulator */
            for (uint32_t b = 0; b < n / 64; b++)
                block_emu::run_block(64, [&] { if (dc) clock_lanes<true, 1>(a, b, lds); else clock_lanes<false, 1>(a, b, lds); });
            return;
        }
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t l = wm_emu_descending ? n - 1 - i : i;
            threadIdx.x = l & 63u;
            if (dc) clock_lanes<true, 1>(a, l >> 6, lds); else clock_lanes<false, 1>(a, l >> 6, lds);
        }
    };
    launch(nullptr, lanes);
    long reruns = 0;
    uint32_t round = 0;
    for (;; round++) {
        list.clear();
        for (uint32_t lane = 0; lane < lanes; lane++) {                  /* k2_verify */
            uint32_t ch, stream, seg;
            lane_decode(g, 1, lane, ch, stream, seg);
            if (!(g.flags & (ch ? WM_F_S1 : WM_F_T1C1)) || seg == 0) continue;
            const size_t sidx = ((size_t)ch * S + stream) * nseg + seg;
            const bool differs = std::memcmp(&st_start[sidx], &st_final[sidx - 1], sizeof(WmClkState)) != 0;
            bad[((size_t)ch * nseg + seg) * S + stream] = differs;          /* the verdict per segment ([chain][segment][capture]), stable during the launch that follows */
            /* a launch that walks chains is given the HEADS of runs of listed segments only (k2_verify_lane's `heads`; lanes go segment-major,
             * so the predecessor's verdict of this round is there already) */
            const bool walks = wm_emu_chains && round >= 1;
            if (differs && !(walks && seg > 1 && bad[((size_t)ch * nseg + seg - 1) * S + stream])) list.push_back(lane);
        }
        if (list.empty()) break;
        if (round > nseg + 1) return -1;
        reruns += (long)list.size();
        a.bad = (wm_emu_chains && round >= 1) ? bad.data() : nullptr;   /* the first list round re-runs lone segments (wm_api.hip fr_launch) */
        launch(list.data(), (uint32_t)list.size());
    }
    WmClkState *c = (WmClkState *)carry;                                 /* k_carry */
    for (uint32_t r = 0; r < rows; r++) c[r] = st_final[(size_t)r * nseg + nseg - 1];
    if (wm_emu_seen_out) std::memcpy(wm_emu_seen_out, seen.data(), seen.size() * sizeof(uint32_t));
    if (wm_emu_states_out) {
        std::memcpy(wm_emu_states_out, st_start.data(), st_start.size() * sizeof(WmClkState));
        std::memcpy((char *)wm_emu_states_out + st_start.size() * sizeof(WmClkState), st_final.data(), st_final.size() * sizeof(WmClkState));
    }
    if (err_out) *err_out = err;
    if (rounds_out) *rounds_out = round;
    return reruns;
}

unsigned wm_emu_clock_state_bytes(void) { return sizeof(WmClkState); }

}
