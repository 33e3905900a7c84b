/* rla_emu.cpp -- TEST INFRASTRUCTURE (never part of the product): the run-length framer's device
 * code (rtl-wmbus_amd/csrc/wm_k2_rla.h, the very source hipcc compiles for gfx950) built for the
 * host behind a small shim and executed lane by lane, with the speculative-start / verify / re-run
 * loop of wm_api.hip's run_segments around it.  Lanes of this kernel never talk to each other (no
 * barrier, no shuffle), so running them one after the other is the same computation.  Purpose: the
 * framer logic can be checked against the oracle -- and changed -- on a box without a GPU; the GPU
 * tests remain the proof for the compiled kernel. */
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

/* ---- the shim: what the device header needs from the HIP language ---------------------------- */
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct Idx3 { uint32_t x, y, z; };
static Idx3 threadIdx, blockIdx, gridDim;
static inline void __syncthreads(void) {}                           /* the kernels themselves are not run here (lanes one by one) */
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline float __frcp_rn(float x) { return 1.0f / x; }          /* correctly rounded on both sides */
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p |= v; return o; }
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p += v; return o; }
using std::min;

/* measurements (tools/rla_trips.py): edge trips of the first pass per (row, 64 samples) */
extern "C" { uint32_t *wm_emu_trip_tab = nullptr; uint32_t wm_emu_trip_words = 0; }
#define WM_RLA_TRIP_HOOK(stream, sample) do { if (wm_emu_trip_tab && !rerun && (sample) >= mb) wm_emu_trip_tab[(size_t)row * wm_emu_trip_words + ((sample) >> 6)]++; } while (0)

#include "wm_dev.h"
#include "wm_k2_common.h"
#include "wm_k2_rla.h"

extern "C" {

int wm_emu_descending = 1;
int wm_emu_last_rounds = 0;               /* list rounds the last call needed */
int wm_emu_chains = 1;                    /* K2Args.bad: a listed lane walks its chain of listed segments (round 4); 0: every listed segment on its own */
/* optional spill storage (WmSpill, wm_dev.h) for the next wm_emu_rla call: arena, chain [2][S][nseg][WM_SPILL_LEVELS],
 * nchain [2][S][nseg] followed by the bump counter; all zero = none */
static WmSpill emu_spill = {};
void wm_emu_rla_set_spill(uint32_t *arena, uint32_t arena_words, uint32_t *chain, uint32_t *nchain, uint32_t *used)
{
    emu_spill.arena = arena; emu_spill.arena_words = arena_words; emu_spill.chain = chain; emu_spill.nchain = nchain; emu_spill.used = used;
}
unsigned wm_emu_spill_chunk(void) { return WM_SPILL_CHUNK; }
unsigned wm_emu_spill_levels(void) { return WM_SPILL_LEVELS; }
uint32_t *wm_emu_seen_out = nullptr;      /* optional: receives the per-region "access-code chip seen" flags */

/* One push of `M` decimated samples for S captures.  bits: [2][S][Mcap/32] slicer words; carry:
 * [2][S] WmRlaState in/out (zero-initialised = the reset state is NOT implied: pass what
 * wmbus_open would, see rla_reset_state).  chips: [2][S][nseg][cap], counts: [2][S][nseg].
 * Returns the number of re-run lanes (all rounds), or -1 if verification did not converge. */
long wm_emu_rla(const uint32_t *bits, uint32_t S, uint32_t M, uint32_t Mcap, uint32_t flags, uint32_t seg_len,
                uint32_t lookback, uint32_t cap, void *carry, uint32_t *chips, uint32_t *counts, uint32_t *err_out)
{
    WmPush g{};
    g.M = M; g.Mcap = Mcap; g.S = S; g.flags = flags; g.d = 2;
    g.seg_len[0] = seg_len; g.nseg[0] = (M + seg_len - 1) / seg_len; g.nseg_cap[0] = g.nseg[0]; g.cap[0] = cap;
    g.lookback = lookback;
    g.sp = emu_spill;
    const uint32_t rows = 2 * S, nseg = g.nseg[0], lanes = rows * nseg;
    std::vector<WmRlaState> st_start((size_t)rows * nseg), st_final((size_t)rows * nseg);
    std::vector<uint32_t> seen((size_t)rows * nseg, 0), bad((size_t)rows * nseg, 0), list;
    uint32_t err = 0;
    K2Args a{};
    a.g = g; a.bits = const_cast<uint32_t *>(bits); a.chips = chips; a.counts = counts;
    a.st_start = st_start.data(); a.st_final = st_final.data(); a.st_carry = carry;
    a.algo = 0; a.err = &err; a.sync_seen = seen.data();
    a.bad = wm_emu_chains ? bad.data() : nullptr;
    static RlaLds lds;
    rla_lds_init(lds, 0, 1);
    const uint32_t B = 64 * WM_RLA_WPB;
    auto launch = [&](const uint32_t *lst, uint32_t n) {
        a.list = lst; a.n_lanes = n;
        /* descending lane order: a re-run lane reads its predecessor's end state before that predecessor's
         * own re-run of the same launch replaces it, as it mostly happens on the GPU (cascading rounds) */
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t l = wm_emu_descending ? n - 1 - i : i;
            threadIdx.x = l % B;
            rla_lanes(a, l / B, lds);
        }
    };
    launch(nullptr, lanes);
    long reruns = 0;
    for (uint32_t round = 0;; round++) {
        list.clear();
        for (uint32_t lane = 0; lane < lanes; lane++) {                  /* k2_verify */
            uint32_t ch, stream, seg;
            lane_decode(g, 0, lane, ch, stream, seg);
            if (!(g.flags & (ch ? WM_F_S1 : WM_F_T1C1)) || seg == 0) continue;
            const size_t sidx = ((size_t)ch * S + stream) * nseg + seg;
            const bool differs = std::memcmp(&st_start[sidx], &st_final[sidx - 1], sizeof(WmRlaState)) != 0;
            bad[((size_t)ch * nseg + seg) * S + stream] = differs;          /* the verdict per segment ([chain][segment][capture]), stable during the launch that follows */
            /* a launch that walks chains is given the HEADS of runs of listed segments only (k2_verify_lane's `heads`; lanes go segment-major,
             * so the predecessor's verdict of this round is there already) */
            const bool walks = wm_emu_chains && round >= 1;
            if (differs && !(walks && seg > 1 && bad[((size_t)ch * nseg + seg - 1) * S + stream])) list.push_back(lane);
        }
        wm_emu_last_rounds = (int)round;
        if (list.empty()) break;
        if (round > nseg + 1) return -1;
        reruns += (long)list.size();
        a.bad = (wm_emu_chains && round >= 1) ? bad.data() : nullptr;   /* the first list round re-runs lone segments (wm_api.hip fr_launch) */
        launch(list.data(), (uint32_t)list.size());
    }
    WmRlaState *c = (WmRlaState *)carry;                                 /* k_carry */
    for (uint32_t r = 0; r < rows; r++) c[r] = st_final[(size_t)r * nseg + nseg - 1];
    if (wm_emu_seen_out) std::memcpy(wm_emu_seen_out, seen.data(), seen.size() * sizeof(uint32_t));
    if (err_out) *err_out = err;
    return reruns;
}

/* the state a fresh context starts from (wm_api.hip: wmbus_open / rtl_wmbus.c:628-637,717-726) */
void wm_emu_rla_reset_state(void *st) { const WmRlaState r = {0, 8 * 256, 0, 0u, 0u, 0u, 24, 24}; *(WmRlaState *)st = r; }
unsigned wm_emu_rla_state_bytes(void) { return sizeof(WmRlaState); }
unsigned wm_emu_chip_pos_shift(void) { return 3; }

}
