/* wm_k2_common.h -- what the two framer kernels share: launch arguments and the lane numbering.
 * Device code, included by wm_kernels.hip (one translation unit, see the overview there). */
#ifndef WM_K2_COMMON_H
#define WM_K2_COMMON_H

/* =============================================================================================
 * K2: sequential lanes over time segments
 * ===========================================================================================*/
struct K2Args {
    WmPush g;
    const float *dphi;
    const uint8_t *rssi;
    uint32_t *bits;            /* [2][S][Mcap/32]                                          */
    uint32_t *chips;           /* base of this algo's regions: [2][S][nseg_cap][cap]       */
    uint32_t *counts;          /* [2][S][nseg_cap]                                          */
    void *st_start;            /* state each segment's main loop started from               */
    void *st_final;            /* state after the segment's last sample                     */
    void *st_carry;            /* [2][S] exact state carried from the previous push         */
    const uint32_t *list;      /* re-run list of lane ids, or nullptr                       */
    uint32_t n_lanes;
    const uint32_t *n_ptr;     /* if set: the lane count lives on the device (a re-run list k2_verify has just written;
                                  the launch then has a fixed grid whose blocks walk the list) */
    uint32_t algo;             /* WMBUS_ALGO_* of this launch                              */
    uint32_t *err;
    uint32_t *sync_seen;       /* [2][S][nseg_cap]: set when a pass emitted an access-code chip into the region */
    /* checkpoints of the speculative pass, every WM_CK_SAMPLES inside a segment: lane state + chips so
     * far (16 words each).  A re-run stops at the first checkpoint it reproduces: from there on the
     * speculative pass had already been on the exact trajectory. */
    uint32_t *ckpt; uint32_t nck;
    /* [2][nseg_cap][S]: k2_verify's verdict per segment (1: its start did not match its predecessor's end), for the chain
     * walk of the framer kernels' re-run lanes (clock_lanes, rla_lanes); nullptr: every listed segment on its own, as in round 3 */
    const uint32_t *bad;
};

/* Issue priority of the latency-bound kernels behind K1 (framers, verifiers, burst kernels): a clock wave that shares its
 * SIMD with seven demodulation waves gets an eighth of the issue slots under the default round robin, and a context's
 * chain of dependent launches is as long as that makes it.  Build-time switch (0 = leave the default). */
#ifndef WM_FRAMER_PRIO
#define WM_FRAMER_PRIO 0
#endif
#if WM_FRAMER_PRIO
__device__ __forceinline__ void wm_framer_prio() { __builtin_amdgcn_s_setprio(WM_FRAMER_PRIO); }
#else
__device__ __forceinline__ void wm_framer_prio() {}
#endif

__device__ __forceinline__ uint32_t k2_lane_count(const K2Args &a) { return a.n_ptr ? *a.n_ptr : a.n_lanes; }

__device__ __forceinline__ void lane_decode(const WmPush &g, uint32_t algo, uint32_t lane, uint32_t &ch, uint32_t &stream, uint32_t &seg)
{
    /* lane = (ch * nseg + seg) * S + stream : neighbouring lanes = neighbouring streams */
    stream = lane % g.S;
    const uint32_t r = lane / g.S;
    seg = r % g.nseg[algo];
    ch = r / g.nseg[algo];
}

/* Chip i of segment region sidx of framer `algo` (see WmSpill): the primary region, then the segment's chunks. */
__device__ __forceinline__ const uint32_t *wm_chip_ptr(const WmPush &g, const uint32_t *primary, uint32_t algo, uint64_t sidx, uint32_t i)
{
    const uint32_t cap = g.cap[algo];
    if (algo != 0u || i < cap) return primary + sidx * cap + i;
    const uint32_t j = i - cap;
    return g.sp.arena + g.sp.chain[sidx * WM_SPILL_LEVELS + j / WM_SPILL_CHUNK] + j % WM_SPILL_CHUNK;
}

#endif /* WM_K2_COMMON_H */
