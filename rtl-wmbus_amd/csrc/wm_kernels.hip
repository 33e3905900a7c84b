/*
 * wm_kernels.hip -- gfx950 kernels for the rtl-wmbus hot path.
 *
 *   k1_demod2    time-parallel front end, one 976-sample tile per block: cu8 -> [+-325 kHz
 *                shift] -> integer boxcars -> decimate -> polar discriminator (exact fdlibm
 *                atan2f) -> FIR low-pass -> soft symbol; |s| -> EMA -> RSSI byte.
 *                                  (rtl_wmbus.c:1310-1352, 517-586, 369-392, 475-495, 1066-1067)
 *   k1_demod_ppf the same behind the polyphase pre-filter of ppf.h (option; rtl_wmbus.c:258-294).
 *   k1_verify / k1_collect / k1_commit   certify the EMA hand-offs between tiles; an uncertified
 *                tile is repaired by a list launch of k1 (one lane per chain, sequential).
 *   k2_clock     lane = (chain, stream, time segment): DC remover, slicer, squared-signal IIR
 *                band-pass, clock lock, time2 framer.            (rtl_wmbus.c:497-515, 1059,
 *                336-365, 1089-1111, 806-852)
 *   k2_rla       lane = (chain, stream, time segment): run-length framer with deglitch and
 *                bit-length tracking.                            (rtl_wmbus.c:617-803)
 *   k2_verify    compares each segment's start state with its predecessor's end state; a
 *                segment whose speculative start was wrong is re-run from the true state (and
 *                stops at the first checkpoint of the speculative pass it reproduces).
 *   k3_bursts    access-code hits -> the chips (with their RSSI bytes) a host packet decoder
 *                consumes.   k4_flatten: debug/parity view of a whole chip stream.
 *
 * Exactness: every float operation the reference performs is performed here in the same order
 * with separate roundings (wm_exact.h; the file is also built with -ffp-contract=off).  The
 * recurrences (EMA, DC, IIR, run-length state) are evaluated sequentially inside a lane; time
 * parallelism comes from starting lanes early (warm-up / look-back) and PROVING, by bitwise state
 * comparison at the hand-off, that the lane had converged onto the sequential trajectory.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wm_dev.h"
#include "wm_exact.h"

#include "wm_k1_demod.h"
#include "wm_k2_common.h"
#include "wm_k2_clock.h"
#include "wm_k2_clock_sys.h"
#include "wm_k2_rla.h"

/* start[seg] must equal final[seg-1]; mismatching lanes are appended to `list`. */
/* heads: the launch that reads this list walks CHAINS (rla_lanes / clock_lanes / clock_sys_group with K2Args.bad): of a run of consecutive
 * listed segments only the first does anything there, the others return at once -- so only the first is listed (configs[2] listed 11 000
 * segments per context-push for 6 900 walks: 172 waves in the list launch instead of 108).
 * (Round 6 also measured the list in two parts -- the heads of runs of three or more segments in waves of their own: a long run is the
 * T1/C1 chain's framer inside a T1 / C1 telegram, whose chip clock is a PI loop with an integral that never resets there,
 * rtl_wmbus.c:790-796, so no speculative start inside the telegram can match: walks of up to 15 segments in configs[2], a quarter of that
 * chain's walks longer than three -- a clean signal with a third of the noise's edges per step.  No gain: the walk is 120 steps of 16
 * trips whatever shares its wave; removed.) */
__device__ __forceinline__ void k2_verify_lane(const WmPush &g, uint32_t algo, uint32_t lane, const uint32_t *st_start, const uint32_t *st_final, uint32_t words,
                                               uint32_t *list, uint32_t *n_list, uint32_t *bad, uint32_t heads)
{
    if (lane >= 2u * g.nseg[algo] * g.S) return;
    uint32_t ch, stream, seg;
    lane_decode(g, algo, lane, ch, stream, seg);
    if (!(g.flags & (ch ? WM_F_S1 : WM_F_T1C1)) || seg == 0) return;
    const uint64_t sidx = ((uint64_t)ch * g.S + stream) * g.nseg_cap[algo] + seg;
    const uint32_t *p = st_start + sidx * words, *q = st_final + (sidx - 1) * words;
    bool same = true;
    for (uint32_t k = 0; k < words; k++) same &= p[k] == q[k];
    /* the verdict per segment, laid out [chain][segment][capture] so that a wave's 64 captures store one line (capture-major it
     * was 200 000 scattered 4-byte stores per verification: read-modify-write traffic worth 2 ms of every step) */
    if (bad) bad[((uint64_t)ch * g.nseg_cap[algo] + seg) * g.S + stream] = same ? 0u : 1u;
    if (same) return;
    if (heads && bad && seg > 1u) {                      /* my predecessor's verdict, worked out here as its own thread does (segment 0 is never listed) */
        const uint32_t *pp = p - words, *qp = q - words;
        bool prev_same = true;
        for (uint32_t k = 0; k < words; k++) prev_same &= pp[k] == qp[k];
        if (!prev_same) return;                          /* inside a run: its head walks me */
    }
    list[atomicAdd(n_list, 1u)] = lane;
}

__global__ void k2_verify(WmPush g, uint32_t algo, const uint32_t *st_start, const uint32_t *st_final, uint32_t words,
                          uint32_t *list, uint32_t *n_list, uint32_t *bad, uint32_t heads)
{
    wm_framer_prio();
    k2_verify_lane(g, algo, blockIdx.x * blockDim.x + threadIdx.x, st_start, st_final, words, list, n_list, bad, heads);
}

/* The end of a push's framer stage in ONE launch (round 6: it was four): the last verification of either framer and the two
 * carries -- the end state of every row's last segment becomes the next push's start state.  (If a verification still lists
 * segments, wmbus_collect finishes them with the host in the loop and carries again.) */
struct K2Finish {
    const uint32_t *st_start[2], *st_final[2];           /* [framer] */
    uint32_t *list[2], *n_list[2], *bad[2], *carry[2];
    uint32_t words[2], verify[2], do_carry[2], heads[2];
};
__global__ void k2_finish(WmPush g, K2Finish f)
{
    wm_framer_prio();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t algo = 0; algo < 2u; algo++) {
        if (f.verify[algo]) k2_verify_lane(g, algo, t, f.st_start[algo], f.st_final[algo], f.words[algo], f.list[algo], f.n_list[algo], f.bad[algo], f.heads[algo]);
        if (f.do_carry[algo] && t < 2u * g.S)
            for (uint32_t k = 0; k < f.words[algo]; k++)
                f.carry[algo][(uint64_t)t * f.words[algo] + k] = f.st_final[algo][((uint64_t)t * g.nseg_cap[algo] + g.nseg[algo] - 1u) * f.words[algo] + k];
    }
}

#include "wm_k3_bursts.h"
