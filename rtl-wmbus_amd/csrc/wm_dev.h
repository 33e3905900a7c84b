/*
 * wm_dev.h -- layouts shared by the HIP kernels (wm_kernels.hip) and the host pipeline
 * (wm_api.hip).  See DESIGN.md for the data-flow picture.
 *
 * HBM layout (per context, S = n_streams, chain c in {T1C1, S1}, algo a in {RLA, T2A}):
 *   in     u8   [S][HIST_BYTES + max_push_bytes + SLACK]   cu8; HIST = tail of the previous push
 *   dphi   f32  [2][S][Mcap]      soft symbol (FIR output), one per decimated sample
 *   rssi   u8   [2][S][Mcap]      (unsigned) of the filtered magnitude
 *   bits   u32  [2][S][Mcap/32]   slicer output, bit j of word w = sample 32w+j
 *   chips  u32  [2][2][S][nseg][cap_a]  per time segment; word = pos<<3 | value (bit, sync, reset); the RSSI of
 *                                       a chip is rssi[row][sample of the chip] (K3 / K4 insert it into bits 15:8)
 *   state arrays, burst arena (see structs below)
 */
#ifndef WM_DEV_H
#define WM_DEV_H

#include <stdint.h>

#define WM_HIST_BYTES   4096u      /* input history kept in front of each push (2048 samples) */
#define WM_IN_SLACK     256u       /* readable slack behind the staged bytes                  */
#define WM_K1_HALO      48         /* decimated-sample halo: 45 FIR + 1 discriminator, >= EMA warm-up */
#ifndef WM_EMA_WARMUP
#define WM_EMA_WARMUP   32         /* EMA warm-up before a lane's run: trajectories coalesce bitwise within 23
                                      samples (measured); an uncertified hand-off is repaired exactly, not an error */
#endif
#define WM_K1_TILE2     976        /* K1 tile: tile + halo = 1024 = 256 threads x 4           */
#ifndef WM_CLK_WPB
#define WM_CLK_WPB      4          /* clock kernel: independent waves per block (see k2_clock) */
#endif
#ifndef WM_RLA_WPB
#define WM_RLA_WPB      4          /* run-length kernel: waves per block, for the same reason */
#endif
#define WM_CK_SAMPLES   2048       /* clock kernel: distance of the speculative pass's state checkpoints */
#define WM_MAX_DECIM    16u        /* staging for d = 16 with -s: 131 KB of the 160 KB LDS */

#define WM_CHIP_WORD(pos, val) (((pos) << 3) | (val))
#define WM_CHIP_VAL(w)   ((w) & 7u)
#define WM_CHIP_POS(w)   ((w) >> 3)

/* Sync words and window lengths (rtl_wmbus.c:97-103; longest frames t1_c1_packet_decoder.h:95). */
#define WM_SYNC_T1C1      0x543Du
#define WM_SYNC_T1C1_MASK 0xFFFFu
#define WM_SYNC_S1        0x547696u
#define WM_SYNC_S1_MASK   0xFFFFFFu
#define WM_MAXCHIPS_T1C1  (12u * 290u + 1u)
#define WM_MAXCHIPS_S1    (16u * 290u + 1u)
#define WM_RLA_RUN_LIMIT   8192u      /* chips materialised per edge (> WM_MAXCHIPS_*: lossless for the decoders) */

/* Clock-recovery lane state (12 words): IIR history (iir.h:36-45, h1/h2 per section), DC
 * remover (rtl_wmbus.c:497-515), clock lock (rtl_wmbus.c:1043-1044), time2 shift register. */
struct WmClkState {
    float    h[6];
    float    dc_x, dc_y;
    uint32_t clk;          /* the last three clock levels, newest in bit 0              */
    uint32_t sr;           /* time2 chip shift register, masked to the sync length     */
    uint32_t pad[2];
};

/* Run-length framer lane state (8 words), rtl_wmbus.c:617-637 / 705-726. */
struct WmRlaState {
    int32_t  run, bitlen, cum;
    uint32_t state;        /* bit0 = deglitched level, bit1 = reset pending (ours) */
    uint32_t raw;          /* last five raw slicer bits in time order (bit 4 = newest) */
    uint32_t sr;           /* chip shift register, masked                          */
    int32_t  spb0, spb1;   /* S1 samples-per-bit trackers                          */
};

/* Run-length chip storage beyond a segment's primary region.  The reference's bit-length tracker has no floor
 * (rtl_wmbus.c:792-796): an interferer, or a switch combination like -d 3 -s -o -a on a two-mode capture, makes it
 * emit several chips per sample for a while, and its chip loop (:765-779) never gives up.  A segment's primary region
 * holds cap[0] chips; a lane that needs more takes WM_SPILL_CHUNK-word chunks from a per-push arena (atomic bump
 * allocator) and records them in the segment's chain, which later passes over the same segment (re-runs) reuse.
 * Chip i of a segment lives at primary[i] for i < cap[0], else in chunk (i - cap[0]) / WM_SPILL_CHUNK of its chain.
 * Only when the arena (or the chain) is exhausted are chips dropped -- counted, reported as a warning, never an error. */
#define WM_SPILL_CHUNK  2048u      /* words; a multiple of 8 (chips leave in 32-byte groups) */
#define WM_SPILL_LEVELS 16u        /* chunks per segment chain: 32 768 chips beyond the primary region */
struct WmSpill {
    uint32_t *arena;         /* [arena_words], nullptr: no spill storage (host emulation of single kernels) */
    uint32_t *used;          /* words handed out this push                                     */
    uint32_t *chain;         /* [2][S][nseg_cap[0]][WM_SPILL_LEVELS] arena offsets            */
    uint32_t *nchain;        /* [2][S][nseg_cap[0]] chunks the segment owns (reset per push)   */
    uint32_t arena_words;
};

/* Per-push geometry handed to every kernel by value. */
struct WmPush {
    const uint8_t *in;       /* base of stream 0's buffer (history first)             */
    uint64_t in_stride;      /* bytes between streams                                 */
    uint64_t n0;             /* global input-sample index of the first new sample     */
    uint64_t m0;             /* global decimated index of the first new decimated one */
    uint32_t n_new;          /* new input samples this push                           */
    uint32_t M;              /* new decimated samples this push                       */
    uint32_t Mcap;           /* row pitch of dphi/rssi (elements), multiple of 128     */
    uint32_t d;              /* decimation                                            */
    uint32_t S;              /* streams                                               */
    uint32_t lut_n;          /* 32*d                                                  */
    uint32_t lut_phase0;     /* (13*n0) mod lut_n                                     */
    uint32_t flags;          /* WM_F_*                                                */
    /* time segmentation, indexed by framer (0 = run-length, 1 = clock/time2): the two kernels
     * want different segment lengths (cheap look-back vs. long IIR warm-up) */
    uint32_t seg_len[2];     /* C                                                     */
    uint32_t nseg[2];        /* ceil(M / C)                                           */
    uint32_t nseg_cap[2];    /* segment pitch of the state / count arrays             */
    uint32_t cap[2];         /* chips per segment region                              */
    uint32_t warm[2];        /* IIR warm-up per chain                                 */
    uint32_t lookback;       /* RLA speculative lookback                              */
    uint32_t s1_span;        /* 2: an S1 clock lane covers two consecutive segments (its warm-up is twice T1/C1's, so at the
                                same segment length it re-reads 75 % instead of 37 %); 0 / 1: one segment per lane */
    WmSpill sp;              /* run-length chips beyond cap[0]                        */
};

enum {
    WM_F_SHIFT = 1, WM_F_ACCURATE = 2, WM_F_DC = 4, WM_F_T1C1 = 8, WM_F_S1 = 16,
    WM_F_RLA = 32, WM_F_T2A = 64,
    WM_F_APPROX1 = 128, WM_F_APPROX2 = 256     /* atan2_approximation / atan2_approximation2 instead of cargf (options) */
};

/* error word bits (device -> host).  CHIP_OVERFLOW: a time2 region over its proven bound (an internal error);
 * CHIP_TRUNC / BURST_OVERFLOW: storage exhausted, chips / bursts dropped -- warnings, the push still succeeds. */
enum { WM_ERR_CHIP_OVERFLOW = 2, WM_ERR_BURST_OVERFLOW = 4, WM_ERR_CHIP_TRUNC = 8 };

/* Burst = the chips a packet decoder needs after one access-code hit. */
struct WmBurstHdr {
    uint32_t stream;
    uint8_t  chain, algo;
    uint16_t flags;          /* bit0: continuation of a burst cut by the previous push */
    uint32_t chip0;          /* index of the first chip in this push's chip stream     */
    uint32_t n_chips;
    uint64_t pos0;           /* global decimated index of the first chip               */
    uint32_t word_off;       /* offset of the chip words in the arena (u32 units)      */
    uint32_t avail;          /* chips from chip0 to the end of this push's chip stream   */
};
/* burst chip word: [31:11] sample delta from pos0, [10:3] rssi, [2:0] value */

/* A burst that lies entirely inside the push is decoded on the GPU (k3_bursts): the host gets the assembled telegram
 * instead of its chips.  `consumed` = chips the reference's decoder takes from the access-code chip on (that chip
 * included) before it is idle again -- the host needs it to skip access codes that pass meanwhile. */
struct WmPkt {
    uint32_t stream;
    uint8_t  chain, algo;
    uint8_t  status;         /* WM_PKT_DONE: telegram assembled; WM_PKT_ABORT: the decoder gave up after `consumed` chips */
    uint8_t  flags;          /* WM_PKTF_* */
    uint32_t chip0;          /* index of the access-code chip in this push's chip stream */
    uint32_t consumed;
    uint64_t sample;         /* global decimated index of the completing chip (DONE)     */
    uint32_t off;            /* byte offset of the telegram in the byte arena            */
    uint16_t L;              /* expected length with CRC bytes (the decoder's L)         */
    uint8_t  pkt_rssi, rssi_now;
};
enum { WM_PKT_DONE = 1, WM_PKT_ABORT = 2 };
enum { WM_PKTF_C1 = 1, WM_PKTF_FRAME_B = 2, WM_PKTF_ERR3OF6 = 4, WM_PKTF_CRC_OK = 8 };
#define WM_PKT_MAXBYTES 292u

#endif
