/*
 * wmbus_hip.h -- C ABI of libwmbus_hip.so, the MI355X (gfx950) back end for the rtl-wmbus
 * cu8 -> datagram hot path.
 *
 * The reference (xaelsouth/rtl-wmbus) has no library interface: its boundary is the process
 * (stdin cu8, argv switches, stdout lines) and, inside it, the per-sample loop of main()
 * (/root/reference/rtl_wmbus.c:1298-1357) that drives t1_c1_signal_chain (:1038-1116) and
 * s1_signal_chain (:1130-1208), which in turn call the packet decoders
 * (t1_c1_packet_decoder.h:649-712, s1_packet_decoder.h:233-282).  This header is the seam a
 * maintainer would bind instead of that loop; each entry point names what it replaces.
 * INTEGRATION.md shows the replacement main() and the ctypes binding.
 *
 * Plain C, plain pointers and sizes, no C++/torch types.  All functions return 0 on success or
 * a negative WMBUS_E* code; nothing in the library calls exit().  One context serves
 * `n_streams` independent captures that advance in lock step (same byte count per push).
 */
#ifndef WMBUS_HIP_H
#define WMBUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WMBUS_BLOCK_BYTES 4096u   /* the reference consumes stdin in 4096-byte units (rtl_wmbus.c:1249,1301) */

enum {
    WMBUS_OK = 0,
    WMBUS_EINVAL = -1,      /* bad argument / configuration                         */
    WMBUS_ENOMEM = -2,      /* host or device allocation failed                     */
    WMBUS_EDEVICE = -3,     /* HIP runtime error (see wmbus_last_error)             */
    WMBUS_EOVERFLOW = -4,   /* (no longer returned: exhausted chip / burst storage is a wmbus_timing.warnings bit) */
    WMBUS_ENODEVICE = -5    /* no HIP device: this library has no CPU fallback      */
};

enum { WMBUS_CHAIN_T1C1 = 0, WMBUS_CHAIN_S1 = 1 };
enum { WMBUS_ALGO_RLA = 0, WMBUS_ALGO_T2A = 1 };

/* Mirrors the reference's switches (rtl_wmbus.c:855-866, parsed at :892-967). */
typedef struct wmbus_cfg {
    unsigned decimation;        /* -d N, default 2                                   */
    int simultaneous;           /* -s                                                */
    int accurate_atan;          /* 1 unless -a                                       */
    int remove_dc;              /* -o                                                */
    int t1c1_enabled;           /* 0 after -p T                                      */
    int s1_enabled;             /* 0 after -p S                                      */
    int rla_enabled;            /* 0 after -r 0                                      */
    int time2_enabled;          /* 0 after -t 0                                      */
    int show_algorithm;         /* -v                                                */
    int fixed_timestamp;        /* 1: print "TS" instead of the wall clock (tests)   */
    /* batch geometry */
    unsigned n_streams;         /* independent captures processed per push           */
    int device;                 /* HIP device ordinal                                */
    size_t max_push_bytes;      /* capacity per stream per push, multiple of 4096    */
    /* tuning (0 = default) */
    unsigned seg_len;           /* clock-recovery time segment, decimated samples (power of two, 1024 ... 2^20); 0: 32768 -- 65536 for
                                   batches of >= 64 captures at decimation 2 with exact arithmetic and no -s, 16384 for batches of
                                   < 64 captures whose pushes are 2^15 ... 2^19 decimated samples (a live stream); with
                                   clock_waves = 1: 32768 / 16384 / 8192 by batch size */
    unsigned rla_seg_len;       /* run-length framer time segment                    */
    unsigned warmup_t1c1;       /* IIR warm-up before a segment, T1/C1 chain         */
    unsigned warmup_s1;         /* IIR warm-up before a segment, S1 chain            */
    unsigned rla_lookback;      /* speculative run-length lookback                   */
    unsigned host_threads;      /* host decoder threads, 0 = auto                    */
    int keep_taps;              /* 1: wmbus_read_tap may be used (the debug views of the last push).  0: no views -- the context then
                                   computes the RSSI only for the samples a packet decoder reads (the default switches at
                                   decimation 2 ... 5): same datagrams, fewer instructions */
    /* Low-pass in front of the decimator.  BOXCAR = the moving averages the reference's main()
     * runs (rtl_wmbus.c:1333-1344): what every reference binary computes, bit for bit.
     * POLYPHASE = lp_ppf_butter_1600kHz_160kHz_200kHz (rtl_wmbus.c:258-294 over ppf.h:46-59), which
     * the reference defines but never calls: an extension, 1.6 MS/s (decimation 2, no -s) only. */
    int prefilter;
    /* Discriminator arctangent: WMBUS_ATAN_LIBM = cargf / pi, what the reference is built with (atan2.h:7-10,
     * rtl_wmbus.c:523-529); WMBUS_ATAN_APPROX1 / 2 = atan2_approximation / atan2_approximation2 (atan2.h:14-74),
     * the alternatives its source keeps behind `#elif 0` / `#else`: extensions, ignored with -a. */
    int atan_mode;
    unsigned spill_words;       /* tuning: run-length chip spill arena, 32-bit words (0 = default) */
    /* Downstream conveniences, OFF by default (the drop-in prints exactly what the reference prints).  The two
     * framers work on every burst, so a clean telegram is printed twice (README.md:105-108): dedup_twins drops the
     * later of two lines of one capture and mode with the same payload from different framers that complete within one
     * longest-telegram time; only_crc_ok prints CRC-clean telegrams only (what e.g. wmbusmeters keeps). */
    int dedup_twins, only_crc_ok;
    unsigned input_windows;     /* 1 (default, also for 0): one device input window per stream; 2: two, used alternately, so
                                   that wmbus_stage() for the next push may run while the previous one is in flight (the
                                   copies run on their own HIP stream).  wmbus_device_input() names the window the next
                                   wmbus_process() will read. */
    /* 0 (default): every soft symbol, RSSI byte and chip is bit-identical to the reference's.  1: TOLERANCE MODE for the
     * default switches (both chains, cargf arctangent, no -P): the discriminator uses a polynomial arctangent and the two
     * FIR low-passes fused multiply-adds, so soft symbols agree with the reference's to 2e-6 (absolute; they lie in
     * [-1, 1]) instead of bit for bit; RSSI, clock recovery and framers stay exact on those symbols.  A telegram whose
     * decision hangs on the last bits of a soft symbol may decode differently (DESIGN_HISTORY.md section 12 and DESIGN.md section 6 count them). */
    int tolerance_mode;
    /* Tuning and test knobs, 0 = the library's default.  (Rounds 1-4 read these from WMBUS_* environment variables inside the
     * push path; as configuration two contexts of one process can differ, and nothing in a push calls getenv.) */
    unsigned rounds_on_host;    /* 1: no hand-off rounds are enqueued unattended: every hand-off failure is finished by the host-driven
                                   path of wmbus_collect (wmbus_timing.slow_path); the GPU test suites run once with it */
    unsigned rssi_full;         /* 1: the RSSI of every sample in the demodulation kernel even without debug views (A/B of RSSI on demand) */
    unsigned rssi_dense_pm;     /* RSSI on demand pauses for 16 pushes when more than this many per mille of a push's tiles were listed
                                   (0: 200; configs[2] lists 310 and is faster on the full pass) */
    unsigned bursts_to_host;    /* 1: every burst travels to the host packet decoders as chips (round 1's path) instead of being decoded
                                   on the GPU where it lies inside the push */
    unsigned burst_caps[4];     /* tests: burst storage {headers, chip words, packets, bytes} (0: sized from the batch), to reach the
                                   WMBUS_WARN_BURSTS_DROPPED paths */
    unsigned k1_small_tile;     /* 1: the first pass of the demodulation kernel on 976-sample tiles even where the 2000-sample tile of
                                   512 threads is available (decimation 2, no -s, RSSI on demand); A/B */
    unsigned k1_tiles_per_block;/* consecutive tiles a block of that first pass takes, each tile's input loaded while the one before
                                   is computed (0: the default; 1: one tile per block, no prefetch) */
    unsigned clock_waves;       /* how a clock-recovery lane group's filter cascade (iir.h:49-77 behind rtl_wmbus.c:1089-1111) is laid onto
                                   waves: 4 = systolic, the sections on the four waves of a block (round 6; 0: the default); 1 = one wave,
                                   software-pipelined across the sections (rounds 1-5).  Same records in memory, same datagrams; A/B */
} wmbus_cfg;

enum { WMBUS_PREFILTER_BOXCAR = 0, WMBUS_PREFILTER_POLYPHASE = 1 };
enum { WMBUS_ATAN_LIBM = 0, WMBUS_ATAN_APPROX1 = 1, WMBUS_ATAN_APPROX2 = 2 };

typedef struct wmbus_ctx wmbus_ctx;

/* One datagram line as the reference prints it (t1_c1_packet_decoder.h:671-699,
 * s1_packet_decoder.h:248-269), plus the keys that define its place in stdout. */
typedef struct wmbus_line {
    uint32_t stream;
    uint8_t  chain, algo, crc_ok, pad;
    uint64_t sample;            /* global decimated-sample index of the completing chip */
    uint32_t text_off, text_len;/* into the buffer returned by wmbus_lines_text()        */
} wmbus_line;

/* Per-push timings measured with HIP events on the library's own stream (ms). */
typedef struct wmbus_timing {
    float demod_ms;             /* front end + discriminator + FIR + RSSI kernel      */
    float clock_ms;             /* IIR clock recovery + time2 framer (incl. re-runs)  */
    float rla_ms;               /* run-length framer (incl. re-runs)                  */
    float gather_ms;            /* burst extraction (and, without debug views, the RSSI of the tiles the bursts touch) */
    float d2h_ms;               /* burst copy to pinned host memory                   */
    float gpu_total_ms;         /* first kernel start -> last copy done               */
    float host_decode_ms;       /* packet decoders + formatting (wall clock)          */
    unsigned clock_reruns, rla_reruns, ema_retries;
    uint64_t chips[2][2];       /* chips produced in this push, [chain][algo]         */
    uint64_t bursts;            /* candidate bursts handed to the host decoders       */
    float turn_wait_ms;         /* host time spent waiting for this process's turn in the demodulation kernel */
    unsigned warnings;          /* WMBUS_WARN_* of this push (the push succeeded)     */
    unsigned slow_path;         /* 1: hand-off verification needed more rounds than run on the device unattended, or an
                                   RSSI value computed on demand could not be proven and the push took the full pass */
    float rssi_ms;              /* RSSI on demand: the launch over the listed tiles (part of gather_ms); 0 otherwise */
    unsigned rssi_mode;         /* WMBUS_RSSI_*: how this push got its RSSI */
    unsigned rssi_tiles;        /* RSSI on demand: (tile, capture) pairs listed in this push */
    unsigned clock_round[4], rla_round[4];   /* segments re-run in the unattended rounds, round by round (beyond the rounds enqueued: 0) */
} wmbus_timing;

/* wmbus_timing.rssi_mode.  EVERY_SAMPLE: in the demodulation kernel (contexts with debug views, option kernels, cfg.rssi_full).
 * ON_DEMAND: only where a packet decoder reads it.  PAUSED: an on-demand context whose bursts
 * cover most of its tiles takes the full pass for sixteen pushes.  FELL_BACK: a value read could not be proven, the push was
 * finished by the full pass (slow_path is set too). */
enum { WMBUS_RSSI_EVERY_SAMPLE = 0, WMBUS_RSSI_ON_DEMAND = 1, WMBUS_RSSI_PAUSED = 2, WMBUS_RSSI_FELL_BACK = 3 };

/* The reference never gives up on an input (rtl_wmbus.c:729-803 has no bound); neither does a push.  When an
 * interferer makes the run-length framer emit more chips than even the spill arena holds, or more candidate bursts
 * than the burst arena, the excess is dropped (datagrams inside it may be lost), all carried state stays exact. */
enum { WMBUS_WARN_CHIPS_DROPPED = 1, WMBUS_WARN_BURSTS_DROPPED = 2 };

/* Process-wide HIP runtime defaults this library wants, applied through the environment: GPU_MAX_HW_QUEUES=16 unless the
 * caller has set the variable (ROCm maps HIP streams onto 4 hardware queues by default and streams that share a queue
 * serialise: a batch of eight contexts runs at 55 % of its rate on four).  The runtime reads the variable ONCE, at the
 * first HIP call of the process, so this must run before that -- wmbus_batch_open() calls it, the CLI calls it first thing
 * in main(); an application that uses HIP before it opens a batch calls it (or exports the variable) itself.  The library
 * does nothing at load time.  WMBUS_KEEP_HW_QUEUES=1 in the environment turns the call into a no-op.  Not thread-safe
 * against concurrent getenv/setenv (it is a setenv): call it before starting threads. */
void wmbus_runtime_init(void);

void wmbus_default_cfg(wmbus_cfg *cfg);

/* Replaces the state set-up at rtl_wmbus.c:1249-1273,1296 (framer/decoder resets, LUT). */
int  wmbus_open(const wmbus_cfg *cfg, wmbus_ctx **out);
void wmbus_close(wmbus_ctx *ctx);
const char *wmbus_last_error(const wmbus_ctx *ctx);

/* Replaces fread() at rtl_wmbus.c:1301: copy `nbytes` (multiple of 4096) of stream `stream`
 * from host memory into the device input window of the next push (asynchronously, on the context's copy stream;
 * wmbus_process orders its kernels behind the copies). */
int  wmbus_stage(wmbus_ctx *ctx, unsigned stream, const uint8_t *cu8, size_t nbytes);
/* Page-locked host memory for wmbus_stage sources (hipHostMalloc): copies from it run at the PCIe
 * rate and asynchronously; any other host pointer works too, through the driver's bounce buffer. */
void *wmbus_alloc_pinned(size_t nbytes);
void  wmbus_free_pinned(void *p);
/* Same for HBM-resident producers: device address of the stream's input window.  With one input window (the default) the
 * window of a push in flight must stay untouched until wmbus_collect has returned: the rare slow paths of a push (a hand-off
 * repair, the full RSSI pass behind an unprovable on-demand value) read it again at collect time.  wmbus_stage enforces that;
 * a producer that writes through this pointer has to keep to it itself, or open the context with cfg.input_windows = 2. */
void *wmbus_device_input(wmbus_ctx *ctx, unsigned stream);

/* Replaces the sample loop rtl_wmbus.c:1310-1356 for `nbytes` staged bytes of every stream:
 * launches the kernels (asynchronously on the context's stream). */
int  wmbus_process(wmbus_ctx *ctx, size_t nbytes);
/* Waits for the push, runs the host packet decoders, makes the lines available.  Lines are in
 * the reference's stdout order within each stream, streams in ascending order. */
int  wmbus_collect(wmbus_ctx *ctx);

size_t wmbus_lines(const wmbus_ctx *ctx, const wmbus_line **lines);
const char *wmbus_lines_text(const wmbus_ctx *ctx, size_t *len);

int  wmbus_get_timing(const wmbus_ctx *ctx, wmbus_timing *t);

/* Debug taps of the last push (requires cfg.keep_taps): what = "dphi" (f32), "rssi" (u8),
 * "bits" (u8 0/1).  Returns the number of elements written, or a negative error. */
long wmbus_read_tap(wmbus_ctx *ctx, const char *what, int chain, unsigned stream,
                    void *dst, size_t max_elems);
/* All chips of the last push for one stream/chain/algo as u32 words
 * [15:8] rssi, [7:0] value (bit0 data, bit1 sync, bit2 framer reset
 * before this chip); `pos` (optional) receives the global decimated-sample index per chip.  A context
 * opened without cfg.keep_taps computes the RSSI only where a packet decoder reads it: the rssi field is 0 then. */
long wmbus_read_chips(wmbus_ctx *ctx, int chain, int algo, unsigned stream,
                      uint32_t *dst, uint64_t *pos, size_t max_elems);

/* Device self-test of the exact scalar arithmetic the kernels use (correctly rounded sqrt and
 * divide, the restated glibc atan2f, the polar discriminator rtl_wmbus.c:517-534) on n operand
 * pairs: o_sqrt = sqrt(|a|), o_div = a/b, o_atan2 = atan2f(a, b),
 * o_disc[i] = discriminator(i=a[i], q=b[i], i'=b[i+1], q'=a[i+1]). */
int  wmbus_selftest_math(int device, const float *a, const float *b, float *o_sqrt, float *o_div,
                         float *o_atan2, float *o_disc, size_t n);

/* Device self-test of the demodulation kernel's two low-pass filters (fir.h:48-72 with the coefficients of
 * rtl_wmbus.c:369-391, lp_fir_butter_800kHz_100kHz_160kHz / _32kHz_36kHz) on one row: x[0 .. n + 48) holds 48 samples of history followed by the n inputs
 * (n a multiple of 4, at most 976: one tile of the kernel); out[0 .. n) receives the 11-tap filter's outputs, out[n .. 2n) the 46-tap one's. */
int  wmbus_selftest_fir(int device, const float *x, float *out, size_t n);

/* Measurement aid: the host half of wmbus_collect (sort, strip / format, merge: wm_decoder.c, replacing the decoders' fprintf at
 * t1_c1_packet_decoder.h:671-699 / s1_packet_decoder.h:248-269) over the records of the context's last push again, `reps` times, without
 * any GPU work.  Returns the lines of one repetition; *seconds receives the wall clock of all of them. */
long wmbus_debug_replay_decode(wmbus_ctx *ctx, unsigned reps, double *seconds);

/* Number of visible HIP devices (0 if none). */
int  wmbus_device_count(void);

/* ---------------------------------------------------------------------------------------------------------------
 * Many captures on one device.  The reference is one stream per process (rtl_wmbus.c:1298-1356 IS its product); a
 * batch is `cfg->n_streams` of those loops side by side.  The library splits the captures over several receiver
 * contexts (default: 8, whole 64-capture groups each), drives every context on its own thread through the same
 * stage / process / collect steps as above, and overlaps them: one context's demodulation kernel runs beside the other
 * contexts' framer kernels and host decoding, and a context's next push is already on the GPU while its previous one
 * is decoded on the host.  This is what `rtl_wmbus_hip FILE...` and bench.py run.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct wmbus_batch wmbus_batch;

typedef struct wmbus_batch_io {
    /* SOURCE.  fill != NULL (needs cfg.input_windows = 2): called on a worker thread whenever streams
     * [first_stream, first_stream + n_streams) need their next push.  Write the bytes of stream s at
     * slab + (s - first_stream) * pitch (page-locked memory, `cap` bytes per stream) and return the byte count EVERY
     * stream of the group advances by (a multiple of 4096, <= cap; pad ended streams with 128 = no signal), or 0 when
     * the group's input has ended.  Calls for different groups may run concurrently.
     * fill == NULL: the input is resident in the device windows (wmbus_batch_stage / wmbus_batch_device_input):
     * every context makes `passes` pushes of `resident_bytes`. */
    size_t (*fill)(void *user, unsigned first_stream, unsigned n_streams, uint8_t *slab, size_t pitch, size_t cap);
    /* SINK (optional).  The lines of one push of one group, in the reference's stdout order per stream
     * (wmbus_line.stream counts within the whole batch); calls are serialised, a push's lines leave as a unit. */
    void (*lines)(void *user, unsigned first_stream, unsigned n_streams, const wmbus_line *lines, size_t n_lines,
                  const char *text, const wmbus_timing *timing);
    void *user;
    size_t resident_bytes;
    unsigned passes;
    /* 1: `fill` stages the group's bytes itself with wmbus_batch_stage() from page-locked memory of its own (slab is
     * NULL then): no host-side copy between the source and the PCIe transfer. */
    unsigned self_staged;
} wmbus_batch_io;

typedef struct wmbus_batch_stats {
    uint64_t samples;           /* input IQ samples consumed, all streams        */
    uint64_t lines;             /* datagram lines produced                       */
    double   seconds;           /* wall clock of wmbus_batch_run                 */
    unsigned pushes;            /* context pushes                                */
    unsigned warnings;          /* WMBUS_WARN_* seen                             */
} wmbus_batch_stats;

/* cfg->n_streams = all captures of the batch; contexts = 0: the default split.  On failure *out still holds an
 * object whose wmbus_batch_last_error() explains; close it. */
int  wmbus_batch_open(const wmbus_cfg *cfg, unsigned contexts, wmbus_batch **out);
/* The split wmbus_batch_open would make, without opening anything (no device needed): returns the number of contexts and
 * writes the captures of each into counts[0 .. min(contexts, cap)). */
unsigned wmbus_batch_plan(const wmbus_cfg *cfg, unsigned contexts, unsigned *counts, unsigned cap);
void wmbus_batch_close(wmbus_batch *b);
const char *wmbus_batch_last_error(const wmbus_batch *b);
unsigned wmbus_batch_contexts(const wmbus_batch *b);
/* Context i and the streams it serves (its own wmbus_lines / wmbus_get_timing / taps describe its last push). */
wmbus_ctx *wmbus_batch_context(wmbus_batch *b, unsigned i, unsigned *first_stream, unsigned *n_streams);
/* wmbus_stage / wmbus_device_input by batch-wide stream number. */
int  wmbus_batch_stage(wmbus_batch *b, unsigned stream, const uint8_t *cu8, size_t nbytes);
void *wmbus_batch_device_input(wmbus_batch *b, unsigned stream);
/* Runs until every group's source has ended (or `passes` pushes per context).  Returns 0 or the first error. */
int  wmbus_batch_run(wmbus_batch *b, const wmbus_batch_io *io, wmbus_batch_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* WMBUS_HIP_H */
