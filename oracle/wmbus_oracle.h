/*
 * wmbus_oracle.h -- CPU restatement of the rtl-wmbus hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, link or call this; the product (rtl-wmbus_amd/) never does.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py) against
 *   - the unmodified reference binary built by oracle/Makefile (oracle/_ref/rtl_wmbus) on the
 *     reference's two bundled captures x its flag matrix and on synthetic T1/C1/S1 captures,
 *   - the reference's own stage functions reached through oracle/ref_probe.c,
 *   - the committed golden lines in tests/golden/ (SURVEY.md Appendix B).
 * The only arithmetic not under /root/reference is glibc 2.35 libm (atan2f via cargf, sqrtf,
 * cosf/sinf); the oracle calls the same libm, exactly like the reference does.
 *
 * Every function in wmbus_oracle.c cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#ifndef WMBUS_ORACLE_H
#define WMBUS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors the reference's command-line switches (rtl_wmbus.c:855-866, 892-967). */
typedef struct wmo_opts {
    unsigned decimation;     /* -d N   (default 2)                      */
    int simultaneous;        /* -s                                      */
    int accurate_atan;       /* 1 unless -a                             */
    int remove_dc;           /* -o                                      */
    int t1c1_enabled;        /* 0 after -p T                            */
    int s1_enabled;          /* 0 after -p S                            */
    int rla_enabled;         /* 0 after -r 0                            */
    int time2_enabled;       /* 0 after -t 0                            */
    int show_algorithm;      /* -v                                      */
    int fixed_timestamp;     /* 1: print the literal "TS" instead of wall-clock time */
    int prefilter;           /* 0: the moving averages main() calls (rtl_wmbus.c:1333-1344);
                                1: the polyphase low-pass the reference defines but never calls
                                   (rtl_wmbus.c:258-294, ppf.h:46-59), d = 2 only, no -s */
    int atan_mode;           /* 0: cargf / pi (what the reference is built with, atan2.h:7-10, rtl_wmbus.c:523-529);
                                1, 2: atan2_approximation / atan2_approximation2 (atan2.h:14-74), the alternatives
                                the reference keeps behind `#elif 0` / `#else` */
} wmo_opts;

/* The two approximations of atan2.h on arrays (for pinning against the reference's own functions). */
void wmo_atan2_approx(int which, const float *im, const float *re, float *out, size_t n);

/* the low-pass filters alone, from a zeroed history (which: 0 = 11 taps, T1/C1; 1 = 46 taps, S1) */
void wmo_fir(int which, const float *x, float *y, size_t n);

void wmo_default_opts(wmo_opts *o);

enum { WMO_CHAIN_T1C1 = 0, WMO_CHAIN_S1 = 1 };
enum { WMO_ALGO_RLA = 0, WMO_ALGO_T2A = 1 };

/* One recovered chip as handed to a packet decoder (t1_c1_packet_decoder.h:649,
 * s1_packet_decoder.h:233): bit0 = data bit, bit1 = access-code flag.  bit2 is ours:
 * "the run-length framer reset itself (and its decoder) since the previous chip". */
typedef struct wmo_chip {
    uint32_t sample;         /* decimated-sample index at which the chip was emitted */
    uint8_t  chain;          /* WMO_CHAIN_*  */
    uint8_t  algo;           /* WMO_ALGO_*   */
    uint8_t  value;          /* bit0 data, bit1 sync flag, bit2 reset-before */
    uint8_t  rssi;           /* (unsigned) filtered magnitude, as passed to the decoder */
} wmo_chip;

/* Optional per-decimated-sample taps; any pointer may be NULL.  Arrays are indexed
 * [chain][m] with `cap` elements per chain. */
typedef struct wmo_taps {
    size_t   cap;
    float   *iq[2];          /* interleaved i,q after boxcar+decimation (2*cap floats) */
    float   *dphi_raw[2];    /* discriminator output                                  */
    float   *dphi[2];        /* after FIR (+ DC removal with -o): the soft symbol      */
    float   *dphi_fir[2];    /* FIR output before the optional DC removal              */
    float   *rssi[2];        /* filtered magnitude (float, before truncation)          */
    float   *clk[2];         /* IIR band-pass output (after gain)                      */
    uint8_t *bit[2];         /* slicer output                                          */
    wmo_chip *chips;         /* chip log, all chains/algos in emission order           */
    size_t   chips_cap, chips_len;
} wmo_taps;

typedef struct wmo_ctx wmo_ctx;

wmo_ctx *wmo_new(const wmo_opts *opts);
void     wmo_free(wmo_ctx *c);
void     wmo_set_taps(wmo_ctx *c, wmo_taps *taps);

/* Consume whole 4096-byte blocks from `cu8` (rtl_wmbus.c:1249,1298-1308); returns the number
 * of bytes consumed (the partial tail is left to the caller, who drops it at EOF). */
size_t   wmo_feed(wmo_ctx *c, const uint8_t *cu8, size_t nbytes);

/* Number of decimated samples produced so far. */
uint64_t wmo_decimated_count(const wmo_ctx *c);

/* Datagram text produced so far ('\n'-terminated lines, reference format).  The buffer is
 * owned by the context; wmo_clear_output() empties it. */
const char *wmo_output(const wmo_ctx *c, size_t *len);
void        wmo_clear_output(wmo_ctx *c);

/* Convenience for timing: process nbytes and return the number of output lines. */
/* Array helpers for the device arithmetic self-test: this host's libm atan2f, IEEE divide and sqrt. */
void wmo_libm_atan2f(const float *y, const float *x, float *out, size_t n);
void wmo_ieee_div(const float *a, const float *b, float *out, size_t n);
void wmo_ieee_sqrt(const float *a, float *out, size_t n);

size_t wmo_run(const wmo_opts *opts, const uint8_t *cu8, size_t nbytes, char **text_out);
void   wmo_free_text(char *text);

#ifdef __cplusplus
}
#endif
#endif
