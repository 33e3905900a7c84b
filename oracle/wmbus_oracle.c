/*
 * wmbus_oracle.c -- plain-C restatement of the rtl-wmbus cu8 -> datagram path.
 *
 * TEST INFRASTRUCTURE ONLY (see wmbus_oracle.h).  Written from the behaviour of the reference,
 * not copied from it: one re-entrant context per stream, integer state indices instead of
 * function-pointer tables, explicit ring buffers.  Each block cites the reference lines it
 * restates (paths relative to /root/reference).
 *
 * Build with -ffp-contract=off and without -march flags: the reference binary rounds every
 * float multiply and add separately (x86-64 baseline, no FMA), and so must this file.
 */
#define _GNU_SOURCE
#include "wmbus_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------
 * Constants
 * ---------------------------------------------------------------------------------------- */

/* rtl_wmbus.c:372 (11-tap) and :384 (46-tap): float arrays initialised from decimal literals. */
static const float FIR_T1C1[11] = {
    -0.00456638213, -0.002571450348, 0.02689425925, 0.1141330398, 0.2264456422, 0.2793297826,
    0.2264456422, 0.1141330398, 0.02689425925, -0.002571450348, -0.00456638213,
};
static const float FIR_S1[46] = {
    -0.000649081282, -0.0009491938209, -0.001361601657, -0.001910785234, -0.002570133495,
    -0.003251218426, -0.003801634695, -0.004012672882, -0.003636803575, -0.002413585945,
    -0.0001013597693, 0.003488892085, 0.008461671287, 0.01481127545, 0.02240598045,
    0.03098477999, 0.0401679839, 0.04948137286, 0.05839197924, 0.06635211627, 0.07284719662,
    0.07744230649, 0.07982251613, 0.07982251613, 0.07744230649, 0.07284719662, 0.06635211627,
    0.05839197924, 0.04948137286, 0.0401679839, 0.03098477999, 0.02240598045, 0.01481127545,
    0.008461671287, 0.003488892085, -0.0001013597693, -0.002413585945, -0.003636803575,
    -0.004012672882, -0.003801634695, -0.003251218426, -0.002570133495, -0.001910785234,
    -0.001361601657, -0.0009491938209, -0.000649081282,
};

/* rtl_wmbus.c:338-341 (T1/C1, ~100 kHz) and :353-356 (S1, ~32.8 kHz): 3 biquads each,
 * b then a, a[0] == 1 is implied. */
static const float IIR_B_T1C1[9] = {1, 1.999994649, 0.9999946492, 1, -1.99999482, 0.9999948196,
                                    1, 1.703868036e-07, -1.000010531};
static const float IIR_A_T1C1[9] = {1, -1.387139203, 0.9921518712, 1, -1.403492665, 0.9845934971,
                                    1, -1.430055639, 0.9923856172};
static const float IIR_B_S1[9] = {1, 1.999994187, 0.9999941867, 1, -1.999994026, 0.9999940262,
                                  1, -1.605750097e-07, -1.000011787};
static const float IIR_A_S1[9] = {1, -1.92151475, 0.9918135499, 1, -1.922481015, 0.984593497,
                                  1, -1.937432099, 0.9927241336};
static const float IIR_GAIN = 1.874981046e-06; /* both filters, rtl_wmbus.c:338,353 */

/* t1_c1_packet_decoder.h:50-65: 6-chip symbol -> nibble, 0xFF = invalid 3-out-of-6 symbol.
 * Generated here from the code's defining property instead of being listed. */
static uint8_t NIB3OF6[64];
/* t1_c1_packet_decoder.h:68-96: L-field -> total telegram length with CRC bytes (frame A). */
static uint16_t FULL_LEN_A[256];
/* t1_c1_packet_decoder.h:99-133: CRC-16 table for polynomial 0x3D65, MSB first. */
static uint16_t CRC_TAB[256];
static int tables_ready;

static void build_tables(void)
{
    if (tables_ready) return;
    /* EN 13757-4 3-out-of-6 code, nibble -> symbol; equals the inverse of both tables at
     * t1_c1_packet_decoder.h:50-65 (checked by tests/test_oracle_units.py via ref_probe). */
    static const uint8_t sym[16] = {0x16, 0x0D, 0x0E, 0x0B, 0x1C, 0x19, 0x1A, 0x13,
                                    0x2C, 0x25, 0x26, 0x23, 0x34, 0x31, 0x32, 0x29};
    memset(NIB3OF6, 0xFF, sizeof NIB3OF6);
    for (int n = 0; n < 16; n++) NIB3OF6[sym[n]] = (uint8_t)n;
    /* Frame A: the L byte + L bytes + 2 CRC bytes per block; block 1 carries 9 bytes after L,
     * every further block up to 16.  Reproduces the table at t1_c1_packet_decoder.h:68-96
     * (checked via ref_probe; note the closed form at :250-253 disagrees with that table
     * from L = 10 on, and the table is what the decoder uses). */
    for (int L = 0; L < 256; L++) FULL_LEN_A[L] = (uint16_t)(1 + L + 2 * (1 + (L > 9 ? (L - 9 + 15) / 16 : 0)));
    for (int v = 0; v < 256; v++) {
        uint16_t r = (uint16_t)(v << 8);
        for (int k = 0; k < 8; k++) r = (uint16_t)((r & 0x8000) ? ((r << 1) ^ 0x3D65) : (r << 1));
        CRC_TAB[v] = r;
    }
    tables_ready = 1;
}

/* ------------------------------------------------------------------------------------------
 * Packet decoders (host state machines)
 * ---------------------------------------------------------------------------------------- */

typedef struct line_sink {
    char *buf;
    size_t len, cap;
    int show_algorithm, fixed_timestamp;
} line_sink;

static void sink_append(line_sink *s, const char *p, size_t n)
{
    if (s->len + n + 1 > s->cap) {
        size_t nc = s->cap ? s->cap * 2 : 4096;
        while (nc < s->len + n + 1) nc *= 2;
        s->buf = (char *)realloc(s->buf, nc);
        s->cap = nc;
    }
    memcpy(s->buf + s->len, p, n);
    s->len += n;
    s->buf[s->len] = 0;
}

/* rtl_wmbus_util.h:10-39: local wall-clock time with microseconds. */
static void make_timestamp(char *dst, size_t n, int fixed)
{
    if (fixed) { snprintf(dst, n, "TS"); return; }
    struct timeval tv;
    struct tm tmv;
    char fmt[64];
    gettimeofday(&tv, NULL);
    localtime_r(&tv.tv_sec, &tmv);
    strftime(fmt, sizeof fmt, "%Y-%m-%d %H:%M:%S.%%06u", &tmv);
    snprintf(dst, n, fmt, (unsigned)tv.tv_usec);
}

/* t1_c1_packet_decoder.h:463-469 */
static uint16_t crc_wmbus(const uint8_t *d, size_t n)
{
    uint16_t crc = 0;
    while (n--) crc = (uint16_t)(CRC_TAB[*d++ ^ (crc >> 8)] ^ (crc << 8));
    return (uint16_t)~crc;
}

static int crc_block_ok(const uint8_t *d, size_t payload)
{
    return crc_wmbus(d, payload) == (uint16_t)((d[payload] << 8) | d[payload + 1]);
}

/* t1_c1_packet_decoder.h:471-506: frame A, blocks of 10+2 then 16+2, short last block. */
static int crc_ok_frame_a(const uint8_t *d, size_t n)
{
    if (n < 12) return 0;
    int ok = crc_block_ok(d, 10);
    d += 12; n -= 12;
    while (ok && n) {
        size_t blk = n >= 18 ? 18 : n;
        if (blk < 2) return 0; /* the reference would read out of bounds here (UB) */
        ok = crc_block_ok(d, blk - 2);
        d += blk; n -= blk;
    }
    return ok;
}

/* t1_c1_packet_decoder.h:508-536: frame B, blocks of 126+2, short last block. */
static int crc_ok_frame_b(const uint8_t *d, size_t n)
{
    int ok = n >= 12;
    while (ok && n) {
        size_t blk = n >= 128 ? 128 : n;
        if (blk < 2) return 0; /* L-field 128 etc.: the reference reads out of bounds (UB) */
        ok = crc_block_ok(d, blk - 2);
        d += blk; n -= blk;
    }
    return ok;
}

/* t1_c1_packet_decoder.h:551-592: drop the CRC bytes in place (frame A layout). */
static unsigned strip_crc_frame_a(uint8_t *d, unsigned n)
{
    unsigned out = 0;
    if (d[0] > 0 && n >= 12) {
        uint8_t *dst = d + 10;
        const uint8_t *src = d + 12;
        out = 10; n -= 12;
        while (n) {
            unsigned blk = n >= 18 ? 18 : n;
            if (blk < 2) break;
            memmove(dst, src, blk - 2);
            dst += blk - 2; out += blk - 2; src += blk; n -= blk;
        }
    }
    return out;
}

/* t1_c1_packet_decoder.h:595-636: frame B; the L byte is rewritten (-2 per block). */
static unsigned strip_crc_frame_b(uint8_t *d, unsigned n)
{
    unsigned out = 0;
    if (d[0] >= 2 && n >= 12) {
        uint8_t *dst = d;
        const uint8_t *src = d;
        while (n) {
            unsigned blk = n >= 128 ? 128 : n;
            if (blk < 2) break;
            memmove(dst, src, blk - 2);
            dst += blk - 2; out += blk - 2; src += blk; n -= blk;
            d[0] = (uint8_t)(d[0] - 2);
        }
    }
    return out;
}

typedef struct pkt_decoder {
    int state;               /* index into the reference's state tables */
    unsigned cur_rssi, pkt_rssi;
    unsigned err3of6, c1, bframe;
    unsigned l, L, mode, byte;
    uint8_t packet[290 + 2];
    char ts[64];
} pkt_decoder;

static void dec_reset(pkt_decoder *d) { memset(d, 0, sizeof *d); }

/* Header + payload line, t1_c1_packet_decoder.h:670-699 / s1_packet_decoder.h:247-269. */
static void emit_line(line_sink *out, pkt_decoder *d, const char *mode, const char *algo,
                      unsigned crc_ok, unsigned ok3of6, unsigned rssi, int frame_b)
{
    char head[160];
    uint32_t ident;
    memcpy(&ident, d->packet + 4, 4); /* t1_c1_packet_decoder.h:638-645, before stripping */
    int n = snprintf(head, sizeof head, "%s%s;%u;%u;%s;%u;%u;%08X;0x", out->show_algorithm ? algo : "",
                     mode, crc_ok, ok3of6, d->ts, d->pkt_rssi, rssi, ident);
    sink_append(out, head, (size_t)n);
    unsigned len = frame_b ? strip_crc_frame_b(d->packet, d->L) : strip_crc_frame_a(d->packet, d->L);
    for (unsigned k = 0; k < len; k++) {
        char hx[3];
        snprintf(hx, sizeof hx, "%02x", d->packet[k]);
        sink_append(out, hx, 2);
    }
    sink_append(out, "\n", 1);
}

/* t1_c1_packet_decoder.h:649-712 driver with the handlers of :272-460 folded into a switch on
 * the state index of the table at :165-223. */
static void t1c1_decoder_chip(pkt_decoder *d, unsigned chip, unsigned rssi, const char *algo,
                              line_sink *out)
{
    const unsigned b = chip & 1u;
    d->cur_rssi = rssi;
    const int st = d->state++;
    switch (st) {
    case 0: /* :272-278 */
        if (!(chip & 2u)) dec_reset(d);
        break;
    case 1: /* :292-296 */
        d->byte = b; d->pkt_rssi = d->cur_rssi;
        break;
    case 6: /* :298-306 */
        d->byte = (d->byte << 1) | b;
        d->mode = d->byte;
        d->L = NIB3OF6[d->byte] == 0xFF ? 0xFFu : (unsigned)NIB3OF6[d->byte] << 4;
        d->err3of6 = d->c1 = d->bframe = 0;
        break;
    case 7: case 13: case 19: case 26: case 30: case 38: /* first bit of a symbol */
        d->byte = b;
        break;
    case 12: { /* :313-349 */
        d->byte = (d->byte << 1) | b;
        d->mode = (d->mode << 6) | d->byte;
        const unsigned lo = NIB3OF6[d->byte];
        if (d->L == 0xFFu || lo == 0xFFu) {
            if (d->mode == 0x54Cu) { d->bframe = 0; d->state = 26; }      /* C1 mode A */
            else if (d->mode == 0x543u) { d->bframe = 1; d->state = 26; } /* C1 mode B */
            else dec_reset(d);
        } else {
            d->bframe = 0; d->c1 = 0;
            d->L |= lo;
            d->l = 0;
            d->packet[d->l++] = (uint8_t)d->L;
            d->L = FULL_LEN_A[d->L];
        }
        break;
    }
    case 18: { /* :356-366 */
        d->byte = (d->byte << 1) | b;
        const unsigned v = NIB3OF6[d->byte];
        if (v == 0xFFu) d->err3of6 = 1;
        d->packet[d->l] = (uint8_t)(v == 0xFFu ? 0xFFu : v << 4);
        break;
    }
    case 24: { /* :373-392 */
        d->byte = (d->byte << 1) | b;
        const unsigned v = NIB3OF6[d->byte];
        if (v == 0xFFu) d->err3of6 = 1;
        d->packet[d->l++] |= (uint8_t)v;
        if (d->l < d->L) d->state = 13;
        else make_timestamp(d->ts, sizeof d->ts, out->fixed_timestamp);
        break;
    }
    case 29: /* :399-415 */
        d->byte = (d->byte << 1) | b;
        d->mode = (d->mode << 4) | d->byte;
        if (d->byte == 0xDu) d->c1 = 1; else dec_reset(d);
        break;
    case 37: /* :422-438 */
        d->byte = (d->byte << 1) | b;
        d->L = d->byte;
        d->l = 0;
        d->packet[d->l++] = (uint8_t)d->L;
        d->L = d->bframe ? 1u + d->L : FULL_LEN_A[d->L];
        break;
    case 45: /* :445-460 */
        d->byte = (d->byte << 1) | b;
        d->packet[d->l++] = (uint8_t)d->byte;
        if (d->l < d->L) d->state = 38;
        else make_timestamp(d->ts, sizeof d->ts, out->fixed_timestamp);
        break;
    default: /* :286-290 plain shift-in */
        d->byte = (d->byte << 1) | b;
        break;
    }

    if (d->state == 0) return;
    if (d->state == 25 || d->state == 46) { /* :659-702 */
        const unsigned ok = d->bframe ? (unsigned)crc_ok_frame_b(d->packet, d->L)
                                      : (unsigned)crc_ok_frame_a(d->packet, d->L);
        emit_line(out, d, d->c1 ? "C1" : "T1", algo, ok, d->err3of6 ^ 1u, rssi, (int)d->bframe);
        dec_reset(d);
    } else if (rssi < 5u) { /* :705-710, PACKET_CAPTURE_THRESHOLD :35-37 */
        dec_reset(d);
    }
}

/* s1_packet_decoder.h:152-168: every second chip closes a Manchester pair ("01"=1, "10"=0). */
static int s1_pair(pkt_decoder *d, unsigned b)
{
    d->byte = (d->byte << 1) | b;
    const unsigned pair = d->byte & 3u;
    if (pair == 0u || pair == 3u) { dec_reset(d); return 0; }
    d->byte = ((d->byte >> 2) << 1) | (pair == 1u ? 1u : 0u);
    return 1;
}

/* s1_packet_decoder.h:233-282 driver, handlers :132-231, state table :57-96. */
static void s1_decoder_chip(pkt_decoder *d, unsigned chip, unsigned rssi, const char *algo,
                            line_sink *out)
{
    const unsigned b = chip & 1u;
    d->cur_rssi = rssi;
    const int st = d->state++;
    if (st == 0) {
        if (!(chip & 2u)) dec_reset(d);
    } else if (st == 1) {
        d->byte = b; d->pkt_rssi = d->cur_rssi;
    } else if (st == 17) {
        d->byte = b;
    } else if (st == 16) { /* :176-197 */
        if (s1_pair(d, b)) {
            d->L = d->byte;
            d->l = 0;
            d->packet[d->l++] = (uint8_t)d->L;
            d->L = FULL_LEN_A[d->L];
        }
    } else if (st == 32) { /* :204-231 */
        if (s1_pair(d, b)) {
            d->packet[d->l++] = (uint8_t)d->byte;
            if (d->l < d->L) d->state = 17;
            else make_timestamp(d->ts, sizeof d->ts, out->fixed_timestamp);
        }
    } else if ((st & 1) == 0) { /* 2,4,...,14 and 18,...,30 */
        (void)s1_pair(d, b);
    } else {
        d->byte = (d->byte << 1) | b;
    }

    if (d->state == 0) return;
    if (d->state == 33) { /* :243-272 */
        const unsigned ok = (unsigned)crc_ok_frame_a(d->packet, d->L);
        emit_line(out, d, "S1", algo, ok, 1u, rssi, 0);
        dec_reset(d);
    } else if (rssi < 5u) { /* :273-281 */
        dec_reset(d);
    }
}

/* ------------------------------------------------------------------------------------------
 * Per-chain DSP + framers
 * ---------------------------------------------------------------------------------------- */

typedef struct chain_state {
    int is_s1;
    /* boxcar (moving_average_filter.h:36-54; rtl_wmbus.c:165-195) */
    int box_len, box_pos, box_sum[2], box_hist[2][16];
    float i, q;              /* latest boxcar outputs */
    /* discriminator memory (rtl_wmbus.c:519/555, 543/579) */
    float pi, pq;
    /* FIR ring (fir.h:37-72) */
    const float *fir_b; int fir_len, fir_pos; float fir_hist[46];
    /* DC removal (rtl_wmbus.c:497-515) */
    float dc_x, dc_y;
    /* RSSI EMA (rtl_wmbus.c:475-495) */
    float ema;
    /* IIR (iir.h:49-77) */
    const float *iir_b, *iir_a; float iir_h[9];
    /* clock lock (rtl_wmbus.c:1043-1044 / 1135-1136) */
    int old_clock_high; unsigned clock_lock;
    /* time2 framer (rtl_wmbus.c:806-852) */
    uint32_t t2_sr; pkt_decoder t2_dec;
    /* run-length framer (rtl_wmbus.c:617-637, 705-726) */
    int rl_run, rl_bitlen, rl_cum, rl_spb[2]; unsigned rl_state; uint32_t rl_raw, rl_sr;
    int rl_reset_pending;    /* ours: feeds bit2 of the chip log */
    pkt_decoder rl_dec;
} chain_state;

/* ppf.h:36-59 + rtl_wmbus.c:258-294: two phases of 12 taps, one ring per phase and component. */
typedef struct ppf_state { float sum; unsigned phase; float hist[2][12]; int pos[2]; } ppf_state;

struct wmo_ctx {
    wmo_opts o;
    ppf_state ppf[2];        /* i, q */
    unsigned dec_idx;        /* rtl_wmbus.c:1255 */
    size_t lut_n, lut_pos;   /* rtl_wmbus.c:974-1010 */
    float *lut_cos, *lut_msin;
    chain_state ch[2];
    uint64_t m;              /* decimated samples so far */
    line_sink out;
    wmo_taps *taps;
};

/* atan2.h:14-41 (y = imaginary, x = real part of s * conj(s_prev)).  `fabs` there is the double function:
 * |y| + 1e-10f is formed in double and rounded to float by the assignment.  Note that the result is
 * NOT divided by pi although the polynomial's coefficients are (the reference's text, kept as it is). */
static float atan2_approx1(float y, float x)
{
    static const float ONEQTR_PI = M_PI / 4.0;
    static const float THRQTR_PI = 3.0 * M_PI / 4.0;
    float r, angle;
    float abs_y = fabs(y) + 1e-10f;
    if (x < 0.0f) { r = (x + abs_y) / (abs_y - x); angle = THRQTR_PI; }
    else { r = (x - abs_y) / (x + abs_y); angle = ONEQTR_PI; }
    angle += (0.1963f * (float)M_1_PI * r * r - 0.9817f * (float)M_1_PI) * r;
    if (y < 0.0f) angle = -angle;
    return angle;
}

/* atan2.h:44-74 */
static float atan2_approx2(float y, float x)
{
    if (x == 0.0f) {
        if (y > 0.0f) return 0.5f;
        if (y == 0.0f) return 0.0f;
        return -0.5f;
    }
    float atan;
    const float z = y / x;
    if (fabs(z) < 1.0f) {
        atan = z / (1.0f * (float)M_PI + 0.28086f * (float)M_PI * z * z);
        if (x < 0.0f) {
            if (y < 0.0f) return atan - 1.0f;
            return atan + 1.0f;
        }
    } else {
        atan = 0.5f - z / (z * z + 0.28086f) * (float)M_1_PI;
        if (y < 0.0f) return atan - 1.0f;
    }
    return atan;
}

void wmo_atan2_approx(int which, const float *im, const float *re, float *out, size_t n)
{
    for (size_t i = 0; i < n; i++) out[i] = which == 1 ? atan2_approx1(im[i], re[i]) : atan2_approx2(im[i], re[i]);
}

void wmo_default_opts(wmo_opts *o)
{
    memset(o, 0, sizeof *o);
    o->decimation = 2; o->accurate_atan = 1; o->t1c1_enabled = 1; o->s1_enabled = 1;
    o->rla_enabled = 1; o->time2_enabled = 1;
}

static void rla_reset(chain_state *c) /* rtl_wmbus.c:628-637, 717-726 */
{
    c->rl_run = 0; c->rl_bitlen = 8 * 256; c->rl_cum = 0; c->rl_state = 0;
    c->rl_raw = 0; c->rl_sr = 0; c->rl_spb[0] = c->rl_spb[1] = 24;
    dec_reset(&c->rl_dec);
    c->rl_reset_pending = 1;
}

static void chain_init(chain_state *c, int is_s1)
{
    memset(c, 0, sizeof *c);
    c->is_s1 = is_s1;
    c->box_len = is_s1 ? 16 : 8;
    c->fir_b = is_s1 ? FIR_S1 : FIR_T1C1;
    c->fir_len = is_s1 ? 46 : 11;
    c->iir_b = is_s1 ? IIR_B_S1 : IIR_B_T1C1;
    c->iir_a = is_s1 ? IIR_A_S1 : IIR_A_T1C1;
    rla_reset(c);
    c->rl_reset_pending = 0;
}

wmo_ctx *wmo_new(const wmo_opts *opts)
{
    build_tables();
    wmo_ctx *c = (wmo_ctx *)calloc(1, sizeof *c);
    c->o = *opts;
    chain_init(&c->ch[0], 0);
    chain_init(&c->ch[1], 1);
    c->out.show_algorithm = opts->show_algorithm;
    c->out.fixed_timestamp = opts->fixed_timestamp;
    /* rtl_wmbus.c:974-993: fs = d*800 kHz, 25 kHz steps, table of cosf / -sinf. */
    const int fs_khz = (int)opts->decimation * 800;
    c->lut_n = (size_t)(fs_khz / 25);
    c->lut_cos = (float *)malloc((c->lut_n + 1) * sizeof(float));
    c->lut_msin = (float *)malloc((c->lut_n + 1) * sizeof(float));
    for (size_t n = 0; n < c->lut_n; n++) {
        const double phi = (2. * M_PI * (25 * (double)n)) / fs_khz;
        c->lut_cos[n] = cosf(phi);
        c->lut_msin[n] = -sinf(phi);
    }
    return c;
}

void wmo_free(wmo_ctx *c)
{
    if (!c) return;
    free(c->lut_cos); free(c->lut_msin); free(c->out.buf); free(c);
}

void wmo_set_taps(wmo_ctx *c, wmo_taps *t) { c->taps = t; }
uint64_t wmo_decimated_count(const wmo_ctx *c) { return c->m; }
const char *wmo_output(const wmo_ctx *c, size_t *len) { if (len) *len = c->out.len; return c->out.buf ? c->out.buf : ""; }
void wmo_clear_output(wmo_ctx *c) { c->out.len = 0; if (c->out.buf) c->out.buf[0] = 0; }

static void log_chip(wmo_ctx *c, int chain, int algo, unsigned value, unsigned rssi)
{
    wmo_taps *t = c->taps;
    if (!t || !t->chips || t->chips_len >= t->chips_cap) return;
    wmo_chip *e = &t->chips[t->chips_len++];
    e->sample = (uint32_t)c->m; e->chain = (uint8_t)chain; e->algo = (uint8_t)algo;
    e->value = (uint8_t)value; e->rssi = (uint8_t)(rssi > 255u ? 255u : rssi);
}

/* moving_average_filter.h:47-54: integer running sum, output (float)sum/len. */
static float boxcar(chain_state *c, int which, int sample)
{
    c->box_sum[which] += sample - c->box_hist[which][c->box_pos];
    c->box_hist[which][c->box_pos] = sample;
    return (float)c->box_sum[which] / (float)c->box_len;
}

/* fir.h:48-72: y = sum_k b[k]*x[n-k], k ascending, accumulated from 0.0f. */
static float fir_step(chain_state *c, float x)
{
    c->fir_hist[c->fir_pos] = x;
    float acc = 0.0f;
    int p = c->fir_pos;
    for (int k = 0; k < c->fir_len; k++) {
        acc += c->fir_b[k] * c->fir_hist[p];
        p = p ? p - 1 : c->fir_len - 1;
    }
    c->fir_pos = c->fir_pos + 1 == c->fir_len ? 0 : c->fir_pos + 1;
    return acc;
}

/* iir.h:49-77: three direct-form-II sections, gain applied once at the end. */
static float iir_step(chain_state *c, float x)
{
    for (int s = 0; s < 3; s++) {
        const float *a = c->iir_a + 3 * s, *b = c->iir_b + 3 * s;
        float *h = c->iir_h + 3 * s;
        h[0] = x - (a[1] * h[1] + a[2] * h[2]);
        x = b[0] * h[0] + b[1] * h[1] + b[2] * h[2];
        h[2] = h[1];
        h[1] = h[0];
    }
    return x * IIR_GAIN;
}

static void deliver(wmo_ctx *c, chain_state *ch, int algo, unsigned chip, unsigned rssi)
{
    const int chain = ch->is_s1;
    unsigned logged = chip;
    pkt_decoder *d = algo == WMO_ALGO_RLA ? &ch->rl_dec : &ch->t2_dec;
    if (algo == WMO_ALGO_RLA && ch->rl_reset_pending) { logged |= 4u; ch->rl_reset_pending = 0; }
    log_chip(c, chain, algo, logged, rssi);
    const char *tag = algo == WMO_ALGO_RLA ? "rla;" : "t2a;";
    if (chain) s1_decoder_chip(d, chip, rssi, tag, &c->out);
    else t1c1_decoder_chip(d, chip, rssi, tag, &c->out);
}

/* rtl_wmbus.c:729-803 */
static void rla_t1c1(wmo_ctx *c, chain_state *ch, unsigned raw_bit, unsigned rssi)
{
    ch->rl_raw = (ch->rl_raw << 1) | raw_bit;
    const unsigned st = (unsigned)(__builtin_popcount(ch->rl_raw & 0x3Fu) >= 3); /* LUT :126-144 */
    if (ch->rl_state == st) { ch->rl_run++; return; }
    if (ch->rl_run < 5) { rla_reset(ch); ch->rl_state = st; ch->rl_run = 1; return; }
    ch->rl_run *= 256;
    const int half = ch->rl_bitlen / 2;
    if (ch->rl_run <= half) { rla_reset(ch); ch->rl_state = st; ch->rl_run = 1; return; }
    int n;
    for (n = 0; ch->rl_run > half; n++) {
        ch->rl_run -= ch->rl_bitlen;
        unsigned chip = ch->rl_state;
        ch->rl_sr = (ch->rl_sr << 1) | chip;
        if ((ch->rl_sr & 0xFFFFu) == 0x543Du) chip |= 2u; /* :97-99, :773 */
        deliver(c, ch, WMO_ALGO_RLA, chip, rssi);
    }
    ch->rl_cum += ch->rl_run;
    ch->rl_bitlen += (ch->rl_run + ch->rl_cum / 16) / (32 * n); /* :792-796 */
    ch->rl_state = st;
    ch->rl_run = 1;
}

/* rtl_wmbus.c:640-702 */
static void rla_s1(wmo_ctx *c, chain_state *ch, unsigned raw_bit, unsigned rssi)
{
    static const uint8_t deglitch[16] = {0, 1, 0, 1, 0, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1}; /* :149-154 */
    ch->rl_raw = (ch->rl_raw << 1) | raw_bit;
    const unsigned st = deglitch[ch->rl_raw & 0xFu];
    if (ch->rl_state == st) { ch->rl_run++; return; }
    const int spb = (ch->rl_spb[0] + ch->rl_spb[1]) / 2;
    if (spb <= 12 || spb >= 36) { rla_reset(ch); ch->rl_state = st; ch->rl_run = 1; return; }
    const int half = spb / 2;
    const int run0 = ch->rl_run;
    if (run0 <= half) { rla_reset(ch); ch->rl_state = st; ch->rl_run = 1; return; }
    int n;
    for (n = 0; ch->rl_run > half; n++) {
        ch->rl_run -= spb;
        unsigned chip = ch->rl_state;
        ch->rl_sr = (ch->rl_sr << 1) | chip;
        if ((ch->rl_sr & 0xFFFFFFu) == 0x547696u) chip |= 2u; /* :101-103, :688 */
        deliver(c, ch, WMO_ALGO_RLA, chip, rssi);
    }
    ch->rl_spb[ch->rl_state] = run0 / n;
    ch->rl_state = st;
    ch->rl_run = 1;
}

/* rtl_wmbus.c:1038-1116 (T1/C1) and :1130-1208 (S1): one decimated sample through a chain. */
static void chain_step(wmo_ctx *c, chain_state *ch)
{
    const int k = ch->is_s1;
    const float i = ch->i, q = ch->q;
    wmo_taps *t = c->taps;
    const size_t m = (size_t)c->m;
    const int tap = t && m < t->cap;

    /* Discriminator: s * conj(s_prev), then cargf/pi (atan2.h:7-10), or with -a only the
     * imaginary part (rtl_wmbus.c:536-551). */
    float dphi_raw;
    if (c->o.accurate_atan) {
        const float cr = ch->pi, ci = -ch->pq;
        const float re = i * cr - q * ci;
        const float im = i * ci + q * cr;
        dphi_raw = c->o.atan_mode == 1 ? atan2_approx1(im, re) : c->o.atan_mode == 2 ? atan2_approx2(im, re) : atan2f(im, re) * (float)M_1_PI;
    } else {
        dphi_raw = ch->pi * q - i * ch->pq;
    }
    ch->pi = i; ch->pq = q;

    float dphi = fir_step(ch, dphi_raw);
    const float dphi_fir = dphi;
    if (c->o.remove_dc) { /* rtl_wmbus.c:497-515, alpha = 0.999f */
        const float alpha = 0.999f;
        const float y = (1.f + alpha) / 2.f * (dphi - ch->dc_x) + alpha * ch->dc_y;
        ch->dc_x = dphi; ch->dc_y = y;
        dphi = y;
    }
    const unsigned bit = dphi >= 0; /* rtl_wmbus.c:1059 / 1151 */

    float mag = sqrtf(i * i + q * q); /* rtl_wmbus.c:1066-1067 / 1158-1159 */
    ch->ema = 0.6789f * mag + (1.0f - 0.6789f) * ch->ema;
    const unsigned rssi = (unsigned)ch->ema; /* float -> unsigned at the framer call */

    if (tap) {
        if (t->iq[k]) { t->iq[k][2 * m] = i; t->iq[k][2 * m + 1] = q; }
        if (t->dphi_raw[k]) t->dphi_raw[k][m] = dphi_raw;
        if (t->dphi[k]) t->dphi[k][m] = dphi;
        if (t->dphi_fir[k]) t->dphi_fir[k][m] = dphi_fir;
        if (t->rssi[k]) t->rssi[k][m] = ch->ema;
        if (t->bit[k]) t->bit[k][m] = (uint8_t)bit;
    }

    if (c->o.rla_enabled) {
        if (k) rla_s1(c, ch, bit, rssi); else rla_t1c1(c, ch, bit, rssi);
    }

    if (c->o.time2_enabled) { /* rtl_wmbus.c:1089-1111 / 1181-1203 */
        const float y = iir_step(ch, dphi * dphi);
        const int high = y >= 0;
        if (tap && t->clk[k]) t->clk[k][m] = y;
        if (high && !ch->old_clock_high) {
            ch->clock_lock = 1;
        } else if (high) {
            if (ch->clock_lock < 2) ch->clock_lock++;
            else if (ch->clock_lock == 2) {
                ch->clock_lock++;
                unsigned chip = bit;
                ch->t2_sr = (ch->t2_sr << 1) | bit; /* :818-828 / :842-852 */
                if (k ? (ch->t2_sr & 0xFFFFFFu) == 0x547696u : (ch->t2_sr & 0xFFFFu) == 0x543Du) chip |= 2u;
                deliver(c, ch, WMO_ALGO_T2A, chip, rssi);
            }
        }
        ch->old_clock_high = high;
    }
}

/* rtl_wmbus.c:262-266: b[PHASES][COEFFS]; filter[..].fir = {b[1] for phase 0, b[0] for phase 1} (:275-276). */
static const float PPF_B[2][12] = {
    {0.000140535927, 0.0001309279731, 0.00551787474, 0.03160167988, 0.08315031015, 0.1295143636, 0.1295143636,
     0.08315031015, 0.03160167988, 0.00551787474, 0.0001309279731, 0.000140535927},
    {1.102280392e-05, 0.001356012537, 0.01499414005, 0.05525973093, 0.1099887688, 0.1366692652, 0.1099887688,
     0.05525973093, 0.01499414005, 0.001356012537, 1.102280392e-05, 0},
};

/* ppf.h:46-59: at phase == max_phase the running sum restarts; every call adds the current phase's
 * FIR (fir.h:48-72: ring history, taps ascending from the newest sample, accumulated from 0). */
static float ppf_step(ppf_state *f, float sample)
{
    if (f->phase == 2) { f->phase = 0; f->sum = 0; }
    const float *b = PPF_B[f->phase == 0 ? 1 : 0];
    float *hist = f->hist[f->phase];
    int *pos = &f->pos[f->phase];
    hist[*pos] = sample;
    float acc = 0;
    int p = *pos;
    for (int k = 0; k < 12; k++) { acc += b[k] * hist[p]; p = p ? p - 1 : 11; }
    *pos = *pos + 1 == 12 ? 0 : *pos + 1;
    f->sum += acc;
    f->phase++;
    return f->sum;
}

/* rtl_wmbus.c:1298-1357: the per-input-sample loop. */
size_t wmo_feed(wmo_ctx *c, const uint8_t *cu8, size_t nbytes)
{
    const size_t whole = nbytes - nbytes % 4096;
    for (size_t p = 0; p < whole; p += 2) {
        const float fi = (float)cu8[p] - 127.5f;       /* :1312-1313 */
        const float fq = (float)cu8[p + 1] - 127.5f;
        float it = fi, qt = fq, is = fi, qs = fq;
        if (c->o.simultaneous) { /* :997-1031 */
            const float x = c->lut_cos[c->lut_pos], z = c->lut_msin[c->lut_pos];
            c->lut_pos += 13;
            if (c->lut_pos >= c->lut_n) c->lut_pos -= c->lut_n;
            const float ix = fi * x, qx = fq * x, iz = fi * z, qz = fq * z;
            it = ix - qz; qt = qx + iz;
            is = ix + qz; qs = qx - iz;
        }
        /* float -> int truncation happens at the boxcar's int parameter (A.1). */
        chain_state *a = &c->ch[0], *b = &c->ch[1];
        if (c->o.prefilter == 1) {                     /* one filtered (i,q) pair feeds both chains */
            a->i = b->i = ppf_step(&c->ppf[0], fi);
            a->q = b->q = ppf_step(&c->ppf[1], fq);
        } else {
        a->i = boxcar(a, 0, (int)it);
        a->q = boxcar(a, 1, (int)qt);
        a->box_pos = a->box_pos + 1 == a->box_len ? 0 : a->box_pos + 1;
        b->i = boxcar(b, 0, (int)is);
        b->q = boxcar(b, 1, (int)qs);
        b->box_pos = b->box_pos + 1 == b->box_len ? 0 : b->box_pos + 1;
        }

        if (++c->dec_idx < c->o.decimation) continue; /* :1350-1352 */
        c->dec_idx = 0;
        if (c->o.t1c1_enabled) chain_step(c, a);      /* :1354 */
        if (c->o.s1_enabled) chain_step(c, b);        /* :1355 */
        c->m++;
    }
    return whole;
}

size_t wmo_run(const wmo_opts *opts, const uint8_t *cu8, size_t nbytes, char **text_out)
{
    wmo_ctx *c = wmo_new(opts);
    wmo_feed(c, cu8, nbytes);
    size_t lines = 0;
    for (size_t k = 0; k < c->out.len; k++) lines += c->out.buf[k] == '\n';
    if (text_out) {
        *text_out = (char *)malloc(c->out.len + 1);
        memcpy(*text_out, c->out.buf ? c->out.buf : "", c->out.len);
        (*text_out)[c->out.len] = 0;
    }
    wmo_free(c);
    return lines;
}

void wmo_free_text(char *t) { free(t); }

/* ---- array helpers for the device arithmetic self-test (tests/test_gpu_parity.py) ------------- */
void wmo_libm_atan2f(const float *y, const float *x, float *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = atan2f(y[i], x[i]); }
void wmo_ieee_div(const float *a, const float *b, float *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = a[i] / b[i]; }
void wmo_ieee_sqrt(const float *a, float *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = sqrtf(a[i]); }

/* The two low-pass filters alone (rtl_wmbus.c:369-391 over fir.h:48-72), from a zeroed history: which = 0 the 11-tap filter
 * of the T1/C1 chain, 1 the 46-tap one of the S1 chain.  For tests with operands no capture produces (signed zeros). */
void wmo_fir(int which, const float *x, float *y, size_t n)
{
    chain_state c;
    memset(&c, 0, sizeof c);
    c.fir_b = which ? FIR_S1 : FIR_T1C1;
    c.fir_len = which ? 46 : 11;
    for (size_t i = 0; i < n; i++) y[i] = fir_step(&c, x[i]);
}
