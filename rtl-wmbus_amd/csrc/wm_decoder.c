/*
 * wm_decoder.c -- host-side packet decoders, see wm_decoder.h.
 *
 * The reference walks a 47-entry (T1/C1) and a 34-entry (S1) table of per-chip handlers
 * (t1_c1_packet_decoder.h:165-223, s1_packet_decoder.h:57-96).  Here a telegram is a sequence of
 * SYMBOL phases; a phase collects `width` chips and then runs one action.  Phase order:
 *
 *   T1/C1:  IDLE -> L_HI(6) -> L_LO(6) -+-> D_HI(6) -> D_LO(6) -> ... -> DONE        (T1, 3-out-of-6)
 *                                       +-> C_TRAILER(4) -> C_L(8) -> C_D(8) -> ... -> DONE   (C1, NRZ)
 *   S1:     IDLE -> S_L(16) -> S_D(16) -> ... -> DONE                                  (Manchester)
 *
 * After every chip that leaves the decoder mid-telegram the RSSI gate of the reference applies
 * (t1_c1_packet_decoder.h:705-710): rssi < 5 drops the telegram.
 */
#include "wm_decoder.h"

#include <stdio.h>
#include <string.h>
#include <sys/time.h>
#include <time.h>

enum { PH_IDLE = 0, PH_L_HI, PH_L_LO, PH_D_HI, PH_D_LO, PH_C_TRAILER, PH_C_L, PH_C_D, PH_S_L, PH_S_D, PH_DONE };
static const uint8_t PHASE_WIDTH[] = {0, 6, 6, 6, 6, 4, 8, 8, 16, 16, 0};

#define STEP(phase, cnt) ((uint16_t)(((phase) << 5) | (cnt)))
#define STEP_PHASE(s) ((s) >> 5)
#define STEP_CNT(s) ((s) & 31u)

#define RSSI_GATE 5u             /* PACKET_CAPTURE_THRESHOLD, t1_c1_packet_decoder.h:35-37 */
#define C1_MODE_A 0x54Cu         /* t1_c1_packet_decoder.h:39-41 */
#define C1_MODE_B 0x543u
#define C1_TRAILER 0xDu

/* EN 13757-4 3-out-of-6: symbol -> nibble, 0xFF invalid (inverse of the nibble -> symbol code;
 * same mapping as t1_c1_packet_decoder.h:50-65). */
static const uint8_t NIBBLE_OF_SYMBOL[64] = {
    0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0x3, 0xFF, 0x1, 0x2, 0xFF,
    0xFF, 0xFF, 0xFF, 0x7, 0xFF, 0xFF, 0x0, 0xFF, 0xFF, 0x5, 0x6, 0xFF, 0x4, 0xFF, 0xFF, 0xFF,
    0xFF, 0xFF, 0xFF, 0xB, 0xFF, 0x9, 0xA, 0xFF, 0xFF, 0xF, 0xFF, 0xFF, 0x8, 0xFF, 0xFF, 0xFF,
    0xFF, 0xD, 0xE, 0xFF, 0xC, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};

/* Telegram length on air for frame format A: L byte + L bytes + 2 CRC bytes per block (block 1
 * holds 9 bytes after L, later blocks 16).  Equals the table at t1_c1_packet_decoder.h:68-96. */
static unsigned frame_a_length(unsigned L) { return 1u + L + 2u * (1u + (L > 9u ? (L + 6u) / 16u : 0u)); }

uint16_t wm_crc16(const uint8_t *data, size_t n)
{
    /* bitwise MSB-first CRC, polynomial 0x3D65 (t1_c1_packet_decoder.h:463-469 uses a table) */
    uint16_t crc = 0;
    while (n--) {
        crc ^= (uint16_t)(*data++ << 8);
        for (int k = 0; k < 8; k++) crc = (uint16_t)((crc & 0x8000u) ? (((unsigned)crc << 1) ^ 0x3D65u) : ((unsigned)crc << 1));
    }
    return (uint16_t)~crc;
}

void wm_decoder_init(wm_decoder *d, int mode)
{
    memset(d, 0, sizeof *d);
    d->mode = (uint8_t)mode;
}

static void store_first_length(wm_decoder *d, unsigned lfield, unsigned total)
{
    d->l = 0;
    d->packet[d->l++] = (uint8_t)lfield;
    d->L = (uint16_t)total;
}

/* Manchester pair -> bit (s1_packet_decoder.h:35-37): 01 -> 1, 10 -> 0, others invalid. */
static int manchester_byte(uint32_t sym16, unsigned *out)
{
    unsigned v = 0;
    for (int k = 7; k >= 0; k--) {
        const unsigned pair = (sym16 >> (2 * k)) & 3u;
        if (pair == 0u || pair == 3u) return 0;
        v = (v << 1) | (pair == 1u);
    }
    *out = v;
    return 1;
}

int wm_decoder_chip(wm_decoder *d, unsigned chip, unsigned rssi)
{
    const unsigned b = chip & 1u;
    unsigned phase = STEP_PHASE(d->step), cnt = STEP_CNT(d->step);

    if (phase == PH_IDLE) {
        if (!(chip & 2u)) return WM_DEC_IDLE;             /* no access code: stay idle          */
        if (rssi < RSSI_GATE) return WM_DEC_IDLE;         /* armed, then dropped by the gate    */
        d->step = STEP(d->mode == WM_MODE_S1 ? PH_S_L : PH_L_HI, 0);
        d->sym = 0;
        return WM_DEC_RECEIVING;
    }

    if (cnt == 0) {
        d->sym = 0;
        if (phase == PH_L_HI || phase == PH_S_L) d->pkt_rssi = rssi;   /* :292-296 */
    }
    d->sym = (d->sym << 1) | b;
    cnt++;

    /* Manchester pairs are validated as soon as they are complete (s1_packet_decoder.h:152-168) */
    if ((phase == PH_S_L || phase == PH_S_D) && (cnt & 1u) == 0) {
        const unsigned pair = d->sym & 3u;
        if (pair == 0u || pair == 3u) { d->step = 0; return WM_DEC_IDLE; }
    }

    if (cnt == PHASE_WIDTH[phase]) {
        cnt = 0;
        switch (phase) {
        case PH_L_HI:                                     /* :298-306 */
            d->mode_bits = d->sym;
            d->err3of6 = d->c1 = d->frame_b = 0;
            d->L = NIBBLE_OF_SYMBOL[d->sym];              /* 0xFF if invalid */
            phase = PH_L_LO;
            break;
        case PH_L_LO: {                                   /* :313-349 */
            d->mode_bits = (d->mode_bits << 6) | d->sym;
            const unsigned lo = NIBBLE_OF_SYMBOL[d->sym];
            if (d->L == 0xFFu || lo == 0xFFu) {
                if (d->mode_bits == C1_MODE_A) { d->frame_b = 0; phase = PH_C_TRAILER; }
                else if (d->mode_bits == C1_MODE_B) { d->frame_b = 1; phase = PH_C_TRAILER; }
                else { d->step = 0; return WM_DEC_IDLE; }
            } else {
                const unsigned lfield = ((unsigned)d->L << 4) | lo;
                store_first_length(d, lfield, frame_a_length(lfield));
                phase = PH_D_HI;
            }
            break;
        }
        case PH_D_HI: {                                   /* :356-366 */
            const unsigned n = NIBBLE_OF_SYMBOL[d->sym];
            if (n == 0xFFu) d->err3of6 = 1;
            d->packet[d->l] = (uint8_t)(n == 0xFFu ? 0xFFu : n << 4);
            phase = PH_D_LO;
            break;
        }
        case PH_D_LO: {                                   /* :373-392 */
            const unsigned n = NIBBLE_OF_SYMBOL[d->sym];
            if (n == 0xFFu) d->err3of6 = 1;
            d->packet[d->l++] |= (uint8_t)n;
            phase = d->l < d->L ? PH_D_HI : PH_DONE;
            break;
        }
        case PH_C_TRAILER:                                /* :399-415 */
            if (d->sym != C1_TRAILER) { d->step = 0; return WM_DEC_IDLE; }
            d->c1 = 1;
            phase = PH_C_L;
            break;
        case PH_C_L:                                      /* :422-438 */
            store_first_length(d, d->sym, d->frame_b ? 1u + d->sym : frame_a_length(d->sym));
            phase = PH_C_D;
            break;
        case PH_C_D:                                      /* :445-460 */
            d->packet[d->l++] = (uint8_t)d->sym;
            phase = d->l < d->L ? PH_C_D : PH_DONE;
            break;
        case PH_S_L: {                                    /* s1_packet_decoder.h:176-197 */
            unsigned v;
            if (!manchester_byte(d->sym, &v)) { d->step = 0; return WM_DEC_IDLE; }
            store_first_length(d, v, frame_a_length(v));
            phase = PH_S_D;
            break;
        }
        case PH_S_D: {                                    /* s1_packet_decoder.h:204-231 */
            unsigned v;
            if (!manchester_byte(d->sym, &v)) { d->step = 0; return WM_DEC_IDLE; }
            d->packet[d->l++] = (uint8_t)v;
            phase = d->l < d->L ? PH_S_D : PH_DONE;
            break;
        }
        default: break;
        }
    }

    d->step = STEP(phase, cnt);
    if (phase == PH_DONE) return WM_DEC_DONE;
    if (rssi < RSSI_GATE) { d->step = 0; return WM_DEC_IDLE; }   /* :705-710 */
    return WM_DEC_RECEIVING;
}

unsigned wm_decoder_chips_owed(const wm_decoder *d)
{
    const unsigned phase = STEP_PHASE(d->step), cnt = STEP_CNT(d->step);
    const unsigned left = d->L > d->l ? (unsigned)(d->L - d->l) : 0u;
    switch (phase) {
    case PH_IDLE: case PH_DONE: return 0;
    case PH_D_HI: return 12u * left - cnt;
    case PH_D_LO: return 12u * left - 6u - cnt;
    case PH_C_D: return 8u * left - cnt;
    case PH_S_D: return 16u * left - cnt;
    case PH_S_L: return 16u * 290u;
    default: return 12u * 290u;
    }
}

/* CRC check per block structure (t1_c1_packet_decoder.h:471-536). */
static int block_ok(const uint8_t *p, unsigned payload)
{
    return wm_crc16(p, payload) == (uint16_t)((p[payload] << 8) | p[payload + 1]);
}

static int crc_ok_blocks(const uint8_t *p, unsigned n, unsigned first, unsigned next)
{
    /* `first`/`next`: block sizes including their two CRC bytes; the last block may be short */
    if (n < 12u) return 0;
    unsigned blk = first;
    while (n) {
        if (blk > n) blk = n;
        if (blk < 2u) return 0;
        if (!block_ok(p, blk - 2u)) return 0;
        p += blk; n -= blk; blk = next;
    }
    return 1;
}

/* Remove the CRC bytes in place (t1_c1_packet_decoder.h:551-636); returns the new length. */
static unsigned strip_crc(uint8_t *p, unsigned n, int frame_b)
{
    if (n < 12u || (frame_b ? p[0] < 2u : p[0] == 0u)) return 0;
    unsigned out = 0, blk = frame_b ? 128u : 12u, src = 0;
    while (src < n) {
        if (blk > n - src) blk = n - src;
        if (blk < 2u) break;
        memmove(p + out, p + src, blk - 2u);
        out += blk - 2u; src += blk;
        if (frame_b) p[0] = (uint8_t)(p[0] - 2u);         /* L counts the CRC bytes in frame B */
        blk = frame_b ? 128u : 18u;
    }
    return out;
}

int wm_packet_crc_ok(const uint8_t *packet, unsigned L, int frame_b)
{
    return frame_b ? crc_ok_blocks(packet, L, 128u, 128u) : crc_ok_blocks(packet, L, 12u, 18u);
}

/* the line's fields, written by hand: the reference's "%s;%u;%u;%s;%u;%u;%08X;0x" (t1_c1_packet_decoder.h:671-699, s1_packet_decoder.h:248-269)
 * through snprintf was 250 of the 350 ns a line costs, and eight GPUs' worth of lines go through one host (tools/host_replay.py) */
static char *put_str(char *w, char *end, const char *s) { while (*s && w < end) *w++ = *s++; return w; }
static char *put_u(char *w, char *end, unsigned v)
{
    char t[10];
    int n = 0;
    do { t[n++] = (char)('0' + v % 10u); v /= 10u; } while (v);
    while (n && w < end) *w++ = t[--n];
    return w;
}

size_t wm_packet_format(int mode, int c1, int frame_b, int err3of6, int crc_ok, unsigned L, uint8_t *packet, unsigned pkt_rssi,
                        unsigned rssi_now, const char *algo_tag, const char *timestamp, char *out, size_t cap)
{
    static const char hexd[] = "0123456789abcdef", HEXD[] = "0123456789ABCDEF";
    if (cap < 64) { if (cap) out[0] = 0; return 0; }
    const uint32_t ident = (uint32_t)packet[4] | ((uint32_t)packet[5] << 8) | ((uint32_t)packet[6] << 16) | ((uint32_t)packet[7] << 24);
    const char *mname = mode == WM_MODE_S1 ? "S1" : c1 ? "C1" : "T1";
    const unsigned ok3 = mode == WM_MODE_S1 ? 1u : (unsigned)(err3of6 ^ 1);
    char *w = out, *end = out + cap - 16;                   /* room for the ident, "0x", the newline and the terminator */
    if (algo_tag) w = put_str(w, end, algo_tag);
    w = put_str(w, end, mname); *w++ = ';';
    *w++ = (char)('0' + (crc_ok ? 1 : 0)); *w++ = ';';
    *w++ = (char)('0' + ok3); *w++ = ';';
    w = put_str(w, end, timestamp); if (w < end) *w++ = ';';
    w = put_u(w, end, pkt_rssi); if (w < end) *w++ = ';';
    w = put_u(w, end, rssi_now); if (w < end) *w++ = ';';
    for (int k = 7; k >= 0; k--) *w++ = HEXD[(ident >> (4 * k)) & 15u];
    *w++ = ';'; *w++ = '0'; *w++ = 'x';
    size_t n = (size_t)(w - out);
    const unsigned len = strip_crc(packet, L, frame_b);
    for (unsigned k = 0; k < len && n + 3 < cap; k++) {
        out[n++] = hexd[packet[k] >> 4];
        out[n++] = hexd[packet[k] & 15u];
    }
    if (n + 1 < cap) out[n++] = '\n';
    out[n] = 0;
    return n;
}

size_t wm_decoder_format(wm_decoder *d, const char *algo_tag, const char *timestamp,
                         unsigned rssi_now, char *out, size_t cap, int *crc_ok)
{
    const int ok = wm_packet_crc_ok(d->packet, d->L, d->frame_b);
    if (crc_ok) *crc_ok = ok;
    const size_t w = wm_packet_format(d->mode, d->c1, d->frame_b, d->err3of6, ok, d->L, d->packet, d->pkt_rssi, rssi_now,
                                      algo_tag, timestamp, out, cap);
    d->step = 0;
    return w;
}

int wm_twin_check(wm_twin state[2], int chain, int algo, uint64_t sample, const char *text, size_t len)
{
    size_t x = len;
    while (x >= 3 && !(text[x - 3] == ';' && text[x - 2] == '0' && text[x - 1] == 'x')) x--;
    x = x >= 3 ? x - 3 : 0;                                   /* from ";0x" on: the payload */
    uint64_t h = 1469598103934665603ull;                     /* FNV-1a */
    for (size_t i = x; i < len; i++) h = (h ^ (uint8_t)text[i]) * 1099511628211ull;
    /* decimated samples (800 kS/s) of the longest telegram: 290 bytes x 12 chips x 8 samples (T1), x 16 x 24 (S1) */
    const uint64_t window = chain ? 16ull * 290 * 24 + 1024 : 12ull * 290 * 8 + 1024;
    int twin = 0;
    for (int k = 0; k < 2; k++)
        if (state[k].valid && state[k].algo != algo && state[k].hash == h && sample - state[k].sample <= window) { twin = 1; state[k].valid = 0; }
    if (twin) return 1;
    const int slot = (!state[0].valid || (state[1].valid && state[0].sample <= state[1].sample)) ? 0 : 1;
    state[slot].sample = sample; state[slot].hash = h; state[slot].algo = (uint8_t)algo; state[slot].valid = 1;
    return 0;
}

void wm_timestamp_at(char *dst, size_t cap, long sec, long usec)
{
    struct tm tmv;
    char fmt[48];
    time_t t = (time_t)sec;
    if (usec < 0) { usec += 1000000; t -= 1; }
    localtime_r(&t, &tmv);
    strftime(fmt, sizeof fmt, "%Y-%m-%d %H:%M:%S.%%06u", &tmv);
    snprintf(dst, cap, fmt, (unsigned)usec);
}

void wm_timestamp(char *dst, size_t cap)
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    wm_timestamp_at(dst, cap, (long)tv.tv_sec, (long)tv.tv_usec);
}
