/*
 * wm_decoder.h -- host-side Wireless-M-Bus packet decoders (plain C).
 *
 * Same observable behaviour as the reference's chip-driven state machines
 * (/root/reference/t1_c1_packet_decoder.h:649-712, s1_packet_decoder.h:233-282): fed one recovered
 * chip at a time (bit0 = data, bit1 = access-code flag) together with the truncated RSSI, they
 * assemble T1 (3-out-of-6), C1 frame A/B (NRZ) and S1 (Manchester) telegrams, check the CRCs,
 * strip them and format the stdout line.  Implemented as a data-driven interpreter over a
 * per-mode step table rather than as function-pointer tables, re-entrant (one object per
 * stream/chain/framer), and without the per-idle-chip memset of the reference.
 */
#ifndef WM_DECODER_H
#define WM_DECODER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { WM_DEC_IDLE = 0, WM_DEC_RECEIVING = 1, WM_DEC_DONE = 2 };
enum { WM_MODE_T1C1 = 0, WM_MODE_S1 = 1 };

typedef struct wm_decoder {
    uint16_t step;           /* index into the mode's step table, 0 = idle                */
    uint8_t  mode;           /* WM_MODE_*                                                 */
    uint8_t  err3of6, c1, frame_b;
    uint16_t l, L;           /* bytes stored / bytes expected (with CRC bytes)            */
    uint32_t sym;            /* chips of the symbol being assembled                       */
    uint32_t mode_bits;      /* first 12 chips after the access code (C1 detection)       */
    uint32_t pkt_rssi;       /* RSSI at the first chip after the access code              */
    uint8_t  packet[292];
} wm_decoder;

void wm_decoder_init(wm_decoder *d, int mode);

/* Feed one chip.  Returns WM_DEC_IDLE / WM_DEC_RECEIVING / WM_DEC_DONE.  After WM_DEC_DONE call
 * wm_decoder_format() (which also returns the decoder to idle). */
int wm_decoder_chip(wm_decoder *d, unsigned chip, unsigned rssi);

/* Abort reception (the run-length framer reset itself, rtl_wmbus.c:636,725). */
static inline void wm_decoder_abort(wm_decoder *d) { d->step = 0; }

/* Chips still needed to finish the telegram in progress, or an upper bound if its length is
 * not known yet; 0 when idle. */
unsigned wm_decoder_chips_owed(const wm_decoder *d);

/* Format "[algo;]MODE;CRC_OK;3OUTOF6OK;TIMESTAMP;PACKET_RSSI;CURRENT_RSSI;IDENT;0xHEX\n"
 * (t1_c1_packet_decoder.h:671-699).  Returns the number of bytes written (no NUL counted). */
size_t wm_decoder_format(wm_decoder *d, const char *algo_tag, const char *timestamp,
                         unsigned rssi_now, char *out, size_t cap, int *crc_ok);

/* The same line from a telegram that arrives already assembled (the GPU's k3_bursts decodes complete bursts: 3-out-of-6
 * / NRZ / Manchester symbols, CRC verdict): packet[0 .. stored bytes) with its CRC bytes, L = expected length with CRC
 * bytes (what the decoder's d->L holds).  Strips the CRC bytes in place and formats (t1_c1_packet_decoder.h:551-636,
 * 671-699). */
size_t wm_packet_format(int mode, int c1, int frame_b, int err3of6, int crc_ok, unsigned L, uint8_t *packet, unsigned pkt_rssi,
                        unsigned rssi_now, const char *algo_tag, const char *timestamp, char *out, size_t cap);
/* CRC verdict of an assembled telegram (t1_c1_packet_decoder.h:471-536). */
int wm_packet_crc_ok(const uint8_t *packet, unsigned L, int frame_b);

/* Twin filter (option cfg.dedup_twins; off by default).  Both framers work on every burst, so a clean telegram is
 * printed twice (README.md:105-108: "You will eventually get two identical datagrams").  state: two records per
 * (capture, chain), zero-initialised.  Returns 1 if this line is the later of two lines with the same payload (text
 * from ";0x" on) from DIFFERENT framers whose completing samples lie within one longest-telegram time: drop it. */
typedef struct wm_twin { uint64_t sample, hash; uint8_t algo, valid; } wm_twin;
int wm_twin_check(wm_twin state[2], int chain, int algo, uint64_t sample, const char *text, size_t len);

/* Wall-clock timestamp in the reference's format (rtl_wmbus_util.h:10-39). */
void wm_timestamp(char *dst, size_t cap);
/* The same for a given instant (seconds and microseconds since the epoch). */
void wm_timestamp_at(char *dst, size_t cap, long sec, long usec);

/* CRC-16 EN 13757 (poly 0x3D65, init 0, final complement), exposed for tests. */
uint16_t wm_crc16(const uint8_t *data, size_t n);

#ifdef __cplusplus
}
#endif
#endif
