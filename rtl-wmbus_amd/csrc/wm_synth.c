/*
 * wm_synth.c -- deterministic synthetic cu8 capture generator (host C, no GPU).
 *
 * Produces what an RTL-SDR would hand to `rtl_wmbus` on stdin: interleaved unsigned 8-bit I/Q
 * with Gaussian receiver noise and 2-FSK Wireless-M-Bus bursts (EN 13757-4 modes T1, C1 frame
 * A/B, S1).  It is the TX-side counterpart of the receiver: the frame layouts mirror what the
 * reference's decoders accept (t1_c1_packet_decoder.h:165-223, s1_packet_decoder.h:57-96) and
 * the recipe is SURVEY.md section 8(d) / Appendix C.  Used by tests/ (parity inputs) and by
 * bench.py (the 1024-stream workload); it is not on the measured path.
 *
 * RNG: splitmix64-seeded xoshiro256**, Gaussian noise as a sum of four 16-bit uniforms
 * (Irwin-Hall, sigma-matched), so a stream is a pure function of (seed, config).
 */
#include "wm_synth.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct rng { uint64_t s[4]; } rng;

static uint64_t splitmix(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void rng_seed(rng *r, uint64_t seed) { for (int k = 0; k < 4; k++) r->s[k] = splitmix(&seed); }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng *r)
{
    uint64_t *s = r->s;
    const uint64_t res = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return res;
}
static inline double rng_unit(rng *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline unsigned rng_range(rng *r, unsigned lo, unsigned hi) { return lo + (unsigned)(rng_next(r) % (hi - lo + 1)); }
/* ~N(0,1): sum of four U{0..65535}, centred, scaled by 1/sqrt(4*(65536^2-1)/12). */
static inline float rng_gauss(rng *r)
{
    const uint64_t v = rng_next(r);
    const int sum = (int)(v & 0xFFFF) + (int)((v >> 16) & 0xFFFF) + (int)((v >> 32) & 0xFFFF) + (int)(v >> 48);
    return (float)(sum - 131070) * (1.0f / 37837.2f);
}

/* CRC-16 EN 13757 (poly 0x3D65, init 0, MSB first, complemented). */
static uint16_t crc16(const uint8_t *d, size_t n)
{
    uint16_t crc = 0;
    while (n--) {
        crc ^= (uint16_t)(*d++ << 8);
        for (int k = 0; k < 8; k++) crc = (uint16_t)((crc & 0x8000) ? ((crc << 1) ^ 0x3D65) : (crc << 1));
    }
    return (uint16_t)~crc;
}

/* Frame A on-air bytes from the CRC-free telegram t[0..n) (t[0] = L = n-1). */
static size_t frame_a_with_crc(const uint8_t *t, size_t n, uint8_t *out)
{
    size_t o = 0, p = 0;
    size_t blk = n < 10 ? n : 10;
    while (p < n) {
        memcpy(out + o, t + p, blk);
        const uint16_t c = crc16(t + p, blk);
        o += blk; p += blk;
        out[o++] = (uint8_t)(c >> 8); out[o++] = (uint8_t)c;
        blk = n - p < 16 ? n - p : 16;
    }
    return o;
}

/* Frame B (<= 126 data bytes handled): L counts every byte after L including the CRC. */
static size_t frame_b_with_crc(const uint8_t *t, size_t n, uint8_t *out)
{
    memcpy(out, t, n);
    out[0] = (uint8_t)(n - 1 + 2);
    const uint16_t c = crc16(out, n);
    out[n] = (uint8_t)(c >> 8); out[n + 1] = (uint8_t)c;
    return n + 2;
}

typedef struct chipbuf { uint8_t c[8192]; size_t n; } chipbuf;
static void put_bits(chipbuf *b, uint32_t v, int nbits)
{
    for (int k = nbits - 1; k >= 0; k--) if (b->n < sizeof b->c) b->c[b->n++] = (uint8_t)((v >> k) & 1u);
}

static const uint8_t SYM3OF6[16] = {0x16, 0x0D, 0x0E, 0x0B, 0x1C, 0x19, 0x1A, 0x13,
                                    0x2C, 0x25, 0x26, 0x23, 0x34, 0x31, 0x32, 0x29};

static void chips_t1(chipbuf *b, const uint8_t *air, size_t n)
{
    for (int k = 0; k < 19; k++) put_bits(b, 1, 2);       /* 01 x19 */
    put_bits(b, 0x03D, 10);                               /* 0000111101 */
    for (size_t k = 0; k < n; k++) { put_bits(b, SYM3OF6[air[k] >> 4], 6); put_bits(b, SYM3OF6[air[k] & 15], 6); }
    put_bits(b, 0x55, 8);                                 /* postamble */
}
static void chips_c1(chipbuf *b, const uint8_t *air, size_t n, int frame_b)
{
    for (int k = 0; k < 16; k++) put_bits(b, 1, 2);
    put_bits(b, 0x543D, 16);
    put_bits(b, frame_b ? 0x543D : 0x54CD, 16);
    for (size_t k = 0; k < n; k++) put_bits(b, air[k], 8);
    put_bits(b, 0x55, 8);
}
static void chips_s1(chipbuf *b, const uint8_t *air, size_t n)
{
    for (int k = 0; k < 30; k++) put_bits(b, 1, 2);
    put_bits(b, 0x07696, 18);                             /* 000111011010010110 */
    for (size_t k = 0; k < n; k++)
        for (int i = 7; i >= 0; i--) put_bits(b, ((air[k] >> i) & 1) ? 1 : 2, 2);  /* 1->01, 0->10 */
    put_bits(b, 0x55, 8);
}

void wmsynth_default_cfg(wmsynth_cfg *c)
{
    memset(c, 0, sizeof *c);
    c->seed = 0xC0FFEE; c->fs_khz = 1600; c->noise_sigma = 3.0; c->amplitude = 60.0;
    c->frames_per_s = 20.0; c->kinds = WMSYNTH_T1 | WMSYNTH_C1A | WMSYNTH_C1B;
    c->l_min = 10; c->l_max = 60; c->t1c1_center_khz = 0.0; c->s1_center_khz = 0.0;
    c->max_offset_khz = 10.0;
}

size_t wmsynth_generate(const wmsynth_cfg *cfg, uint8_t *out, size_t n_samples,
                        wmsynth_frame *frames, size_t frames_cap)
{
    rng r;
    rng_seed(&r, cfg->seed);
    const double fs = cfg->fs_khz * 1e3;
    const float sigma = (float)cfg->noise_sigma;
    size_t n_frames = 0, pos = 0;
    unsigned seq = 0;
    chipbuf *cb = (chipbuf *)malloc(sizeof *cb);

    while (pos < n_samples) {
        /* exponential gap to the next burst */
        size_t gap = n_samples;
        if (cfg->frames_per_s > 0 && cfg->kinds) {
            const double u = rng_unit(&r);
            const double g = -log(1.0 - u) / cfg->frames_per_s + 0.002;
            gap = (size_t)(g * fs);
        }
        size_t stop = pos + gap < n_samples ? pos + gap : n_samples;
        for (; pos < stop; pos++) {
            const float x = sigma * rng_gauss(&r) + 128.0f, y = sigma * rng_gauss(&r) + 128.0f;
            out[2 * pos] = (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : (int)x);
            out[2 * pos + 1] = (uint8_t)(y < 0 ? 0 : y > 255 ? 255 : (int)y);
        }
        if (pos >= n_samples) break;

        /* pick a burst kind: every 4th burst C1 (A/B alternating) when enabled, S1 when it is
         * the only choice or every 3rd when mixed with T1 */
        unsigned kind = 0;
        {
            const unsigned k = cfg->kinds;
            const int want_c1 = (seq % 4u) == 3u;
            const int want_s1 = (seq % 3u) == 1u;
            if (want_c1 && (k & (WMSYNTH_C1A | WMSYNTH_C1B))) {
                const int b = ((seq / 4u) & 1u) != 0;
                kind = (b && (k & WMSYNTH_C1B)) ? WMSYNTH_C1B : (k & WMSYNTH_C1A) ? WMSYNTH_C1A : WMSYNTH_C1B;
            } else if ((k & WMSYNTH_S1) && (want_s1 || !(k & WMSYNTH_T1))) kind = WMSYNTH_S1;
            else if (k & WMSYNTH_T1) kind = WMSYNTH_T1;
            else if (k & WMSYNTH_S1) kind = WMSYNTH_S1;
            else kind = (k & WMSYNTH_C1A) ? WMSYNTH_C1A : WMSYNTH_C1B;
            seq++;
        }

        /* telegram without CRC: L C M M A A A A V T data... */
        uint8_t tel[256], air[300];
        const unsigned L = rng_range(&r, (unsigned)cfg->l_min, (unsigned)cfg->l_max);
        tel[0] = (uint8_t)L; tel[1] = 0x44;
        for (unsigned k = 2; k <= L; k++) tel[k] = (uint8_t)rng_next(&r);
        const size_t tel_n = (size_t)L + 1;
        size_t air_n;
        cb->n = 0;
        double chip_rate, dev_hz, centre_khz;
        if (kind == WMSYNTH_T1) { air_n = frame_a_with_crc(tel, tel_n, air); chips_t1(cb, air, air_n); chip_rate = 100e3; dev_hz = 50e3; centre_khz = cfg->t1c1_center_khz; }
        else if (kind == WMSYNTH_C1A) { air_n = frame_a_with_crc(tel, tel_n, air); chips_c1(cb, air, air_n, 0); chip_rate = 100e3; dev_hz = 45e3; centre_khz = cfg->t1c1_center_khz; }
        else if (kind == WMSYNTH_C1B) { air_n = frame_b_with_crc(tel, tel_n, air); chips_c1(cb, air, air_n, 1); chip_rate = 100e3; dev_hz = 45e3; centre_khz = cfg->t1c1_center_khz; }
        else { air_n = frame_a_with_crc(tel, tel_n, air); chips_s1(cb, air, air_n); chip_rate = 32768.0; dev_hz = 50e3; centre_khz = cfg->s1_center_khz; }

        const double f_off = centre_khz * 1e3 + (2.0 * rng_unit(&r) - 1.0) * cfg->max_offset_khz * 1e3;
        double phase = 2.0 * M_PI * rng_unit(&r);
        const size_t burst = (size_t)((double)cb->n * fs / chip_rate);
        const float amp = (float)cfg->amplitude;

        if (frames && n_frames < frames_cap) {
            wmsynth_frame *f = &frames[n_frames];
            memset(f, 0, sizeof *f);
            f->kind = kind; f->start_sample = (uint64_t)pos; f->n_samples = (uint32_t)burst;
            f->len = (uint16_t)tel_n;
            memcpy(f->telegram, tel, tel_n);
            /* the receiver prints frame B with L rewritten to the CRC-free length, i.e. the
             * same bytes as tel[] (t1_c1_packet_decoder.h:618,630) */
            f->complete = (uint8_t)(pos + burst <= n_samples);
        }
        n_frames++;

        const double w_c = 2.0 * M_PI * f_off / fs, w_d = 2.0 * M_PI * dev_hz / fs;
        for (size_t k = 0; k < burst && pos < n_samples; k++, pos++) {
            size_t ci = (size_t)((double)k * chip_rate / fs);
            if (ci >= cb->n) ci = cb->n - 1;
            phase += w_c + (cb->c[ci] ? w_d : -w_d);
            if (phase > M_PI) phase -= 2.0 * M_PI; else if (phase < -M_PI) phase += 2.0 * M_PI;
            const float x = amp * (float)cos(phase) + sigma * rng_gauss(&r) + 128.0f;
            const float y = amp * (float)sin(phase) + sigma * rng_gauss(&r) + 128.0f;
            out[2 * pos] = (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : (int)x);
            out[2 * pos + 1] = (uint8_t)(y < 0 ? 0 : y > 255 ? 255 : (int)y);
        }
    }
    free(cb);
    return n_frames;
}
