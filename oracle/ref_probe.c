/*
 * ref_probe.c -- TEST INFRASTRUCTURE: reach the reference's static stage functions.
 *
 * This file is OUR driver; it textually includes the reference translation unit from where it
 * lies (the Makefile passes -I/root/reference) so that every `static` function and table of
 * rtl_wmbus.c and its headers is callable.  Nothing of the reference is copied here.
 *
 * Modes (argv[1]):
 *   tables              print the decoder tables (3-out-of-6, L->length, CRC) as text
 *   stages <prefix> [a] stdin = cu8 (d = 2, no -s/-o).  Runs the reference's own stage functions
 *                       in the order main()/the signal chains call them (rtl_wmbus.c:1310-1356,
 *                       1047-1111, 1139-1203) and writes <prefix>.{iq,draw,dphi,rssi,clk}{0,1}.f32
 *                       taps (chain 0 = T1/C1, 1 = S1).  Optional flag letters: 'a' = inaccurate atan
 *                       (-a); 'p' = the polyphase low-pass lp_ppf_butter_1600kHz_160kHz_200kHz
 *                       (rtl_wmbus.c:258-294, ppf.h:46-59; defined by the reference but never called
 *                       from its main()) in place of the two moving averages: one (i,q) pair feeds
 *                       both chains.
 *   atan <1|2>          stdin = float pairs (imaginary, real); stdout = the reference's atan2_approximation /
 *                       atan2_approximation2 (atan2.h:14-74) of each pair, as raw floats
 *   fir <0|1>           stdin = raw floats; stdout = the reference's own low-pass (0: lp_fir_butter_800kHz_100kHz_160kHz,
 *                       1: lp_fir_butter_800kHz_32kHz_36kHz, rtl_wmbus.c:369-391 over fir.h:48-72) of the sequence, from a
 *                       zeroed history, as raw floats -- for operands no capture produces (signed zeros, subnormals)
 *   chips <chain> <algo-tag>   stdin = (chip value, rssi) byte pairs -> reference packet decoder;
 *                       datagram lines appear on stdout exactly as the reference prints them.
 */
#define main rtl_wmbus_reference_main
#include "rtl_wmbus.c"
#undef main

static void dump(const char *prefix, const char *name, int chain, const float *v, size_t n)
{
    char path[512];
    snprintf(path, sizeof path, "%s.%s%d.f32", prefix, name, chain);
    FILE *f = fopen(path, "wb");
    fwrite(v, sizeof(float), n, f);
    fclose(f);
}

static int mode_tables(void)
{
    printf("HI");
    for (int i = 0; i < 64; i++) printf(" %u", HIGH_NIBBLE_3OUTOF6[i]);
    printf("\nLO");
    for (int i = 0; i < 64; i++) printf(" %u", LOW_NIBBLE_3OUTOF6[i]);
    printf("\nLEN");
    for (int i = 0; i < 256; i++) printf(" %u", FULL_TLG_LENGTH_FROM_L_FIELD[i]);
    printf("\nCRC");
    for (int i = 0; i < 256; i++) printf(" %u", CRC16_DNP_TABLE[i]);
    printf("\nDEGLITCH_T");
    for (int i = 0; i < 128; i++) printf(" %u", deglitch_filter_t1_c1[i]);
    printf("\nDEGLITCH_S");
    for (int i = 0; i < 16; i++) printf(" %u", deglitch_filter_s1[i]);
    printf("\n");
    return 0;
}

static int mode_stages(const char *prefix, int inaccurate, int use_ppf)
{
    size_t cap = 1 << 20, n = 0;
    uint8_t *in = malloc(cap);
    for (;;) {
        if (n == cap) in = realloc(in, cap *= 2);
        size_t g = fread(in + n, 1, cap - n, stdin);
        if (!g) break;
        n += g;
    }
    n -= n % 4096;
    const size_t M = n / 4;
    float *iq[2], *draw[2], *dphi[2], *rssi[2], *clk[2];
    for (int c = 0; c < 2; c++) {
        iq[c] = calloc(2 * M + 2, 4); draw[c] = calloc(M + 1, 4); dphi[c] = calloc(M + 1, 4);
        rssi[c] = calloc(M + 1, 4); clk[c] = calloc(M + 1, 4);
    }
    size_t m = 0;
    unsigned idx = 0;
    for (size_t k = 0; k < n; k += 2) {
        const float i_unfilt = ((float)(in[k]) - 127.5f), q_unfilt = ((float)(in[k + 1]) - 127.5f);
        float it, qt, is, qs;
        if (use_ppf) {
            it = is = lp_ppf_butter_1600kHz_160kHz_200kHz(i_unfilt, 0);
            qt = qs = lp_ppf_butter_1600kHz_160kHz_200kHz(q_unfilt, 1);
        } else {
            it = moving_average_t1_c1(i_unfilt, 0); qt = moving_average_t1_c1(q_unfilt, 1);
            is = moving_average_s1(i_unfilt, 0); qs = moving_average_s1(q_unfilt, 1);
        }
        if (++idx < 2) continue;
        idx = 0;
        iq[0][2 * m] = it; iq[0][2 * m + 1] = qt; iq[1][2 * m] = is; iq[1][2 * m + 1] = qs;
        draw[0][m] = inaccurate ? polar_discriminator_t1_c1_inaccurate(it, qt) : polar_discriminator_t1_c1(it, qt);
        dphi[0][m] = lp_fir_butter_800kHz_100kHz_160kHz(draw[0][m]);
        rssi[0][m] = rssi_filter_t1_c1(sqrtf(it * it + qt * qt));
        clk[0][m] = bp_iir_cheb1_800kHz_90kHz_98kHz_102kHz_110kHz(dphi[0][m] * dphi[0][m]);
        draw[1][m] = inaccurate ? polar_discriminator_s1_inaccurate(is, qs) : polar_discriminator_s1(is, qs);
        dphi[1][m] = lp_fir_butter_800kHz_32kHz_36kHz(draw[1][m]);
        rssi[1][m] = rssi_filter_s1(sqrtf(is * is + qs * qs));
        clk[1][m] = bp_iir_cheb1_800kHz_22kHz_30kHz_34kHz_42kHz(dphi[1][m] * dphi[1][m]);
        m++;
    }
    for (int c = 0; c < 2; c++) {
        dump(prefix, "iq", c, iq[c], 2 * m); dump(prefix, "draw", c, draw[c], m); dump(prefix, "dphi", c, dphi[c], m);
        dump(prefix, "rssi", c, rssi[c], m); dump(prefix, "clk", c, clk[c], m);
    }
    return 0;
}

static int mode_chips(int chain, const char *tag)
{
    struct t1_c1_packet_decoder_work t;
    struct s1_packet_decoder_work s;
    reset_t1_c1_packet_decoder(&t);
    reset_s1_packet_decoder(&s);
    int v, r;
    while ((v = getchar()) != EOF && (r = getchar()) != EOF) {
        if (v & 4) { reset_t1_c1_packet_decoder(&t); reset_s1_packet_decoder(&s); }
        if (chain == 0) t1_c1_packet_decoder((unsigned)v & 3u, (unsigned)r, &t, tag);
        else s1_packet_decoder((unsigned)v & 3u, (unsigned)r, &s, tag);
    }
    return 0;
}

static int mode_atan(int which)
{
    float p[2];
    while (fread(p, sizeof(float), 2, stdin) == 2) {
        const float complex s = p[1] + p[0] * _Complex_I;
        const float r = which == 1 ? atan2_approximation(s) : atan2_approximation2(s);
        fwrite(&r, sizeof r, 1, stdout);
    }
    return 0;
}

static int mode_fir(int which)
{
    float x;
    while (fread(&x, sizeof x, 1, stdin) == 1) {
        const float y = which == 0 ? lp_fir_butter_800kHz_100kHz_160kHz(x) : lp_fir_butter_800kHz_32kHz_36kHz(x);
        fwrite(&y, sizeof y, 1, stdout);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 3 && !strcmp(argv[1], "fir")) return mode_fir(atoi(argv[2]));
    if (argc >= 3 && !strcmp(argv[1], "atan")) return mode_atan(atoi(argv[2]));
    if (argc >= 2 && !strcmp(argv[1], "tables")) return mode_tables();
    if (argc >= 3 && !strcmp(argv[1], "stages"))
        return mode_stages(argv[2], argc >= 4 && strchr(argv[3], 'a') != NULL, argc >= 4 && strchr(argv[3], 'p') != NULL);
    if (argc >= 4 && !strcmp(argv[1], "chips")) { opts_show_used_algorithm = 1; return mode_chips(atoi(argv[2]), argv[3]); }
    fprintf(stderr, "usage: ref_probe tables | stages <prefix> [a][p] | atan <1|2> | fir <0|1> | chips <chain> <tag>\n");
    return 2;
}
