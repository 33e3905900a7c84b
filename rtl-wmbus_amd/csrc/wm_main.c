/*
 * wm_main.c -- `rtl_wmbus_hip`: drop-in for the reference's command line.
 *
 *   rtl_sdr -f 868.95M -s 1600000 - 2>/dev/null | rtl_wmbus_hip [-o] [-a] [-d N] [-p T|S] [-r 0] [-t 0] [-v] [-s] [-f] [-V]
 *
 * Same switches, usage text layout, exit codes and stdout line format as
 * /root/reference/rtl_wmbus.c:869-967,1217-1372; the per-sample loop (:1298-1357) is replaced by
 * wmbus_stage()/wmbus_process()/wmbus_collect() from libwmbus_hip.so.  Plain C; the GPU is only
 * reached through the C ABI in include/wmbus_hip.h.
 * Extensions (letters the reference does not use): -B bytes per GPU push (default 1 MiB),
 * -G HIP device ordinal.
 */
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "wmbus_hip.h"

#define VERSION "rtl_wmbus_hip 0.1 (MI355X/gfx950 back end)"

static void print_usage(const char *prog)
{
    fprintf(stdout, "rtl_wmbus: " VERSION "\n\n");
    fprintf(stdout, "Usage %s:\n", prog);
    fprintf(stdout, "\t-o remove DC offset\n");
    fprintf(stdout, "\t-a accelerate (use an inaccurate atan version)\n");
    fprintf(stdout, "\t-r 0 to disable run length algorithm\n");
    fprintf(stdout, "\t-t 0 to disable time2 algorithm\n");
    fprintf(stdout, "\t-d 2 set decimation rate to 2 (defaults to 2 if omitted)\n");
    fprintf(stdout, "\t-v show used algorithm in the output\n");
    fprintf(stdout, "\t-V show version\n");
    fprintf(stdout, "\t-s receive S1 and T1/C1 datagrams simultaneously. rtl_sdr _MUST_ be set to 868.625MHz (-f 868.625M)\n");
    fprintf(stdout, "\t-p [T,S] to disable processing T1/C1 or S1 mode\n");
    fprintf(stdout, "\t-f exit if flow of incoming data stops\n");
    fprintf(stdout, "\t-B bytes per GPU push (multiple of 4096, default 1048576)\n");
    fprintf(stdout, "\t-G HIP device ordinal (default 0)\n");
    fprintf(stdout, "\t-P polyphase low-pass (ppf.h) instead of the moving average before decimation (1.6 MS/s, -d 2, no -s)\n");
    fprintf(stdout, "\t-h print this help\n");
}

static void on_alarm(int signo)
{
    (void)signo;
    static const char msg[] = "rtl_wmbus: exiting since incoming data stopped flowing!\n";
    if (write(2, msg, sizeof msg - 1) < 0) _exit(EXIT_FAILURE);
    _exit(EXIT_FAILURE);
}

static int flush_push(wmbus_ctx *ctx, const unsigned char *buf, size_t n)
{
    int rc = wmbus_stage(ctx, 0, buf, n);
    if (!rc) rc = wmbus_process(ctx, n);
    if (!rc) rc = wmbus_collect(ctx);
    if (rc) { fprintf(stderr, "rtl_wmbus_hip: %s\n", wmbus_last_error(ctx)); return rc; }
    size_t len = 0;
    const char *text = wmbus_lines_text(ctx, &len);
    if (len) { fwrite(text, 1, len, stdout); fflush(stdout); }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc == 1 && isatty(0)) { print_usage(argv[0]); return 0; }

    wmbus_cfg cfg;
    wmbus_default_cfg(&cfg);
    cfg.max_push_bytes = 1u << 20;
    int check_flow = 0, opt;
    while ((opt = getopt(argc, argv, "ofad:p:r:vVst:B:G:P")) != -1) {
        switch (opt) {
        case 'o': cfg.remove_dc = 1; break;
        case 'f': check_flow = 1; break;
        case 'a': cfg.accurate_atan = 0; break;
        case 'p':
            if (!strcmp(optarg, "T") || !strcmp(optarg, "t")) cfg.t1c1_enabled = 0;
            else if (!strcmp(optarg, "S") || !strcmp(optarg, "s")) cfg.s1_enabled = 0;
            else { print_usage(argv[0]); return EXIT_FAILURE; }
            break;
        case 'r': if (strcmp(optarg, "0")) { print_usage(argv[0]); return EXIT_FAILURE; } cfg.rla_enabled = 0; break;
        case 't': if (strcmp(optarg, "0")) { print_usage(argv[0]); return EXIT_FAILURE; } cfg.time2_enabled = 0; break;
        case 'd': cfg.decimation = (unsigned)strtoul(optarg, NULL, 10); break;
        case 's': cfg.simultaneous = 1; break;
        case 'v': cfg.show_algorithm = 1; break;
        case 'V': fprintf(stdout, "rtl_wmbus: " VERSION "\n"); return EXIT_SUCCESS;
        case 'B': cfg.max_push_bytes = (size_t)strtoull(optarg, NULL, 10) / WMBUS_BLOCK_BYTES * WMBUS_BLOCK_BYTES; break;
        case 'G': cfg.device = atoi(optarg); break;
        case 'P': cfg.prefilter = WMBUS_PREFILTER_POLYPHASE; break;
        default: print_usage(argv[0]); return EXIT_FAILURE;
        }
    }
    if (getenv("WMBUS_FIXED_TS")) cfg.fixed_timestamp = 1;

    if (check_flow) {
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_handler = on_alarm;
        sigemptyset(&sa.sa_mask);
        fprintf(stderr, "rtl_wmbus: monitoring flow\n");
        sigaction(SIGALRM, &sa, NULL);
    }

    wmbus_ctx *ctx = NULL;
    int rc = wmbus_open(&cfg, &ctx);
    if (rc) {
        fprintf(stderr, "rtl_wmbus_hip: cannot open GPU back end: %s\n", ctx ? wmbus_last_error(ctx) : "out of memory");
        wmbus_close(ctx);
        return EXIT_FAILURE;
    }

    unsigned char *buf = malloc(cfg.max_push_bytes);
    size_t fill = 0;
    if (!buf) return EXIT_FAILURE;
    for (;;) {
        if (check_flow) alarm(2);
        const size_t got = fread(buf + fill, WMBUS_BLOCK_BYTES, 1, stdin);   /* whole blocks only */
        if (check_flow) alarm(0);
        if (got != 1) break;                                                 /* EOF: partial tail dropped */
        fill += WMBUS_BLOCK_BYTES;
        if (fill == cfg.max_push_bytes) { if (flush_push(ctx, buf, fill)) { rc = 1; break; } fill = 0; }
    }
    if (!rc && fill) rc = flush_push(ctx, buf, fill) ? 1 : 0;
    free(buf);
    wmbus_close(ctx);
    return rc ? EXIT_FAILURE : EXIT_SUCCESS;
}
